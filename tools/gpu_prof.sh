#!/bin/bash
# The driver's bench line, the same command under rocprofv3 --kernel-trace --stats, and the PMC traffic passes of the
# scan kernel (counters in runs of their own, kernel trace only).  Outputs under gpurun_out/<tag>/.
TAG=${1:-prof}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 900 python bench.py ) > $OUT/bench.json 2> $OUT/bench.err
cat $OUT/bench.json
( time timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/prof -o trace -- python bench.py --no-pmc --no-c3-file ) > $OUT/prof_bench.json 2> $OUT/prof.err
DB=$(find $OUT/prof -name '*.db' | head -1)
[ -n "$DB" ] && python tools/rocprof_summary.py $DB $OUT/kernel_stats.txt && grep 'fx::' $OUT/kernel_stats.txt
rm -rf $OUT/prof
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$C -o pmc -- python bench.py --no-pmc --steps 5 --warmup 1 --no-cpu-baseline --no-verify --no-e2e --no-c3 --no-c4 > $OUT/pmc_$C.json 2> $OUT/pmc_$C.err
done
python tools/pmc_summary.py $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE "k_span_scan<0>" 3050025703 $OUT/pmc_k_span_scan.json | head -20
find $OUT -name '*.csv' -size +5M -delete
