#!/bin/bash
# The round's validation pass on the GPU box: parity tests, smoke, the default bench line (timed), the kernel trace of the same command.
TAG=${1:-round}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests -m gpu -x -q ) > $OUT/pytest.log 2>&1
tail -5 $OUT/pytest.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print(\"smoke ok\")" ) > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
( time timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err ) 2> $OUT/bench.time; cat $OUT/bench.time
cat $OUT/bench.json
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o trace -- python bench.py --no-pmc --no-cpu-baseline > $OUT/prof_bench.json 2> $OUT/prof.err
DB=$(find $OUT/prof -name '*.db' | head -1)
[ -n "$DB" ] && python tools/rocprof_summary.py $DB $OUT/kernel_stats.txt && grep 'fx::' $OUT/kernel_stats.txt
find $OUT/prof -name '*.db' -size +20M -delete
# HBM traffic and instruction counters of the FASTQ build kernels (one-read and two-read forms)
bash tools/gpu_pmc_fq_one.sh $TAG/pmc_fq_traffic 2e7 > $OUT/pmc_fq_traffic.log 2>&1
bash tools/gpu_pmc_fq_one_sq.sh $TAG/pmc_fq_sq 2e7 > $OUT/pmc_fq_sq.log 2>&1; tail -24 $OUT/pmc_fq_sq.log
python tools/bgzf_group_probe.py 3.0 128 256 > $OUT/bgzf_group_probe.json 2> $OUT/bgzf_group_probe.err; cat $OUT/bgzf_group_probe.json
python tools/fetch_many_breakdown.py > $OUT/fetch_many_breakdown.json 2> $OUT/fetch_many_breakdown.err; tail -3 $OUT/fetch_many_breakdown.json
python tools/getter_breakdown.py > $OUT/getter_breakdown.json 2> $OUT/getter_breakdown.err; tail -2 $OUT/getter_breakdown.json | cut -c1-600
python tools/iter_rate.py > $OUT/iter_rate.json 2> $OUT/iter_rate.err; tail -2 $OUT/iter_rate.json | cut -c1-600
