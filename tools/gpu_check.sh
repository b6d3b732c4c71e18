#!/bin/bash
# One GPU-box pass: parity tests, bench line, kernel trace, scale tools.  Outputs under gpurun_out/<tag>/.
TAG=${1:-run}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -x -q ) > $OUT/pytest.log 2>&1
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print(\"smoke ok\")" ) > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
tail -3 $OUT/pytest.log
timeout 600 python bench.py --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
cat $OUT/bench.json
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o trace -- python bench.py --no-pmc --steps 20 --warmup 3 --no-cpu-baseline > $OUT/prof_bench.json 2> $OUT/prof.err
DB=$(find $OUT/prof -name '*.db' | head -1)
[ -n "$DB" ] && python tools/rocprof_summary.py $DB $OUT/kernel_stats.txt && grep 'fx::' $OUT/kernel_stats.txt
find $OUT/prof -name '*.db' -size +20M -delete
timeout 600 python tools/fastq_scale.py 2e7 > $OUT/fastq_scale.json 2> $OUT/fastq_scale.err; cat $OUT/fastq_scale.json
timeout 900 python tools/bgzf_scale.py 3.0 > $OUT/bgzf_scale.json 2> $OUT/bgzf_scale.err; cat $OUT/bgzf_scale.json
timeout 600 python tools/e2e_file.py 3.0 > $OUT/e2e.json 2> $OUT/e2e.err; cat $OUT/e2e.json
timeout 600 python tools/manyrec_scale.py 5e6 300 > $OUT/manyrec.json 2> $OUT/manyrec.err; cat $OUT/manyrec.json
nproc; free -g | head -2
