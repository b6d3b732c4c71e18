#!/bin/bash
# round 5 evidence: parity tests, smoke, the default bench line (timed), the same command under rocprofv3 --kernel-trace --stats
TAG=${1:-r05f}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $OUT/pytest.log 2>&1
tail -4 $OUT/pytest.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print(\"smoke ok\")" ) > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
( time timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err ) 2> $OUT/bench.time; cat $OUT/bench.time
cut -c1-600 $OUT/bench.json
( time timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/prof -o trace -- python bench.py --no-pmc --no-cpu-baseline > $OUT/prof_bench.json 2> $OUT/prof.err ) 2> $OUT/prof.time; cat $OUT/prof.time
DB=$(find $OUT/prof -name '*.db' | head -1)
[ -n "$DB" ] && python tools/rocprof_summary.py $DB $OUT/kernel_stats.txt && grep 'fx::' $OUT/kernel_stats.txt | head -70
rm -rf $OUT/prof
