#!/bin/bash
# round 5 validation pass: parity tests, smoke, the default bench line (timed)
TAG=${1:-r05v}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $OUT/pytest.log 2>&1
tail -5 $OUT/pytest.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print(\"smoke ok\")" ) > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
( time timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err ) 2> $OUT/bench.time; cat $OUT/bench.time
cat $OUT/bench.json
