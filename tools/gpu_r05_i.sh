#!/bin/bash
OUT=gpurun_out/r05i
mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -k "one_read or line_records" > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
FX_TRACE=1 FX_TRACE_ALLOC=1 timeout 600 python tools/c3_outlier_probe.py 1e8 2e6 full > $OUT/outlier.json 2> $OUT/outlier.err; python -c "import json;d=json.load(open('$OUT/outlier.json'));print(d['full_runs'])"; grep "open plain\|scratch" $OUT/outlier.err | tail -12
