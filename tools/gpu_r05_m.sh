#!/bin/bash
OUT=gpurun_out/r05m
mkdir -p $OUT
export TMPDIR=/tmp
python tools/first_open_probe.py make 1e8 > $OUT/make.json 2> $OUT/make.err; cat $OUT/make.json
FX_TRACE_ALLOC=1 python tools/first_open_probe.py open > $OUT/open_pool.json 2> $OUT/open_pool.err; cat $OUT/open_pool.json; grep fxgpu $OUT/open_pool.err | head -8
rm -f /dev/shm/fx_first_open.fq
timeout 600 python -m pytest tests/test_gpu_windows.py tests/test_gpu_api.py -x -q > $OUT/pytest.log 2>&1; tail -2 $OUT/pytest.log
