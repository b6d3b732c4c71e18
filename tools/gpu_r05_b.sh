#!/bin/bash
# round 5: the device-formatted .fxi -- parity tests, then the phases of Fastq(path) at 1e8 reads
OUT=gpurun_out/r05b
mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_fxi_dev.py -x -q ) > $OUT/pytest_fxi.log 2>&1; tail -15 $OUT/pytest_fxi.log
FX_TRACE=1 timeout 600 python tools/c3_phases.py 1e8 > $OUT/c3_phases_1e8.json 2> $OUT/c3_phases_1e8.err; cat $OUT/c3_phases_1e8.json; grep fxgpu $OUT/c3_phases_1e8.err | tail -4
