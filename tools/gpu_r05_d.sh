#!/bin/bash
OUT=gpurun_out/r05d
mkdir -p $OUT
export TMPDIR=/tmp
python tools/fq_build_bench.py 1e8 > $OUT/fq_rows_wg64.json 2> $OUT/fq_rows_wg64.err; cat $OUT/fq_rows_wg64.json
timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -k "fastq" > $OUT/pytest_fq.log 2>&1; tail -2 $OUT/pytest_fq.log
