#!/bin/bash
OUT=gpurun_out/r05e
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_api.py tests/test_gpu_vs_reference.py tests/test_gpu_shards.py -x -q -k "fastq or Fastq or fq" > $OUT/pytest_fq.log 2>&1; tail -12 $OUT/pytest_fq.log
python tools/fastq_scale.py 2e7 > $OUT/fastq_scale.json 2> $OUT/fastq_scale.err; cut -c1-900 $OUT/fastq_scale.json; tail -2 $OUT/fastq_scale.err
