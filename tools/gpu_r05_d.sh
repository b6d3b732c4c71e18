#!/bin/bash
OUT=gpurun_out/r05d
mkdir -p $OUT
export TMPDIR=/tmp
for G in 2048 4096 1024 100000000; do FX_FQ_ROWS_GRID=$G python tools/fq_build_bench.py 1e8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('grid $G', d['build_ms'], d['kernels_ms_avg']['k_fastq_rows'], d['kernels_ms_avg']['k_fastq_lines'], d['rows_equal_truth'])"; done
timeout 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_shards.py -x -q -k "fastq" > $OUT/pytest_fq.log 2>&1; tail -2 $OUT/pytest_fq.log
