#!/bin/bash
OUT=gpurun_out/r05j
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_api.py tests/test_gpu_vs_reference.py tests/test_gpu_fastx.py -x -q -k "fastq or Fastq or fq or fastx or Fastx" > $OUT/pytest_fq.log 2>&1; tail -3 $OUT/pytest_fq.log
python tools/fq_one_probe.py 1e8 > $OUT/fq_one.json 2> $OUT/fq_one.err; cat $OUT/fq_one.json
FX_FQ_CRLF=1 python tools/fq_one_probe.py 1e8 > $OUT/fq_one_crlf_kernel.json 2> $OUT/fq_one_crlf_kernel.err; cat $OUT/fq_one_crlf_kernel.json
python tools/fastx_scale.py > $OUT/fastx_scale.json 2> $OUT/fastx_scale.err; cut -c1-1500 $OUT/fastx_scale.json
