python -m pytest tests/test_gpu_kernels.py tests/test_gpu_vs_reference.py tests/test_gpu_api.py tests/test_gpu_windows.py tests/test_gpu_shards.py -x -q -k "fastq or Fastq" 2>&1 | tail -3
python tools/fastq_scale.py 1e8 2>&1 | tail -1 | cut -c1-330
