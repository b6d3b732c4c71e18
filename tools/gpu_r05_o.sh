#!/bin/bash
OUT=gpurun_out/r05o
mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_api.py tests/test_gpu_vs_reference.py -x -q -k "fastq or Fastq or fq" > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
python tools/fq_one_probe.py 1e8 > $OUT/fq_one.json 2> $OUT/fq_one.err; cat $OUT/fq_one.json
timeout 600 python bench.py --no-c4 --no-e2e --no-pmc --no-cpu-baseline --steps 2 --warmup 1 --gbp 0.1 > $OUT/bench.json 2> $OUT/bench.err
python -c "
import json;d=json.load(open('$OUT/bench.json'));e=d['c3']['e2e_full'];print(e['Fastq_ctor_s'], e['constructor_runs']); print(e['phases_s']); f=d['c3']['full']; print(f['index_build_ms'], f['full_index_one_read_ms'], f['kernels_one_read_ms_avg'], f['roofline_issue']['k_fastq_lines_comp']['floor_over_measured'])"
