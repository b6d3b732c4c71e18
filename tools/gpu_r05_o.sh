#!/bin/bash
OUT=gpurun_out/r05o
mkdir -p $OUT
export TMPDIR=/tmp
FX_TRACE_ALLOC=1 FX_TRACE=1 timeout 600 python bench.py --no-c4 --no-e2e --no-pmc --no-cpu-baseline --steps 2 --warmup 1 --gbp 0.1 > $OUT/bench.json 2> $OUT/bench.err
grep -n "scratch\|open plain\|fxi" $OUT/bench.err | tail -40 | cut -c1-220
python -c "
import json;d=json.load(open('$OUT/bench.json'));e=d['c3']['e2e_full'];print(e['Fastq_ctor_s'], e['first_constructor_of_the_process'], e['phases_s'])"
timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -k "one_read or misleads" > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
