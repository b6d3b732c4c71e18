#!/bin/bash
OUT=gpurun_out/r05n
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_fxi_dev.py -x -q > $OUT/pytest.log 2>&1; tail -2 $OUT/pytest.log
C3_REPS=3 timeout 600 python tools/c3_phases.py 1e8 > $OUT/c3_phases.json 2> $OUT/c3_phases.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/r05n/c3_phases.json'))
for r in d['ctor_runs']:
    ip=r['index_phases']
    print(r['mode'], r['Fastq_ctor_s'], 'staging', r['build_phases']['staging_s'], 'fxi', r['build_phases']['fxi_s'], 'write_call', round(ip['write_call'],3), 'laps', round(sum(ip[k] for k in ('table_shape','table_kernels','table_to_file','file_grown','index_shape','index_kernels','index_to_file','host_levels_and_header')),3), 'sort', round(ip['name_sort'],3))
PY
