#!/bin/bash
OUT=gpurun_out/r05p
mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 900 python bench.py --no-pmc > $OUT/bench.json 2> $OUT/bench.err ) 2> $OUT/bench.time; head -2 $OUT/bench.time
python -c "
import json;d=json.load(open('$OUT/bench.json'));e=d['c3']['e2e_full'];print(e['Fastq_ctor_s'], e['constructor_runs']); print(e['phases_s']); print(d['c3']['file_sample']['Fastq_ctor_full_index_s'], d['c3']['file_sample']['ctor_phases_s'])"
C3_REPS=2 timeout 600 python tools/c3_phases.py 1e8 > $OUT/c3_phases_bind.json 2> $OUT/c3_phases.err
FX_FXI_NO_BIND=1 C3_REPS=2 timeout 600 python tools/c3_phases.py 1e8 > $OUT/c3_phases_nobind.json 2>> $OUT/c3_phases.err
python - <<'PY'
import json
for t in ('bind','nobind'):
    d=json.load(open('gpurun_out/r05p/c3_phases_%s.json'%t))
    for r in d['ctor_runs']:
        ip=r['index_phases']
        print(t, r['mode'], r['Fastq_ctor_s'], 'staging', r['build_phases']['staging_s'], 'to_file', round(ip['table_to_file']+ip['index_to_file'],3), 'grown', round(ip['file_grown'],3))
PY
