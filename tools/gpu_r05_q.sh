#!/bin/bash
OUT=gpurun_out/r05q
mkdir -p $OUT
export TMPDIR=/tmp
for i in 1 2 3 4; do
timeout 600 python bench.py --no-c4 --no-e2e --no-pmc --gbp 0.1 --steps 2 --warmup 1 --no-c3-file > $OUT/b$i.json 2> $OUT/b$i.err
python -c "
import json;d=json.load(open('$OUT/b$i.json'));fs=d['c3']['file_sample'];print('sample', fs['Fastq_ctor_full_index_s'], fs['ctor_phases_s'].get('scan_s'), fs['ctor_phases_s'].get('scan_laps_s'), 'settle', d['c3'].get('device_memory_settled_after_s'))"
done
