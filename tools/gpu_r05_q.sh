#!/bin/bash
OUT=gpurun_out/r05q
mkdir -p $OUT
export TMPDIR=/tmp
for i in 1 2 3; do
( time timeout 900 python bench.py --no-pmc > $OUT/bench$i.json 2> $OUT/bench$i.err ) 2> $OUT/bench$i.time; head -2 $OUT/bench$i.time | tail -1
python -c "
import json;d=json.load(open('$OUT/bench$i.json'));e=d['c3']['e2e_full'];fs=d['c3']['file_sample'];print('ctor', e['Fastq_ctor_s'], [r['Fastq_ctor_s'] for r in e['constructor_runs']], 'sample', fs['Fastq_ctor_full_index_s'], fs['ctor_phases_s'].get('scan_laps_s',{}).get('count_pass_enqueue'), 'settle', d.get('device_memory_settled_after_s'), d['c3'].get('device_memory_settled_after_s'), 'e2e', d['e2e']['fxi_durable_s'], d.get('speedup_vs_cpu'))"
done
