#!/bin/bash
# the compiled reference on the WHOLE configs[2] file (VERDICT r4 missing #4): a small rehearsal first, then 1e8 reads
OUT=gpurun_out/r05h
mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python bench.py --no-c4 --no-e2e --no-pmc --no-cpu-baseline --steps 2 --warmup 1 --gbp 0.1 --c3-reads 3e6 --c3-sample 1e6 --c3-reference-full > $OUT/small.json 2> $OUT/small.err || { tail -5 $OUT/small.err; exit 1; }
python -c "import json;d=json.load(open('$OUT/small.json'));print(json.dumps(d['c3']['e2e_full']['reference_on_the_whole_file']))"
( time timeout 1500 python bench.py --no-c4 --no-e2e --no-pmc --no-cpu-baseline --steps 2 --warmup 1 --gbp 0.1 --c3-reference-full > $OUT/full.json 2> $OUT/full.err ) 2> $OUT/full.time
cat $OUT/full.time; tail -3 $OUT/full.err
python -c "import json;d=json.load(open('$OUT/full.json'));print(json.dumps(d['c3']['e2e_full']))"
