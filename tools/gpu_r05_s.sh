#!/bin/bash
# the compiled reference on the WHOLE configs[2] file beside the product, at the last commit of the round
OUT=gpurun_out/r05s
mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 330 python bench.py --no-c4 --no-e2e --no-pmc --no-cpu-baseline --steps 2 --warmup 1 --gbp 0.1 --c3-reference-full > $OUT/full.json 2> $OUT/full.err ) 2> $OUT/full.time
cat $OUT/full.time; tail -3 $OUT/full.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r05s/full.json"))
e=d["c3"]["e2e_full"]
print({k:(v if not isinstance(v,(dict,list)) else "...") for k,v in e.items()})
print(e.get("reference_on_the_whole_file"))
PY
