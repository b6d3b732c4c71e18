#!/bin/bash
OUT=gpurun_out/r05d
mkdir -p $OUT
export TMPDIR=/tmp
python tools/fq_build_bench.py 1e8 > $OUT/fq_rows_wg.json 2> $OUT/fq_rows_wg.err; cat $OUT/fq_rows_wg.json
FX_FQ_ROWS_NT=1 python tools/fq_build_bench.py 1e8 > $OUT/fq_rows_wg_nt.json 2> $OUT/fq_rows_wg_nt.err; cat $OUT/fq_rows_wg_nt.json
