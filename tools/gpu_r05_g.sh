#!/bin/bash
# the driver's multi-GPU command shape on ONE GPU: 8 ranks over gloo sharing the device, full default sizes (weak 8 x 3 Gbp + strong + fastq_strong)
OUT=gpurun_out/r05g
mkdir -p $OUT
export TMPDIR=/tmp
( time FX_BENCH_BACKEND=gloo timeout 1500 python bench.py --gpus 8 > $OUT/bench_8ranks_gloo.json 2> $OUT/bench_8ranks_gloo.err ) 2> $OUT/bench_8ranks.time
cat $OUT/bench_8ranks.time; cut -c1-3000 $OUT/bench_8ranks_gloo.json; tail -5 $OUT/bench_8ranks_gloo.err
rocm-smi --showmeminfo vram 2>/dev/null | head -8
