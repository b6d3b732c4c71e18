#!/bin/bash
# PMC passes for the composition kernels (they run in bench.py's verification leg, outside the timed step).
# Outputs under gpurun_out/<tag>/; tools/pmc_dump.py prints the per-dispatch counters.
TAG=${1:-pmc_comp}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace --output-format csv -d $OUT/sq -o pmc -- python bench.py --no-pmc --steps 2 --warmup 1 --no-cpu-baseline > $OUT/sq.json 2> $OUT/sq.err
python tools/pmc_dump.py $OUT/sq > $OUT/sq_k_fasta_comp.txt; tail -4 $OUT/sq_k_fasta_comp.txt
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/$C -o pmc -- python bench.py --no-pmc --steps 2 --warmup 1 --no-cpu-baseline > $OUT/$C.json 2> $OUT/$C.err
  python tools/pmc_dump.py $OUT/$C > $OUT/${C}_k_fasta_comp.txt; tail -2 $OUT/${C}_k_fasta_comp.txt
done
find $OUT -name '*.csv' -size +2M -delete
