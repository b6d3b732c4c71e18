#!/bin/bash
# instruction counters of the plain (two-read form) FASTQ build kernels: k_fastq_lines, k_fastq_rows_wg
OUT=gpurun_out/r05l
mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace --output-format csv -d $OUT/sq -o pmc -- python tools/fq_build_bench.py 2e7 > $OUT/sq.json 2> $OUT/sq.err
FX_PMC_KERNEL=k_fastq python tools/pmc_dump.py $OUT/sq | tail -8 > $OUT/sq_fq_plain.txt; cat $OUT/sq_fq_plain.txt
rm -rf $OUT/sq
