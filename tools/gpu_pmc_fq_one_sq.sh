#!/bin/bash
# instruction counters of the FASTQ build kernels, one-read and two-read (tools/fq_one_probe.py).  Outputs under gpurun_out/<tag>/.
TAG=${1:-pmc_fq_sq}
N=${2:-2e7}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace --output-format csv -d $OUT/sq -o pmc -- python tools/fq_one_probe.py $N > $OUT/sq.json 2> $OUT/sq.err
FX_PMC_KERNEL=k_fastq python tools/pmc_dump.py $OUT/sq | tail -12 > $OUT/sq_fq.txt; cat $OUT/sq_fq.txt
timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_WAIT_ANY --kernel-trace --output-format csv -d $OUT/sq2 -o pmc -- python tools/fq_one_probe.py $N > $OUT/sq2.json 2> $OUT/sq2.err
FX_PMC_KERNEL=k_fastq python tools/pmc_dump.py $OUT/sq2 | tail -12 > $OUT/sq2_fq.txt; cat $OUT/sq2_fq.txt
find $OUT -name '*.csv' -size +2M -delete
