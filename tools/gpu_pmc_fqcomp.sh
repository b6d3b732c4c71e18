#!/bin/bash
# PMC passes for k_fastq_comp (tools/fq_comp_probe.py): HBM traffic, L2 and L1 requests.  Outputs under gpurun_out/<tag>/.
TAG=${1:-pmc_fqcomp}
N=${2:-1e7}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
i=0
for SET in "FETCH_SIZE TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_ACCESSES_sum TA_BUSY_avr" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/p$i -o pmc -- python tools/fq_comp_probe.py $N > $OUT/p$i.json 2> $OUT/p$i.err
  FX_PMC_KERNEL=k_fastq_comp python tools/pmc_dump.py $OUT/p$i | tail -2 | tee $OUT/p${i}_fqcomp.txt
done
find $OUT -name '*.csv' -size +2M -delete
