#!/bin/bash
# PMC passes for the gather kernels (tools/fetch_probe.py).  Outputs under gpurun_out/<tag>/.
TAG=${1:-pmc_fetch}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
run() {  # name, counters...
  n=$1; shift
  timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$n -o pmc -- python tools/fetch_probe.py 3.0 2e7 > $OUT/$n.json 2> $OUT/$n.err
  FX_PMC_KERNEL=fetch python tools/pmc_dump.py $OUT/$n > $OUT/${n}_fetch.txt
  find $OUT/$n -name '*.csv' -size +2M -delete
}
run sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
run sq2 SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_LEVEL_VMEM
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum
for n in sq sq2 fetch write tcc; do echo "== $n"; awk 'NR%5==1' $OUT/${n}_fetch.txt | head -8; done
