#!/bin/bash
# PMC passes for the BGZF kernels (tools/bgzf_decode_probe.py opens the C4 file a few times).  Outputs under gpurun_out/<tag>/.
TAG=${1:-pmc_bgzf}
GBP=${2:-1.0}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace --output-format csv -d $OUT/sq -o pmc -- python tools/bgzf_decode_probe.py $GBP > $OUT/sq.json 2> $OUT/sq.err
FX_PMC_KERNEL=k_bgzf python tools/pmc_dump.py $OUT/sq > $OUT/sq_k_bgzf.txt; tail -4 $OUT/sq_k_bgzf.txt
timeout 600 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $OUT/sq2 -o pmc -- python tools/bgzf_decode_probe.py $GBP > $OUT/sq2.json 2> $OUT/sq2.err
FX_PMC_KERNEL=k_bgzf python tools/pmc_dump.py $OUT/sq2 > $OUT/sq2_k_bgzf.txt; tail -4 $OUT/sq2_k_bgzf.txt
timeout 600 rocprofv3 --pmc SQ_IFETCH SQ_INST_LEVEL_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --kernel-trace --output-format csv -d $OUT/sq3 -o pmc -- python tools/bgzf_decode_probe.py $GBP > $OUT/sq3.json 2> $OUT/sq3.err
FX_PMC_KERNEL=k_bgzf python tools/pmc_dump.py $OUT/sq3 > $OUT/sq3_k_bgzf.txt; tail -4 $OUT/sq3_k_bgzf.txt; tail -3 $OUT/sq3.err
find $OUT -name '*.csv' -size +2M -delete
