#!/bin/bash
# HBM traffic and times of the kernels that format the index file on the device (k_fxi_*) and of k_fastq_rows_wg: 2e7 reads
OUT=gpurun_out/r05r
mkdir -p $OUT
export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/$C -o pmc -- python tools/fxi_pmc_probe.py 2e7 > $OUT/$C.json 2> $OUT/$C.err
  for K in k_fxi k_fastq_rows k_sort k_rs_; do FX_PMC_KERNEL=$K python tools/pmc_dump.py $OUT/$C; done > $OUT/pmc_$C.txt
  rm -rf $OUT/$C
done
timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/prof -o trace -- python tools/fxi_pmc_probe.py 2e7 > $OUT/stats.json 2> $OUT/stats.err
DB=$(find $OUT/prof -name '*.db' | head -1)
[ -n "$DB" ] && python tools/rocprof_summary.py $DB $OUT/kernel_stats.txt && grep -E 'k_fxi|k_fastq_rows|k_sort|k_rs_' $OUT/kernel_stats.txt
rm -rf $OUT/prof
cat $OUT/stats.json; wc -l $OUT/pmc_*.txt
