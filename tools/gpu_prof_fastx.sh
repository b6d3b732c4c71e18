#!/bin/bash
# rocprofv3 kernel trace of the kseq path (tools/fastx_scale.py: 10 M-read FASTQ stream, 3.1 GB FASTA stream, file iteration),
# once by default and once with every line through the walk.  Outputs under gpurun_out/<tag>/.
TAG=${1:-prof_fastx}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for MODE in default walk_only; do
  if [ $MODE = walk_only ]; then export FX_KSEQ_WALK_ONLY=1; else unset FX_KSEQ_WALK_ONLY; fi
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_$MODE -o trace -- python tools/fastx_scale.py > $OUT/$MODE.json 2> $OUT/$MODE.err
  DB=$(find $OUT/prof_$MODE -name '*.db' | head -1)
  [ -n "$DB" ] && python tools/rocprof_summary.py $DB $OUT/kernel_stats_$MODE.txt && grep 'k_kq\|^kernel' $OUT/kernel_stats_$MODE.txt
  rm -rf $OUT/prof_$MODE
done
