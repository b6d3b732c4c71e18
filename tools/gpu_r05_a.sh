#!/bin/bash
# round 5, first pass on the GPU box: where Fastq(path) of a large file spends its time, and how fast bytes get into a tmpfs file
OUT=gpurun_out/r05a
mkdir -p $OUT
export TMPDIR=/tmp
nproc > $OUT/host.txt; free -g >> $OUT/host.txt; df -h /dev/shm /tmp >> $OUT/host.txt; uname -r >> $OUT/host.txt
gcc -O2 -o /tmp/fwp tools/filewrite_probe.c -lpthread
for m in "0 16 0" "0 16 1" "0 8 0" "0 8 1" "1 1 0" "1 16 0" "1 16 1" "0 32 1" "1 4 0" "0 4 0"; do set -- $m; /tmp/fwp /dev/shm/fwp.bin 8192 $2 $1 $3; done > $OUT/filewrite.txt 2>&1
cat $OUT/filewrite.txt
timeout 600 python tools/c3_phases.py 5e7 > $OUT/c3_phases.json 2> $OUT/c3_phases.err; cat $OUT/c3_phases.json; tail -3 $OUT/c3_phases.err
