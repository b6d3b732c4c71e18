#!/bin/bash
OUT=gpurun_out/r05c
mkdir -p $OUT
gcc -O2 -o /tmp/fwp2 tools/filewrite_probe2.c -lpthread 2>/dev/null
for cfg in "A 16" "A 8" "A 4" "B 16" "B 8" "B 32" "F 16" "F 8" "F 32" "F 4" "E 4" "E 8" "E 16" "E 2" "P 1" "P 2" "P 4" "F 16 2048" "F 16 8192 32" "F 16 8192 512" "B 16 2048" "E 4 2048"; do set -- $cfg; /tmp/fwp2 /dev/shm/fwp2.bin 10240 $1 $2 ${3:-8192} ${4:-128}; done > $OUT/filewrite2.txt 2>&1
cat $OUT/filewrite2.txt
