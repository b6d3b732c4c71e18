#!/bin/bash
# round 5 evidence, part 1: instruction counters of the FASTQ build kernels (one-read and two-read), the phases of Fastq(path)
# at 1e8 reads with and without the early room, the per-object rates from ONE run
OUT=gpurun_out/r05k
mkdir -p $OUT
export TMPDIR=/tmp
bash tools/gpu_pmc_fq_one_sq.sh r05k/pmc_fq_sq 2e7 > $OUT/pmc_fq_sq.log 2>&1; tail -30 $OUT/pmc_fq_sq.log
C3_REPS=2 timeout 600 python tools/c3_phases.py 1e8 > $OUT/c3_phases.json 2> $OUT/c3_phases.err; cut -c1-2500 $OUT/c3_phases.json
timeout 600 python tools/iter_rate.py > $OUT/iter_rate.json 2> $OUT/iter_rate.err; tail -2 $OUT/iter_rate.json | cut -c1-1500
