#!/bin/bash
# HBM traffic of the one-read FASTQ build (k_fastq_lines_comp) and of the two-read form: FETCH_SIZE / WRITE_SIZE in separate passes
# over tools/fq_one_probe.py.  Outputs under gpurun_out/<tag>/.
TAG=${1:-pmc_fq_one}
N=${2:-2e7}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/$C -o pmc -- python tools/fq_one_probe.py $N > $OUT/$C.json 2> $OUT/$C.err
done
NB=$(python -c "print(int(float('$N')) * 348)")
python tools/pmc_summary.py $OUT/FETCH_SIZE $OUT/WRITE_SIZE fx::k_fastq_lines_comp $NB $OUT/traffic_one_read.json | grep -v '^  "fx::k_\(rs\|sort\|cnt\)' | head -60
find $OUT -name '*.csv' -size +2M -delete
