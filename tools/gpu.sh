#!/bin/bash
# ONE script for every pass on the GPU box (round 6; replaces the twenty gpu_r05_*.sh one-offs):
#   gpurun -- 'bash tools/gpu.sh <tag> <step> [<step> ...]'          outputs under gpurun_out/<tag>/
# steps (any order, any subset; a step's arguments follow it as step:arg1:arg2):
#   pytest:<file>[:<file>...]  some test files (-m gpu)
#   tests          pytest -m gpu                           smoke       __graft_entry__.smoke()
#   bench          the driver's default bench line, timed  prof        the same command under rocprofv3 --kernel-trace --stats
#   bench_c3full   the bench with the reference on the WHOLE configs[2] file (10^8 reads: ~3 minutes of one core)
#   bench_c4full   the bench with ALL 1 M queries of the C4 leg answered by the reference too (builder-run evidence)
#   bench8         the driver's --gpus 8 command shape on this ONE device (gloo; all ranks share it)
#   counters       rocprofv3 -L (the counter names of this box)
#   valuprobe      tools/valuprobe.hip  (cycles per wave64 VALU instruction)
#   firsttouch[:GB] tools/firsttouch.hip (the first host-to-device copy into fresh device memory, with and without a touching kernel)
#   gathercal      tools/gathercal.hip under FETCH_SIZE and the raw request counters (calibration of the gather traffic)
#   bgzf_pmc[:gbp] counter passes over the BGZF kernels, whole file in ONE launch (FX_BGZF_GROUP=0)
#   bgzf_libs:gbp:lib[@K=V,...]:...   k_bgzf_* times for experiment builds (build/libfxgpu_*.so; pyfastx_amd/csrc/libfxgpu.so = the product)
#   bgzf_dbg[:gbp] the phase probes of k_bgzf_decode_par (FX_BGZF_DBG=8 / 1 / 2 / 4: header+tables / +A / +A2 / +B without stores)
#   pmc_fq[:n] pmc_fq_one[:n] pmc_fqcomp[:n] pmc_comp[:filter] pmc_fetch pmc_fxi[:n] pmc_sort[:n]   the standing counter passes (traffic, SQ, TCC) of a probe
#   prof_fastx     kernel trace of the kseq walk (tools/fastx_scale.py), default and walk-only
#   py:<script>:args...   any tools/*.py probe, stdout to <script>.json
TAG=${1:-run}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
pmc_pass() {   # name, kernel filter, command..., then -- counters
  local n=$1 filt=$2; shift 2
  local cmd=(); while [ "$1" != "--" ]; do cmd+=("$1"); shift; done; shift
  timeout 900 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$n -o pmc -- "${cmd[@]}" > $OUT/$n.out 2> $OUT/$n.err || { echo "pass $n failed"; tail -3 $OUT/$n.err; }
  FX_PMC_KERNEL=$filt python tools/pmc_dump.py $OUT/$n > $OUT/$n.txt
  find $OUT/$n -name '*.csv' -size +2M -delete
}
for STEP in "$@"; do
  IFS=: read -r S A1 A2 A3 A4 A5 A6 A7 A8 <<< "$STEP"
  echo "=== $STEP"
  case $S in
  tests) ( time timeout 1800 python -m pytest tests -m gpu -x -q ) > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log ;;
  pytest) FILES=""; for f in $A1 $A2 $A3 $A4 $A5 $A6 $A7 $A8; do FILES="$FILES tests/$f"; done
    ( time timeout 1800 python -m pytest $FILES -m gpu -x -q ) > $OUT/pytest_${A1%.py}.log 2>&1; tail -15 $OUT/pytest_${A1%.py}.log | grep -v "^$" | tail -12 ;;
  smoke) ( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log ;;
  bench) ( time timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err ) 2> $OUT/bench.time; cat $OUT/bench.time; cut -c1-800 $OUT/bench.json ;;
  prof)
    ( time timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/prof -o trace -- python bench.py --no-pmc --no-cpu-baseline > $OUT/prof_bench.json 2> $OUT/prof.err ) 2> $OUT/prof.time; cat $OUT/prof.time
    DB=$(find $OUT/prof -name '*.db' | head -1)
    [ -n "$DB" ] && python tools/rocprof_summary.py $DB $OUT/kernel_stats.txt && grep 'fx::' $OUT/kernel_stats.txt | head -80
    rm -rf $OUT/prof ;;
  bench_c3full) ( time timeout 2700 python bench.py --c3-reference-full --no-c4 > $OUT/bench_c3full.json 2> $OUT/bench_c3full.err ) 2> $OUT/bench_c3full.time; cat $OUT/bench_c3full.time; python -c "import json; d=json.loads([l for l in open('$OUT/bench_c3full.json') if l.startswith('{')][-1]); print(json.dumps(d['c3']['e2e_full'])[:3500])" ;;
  bench_c4full) ( time timeout 2400 python bench.py --c4-reference-full --no-c3 > $OUT/bench_c4full.json 2> $OUT/bench_c4full.err ) 2> $OUT/bench_c4full.time; cat $OUT/bench_c4full.time; python -c "import json; d=json.loads([l for l in open('$OUT/bench_c4full.json') if l.startswith('{')][-1]); print(json.dumps(d['c4'])[:3000])" ;;
  bench8) ( time FX_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 8 > $OUT/bench8.json 2> $OUT/bench8.err ) 2> $OUT/bench8.time; cat $OUT/bench8.time; python -c "import json,sys; d=json.loads([l for l in open('$OUT/bench8.json') if l.startswith('{')][-1]); print(json.dumps(d.get('fastq_strong')))"; tail -3 $OUT/bench8.err ;;
  counters) rocprofv3 -L > $OUT/counters.txt 2>&1; grep -c . $OUT/counters.txt ;;
  valuprobe)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/valuprobe tools/valuprobe.hip 2> /dev/null && /tmp/valuprobe > $OUT/valuprobe.txt 2>&1
    python tools/mixprobe.py >> $OUT/valuprobe.txt 2>> $OUT/mixprobe.err; cat $OUT/valuprobe.txt ;;
  firsttouch) /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/firsttouch tools/firsttouch.hip 2> /dev/null && /tmp/firsttouch ${A1:-16} > $OUT/firsttouch.txt 2>&1; cat $OUT/firsttouch.txt ;;
  gathercal)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-align-mismatch -o /tmp/gathercal tools/gathercal.hip 2> /dev/null
    /tmp/gathercal > $OUT/gathercal_known.txt 2>&1; cat $OUT/gathercal_known.txt
    pmc_pass gc_fetch cal_ /tmp/gathercal -- FETCH_SIZE
    pmc_pass gc_req cal_ /tmp/gathercal -- TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum
    pmc_pass gc_hit cal_ /tmp/gathercal -- TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
    python tools/gathercal_summary.py $OUT > $OUT/gathercal.txt; cat $OUT/gathercal.txt ;;
  bgzf_pmc)
    export FX_PROBE_FILE=/tmp/c4_probe.fa.gz FX_BGZF_GROUP=0
    P="python tools/bgzf_decode_probe.py ${A1:-3.0}"
    $P > $OUT/bgzf_plain.out 2>&1; tail -3 $OUT/bgzf_plain.out
    pmc_pass bz1 k_bgzf $P -- SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
    pmc_pass bz2 k_bgzf $P -- SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS
    pmc_pass bz3 k_bgzf $P -- SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_IFETCH SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC
    pmc_pass bz4 k_bgzf $P -- TA_BUSY_sum TA_TA_BUSY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum
    pmc_pass bz5 k_bgzf $P -- TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE
    pmc_pass bz6 k_bgzf $P -- FETCH_SIZE
    pmc_pass bz7 k_bgzf $P -- WRITE_SIZE
    pmc_pass bz8 k_bgzf $P -- TA_BUSY_avr TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
    pmc_pass bz9 k_bgzf $P -- TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_UTCL1_TRANSLATION_MISS_sum
    for n in bz1 bz2 bz3 bz4 bz5 bz6 bz7 bz8 bz9; do echo "-- $n"; grep decode_par $OUT/$n.txt | tail -1; done
    unset FX_BGZF_GROUP ;;
  bgzf_libs)
    export FX_PROBE_FILE=/tmp/c4_probe.fa.gz FX_BGZF_GROUP=0
    python tools/bgzf_decode_probe.py ${A1:-3.0} $A2 $A3 $A4 $A5 $A6 $A7 $A8 > $OUT/bgzf_libs.txt 2>&1; grep -v in-process $OUT/bgzf_libs.txt | cut -c1-400
    unset FX_BGZF_GROUP ;;
  bgzf_dbg)
    export FX_PROBE_FILE=/tmp/c4_probe.fa.gz FX_BGZF_GROUP=0
    for d in 8 1 2 4 5; do FX_BGZF_DBG=$d python tools/bgzf_decode_probe.py ${A1:-3.0} 2>&1 | grep "dbg=" | tail -1; done | tee $OUT/bgzf_dbg.txt
    unset FX_BGZF_GROUP ;;
  pmc_fq|pmc_fq_one|pmc_fqcomp|pmc_comp|pmc_fetch|pmc_fxi|pmc_sort)
    # the standing counter passes of a probe, one rocprofv3 run per set (counters in runs of their own, kernel trace only):
    #   pmc_fq[:n]      tools/fq_build_bench.py   k_fastq        pmc_fq_one[:n]  tools/fq_one_probe.py  k_fastq (one-read build)
    #   pmc_fqcomp[:n]  tools/fq_comp_probe.py    k_fastq_comp   pmc_comp        bench.py (2 steps)     k_fasta_comp / k_scan_comp / k_span_scan
    #   pmc_fetch       tools/fetch_probe.py      fetch          pmc_fxi[:n]     tools/fxi_pmc_probe.py k_fxi
    #   pmc_sort[:n]    tools/sort_probe.py       k_sort_* / k_rs_* (the name sort alone)
    case $S in
      pmc_fq) P="python tools/fq_build_bench.py ${A1:-2e7}"; F=k_fastq ;;
      pmc_fq_one) P="python tools/fq_one_probe.py ${A1:-2e7}"; F=k_fastq ;;
      pmc_fqcomp) P="python tools/fq_comp_probe.py ${A1:-1e7}"; F=k_fastq_comp ;;
      pmc_comp) P="python bench.py --no-pmc --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-c3 --no-c4"; F=${A1:-k_} ;;
      pmc_fetch) P="python tools/fetch_probe.py 3.0 2e7"; F=fetch ;;
      pmc_fxi) P="python tools/fxi_pmc_probe.py ${A1:-2e7}"; F=k_fxi ;;
      pmc_sort) P="python tools/sort_probe.py ${A1:-1e8}"; F=k_ ;;
    esac
    pmc_pass ${S}_fetch $F $P -- FETCH_SIZE
    pmc_pass ${S}_write $F $P -- WRITE_SIZE
    pmc_pass ${S}_sq $F $P -- SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
    pmc_pass ${S}_sq2 $F $P -- SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_LEVEL_VMEM
    pmc_pass ${S}_tcc $F $P -- TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum
    for n in fetch write sq sq2 tcc; do echo "-- $n"; awk 'NR%3==1' $OUT/${S}_$n.txt | tail -8 | cut -c1-400; done ;;
  prof_fastx)
    for MODE in default walk_only; do
      if [ $MODE = walk_only ]; then export FX_KSEQ_WALK_ONLY=1; else unset FX_KSEQ_WALK_ONLY; fi
      timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_$MODE -o trace -- python tools/fastx_scale.py > $OUT/$MODE.json 2> $OUT/$MODE.err
      DB=$(find $OUT/prof_$MODE -name '*.db' | head -1)
      [ -n "$DB" ] && python tools/rocprof_summary.py $DB $OUT/kernel_stats_$MODE.txt && grep 'k_kq\|^kernel' $OUT/kernel_stats_$MODE.txt
      rm -rf $OUT/prof_$MODE
    done ;;
  py) timeout 1200 python tools/$A1 $A2 $A3 $A4 $A5 $A6 > $OUT/${A1%.py}.json 2> $OUT/${A1%.py}.err; tail -3 $OUT/${A1%.py}.json | cut -c1-2500; tail -2 $OUT/${A1%.py}.err ;;
  *) echo "unknown step $S" ;;
  esac
done
