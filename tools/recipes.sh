#!/bin/bash
# One entry point for the measurement recipes that run on the GPU box (each used to be a one-off tools/_*.sh):
#   gpurun --timeout 1800 -- 'bash tools/recipes.sh <recipe> [args]'         environment knobs (TMX_*, P, N, WORKLOAD ...) pass through
# rocprofv3 passes never combine --pmc with anything but --kernel-trace.  Everything is written under gpurun_out/; copy what is judged into
# profiles/.  `bash tools/recipes.sh list` prints the recipes.
set -u
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
R=${1:-list}; shift || true
db() { find "$1" -name "*.db" | sort; }

case "$R" in
list) grep -E '^[a-z0-9_|]+\) +#' "$0" | sed 's/) *#/  --/' ;;

tl)  # kernel timeline of one full-batch step (P, N, WORKLOAD select the shape)
  rm -rf gpurun_out/tl
  rocprofv3 --kernel-trace -d gpurun_out/tl -o tl -- python tools/profile_step.py > gpurun_out/tl.log 2>&1
  python tools/step_timeline.py $(db gpurun_out/tl | head -1) ;;

tl1)  # kernel timeline of one step at P proofs (default 1), memory copies included
  export P=${P:-1}
  rm -rf gpurun_out/tl1
  WARM=5 STEPS=10 rocprofv3 --kernel-trace --memory-copy-trace -d gpurun_out/tl1 -o tl -- python tools/profile_step.py > gpurun_out/tl1.log 2>&1
  python tools/step_timeline.py $(db gpurun_out/tl1 | head -1); tail -1 gpurun_out/tl1.log ;;

ab_vals)  # one knob, explicit values, several sizes: KNOB=TMX_PHASE1_MAX VALS="0 1000000" SIZES="24 32 64"
  for p in ${SIZES:-32 64}; do for v in $VALS; do
    echo -n "P=$p $KNOB=$v  "; env P=$p $KNOB=$v timeout 300 python tools/ab_lib.py ${REPS:-2} tendermintx_amd/libtmx.so 2>&1 | tail -1 | cut -c30-200
  done; done ;;

ab_round)  # build_ab/prev.so vs build_ab/cur.so: steps at several sizes + isolated small launches warm / cold
  for p in ${SIZES:-256 1024 64}; do
    echo "== P=$p"; env P=$p timeout 600 python tools/ab_lib.py ${REPS:-3} build_ab/prev.so build_ab/cur.so 2>&1 | tail -2 | cut -c1-150
  done
  for cfg in ${SMALL:-1,128 1,512 8,128}; do
    IFS=, read p n <<< "$cfg"
    for l in prev cur prev cur; do echo -n "$l "; TMX_LIB=$PWD/build_ab/$l.so P=$p N=$n timeout 300 python tools/tiny_probe.py 2>&1 | tail -1 | cut -c1-200; done
  done ;;

chain)  # the chain kernels' durations with the EdDSA stage alone, beside k_proof, in the full step
  OUT=gpurun_out/chain; rm -rf $OUT; mkdir -p $OUT
  for m in ed noser full; do
    MODE=$m rocprofv3 --kernel-trace -d $OUT/$m -o t -- python tools/chain_probe.py > $OUT/$m.log 2>&1
    echo "== $m"; python tools/rocpd_summary.py $(db $OUT/$m) | grep -E "k_ed_|k_proof|k_serialize|k_verdict" | cut -c1-110
  done ;;

stress)  # the EdDSA stage beside a stream of HBM writes / reads
  OUT=gpurun_out/stress; rm -rf $OUT; mkdir -p $OUT
  for k in alone write read; do
    KINDS=$k rocprofv3 --kernel-trace -d $OUT/$k -o t -- python tools/stress_probe.py > $OUT/$k.log 2>&1
    echo "== $k"; tail -1 $OUT/$k.log; python tools/rocpd_summary.py $(db $OUT/$k) | grep -E "k_ed_" | cut -c1-110
  done ;;

trace_prof)  # Level-2 writer: per-kernel time, then FETCH_SIZE / WRITE_SIZE passes
  rm -rf gpurun_out/trprof*
  rocprofv3 --kernel-trace -d gpurun_out/trprof_t -o t -- python tools/trace_bench.py > gpurun_out/trprof_t.log 2>&1
  python tools/rocpd_summary.py $(db gpurun_out/trprof_t) | grep -i "trace\|kernel " | cut -c1-190
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c -d gpurun_out/trprof_$c -o t -- python tools/trace_bench.py > gpurun_out/trprof_$c.log 2>&1
    python tools/rocpd_summary.py $(db gpurun_out/trprof_$c) | grep -i "k_trace.*SIZE" | cut -c1-200
  done ;;

ladder_ab)  # Level-2 ladders: parity first, then rows-per-inversion x segments (CFGS="16,4 32,4 ...": rows,segs[,side])
  timeout 900 python -m pytest tests/test_trace.py tests/test_commit_pipeline.py -m gpu -x -q 2>&1 | tail -3
  TMX_FUZZ_TRACE_SEEDS=${TRACE_SEEDS:-12} timeout 900 python -m pytest tests/test_fuzz_extended.py -m gpu -x -q -k fuzz_trace_rows 2>&1 | tail -2
  for cfg in ${CFGS:-16,4 8,4 32,4 64,4 16,8 32,8 16,2}; do
    IFS=, read r g sd <<< "$cfg"; sd=${sd:-1}
    echo "ROWS=$r SEGS=$g SIDE=$sd: $(TMX_TRACE_ROWS=$r TMX_TRACE_SEGS=$g TMX_TRACE_SIDE=$sd SECTIONS=ladders timeout 300 python tools/trace_bench.py 2>&1 | grep -i 'ladders' | tr -s ' ')"
  done ;;

ladder_pmc)  # stall picture of the two ladder passes (PMCS="A B;C D" overrides the counter sets)
  rm -rf gpurun_out/ldpmc*; export SECTIONS=ladders
  IFS=';'
  for c in ${PMCS:-SQ_INSTS_VALU SQ_INSTS_SALU;SQ_WAVE_CYCLES SQ_BUSY_CYCLES;SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY;SQ_WAIT_INST_ANY SQ_WAIT_ANY;SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT;SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM;SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SALU}; do
    name=$(echo $c | tr ' ' '_'); IFS=' '
    timeout 300 rocprofv3 --kernel-trace --pmc $c -d gpurun_out/ldpmc_$name -o t -- python tools/trace_bench.py > gpurun_out/ldpmc_$name.log 2>&1
    echo "== $c"; python tools/rocpd_summary.py $(db gpurun_out/ldpmc_$name) 2>/dev/null | grep -i "k_trace_ladder" | cut -c1-60,80-200
    IFS=';'
  done ;;

ntt_prof)  # NTT / LDE: timings, kernel trace, VALU / LDS / FETCH / WRITE passes of one shape
  OUT=gpurun_out/ntt; rm -rf $OUT; mkdir -p $OUT
  python tools/ntt_bench.py 2>/dev/null | grep "^{" > $OUT/ntt_bench.jsonl
  rocprofv3 --kernel-trace --stats -d $OUT/trace -o ntt -- python tools/ntt_bench.py > $OUT/trace.log 2>&1
  export NTT_ONLY=${NTT_ONLY:-20x256}
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_LDS SQ_INSTS_SALU -d $OUT/pmc_valu -o ntt -- python tools/ntt_bench.py > $OUT/pmc.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/pmc_lds -o ntt -- python tools/ntt_bench.py > $OUT/pmc1.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o ntt -- python tools/ntt_bench.py > $OUT/pmc2.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o ntt -- python tools/ntt_bench.py > $OUT/pmc3.log 2>&1
  python tools/rocpd_summary.py $(db $OUT) > $OUT/rocprofv3_summary.txt 2>&1
  cat $OUT/ntt_bench.jsonl | cut -c1-300; grep -E "k_ntt|k_lde|kernel " $OUT/rocprofv3_summary.txt | cut -c1-170 | head -40 ;;

poseidon_prof)  # Poseidon Merkle commitment: timings, per-kernel time, VALU instruction count
  rm -rf gpurun_out/pos; mkdir -p gpurun_out/pos
  python tools/poseidon_bench.py > gpurun_out/pos/bench.jsonl 2>gpurun_out/pos/bench.err
  rocprofv3 --kernel-trace --stats -d gpurun_out/pos/trace -o pos -- python tools/poseidon_bench.py 21 64 4 > gpurun_out/pos/trace.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES -d gpurun_out/pos/pmc -o pos -- python tools/poseidon_bench.py 21 64 4 > gpurun_out/pos/pmc.log 2>&1
  python tools/rocpd_summary.py $(db gpurun_out/pos) > gpurun_out/pos/summary.txt 2>&1
  cat gpurun_out/pos/bench.jsonl; grep -E "poseidon|counter" gpurun_out/pos/summary.txt | head -20 ;;

tiny_parts)  # role / stage timings of k_tiny / k_tiny_tail: needs `bash tools/build_variant.sh tinyprof -DTMX_TINY_PROF` first
  export TMX_LIB=build_ab/tinyprof.so QUICK=1
  for d in 0 0x1e 0x11e 0x21e 0x41e 0x1d 0x1b 0x17 0x0f 0x1f 0x1000 0x2000 0x4000 0x8000 0xc000; do
    TMX_TINY_DBG=$d timeout 120 python tools/tiny_probe.py 2>&1 | grep DBG
  done ;;

round)  # the round's evidence in one GPU call: bash tools/recipes.sh round r05   (writes gpurun_out/<tag>/)
  TAG=${1:-rXX}; O=gpurun_out/$TAG
  bash tools/collect_profiles.sh $TAG > gpurun_out/${TAG}_collect.log 2>&1
  P=1 bash "$0" tl1 > $O/single_proof_timeline.txt 2>&1
  P=32 bash "$0" tl1 > $O/p32_timeline.txt 2>&1
  bash "$0" tl > $O/step_timeline.txt 2>&1
  (for p in 1 4 8; do P=$p timeout 200 python tools/tiny_probe.py | tail -1; done; TMX_TINY=0 timeout 200 python tools/tiny_probe.py | tail -1
   P=1 N=512 timeout 200 python tools/tiny_probe.py | tail -1; P=1 N=32 timeout 200 python tools/tiny_probe.py | tail -1) > $O/tiny_probe.txt 2>&1
  timeout 300 python tools/churn_probe.py > $O/churn_probe.txt 2>&1
  tail -3 gpurun_out/${TAG}_collect.log; cut -c1-220 $O/tiny_probe.txt; tail -12 $O/single_proof_timeline.txt; tail -40 $O/step_timeline.txt; tail -2 $O/churn_probe.txt ;;

commit_prof)  # commit pipeline per section under a kernel trace: bash tools/recipes.sh commit_prof r05
  TAG=${1:-rXX}; O=gpurun_out/$TAG; mkdir -p $O
  rocprofv3 --kernel-trace --stats -d $O/commit_trace -o commit -- python tools/commit_bench.py > $O/commit_bench.jsonl 2> $O/commit_bench.err
  python tools/rocpd_summary.py $(db $O/commit_trace | head -1) > $O/commit_rocprofv3_summary.txt 2>&1
  tail -5 $O/commit_bench.jsonl | cut -c1-400 ;;

*) echo "unknown recipe $R"; exit 2 ;;
esac
