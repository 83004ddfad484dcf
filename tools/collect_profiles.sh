#!/bin/bash
# Collects the rocprofv3 evidence of a round on the GPU box: kernel trace of bench.py (the judged command) and, in separate runs,
# PMC passes over tools/profile_step.py (never combined with sys/hip/hsa traces).  Usage: bash tools/collect_profiles.sh <tag>
set -u
TAG=${1:-rXX}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
python bench.py > "$OUT/bench.out" 2> "$OUT/bench.err"
tail -1 "$OUT/bench.out" > "$OUT/bench.json"                       # the record line, byte for byte (what the driver parses)
cp bench_extras.json "$OUT/bench_extras.json" 2>/dev/null           # the full record
tools/microbench/valu_isa > "$OUT/valu_isa.txt" 2>&1
TMX_BENCH_NO_PMC=1 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o bench -- python bench.py --no-cpu-baseline > "$OUT/bench_under_rocprof.log" 2>&1
rocprofv3 --kernel-trace -d "$OUT/trace_step" -o step -- python tools/profile_step.py > "$OUT/step_trace.log" 2>&1
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES"; do
  name=$(echo $pass | tr ' ' '_' | tr 'A-Z' 'a-z')
  rocprofv3 --kernel-trace --pmc $pass -d "$OUT/pmc_$name" -o step -- python tools/profile_step.py > "$OUT/pmc_$name.log" 2>&1
done
python tools/rocpd_summary.py $(find "$OUT" -name "*.db" | sort) > "$OUT/rocprofv3_summary.txt" 2>&1
python tools/pmc_to_json.py "$OUT" > "$OUT/pmc.json" 2> "$OUT/pmc_to_json.err"
tail -1 "$OUT/bench.json" | cut -c1-400
cat "$OUT/step_trace.log" | tail -1
