# clocks + pass-1 sweep probe: bash tools/_g2.sh
cd "$GRAFT_REPO_ROOT"
export TRACE_SEEDS=${TRACE_SEEDS:-6}
bash tools/_ladder_ab.sh 2>&1 | tail -12
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
SECTIONS=ladders rocprofv3 --kernel-trace -d gpurun_out/ld_t -o t -- python tools/trace_bench.py > gpurun_out/ld_t.log 2>&1
python tools/rocpd_summary.py $(find gpurun_out/ld_t -name "*.db") | grep -i "ladder\|kernel " | cut -c1-190
