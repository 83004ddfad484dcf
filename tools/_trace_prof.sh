# per-kernel time and HBM traffic of the Level-2 writer: bash tools/_trace_prof.sh
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/trprof*
rocprofv3 --kernel-trace -d gpurun_out/trprof_t -o t -- python tools/trace_bench.py > gpurun_out/trprof_t.log 2>&1
python tools/rocpd_summary.py $(find gpurun_out/trprof_t -name "*.db") | grep -i "trace\|kernel " | cut -c1-190
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d gpurun_out/trprof_$c -o t -- python tools/trace_bench.py > gpurun_out/trprof_$c.log 2>&1
  python tools/rocpd_summary.py $(find gpurun_out/trprof_$c -name "*.db") | grep -i "k_trace.*SIZE" | cut -c1-200
done
