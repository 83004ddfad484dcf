cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/chain
rm -rf $OUT; mkdir -p $OUT
for m in ed noser full; do
  MODE=$m rocprofv3 --kernel-trace -d $OUT/$m -o t -- python tools/chain_probe.py > $OUT/$m.log 2>&1
  echo "== $m"
  python tools/rocpd_summary.py $(find $OUT/$m -name "*.db") | grep -E "k_ed_|k_proof|k_serialize|k_verdict" | cut -c1-110
done
