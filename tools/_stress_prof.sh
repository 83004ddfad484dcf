cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/stress
rm -rf $OUT; mkdir -p $OUT
for k in alone write read; do
  KINDS=$k rocprofv3 --kernel-trace -d $OUT/$k -o t -- python tools/stress_probe.py > $OUT/$k.log 2>&1
  echo "== $k"; tail -1 $OUT/$k.log
  python tools/rocpd_summary.py $(find $OUT/$k -name "*.db") | grep -E "k_ed_" | cut -c1-110
done
