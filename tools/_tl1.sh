# kernel timeline of one step at P proofs (default 1): P=32 bash tools/_tl1.sh   (environment knobs pass through)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export P=${P:-1}
rm -rf gpurun_out/tl1
WARM=5 STEPS=10 rocprofv3 --kernel-trace --memory-copy-trace -d gpurun_out/tl1 -o tl -- python tools/profile_step.py > gpurun_out/tl1.log 2>&1
python tools/step_timeline.py $(find gpurun_out/tl1 -name "*.db" | head -1)
tail -1 gpurun_out/tl1.log
