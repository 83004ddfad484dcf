# kernel timeline of one full-batch step: bash tools/_tl.sh   (environment knobs pass through; P, N, WORKLOAD select the shape)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/tl
rocprofv3 --kernel-trace -d gpurun_out/tl -o tl -- python tools/profile_step.py > gpurun_out/tl.log 2>&1
python tools/step_timeline.py $(find gpurun_out/tl -name "*.db" | head -1)
