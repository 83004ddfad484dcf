cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d gpurun_out/tl -o tl -- python tools/ser_span_ab.py > gpurun_out/tl.log 2>&1
ls gpurun_out/tl
python tools/step_timeline.py $(ls gpurun_out/tl/*.db | head -1)
