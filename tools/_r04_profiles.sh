# round-4 evidence in one GPU call: bash tools/_r04_profiles.sh   (writes gpurun_out/r04/...)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
bash tools/collect_profiles.sh r04 > gpurun_out/r04_collect.log 2>&1
P=1 bash tools/_tl1.sh > gpurun_out/r04/single_proof_timeline.txt 2>&1
P=32 bash tools/_tl1.sh > gpurun_out/r04/p32_timeline.txt 2>&1
bash tools/_tl.sh > gpurun_out/r04/step_timeline.txt 2>&1
(for p in 1 4 8; do P=$p timeout 200 python tools/tiny_probe.py | tail -1; done; TMX_TINY=0 timeout 200 python tools/tiny_probe.py | tail -1; P=1 N=512 timeout 200 python tools/tiny_probe.py | tail -1; P=1 N=32 timeout 200 python tools/tiny_probe.py | tail -1) > gpurun_out/r04/tiny_probe.txt 2>&1
rocprofv3 --kernel-trace --stats -d gpurun_out/r04/commit_trace -o commit -- python tools/commit_bench.py > gpurun_out/r04/commit_bench.jsonl 2> gpurun_out/r04/commit_bench.err
python tools/rocpd_summary.py $(find gpurun_out/r04/commit_trace -name "*.db" | head -1) > gpurun_out/r04/commit_rocprofv3_summary.txt 2>&1
tail -3 gpurun_out/r04_collect.log; cat gpurun_out/r04/tiny_probe.txt | cut -c1-220; cat gpurun_out/r04/single_proof_timeline.txt | tail -12; tail -5 gpurun_out/r04/commit_bench.jsonl
