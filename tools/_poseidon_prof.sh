# Poseidon Merkle commitment: timings, then per-kernel time and the VALU instruction count (separate PMC pass) of one shape
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pos; mkdir -p gpurun_out/pos
python tools/poseidon_bench.py > gpurun_out/pos/bench.jsonl 2>gpurun_out/pos/bench.err
rocprofv3 --kernel-trace --stats -d gpurun_out/pos/trace -o pos -- python tools/poseidon_bench.py 21 64 4 > gpurun_out/pos/trace.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES -d gpurun_out/pos/pmc -o pos -- python tools/poseidon_bench.py 21 64 4 > gpurun_out/pos/pmc.log 2>&1
python tools/rocpd_summary.py $(find gpurun_out/pos -name "*.db" | sort) > gpurun_out/pos/summary.txt 2>&1
cat gpurun_out/pos/bench.jsonl; grep -E "poseidon|counter" gpurun_out/pos/summary.txt | head -20
