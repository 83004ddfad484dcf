import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from tendermintx_amd import Context, _lib
from tendermintx_amd.context import KIND_SKIP
from tendermintx_amd.synth import bench_workload
P, n = 256, 128
w = bench_workload("survey8d", n, P, seed=0x544D58)
dev = torch.device("cuda:0")
d = [torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev) for b in (w.proofs, w.targets, w.trusteds)]
stride = int(_lib.lib().tmx_elem_stride(KIND_SKIP, n))
out = torch.empty(P * stride, dtype=torch.int64, device=dev)
rep = torch.empty(P * 64, dtype=torch.uint8, device=dev)
streams = {"default": torch.cuda.current_stream(dev), "own": torch.cuda.Stream(dev), "own_high": torch.cuda.Stream(dev, priority=-1)}
ctx = Context(n, b"celestia", 100800, device=0, max_batch=P)
def run(s, k):
    for _ in range(k):
        ctx.witness_batch_device(KIND_SKIP, P, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), out.data_ptr(), rep.data_ptr(), s.cuda_stream)
    torch.cuda.synchronize()
for rnd in range(3):
    for name, s in streams.items():
        run(s, 10)
        t0 = time.perf_counter(); run(s, 50); ms = 1e3 * (time.perf_counter() - t0) / 50
        print(f"round {rnd} stream {name}: {ms:.4f} ms/step", flush=True)
