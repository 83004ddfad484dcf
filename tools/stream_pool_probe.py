import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tendermintx_amd import Context, _lib
from tendermintx_amd.context import KIND_SKIP
from tendermintx_amd.synth import Workload
P, n = 256, 128
w = Workload(KIND_SKIP, n, P, n, chain_id=b"celestia", seed=7)
dev = torch.device("cuda:0")
d_proofs = torch.frombuffer(bytearray(w.proofs), dtype=torch.uint8).to(dev)
d_targets = torch.frombuffer(bytearray(w.targets), dtype=torch.uint8).to(dev)
d_trusteds = torch.frombuffer(bytearray(w.trusteds), dtype=torch.uint8).to(dev)
stride = int(_lib.lib().tmx_elem_stride(KIND_SKIP, n))
d_out = torch.empty(P * stride, dtype=torch.int64, device=dev)
d_rep = torch.empty(P * 64, dtype=torch.uint8, device=dev)
stream = torch.cuda.Stream(dev)
def run(ctx, k):
    for _ in range(k):
        ctx.witness_batch_device(KIND_SKIP, P, d_proofs.data_ptr(), d_targets.data_ptr(), d_trusteds.data_ptr(), d_out.data_ptr(), d_rep.data_ptr(), stream.cuda_stream)
    torch.cuda.synchronize(dev)
def step(tag):
    c = Context(n, b"celestia", 100800, device=0, max_batch=P)
    run(c, 10)
    t0 = time.perf_counter(); run(c, 30); ms = (time.perf_counter() - t0) / 30 * 1e3
    print(f"{tag}: step {ms:.4f} ms", {k: round(v, 3) for k, v in c.kernel_ms_mean(20).items()}, flush=True)
    return c
mode = sys.argv[1]
keep = []
if mode.startswith("streams"):
    pr = [int(x) for x in mode[7:].split(",")] if len(mode) > 7 else [-1, 0, 0, 0]
    keep = [torch.cuda.Stream(dev, priority=p) for p in pr]
    for s in keep:
        with torch.cuda.stream(s):
            torch.zeros(16, device=dev)
    torch.cuda.synchronize()
elif mode == "mem":
    keep = [torch.empty(64 << 20, dtype=torch.uint8, device=dev) for _ in range(16)]
elif mode == "ctx":
    keep = [Context(n, b"celestia", 100800, device=0, max_batch=P)]
elif mode == "ctx_small":
    keep = [Context(8, b"celestia", 100800, device=0, max_batch=2)]
c = step(mode + " first measured ctx")
c.close()
c = step(mode + " second measured ctx")
