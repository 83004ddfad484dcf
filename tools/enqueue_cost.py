#!/usr/bin/env python3
"""Host-side enqueue time of one batch vs its GPU time (is the step launch-bound?)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tendermintx_amd import Context, _lib
from tendermintx_amd.context import KIND_SKIP
from tendermintx_amd.synth import Workload
P, n = int(os.environ.get("P", "256")), int(os.environ.get("N", "128"))
w = Workload(KIND_SKIP, n, P, n, chain_id=b"celestia", seed=7)
dev = torch.device("cuda:0")
d = [torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev) for b in (w.proofs, w.targets, w.trusteds)]
stride = int(_lib.lib().tmx_elem_stride(KIND_SKIP, n))
out = torch.empty(P * stride, dtype=torch.int64, device=dev)
rep = torch.empty(P * 64, dtype=torch.uint8, device=dev)
s = torch.cuda.Stream(dev)
ctx = Context(n, b"celestia", 100800, device=0, max_batch=P)
def call():
    ctx.witness_batch_device(KIND_SKIP, P, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), out.data_ptr(), rep.data_ptr(), s.cuda_stream)
for _ in range(10): call()
torch.cuda.synchronize()
for K in (1, 50):
    t0 = time.perf_counter()
    for _ in range(K): call()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"K={K}: enqueue {1e3*(t1-t0)/K:.4f} ms/call, total {1e3*(t2-t0)/K:.4f} ms/call", flush=True)
