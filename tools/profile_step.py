#!/usr/bin/env python3
"""The bench workload and nothing else (one context, WARM + STEPS full batches): the command the PMC passes in profiles/ run on, so
that every dispatch of a counter database belongs to a full batch and per-batch figures are sum / (WARM + STEPS)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from tendermintx_amd import Context  # noqa: E402
from tendermintx_amd.context import KIND_SKIP  # noqa: E402
from tendermintx_amd.synth import bench_workload  # noqa: E402

P, n = int(os.environ.get("P", "256")), int(os.environ.get("N", "128"))
WARM, STEPS = int(os.environ.get("WARM", "2")), int(os.environ.get("STEPS", "6"))
WORKLOAD = os.environ.get("WORKLOAD", "survey8d")  # bench.py's default workload
w = bench_workload(WORKLOAD, n, P, seed=0x544D58)
dev = torch.device("cuda:0")
d = [torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev) for b in (w.proofs, w.targets, w.trusteds)]
ctx = Context(n, b"celestia", 100800, device=0, max_batch=P)
stride = ctx.elem_stride(KIND_SKIP)
out = torch.empty(P * stride, dtype=torch.int64, device=dev)
rep = torch.empty(P * 64, dtype=torch.uint8, device=dev)
s = torch.cuda.Stream(dev)
def run(k):
    for _ in range(k):
        ctx.witness_batch_device(KIND_SKIP, P, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), out.data_ptr(), rep.data_ptr(), s.cuda_stream)
    torch.cuda.synchronize()
run(WARM)
t0 = time.perf_counter()
run(STEPS)
print(f"profile_step: {1e3 * (time.perf_counter() - t0) / STEPS:.4f} ms/step over {STEPS} steps (+{WARM} warm), P={P} N={n} workload={WORKLOAD}")
NEW_KEYS = int(os.environ.get("NEW_KEYS", "0"))  # then: one more step in which that many keys are new to the cache, and a plain one behind it
if NEW_KEYS:                                      # (tools/step_timeline.py prints the step before the last: the one with the new keys)
    from tendermintx_amd.synth import Workload
    wj = Workload(0, n, 1, NEW_KEYS, chain_id=b"celestia", seed=0x700000 + NEW_KEYS, signed_permille=1000)
    d1 = [torch.frombuffer(bytearray(a + b[len(a):]), dtype=torch.uint8).to(dev) for a, b in ((wj.proofs, w.proofs), (wj.targets, w.targets), (wj.trusteds, w.trusteds))]
    ctx.witness_batch_device(KIND_SKIP, P, d1[0].data_ptr(), d1[1].data_ptr(), d1[2].data_ptr(), out.data_ptr(), rep.data_ptr(), s.cuda_stream)
    torch.cuda.synchronize()
    run(1)
ctx.close()
