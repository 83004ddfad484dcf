#!/usr/bin/env python3
"""Two contexts on two caller streams, full batches alternating between them, against one context on one stream: what the device does when
the tail of one step (row writes, HBM-bound) may overlap the head of the next (dedup and hash role, a few latency-bound waves).
usage: python tools/two_ctx.py [steps]        (P, N, WORKLOAD from the environment; GPU_MAX_HW_QUEUES=8 for a queue per stream)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from tendermintx_amd import Context, _lib  # noqa: E402
from tendermintx_amd.context import KIND_SKIP  # noqa: E402
from tendermintx_amd.synth import bench_workload  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
P, n = int(os.environ.get("P", "256")), int(os.environ.get("N", "128"))
w = bench_workload(os.environ.get("WORKLOAD", "survey8d"), n, P, seed=7)
dev = torch.device("cuda:0")
up = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
d = (up(w.proofs), up(w.targets), up(w.trusteds))
stride = int(_lib.lib().tmx_elem_stride(KIND_SKIP, n))


def make():
    return (Context(n, b"celestia", 100800, device=0, max_batch=P), torch.cuda.Stream(dev), torch.empty(P * stride, dtype=torch.int64, device=dev),
            torch.zeros(P * 64, dtype=torch.uint8, device=dev))


def run(slots, k):
    for i in range(k):
        ctx, st, out, rep = slots[i % len(slots)]
        ctx.witness_batch_device(KIND_SKIP, P, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), out.data_ptr(), rep.data_ptr(), st.cuda_stream)
    torch.cuda.synchronize(dev)


for n_ctx in (1, 2, 1, 2, 3):
    slots = [make() for _ in range(n_ctx)]
    run(slots, 12)
    t0 = time.perf_counter()
    run(slots, steps)
    ms = 1e3 * (time.perf_counter() - t0) / steps
    ok = all(int(s[3].cpu().numpy().reshape(-1, 64)[:, 32:36].copy().view("uint32").sum()) == P for s in slots)
    print(f"{n_ctx} context(s) / stream(s), {steps} steps of {P} proofs: {ms:.4f} ms per step, all_ok {ok}")
    for s in slots:
        s[0].close()
