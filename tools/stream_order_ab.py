#!/usr/bin/env python3
"""Which stream the caller hands to the device entry point, and when it was created relative to the context (the HIP runtime maps streams to
its four hardware queues in creation order): one subprocess per variant, alternating.  usage: stream_order_ab.py REPS"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time
sys.path.insert(0, %r)
import torch
from tendermintx_amd import Context, _lib
from tendermintx_amd.context import KIND_SKIP
from tendermintx_amd.synth import bench_workload
P, n = int(os.environ.get("P", "256")), 128
mode = os.environ["MODE"]
w = bench_workload("survey8d", n, P, seed=0x544D58)
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
d = [torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev) for b in (w.proofs, w.targets, w.trusteds)]
stride = int(_lib.lib().tmx_elem_stride(KIND_SKIP, n))
out = torch.empty(P * stride, dtype=torch.int64, device=dev)
rep = torch.empty(P * 64, dtype=torch.uint8, device=dev)
s = None
if mode == "own_before":
    s = torch.cuda.Stream(dev)
ctx = Context(n, b"celestia", 100800, device=0, max_batch=P)
if mode == "own_after":
    s = torch.cuda.Stream(dev)
if mode == "default":
    s = torch.cuda.current_stream(dev)
if mode == "own_after_x2":   # two streams created after the context, the second one used
    s0 = torch.cuda.Stream(dev); s = torch.cuda.Stream(dev)
def run(k):
    for _ in range(k):
        ctx.witness_batch_device(KIND_SKIP, P, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), out.data_ptr(), rep.data_ptr(), s.cuda_stream)
    torch.cuda.synchronize()
res = []
for r in range(3):
    run(10)
    t0 = time.perf_counter(); run(50); res.append(1e3 * (time.perf_counter() - t0) / 50)
print("RES", min(res), sum(res) / len(res))
''' % ROOT
reps = int(sys.argv[1])
modes = ["default", "own_before", "own_after", "own_after_x2"]
res = {m: [] for m in modes}
for r in range(reps):
    for m in modes:
        o = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, MODE=m), capture_output=True, text=True)
        line = [x for x in o.stdout.splitlines() if x.startswith("RES")]
        if line:
            res[m].append(float(line[0].split()[1]))
        else:
            print(m, "FAILED", o.stderr[-300:])
for m in modes:
    if res[m]:
        print(f"{m:14s} min-of-3 per process: " + " ".join(f"{x:.4f}" for x in res[m]), flush=True)
