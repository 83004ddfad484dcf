#!/usr/bin/env python3
"""A/B of library builds on the bench workload: ab_lib.py REPS libA.so libB.so ...  (one subprocess per measurement, alternating;
differences between gpurun boxes are larger than most kernel-level changes, so variants are compared inside one call)."""
import os
import statistics
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
P, n = int(os.environ.get("P", "256")), int(os.environ.get("N", "128"))
w = bench_workload(os.environ.get("WORKLOAD", "survey8d"), n, P, seed=7)
dev = torch.device("cuda:0")
d = [torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev) for b in (w.proofs, w.targets, w.trusteds)]
stride = int(_lib.lib().tmx_elem_stride(KIND_SKIP, n))
out = torch.empty(P * stride, dtype=torch.int64, device=dev)
rep = torch.empty(P * 64, dtype=torch.uint8, device=dev)
s = torch.cuda.Stream(dev)
ctx = Context(n, b"celestia", 100800, device=0, max_batch=P)
def run(k):
    for _ in range(k):
        ctx.witness_batch_device(KIND_SKIP, P, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), out.data_ptr(), rep.data_ptr(), s.cuda_stream)
    torch.cuda.synchronize()
res = []
for r in range(int(os.environ.get("INNER", "3"))):
    run(10)
    t0 = time.perf_counter(); run(40); res.append(1e3 * (time.perf_counter() - t0) / 40)
ok = bool((rep.view(P, 64)[:, 32] == 1).all().item())
print("RES", min(res), ok, {k: round(v, 3) for k, v in ctx.kernel_ms_mean(40).items()})
''' % ROOT
reps = int(sys.argv[1])
libs = sys.argv[2:]
res = {l: [] for l in libs}
last = {}
for r in range(reps):
    for l in libs:
        env = dict(os.environ, TMX_LIB=os.path.abspath(l))
        o = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
        line = [x for x in o.stdout.splitlines() if x.startswith("RES")]
        if not line:
            print(l, "FAILED", o.stderr[-400:])
            continue
        f = line[0].split(None, 3)
        res[l].append(float(f[1]))
        last[l] = (f[2], f[3])
for l in libs:
    if res[l]:
        print(f"{os.path.basename(l):28s} step min {min(res[l]):.4f} mean {statistics.mean(res[l]):.4f} max {max(res[l]):.4f} all_ok {last[l][0]} {last[l][1]}", flush=True)
