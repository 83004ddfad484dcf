#!/usr/bin/env python3
"""A/B of environment knobs on the bench workload (256 proofs x 128 validators): alternates the configurations, fresh context each time.
usage: ab_env.py REPS KEY=V1,V2,...   e.g.  ab_env.py 5 TMX_SCHEDULE=warm,cold"""
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from tendermintx_amd import Context, _lib  # noqa: E402
from tendermintx_amd.context import KIND_SKIP  # noqa: E402
from tendermintx_amd.synth import bench_workload  # noqa: E402

reps = int(sys.argv[1])
key, vals = sys.argv[2].split("=")
vals = vals.split(",")
P, n = int(os.environ.get("P", "256")), int(os.environ.get("N", "128"))
w = bench_workload(os.environ.get("WORKLOAD", "survey8d"), n, P, seed=7)
dev = torch.device("cuda:0")
d_proofs = torch.frombuffer(bytearray(w.proofs), dtype=torch.uint8).to(dev)
d_targets = torch.frombuffer(bytearray(w.targets), dtype=torch.uint8).to(dev)
d_trusteds = torch.frombuffer(bytearray(w.trusteds), dtype=torch.uint8).to(dev)
stride = int(_lib.lib().tmx_elem_stride(KIND_SKIP, n))
d_out = torch.empty(P * stride, dtype=torch.int64, device=dev)
d_rep = torch.empty(P * 64, dtype=torch.uint8, device=dev)
stream = torch.cuda.Stream(dev, priority=int(os.environ.get("STREAM_PRIO", "0")))  # -1 = a high-priority caller's stream
res = {v: [] for v in vals}
kern = {v: None for v in vals}
for r in range(reps):
    for v in vals:
        os.environ[key] = v
        ctx = Context(n, b"celestia", 100800, device=0, max_batch=P)
        def run(k):
            for _ in range(k):
                ctx.witness_batch_device(KIND_SKIP, P, d_proofs.data_ptr(), d_targets.data_ptr(), d_trusteds.data_ptr(), d_out.data_ptr(),
                                         d_rep.data_ptr(), stream.cuda_stream)
            torch.cuda.synchronize(dev)
        run(10)
        t0 = time.perf_counter()
        run(40)
        res[v].append(1e3 * (time.perf_counter() - t0) / 40)
        kern[v] = ctx.kernel_ms_mean(40)
        ctx.close()
for v in vals:
    xs = res[v]
    print(f"{key}={v}: step mean {statistics.mean(xs):.4f} min {min(xs):.4f} max {max(xs):.4f}  kernels {({k: round(x, 3) for k, x in kern[v].items()})}", flush=True)
