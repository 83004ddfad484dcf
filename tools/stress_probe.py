#!/usr/bin/env python3
"""Development aid: the EdDSA stage of a batch (the latency chain keys -> anchors -> multiples -> walk -> finish) timed alone, beside a
stream of pure HBM writes, beside a stream of pure HBM reads and beside a stream of arithmetic -- which kind of company stretches it."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from tendermintx_amd import Context  # noqa: E402
from tendermintx_amd.synth import bench_workload  # noqa: E402

P, n = int(os.environ.get("P", "256")), int(os.environ.get("N", "128"))
w = bench_workload(os.environ.get("WORKLOAD", "survey8d"), n, P, seed=7)
dev = torch.device("cuda:0")
d_t = torch.frombuffer(bytearray(w.targets), dtype=torch.uint8).to(dev)
ed = torch.empty(P * n * 448, dtype=torch.uint8, device=dev)
s = torch.cuda.Stream(dev)
bg = torch.cuda.Stream(dev, priority=0)
ctx = Context(n, b"celestia", 100800, device=0, max_batch=P)
big = torch.empty(1 << 28, dtype=torch.int64, device=dev)       # 2 GB
src = torch.ones(1 << 28, dtype=torch.int64, device=dev)
small = torch.rand(1 << 22, device=dev)


def stress(kind, reps):
    with torch.cuda.stream(bg):
        for _ in range(reps):
            if kind == "write":
                big.zero_()
            elif kind == "read":
                src.sum()
            elif kind == "copy":
                big.copy_(src)
            elif kind == "valu":
                x = small
                for _ in range(8):
                    x = torch.sin(x) * 1.0001 + 0.5
            elif kind == "valu_int":
                y = src[: 1 << 22]
                for _ in range(8):
                    y = y * 6364136223846793005 + 1442695040888963407


def run(kind):
    for _ in range(5):
        ctx.eddsa_lanes_device(P * n, d_t.data_ptr(), ed.data_ptr(), s.cuda_stream)
    torch.cuda.synchronize()
    ts = []
    for _ in range(12):
        if kind != "alone":
            stress(kind, 6 if kind in ("write", "read", "copy") else 40)
        time.sleep(0.0002)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(s):
            e0.record(s)
            ctx.eddsa_lanes_device(P * n, d_t.data_ptr(), ed.data_ptr(), s.cuda_stream)
            e1.record(s)
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    print(f"{kind:10s} EdDSA stage median {ts[len(ts)//2]:.4f} ms  min {ts[0]:.4f}  max {ts[-1]:.4f}", flush=True)


for kind in os.environ.get("KINDS", "alone,write,read,copy,valu,valu_int,alone").split(","):
    run(kind)
ctx.close()
