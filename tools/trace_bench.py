#!/usr/bin/env python3
"""Level-2 trace-row writer on the bench workload: time per section and achieved HBM write rate.  P, N, WORKLOAD from the environment."""
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
w = bench_workload(os.environ.get("WORKLOAD", "survey8d"), n, P, seed=7)
dev = torch.device("cuda:0")
d = [torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev) for b in (w.proofs, w.targets, w.trusteds)]
ctx = Context(n, b"celestia", 100800, device=0, max_batch=P)
out = torch.empty(P * ctx.elem_stride(KIND_SKIP), dtype=torch.int64, device=dev)
rep = torch.empty(P * 64, dtype=torch.uint8, device=dev)
te = ctx.trace_elem_count(KIND_SKIP)
tr = torch.empty(P * te, dtype=torch.int64, device=dev)
s = torch.cuda.Stream(dev)
ctx.witness_batch_device(KIND_SKIP, P, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), out.data_ptr(), rep.data_ptr(), s.cuda_stream)
torch.cuda.synchronize()
tn, sz = 0, n
while sz > 1:
    sz = (sz + 1) // 2
    tn += sz
secs = {"ladders": (1, n * 2 * 256 * 65), "sha512": (2, n * 2880), "sha256": (4, n * 2 * 576), "match": (8, n * n), "tree": (16, 2 * tn * 1152),
        "header": (32, 20 * 1152), "all": (63, te)}
only = os.environ.get("SECTIONS")
for name, (mask, elems) in secs.items():
    if only and name not in only.split(","):
        continue
    for _ in range(2):
        ctx.trace_rows_device(KIND_SKIP, P, d[1].data_ptr(), d[2].data_ptr(), tr.data_ptr(), mask, s.cuda_stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 5
    for _ in range(K):
        ctx.trace_rows_device(KIND_SKIP, P, d[1].data_ptr(), d[2].data_ptr(), tr.data_ptr(), mask, s.cuda_stream)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / K
    gb = P * elems * 8 / 1e9
    print(f"trace {name:8s} P={P} N={n}: {ms:9.3f} ms  {gb:7.3f} GB  {gb / (ms * 1e-3):8.1f} GB/s = {gb / (ms * 1e-3) / 8000:.3f} of 8 TB/s", flush=True)
ctx.close()
