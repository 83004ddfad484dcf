#!/usr/bin/env python3
"""Development aid: run one part of a step repeatedly (MODE = ed | noser | full) so that a rocprofv3 kernel trace shows how long the
kernels of the EdDSA chain take beside each kind of company.  Environment: P, N, WORKLOAD, MODE, plus any TMX_* knob."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from tendermintx_amd import Context, _lib  # noqa: E402
from tendermintx_amd.context import KIND_SKIP  # noqa: E402
from tendermintx_amd.synth import bench_workload  # noqa: E402

P, n = int(os.environ.get("P", "256")), int(os.environ.get("N", "128"))
mode = os.environ.get("MODE", "full")
w = bench_workload(os.environ.get("WORKLOAD", "survey8d"), n, P, seed=7)
dev = torch.device("cuda:0")
d = [torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev) for b in (w.proofs, w.targets, w.trusteds)]
stride = int(_lib.lib().tmx_elem_stride(KIND_SKIP, n))
out = torch.empty(P * stride, dtype=torch.int64, device=dev)
ed = torch.empty(P * n * 448, dtype=torch.uint8, device=dev)
rep = torch.empty(P * 64, dtype=torch.uint8, device=dev)
s = torch.cuda.Stream(dev)
ctx = Context(n, b"celestia", 100800, device=0, max_batch=P)
for _ in range(24):
    if mode == "ed":
        ctx.eddsa_lanes_device(P * n, d[1].data_ptr(), ed.data_ptr(), s.cuda_stream)
    elif mode == "noser":
        ctx.witness_batch_device(KIND_SKIP, P, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), None, rep.data_ptr(), s.cuda_stream)
    else:
        ctx.witness_batch_device(KIND_SKIP, P, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), out.data_ptr(), rep.data_ptr(), s.cuda_stream)
    torch.cuda.synchronize()
ctx.close()
