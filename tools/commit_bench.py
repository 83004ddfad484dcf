#!/usr/bin/env python3
"""The commit pipeline alone (tmx_trace_commit_device) on the bench workload: P proofs x N lanes, one JSON line per section.
   P=256 SECTIONS=sha512,sha256 python tools/commit_bench.py          (rocprofv3 --kernel-trace --stats -- ... gives the per-kernel split)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from tendermintx_amd import Context, _lib  # noqa: E402
from tendermintx_amd.context import KIND_SKIP  # noqa: E402
from tendermintx_amd.synth import bench_workload  # noqa: E402

P, n = int(os.environ.get("P", "256")), int(os.environ.get("N", "128"))
names = os.environ.get("SECTIONS", "sha512,sha256,tree,header").split(",")
SEC = {"ladders": _lib.TRACE_LADDERS, "sha512": _lib.TRACE_SHA512, "sha256": _lib.TRACE_SHA256, "tree": _lib.TRACE_TREE, "header": _lib.TRACE_HEADER}
w = bench_workload("survey8d", n, P, seed=0x544D58)
dev = torch.device("cuda:0")
d = [torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev) for b in (w.proofs, w.targets, w.trusteds)]
ctx = Context(n, b"celestia", 100800, device=0, max_batch=P)
out = torch.empty(P * ctx.elem_stride(KIND_SKIP), dtype=torch.int64, device=dev)
rep = torch.empty(P * 64, dtype=torch.uint8, device=dev)
tr = torch.empty(P * ctx.trace_elem_count(KIND_SKIP), dtype=torch.int64, device=dev)
ctx.witness_batch_device(KIND_SKIP, P, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), out.data_ptr(), rep.data_ptr(), 0)
ctx.trace_rows_device(KIND_SKIP, P, d[1].data_ptr(), d[2].data_ptr(), tr.data_ptr(), _lib.TRACE_ALL, 0)
cap = torch.zeros(64, dtype=torch.int64, device=dev)
for name in names:
    log_rows, width = ctx.trace_commit_shape(KIND_SKIP, SEC[name])
    for _ in range(int(os.environ.get("REPS", "3"))):
        ctx.trace_commit_device(KIND_SKIP, P, SEC[name], 3, 4, tr.data_ptr(), cap.data_ptr(), 0)
    ms = ctx.trace_commit_last_ms()
    cols = P * width
    perms = (1 << (log_rows + 3)) * ((cols + 7) // 8 + 1)
    print(json.dumps({"section": name, "proofs": P, "n": n, "columns": cols, "log_rows": log_rows, "ms": {k: round(v, 4) for k, v in ms.items()},
                      "lde_gbytes_extended": round((cols << (log_rows + 3)) * 8 / 1e9, 3), "gperm_per_s": round(perms / (ms["merkle"] * 1e-3) / 1e9, 3)}), flush=True)
ctx.close()
