#!/usr/bin/env python3
"""What the parts of a step cost on their own (development aid): the whole step, the step without any serialization (reports only),
the EdDSA stage alone, the unsplit serializer alone.  Environment: P, N, WORKLOAD, plus any TMX_* knob."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from tendermintx_amd import Context, _lib  # noqa: E402
from tendermintx_amd.context import KIND_SKIP  # noqa: E402
from tendermintx_amd.synth import bench_workload  # noqa: E402

P, n = int(os.environ.get("P", "256")), int(os.environ.get("N", "128"))
w = bench_workload(os.environ.get("WORKLOAD", "survey8d"), n, P, seed=7)
dev = torch.device("cuda:0")
d = [torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev) for b in (w.proofs, w.targets, w.trusteds)]
stride = int(_lib.lib().tmx_elem_stride(KIND_SKIP, n))
out = torch.empty(P * stride, dtype=torch.int64, device=dev)
ed = torch.empty(P * n * 448, dtype=torch.uint8, device=dev)
rep = torch.empty(P * 64, dtype=torch.uint8, device=dev)
s = torch.cuda.Stream(dev)
ctx = Context(n, b"celestia", 100800, device=0, max_batch=P)


def timed(fn, k=40, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / k


full = timed(lambda: ctx.witness_batch_device(KIND_SKIP, P, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), out.data_ptr(), rep.data_ptr(), s.cuda_stream))
kms = ctx.kernel_ms_mean(40)
noser = timed(lambda: ctx.witness_batch_device(KIND_SKIP, P, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), None, rep.data_ptr(), s.cuda_stream))
kms2 = ctx.kernel_ms_mean(40)
edonly = timed(lambda: ctx.eddsa_lanes_device(P * n, d[1].data_ptr(), ed.data_ptr(), s.cuda_stream))
print(f"P={P} N={n}: full step {full:.4f} ms {({k: round(v, 3) for k, v in kms.items()})} | without serialization {noser:.4f} ms "
      f"{({k: round(v, 3) for k, v in kms2.items()})} | EdDSA stage alone {edonly:.4f} ms")
ctx.close()
