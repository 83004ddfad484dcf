#!/usr/bin/env python3
"""Goldilocks NTT / LDE throughput on one GPU (HIP events on the launch stream) beside the CPU oracle on one column.
Two HBM fractions per shape: `frac_passes` counts what the kernels really move -- one read + one write of every element per pass (one pass up
to 2^11, two above: the four-step split) -- and `frac_1r1w` the algorithmic minimum of ONE read and ONE write of the column (what a transform
that kept a 2^20-element column on chip would move).  Neither is the kernel's roof: `valu_issue_frac` is (profiles/r04_ntt_rocprofv3_summary.txt:
6.65e8 wave-level VALU instructions per pass at 2^20 x 256 = 158 per element and pass, 15.8 per element and stage, scaled by the stage count)
the fraction of the chip's issue slots the butterflies need in the measured time."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle", "py"))
import torch  # noqa: E402
import tendermintx_amd as tmx  # noqa: E402
import oracle_c as oc  # noqa: E402

P = 2**64 - 2**32 + 1
ctx = tmx.Context(4, b"celestia", max_batch=1)
s = torch.cuda.current_stream().cuda_stream
rng = np.random.default_rng(1)
rows = []
CASES = ((10, 4096), (11, 2048), (16, 256), (20, 64), (20, 256), (22, 16))
if os.environ.get("NTT_ONLY"):  # e.g. NTT_ONLY=20x256 (profiling runs)
    CASES = tuple(tuple(int(v) for v in c.split("x")) for c in os.environ["NTT_ONLY"].split(","))
for log_n, cols in CASES:
    n = 1 << log_n
    x = rng.integers(0, P, size=(cols, n), dtype=np.uint64)
    d = torch.from_numpy(x.view(np.int64)).to("cuda:0")
    out = torch.empty_like(d)
    for _ in range(3):
        ctx.ntt_device(log_n, cols, d.data_ptr(), out.data_ptr(), False, s)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    e0.record()
    for _ in range(reps):
        ctx.ntt_device(log_n, cols, d.data_ptr(), out.data_ptr(), False, s)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    passes = 1 if log_n <= 11 else 2
    nbytes = passes * 2 * 8 * n * cols
    a = time.perf_counter()
    ref = oc.ntt(x[0])
    cpu_ms = 1e3 * (time.perf_counter() - a)
    ok = bool(np.array_equal(out[0].cpu().numpy().view(np.uint64), ref))
    rows.append({"log_n": log_n, "cols": cols, "ms": round(ms, 4), "elements_per_s": round(n * cols / (ms * 1e-3), 0),
                 "algorithmic_gbs": round(nbytes / (ms * 1e-3) / 1e9, 1), "frac_passes": round(nbytes / (ms * 1e-3) / 8e12, 3),
                 "frac_1r1w": round(2 * 8 * n * cols / (ms * 1e-3) / 8e12, 3), "passes": passes,
                 "valu_issue_frac": round(15.8 * log_n * n * cols / 64 * 4.07 / (1024 * 2.4e9) / (ms * 1e-3), 3),
                 "butterfly_mul_per_s": round(n * cols * log_n / 2 / (ms * 1e-3), 0),
                 "cpu_oracle_ms_one_column": round(cpu_ms, 2), "speedup_vs_one_core": round(cpu_ms * cols / ms, 0), "bit_exact_col0": ok})
    print(json.dumps(rows[-1]), flush=True)
if os.environ.get("NTT_ONLY"):
    ctx.close()
    sys.exit(0)
log_n, lb, cols = 18, 3, 32
x = rng.integers(0, P, size=(cols, 1 << log_n), dtype=np.uint64)
d = torch.from_numpy(x.view(np.int64)).to("cuda:0")
out = torch.empty((cols, 1 << (log_n + lb)), dtype=torch.int64, device="cuda:0")
for _ in range(2):
    ctx.lde_device(log_n, lb, cols, d.data_ptr(), out.data_ptr(), s)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    ctx.lde_device(log_n, lb, cols, d.data_ptr(), out.data_ptr(), s)
e1.record()
torch.cuda.synchronize()
print(json.dumps({"lde": {"log_n": log_n, "log_blowup": lb, "cols": cols, "ms": round(e0.elapsed_time(e1) / 5, 4)}}))
ctx.close()
