#!/usr/bin/env python3
"""Poseidon-Goldilocks Merkle commitment of LDE-sized columns: time, permutations/s, S-box field products/s, and the fraction of the
VALU issue roof (the kernel's bound: 8 B read per hashed element, ~2.8 k wave-instructions per permutation-wave).
usage: poseidon_bench.py [log_rows n_cols cap_height] ...   (default: the shapes quoted in DESIGN.md)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from tendermintx_amd import Context  # noqa: E402

SIMDS, CLOCK_GHZ, CYC = 1024, 2.4, 4.07          # bench.py: issue cost of a wave64 VALU instruction per SIMD (profiles/r03_valu_isa.txt)
INSTS_PER_PERM_WAVE = float(os.environ.get("POS_INSTS", "0")) or None   # from the PMC pass (tools/recipes.sh poseidon_prof); None: not reported
shapes = [(16, 256, 4), (19, 64, 4), (21, 64, 4), (21, 256, 4), (22, 16, 4)]
if len(sys.argv) > 3:
    a = [int(x) for x in sys.argv[1:]]
    shapes = [tuple(a[i:i + 3]) for i in range(0, len(a), 3)]
dev = torch.device("cuda", 0)
ctx = Context(4, b"celestia")
for log_n, n_cols, cap in shapes:
    cols = torch.randint(0, 2**62, (n_cols << log_n,), dtype=torch.int64, device=dev)
    nd = ctx.poseidon_merkle_digests(log_n, cap)
    lv = torch.empty((nd, 4), dtype=torch.int64, device=dev)
    for _ in range(2):
        ctx.poseidon_merkle_device(log_n, n_cols, cols.data_ptr(), cap, lv.data_ptr(), 0)
    torch.cuda.synchronize(dev)
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.poseidon_merkle_device(log_n, n_cols, cols.data_ptr(), cap, lv.data_ptr(), 0)
    torch.cuda.synchronize(dev)
    ms = 1e3 * (time.perf_counter() - t0) / reps
    rows = 1 << log_n
    perms = rows * ((n_cols + 7) // 8 if n_cols > 4 else 0) + (rows - (1 << cap))
    res = {"log_rows": log_n, "n_cols": n_cols, "cap_height": cap, "ms": round(ms, 4), "permutations": perms,
           "mperm_per_s": round(perms / ms / 1e3, 1), "sbox_field_mul_per_s": round(perms * 118 * 4 / (ms * 1e-3), 0),
           "hashed_bytes": 8 * rows * n_cols, "read_gbs": round(8 * rows * n_cols / (ms * 1e-3) / 1e9, 1)}
    if INSTS_PER_PERM_WAVE:
        waves = perms / 64
        res["valu_issue_frac"] = round(waves * INSTS_PER_PERM_WAVE * CYC / (SIMDS * ms * 1e-3 * CLOCK_GHZ * 1e9), 4)
    print(json.dumps(res), flush=True)
    del cols, lv
ctx.close()
