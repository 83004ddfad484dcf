#!/usr/bin/env python3
"""Where and when the waves of one step ran, SIMD by SIMD (a -DTMX_WAVE_PROF build: `bash tools/build_variant.sh waveprof -DTMX_WAVE_PROF`,
then `TMX_LIB=$PWD/build_ab/waveprof.so python tools/wave_prof.py`).  Every wave of the instrumented kernels (1 walk, 2 hash, 3 s*B, 4 finish,
5 k_proof, 6 k_serialize_few) records HW_ID / XCC_ID and s_memrealtime at its first and last instruction; this prints, per kernel, the
distribution of wave lifetimes, when the first / median / last wave ended, and how the walk's waves fared on SIMDs they shared with a
k_proof wave against SIMDs they did not (P, N, WORKLOAD from the environment)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from tendermintx_amd import Context, _lib  # noqa: E402
from tendermintx_amd.context import KIND_SKIP  # noqa: E402
from tendermintx_amd.synth import bench_workload  # noqa: E402

NAMES = {1: "walk", 2: "hash", 3: "s*B", 4: "finish", 5: "k_proof", 6: "ser_few"}
P, n = int(os.environ.get("P", "256")), int(os.environ.get("N", "128"))
w = bench_workload(os.environ.get("WORKLOAD", "survey8d"), n, P, seed=7)
dev = torch.device("cuda:0")
up = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
d_proofs, d_targets, d_trusteds = up(w.proofs), up(w.targets), up(w.trusteds)
L = _lib.lib()
stride = int(L.tmx_elem_stride(KIND_SKIP, n))
d_out = torch.empty(P * stride, dtype=torch.int64, device=dev)
d_rep = torch.empty(P * 64, dtype=torch.uint8, device=dev)
stream = torch.cuda.Stream(dev)
ctx = Context(n, b"celestia", 100800, device=0, max_batch=P)


MODE = os.environ.get("MODE", "full")   # full | noser (no witness rows: no serializer) | ed (the EdDSA stage alone)
d_ed = torch.empty(P * n * 448, dtype=torch.uint8, device=dev)


def step(k=1):
    for _ in range(k):
        if MODE == "ed":
            ctx.eddsa_lanes_device(P * n, d_targets.data_ptr(), d_ed.data_ptr(), stream.cuda_stream)
        else:
            ctx.witness_batch_device(KIND_SKIP, P, d_proofs.data_ptr(), d_targets.data_ptr(), d_trusteds.data_ptr(),
                                     None if MODE == "noser" else d_out.data_ptr(), d_rep.data_ptr(), stream.cuda_stream)
    torch.cuda.synchronize(dev)


prof = L.tmx_debug_wave_prof
prof.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.c_int]
prof.restype = C.c_int
step(12)
cnt = C.c_uint32(0)
assert prof(None, 0, C.byref(cnt), 1) == 0   # reset
step(1)
buf = np.zeros((1 << 16, 8), dtype=np.uint32)
assert prof(buf.ctypes.data, buf.shape[0], C.byref(cnt), 1) == 0
r = buf[:cnt.value]
tag, hw, xcc = r[:, 0], r[:, 1], r[:, 2] & 0xf
t0 = r[:, 4].astype(np.uint64) | (r[:, 5].astype(np.uint64) << np.uint64(32))
t1 = r[:, 6].astype(np.uint64) | (r[:, 7].astype(np.uint64) << np.uint64(32))
simd = (xcc.astype(np.uint64) << np.uint64(16)) | (((hw >> 13) & 7).astype(np.uint64) << np.uint64(12)) | (((hw >> 12) & 1).astype(np.uint64) << np.uint64(11)) \
    | (((hw >> 8) & 0xf).astype(np.uint64) << np.uint64(4)) | ((hw >> 4) & 3).astype(np.uint64)
base = t0.min()
us_per_tick = 1 / 100.0   # s_memrealtime: the constant 100-MHz counter
print(f"MODE={MODE}: {cnt.value} wave records; recorded span {float(t1.max() - base) * us_per_tick:.1f} us")
for t in sorted(NAMES):
    m = tag == t
    if not m.any():
        continue
    s0, e0 = (t0[m] - base) * us_per_tick, (t1[m] - base) * us_per_tick
    life = e0 - s0
    print(f"{NAMES[t]:8s} waves {m.sum():5d}  start {s0.min():7.1f} .. {np.median(s0):7.1f} .. {s0.max():7.1f}   end {e0.min():7.1f} .. {np.median(e0):7.1f} .. {e0.max():7.1f}   "
          f"life min {life.min():6.1f} med {np.median(life):6.1f} p90 {np.percentile(life, 90):6.1f} max {life.max():6.1f} us;  SIMDs used {len(set(simd[m]))}")
# the walk's waves on SIMDs shared with k_proof / s*B waves (overlapping in time) against the others
mw = tag == 1
for other, name in ((5, "k_proof"), (3, "s*B"), (6, "ser_few")):
    mo = tag == other
    if not mw.any() or not mo.any():
        continue
    by_simd = {}
    for s_, a, b in zip(simd[mo], t0[mo], t1[mo]):
        by_simd.setdefault(int(s_), []).append((a, b))
    shared, alone = [], []
    for s_, a, b in zip(simd[mw], t0[mw], t1[mw]):
        ov = any(x < b and a < y for x, y in by_simd.get(int(s_), ()))
        (shared if ov else alone).append(float(b - a) * us_per_tick)
    if shared and alone:
        print(f"walk waves sharing a SIMD with a {name:8s} wave: {len(shared):5d}, life med {np.median(shared):6.1f} p90 {np.percentile(shared, 90):6.1f} us | "
              f"not sharing: {len(alone):5d}, life med {np.median(alone):6.1f} p90 {np.percentile(alone, 90):6.1f} us")
# occupancy picture: waves per SIMD of each kernel at a few instants of the walk
if mw.any():
    ws, we = float((t0[mw].min() - base)) * us_per_tick, float((t1[mw].max() - base)) * us_per_tick
    for frac in (0.1, 0.5, 0.9):
        at = base + np.uint64((ws + frac * (we - ws)) / us_per_tick)
        live = (t0 <= at) & (t1 > at)
        print(f"at {ws + frac * (we - ws):6.1f} us: " + ", ".join(f"{NAMES[t]} {int((live & (tag == t)).sum())}" for t in sorted(NAMES) if (tag == t).any()) +
              f" waves resident; busiest SIMD holds {np.bincount(np.unique(simd[live], return_inverse=True)[1]).max() if live.any() else 0}")
ctx.close()
