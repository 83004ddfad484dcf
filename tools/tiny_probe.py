#!/usr/bin/env python3
"""Small-launch latency probe: isolated calls (device idle before, host clock from the enqueue to the end of the caller's stream) at
P proofs x N lanes, warm and cold (after tmx_key_cache_flush), bit-compared with the CPU oracle.  TMX_TINY=0 gives the classic launch
graph of the same build:   P=1 N=128 python tools/tiny_probe.py ;  TMX_TINY=0 P=1 python tools/tiny_probe.py"""
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle", "py"))
import numpy as np
import torch
import oracle_c as oc  # checker only
from tendermintx_amd import Context, _lib
from tendermintx_amd.context import KIND_SKIP
from tendermintx_amd.synth import bench_workload

P, n = int(os.environ.get("P", "1")), int(os.environ.get("N", "128"))
w = bench_workload(os.environ.get("WORKLOAD", "survey8d"), n, P, seed=0x544D58)
dev = torch.device("cuda:0")
d = [torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev) for b in (w.proofs, w.targets, w.trusteds)]
stride, count = int(_lib.lib().tmx_elem_stride(KIND_SKIP, n)), int(_lib.lib().tmx_elem_count(KIND_SKIP, n))
out = torch.zeros(P * stride, dtype=torch.int64, device=dev)
rep = torch.zeros(P * 64, dtype=torch.uint8, device=dev)
s = torch.cuda.current_stream(dev)
ctx = Context(n, b"celestia", 100800, device=0, max_batch=P)


def call():
    torch.cuda.synchronize(dev)
    a = time.perf_counter()
    ctx.witness_batch_device(KIND_SKIP, P, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), out.data_ptr(), rep.data_ptr(), s.cuda_stream)
    s.synchronize()
    b = time.perf_counter()
    torch.cuda.synchronize(dev)
    return 1e3 * (b - a)


def parity(tag):
    want, _ = oc.witness_batch(KIND_SKIP, P, w.proofs, w.targets, w.trusteds, n, b"celestia", 100800, n_threads=8)
    got = out.view(P, stride)[:, :count].cpu().numpy().view(np.uint64)
    ok = np.array_equal(got, want)
    print(f"parity {tag}: {ok}", flush=True)
    if not ok:
        bad = np.argwhere(got != want)
        print("  first mismatches (proof, element):", bad[:12].tolist(), "count", len(bad))
    return ok


QUICK = os.environ.get("QUICK") == "1"   # warm timings only, no parity (profiling builds whose outputs are wrong on purpose)
if QUICK:
    parity = lambda tag: True
cold = []
for _ in range(0 if QUICK else 8):
    ctx.key_cache_flush()
    out.zero_()
    cold.append(call())
cold = cold or [0.0]
ok = parity("cold")
for _ in range(5):
    call()
out.zero_()
warm = [call() for _ in range(40)]
ok = parity("warm") and ok
st = ctx.key_cache_stats()
print(f"DBG={os.environ.get('TMX_TINY_DBG', '0')} P={P} N={n} TMX_TINY={os.environ.get('TMX_TINY', 'default')}: warm median {statistics.median(warm):.4f} ms (min {min(warm):.4f}), "
      f"cold median {statistics.median(cold):.4f} ms; kernels {ctx.kernel_ms_mean(20)}; cache {st['last_new_keys']} new / {st['last_hit_lanes']} hit lanes",
      flush=True)
ctx.close()
sys.exit(0 if ok else 1)
