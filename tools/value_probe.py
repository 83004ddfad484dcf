"""Host-to-host latency of the typed-value entry point (tmx_inputs_value_batch) beside the element-row entry points: one proof and the
256-proof batch at N = 128, pageable and page-locked buffers.  `TMX_VALUE_DIRECT_MAX=<bytes>` (read once per process) bounds the size up to
which a page-locked `out` is written by the device itself instead of through device staging + one copy.
usage: python tools/value_probe.py [n_max] [proofs]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402  (the HIP runtime the library runs on)

from tendermintx_amd import KIND_SKIP, Context, _lib  # noqa: E402
from tendermintx_amd.synth import bench_workload  # noqa: E402


def med(xs):
    xs = sorted(xs)
    return xs[len(xs) // 2]


def timeit(fn, reps, warm=3):
    for _ in range(warm):
        fn()
    xs = []
    for _ in range(reps):
        a = time.perf_counter()
        fn()
        xs.append(1e3 * (time.perf_counter() - a))
    return round(med(xs), 4), round(min(xs), 4)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    P = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    wl = bench_workload("survey8d", n, P, seed=0x544D58)
    out = {"n": n, "proofs": P, "direct_max": os.environ.get("TMX_VALUE_DIRECT_MAX", "default")}
    with Context(n, b"celestia", 100800, max_batch=P) as ctx:
        layh, laya = ctx.value_layout(KIND_SKIP, _lib.SEC_HINT), ctx.value_layout(KIND_SKIP, _lib.SEC_ALL)
        out["value_bytes"] = {"hint": int(layh.bytes), "all": int(laya.bytes), "row_u64": int(ctx.elem_stride(KIND_SKIP) * 8)}
        pin_out = ctx.host_alloc(P * laya.bytes)
        pp, pt, pr = ctx.host_alloc(len(wl.proofs)), ctx.host_alloc(len(wl.targets)), ctx.host_alloc(len(wl.trusteds))
        pp[:] = np.frombuffer(wl.proofs, dtype=np.uint8); pt[:] = np.frombuffer(wl.targets, dtype=np.uint8); pr[:] = np.frombuffer(wl.trusteds, dtype=np.uint8)
        page_out = np.zeros(P * laya.bytes, dtype=np.uint8)
        ctx.inputs_value_batch(KIND_SKIP, pp, pt, pr, _lib.SEC_ALL, out=pin_out)     # warm the key cache with the whole batch's keys
        for name, k in (("single", 1), ("batch", P)):
            a, b, c = pp[:2336 * k], pt[:256 * n * k], pr[:48 * n * k]
            ab, bb, cb = wl.proofs[:2336 * k], wl.targets[:256 * n * k], wl.trusteds[:48 * n * k]
            reps = 40 if k == 1 else 8
            r = {}
            r["value_hint_pinned"] = timeit(lambda: ctx.inputs_value_batch(KIND_SKIP, a, b, c, _lib.SEC_HINT, out=pin_out), reps)
            r["value_all_pinned"] = timeit(lambda: ctx.inputs_value_batch(KIND_SKIP, a, b, c, _lib.SEC_ALL, out=pin_out), reps)
            r["value_hint_pageable"] = timeit(lambda: ctx.inputs_value_batch(KIND_SKIP, ab, bb, cb, _lib.SEC_HINT, out=page_out), reps)
            r["value_hint_pageable_in_pinned_out"] = timeit(lambda: ctx.inputs_value_batch(KIND_SKIP, ab, bb, cb, _lib.SEC_HINT, out=pin_out), reps)
            # device-resident: the Level-1 kernels + k_pack_value on torch's stream, buffers in HBM
            dev = torch.device("cuda:0")
            d = lambda x: torch.frombuffer(bytearray(x), dtype=torch.uint8).to(dev)
            dp, dt_, dr = d(ab), d(bb), d(cb)
            dout = torch.zeros(k * laya.bytes, dtype=torch.uint8, device=dev)
            s = int(torch.cuda.current_stream().cuda_stream)

            def devcall(sec):
                ctx.inputs_value_batch_device(KIND_SKIP, k, dp.data_ptr(), dt_.data_ptr(), dr.data_ptr(), dout.data_ptr(), sec, stream=s)
                torch.cuda.current_stream().synchronize()
            r["value_hint_device"] = timeit(lambda: devcall(_lib.SEC_HINT), reps)
            r["value_all_device"] = timeit(lambda: devcall(_lib.SEC_ALL), reps)
            if k == 1:
                rows = torch.empty(ctx.elem_stride(KIND_SKIP), dtype=torch.int64, pin_memory=True).numpy().view(np.uint64)
                r["row_u64_pinned_out"] = timeit(lambda: ctx.witness_batch(KIND_SKIP, ab, bb, cb, out=rows), reps)
                r["row_hint_u32_pinned_out"] = timeit(lambda: ctx.witness_batch_hint(KIND_SKIP, ab, bb, cb, out=rows), reps)
            out[name] = {kk: {"median_ms": v[0], "min_ms": v[1]} for kk, v in r.items()}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
