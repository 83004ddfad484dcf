#!/usr/bin/env python3
"""Fused rows (TMX_FUSED_ROWS=<b>:<w>, layout.h FusedRows): bit parity of every configuration against the unfused build of the same library
(and the oracle on a sample), then the step time per configuration and batch size, alternating inside one process.
usage: fused_probe.py REPS CONFIG[,CONFIG...]   e.g.  fused_probe.py 3 0:0,4:0,8:0,4:2     (P list from $PS, default 256,512,1024)"""
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle", "py"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from tendermintx_amd import Context, _lib  # noqa: E402
from tendermintx_amd.context import KIND_SKIP  # noqa: E402
from tendermintx_amd.synth import bench_workload  # noqa: E402

reps = int(sys.argv[1])
cfgs = sys.argv[2].split(",")
Ps = [int(x) for x in os.environ.get("PS", "256,512,1024").split(",")]
n = 128
dev = torch.device("cuda:0")
stride = int(_lib.lib().tmx_elem_stride(KIND_SKIP, n))
stream = torch.cuda.Stream(dev)
up = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)

# ---- parity: 160 proofs (warm split schedule, s*B a launch of its own), every configuration against TMX_FUSED_ROWS=0:0 and the oracle
P = 160
w = bench_workload("survey8d", n, P, seed=11)
d = [up(b) for b in (w.proofs, w.targets, w.trusteds)]
ref = None
for cfg in ["0:0"] + [c for c in cfgs if c != "0:0"]:
    os.environ["TMX_FUSED_ROWS"] = cfg
    with Context(n, b"celestia", 100800, device=0, max_batch=P) as ctx:
        outs = []
        for it in range(4):   # cold, then warm three times (the counters alternate)
            o = torch.full((P * stride,), -1, dtype=torch.int64, device=dev)
            r = torch.zeros(P * 64, dtype=torch.uint8, device=dev)
            torch.cuda.synchronize(dev)
            ctx.witness_batch_device(KIND_SKIP, P, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), o.data_ptr(), r.data_ptr(), stream.cuda_stream)
            torch.cuda.synchronize(dev)
            outs.append(o)
        if ref is None:
            ref = outs[-1]
            import oracle_c as oc
            want, _ = oc.witness_batch(KIND_SKIP, 8, w.proofs[:8 * 2336], w.targets[:8 * n * 256], w.trusteds[:8 * n * 48], n, b"celestia", 100800, n_threads=8)
            cnt = ctx.elem_count(KIND_SKIP)
            got = ref.view(P, stride)[:8, :cnt].cpu().numpy().view(np.uint64)
            print("unfused == oracle on 8 proofs:", bool(np.array_equal(got, want)), flush=True)
        for it, o in enumerate(outs):
            cnt = ctx.elem_count(KIND_SKIP)
            same = torch.equal(o.view(P, stride)[:, :cnt], ref.view(P, stride)[:, :cnt])
            if not same:
                bad = (o.view(P, stride)[:, :cnt] != ref.view(P, stride)[:, :cnt]).nonzero()
                print(f"PARITY FAIL cfg {cfg} call {it}: {len(bad)} elements differ, first {bad[:5].tolist()}", flush=True)
                sys.exit(1)
    print(f"parity ok: TMX_FUSED_ROWS={cfg}", flush=True)

# ---- timing
for P in Ps:
    w = bench_workload("survey8d", n, P, seed=7)
    d = [up(b) for b in (w.proofs, w.targets, w.trusteds)]
    d_out = torch.empty(P * stride, dtype=torch.int64, device=dev)
    d_rep = torch.empty(P * 64, dtype=torch.uint8, device=dev)
    res = {c: [] for c in cfgs}
    kern = {}
    for r in range(reps):
        for cfg in cfgs:
            os.environ["TMX_FUSED_ROWS"] = cfg
            ctx = Context(n, b"celestia", 100800, device=0, max_batch=P)

            def run(k):
                for _ in range(k):
                    ctx.witness_batch_device(KIND_SKIP, P, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d_out.data_ptr(), d_rep.data_ptr(), stream.cuda_stream)
                torch.cuda.synchronize(dev)
            run(10)
            t0 = time.perf_counter()
            run(30)
            res[cfg].append(1e3 * (time.perf_counter() - t0) / 30)
            kern[cfg] = ctx.kernel_ms_mean(30)
            ctx.close()
    for cfg in cfgs:
        xs = res[cfg]
        print(f"P={P} TMX_FUSED_ROWS={cfg}: step mean {statistics.mean(xs):.4f} min {min(xs):.4f} max {max(xs):.4f}  {({k: round(x, 3) for k, x in kern[cfg].items()})}", flush=True)
