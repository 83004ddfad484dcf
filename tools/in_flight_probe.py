#!/usr/bin/env python3
"""How much of the chip does ONE witness step leave idle?  The bench workload (P proofs at N lanes) as
  one      one context, one stream, P proofs per call                                  (the bench line's step)
  split2   two contexts on two streams, P/2 proofs each, enqueued side by side          (one step's work as two concurrent halves)
  split4   four contexts, P/4 each
  two      two contexts on two streams, P proofs each, alternating                      (two steps in flight: ms per step = pair / 2)
Every figure: ms per P proofs, host clock around enqueue + synchronize, best of INNER runs of 40 after 10 warm-up.
in_flight_probe.py [P] [N]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tendermintx_amd import Context, _lib  # noqa: E402
from tendermintx_amd.context import KIND_SKIP  # noqa: E402
from tendermintx_amd.synth import bench_workload  # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n = int(sys.argv[2]) if len(sys.argv) > 2 else 128
INNER = int(os.environ.get("INNER", "5"))
dev = torch.device("cuda:0")
stride = int(_lib.lib().tmx_elem_stride(KIND_SKIP, n))
w = bench_workload(os.environ.get("WORKLOAD", "survey8d"), n, P, seed=7)
psz, tsz = len(w.proofs) // P, len(w.targets) // P
rsz = len(w.trusteds) // P


class Part:
    def __init__(self, lo, hi):
        self.p = hi - lo
        self.d = [torch.frombuffer(bytearray(b[lo * z:hi * z]), dtype=torch.uint8).to(dev) for b, z in ((w.proofs, psz), (w.targets, tsz), (w.trusteds, rsz))]
        self.out = torch.empty(self.p * stride, dtype=torch.int64, device=dev)
        self.rep = torch.empty(self.p * 64, dtype=torch.uint8, device=dev)
        self.s = torch.cuda.Stream(dev)
        self.ctx = Context(n, b"celestia", 100800, device=0, max_batch=self.p)

    def go(self):
        self.ctx.witness_batch_device(KIND_SKIP, self.p, self.d[0].data_ptr(), self.d[1].data_ptr(), self.d[2].data_ptr(), self.out.data_ptr(),
                                      self.rep.data_ptr(), self.s.cuda_stream)

    def ok(self):
        return bool((self.rep.view(self.p, 64)[:, 32] == 1).all().item())


def measure(parts, per_iter_proofs):
    def run(k):
        for _ in range(k):
            for q in parts:
                q.go()
        torch.cuda.synchronize()
    best = 1e9
    for _ in range(INNER):
        run(10)
        t0 = time.perf_counter()
        run(40)
        best = min(best, 1e3 * (time.perf_counter() - t0) / 40)
    return best * P / per_iter_proofs, all(q.ok() for q in parts)


def split(k):
    return [Part(i * P // k, (i + 1) * P // k) for i in range(k)]


res = {}
res["one"] = measure(split(1), P)
res["split2"] = measure(split(2), P)
res["split4"] = measure(split(4), P)
res["two"] = measure([Part(0, P), Part(0, P)], 2 * P)
res["three"] = measure([Part(0, P), Part(0, P), Part(0, P)], 3 * P)
for k, (ms, ok) in res.items():
    print(f"{k:8s} {ms:.4f} ms per {P} proofs  all_ok {ok}", flush=True)
