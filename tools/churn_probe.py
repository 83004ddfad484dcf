"""key_cache.churn of bench.py on its own, as an in-process A/B over one environment knob (fresh context per configuration, the
configurations alternating, so that clock / box differences cancel): one 256-proof step in which exactly k of the batch's 401 distinct keys
are new to the cache, the schedule hint saying warm.
usage: python tools/churn_probe.py [KEY=V1,V2,...] [rounds]      e.g.  TMX_WALK_SPLIT=1,0 3"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from tendermintx_amd import KIND_SKIP, Context  # noqa: E402
from tendermintx_amd.synth import Workload, bench_workload  # noqa: E402


def med(xs):
    xs = sorted(xs)
    return xs[len(xs) // 2]


def main():
    key, vals = (sys.argv[1].split("=") + [""])[:2] if len(sys.argv) > 1 and "=" in sys.argv[1] else ("TMX_NONE", "x")
    vals = vals.split(",")
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    P, n = 256, 128
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream(dev)
    wl = bench_workload("survey8d", n, P, seed=0x544D58)
    up = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
    base = tuple(up(b) for b in (wl.proofs, wl.targets, wl.trusteds))
    acc = {v: {} for v in vals}
    d_out = d_rep = None
    for rnd in range(rounds):
        for v in vals:
            os.environ[key] = v
            ctx = Context(n, b"celestia", 100800, max_batch=P)
            if d_out is None:
                d_out = torch.empty((P, ctx.elem_stride(KIND_SKIP)), dtype=torch.int64, device=dev)
                d_rep = torch.zeros(P * 64, dtype=torch.uint8, device=dev)

            def run(k, bufs=None):
                dp, dt, dr = bufs or base
                for _ in range(k):
                    ctx.witness_batch_device(KIND_SKIP, P, dp.data_ptr(), dt.data_ptr(), dr.data_ptr(), d_out.data_ptr(), d_rep.data_ptr(), stream.cuda_stream)

            def timed(k, bufs=None):
                torch.cuda.synchronize(dev)
                a = time.perf_counter()
                run(k, bufs)
                stream.synchronize()
                b = time.perf_counter()
                torch.cuda.synchronize(dev)
                return 1e3 * (b - a) / k

            run(30)
            acc[v].setdefault("warm", []).append(timed(40))
            for new_keys, reps in ((0, 8), (1, 8), (4, 8), (40, 8), (401, 6)):  # (0: the same single step behind a synchronize, no new key)
                xs = []
                for j in range(reps):
                    if new_keys == 0:
                        bufs = None
                    elif new_keys == 401:
                        wj = bench_workload("survey8d", n, P, seed=0x600000 + 977 * j + 13 * rnd)
                        bufs = tuple(up(b) for b in (wj.proofs, wj.targets, wj.trusteds))
                    else:
                        wj = Workload(0, n, 1, new_keys, chain_id=b"celestia", seed=0x700000 + 31 * j + new_keys + 1000 * rnd, signed_permille=1000)
                        bufs = tuple(up(a + b[len(a):]) for a, b in ((wj.proofs, wl.proofs), (wj.targets, wl.targets), (wj.trusteds, wl.trusteds)))
                    run(2)
                    xs.append(timed(1, bufs))
                acc[v].setdefault(str(new_keys if new_keys != 401 else 400), []).append(med(xs))
            run(3)
            cold = []
            for _ in range(6):
                ctx.key_cache_flush()
                cold.append(timed(1))
            acc[v].setdefault("cold", []).append(med(cold))
            ok = int(d_rep.cpu().numpy().reshape(-1, 64)[:, 32:36].copy().view("uint32").sum())
            assert ok == P
            ctx.close()
    for v in vals:
        print(json.dumps({"config": f"{key}={v}", **{k: round(med(x), 4) for k, x in acc[v].items()}, "all": {k: [round(y, 4) for y in x] for k, x in acc[v].items()}}))


if __name__ == "__main__":
    main()
