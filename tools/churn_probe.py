"""key_cache.churn of bench.py on its own: one 256-proof step in which exactly k of the batch's 401 distinct keys are new to the cache, the
schedule hint saying warm.  Rows checked against nothing here (tests/test_key_cache.py does that); this is the timing tool.
usage: python tools/churn_probe.py [proofs]      (TMX_WALK_SPLIT=0: the round-4 schedule)"""
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
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    n = 128
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream(dev)
    wl = bench_workload("survey8d", n, P, seed=0x544D58)
    up = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
    base = tuple(up(b) for b in (wl.proofs, wl.targets, wl.trusteds))
    ctx = Context(n, b"celestia", 100800, max_batch=P)
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
    out = {"proofs": P, "walk_split": os.environ.get("TMX_WALK_SPLIT", "1"), "warm": round(timed(30), 4)}
    churn = {}
    for new_keys, reps in ((0, 12), (1, 12), (4, 12), (40, 12), (401, 8)):
        xs = []
        seen = 0
        for j in range(reps):
            if new_keys == 0:
                bufs = None
            elif new_keys == 401:
                wj = bench_workload("survey8d", n, P, seed=0x600000 + 977 * j)
                bufs = tuple(up(b) for b in (wj.proofs, wj.targets, wj.trusteds))
            else:
                wj = Workload(0, n, 1, new_keys, chain_id=b"celestia", seed=0x700000 + 31 * j + new_keys, signed_permille=1000)
                bufs = tuple(up(a + b[len(a):]) for a, b in ((wj.proofs, wl.proofs), (wj.targets, wl.targets), (wj.trusteds, wl.trusteds)))
            run(2)
            xs.append(timed(1, bufs))
            seen = ctx.key_cache_stats()["last_new_keys"]
        churn[str(seen)] = round(med(xs), 4)
    out["ms_per_step_by_new_keys"] = churn
    run(3)
    ctx.key_cache_flush()
    cold = []
    for _ in range(8):
        ctx.key_cache_flush()
        cold.append(timed(1))
    out["cold"] = round(med(cold), 4)
    ok = int(d_rep.cpu().numpy().reshape(-1, 64)[:, 32:36].copy().view("uint32").sum())
    out["all_ok_last"] = ok == P
    print(json.dumps(out))
    ctx.close()


if __name__ == "__main__":
    main()
