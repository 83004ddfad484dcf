"""Two modes.  `run <new_keys>`: warm the cache with the bench batch, then ONE 256-proof step with that many new keys (run it under
rocprofv3 --kernel-trace).  `show <results.db>`: the kernel timeline of that last step (from its k_proof launch on)."""
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(new_keys):
    import torch
    from tendermintx_amd import KIND_SKIP, Context
    from tendermintx_amd.synth import Workload, bench_workload
    P, n = 256, 128
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream(dev)
    wl = bench_workload("survey8d", n, P, seed=0x544D58)
    up = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
    base = tuple(up(b) for b in (wl.proofs, wl.targets, wl.trusteds))
    ctx = Context(n, b"celestia", 100800, max_batch=P)
    d_out = torch.empty((P, ctx.elem_stride(KIND_SKIP)), dtype=torch.int64, device=dev)
    d_rep = torch.zeros(P * 64, dtype=torch.uint8, device=dev)

    def step(bufs):
        dp, dt, dr = bufs
        ctx.witness_batch_device(KIND_SKIP, P, dp.data_ptr(), dt.data_ptr(), dr.data_ptr(), d_out.data_ptr(), d_rep.data_ptr(), stream.cuda_stream)
    for _ in range(20):
        step(base)
    torch.cuda.synchronize(dev)
    bufs = base
    if new_keys:
        wj = Workload(0, n, 1, new_keys, chain_id=b"celestia", seed=0x700123 + new_keys, signed_permille=1000)
        bufs = tuple(up(a + b[len(a):]) for a, b in ((wj.proofs, wl.proofs), (wj.targets, wl.targets), (wj.trusteds, wl.trusteds)))
    torch.cuda.synchronize(dev)
    step(bufs)
    torch.cuda.synchronize(dev)
    print("last_new_keys", ctx.key_cache_stats()["last_new_keys"])
    ctx.close()


def show(path):
    db = sqlite3.connect(path)
    rows = db.execute("select name,start,end,stream_id from kernels order by start").fetchall()
    starts = [i for i, r in enumerate(rows) if "k_proof" in r[0]]
    i0 = starts[-1]
    t0 = min(r[1] for r in rows[i0:])
    for r in rows[i0:]:
        print(f"{r[0][:34]:34s} start {((r[1]-t0)/1e3):8.1f}  end {((r[2]-t0)/1e3):8.1f}  dur {((r[2]-r[1])/1e3):7.1f} us  stream {r[3]}")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(int(sys.argv[2]))
    else:
        show(sys.argv[2])
