"""One rank of the collective-abort test (tests/test_world2_one_gpu.py): two processes on cuda:0 joined through the fake RCCL.  Rank 1's context
is too small for its shard (max_batch 2, shard 4): tmx_witness_batch_sharded_device(gather = 1) fails LOCALLY on rank 1 in front of the
exchange.  The contract (include/tmx.h "FAILURE CONTRACT"): rank 1 aborts its communicator and returns TMX_ERR_RCCL naming the local cause;
rank 0, which computed its shard and entered the exchange, returns TMX_ERR_RCCL too instead of waiting for ever; both contexts refuse sharded
calls until tmx_comm_create is called again; then a batch that fits (4 proofs, 2 per rank) runs and every row equals the oracle's.
TEST INFRASTRUCTURE.  usage: abort_worker.py <rank> <world> <outdir>"""
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "oracle", "py"), os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def exchange_id(rank, outdir, tag):
    from tendermintx_amd import sharding
    path = os.path.join(outdir, f"id_{tag}.bin")
    if rank == 0:
        u = sharding.unique_id()
        with open(path + ".tmp", "wb") as f:
            f.write(u)
        os.replace(path + ".tmp", path)
        return u
    t0 = time.time()
    while not os.path.exists(path):
        if time.time() - t0 > 120:
            raise TimeoutError(tag)
        time.sleep(0.01)
    return open(path, "rb").read()


def main():
    rank, world, outdir = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    import numpy as np
    import torch
    import oracle_c as oc
    import tendermintx_amd as tmx
    from tendermintx_amd import sharding
    from tendermintx_amd._lib import TmxError
    from tendermintx_amd.synth import Workload
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    up = lambda *bs: [torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev) for b in bs]
    n, P = 32, 8
    wl = Workload(0, n, P, n - 3, chain_id=b"celestia", seed=777, signed_permille=900)
    d = up(wl.proofs, wl.targets, wl.trusteds)
    log = []
    with tmx.Context(n, b"celestia", device=0, max_batch=4 if rank == 0 else 2) as ctx:
        ctx.comm_create(exchange_id(rank, outdir, "a"), rank, world)
        stride, count = ctx.elem_stride(0), ctx.elem_count(0)
        out = torch.zeros((P, stride), dtype=torch.int64, device=dev)
        rep = torch.zeros(P * 64, dtype=torch.uint8, device=dev)
        t0 = time.time()
        try:
            sharding.proof_sharded_batch(ctx, 0, P, d[0], d[1], d[2], out, rep, gather=True)
            ctx.comm_sync(int(torch.cuda.current_stream().cuda_stream), timeout_ms=60000)
            raise AssertionError(f"rank {rank}: the sharded call succeeded although rank 1's shard does not fit its context")
        except TmxError as e:
            dt = time.time() - t0
            assert e.status == -7, f"rank {rank}: status {e.status} ({e})"
            assert dt < 60, f"rank {rank} needed {dt:.1f} s to leave the collective"
            if rank == 1:
                assert "local failure in front of a collective" in str(e) and "max_batch" in str(e), str(e)
            log.append(f"rank {rank} left the failed collective after {dt:.2f} s: {str(e)[:120]}")
        torch.cuda.synchronize(dev)
        # the context refuses sharded calls until there is a new communicator ...
        try:
            sharding.proof_sharded_batch(ctx, 0, 4, d[0], d[1], d[2], out, rep, gather=True)
            raise AssertionError("a sharded call on an aborted communicator must fail")
        except TmxError as e:
            assert e.status == -7 and "aborted" in str(e), str(e)
        # ... while the non-sharded entry points still work
        e1, _ = ctx.witness_batch(0, wl.proofs[:2336], wl.targets[:256 * n], wl.trusteds[:48 * n])
        want, oreps = oc.witness_batch(0, P, wl.proofs, wl.targets, wl.trusteds, n, b"celestia", 100800, n_threads=4)
        assert np.array_equal(e1[0], want[0])
        # ... and a new communicator makes it whole again: 4 proofs, 2 per rank, gathered
        ctx.comm_create(exchange_id(rank, outdir, "b"), rank, world)
        out.zero_(); rep.zero_()
        sharding.proof_sharded_batch(ctx, 0, 4, d[0], d[1], d[2], out, rep, gather=True)
        ctx.comm_sync(int(torch.cuda.current_stream().cuda_stream), timeout_ms=60000)
        got = out[:4, :count].cpu().numpy().view(np.uint64)
        assert np.array_equal(got, want[:4]), f"rows after the recovery differ from the oracle on rank {rank}"
        log.append("new communicator: 4 proofs sharded + gathered, bit-exact")
    return log


if __name__ == "__main__":
    rank, outdir = int(sys.argv[1]), sys.argv[3]
    try:
        msg = "ok\n" + "\n".join(main())
    except BaseException:
        msg = "FAIL\n" + traceback.format_exc()
    with open(os.path.join(outdir, f"rank{rank}.txt"), "w") as f:
        f.write(msg)
    sys.exit(0 if msg.startswith("ok") else 1)
