"""The HIP path under more than one RCCL rank (one process per GPU).  Needs >= 2 GPUs in the box: skipped on the 1-GPU boxes the
round-end test tier runs on; the exchange logic itself is covered on CPU by tests/test_sharding_gloo.py (gloo, world_size 2)."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, q):
    for p in (ROOT, os.path.join(ROOT, "oracle", "py")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    dev = torch.device("cuda", rank)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        import oracle_c as oc
        import tendermintx_amd as tmx
        from tendermintx_amd import sharding
        from tendermintx_amd.synth import Workload
        # ---- BASELINE configs[3]: one batch sharded over the ranks through the C entry point (tmx_witness_batch_sharded_device: slices in
        # place, one grouped RCCL exchange), bit-exact vs the oracle on every rank
        n, P = 32, 11
        wl = Workload(0, n, P, 29, chain_id=b"celestia", seed=4321, signed_permille=900)
        d = [torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev) for b in (wl.proofs, wl.targets, wl.trusteds)]
        lo, hi = sharding.shard_range(P, rank, world)
        with tmx.Context(n, b"celestia", device=rank, max_batch=hi - lo) as ctx:
            assert sharding.connect(ctx) == (rank, world) and ctx.comm_info() == (rank, world)
            out = torch.zeros((P, ctx.elem_stride(0)), dtype=torch.int64, device=dev)
            rep = torch.zeros(P * 64, dtype=torch.uint8, device=dev)
            sharding.proof_sharded_batch(ctx, 0, P, d[0], d[1], d[2], out, rep, gather=True)
            torch.cuda.synchronize(dev)
            count = ctx.elem_count(0)
            want, oreps = oc.witness_batch(0, P, wl.proofs, wl.targets, wl.trusteds, n, b"celestia", 100800, n_threads=4)
            assert np.array_equal(out[:, :count].cpu().numpy().view(np.uint64), want)
            assert [bool(rep.cpu().numpy()[64 * p + 32]) for p in range(P)] == [bool(r["all_ok"]) for r in oreps]
            # without the gather only this rank's rows are written
            out.zero_()
            sharding.proof_sharded_batch(ctx, 0, P, d[0], d[1], d[2], out, rep, gather=False)
            torch.cuda.synchronize(dev)
            got = out[:, :count].cpu().numpy().view(np.uint64)
            assert np.array_equal(got[lo:hi], want[lo:hi]) and not got[:lo].any() and not got[hi:].any()
        # ---- BASELINE configs[4]: one proof, validator lanes sharded, one exchange of the EdDSA lane records (tmx_witness_validator_sharded_device)
        n = 64
        wl = Workload(0, n, 1, 50, chain_id=b"celestia", seed=99, signed_permille=900)
        d = [torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev) for b in (wl.proofs, wl.targets, wl.trusteds)]
        with tmx.Context(n, b"celestia", device=rank, max_batch=1) as ctx:
            sharding.connect(ctx)
            elems, rep = sharding.validator_sharded_skip(ctx, 0, d[0], d[1], d[2])
            torch.cuda.synchronize(dev)
            got = elems.cpu().numpy().view(np.uint64)
        want, orep = oc.witness(0, wl.proofs, wl.targets, wl.trusteds, b"celestia", 100800)
        assert np.array_equal(got, want) and orep["all_ok"] and bytes(rep.cpu().numpy()[:32]) == orep["header"]
        q.put((rank, "ok"))
    except Exception:
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc()))
        raise
    finally:
        dist.destroy_process_group()


def test_hip_path_under_two_rccl_ranks(built_lib, oracle):
    import torch
    import torch.multiprocessing as mp
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs (the driver's 8-GPU node); exchange logic covered by tests/test_sharding_gloo.py")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + (os.getpid() % 200)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(results) == [(0, "ok"), (1, "ok")], results
