"""world_size-2 gloo tests (CPU) of the N > 1 path: proof sharding, row reassembly, and the validator-sharded EdDSA
exchange.  The compute callable is injected: here the oracle stands in for k_eddsa (the product default needs a GPU)."""
import os
import struct
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _ed_record(tr):
    return tr["digest"] + tr["h"] + b"".join(tr["pt"]) + struct.pack("<II", int(tr["ok"]), int(tr["decode_ok"])) + bytes(24)


def _worker(rank, world, port, q):
    for p in (ROOT, os.path.join(ROOT, "oracle", "py")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle_c as oc
        from tendermintx_amd import sharding
        from tendermintx_amd.synth import Workload
        # ---- shard_range covers everything exactly once, sizes differ by <= 1
        for n in (1, 5, 37, 128, 257):
            spans = [sharding.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        # ---- proof-sharded batch: each rank computes its slice, gather_rows reassembles the full batch everywhere
        n, P = 8, 5
        wl = Workload(0, n, P, 7, chain_id=b"celestia", seed=99, signed_permille=900)
        lo, hi = sharding.shard_range(P, rank, world)
        elems, reps = oc.witness_batch(0, hi - lo, wl.proofs[lo * 2336:hi * 2336], wl.targets[lo * n * 256:hi * n * 256],
                                       wl.trusteds[lo * n * 48:hi * n * 48], n, b"celestia", 100800)
        full = sharding.gather_rows(torch.from_numpy(elems.astype(np.int64)), P)
        want, _ = oc.witness_batch(0, P, wl.proofs, wl.targets, wl.trusteds, n, b"celestia", 100800)
        assert np.array_equal(full.numpy().view(np.uint64), want)
        # ---- validator-sharded single proof: EdDSA lanes split, one all-gather, identical records on every rank
        n = 13  # odd on purpose: uneven shards
        wl = Workload(0, n, 1, 11, chain_id=b"celestia", seed=5, signed_permille=800)
        lanes = torch.frombuffer(bytearray(wl.targets), dtype=torch.uint8).view(n, 256)
        dpk, dsig = oc.dummy()
        calls = []

        def eddsa_fn(sl):
            calls.append(sl.shape[0])
            out = []
            for row in sl.numpy():
                b = row.tobytes()
                if b[223] & 1:
                    out.append(_ed_record(oc.eddsa_trace(b[:32], b[32:96], b[96:96 + struct.unpack_from("<H", b, 220)[0]])))
                else:
                    out.append(_ed_record(oc.eddsa_trace(dpk, dsig, bytes(32))))
            return torch.frombuffer(bytearray(b"".join(out)), dtype=torch.uint8).view(len(out), 448)

        ed = sharding.validator_sharded_eddsa(lanes, eddsa_fn)
        lo, hi = sharding.shard_range(n, rank, world)
        assert calls == [hi - lo]
        want_ed = eddsa_fn(lanes)
        assert torch.equal(ed, want_ed)
        # every lane of the gathered records verifies (ok word at byte 416)
        assert all(int.from_bytes(bytes(r[416:420].tolist()), "little") == 1 for r in ed)
        q.put((rank, "ok"))
    except Exception as e:  # surface the failure in the parent
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc()))
        raise e
    finally:
        dist.destroy_process_group()


def test_sharding_world_size_2(oracle):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(results) == [(0, "ok"), (1, "ok")], results
