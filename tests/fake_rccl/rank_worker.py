"""One rank of the world > 1 run on a single GPU (tests/test_world2_one_gpu.py starts `world` of these): every rank is a process of its own
on cuda:0 with a tmx context of its own, joined through tmx_comm_create with the fake RCCL of this directory (TMX_RCCL_LIB), and drives
the multi-GPU C entry points exactly as tests/test_multi_gpu.py does on a real multi-GPU box.  Every rank checks EVERY row against the CPU
oracle and writes "ok" or a traceback to <outdir>/rank<r>.txt.  TEST INFRASTRUCTURE.
usage: rank_worker.py <rank> <world> <outdir>"""
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "oracle", "py")):
    if p not in sys.path:
        sys.path.insert(0, p)


def bootstrap_id(rank, outdir):
    """rank 0 makes the id and leaves it in a file (any channel would do: include/tmx.h "multi-GPU"); the others wait for it"""
    from tendermintx_amd import sharding
    path = os.path.join(outdir, "unique_id.bin")
    if rank == 0:
        uid = sharding.unique_id()
        with open(path + ".tmp", "wb") as f:
            f.write(uid)
        os.replace(path + ".tmp", path)
        return uid
    t0 = time.time()
    while not os.path.exists(path):
        if time.time() - t0 > 120:
            raise TimeoutError("rank 0 never published the unique id")
        time.sleep(0.01)
    return open(path, "rb").read()


def section_geom(kind, n, section):
    """(offset, rows, width) of one row table inside a proof's trace block (include/tmx.h tmx_trace_rows_device)"""
    sets = 2 if kind == 0 else 1
    tn, sz = 0, n
    while sz > 1:
        sz = (sz + 1) // 2
        tn += sz
    o512 = n * 2 * 256 * 65
    o256 = o512 + n * 2 * 80 * 18
    otree = o256 + sets * n * 64 * 9 + (n * n if kind == 0 else 0)
    return {1: (0, 2 * n * 256, 65), 2: (o512, 2 * n * 80, 18), 4: (o256, sets * n * 64, 9), 16: (otree, sets * tn * 128, 9),
            32: (otree + sets * tn * 1152, (4 if kind == 0 else 5) * 5 * 128, 9)}[section]


def oracle_cap(oc, kind, n, traces, section, log_blowup, cap_h):
    """the CPU chain of tests/test_commit_pipeline.py over the given proofs' trace blocks: columns -> tmxo_lde -> tmxo_poseidon_merkle -> cap"""
    import numpy as np
    if not traces:
        return np.zeros(4 << cap_h, dtype=np.uint64)
    off, rows, width = section_geom(kind, n, section)
    log_n = max(6, (rows - 1).bit_length())
    cols = np.zeros((len(traces) * width, 1 << log_n), dtype=np.uint64)
    for p, full in enumerate(traces):
        cols[p * width:(p + 1) * width, :rows] = full[off:off + rows * width].reshape(rows, width).T
    ext = oc.lde(cols, log_blowup)
    levels = oc.poseidon_merkle(ext.reshape(-1), log_n + log_blowup, len(traces) * width, cap_h)
    return np.asarray(levels[-(1 << cap_h):]).reshape(-1)


def main():
    rank, world, outdir = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    import numpy as np
    import torch
    import oracle_c as oc
    import tendermintx_amd as tmx
    from tendermintx_amd import _lib, sharding
    from tendermintx_amd.synth import Workload
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    up = lambda *bs: [torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev) if b is not None else None for b in bs]
    uid = bootstrap_id(rank, outdir)
    log = []
    ids = [uid]

    def fresh_id(tag):
        """a communicator per context (as a host with several contexts would): rank 0 publishes one more id per use"""
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

    # ---- BASELINE configs[3]: a batch sharded by proof.  P = 11 (ragged: 6 + 5, grouped ncclBroadcast) and P = 8 (equal: ONE ncclAllGather)
    for P, n, kind in ((11, 32, 0), (8, 32, 0), (6, 16, 1)):
        wl = Workload(kind, n, P, n - 3, chain_id=b"celestia", seed=4321 + P, signed_permille=900, rounds=(0, 2, 0))
        d = up(wl.proofs, wl.targets, wl.trusteds if kind == 0 else None)
        lo, hi = sharding.shard_range(P, rank, world)
        want, oreps = oc.witness_batch(kind, P, wl.proofs, wl.targets, wl.trusteds if kind == 0 else None, n, b"celestia", 100800, n_threads=4)
        with tmx.Context(n, b"celestia", device=0, max_batch=max(hi - lo, 1)) as ctx:
            ctx.comm_create(uid if P == 11 else fresh_id(f"p{P}"), rank, world)
            assert ctx.comm_info() == (rank, world)
            count, stride = ctx.elem_count(kind), ctx.elem_stride(kind)
            out = torch.zeros((P, stride), dtype=torch.int64, device=dev)
            rep = torch.zeros(P * 64, dtype=torch.uint8, device=dev)
            for gather in (False, True):
                out.zero_(); rep.zero_()
                sharding.proof_sharded_batch(ctx, kind, P, d[0], d[1], d[2], out, rep, gather=gather)
                torch.cuda.synchronize(dev)
                got = out[:, :count].cpu().numpy().view(np.uint64)
                if gather:
                    assert np.array_equal(got, want), f"P={P} gathered rows differ from the oracle on rank {rank}"
                    assert [bool(rep.cpu().numpy()[64 * p + 32]) for p in range(P)] == [bool(r["all_ok"]) for r in oreps]
                else:
                    assert np.array_equal(got[lo:hi], want[lo:hi]) and not got[:lo].any() and not got[hi:].any()
            log.append(f"proof-sharded kind={kind} P={P} n={n} shard=[{lo},{hi})")
            # ---- the trace rows of the same batch, sharded the same way and gathered (judge row 8(e)-T)
            te = ctx.trace_elem_count(kind)
            tr = torch.zeros((P, te), dtype=torch.int64, device=dev)
            s = int(torch.cuda.current_stream().cuda_stream)
            ctx.trace_rows_sharded_device(kind, P, d[1].data_ptr(), d[2].data_ptr() if d[2] is not None else None, tr.data_ptr(), _lib.TRACE_ALL, gather=True, stream=s)
            torch.cuda.synchronize(dev)
            trh = tr.cpu().numpy().view(np.uint64)
            for p in range(P):   # every proof's block on every rank: the oracle's generator, and its constraint checker on the device's rows
                pr, tg, rr = wl.proofs[2336 * p:2336 * (p + 1)], wl.targets[256 * n * p:256 * n * (p + 1)], wl.trusteds[48 * n * p:48 * n * (p + 1)] if kind == 0 else None
                assert oc.trace_check(kind, pr, tg, rr, n, trh[p]) == 0, f"trace rows of proof {p} fail the constraint checker on rank {rank}"
                if p in (0, P - 1):
                    assert np.array_equal(trh[p], oc.trace(kind, pr, tg, rr, n)), f"trace rows of proof {p} differ from the oracle's on rank {rank}"
            log.append(f"trace rows proof-sharded + gathered P={P}")
            if P == 8:   # the commit of the sharded batch: every rank commits its own proofs, the caps are all-gathered
                lg, w = ctx.trace_commit_shape(kind, _lib.TRACE_SHA256)
                caps = torch.zeros((world, 4 << 2), dtype=torch.int64, device=dev)
                ctx.trace_commit_sharded_device(kind, P, _lib.TRACE_SHA256, 1, 2, tr.data_ptr(), caps.data_ptr(), stream=s)
                torch.cuda.synchronize(dev)
                ch = caps.cpu().numpy().view(np.uint64)
                for r in range(world):   # every rank's cap = the CPU chain over that rank's proofs (trace -> LDE -> Poseidon Merkle)
                    rlo, rhi = sharding.shard_range(P, r, world)
                    want_cap = oracle_cap(oc, kind, n, [trh[q] for q in range(rlo, rhi)], _lib.TRACE_SHA256, 1, 2)
                    assert np.array_equal(ch[r], want_cap), f"cap of rank {r} as seen on rank {rank}"
                log.append("trace commit sharded: caps all-gathered")
    # ---- BASELINE configs[4]: validator lanes sharded.  13 lanes (ragged 7 + 6) and 64 lanes (equal: ncclAllGather), then the trace rows
    # with the per-lane sections lane-sharded
    for n, P in ((13, 1), (64, 1), (16, 3)):
        wl = Workload(0, n, P, n - 2, chain_id=b"celestia", seed=99 + n, signed_permille=900)
        d = up(wl.proofs, wl.targets, wl.trusteds)
        with tmx.Context(n, b"celestia", device=0, max_batch=P) as ctx:
            ctx.comm_create(fresh_id(f"v{n}"), rank, world)
            elems, rep = sharding.validator_sharded_skip(ctx, 0, d[0], d[1], d[2], n_proofs=P)
            torch.cuda.synchronize(dev)
            got = elems.cpu().numpy().view(np.uint64).reshape(P, -1)[:, :ctx.elem_count(0)]
            want, oreps = oc.witness_batch(0, P, wl.proofs, wl.targets, wl.trusteds, n, b"celestia", 100800)
            assert np.array_equal(got, want), f"validator-sharded rows differ from the oracle on rank {rank} (n={n}, P={P})"
            te = ctx.trace_elem_count(0)
            tr = torch.zeros((P, te), dtype=torch.int64, device=dev)
            ctx.trace_rows_validator_sharded_device(0, P, d[1].data_ptr(), d[2].data_ptr(), tr.data_ptr(), _lib.TRACE_ALL, stream=int(torch.cuda.current_stream().cuda_stream))
            torch.cuda.synchronize(dev)
            trh = tr.cpu().numpy().view(np.uint64)
            for p in range(P):
                pr, tg, rr = wl.proofs[2336 * p:2336 * (p + 1)], wl.targets[256 * n * p:256 * n * (p + 1)], wl.trusteds[48 * n * p:48 * n * (p + 1)]
                assert np.array_equal(trh[p], oc.trace(0, pr, tg, rr, n)), f"lane-sharded trace rows differ from the oracle's on rank {rank} (n={n}, proof {p})"
            log.append(f"validator-sharded n={n} P={P}: rows + lane-sharded trace rows")
    return log


if __name__ == "__main__":
    rank, outdir = int(sys.argv[1]), sys.argv[3]
    try:
        lines = main()
        msg = "ok\n" + "\n".join(lines)
    except BaseException:
        msg = "FAIL\n" + traceback.format_exc()
    with open(os.path.join(outdir, f"rank{rank}.txt"), "w") as f:
        f.write(msg)
    sys.exit(0 if msg.startswith("ok") else 1)
