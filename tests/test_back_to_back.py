"""Back-to-back calls on one context without a synchronize in between -- what a prover that keeps the device busy does -- give the bits of the
same batch computed alone: the schedule's cross-stream joins (four streams, the late cache epilogue, the new-key pipeline beside the
resident lanes, the dedup beside the hash role of small batches) order every kernel against the NEXT call's kernels too.  Batches with and
without new keys alternate, the key cache is flushed now and then; rows and reports are compared bit for bit.  (The batches themselves are
checked against the oracle in test_gpu_parity.py / test_key_cache.py; this test is about ordering.)"""
import os

import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("P, n, rounds", [(24, 128, 60), (40, 64, 60), (130, 128, 30), (256, 128, 20)])
def test_back_to_back_calls_give_the_bits_of_a_call_alone(built_lib, P, n, rounds):
    import torch
    from tendermintx_amd import Context, KIND_SKIP
    from tendermintx_amd.synth import Workload, bench_workload
    dev = torch.device("cuda:0")
    up = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
    base = bench_workload("survey8d", n, P, seed=21 + P)
    variants = [tuple(up(b) for b in (base.proofs, base.targets, base.trusteds))]
    for k, seed in ((1, 5), (3, 6)):   # k proofs over validator sets the cache has not seen, in front of the batch
        f = Workload(0, n, k, min(n, 90), chain_id=b"celestia", seed=7000 + 13 * seed + P, signed_permille=950, n_sets=k)
        variants.append(tuple(up(a + b[len(a):]) for a, b in ((f.proofs, base.proofs), (f.targets, base.targets), (f.trusteds, base.trusteds))))
    st = torch.cuda.Stream(dev)
    with Context(n, b"celestia", 100800, device=0, max_batch=P) as ctx:
        stride = ctx.elem_stride(KIND_SKIP)

        def buffers():
            o, r = torch.empty(P * stride, dtype=torch.int64, device=dev), torch.zeros(P * 64, dtype=torch.uint8, device=dev)
            torch.cuda.synchronize(dev)   # (the fill runs on torch's stream, the calls on `st`)
            return o, r

        def call(v, o, r):
            ctx.witness_batch_device(KIND_SKIP, P, v[0].data_ptr(), v[1].data_ptr(), v[2].data_ptr(), o.data_ptr(), r.data_ptr(), st.cuda_stream)

        want = []
        for v in variants:
            o, r = buffers()
            call(v, o, r)
            torch.cuda.synchronize(dev)
            want.append((o, r))
        outs = [buffers() for _ in range(4)]
        for it in range(rounds):
            order = [(it + j) % 3 if it % 3 == 0 else 0 for j in range(4)]
            if it % 7 == 3:
                ctx.key_cache_flush()
            for j, vi in enumerate(order):
                call(variants[vi], *outs[j])
            torch.cuda.synchronize(dev)
            for j, vi in enumerate(order):
                assert torch.equal(outs[j][1], want[vi][1]), f"reports of call {j} of round {it} (variant {vi}, order {order})"
                assert torch.equal(outs[j][0], want[vi][0]), f"rows of call {j} of round {it} (variant {vi}, order {order})"


@pytest.mark.parametrize("n", [128, 64])
def test_mixed_sizes_back_to_back(built_lib, n):
    """The transitions that CHANGE which side stream owns the key pipeline's tail, without a synchronize between the calls: a tiny
    call (<= 512 lanes: tail on side2) -> a split hash-first batch (dedup on side3) -> a large split batch (dedup on s, tail on side3) -> a tiny
    call again -> a non-batch eddsa_lanes call, with and without new keys and with cache flushes (cold schedule: tail on side2).  One context,
    one stream, max_batch = the largest; every call's rows and reports equal the same call made alone behind a synchronize."""
    import numpy as np
    import torch
    from tendermintx_amd import Context, KIND_SKIP
    from tendermintx_amd.synth import Workload, bench_workload
    dev = torch.device("cuda:0")
    up = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
    sizes = [1, 2, 24, 256, 3, 40, 130] + ([800] if n == 128 else [])   # (800 x 128 lanes: the throughput regime's schedule, api.cpp THROUGHPUT_LANES = 81 920)
    rounds = int(os.environ.get("TMX_SOAK_ROUNDS", "40"))                # (a soak: TMX_SOAK_ROUNDS=3000)
    Pmax = max(sizes)
    base = bench_workload("survey8d", n, Pmax, seed=77 + n)
    fresh = [Workload(0, n, 3, min(n, 90), chain_id=b"celestia", seed=8100 + 17 * k + n, signed_permille=950, n_sets=3) for k in range(2)]

    def variant(P, f):   # the first P proofs of the base batch; f: proofs over validator sets the cache has not seen in front (new keys)
        pr, tg, tr = base.proofs[:2336 * P], base.targets[:256 * n * P], base.trusteds[:48 * n * P]
        if f is not None:
            k = min(3, P)
            pr, tg, tr = f.proofs[:2336 * k] + pr[2336 * k:], f.targets[:256 * n * k] + tg[256 * n * k:], f.trusteds[:48 * n * k] + tr[48 * n * k:]
        return tuple(up(b) for b in (pr, tg, tr))

    calls = [(P, variant(P, f)) for P in sizes for f in (None, fresh[0], fresh[1])]
    st = torch.cuda.Stream(dev)
    with Context(n, b"celestia", 100800, device=0, max_batch=Pmax) as ctx:
        stride = ctx.elem_stride(KIND_SKIP)

        def buffers(P):
            o, r = torch.empty(P * stride, dtype=torch.int64, device=dev), torch.zeros(P * 64, dtype=torch.uint8, device=dev)
            return o, r

        def call(P, v, o, r):
            ctx.witness_batch_device(KIND_SKIP, P, v[0].data_ptr(), v[1].data_ptr(), v[2].data_ptr(), o.data_ptr(), r.data_ptr(), st.cuda_stream)

        want = []
        for P, v in calls:
            o, r = buffers(P)
            torch.cuda.synchronize(dev)
            call(P, v, o, r)
            torch.cuda.synchronize(dev)
            want.append((o, r))
        rng = np.random.default_rng(5 + n)
        for it in range(rounds):
            seq = [int(x) for x in rng.integers(0, len(calls), 6)]
            if it % 8 == 4 and 800 in sizes:   # across the regime boundary and back: large split -> throughput regime -> tiny -> throughput regime -> hash-first split
                seq = [sizes.index(256) * 3, sizes.index(800) * 3 + (it // 8) % 3, sizes.index(1) * 3, sizes.index(800) * 3, sizes.index(40) * 3 + 1, sizes.index(800) * 3 + 2]
            if it % 4 == 0:   # the transitions named above, in order: tiny -> hash-first split -> large split -> tiny -> hash-first split
                seq = [sizes.index(1) * 3, sizes.index(24) * 3 + (it // 4) % 3, sizes.index(256) * 3, sizes.index(2) * 3 + 1, sizes.index(40) * 3, sizes.index(130) * 3 + 2]
            outs = [buffers(calls[i][0]) for i in seq]
            torch.cuda.synchronize(dev)
            if it % 5 == 2:
                ctx.key_cache_flush()
            for i, (o, r) in zip(seq, outs):
                call(calls[i][0], calls[i][1], o, r)
            torch.cuda.synchronize(dev)
            for j, (i, (o, r)) in enumerate(zip(seq, outs)):
                assert torch.equal(r, want[i][1]), f"reports of call {j} of round {it} (sizes {[calls[k][0] for k in seq]})"
                assert torch.equal(o, want[i][0]), f"rows of call {j} of round {it} (sizes {[calls[k][0] for k in seq]})"


def test_two_contexts_on_one_device(built_lib):
    """Two or three contexts of one device, each on a caller stream of its own, batches alternating without a synchronize (a host with a context
    per worker thread): they share the device's three internal streams and nothing else -- rows and reports of every call equal the same batch
    computed alone, and a context destroyed in between leaves the others intact.  (Correctness only: five streams on the runtime's four hardware
    queues cost time -- 0.49 - 0.64 instead of 0.38 ms per 256-proof step, with or without making the contexts take turns by event:
    profiles/r06_two_ctx_and_proof_chunk_ab.txt -- so INTEGRATION.md recommends one context per process and GPU.)"""
    import torch
    from tendermintx_amd import Context, KIND_SKIP
    from tendermintx_amd.synth import bench_workload
    dev = torch.device("cuda:0")
    up = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
    n = 128
    shapes = [(130, 31), (24, 32), (256, 33)]     # (proofs, seed) per context: a split batch, a hash-first batch, the bench shape
    ctxs, want = [], []
    for P, seed in shapes:
        w = bench_workload("survey8d", n, P, seed=seed)
        c = Context(n, b"celestia", 100800, device=0, max_batch=P)
        st = torch.cuda.Stream(dev)
        d = tuple(up(b) for b in (w.proofs, w.targets, w.trusteds))
        stride = c.elem_stride(KIND_SKIP)
        ctxs.append((c, st, d, P, stride))
    try:
        def call(k, o, r):
            c, st, d, P, _ = ctxs[k]
            c.witness_batch_device(KIND_SKIP, P, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), o.data_ptr(), r.data_ptr(), st.cuda_stream)

        def buffers(k):
            _, _, _, P, stride = ctxs[k]
            return torch.empty(P * stride, dtype=torch.int64, device=dev), torch.zeros(P * 64, dtype=torch.uint8, device=dev)

        for k in range(len(ctxs)):     # every batch alone (twice: cold, warm), behind a synchronize
            for _ in range(2):
                o, r = buffers(k)
                torch.cuda.synchronize(dev)
                call(k, o, r)
                torch.cuda.synchronize(dev)
            want.append((o, r))
        for it in range(12):
            order = [(it + j) % len(ctxs) for j in range(6)]
            outs = [buffers(k) for k in order]
            torch.cuda.synchronize(dev)
            for k, (o, r) in zip(order, outs):
                call(k, o, r)
            torch.cuda.synchronize(dev)
            for j, (k, (o, r)) in enumerate(zip(order, outs)):
                assert torch.equal(r, want[k][1]) and torch.equal(o, want[k][0]), f"call {j} of round {it} (context {k}, order {order})"
            if it == 7:                # the context that enqueued last goes away
                ctxs[order[-1]][0].close()
                gone = order[-1]
                ctxs = [x for i, x in enumerate(ctxs) if i != gone]
                want = [x for i, x in enumerate(want) if i != gone]
    finally:
        for c, *_ in ctxs:
            c.close()
