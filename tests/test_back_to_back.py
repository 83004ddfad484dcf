"""Back-to-back calls on one context without a synchronize in between -- what a prover that keeps the device busy does -- give the bits of the
same batch computed alone: the schedule's cross-stream joins (four streams, the late cache epilogue, the new-key pipeline beside the
resident lanes, the dedup beside the hash role of small batches) order every kernel against the NEXT call's kernels too.  Batches with and
without new keys alternate, the key cache is flushed now and then; rows and reports are compared bit for bit.  (The batches themselves are
checked against the oracle in test_gpu_parity.py / test_key_cache.py; this test is about ordering.)"""
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
