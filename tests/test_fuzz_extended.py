"""Extended differential fuzz of the HIP path against the oracle (opt-in depth: TMX_FUZZ_SEEDS=N, default 64: a second and a half on the GPU box).
Heavier than test_gpu_parity.py::test_random_shapes_and_bit_flips: whole bytes replaced by random and extreme values, field-aware
extremes (lengths, powers, flags), duplicated keys across the two validator sets, several mutations per proof.  The verdicts are free to
be anything; elements and reports must equal the oracle's bit for bit."""
import os
import struct

import numpy as np
import pytest

from test_gpu_parity import _check_vs_oracle

pytestmark = pytest.mark.gpu

SEEDS = int(os.environ.get("TMX_FUZZ_SEEDS", "64"))
NSET = tuple(int(v) for v in os.environ.get("TMX_FUZZ_NSET", "1,2,4,7,16,31,32,33,64,100,128").split(","))   # e.g. 200,256,300,512: the wide k_proof
CHAIN_IDS = (b"celestia", b"mocha-4", b"a", b"thirteen-char")          # what the synthetic sign-bytes can carry
CTX_CHAIN_IDS = CHAIN_IDS + (b"x" * 50, b"celestia-but-longer")              # what a context can be configured for
EXTREME_BYTES = (0x00, 0x01, 0x7f, 0x80, 0xfe, 0xff)
EXTREME_U64 = (0, 1, 2**62, 2**63 - 1, 2**63, 2**63 + 5, 2**64 - 1)


def _pick(rng, seq):
    return seq[int(rng.integers(0, len(seq)))]


@pytest.fixture(scope="module")
def tmx(built_lib):
    import tendermintx_amd
    return tendermintx_amd


def _mutated_batch(seed):
    from tendermintx_amd.synth import Workload
    rng = np.random.default_rng(0x7E57 + 7919 * seed)
    kind = int(rng.integers(0, 2))
    n = _pick(rng, NSET)
    P = int(rng.integers(1, 24))
    nb = int(rng.integers(1, n + 1))
    wl_chain = _pick(rng, CHAIN_IDS)
    ctx_chain = wl_chain if rng.random() < 0.8 else _pick(rng, CTX_CHAIN_IDS)   # (a context configured for another chain: checks fail, parity holds)
    skip_max = _pick(rng, (1, 2, 1000, 100800, 2**40))
    wl = Workload(kind, n, P, nb, chain_id=wl_chain, seed=int(rng.integers(1, 2**31)), signed_permille=int(rng.integers(300, 1001)),
                  rounds=(0, int(rng.integers(0, 7)), 0))
    proofs, targets = bytearray(wl.proofs), bytearray(wl.targets)
    trusteds = bytearray(wl.trusteds) if kind == 0 else None
    for p in range(P):
        if rng.random() < 0.25:
            continue  # some proofs stay pristine
        for _ in range(int(rng.integers(1, 7))):
            mode = int(rng.integers(0, 10))
            t0 = p * n * 256
            lane = t0 + int(rng.integers(0, n)) * 256
            if mode == 0:      # any byte of the proof record := random (the set sizes stay: nb > N is a host-side error, covered elsewhere)
                off = int(rng.integers(0, 2336))
                if off not in range(56, 64):
                    proofs[p * 2336 + off] = int(rng.integers(0, 256))
            elif mode == 1:    # ... := an extreme byte
                off = int(rng.integers(0, 2336))
                if off not in range(56, 64):
                    proofs[p * 2336 + off] = _pick(rng, EXTREME_BYTES)
            elif mode == 2:    # any byte of a target lane := random / extreme
                targets[lane + int(rng.integers(0, 256))] = int(rng.integers(0, 256)) if rng.random() < 0.5 else _pick(rng, EXTREME_BYTES)
            elif mode == 3:    # voting power extremes
                targets[lane + 224:lane + 232] = struct.pack("<Q", _pick(rng, EXTREME_U64))
            elif mode == 4:    # lengths and flags
                which = int(rng.integers(0, 3))
                if which == 0:
                    targets[lane + 222] = _pick(rng, (0, 1, 33, 34, 45, 46, 47, 80, 255))
                elif which == 1:
                    targets[lane + 220:lane + 222] = struct.pack("<H", _pick(rng, (0, 1, 63, 64, 111, 112, 123, 124, 125, 300, 65535)))
                else:
                    targets[lane + 223] = int(rng.integers(0, 256))
            elif mode == 5 and trusteds is not None:   # trusted lane: any byte, power extremes, a key copied from the target set
                j = p * n * 48 + int(rng.integers(0, n)) * 48
                which = int(rng.integers(0, 3))
                if which == 0:
                    trusteds[j + int(rng.integers(0, 48))] = int(rng.integers(0, 256))
                elif which == 1:
                    trusteds[j + 32:j + 40] = struct.pack("<Q", _pick(rng, EXTREME_U64))
                else:
                    trusteds[j:j + 32] = targets[lane:lane + 32]
            elif mode == 6:    # a key duplicated inside the target set (and its signature, or not)
                other = t0 + int(rng.integers(0, n)) * 256
                targets[other:other + 32] = targets[lane:lane + 32]
                if rng.random() < 0.5:
                    targets[other + 32:other + 96] = targets[lane + 32:lane + 96]
            elif mode == 7:    # signature bytes: the s half at and above the group order, R := small-order / non-canonical encodings
                which = int(rng.integers(0, 4))
                if which == 0:
                    targets[lane + 64:lane + 96] = bytes([0xff] * 32)
                elif which == 1:
                    targets[lane + 64:lane + 96] = (2**252 + 27742317777372353535851937790883648493).to_bytes(32, "little")  # s = l
                elif which == 2:
                    targets[lane + 32:lane + 64] = bytes([1] + [0] * 31)   # R = the identity
                else:
                    targets[lane + 32:lane + 64] = bytes([0xed] + [0xff] * 30 + [0x7f])  # y = p (non-canonical zero)
            elif mode == 8:    # public key: small order / non-canonical / not on the curve
                keys = (bytes([1] + [0] * 31), bytes([0] * 32), bytes([0xec] + [0xff] * 30 + [0x7f]), bytes([0xee] + [0xff] * 30 + [0x7f]),
                        bytes([2] + [0] * 31), bytes([0xff] * 32))
                targets[lane:lane + 32] = keys[int(rng.integers(0, len(keys)))]
            else:              # a run of bytes of the sign-bytes message zeroed or randomized
                a = int(rng.integers(96, 220))
                b = min(220, a + int(rng.integers(1, 24)))
                targets[lane + a:lane + b] = bytes(b - a) if rng.random() < 0.5 else rng.integers(0, 256, b - a, dtype=np.uint8).tobytes()
    assert len(targets) == len(wl.targets) and len(proofs) == len(wl.proofs)
    return kind, n, bytes(proofs), bytes(targets), bytes(trusteds) if trusteds is not None else None, ctx_chain, skip_max


SEED0 = int(os.environ.get("TMX_FUZZ_SEED0", "0"))


@pytest.mark.parametrize("seed", range(SEED0, SEED0 + SEEDS))
def test_fuzz_bytes_and_field_extremes(tmx, oracle, seed):
    kind, n, proofs, targets, trusteds, chain_id, skip_max = _mutated_batch(seed)
    _check_vs_oracle(tmx, oracle, kind, n, proofs, targets, trusteds, chain_id, skip_max, repeat=2)   # cold, then from the key cache


TRACE_SEEDS = int(os.environ.get("TMX_FUZZ_TRACE_SEEDS", "4"))


@pytest.mark.parametrize("seed", range(TRACE_SEEDS))
def test_fuzz_trace_rows(tmx, oracle, seed):
    """The Level-2 rows of mutated batches (undecodable and small-order keys, non-canonical scalars, duplicated keys ...): bit-exact vs the
    oracle's generator and accepted by its constraint checker."""
    from test_trace import _gpu_trace
    for s in range(1000 * seed, 1000 * seed + 40):   # the first batch of this seed's range that is small enough for the generator
        kind, n, proofs, targets, trusteds, _, _ = _mutated_batch(s)
        P = len(proofs) // 2336
        if n <= 16 and P <= 8:
            break
    else:
        pytest.skip("no small batch in this seed range")
    got = _gpu_trace(tmx, kind, n, proofs, targets, trusteds)
    for p in range(P):
        t = targets[p * n * 256:(p + 1) * n * 256]
        r = trusteds[p * n * 48:(p + 1) * n * 48] if kind == 0 else None
        pr = proofs[p * 2336:(p + 1) * 2336]
        assert np.array_equal(got[p], oracle.trace(kind, pr, t, r, n)), (s, p)
        assert oracle.trace_check(kind, pr, t, r, n, got[p]) == 0, (s, p)


def test_one_context_many_different_calls(tmx, oracle):
    """State carried between calls of ONE context (hash-table parity, cleared-table event, tiny-launch path vs per-key tables, scratch buffers,
    the events of the stream schedule): 80 calls of random kind, batch size (1 .. 96 proofs: both sides of every size threshold), mutation
    and entry point -- host rows, hint-only u32 rows, device rows on two alternating streams -- each checked against the oracle."""
    import torch
    from tendermintx_amd import _lib
    n = 64
    rng = np.random.default_rng(77)
    dev = torch.device("cuda", 0)
    streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
    with tmx.Context(n, b"celestia", 100800, max_batch=96) as ctx:
        sizes = set()
        for it in range(80):
            s0 = int(rng.integers(0, 20000))
            saved = NSET
            try:   # up to four mutated batches of one kind glued together: 1 .. 92 proofs = 64 .. 5888 lanes
                globals()["NSET"] = (n,)
                kind, _, proofs, targets, trusteds, _, _ = _mutated_batch(s0)
                for extra in range(int(rng.integers(0, 4))):
                    for s1 in range(s0 + 1 + 50 * extra, s0 + 50 * (extra + 1)):
                        k2, _, p2, t2, r2, _, _ = _mutated_batch(s1)
                        if k2 == kind:
                            proofs, targets = proofs + p2, targets + t2
                            trusteds = trusteds + r2 if trusteds is not None else None
                            break
            finally:
                globals()["NSET"] = saved
            P = len(proofs) // 2336
            sizes.add(P)
            want, oreps = oracle.witness_batch(kind, P, proofs, targets, trusteds, n, b"celestia", 100800, n_threads=8)
            mode = int(rng.integers(0, 3))
            if mode == 0:
                elems, reps = ctx.witness_batch(kind, proofs, targets, trusteds)
                assert np.array_equal(elems, want) and reps == oreps, (it, s0)
            elif mode == 1:
                rows, reps = ctx.witness_batch_hint(kind, proofs, targets, trusteds)
                h = ctx.hint_elem_count(kind)
                assert np.array_equal(rows.astype(np.uint64), want[:, :h]) and reps == oreps, (it, s0)
            else:
                st = streams[it & 1]
                d = [torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev) if b else None for b in (proofs, targets, trusteds)]
                out = torch.zeros((P, ctx.elem_stride(kind)), dtype=torch.int64, device=dev)
                rep = torch.zeros(P * 64, dtype=torch.uint8, device=dev)
                torch.cuda.synchronize(dev)   # (calls of one context on different streams must be ordered by the caller: include/tmx.h)
                ctx.witness_batch_device(kind, P, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr() if d[2] is not None else None,
                                         out.data_ptr(), rep.data_ptr(), st.cuda_stream)
                st.synchronize()
                got = out.cpu().numpy().view(np.uint64)[:, :want.shape[1]]
                assert np.array_equal(got, want), (it, s0)
        assert min(sizes) <= 8 and max(sizes) >= 64   # (both the tiny-launch path and the split tail were on)


@pytest.mark.parametrize("n_proofs", [int(os.environ.get("TMX_FUZZ_BIG", "300"))])
def test_big_mutated_batch(tmx, oracle, n_proofs):
    """Mutated proofs at N = 128 glued into one large batch: the size-dependent schedules (walk in parts, tail aside, leaves first, the
    split tail and the larger serializer cap from 131 072 lanes: TMX_FUZZ_BIG=1100) see hostile inputs too."""
    n = 128
    saved = NSET
    try:
        globals()["NSET"] = (n,)
        kind0, parts = 0, []
        s = 0
        while sum(len(p[0]) // 2336 for p in parts) < n_proofs:
            kind, _, proofs, targets, trusteds, _, _ = _mutated_batch(50000 + s)
            s += 1
            if kind == kind0:
                parts.append((proofs, targets, trusteds))
    finally:
        globals()["NSET"] = saved
    proofs = b"".join(p[0] for p in parts)[:n_proofs * 2336]
    targets = b"".join(p[1] for p in parts)[:n_proofs * n * 256]
    trusteds = b"".join(p[2] for p in parts)[:n_proofs * n * 48]
    # (cold, then warm on the same context: the split schedule, the proof-major capped row writer and -- from 512 proofs on -- the throughput
    # regime's settings see the hostile inputs as well)
    _check_vs_oracle(tmx, oracle, kind0, n, proofs, targets, trusteds, b"celestia", threads=16, repeat=int(os.environ.get("TMX_FUZZ_BIG_REPEAT", "2")))
