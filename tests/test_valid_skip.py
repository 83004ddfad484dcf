"""`is_valid_skip` / `find_block_to_request` (reference circuits/input/tendermint_utils.rs:444-482,
circuits/input/mod.rs:160-186) -- the caller of the skip path -- as one batched launch.  CPU: codec + oracle; GPU: parity."""
import os
import struct

import numpy as np
import pytest

import tmx_model as m
from conftest import GOLDEN

FX = os.path.join(GOLDEN, "fixtures", "mocha-4")
PAIRS = [(10000, 10500), (3000, 3100), (10500, 157001), (157001, 157001), (10000, 10001), (157001, 10500), (3000, 10500)]


def _full_shared(start, target, sigs):
    """shared power without the reference's early exit (what the batched kernel reduces)"""
    sh = 0
    for s in start:
        for t in target:
            if t[:20] == s[:20]:
                sh += sum(struct.unpack_from("<Q", t, 24)[0] for g in sigs if g[20] and g[:20] == t[:20])
                break
    return sh & m.U64


def test_codec_and_oracle_on_fixtures(built_lib, oracle):
    from tendermintx_amd.circuits import InputDataFetcher
    f, pf = InputDataFetcher(FX), m.FixtureFetcher(FX)
    expect = {(10000, 10500): True, (3000, 3100): True, (10500, 157001): False, (157001, 157001): True}
    for a, b in PAIRS:
        s, t, g = m.skipcheck_records(pf.validators(a), pf.validators(b), pf.signed_header(b)["commit"])
        cs, ns, ct, nt, cg, ng = f.get_skipcheck_inputs(128, a, b)
        assert cs[:32 * ns] == b"".join(s) and ct[:32 * nt] == b"".join(t) and cg[:32 * ng] == b"".join(g)
        assert cs[32 * ns:] == bytes(32 * (128 - ns))
        r_py, r_c = m.is_valid_skip(s, t, g), oracle.is_valid_skip(b"".join(s), b"".join(t), b"".join(g))
        assert r_py == r_c
        if (a, b) in expect:
            assert r_c[0] == expect[(a, b)]
        # the early exit never changes the verdict
        assert r_c[0] == (float(r_c[2]) * (1.0 / 3.0) <= float(_full_shared(s, t, g)))


def _random_case(rng, n_max, pool=None):
    """start / target / sigs with overlaps, strangers, duplicate signature addresses, nil and absent votes, and total powers chosen
    so that shared sits exactly on, just below or just above total/3 (the f64 product is what decides)."""
    ns, nt = int(rng.integers(1, n_max + 1)), int(rng.integers(1, n_max + 1))
    pool = pool or [rng.bytes(20) for _ in range(n_max * 2)]
    t_addr = [pool[i] for i in rng.permutation(len(pool))[:nt]]
    mode = int(rng.integers(0, 4))
    if mode == 0:
        powers = [int(rng.integers(1, 30_000_000)) for _ in range(nt)]
    elif mode == 1:
        powers = [3] * nt                                   # total = 3 nt: exact thirds
    elif mode == 2:
        powers = [int(rng.integers(2**50, 2**53)) for _ in range(nt)]   # beyond f64 integer precision once summed
    else:
        powers = [1] * nt
    target = [m.pack_addr(a, True, p) for a, p in zip(t_addr, powers)]
    s_addr = [t_addr[int(rng.integers(0, nt))] if rng.random() < 0.6 else pool[int(rng.integers(0, len(pool)))] for _ in range(ns)]
    start = [m.pack_addr(a, True, int(rng.integers(1, 1000))) for a in s_addr]
    sigs = []
    for a in t_addr:
        r = rng.random()
        if r < 0.6:
            sigs.append(m.pack_addr(a, True, 0))            # commit or nil vote: both carry the address
        elif r < 0.8:
            sigs.append(m.pack_addr(bytes(20), False, 0))   # absent
        else:
            sigs.append(m.pack_addr(t_addr[0], True, 0))    # duplicate address
    return start, target, sigs


def test_oracle_c_equals_model_random(oracle):
    rng = np.random.default_rng(12)
    for _ in range(300):
        s, t, g = _random_case(rng, 12)
        assert m.is_valid_skip(s, t, g) == oracle.is_valid_skip(b"".join(s), b"".join(t), b"".join(g))


@pytest.mark.gpu
def test_valid_skip_batch_gpu(built_lib, oracle):
    import tendermintx_amd as tmx
    rng = np.random.default_rng(99)
    n_max, n_cand = 32, 200
    with tmx.Context(n_max, b"celestia") as ctx:
        pool = [rng.bytes(20) for _ in range(n_max + 8)]   # one address universe: start and every candidate draw from it
        start = [m.pack_addr(a, True, 1) for a in pool[:20]]
        cases = []
        for _ in range(n_cand):
            _, t, g = _random_case(rng, n_max, pool)
            cases.append((t, g))
        pad = lambda recs: b"".join(recs).ljust(n_max * 32, b"\0")
        valid, shared, total = ctx.valid_skip_batch(pad(start), len(start), b"".join(pad(t) for t, _ in cases), [len(t) for t, _ in cases],
                                                    b"".join(pad(g) for _, g in cases), [len(g) for _, g in cases])
        n_true = 0
        for (t, g), v, sh, to in zip(cases, valid, shared, total):
            ov, _, ot = oracle.is_valid_skip(b"".join(start), b"".join(t), b"".join(g))
            assert v == ov and to == ot and sh == _full_shared(start, t, g)
            n_true += v
        assert 0 < n_true < n_cand


@pytest.mark.gpu
def test_find_block_to_request_gpu(built_lib, oracle):
    """find_block_to_request over the fixture heights: candidates are fetched up front and judged in one launch."""
    import tendermintx_amd as tmx
    from tendermintx_amd.circuits import InputDataFetcher
    pf = m.FixtureFetcher(FX)

    class Fetcher(InputDataFetcher):  # the descent 10500 -> 10250 -> ... needs blocks that are not fixtures: serve the nearest fixture
        def _read(self, height, name):
            have = [3000, 3001, 3100, 10000, 10001, 10500, 10501, 157001]
            near = min(have, key=lambda h: abs(h - height))
            return super()._read(near, name)

    with tmx.Context(128, b"mocha-4") as ctx:
        f = Fetcher(FX)
        assert f.find_block_to_request(ctx, 10000, 10500) == 10500       # valid at the first candidate
        assert f.find_block_to_request(ctx, 10000, 10001) == 10001       # adjacent: the loop returns at once
        # one candidate judged through the plain entry point, against the oracle
        s, ns, t, nt, g, ng = f.get_skipcheck_inputs(128, 10500, 157001)
        valid, shared, total = ctx.valid_skip_batch(s, ns, t, [nt], g, [ng])
        ov, osh, ot = oracle.is_valid_skip(s[:32 * ns], t[:32 * nt], g[:32 * ng])
        assert valid == [ov] == [False] and total == [ot] and shared == [osh]
