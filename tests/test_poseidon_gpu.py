"""Poseidon over Goldilocks + Merkle caps on the GPU (tendermintx_amd/csrc/poseidon.hip through the C ABI) against the CPU oracle
(oracle/c/tmxo_poseidon.c, itself pinned against the independent Python model and the algebraic self-checks in
tests/test_poseidon_oracle.py): bit-exact, including non-canonical inputs, injected constants (the general MDS path) and, at full LDE
sizes, through size-independent properties.  plonky2's tables are absent from the reference tree (Cargo.lock:2957-2982): parity unpinned."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
P = 2**64 - 2**32 + 1


@pytest.fixture(scope="module")
def tmx(built_lib):
    import tendermintx_amd
    return tendermintx_amd


@pytest.fixture(scope="module")
def ctx(tmx):
    c = tmx.Context(4, b"celestia")
    yield c
    c.close()


def _states(seed, n):
    rng = np.random.default_rng(seed)
    s = rng.integers(0, 2**64, (n, 12), dtype=np.uint64)
    s[0] = 0
    s[1] = P - 1
    s[2] = 2**64 - 1            # above p: taken mod p
    s[3] = np.arange(12, dtype=np.uint64)
    s[4] = P                    # = 0 mod p
    return s


def test_permutation_vs_oracle(ctx, oracle):
    s = _states(1, 1000)
    got = ctx.poseidon_permute(s)
    want = oracle.poseidon_permute(s)
    assert np.array_equal(got, want) and int(got.max()) < P


def test_default_constants_are_the_grain_stream(ctx, oracle):
    """the library's own Grain LFSR (api.cpp), the oracle's (C) and the model's (Python) give one stream: a permutation of the zero state
    depends on every constant"""
    import poseidon_model as pm
    z = np.zeros((1, 12), dtype=np.uint64)
    assert [int(x) for x in ctx.poseidon_permute(z)[0]] == pm.Poseidon().permute([0] * 12)


def test_default_constants_are_flagged(tmx, built_lib):
    """the context says whether its round constants were injected (the Grain defaults are not the reference prover's: ADVICE r3)"""
    with tmx.Context(4, b"x") as c:
        assert built_lib.tmx_poseidon_constants_injected(c._h) == 0
        c.poseidon_set_constants(mds_circ=[17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20])
        assert built_lib.tmx_poseidon_constants_injected(c._h) == 0          # an MDS alone does not pin the constants
        c.poseidon_set_constants(round_constants=list(range(1, 361)))
        assert built_lib.tmx_poseidon_constants_injected(c._h) == 1


def test_injected_constants_and_general_mds(tmx, oracle):
    """another table (as a maintainer would inject plonky2's): large MDS entries take the general product path"""
    import poseidon_model as pm
    rng = np.random.default_rng(9)
    rc = [int(x) % P for x in rng.integers(0, 2**63, 360, dtype=np.uint64)]
    s = _states(2, 300)
    with tmx.Context(4, b"celestia") as c:
        for circ, diag in (([int(x) for x in rng.integers(1, 50, 12)], [5] + [0] * 11),
                           ([int(x) % P for x in rng.integers(0, 2**63, 12, dtype=np.uint64)], [int(x) % P for x in rng.integers(0, 2**63, 12, dtype=np.uint64)])):
            c.poseidon_set_constants(rc, circ, diag)
            got = c.poseidon_permute(s)
            model = pm.Poseidon(rc, circ, diag)
            for i in (0, 1, 2, 7, 299):
                assert [int(x) for x in got[i]] == model.permute([int(x) for x in s[i]]), (i, circ[0])
            try:
                oracle.poseidon_set_constants(rc, circ, diag)
                assert np.array_equal(got, oracle.poseidon_permute(s))
            finally:
                oracle.poseidon_set_constants(pm.grain_constants(), pm.MDS_CIRC, pm.MDS_DIAG)


def test_small_mds_path_at_its_bounds(tmx, oracle):
    """the 32-bit-limb MDS layer (every entry < 2^16) with every entry at 2^16 - 1, the diagonal too (it joins the circulant's entry 0:
    2^17 - 2), every round constant p - 1 (they enter as multiply-adds of the row accumulators) and all-ones states: the largest
    accumulators the layer can see (< 2^54, poseidon.hip: pos_fold) -- against the Python model and the C oracle"""
    import poseidon_model as pm
    rc, circ, diag = [P - 1] * 360, [65535] * 12, [65535] * 12
    s = _states(5, 200)
    s[5] = 2**64 - 1
    s[6] = 2**32 - 1
    s[7] = (2**64 - 1) ^ (2**32 - 1)
    with tmx.Context(4, b"celestia") as c:
        c.poseidon_set_constants(rc, circ, diag)
        got = c.poseidon_permute(s)
        model = pm.Poseidon(rc, circ, diag)
        for i in (0, 1, 2, 5, 6, 7, 199):
            assert [int(x) for x in got[i]] == model.permute([int(x) % P for x in s[i]]), i
        try:
            oracle.poseidon_set_constants(rc, circ, diag)
            assert np.array_equal(got, oracle.poseidon_permute(s))
        finally:
            oracle.poseidon_set_constants(pm.grain_constants(), pm.MDS_CIRC, pm.MDS_DIAG)


def test_field_product_rare_borrow_path(tmx, oracle):
    """gl_mul_lazy corrects `lo - hi_hi` behind a wave-wide branch (lo < hi_hi happens once in 2^32 random products, so random states
    never take it).  With zero round constants the first S-box squares the state elements themselves: 2^48, 3 * 2^48, 2^63 and friends
    square to lo = 0 with hi_hi != 0 -- lanes that borrow beside lanes that do not, in one wave and in a wave where every lane borrows."""
    import poseidon_model as pm
    rc = [0] * 360
    s = _states(6, 192)
    special = [2**48, 3 * 2**48, 2**63, 2**56 + 2**48, P - 2**48, 2**48 + 1, 2**62, 5 * 2**52]
    for k, v in enumerate(special):
        s[8 + k] = v                      # wave 0: a few borrowing lanes among random ones
    s[128:192] = 2**48                    # wave 2: every lane borrows
    s[128:192, 3] = np.arange(64, dtype=np.uint64) * np.uint64(2**48)
    with tmx.Context(4, b"celestia") as c:
        c.poseidon_set_constants(rc, pm.MDS_CIRC, pm.MDS_DIAG)
        got = c.poseidon_permute(s)
        model = pm.Poseidon(rc, pm.MDS_CIRC, pm.MDS_DIAG)
        for i in (0, 8, 9, 10, 11, 12, 13, 14, 15, 128, 150, 191):
            assert [int(x) for x in got[i]] == model.permute([int(x) % P for x in s[i]]), i
        try:
            oracle.poseidon_set_constants(rc, pm.MDS_CIRC, pm.MDS_DIAG)
            assert np.array_equal(got, oracle.poseidon_permute(s))
        finally:
            oracle.poseidon_set_constants(pm.grain_constants(), pm.MDS_CIRC, pm.MDS_DIAG)


@pytest.mark.parametrize("log_n,n_cols,cap", [(3, 3, 0), (4, 4, 2), (6, 5, 1), (8, 8, 4), (10, 9, 0), (9, 20, 3), (12, 135, 4), (5, 300, 5)])
def test_merkle_vs_oracle(ctx, oracle, log_n, n_cols, cap):
    import torch
    rng = np.random.default_rng(1000 * log_n + n_cols)
    cols = rng.integers(0, 2**64, n_cols << log_n, dtype=np.uint64)
    dev = torch.device("cuda", 0)
    d_cols = torch.from_numpy(cols.view(np.int64)).to(dev)
    nd = ctx.poseidon_merkle_digests(log_n, cap)
    d_lv = torch.full((nd, 4), -1, dtype=torch.int64, device=dev)
    ctx.poseidon_merkle_device(log_n, n_cols, d_cols.data_ptr(), cap, d_lv.data_ptr(), 0)
    torch.cuda.synchronize(dev)
    got = d_lv.cpu().numpy().view(np.uint64)
    want = oracle.poseidon_merkle(cols, log_n, n_cols, cap)
    assert got.shape == want.shape and np.array_equal(got, want)


def test_merkle_of_an_lde_at_full_size_properties(ctx, oracle):
    """2^16 -> 2^19 coset LDE of 64 columns, committed with a cap of 2^4: (a) the leaf digests of sampled rows equal the oracle's hash of
    those rows, (b) every sampled inner node is two_to_one of its children (checked by the oracle), (c) committing twice gives the same
    cap, (d) changing ONE element of ONE column changes the leaf of its row, exactly one node per level above it and exactly one cap entry."""
    import torch
    import oracle_c as oc
    log_n, blow, n_cols, cap = 16, 3, 64, 4
    log_m = log_n + blow
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(3)
    x = torch.from_numpy(rng.integers(0, P, n_cols << log_n, dtype=np.uint64).view(np.int64)).to(dev)
    y = torch.empty(n_cols << log_m, dtype=torch.int64, device=dev)
    ctx.lde_device(log_n, blow, n_cols, x.data_ptr(), y.data_ptr(), 0)
    nd = ctx.poseidon_merkle_digests(log_m, cap)
    lv = torch.empty((nd, 4), dtype=torch.int64, device=dev)
    ctx.poseidon_merkle_device(log_m, n_cols, y.data_ptr(), cap, lv.data_ptr(), 0)
    torch.cuda.synchronize(dev)
    a = lv.cpu().numpy().view(np.uint64).copy()
    ynp = y.cpu().numpy().view(np.uint64)
    L = oc.lib()
    import ctypes as C
    L.tmxo_poseidon_hash_no_pad.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
    L.tmxo_poseidon_two_to_one.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    out = np.zeros(4, dtype=np.uint64)
    for r in (0, 1, 12345, (1 << log_m) - 1):
        row = np.ascontiguousarray(ynp[r::1 << log_m][:n_cols])
        L.tmxo_poseidon_hash_no_pad(row.ctypes.data, n_cols, out.ctypes.data)
        assert np.array_equal(a[r], out)
    off = 0
    for k in range(log_m - cap):
        cnt = 1 << (log_m - k)
        for i in (0, 1, cnt // 2 - 1, (7919 * (k + 1)) % (cnt // 2)):
            l, rr = np.ascontiguousarray(a[off + 2 * i]), np.ascontiguousarray(a[off + 2 * i + 1])
            L.tmxo_poseidon_two_to_one(l.ctypes.data, rr.ctypes.data, out.ctypes.data)
            assert np.array_equal(a[off + cnt + i], out), (k, i)
        off += cnt
    lv2 = torch.empty_like(lv)
    ctx.poseidon_merkle_device(log_m, n_cols, y.data_ptr(), cap, lv2.data_ptr(), 0)
    assert torch.equal(lv, lv2)
    row, col = 54321, 17
    y[(col << log_m) + row] += 1
    ctx.poseidon_merkle_device(log_m, n_cols, y.data_ptr(), cap, lv2.data_ptr(), 0)
    torch.cuda.synchronize(dev)
    b = lv2.cpu().numpy().view(np.uint64)
    diff = np.argwhere((a != b).any(axis=1)).ravel()
    expect, off, idx = [], 0, row
    for k in range(log_m - cap + 1):
        expect.append(off + idx)
        off += 1 << (log_m - k)
        idx >>= 1
    assert diff.tolist() == expect
