"""GPU parity of the Goldilocks NTT / coset LDE (tmx_ntt_goldilocks_device, tmx_lde_goldilocks_device) against oracle/c/tmxo_ntt.c:
bit-exact at every size the oracle finishes in seconds, size-independent properties at the full four-step sizes."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

P = 2**64 - 2**32 + 1


@pytest.fixture(scope="module")
def env():
    import torch
    import tendermintx_amd as tmx
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "py"))
    import oracle_c as oc
    ctx = tmx.Context(4, b"celestia", max_batch=1)
    yield torch, ctx, oc
    ctx.close()


def test_domain_conventions(env):
    """Default domain = the constants recalled from plonky2's GoldilocksField; tmx_ntt_set_domain switches library and oracle to g = 7
    (Plonky3 / winterfell) and back: the transform of a delta shows the root in use, an LDE the coset shift."""
    torch, ctx, oc = env
    s = torch.cuda.current_stream().cuda_stream
    delta = np.zeros((1, 16), dtype=np.uint64)
    delta[0, 1] = 1
    try:
        for root, shift in (oc.PLONKY2_DOMAIN, oc.G7_DOMAIN, oc.PLONKY2_DOMAIN):
            ctx.ntt_set_domain(root, shift)
            oc.ntt_set_domain(root, shift)
            d = _to_dev(torch, delta)
            out = torch.empty_like(d)
            ctx.ntt_device(4, 1, d.data_ptr(), out.data_ptr(), False, s)
            torch.cuda.synchronize()
            w = pow(root, 1 << 28, P)
            assert _to_host(out)[0].tolist() == [pow(w, j, P) for j in range(16)] == oc.ntt(delta[0]).tolist()
            x = np.arange(1, 9, dtype=np.uint64).reshape(1, 8)
            dx = _to_dev(torch, x)
            lo = torch.empty(32, dtype=torch.int64, device="cuda:0")
            ctx.lde_device(3, 2, 1, dx.data_ptr(), lo.data_ptr(), s)
            torch.cuda.synchronize()
            assert np.array_equal(_to_host(lo), oc.lde(x[0], 2))
        with pytest.raises(Exception):
            ctx.ntt_set_domain(5, 7)          # 5 is not a primitive 2^32-th root of unity
    finally:
        ctx.ntt_set_domain(*oc.PLONKY2_DOMAIN)
        oc.ntt_set_domain(*oc.PLONKY2_DOMAIN)


def _to_dev(torch, a):
    return torch.from_numpy(a.view(np.int64)).to("cuda:0")


def _to_host(t):
    return t.cpu().numpy().view(np.uint64)


def _rand(rng, shape):
    a = rng.integers(0, P, size=shape, dtype=np.uint64)
    flat = a.reshape(-1)
    flat[:4] = [P - 1, 0, 1, 2**64 - 1][: min(4, flat.size)]   # edge values, incl. a non-canonical input (taken mod p)
    return a


@pytest.mark.parametrize("log_n,cols", [(0, 5), (1, 3), (2, 1), (5, 7), (8, 33), (10, 4), (11, 3), (12, 2), (13, 5), (16, 2), (17, 1)])
def test_ntt_matches_oracle(env, log_n, cols):
    torch, ctx, oc = env
    rng = np.random.default_rng(1000 + log_n)
    x = _rand(rng, (cols, 1 << log_n))
    d = _to_dev(torch, x)
    out = torch.empty_like(d)
    s = torch.cuda.current_stream().cuda_stream
    ctx.ntt_device(log_n, cols, d.data_ptr(), out.data_ptr(), False, s)
    torch.cuda.synchronize()
    want = oc.ntt(x)
    assert np.array_equal(_to_host(out), want)
    ctx.ntt_device(log_n, cols, out.data_ptr(), out.data_ptr(), True, s)   # in place, inverse
    torch.cuda.synchronize()
    assert np.array_equal(_to_host(out), x % np.uint64(P))
    assert np.array_equal(oc.ntt(want, inverse=True), x % np.uint64(P))


@pytest.mark.parametrize("log_n,cols", [(4, 3), (10, 4), (11, 8), (13, 2), (16, 1)])
@pytest.mark.parametrize("fill", ["all_ones", "p_minus_1", "above_p", "extremes"])
def test_ntt_noncanonical_and_extreme_inputs(env, log_n, cols, fill):
    """The kernel keeps values as arbitrary 64-bit representatives between its stages (only the second operand of a butterfly and the
    final store are canonical): inputs at and above p, all 2^64 - 1, and alternating extremes must still come out bit-exact."""
    torch, ctx, oc = env
    n = 1 << log_n
    rng = np.random.default_rng(7000 + log_n)
    if fill == "all_ones":
        x = np.full((cols, n), 2**64 - 1, dtype=np.uint64)
    elif fill == "p_minus_1":
        x = np.full((cols, n), P - 1, dtype=np.uint64)
    elif fill == "above_p":
        x = rng.integers(P, 2**64, size=(cols, n), dtype=np.uint64, endpoint=False)
    else:
        x = rng.choice(np.array([0, 1, P - 1, P, P + 1, 2**64 - 1, 2**32 - 1, 2**32, 2**63], dtype=np.uint64), size=(cols, n))
    d = _to_dev(torch, x)
    out = torch.empty_like(d)
    s = torch.cuda.current_stream().cuda_stream
    for inverse in (False, True):
        ctx.ntt_device(log_n, cols, d.data_ptr(), out.data_ptr(), inverse, s)
        torch.cuda.synchronize()
        assert np.array_equal(_to_host(out), oc.ntt(x, inverse=inverse)), (fill, inverse)


@pytest.mark.parametrize("log_n", range(0, 19))
def test_ntt_every_size_and_odd_column_counts(env, log_n):
    """Every transform size up to 2^18 with 1, 2, 3, 5 and 17 columns: every instantiation of the tile kernel (sub-transform lengths 2^0 ..
    2^11, the fixed tile shapes and the generic one that few columns fall back to, partially filled tiles), forward and inverse."""
    torch, ctx, oc = env
    s = torch.cuda.current_stream().cuda_stream
    for cols in (1, 2, 3, 5, 17):
        if log_n >= 17 and cols > 3:
            continue
        rng = np.random.default_rng(31 * log_n + cols)
        x = _rand(rng, (cols, 1 << log_n))
        d = _to_dev(torch, x)
        out = torch.empty_like(d)
        for inverse in (False, True):
            ctx.ntt_device(log_n, cols, d.data_ptr(), out.data_ptr(), inverse, s)
            torch.cuda.synchronize()
            assert np.array_equal(_to_host(out), oc.ntt(x, inverse=inverse)), (log_n, cols, inverse)


# (round 4: the scale vector rides on the inverse transform's last pass and the forward transform skips the zero padding -- one-pass and
# two-pass shapes of either transform, many columns (the fixed-shape tile kernels), and a split whose N2 exceeds N: the padded form)
@pytest.mark.parametrize("log_n,log_blowup,cols", [(0, 1, 2), (3, 3, 3), (8, 2, 5), (10, 3, 2), (12, 1, 3), (13, 3, 1), (11, 3, 70), (15, 3, 9),
                                                   (4, 8, 3), (6, 6, 17), (9, 2, 33), (12, 3, 16)])
def test_lde_matches_oracle(env, log_n, log_blowup, cols):
    torch, ctx, oc = env
    rng = np.random.default_rng(50 * log_n + log_blowup)
    x = _rand(rng, (cols, 1 << log_n))
    d = _to_dev(torch, x)
    out = torch.empty((cols, 1 << (log_n + log_blowup)), dtype=torch.int64, device="cuda:0")
    ctx.lde_device(log_n, log_blowup, cols, d.data_ptr(), out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(_to_host(out), oc.lde(x, log_blowup))


def test_full_size_properties(env):
    """2^20 x 8 columns and the largest size 2^22: round trip, linearity, a delta goes to the powers of omega (spot-checked with
    big-int arithmetic), LDE of a constant is the constant."""
    torch, ctx, oc = env
    s = torch.cuda.current_stream().cuda_stream
    rng = np.random.default_rng(99)
    for log_n, cols in ((20, 8), (22, 2)):
        n = 1 << log_n
        a, b = _rand(rng, (cols, n)), _rand(rng, (cols, n))
        da, db = _to_dev(torch, a), _to_dev(torch, b)
        fa, fb = torch.empty_like(da), torch.empty_like(db)
        ctx.ntt_device(log_n, cols, da.data_ptr(), fa.data_ptr(), False, s)
        ctx.ntt_device(log_n, cols, db.data_ptr(), fb.data_ptr(), False, s)
        back = torch.empty_like(da)
        ctx.ntt_device(log_n, cols, fa.data_ptr(), back.data_ptr(), True, s)
        torch.cuda.synchronize()
        assert np.array_equal(_to_host(back), a % np.uint64(P))
        # linearity on a sample of positions (big-int reference)
        ab = ((a.astype(object) + b.astype(object)) % P).astype(np.uint64)
        dab = _to_dev(torch, ab)
        fab = torch.empty_like(dab)
        ctx.ntt_device(log_n, cols, dab.data_ptr(), fab.data_ptr(), False, s)
        torch.cuda.synchronize()
        ha, hb, hab = _to_host(fa), _to_host(fb), _to_host(fab)
        idx = rng.integers(0, n, size=64)
        for c in range(cols):
            for i in idx:
                assert (int(ha[c, i]) + int(hb[c, i])) % P == int(hab[c, i])
        # delta at position 3 -> omega^(3 j)
        delta = np.zeros((1, n), dtype=np.uint64)
        delta[0, 3] = 1
        dd = _to_dev(torch, delta)
        fd = torch.empty_like(dd)
        ctx.ntt_device(log_n, 1, dd.data_ptr(), fd.data_ptr(), False, s)
        torch.cuda.synchronize()
        hd = _to_host(fd)
        w = oc.gl_root(log_n)
        for j in [0, 1, 2, 12345, n // 2, n - 1]:
            assert int(hd[0, j]) == pow(w, 3 * j, P)
    const = np.full((2, 1 << 18), 123456789, dtype=np.uint64)
    dc = _to_dev(torch, const)
    out = torch.empty((2, 1 << 21), dtype=torch.int64, device="cuda:0")
    ctx.lde_device(18, 3, 2, dc.data_ptr(), out.data_ptr(), s)
    torch.cuda.synchronize()
    assert (_to_host(out) == 123456789).all()


def test_bad_arguments(env):
    torch, ctx, oc = env
    import tendermintx_amd as tmx
    d = torch.zeros(8, dtype=torch.int64, device="cuda:0")
    with pytest.raises(tmx.TmxError):
        ctx.ntt_device(23, 1, d.data_ptr(), d.data_ptr())
    with pytest.raises(tmx.TmxError):
        ctx.lde_device(20, 3, 1, d.data_ptr(), d.data_ptr())
