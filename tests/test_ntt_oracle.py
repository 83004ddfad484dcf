"""CPU tests of the Goldilocks NTT / LDE oracle (oracle/c/tmxo_ntt.c) against an independent pure-Python big-int model: the
definition itself (O(n^2) DFT), known constants, and the size-independent properties the GPU tests rely on at full size."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle", "py"))
import oracle_c as oc  # noqa: E402

P = 2**64 - 2**32 + 1


DOMAIN = {"root": oc.PLONKY2_DOMAIN[0], "shift": oc.PLONKY2_DOMAIN[1]}   # the default of both the oracle and the library


@pytest.fixture(params=["plonky2", "g7"], autouse=True)
def domain(request):
    """Every test runs on both conventions: the constants recalled from plonky2's GoldilocksField (default) and g = 7."""
    root, shift = oc.PLONKY2_DOMAIN if request.param == "plonky2" else oc.G7_DOMAIN
    DOMAIN["root"], DOMAIN["shift"] = root, shift
    oc.ntt_set_domain(root, shift)
    yield request.param
    oc.ntt_set_domain(*oc.PLONKY2_DOMAIN)
    DOMAIN["root"], DOMAIN["shift"] = oc.PLONKY2_DOMAIN


def _root(log_n):
    return pow(DOMAIN["root"], 1 << (32 - log_n), P)


def _dft(x, inverse=False):
    n = len(x)
    w = _root(n.bit_length() - 1)
    if inverse:
        w = pow(w, P - 2, P)
    out = [sum(int(x[i]) * pow(w, i * j, P) for i in range(n)) % P for j in range(n)]
    if inverse:
        ninv = pow(n, P - 2, P)
        out = [(v * ninv) % P for v in out]
    return out


def test_constants(domain):
    # g = 7 (Plonky3 / winterfell): the 2^32-th root of unity it generates
    assert pow(7, (P - 1) >> 32, P) == 1753635133440165772 == 0x185629DCDA58878C == oc.G7_DOMAIN[0]
    # the pair recalled from plonky2's GoldilocksField is self-consistent: POWER_OF_TWO_GENERATOR = MULTIPLICATIVE_GROUP_GENERATOR^((p-1)/2^32),
    # and the latter generates the whole multiplicative group (p - 1 = 2^32 * 3 * 5 * 17 * 257 * 65537)
    root, gen = oc.PLONKY2_DOMAIN
    assert pow(gen, (P - 1) >> 32, P) == root and all(pow(gen, (P - 1) // q, P) != 1 for q in (2, 3, 5, 17, 257, 65537))
    assert oc.gl_root(32) == DOMAIN["root"] and oc.gl_root(1) == P - 1 and oc.gl_root(0) == 1
    for k in range(1, 33):
        w = oc.gl_root(k)
        assert w == _root(k) and pow(w, 1 << k, P) == 1 and pow(w, 1 << (k - 1), P) == P - 1


@pytest.mark.parametrize("log_n", [0, 1, 2, 3, 5, 7])
def test_ntt_is_the_dft(log_n):
    rng = np.random.default_rng(100 + log_n)
    n = 1 << log_n
    x = rng.integers(0, P, size=n, dtype=np.uint64)
    x[: min(n, 3)] = [P - 1, 0, 1][: min(n, 3)]
    assert oc.ntt(x).tolist() == _dft(x)
    assert oc.ntt(x, inverse=True).tolist() == _dft(x, inverse=True)
    assert np.array_equal(oc.ntt(oc.ntt(x), inverse=True), x)


def test_ntt_properties_large():
    rng = np.random.default_rng(7)
    n = 1 << 12
    a = rng.integers(0, P, size=n, dtype=np.uint64)
    b = rng.integers(0, P, size=n, dtype=np.uint64)
    fa, fb = oc.ntt(a), oc.ntt(b)
    s = np.array([(int(u) + int(v)) % P for u, v in zip(a, b)], dtype=np.uint64)
    assert oc.ntt(s).tolist() == [(int(u) + int(v)) % P for u, v in zip(fa, fb)]          # linearity
    delta = np.zeros(n, dtype=np.uint64)
    delta[1] = 1
    w = _root(12)
    assert oc.ntt(delta)[:4].tolist() == [1, w, w * w % P, pow(w, 3, P)]                   # delta_1 -> powers of omega
    # cyclic convolution theorem on a sparse pair
    c = np.zeros(n, dtype=np.uint64)
    d = np.zeros(n, dtype=np.uint64)
    c[[0, 5, n - 1]] = [3, P - 2, 11]
    d[[1, 2]] = [7, 13]
    conv = np.zeros(n, dtype=object)
    for i in (0, 5, n - 1):
        for j in (1, 2):
            conv[(i + j) % n] = (conv[(i + j) % n] + int(c[i]) * int(d[j])) % P
    prod = np.array([(int(u) * int(v)) % P for u, v in zip(oc.ntt(c), oc.ntt(d))], dtype=np.uint64)
    assert oc.ntt(prod, inverse=True).tolist() == [int(v) for v in conv]


@pytest.mark.parametrize("log_n,log_blowup", [(0, 1), (2, 1), (3, 3), (6, 2)])
def test_lde_evaluates_the_interpolant_on_the_coset(log_n, log_blowup):
    rng = np.random.default_rng(31 * log_n + log_blowup)
    n, m = 1 << log_n, 1 << (log_n + log_blowup)
    x = rng.integers(0, P, size=n, dtype=np.uint64)
    coeffs = _dft(x, inverse=True)
    wm = _root(log_n + log_blowup)
    want = [sum(c * pow(DOMAIN["shift"] * pow(wm, j, P) % P, i, P) for i, c in enumerate(coeffs)) % P for j in range(m)]
    assert oc.lde(x, log_blowup).tolist() == want
