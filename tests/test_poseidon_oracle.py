"""Poseidon over Goldilocks + Merkle caps, CPU tier: the C oracle (oracle/c/tmxo_poseidon.c) against the independent Python model
(oracle/py/poseidon_model.py), the Grain-LFSR constant stream against a second implementation, and the algebraic self-checks that hold for
ANY correct Poseidon instance -- the permutation is a bijection (inverse S-box x^d, 7 d = 1 mod p - 1; MDS^-1), the MDS matrix is
invertible and circulant-plus-diagonal.  plonky2's own tables are not in the reference tree (Cargo.lock:2957-2982): parity unpinned, the
constants are injectable (tmx_poseidon_set_constants) and the injection path is tested with a second table."""
import numpy as np
import pytest

import poseidon_model as pm

P = pm.P


def test_grain_stream_two_implementations(oracle):
    rc, circ, diag = oracle.poseidon_constants()
    assert rc == pm.grain_constants() and len(set(rc)) == 360 and all(0 <= x < P for x in rc)
    assert circ == pm.MDS_CIRC and diag == pm.MDS_DIAG
    assert rc[0] != 0xb585f766f2144405     # NOT plonky2's table (its first constant as recalled): the defaults are the paper's stream


def test_mds_is_invertible_and_small():
    m = pm.mds_matrix()
    inv = pm.mat_inverse(m)
    ident = [[sum(m[r][k] * inv[k][c] for k in range(12)) % P for c in range(12)] for r in range(12)]
    assert ident == [[1 if r == c else 0 for c in range(12)] for r in range(12)]
    assert max(max(row) for row in m) < 64          # what the kernel's 32-bit-limb MDS relies on: sum of 12 products fits 2^6 * 12 * 2^32


def test_sbox_exponent_is_a_permutation_of_the_field():
    from math import gcd
    assert gcd(7, P - 1) == 1 and all(gcd(a, P - 1) != 1 for a in (2, 3, 5))   # 7 is the smallest exponent that works for this p
    d = pow(7, -1, P - 1)
    for x in (0, 1, 2, P - 1, 0x123456789abcdef, 2**63):
        assert pow(pow(x, 7, P), d, P) == x % P


def test_permutation_c_vs_model_and_bijection(oracle):
    rng = np.random.default_rng(5)
    states = [[0] * 12, [P - 1] * 12, list(range(12)), [2**64 - 1] * 12] + rng.integers(0, 2**63, (12, 12), dtype=np.uint64).tolist()
    pos = pm.Poseidon()
    got = oracle.poseidon_permute(np.array(states, dtype=np.uint64))
    for s, g in zip(states, got):
        want = pos.permute(s)
        assert [int(x) for x in g] == want
        assert pos.permute_inverse(want) == [x % P for x in s]          # the permutation is a bijection


def test_injected_constants_reach_both(oracle):
    rng = np.random.default_rng(6)
    rc = [int(x) % P for x in rng.integers(0, 2**63, 360, dtype=np.uint64)]
    circ = [int(x) for x in rng.integers(1, 60, 12)]
    try:
        oracle.poseidon_set_constants(rc, circ, [3] + [0] * 11)
        s = list(range(100, 112))
        got = [int(x) for x in oracle.poseidon_permute(np.array([s], dtype=np.uint64))[0]]
        assert got == pm.Poseidon(rc, circ, [3] + [0] * 11).permute(s)
    finally:
        oracle.poseidon_set_constants(pm.grain_constants(), pm.MDS_CIRC, pm.MDS_DIAG)


@pytest.mark.parametrize("log_n,n_cols,cap", [(3, 3, 0), (4, 5, 2), (5, 20, 1), (4, 8, 4), (6, 9, 3)])
def test_merkle_c_vs_model(oracle, log_n, n_cols, cap):
    rng = np.random.default_rng(log_n * 100 + n_cols)
    cols = rng.integers(0, 2**64, n_cols << log_n, dtype=np.uint64)
    got = oracle.poseidon_merkle(cols, log_n, n_cols, cap)
    rows = [[int(cols[(c << log_n) + r]) for c in range(n_cols)] for r in range(1 << log_n)]
    levels = pm.Poseidon().merkle(rows, cap)
    flat = [d for lvl in levels for d in lvl]
    assert len(levels[-1]) == 1 << cap and got.shape[0] == len(flat)
    assert [[int(x) for x in d] for d in got] == flat
