"""Property tests (hypothesis) of the CPU oracle: the plain-C restatement against the pure-Python model and against first principles,
on the byte-level pieces whose semantics the reference's gadgets fix (SURVEY §4 asks for these)."""
import hashlib

from hypothesis import given, settings
from hypothesis import strategies as st

import ed25519_model as ed
import tm_encoding as tm
import tmx_model as m

u63 = st.integers(min_value=0, max_value=2**63 - 1)
u64 = st.integers(min_value=0, max_value=2**64 - 1)


@settings(max_examples=300, deadline=None)
@given(u63)
def test_varint9_roundtrip_and_shape(oracle, v):
    """shared.rs:67-156: nine bytes, continuation bits below the last non-zero septet, trailing zeros; decodes back to v"""
    b = oracle.varint9(v)
    assert b == tm.varint9(v) and len(b) == 9
    std = tm.varint(v)
    assert b[:len(std)] == std and b[len(std):] == bytes(9 - len(std))
    x = 0
    for i, byte in enumerate(b):
        x |= (byte & 0x7F) << (7 * i)
    assert x == v


@settings(max_examples=200, deadline=None)
@given(st.binary(min_size=32, max_size=32), u63)
def test_marshal_and_leaf(oracle, pk, power):
    """validator.rs:185-229: 0a 22 0a 20 pk 10 varint9; the leaf hash over 1 + byte_length bytes equals the protobuf leaf"""
    mv = oracle.marshal_validator(pk, power)
    assert mv == b"\x0a\x22\x0a\x20" + pk + b"\x10" + tm.varint9(power)
    wire = tm.validator_bytes(pk, power)
    if power:
        assert mv[:len(wire)] == wire
        assert oracle.sha256(b"\x00" + mv[:len(wire)]) == tm.leaf_hash(wire)


@settings(max_examples=150, deadline=None)
@given(st.lists(u63, min_size=1, max_size=24), st.data())
def test_tally_matches_model_and_bigint(oracle, powers, data):
    """voting.rs:31-109 with wrap-around semantics; cross-checked against unbounded integers"""
    n = len(powers)
    nb = data.draw(st.integers(min_value=0, max_value=n + 2))
    group = data.draw(st.lists(st.booleans(), min_size=n, max_size=n))
    num, den = data.draw(st.sampled_from([(1, 3), (2, 3)]))
    c, p = oracle.tally(powers, nb, group, num, den), m.tally(powers, nb, group, num, den)
    for k in ("gt", "total", "acc", "scaled_acc", "scaled_total", "tot_prefix", "acc_prefix", "no_overflow"):
        assert c[k] == p[k], k
    total = sum(powers[:nb]) if nb < n else sum(powers)
    acc = sum(x for x, g in zip(powers, group) if g)
    if max(total * num, acc * den, total, acc) < 2**64:
        assert c["no_overflow"] and c["gt"] == (acc * den > total * num)
    else:
        assert not c["no_overflow"]


@settings(max_examples=60, deadline=None)
@given(st.integers(min_value=1, max_value=70), st.data())
def test_fixed_shape_tree_property(oracle, n, data):
    nb = data.draw(st.integers(min_value=1, max_value=n))
    leaves = [hashlib.sha256(bytes([i & 255, i >> 8, n])).digest() for i in range(n)]
    nodes, root = oracle.fixed_shape_tree(leaves, nb)
    assert root == tm.root_from_leaf_hashes(leaves[:nb])
    assert len(nodes) == 32 * m.tree_nodes_count(n)


@settings(max_examples=25, deadline=None)
@given(st.binary(min_size=32, max_size=32), st.binary(min_size=0, max_size=124))
def test_sign_verify_roundtrip(oracle, seed, msg):
    pk, sig = oracle.pubkey(seed), oracle.sign(seed, msg)
    t = oracle.eddsa_trace(pk, sig, msg)
    assert t["ok"] and t["pt"][4] == t["pt"][8] and t["pt"][5] == t["pt"][9]       # s*B == R + h*A, affine
    assert int.from_bytes(t["h"], "little") == int.from_bytes(hashlib.sha512(sig[:32] + pk + msg).digest(), "little") % ed.L
    flipped = bytes([sig[0] ^ 1]) + sig[1:]
    assert not oracle.eddsa_trace(pk, flipped, msg)["ok"]
