"""Pins the oracle (oracle/c plain-C restatement and oracle/py big-int model) against every known answer the
reference's own tests hold for this path, RFC 8032, hashlib, and the committed goldens.  CPU only."""
import hashlib
import json
import os

import numpy as np
import pytest

import ed25519_model as ed
import tm_encoding as tm
import tmx_model as m
from conftest import GOLDEN


def test_sha2_matches_hashlib(oracle):
    rng = np.random.default_rng(1)
    for n in [0, 1, 3, 55, 56, 63, 64, 65, 111, 112, 119, 120, 127, 128, 129, 188, 239, 240, 1000]:
        b = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        assert oracle.sha256(b) == hashlib.sha256(b).digest()
        assert oracle.sha512(b) == hashlib.sha512(b).digest()
    assert oracle.sha256(b"abc").hex() == "ba7816bf8f01cfea414140de5dae2223b00361a396177a9cb410ff61f20015ad"  # FIPS 180-4


def test_varint_table(oracle, kat):
    """reference circuits/builder/shared.rs:236-250"""
    for value, hexs in kat["varint"]:
        want = bytes.fromhex(hexs)
        for got in (oracle.varint9(value), tm.varint9(value)):
            assert got[:len(want)] == want and got[len(want):] == bytes(9 - len(want))
        assert tm.varint(value) == want


def test_marshal_validator(oracle, kat):
    """reference circuits/builder/validator.rs:282-287"""
    k = kat["marshal"]
    want = bytes.fromhex(k["expected"])
    got = oracle.marshal_validator(bytes.fromhex(k["pubkey"]), k["power"])
    assert got[:len(want)] == want and got[len(want):] == bytes(46 - len(want))
    assert tm.validator_bytes(bytes.fromhex(k["pubkey"]), k["power"]) == want


def test_validators_hash_table(oracle, kat):
    """reference circuits/builder/validator.rs:333-338, 359-362, 383: in-circuit root == native RFC-6962 root"""
    for batch, root in zip(kat["validators_hash"], kat["validators_hash_roots"]):
        leaves = [oracle.sha256(b"\x00" + bytes.fromhex(x)) for x in batch]
        assert oracle.rfc6962_root(leaves).hex() == root
        _, fixed_root = oracle.fixed_shape_tree(leaves, len(leaves))
        assert fixed_root.hex() == root
        assert tm.fixed_shape_layers(leaves, len(leaves))[1].hex() == root


def test_hash_in_message_vector(kat):
    """reference circuits/builder/verify.rs:597-601"""
    k = kat["hash_in_message"]
    msg = bytes.fromhex(k["message"]).ljust(124, b"\0")
    chk = m.sigdata_checks(msg, bytes.fromhex(k["header"]), 0x232de, k["round"], True, True)
    assert chk[0] and chk[1] and chk[2]  # hash in message, precommit, height 144094 little-endian at [4..12]


def test_threshold_table(oracle, kat):
    """reference circuits/builder/voting.rs:127-146 (2/3 threshold, total = sum of all four powers)"""
    for powers, in_group, expect in kat["threshold"]:
        for t in (oracle.tally(powers, 4, in_group, 2, 3), m.tally(powers, 4, in_group, 2, 3)):
            assert t["gt"] == expect and t["total"] == sum(powers) and t["no_overflow"]


def test_rfc8032_vectors(oracle, kat):
    for seed, pk, msg, sig in kat["rfc8032"]:
        seed, pk, msg, sig = (bytes.fromhex(x) for x in (seed, pk, msg, sig))
        assert oracle.pubkey(seed) == pk and ed.keypair_from_seed(seed)[2] == pk
        assert oracle.sign(seed, msg) == sig and ed.sign(seed, msg) == sig
        assert oracle.eddsa_trace(pk, sig, msg)["ok"] and ed.verify_trace(pk, sig, msg)["ok"]
        bad = bytearray(sig)
        bad[3] ^= 0x10
        assert not oracle.eddsa_trace(pk, bytes(bad), msg)["ok"]


def test_dummy_constants(oracle, kat):
    """DUMMY_PUBLIC_KEY / DUMMY_SIGNATURE (imported from plonky2x at reference conversion.rs:3-5) = RFC 8032 pair of seed
    01x32 over 00x32; the literals embedded in kernels.hip / codec.cpp must equal the derived values."""
    d = kat["dummy"]
    pk, sig = oracle.dummy()
    assert pk.hex() == d["public_key"] and sig.hex() == d["signature"]
    assert ed.DUMMY_PUBLIC_KEY.hex() == d["public_key"] and ed.DUMMY_SIGNATURE.hex() == d["signature"]
    assert oracle.eddsa_trace(pk, sig, bytes(32))["ok"]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    words = [int.from_bytes(pk[4 * i:4 * i + 4], "little") for i in range(8)] + [int.from_bytes(sig[4 * i:4 * i + 4], "little") for i in range(16)]
    src = open(os.path.join(root, "tendermintx_amd", "csrc", "kernels.hip")).read().lower()
    for w in words:
        assert f"0x{w:08x}u" in src
    csrc = open(os.path.join(root, "tendermintx_amd", "csrc", "codec.cpp")).read().lower().replace(" ", "").replace("\n", "")
    assert ",".join(f"0x{b:02x}" for b in pk) in csrc and ",".join(f"0x{b:02x}" for b in sig) in csrc


def test_eddsa_c_equals_bigint_model(oracle):
    rng = np.random.default_rng(7)
    for i in range(12):
        seed = rng.integers(0, 256, 32, dtype=np.uint8).tobytes()
        msg = rng.integers(0, 256, 5 * i, dtype=np.uint8).tobytes()
        pk, sig = oracle.pubkey(seed), oracle.sign(seed, msg)
        t, p = oracle.eddsa_trace(pk, sig, msg), ed.verify_trace(pk, sig, msg)
        assert t["ok"] and p["ok"] and t["digest"] == p["digest"] and int.from_bytes(t["h"], "little") == p["h"]
        for k, name in enumerate(("A", "R", "sB", "hA", "sum")):
            assert int.from_bytes(t["pt"][2 * k], "little") == p[name][0]
            assert int.from_bytes(t["pt"][2 * k + 1], "little") == p[name][1]
    # undecodable public key: y with no matching x
    bad_pk = (2).to_bytes(32, "little")
    assert ed.decompress(bad_pk) is None
    t = oracle.eddsa_trace(bad_pk, oracle.sign(bytes(32), b"x"), b"x")
    assert not t["ok"] and not t["decode_ok"] and all(x == bytes(32) for x in t["pt"])


def test_sc_reduce(oracle):
    rng = np.random.default_rng(3)
    for _ in range(50):
        b = rng.integers(0, 256, 64, dtype=np.uint8).tobytes()
        assert int.from_bytes(oracle.sc_reduce512(b), "little") == int.from_bytes(b, "little") % ed.L
    assert int.from_bytes(oracle.sc_reduce512(b"\xff" * 64), "little") == (2**512 - 1) % ed.L


def test_fixed_shape_tree_equals_rfc6962(oracle):
    """the in-circuit pairwise rule equals the RFC-6962 root of the first nb leaves, every nb, also odd N"""
    for n in [1, 2, 3, 4, 5, 7, 8, 13, 32, 33, 100, 128]:
        leaves = [hashlib.sha256(bytes([i & 255, n & 255, i >> 8])).digest() for i in range(n)]
        for nb in sorted({1, 2, n // 3 + 1, n - 1 if n > 1 else 1, n}):
            nodes, root = oracle.fixed_shape_tree(leaves, nb)
            assert root == oracle.rfc6962_root(leaves[:nb]) == tm.root_from_leaf_hashes(leaves[:nb])
            layers, proot = tm.fixed_shape_layers(leaves, nb)
            assert proot == root and b"".join(b"".join(l) for l in layers) == nodes


def test_oracle_reproduces_goldens(oracle, cases):
    """Every golden case: C oracle == committed sha256 / full element stream / Level-0 header / verdicts."""
    for name, c in cases.items():
        proof, target = bytes.fromhex(c["proof"]), bytes.fromhex(c["target"])
        trusted = bytes.fromhex(c["trusted"]) if c["trusted"] else None
        w, rep = oracle.witness(c["kind"], proof, target, trusted, c["chain_id"].encode(), c["skip_max"])
        assert len(w) == c["elem_count"] == oracle.elem_count(c["kind"], c["n"]), name
        assert hashlib.sha256(w.tobytes()).hexdigest() == c["elems_sha256"], name
        assert rep["header"].hex() == c["header"] and rep["all_ok"] == c["all_ok"] and rep["fail_mask"] == c["fail_mask"], name
        assert rep["first_bad_sig"] == c["first_bad_sig"] and rep["gt_target"] == c["gt_target"], name
        path = os.path.join(GOLDEN, f"elems_{name}.npz")
        if os.path.exists(path):
            assert np.array_equal(np.load(path)["elems"], w), name
        assert int(w.max()) < 2**32  # every element is a canonical Goldilocks value


def test_signed_block_fixtures_pin_the_oracle(oracle):
    """The nine fixture heights of the reference that hold only `signed_block.json` (SignedBlockResponse, tendermint_utils.rs:52-55, 97-112;
    mocha-4 10002-10004, 11000, 11001, 11105, 15000, 50000, 157000: 2 / 8 / 9 / 34 / 100 validators, 21-of-34 and 47-of-100 signing -- the only
    real data with many absent votes): header tree root = commit.block_id.hash, validator tree root = header.validators_hash, and every
    flag-2 signature verifies under the cofactor-less equation the circuit checks AND under RFC 8032's -- by the C oracle, from the
    committed data alone (tests/golden/signed_blocks.json holds what oracle/gen_golden.py derived with the big-int model)."""
    sb = json.load(open(os.path.join(GOLDEN, "signed_blocks.json")))
    f = m.FixtureFetcher(os.path.join(GOLDEN, "fixtures", "mocha-4"))
    assert sorted(map(int, sb)) == [10002, 10003, 10004, 11000, 11001, 11105, 15000, 50000, 157000]
    n_sigs = 0
    for h, g in sb.items():
        sh, vs = f.signed_header(int(h)), f.validators(int(h))
        leaf_hashes = [oracle.sha256(b"\x00" + x) for x in tm.header_leaves(sh["header"])]
        assert oracle.rfc6962_root(leaf_hashes).hex() == g["header_hash"] == sh["commit"]["block_id"]["hash"].lower(), h
        vleaves = []
        for v in vs:   # marshal_tendermint_validator (validator.rs:185-207): 46 bytes, the first validator_byte_length of them hashed (:209-229)
            pk, power = tm.b64(v["pub_key"]["value"]), int(v["voting_power"])
            vleaves.append(oracle.sha256(b"\x00" + oracle.marshal_validator(pk, power)[:len(tm.validator_bytes(pk, power))]))
        assert oracle.rfc6962_root(vleaves).hex() == g["validators_hash"] == sh["header"]["validators_hash"].lower(), h
        assert oracle.fixed_shape_tree(vleaves + [bytes(32)] * (128 - len(vs)), len(vs))[1].hex() == g["validators_hash"], h
        assert [cs["block_id_flag"] for cs in sh["commit"]["signatures"]] == g["flags"] and len(vs) == g["validators"]
        for lane in g["lanes"]:
            if lane is None:
                continue
            t = oracle.eddsa_trace(bytes.fromhex(lane["pubkey"]), bytes.fromhex(lane["signature"]), bytes.fromhex(lane["message"]))
            assert t["ok"] and lane["ok_cofactorless"] and lane["ok_rfc8032"] and t["h"].hex() == lane["h"], h
            n_sigs += 1
    assert n_sigs == 2 + 2 + 2 + 8 + 8 + 8 + 21 + 47 + 53


def test_public_io_level0(cases, kat):
    """Level-0 outputs implied by the reference's end-to-end tests (skip.rs:197-199, 259-262; step.rs:178-180, 237-253)"""
    skip = {(3000, 3100): "skip_3000_3100_n4", (10000, 10500): "skip_10000_10500_n4"}
    for inp, out in kat["public_io"]["skip"]:
        b = bytes.fromhex(inp)
        key = (int.from_bytes(b[:8], "big"), int.from_bytes(b[40:], "big"))
        c = cases[skip[key]]
        assert c["header"] == out and c["all_ok"]
        assert m.unpack_proof(bytes.fromhex(c["proof"]))["hash"] == b[8:40]
    step = {3000: "step_3000_n4", 10000: "step_10000_n2", 10500: "step_10500_n4"}
    for inp, out in kat["public_io"]["step"]:
        b = bytes.fromhex(inp)
        c = cases[step[int.from_bytes(b[:8], "big")]]
        assert c["header"] == out and c["all_ok"]
        assert m.unpack_proof(bytes.fromhex(c["proof"]))["hash"] == b[8:40]


def test_model_equals_c_on_small_cases(oracle, cases):
    for name in ("skip_10000_10500_n4", "step_10500_n4", "skip_10000_10500_n4_wrongchain", "step_10002_n2", "step_11000_n8", "skip_11000_11105_n16"):
        c = cases[name]
        n = c["n"]
        proof, target = bytes.fromhex(c["proof"]), bytes.fromhex(c["target"])
        trusted = bytes.fromhex(c["trusted"]) if c["trusted"] else None
        tl = [target[256 * i:256 * (i + 1)] for i in range(n)]
        rl = [trusted[48 * i:48 * (i + 1)] for i in range(n)] if trusted else None
        w, rep = m.witness(c["kind"], proof, tl, rl, c["chain_id"].encode(), c["skip_max"])
        wc, repc = oracle.witness(c["kind"], proof, target, trusted, c["chain_id"].encode(), c["skip_max"])
        assert np.array_equal(np.array(w, dtype=np.uint64), wc) and rep["fail_mask"] == repc["fail_mask"]


def test_tally_wraparound_semantics(oracle):
    """voting.rs:91-105: u64 products wrap and are caught by the division check; sums that wrap clear no_overflow"""
    big = 2**63 - 1
    for t in (oracle.tally([big, big, 5, 0], 4, [1, 1, 1, 0], 2, 3), m.tally([big, big, 5, 0], 4, [1, 1, 1, 0], 2, 3)):
        assert not t["no_overflow"]
        assert t["total"] == (2 * big + 5) % 2**64
    t = oracle.tally([big, 0, 0, 0], 4, [1, 0, 0, 0], 2, 3)
    assert not t["no_overflow"] and t["scaled_acc"] == (3 * big) % 2**64 and t["scaled_total"] == (2 * big) % 2**64


def test_multithreaded_batch_equals_single(oracle, cases):
    c = cases["skip_10000_10500_n4"]
    proof, target, trusted = (bytes.fromhex(c[k]) for k in ("proof", "target", "trusted"))
    a, ra = oracle.witness_batch(0, 6, proof * 6, target * 6, trusted * 6, 4, b"mocha-4", 100800, n_threads=3)
    b, rb = oracle.witness_batch(0, 6, proof * 6, target * 6, trusted * 6, 4, b"mocha-4", 100800, n_threads=1)
    assert np.array_equal(a, b) and ra == rb and all(r["all_ok"] for r in ra)


def test_sc_fold_model():
    """Model of the radix-2^21 folding reduction used by the HIP path (tendermintx_amd/csrc/sc25519.hpp::sc_reduce512):
    same steps in Python ints, checked against x % l and for intermediate magnitudes < 2^62 (int64 accumulators)."""
    ell = ed.L
    c = ell - 2**252
    digs, t = [], c
    for _ in range(6):
        d = t & ((1 << 21) - 1)
        if d >= 1 << 20:
            d -= 1 << 21
        digs.append(d)
        t = (t - d) >> 21
    assert t == 0
    m = [-d for d in digs]
    assert m == [666643, 470296, 654183, -997805, 136657, -683901]  # the constants hard-coded in sc25519.hpp
    rng = np.random.default_rng(21)
    xs = [0, 1, ell - 1, ell, ell + 1, 2**512 - 1, 2**511, 2**252, 2**252 - 1] + [int.from_bytes(rng.bytes(64), "little") for _ in range(3000)]
    for x in xs:
        s = [(x >> (21 * i)) & 0x1FFFFF for i in range(23)] + [x >> 483]
        worst = 0

        def fold(i):
            nonlocal worst
            for k in range(6):
                s[i - 12 + k] += s[i] * m[k]
                worst = max(worst, abs(s[i - 12 + k]))
            s[i] = 0
        for i in range(23, 17, -1):
            fold(i)
        for i in range(6, 17):
            cy = (s[i] + (1 << 20)) >> 21
            s[i] -= cy << 21
            s[i + 1] += cy
        for i in range(17, 11, -1):
            fold(i)
        for i in range(12):
            cy = (s[i] + (1 << 20)) >> 21
            s[i] -= cy << 21
            s[i + 1] += cy
        for _ in range(2):
            fold(12)
            for i in range(12):
                cy = s[i] >> 21
                s[i] -= cy << 21
                s[i + 1] += cy
        assert s[12] in (0, 1) and all(0 <= v < 1 << 21 for v in s[:12]) and worst < 2**62
        y = sum(v << (21 * i) for i, v in enumerate(s[:13]))
        for _ in range(2):
            if y >= ell:
                y -= ell
        assert y == x % ell


def test_oracle_under_address_and_ub_sanitizers():
    """`make asan` (oracle/c/Makefile): the witness, the trace generator and its checker on a golden case under -fsanitize=address,undefined,
    in a subprocess (the sanitizer runtime has to be loaded first)."""
    import os
    import subprocess
    import sys
    cdir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "c")
    subprocess.check_call(["make", "-s", "-C", cdir, "asan"])
    asan_rt = subprocess.check_output(["gcc", "-print-file-name=libasan.so"]).decode().strip()
    code = r'''
import ctypes as C, json, os, sys
import numpy as np
L = C.CDLL(os.path.join(sys.argv[1], "libtmx_oracle_asan.so"))
c = json.load(open(sys.argv[2]))["skip_10000_10500_n4"]
p, t, r = bytes.fromhex(c["proof"]), bytes.fromhex(c["target"]), bytes.fromhex(c["trusted"])
L.tmxo_elem_count.restype = C.c_size_t; L.tmxo_elem_count.argtypes = [C.c_int, C.c_size_t]
out = np.zeros(L.tmxo_elem_count(0, 4), dtype=np.uint64)
rep = (C.c_uint8 * 64)()
assert L.tmxo_witness(0, p, t, r, C.c_uint32(4), b"mocha-4", C.c_uint32(7), C.c_uint64(100800), out.ctypes.data_as(C.c_void_p), rep) == 0
L.tmxo_trace_elem_count.restype = C.c_size_t; L.tmxo_trace_elem_count.argtypes = [C.c_int, C.c_size_t]
tr = np.zeros(L.tmxo_trace_elem_count(0, 4), dtype=np.uint64)
assert L.tmxo_trace(0, p, t, r, C.c_uint32(4), tr.ctypes.data_as(C.c_void_p)) == 0
L.tmxo_trace_check.restype = C.c_longlong
assert L.tmxo_trace_check(0, p, t, r, C.c_uint32(4), tr.ctypes.data_as(C.c_void_p)) == 0
print("asan ok")
'''
    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cases.json")
    env = dict(os.environ, LD_PRELOAD=asan_rt, ASAN_OPTIONS="detect_leaks=0")
    out = subprocess.run([sys.executable, "-c", code, cdir, golden], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "asan ok" in out.stdout, out.stderr[-2000:]
