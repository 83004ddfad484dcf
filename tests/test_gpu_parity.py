"""GPU parity tests proper: the HIP path, called through the C ABI, against the CPU oracle on the same inputs
(bit-exact: all values are integers / bytes), against the committed goldens, and -- at BASELINE.json's full sizes --
through size-independent properties.  Written to read like the reference's own tests (circuits/skip.rs:157-296,
circuits/step.rs:141-268): same fixtures, same public inputs, same names."""
import hashlib
import json
import os
import struct

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu

FX = os.path.join(GOLDEN, "fixtures", "mocha-4")
MOCHA = b"mocha-4"


@pytest.fixture(scope="module")
def tmx(built_lib):
    import tendermintx_amd
    return tendermintx_amd


def _case_inputs(c):
    return (bytes.fromhex(c["proof"]), bytes.fromhex(c["target"]), bytes.fromhex(c["trusted"]) if c["trusted"] else None)


def _check_vs_oracle(tmx, oracle, kind, n, proofs, targets, trusteds, chain_id, skip_max=100800, ctx=None, threads=8, repeat=1):
    """repeat > 1: the same call again on the same context (its key cache now holds the batch's keys): every run must equal the oracle"""
    P = len(proofs) // 2336
    own = ctx is None
    ctx = ctx or tmx.Context(n, chain_id, skip_max, max_batch=P)
    try:
        runs = [ctx.witness_batch(kind, proofs, targets, trusteds) for _ in range(repeat)]
    finally:
        if own:
            ctx.close()
    want, oreps = oracle.witness_batch(kind, P, proofs, targets, trusteds, n, chain_id, skip_max, n_threads=threads)
    for k, (elems, reps) in enumerate(runs):
        if not np.array_equal(elems, want):
            bad = np.argwhere(elems != want)
            raise AssertionError(f"GPU != oracle (run {k}) at (proof, element) {bad[:10].tolist()} ({len(bad)} differences)")
        assert reps == oreps, f"run {k}"
    return runs[0]


# ------------------------------------------------------------------------------------------------ goldens
def test_golden_cases_bit_exact(tmx, oracle, cases):
    for name, c in sorted(cases.items()):
        proof, target, trusted = _case_inputs(c)
        elems, reps = _check_vs_oracle(tmx, oracle, c["kind"], c["n"], proof, target, trusted, c["chain_id"].encode(), c["skip_max"])
        assert hashlib.sha256(np.ascontiguousarray(elems[0]).tobytes()).hexdigest() == c["elems_sha256"], name
        r = reps[0]
        assert r["header"].hex() == c["header"] and r["all_ok"] == c["all_ok"] and r["fail_mask"] == c["fail_mask"], name
        assert r["first_bad_sig"] == c["first_bad_sig"] and r["gt_target"] == c["gt_target"], name
        path = os.path.join(GOLDEN, f"elems_{name}.npz")
        if os.path.exists(path):
            assert np.array_equal(np.load(path)["elems"], elems[0]), name


# ------------------------------------------------------------------------------------------------ the reference's own tests
def _skip_template(tmx, n, trusted_header_hex, trusted_block, target_block, expect_hex):
    """test_skip_template (skip.rs:219-250) at the value level: target_header out, all constraints satisfied."""
    circ = tmx.SkipCircuit(n, tmx.MOCHA_4_CHAIN_ID_BYTES, tmx.SKIP_MAX, fetcher=tmx.InputDataFetcher(FX))
    try:
        elems, rep = circ.hint(trusted_block, bytes.fromhex(trusted_header_hex), target_block)
        assert rep["all_ok"] and rep["header"].hex().upper() == expect_hex.upper()
        assert len(elems) == circ.ctx.elem_count(0)
        # the first 256 elements are target_header as 32 x 8 big-endian bits (Bytes32Variable)
        bits = elems[:256].reshape(32, 8)
        assert bytes(int("".join(str(int(b)) for b in row), 2) for row in bits) == rep["header"]
    finally:
        circ.close()


def test_skip_circuit_with_input_bytes(tmx, kat):
    """skip.rs:188-217: block 3000 with requested block 3100, N = 4"""
    circ = tmx.SkipCircuit(4, tmx.MOCHA_4_CHAIN_ID_BYTES, fetcher=tmx.InputDataFetcher(FX))
    inp, out = kat["public_io"]["skip"][0]
    assert circ.prove_public(bytes.fromhex(inp)).hex() == out
    circ.close()


def test_skip_small(tmx):
    """skip.rs:252-265"""
    _skip_template(tmx, 4, "A0123D5E4B8B8888A61F931EE2252D83568B97C223E0ECA9795B29B8BD8CBA2D", 10000, 10500,
                   "E2BA1B86926925A69C2FCC32E5178E7E6653D386C956BB975142FA73211A9444")


def test_skip_medium(tmx):
    """skip.rs:267-282"""
    _skip_template(tmx, 32, "A0123D5E4B8B8888A61F931EE2252D83568B97C223E0ECA9795B29B8BD8CBA2D", 10000, 10500,
                   "E2BA1B86926925A69C2FCC32E5178E7E6653D386C956BB975142FA73211A9444")


def test_skip_wrong_trusted_hash_panics(tmx):
    """input/mod.rs:450-455: a wrong trusted header hash fails the sanity assert"""
    circ = tmx.SkipCircuit(4, tmx.MOCHA_4_CHAIN_ID_BYTES, fetcher=tmx.InputDataFetcher(FX))
    with pytest.raises(AssertionError, match="Trusted header hash doesn't pass sanity check"):
        circ.hint(10000, bytes(32), 10500)
    circ.close()


def _step_template(tmx, n, block_height, header_hex, expect_hex):
    """test_step_template (step.rs:200-228)"""
    circ = tmx.StepCircuit(n, tmx.MOCHA_4_CHAIN_ID_BYTES, fetcher=tmx.InputDataFetcher(FX))
    try:
        _, rep = circ.hint(block_height, bytes.fromhex(header_hex))
        assert rep["all_ok"] and rep["header"].hex().upper() == expect_hex.upper()
    finally:
        circ.close()


def test_step_circuit_with_input_bytes(tmx, kat):
    """step.rs:170-198"""
    circ = tmx.StepCircuit(4, tmx.MOCHA_4_CHAIN_ID_BYTES, fetcher=tmx.InputDataFetcher(FX))
    inp, out = kat["public_io"]["step"][0]
    assert circ.prove_public(bytes.fromhex(inp)).hex() == out
    circ.close()


def test_step_small(tmx):
    """step.rs:230-241"""
    _step_template(tmx, 2, 10000, "A0123D5E4B8B8888A61F931EE2252D83568B97C223E0ECA9795B29B8BD8CBA2D",
                   "F2A340CC2AEF6FE163254B326A52334B45793EB11417029F9548418F88B38E26")


def test_step_with_dummy(tmx):
    """step.rs:243-254: validator 2 of block 10501 voted nil -> dummy-signature lane"""
    _step_template(tmx, 4, 10500, "E2BA1B86926925A69C2FCC32E5178E7E6653D386C956BB975142FA73211A9444",
                   "CD3E0F3E47FDAC9ABE1C98CF6BE241BC23A8779E67DF068832F7F43E2DB7B05B")


def test_step_large(tmx):
    """step.rs:256-267: N = 100 (the reference's VALIDATOR_SET_SIZE_MAX, not a power of two)"""
    _step_template(tmx, 100, 10500, "E2BA1B86926925A69C2FCC32E5178E7E6653D386C956BB975142FA73211A9444",
                   "CD3E0F3E47FDAC9ABE1C98CF6BE241BC23A8779E67DF068832F7F43E2DB7B05B")


def test_signed_block_heights_through_the_hint(tmx, oracle, cases):
    """The nine fixture heights that hold only `signed_block.json` (SignedBlockResponse, tendermint_utils.rs:52-55, 97-112), end to end:
    fetcher -> the library's JSON codec -> HIP path, chained the way a light client advances: step 10002 -> 10003 -> 10004 with each
    output header fed in as the next trusted hash, step 11000 -> 11001, skip 11000 -> 11105 / 15000 -> 50000 / 50000 -> 157000 (47- and
    53-of-100 signing: real data with many absent votes).  Headers = the fixtures' block_id.hash; verdicts and rows = the goldens'."""
    sb = json.load(open(os.path.join(GOLDEN, "signed_blocks.json")))
    f = tmx.InputDataFetcher(FX)
    h = bytes.fromhex(sb["10002"]["header_hash"])
    for prev, n in ((10002, 2), (10003, 4)):
        circ = tmx.StepCircuit(n, tmx.MOCHA_4_CHAIN_ID_BYTES, fetcher=f)
        elems, rep = circ.hint(prev, h)
        c = cases[f"step_{prev}_n{n}"]
        assert rep["all_ok"] and rep["header"].hex() == sb[str(prev + 1)]["header_hash"] == c["header"]
        assert hashlib.sha256(np.ascontiguousarray(elems).tobytes()).hexdigest() == c["elems_sha256"]
        h = rep["header"]
        circ.close()
    circ = tmx.StepCircuit(8, tmx.MOCHA_4_CHAIN_ID_BYTES, fetcher=f)
    _, rep = circ.hint(11000, bytes.fromhex(sb["11000"]["header_hash"]))
    assert rep["all_ok"] and rep["header"].hex() == sb["11001"]["header_hash"]
    circ.close()
    for a, b, n in ((11000, 11105, 16), (15000, 50000, 128), (50000, 157000, 128)):
        c = cases[f"skip_{a}_{b}_n{n}"]
        circ = tmx.SkipCircuit(n, tmx.MOCHA_4_CHAIN_ID_BYTES, tmx.SKIP_MAX, fetcher=f)
        proof, target, trusted = f.get_skip_inputs(n, a, bytes.fromhex(sb[str(a)]["header_hash"]), b)
        assert proof.hex() == c["proof"] and target.hex() == c["target"] and trusted.hex() == c["trusted"]
        elems, reps = circ.ctx.witness_batch(0, proof, target, trusted)
        assert reps[0]["header"].hex() == sb[str(b)]["header_hash"] == c["header"]
        assert reps[0]["all_ok"] == c["all_ok"] and reps[0]["fail_mask"] == c["fail_mask"] and reps[0]["gt_target"] == c["gt_target"]
        assert hashlib.sha256(np.ascontiguousarray(elems[0]).tobytes()).hexdigest() == c["elems_sha256"]
        circ.close()


def test_step_wrong_prev_hash_panics(tmx):
    circ = tmx.StepCircuit(4, tmx.MOCHA_4_CHAIN_ID_BYTES, fetcher=tmx.InputDataFetcher(FX))
    with pytest.raises(AssertionError, match="Prev header hash doesn't pass sanity check"):
        circ.hint(10500, bytes(32))
    circ.close()


# ------------------------------------------------------------------------------------------------ EdDSA lanes
def _lane(pk, sig, msg, signed=True, power=1, vlen=38):
    return struct.pack("<32s64s124sHBBQ24x", pk, sig, msg.ljust(124, b"\0"), len(msg), vlen, 3 if signed else 2, power)


def _ed_record(tr):
    return tr["digest"] + tr["h"] + b"".join(tr["pt"]) + struct.pack("<II", int(tr["ok"]), int(tr["decode_ok"])) + bytes(24)


def test_eddsa_lanes_vs_oracle(tmx, oracle, kat):
    rng = np.random.default_rng(11)
    lanes, want = [], []

    def add(pk, sig, msg, signed=True):
        lanes.append(_lane(pk, sig, msg, signed))
        if signed:
            want.append(_ed_record(oracle.eddsa_trace(pk, sig, msg)))
        else:
            dpk, dsig = oracle.dummy()
            want.append(_ed_record(oracle.eddsa_trace(dpk, dsig, bytes(32))))

    for seed, pk, msg, sig in kat["rfc8032"]:
        add(bytes.fromhex(pk), bytes.fromhex(sig), bytes.fromhex(msg))
    ell = 2**252 + 27742317777372353535851937790883648493
    for i in range(40):
        seed = rng.integers(0, 256, 32, dtype=np.uint8).tobytes()
        msg = rng.integers(0, 256, int(rng.integers(0, 125)), dtype=np.uint8).tobytes()
        pk, sig = oracle.pubkey(seed), oracle.sign(seed, msg)
        add(pk, sig, msg)
        mode = i % 8
        b = bytearray(sig)
        if mode == 0:
            b[int(rng.integers(0, 32))] ^= 1 << int(rng.integers(0, 8))      # corrupt R
            add(pk, bytes(b), msg)
        elif mode == 1:
            b[32 + int(rng.integers(0, 31))] ^= 1 << int(rng.integers(0, 8))  # corrupt s
            add(pk, bytes(b), msg)
        elif mode == 2:
            s = int.from_bytes(sig[32:], "little") + ell                      # non-canonical s (same point s*B)
            add(pk, sig[:32] + s.to_bytes(32, "little"), msg)
        elif mode == 3:
            add(pk, sig[:32] + b"\xff" * 32, msg)                             # s = 2^256 - 1 (recoding carry-out path)
        elif mode == 4:
            add((2).to_bytes(32, "little"), sig, msg)                         # undecodable public key
        elif mode == 5:
            add(pk, (2**255 - 19 + 1).to_bytes(32, "little") + sig[32:], msg)  # non-canonical y = p + 1 for R
        elif mode == 6:
            add(pk, sig, msg + b"x" if len(msg) < 124 else msg[:-1])          # wrong message
        else:
            add(pk, sig, msg, signed=False)                                   # unsigned lane -> dummy triple
    add(bytes(32), bytes(64), b"")                                            # y = 0
    add(b"\x01" + bytes(31), b"\x01" + bytes(63), b"")                        # identity point, s = 0
    add(b"\x01" + bytes(30) + b"\x80", bytes(64), b"")                        # x = 0 with sign bit set: rejected
    with tmx.Context(64, max_batch=4) as ctx:
        got = ctx.eddsa_lanes(b"".join(lanes))
    for i, w in enumerate(want):
        assert bytes(got[i]) == w, f"lane {i}"


# ------------------------------------------------------------------------------------------------ synthetic + adversarial
@pytest.mark.parametrize("kind,n,nb,permille", [(0, 4, 4, 1000), (0, 32, 32, 900), (0, 32, 21, 1000), (1, 32, 32, 900),
                                                 (0, 100, 77, 850), (1, 128, 100, 900), (0, 128, 128, 1000)])
def test_synthetic_batches(tmx, oracle, kind, n, nb, permille):
    from tendermintx_amd.synth import Workload
    wl = Workload(kind, n, 5, nb, chain_id=b"celestia", seed=1000 + n + nb, signed_permille=permille, rounds=(0, 3, 0, 2**40 + 7, 1))
    _, reps = _check_vs_oracle(tmx, oracle, kind, n, wl.proofs, wl.targets, wl.trusteds, b"celestia")
    assert all(r["all_ok"] for r in reps)


def test_limb_parallel_field_arithmetic_agrees_with_big_ints(tmx):
    """fe16.hpp (the table chain's arithmetic: sixteen 16-bit limbs across a DPP row): products, point doublings and both conversions
    against Python integers, at the bounds the kernels rely on."""
    P = 2**255 - 19
    rng = np.random.default_rng(5)
    n, d = 64, 7
    words = np.zeros((n, 128), dtype=np.uint32)
    words[:, :64] = rng.integers(0, 196609, size=(n, 64))            # A: what from_limbs10 can produce (3 * 2^16)
    words[:, 64:] = rng.integers(0, 441506, size=(n, 64))            # B: the largest limbs mul() accepts
    words[0, :64] = 196608; words[0, 64:] = 441505                   # all-maximal
    words[1, :64] = 0
    # B's first 40 words double as four signed ten-limb elements for the conversion test
    l10 = np.zeros((n, 40), dtype=np.int64)
    for j in range(10):
        bound = (1 << 26) - 1 if j % 2 == 0 else (1 << 25) - 1
        l10[:, j::10] = rng.integers(-bound, bound + 1, size=(n, 4))
    l10[2] = np.tile([(1 << 26) - 1 if j % 2 == 0 else (1 << 25) - 1 for j in range(10)], 4)
    l10[3] = -l10[2]
    conv_words = words.copy()
    conv_words[:, 64:104] = l10.astype(np.int32).view(np.uint32).reshape(n, 40)
    val16 = lambda l: sum(int(x) << (16 * k) for k, x in enumerate(l))
    off = [(51 * j + 1) // 2 for j in range(10)]
    val10 = lambda l: sum(int(x) << off[j] for j, x in enumerate(l))
    ctx = tmx.Context(4, b"x", max_batch=1)
    try:
        out = ctx.selftest_f16(words, d)
        out2 = ctx.selftest_f16(conv_words, 0)
    finally:
        ctx.close()
    for i in range(n):
        a = [val16(words[i, 16 * r:16 * r + 16]) for r in range(4)]
        b = [val16(words[i, 64 + 16 * r:64 + 16 * r + 16]) for r in range(4)]
        for r in range(4):
            got = out[i, 16 * r:16 * r + 16]
            assert got.max() <= 65536 + 2280 and val16(got) % P == a[r] * b[r] % P, (i, r)
        X, Y, Z = a[0], a[1], a[2]
        for _ in range(d):
            xx, yy, zz, s = X * X, Y * Y, Z * Z, (X + Y) ** 2
            h, e, g = yy + xx, s - xx - yy, yy - xx
            f = 2 * zz - g
            X, Y, Z, T = e * f % P, h * g % P, g * f % P, e * h % P
        got = [val16(out[i, 64 + 16 * r:64 + 16 * r + 16]) % P for r in range(4)]
        assert got == [X, Y, Z, T], i
        for r in range(4):
            t = out[i, 128 + 10 * r:128 + 10 * r + 10].astype(np.int64)
            assert val10(t) % P == a[r] % P and all(0 <= t[j] < (1 << 27) for j in range(10)), (i, r)  # (fe_carry32 follows)
            g16 = out2[i, 192 + 16 * r:192 + 16 * r + 16]
            assert g16.max() < 3 << 16 and val16(g16) % P == val10(l10[i, 10 * r:10 * r + 10]) % P, (i, r)


def test_field_inversions_agree_with_big_int_arithmetic(tmx):
    """k_ed_fin inverts with Bernstein-Yang division steps; the self-test hook returns that and the Fermat chain for caller values.
    Edge values (0, +-1, 2^k +- 1, p - small, non-canonical representatives up to 2^256 - 1, limb-boundary patterns) and 4096 random ones
    against pow(x, p - 2, p)."""
    p = 2**255 - 19
    rng = np.random.default_rng(25519)
    vals = [0, 1, 2, 3, 19, 20, p - 1, p - 2, p - 19, p, p + 1, 2**255 - 1, 2**255, 2**256 - 1, 2**254, 2**252 + 27742317777372353535851937790883648493]
    vals += [2**k for k in range(0, 256, 5)] + [2**k - 1 for k in range(1, 256, 7)] + [p - 2**k for k in range(0, 255, 9)]
    vals += [sum(((1 << 30) - 1) << (30 * i) for i in range(0, 9, 2)) % 2**256, sum(1 << (30 * i) for i in range(9)), (1 << 255) - (1 << 30)]
    vals += [int.from_bytes(rng.bytes(32), "little") for _ in range(4096)]
    with tmx.Context(4, b"celestia", max_batch=1) as ctx:
        got = ctx.selftest_fe_invert(vals)
    for v, (fermat, safegcd) in zip(vals, got):
        want = pow(v % p, p - 2, p)
        assert fermat == want and safegcd == want, hex(v)


@pytest.mark.parametrize("seed", range(32))
def test_random_shapes_and_bit_flips(tmx, oracle, seed):
    """Fuzz: random VALIDATOR_SET_SIZE_MAX (incl. odd sizes), batch size, real set size, signer fraction, rounds, kind, and a few
    random single-bit flips anywhere in the proof / validator / trusted records of every other proof.  Elements and reports must
    equal the oracle's bit for bit whatever the verdict is."""
    from tendermintx_amd.synth import Workload
    rng = np.random.default_rng(20260928 + seed)
    kind = int(rng.integers(0, 2))
    n = int(rng.choice([1, 2, 3, 5, 8, 13, 21, 32, 47, 64, 100]))
    P = int(rng.integers(1, 40))
    nb = int(rng.integers(1, n + 1))
    wl = Workload(kind, n, P, nb, chain_id=b"celestia", seed=int(rng.integers(1, 2**31)), signed_permille=int(rng.integers(500, 1001)),
                  rounds=(0, int(rng.integers(0, 5))))
    proofs, targets = bytearray(wl.proofs), bytearray(wl.targets)
    trusteds = bytearray(wl.trusteds) if kind == 0 else None
    for p in range(0, P, 2):
        for _ in range(int(rng.integers(1, 4))):
            which = int(rng.integers(0, 3 if kind == 0 else 2))
            if which == 0:
                off = p * 2336 + int(rng.integers(0, 2336))
                if (off - p * 2336) in range(56, 64):
                    continue      # nb_a / nb_b > n is a host-side error (TMX_ERR_SET_TOO_LARGE), covered elsewhere
                proofs[off] ^= 1 << int(rng.integers(0, 8))
            elif which == 1:
                targets[p * n * 256 + int(rng.integers(0, n * 256))] ^= 1 << int(rng.integers(0, 8))
            else:
                trusteds[p * n * 48 + int(rng.integers(0, n * 48))] ^= 1 << int(rng.integers(0, 8))
    _check_vs_oracle(tmx, oracle, kind, n, bytes(proofs), bytes(targets), bytes(trusteds) if trusteds is not None else None, b"celestia")


def test_adversarial_mutations(tmx, oracle):
    """Random corruption of every input field: verdicts may flip, parity with the oracle must not."""
    from tendermintx_amd.synth import Workload
    n, P = 32, 24
    rng = np.random.default_rng(5)
    for kind in (0, 1):
        wl = Workload(kind, n, P, 29, chain_id=b"celestia", seed=77 + kind, signed_permille=950, rounds=(0, 0, 4))
        proofs, targets = bytearray(wl.proofs), bytearray(wl.targets)
        trusteds = bytearray(wl.trusteds) if kind == 0 else None
        for p in range(1, P):  # proof 0 stays pristine
            mode = p % 12
            t0 = p * n * 256
            lane = t0 + int(rng.integers(0, 29)) * 256
            if mode == 0:
                targets[lane + int(rng.integers(0, 32))] ^= 0x04                    # pubkey
            elif mode == 1:
                targets[lane + 32 + int(rng.integers(0, 64))] ^= 0x20               # signature
            elif mode == 2:
                targets[lane + 96 + int(rng.integers(0, 100))] ^= 0x01              # message byte
            elif mode == 3:
                targets[lane + 224:lane + 232] = struct.pack("<Q", 2**63 + 5)       # power with bit 63
            elif mode == 4:
                targets[lane + 224:lane + 232] = struct.pack("<Q", 2**63 - 1)       # sums wrap
                targets[lane + 256 + 224:lane + 256 + 232] = struct.pack("<Q", 2**63 - 1)
                targets[lane + 512 + 224:lane + 512 + 232] = struct.pack("<Q", 2**63 - 1)
            elif mode == 5:
                targets[lane + 222] = int(rng.integers(30, 60))                     # validator_byte_length (incl. > 46)
            elif mode == 6:
                targets[lane + 220:lane + 222] = struct.pack("<H", int(rng.integers(0, 300)))  # message length (incl. > 124)
            elif mode == 7:
                targets[lane + 223] ^= 1                                            # signed flag flipped
            elif mode == 8:
                proofs[p * 2336 + 56:p * 2336 + 60] = struct.pack("<I", int(rng.integers(0, 40)))  # nb (incl. > N)
            elif mode == 9:
                hb = p * 2336 + 64 + int(rng.integers(0, 2)) * 1136
                li = int(rng.integers(0, 14))
                proofs[hb + 16 + 80 * li + int(rng.integers(0, 30))] ^= 0x80        # a header field byte
            elif mode == 10:
                hb = p * 2336 + 64
                proofs[hb + int(rng.integers(0, 14))] = int(rng.integers(0, 120))   # a header field length (incl. > 79)
            else:
                proofs[p * 2336 + 16 + int(rng.integers(0, 32))] ^= 0x10            # public trusted / prev hash
                if trusteds is not None:
                    j = p * n * 48 + int(rng.integers(0, 29)) * 48
                    trusteds[j:j + 32] = targets[t0:t0 + 32]                         # duplicate pubkey into the trusted set
        elems, reps = _check_vs_oracle(tmx, oracle, kind, n, bytes(proofs), bytes(targets), bytes(trusteds) if trusteds else None,
                                       b"celestia")
        assert reps[0]["all_ok"] and sum(1 for r in reps if not r["all_ok"]) >= P // 2


@pytest.mark.parametrize("kind", [0, 1])
def test_header_field_length_edges(tmx, oracle, kind):
    """Every field-length byte of both headers set to 0, 1, 2, 33, 79, 80 and 255 (one proof per combination): empty and oversized
    fields, among them the EMPTY height field whose re-encoded leaf `00 08 varint9` is then hashed over one byte only -- the kernel
    hashed the 08 tag along (found by test_fuzz_extended.py, seed 1452; the oracle had it right)."""
    from tendermintx_amd.synth import Workload
    n, values = 8, (0, 1, 2, 33, 79, 80, 255)
    P = 2 * 14 * len(values)
    wl = Workload(kind, n, P, 7, chain_id=b"celestia", seed=4242 + kind, signed_permille=1000, rounds=(0, 2))
    proofs = bytearray(wl.proofs)
    p = 0
    for hdr in range(2):
        for field in range(14):
            for v in values:
                proofs[p * 2336 + 64 + hdr * 1136 + field] = v
                p += 1
    _check_vs_oracle(tmx, oracle, kind, n, bytes(proofs), wl.targets, wl.trusteds if kind == 0 else None, b"celestia")


def test_threshold_edges(tmx, oracle):
    """exactly 2/3 is not enough (strict >, voting.rs:108); one more unit is"""
    from tendermintx_amd.synth import Workload
    n = 4
    wl = Workload(1, n, 1, 3, chain_id=b"celestia", seed=9, signed_permille=1000, rounds=(0,))
    for powers, signed, expect in [((10, 10, 10), (1, 1, 0), False), ((10, 10, 9), (1, 1, 0), True), ((1, 1, 1), (1, 1, 1), True)]:
        t = bytearray(wl.targets)
        for i in range(3):
            t[i * 256 + 224:i * 256 + 232] = struct.pack("<Q", powers[i])
            if not signed[i]:
                t[i * 256 + 223] &= 0xFE
        _, reps = _check_vs_oracle(tmx, oracle, 1, n, wl.proofs, bytes(t), None, b"celestia")
        assert reps[0]["gt_target"] == expect


# ------------------------------------------------------------------------------------------------ batch / device paths, properties
def test_batch_equals_individual_and_is_idempotent(tmx, oracle):
    from tendermintx_amd.synth import Workload
    n, P = 32, 37
    wl = Workload(0, n, P, 30, chain_id=b"celestia", seed=4242, signed_permille=900)
    with tmx.Context(n, b"celestia", max_batch=P) as ctx:
        a, ra = ctx.witness_batch(0, wl.proofs, wl.targets, wl.trusteds)
        b, rb = ctx.witness_batch(0, wl.proofs, wl.targets, wl.trusteds)
        assert np.array_equal(a, b) and ra == rb                              # idempotent
        for p in (0, 17, 36):
            e, r = ctx.witness_batch(0, wl.proofs[p * 2336:(p + 1) * 2336], wl.targets[p * n * 256:(p + 1) * n * 256],
                                     wl.trusteds[p * n * 48:(p + 1) * n * 48])
            assert np.array_equal(e[0], a[p]) and r[0] == ra[p]
        # permuting the proofs of a batch permutes the rows
        perm = np.random.default_rng(0).permutation(P)
        pp = b"".join(wl.proofs[p * 2336:(p + 1) * 2336] for p in perm)
        tt = b"".join(wl.targets[p * n * 256:(p + 1) * n * 256] for p in perm)
        rr = b"".join(wl.trusteds[p * n * 48:(p + 1) * n * 48] for p in perm)
        c, _ = ctx.witness_batch(0, pp, tt, rr)
        assert np.array_equal(c, a[perm])
        with pytest.raises(tmx.TmxError):
            ctx.witness_batch(0, wl.proofs + wl.proofs[:2336], wl.targets + wl.targets[:n * 256], wl.trusteds + wl.trusteds[:n * 48])


def test_device_resident_path_with_torch(tmx, oracle):
    """tmx_witness_batch_device on PyTorch-owned HBM buffers and PyTorch's current stream (one HIP runtime per process)"""
    import torch
    from tendermintx_amd.synth import Workload
    n, P = 128, 12
    wl = Workload(0, n, P, 128, chain_id=b"celestia", seed=31337, signed_permille=900)
    dev = torch.device("cuda", 0)
    d = [torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev) for b in (wl.proofs, wl.targets, wl.trusteds)]
    with tmx.Context(n, b"celestia", max_batch=P) as ctx:
        stride, count = ctx.elem_stride(0), ctx.elem_count(0)
        out = torch.full((P, stride), -1, dtype=torch.int64, device=dev)
        rep = torch.zeros(P * 64, dtype=torch.uint8, device=dev)
        s = torch.cuda.Stream(dev)
        with torch.cuda.stream(s):
            ctx.witness_batch_device(0, P, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), out.data_ptr(), rep.data_ptr(), s.cuda_stream)
        s.synchronize()
        ms = ctx.kernel_ms_mean(1)
        # (1536 lanes = the largest small launch: k_tiny is attributed ONCE, to the EdDSA slot -- the four figures are disjoint intervals of the launch)
        assert all(v >= 0 for v in ms.values()) and ms["k_eddsa"] > 0 and ms["k_verdict"] > 0 and ms["k_proof"] == 0
    got = out[:, :count].cpu().numpy().view(np.uint64)
    want, oreps = oracle.witness_batch(0, P, wl.proofs, wl.targets, wl.trusteds, n, b"celestia", 100800, n_threads=8)
    assert np.array_equal(got, want)
    assert int(out[:, count:].abs().sum().item()) == 0                       # row padding is zero-filled
    assert bytes(rep.cpu().numpy()[:32]) == oreps[0]["header"]


def test_full_size_n128_batch256(tmx, oracle):
    """BASELINE configs[2]/[3]: N = 128, 256 proofs: every row bit-exact vs the (multi-threaded) oracle, all proofs verify,
    checksum-of-checksums stable across two runs."""
    from tendermintx_amd.synth import Workload
    n, P = 128, 256
    wl = Workload(0, n, P, 128, chain_id=b"celestia", seed=0x544D58, signed_permille=1000)
    with tmx.Context(n, b"celestia", max_batch=P) as ctx:
        elems, reps = _check_vs_oracle(tmx, oracle, 0, n, wl.proofs, wl.targets, wl.trusteds, b"celestia", ctx=ctx, threads=os.cpu_count() or 8)
        again, _ = ctx.witness_batch(0, wl.proofs, wl.targets, wl.trusteds)
    assert all(r["all_ok"] for r in reps)
    h1 = hashlib.sha256(b"".join(hashlib.sha256(np.ascontiguousarray(r).tobytes()).digest() for r in elems)).hexdigest()
    h2 = hashlib.sha256(b"".join(hashlib.sha256(np.ascontiguousarray(r).tobytes()).digest() for r in again)).hexdigest()
    assert h1 == h2


def test_stress_n512(tmx, oracle):
    """BASELINE configs[4]: VALIDATOR_SET_SIZE_MAX = 512, full and partially filled sets"""
    from tendermintx_amd.synth import Workload
    for nb in (512, 300):
        wl = Workload(0, 512, 3, nb, chain_id=b"celestia", seed=512 + nb, signed_permille=900)
        _, reps = _check_vs_oracle(tmx, oracle, 0, 512, wl.proofs, wl.targets, wl.trusteds, b"celestia")
        assert all(r["all_ok"] for r in reps)


def test_set_too_large_and_capacity_errors(tmx):
    from tendermintx_amd.synth import Workload
    wl = Workload(0, 4, 1, 4, seed=1)
    bad = bytearray(wl.proofs)
    bad[56:60] = struct.pack("<I", 5)  # nb_a > N through the host entry point -> TMX_ERR_SET_TOO_LARGE (mod.rs:439-444)
    with tmx.Context(4, b"celestia") as ctx:
        with pytest.raises(tmx.TmxError) as e:
            ctx.witness_batch(0, bytes(bad), wl.targets, wl.trusteds)
        assert e.value.status == -2


def test_section_selection_and_u32_transfer_format(tmx, oracle):
    """tmx_witness_batch_opts: hint-only / derived-only / full rows, as u64 and narrowed to u32, are the same values as the full row
    (every element of this witness is < 2^32); the device entry point with a section selection leaves the other section unwritten."""
    import torch
    from tendermintx_amd import _lib
    from tendermintx_amd.synth import Workload
    for kind, n, P in ((0, 32, 9), (1, 5, 3), (0, 128, 12)):
        wl = Workload(kind, n, P, max(1, n - 3), chain_id=b"celestia", seed=31 + n, signed_permille=900)
        want, oreps = oracle.witness_batch(kind, P, wl.proofs, wl.targets, wl.trusteds, n, b"celestia", 100800, n_threads=8)
        assert int(want.max()) < 2**32
        with tmx.Context(n, b"celestia", max_batch=P) as ctx:
            hint = ctx.hint_elem_count(kind)
            for sections, lo, hi in ((_lib.SEC_HINT, 0, hint), (_lib.SEC_DERIVED, hint, want.shape[1]), (_lib.SEC_ALL, 0, want.shape[1])):
                for fmt in ("u64", "u32"):
                    got, reps = ctx.witness_batch_opts(kind, wl.proofs, wl.targets, wl.trusteds, sections, fmt)
                    assert got.dtype == (np.uint32 if fmt == "u32" else np.uint64) and got.shape == (P, hi - lo)
                    assert np.array_equal(got.astype(np.uint64), want[:, lo:hi]), (kind, n, sections, fmt)
                    assert reps == oreps
            got, _ = ctx.witness_batch_hint(kind, wl.proofs, wl.targets, wl.trusteds)
            assert got.dtype == np.uint32 and np.array_equal(got.astype(np.uint64), want[:, :hint])
            # device entry point, hint only: H is written, D keeps the caller's fill (apart from the few seam spans)
            dev = torch.device("cuda", 0)
            d = [torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev) if b else None for b in (wl.proofs, wl.targets, wl.trusteds)]
            out = torch.full((P, ctx.elem_stride(kind)), -1, dtype=torch.int64, device=dev)
            rep = torch.zeros(P * 64, dtype=torch.uint8, device=dev)
            ctx.witness_batch_device_sections(kind, P, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr() if d[2] is not None else None,
                                              out.data_ptr(), rep.data_ptr(), _lib.SEC_HINT, 0)
            torch.cuda.synchronize(dev)
            o = out.cpu().numpy()
            assert np.array_equal(o[:, :hint].view(np.uint64), want[:, :hint])
            assert (o[:, hint:] == -1).mean() > 0.9
            with pytest.raises(tmx.TmxError):
                ctx.witness_batch_opts(kind, wl.proofs, wl.targets, wl.trusteds, 4, "u64")


@pytest.mark.parametrize("kind, n, P", [(0, 128, 240), (0, 128, 96), (1, 64, 200)])
def test_section_selection_on_the_large_path(tmx, oracle, kind, n, P):
    """The device entry point with a section selection on both forms of the one-launch tail (k_verdict_tail_wide: from 28 672 lanes on the
    high-priority stream beside D.1a, below on the caller's stream behind it): the selected section equals the full row's, the other one keeps the caller's fill except for the seam spans (whole
    spans of 256 elements that straddle a section boundary or the row end: always written, with the right values)."""
    import torch
    from tendermintx_amd import _lib
    from tendermintx_amd.synth import Workload
    wl = Workload(kind, n, P, max(1, n - 5), chain_id=b"celestia", seed=4100 + n, signed_permille=930, n_sets=3)
    want, oreps = oracle.witness_batch(kind, P, wl.proofs, wl.targets, wl.trusteds, n, b"celestia", 100800, n_threads=16)
    dev = torch.device("cuda", 0)
    d = [torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev) if b else None for b in (wl.proofs, wl.targets, wl.trusteds)]
    with tmx.Context(n, b"celestia", max_batch=P) as ctx:
        hint, count, stride = ctx.hint_elem_count(kind), ctx.elem_count(kind), ctx.elem_stride(kind)
        for sections, lo, hi in ((_lib.SEC_HINT, 0, hint), (_lib.SEC_DERIVED, hint, count), (_lib.SEC_ALL, 0, count)):
            out = torch.full((P, stride), -1, dtype=torch.int64, device=dev)
            rep = torch.zeros(P * 64, dtype=torch.uint8, device=dev)
            for _ in range(2):  # (cold, then warm: both schedules)
                ctx.witness_batch_device_sections(kind, P, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr() if d[2] is not None else None,
                                                  out.data_ptr(), rep.data_ptr(), sections, 0)
            torch.cuda.synchronize(dev)
            o = out.cpu().numpy()
            assert np.array_equal(o[:, lo:hi].view(np.uint64), want[:, lo:hi]), (kind, n, sections)
            other = np.ones(count, dtype=bool)
            other[lo:hi] = False
            written = (o[:, :count] != -1) & other[None, :]
            # whatever was written outside the selection has the row's value, and it is a small share (seam spans only)
            assert np.array_equal(o[:, :count].view(np.uint64)[written], want[written]), (kind, n, sections)
            assert sections == _lib.SEC_ALL or written.mean() < 0.05, (kind, n, sections, float(written.mean()))
            reps = np.frombuffer(rep.cpu().numpy().tobytes(), dtype=np.uint32).reshape(P, 16)
            assert [int(r[8]) for r in reps] == [int(bool(x["all_ok"])) for x in oreps]


def test_device_path_flags_nb_above_n(tmx, oracle):
    """The device entry points cannot refuse nb > N before enqueueing (the records are in HBM): tmx_report.precond carries the host
    assert of input/mod.rs:439-444 / 338-342 instead, and the values follow the circuit (every lane enabled)."""
    import torch
    from tendermintx_amd.synth import Workload
    n, P = 8, 3
    wl = Workload(0, n, P, 8, chain_id=b"celestia", seed=99, signed_permille=1000)
    proofs = bytearray(wl.proofs)
    proofs[1 * 2336 + 56:1 * 2336 + 60] = struct.pack("<I", n + 1)       # proof 1: nb_a > N
    proofs[2 * 2336 + 60:2 * 2336 + 64] = struct.pack("<I", 4 * n)        # proof 2: nb_b > N
    dev = torch.device("cuda", 0)
    d = [torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev) for b in (bytes(proofs), wl.targets, wl.trusteds)]
    with tmx.Context(n, b"celestia", max_batch=P) as ctx:
        out = torch.zeros((P, ctx.elem_stride(0)), dtype=torch.int64, device=dev)
        rep = torch.zeros(P * 64, dtype=torch.uint8, device=dev)
        ctx.witness_batch_device(0, P, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), out.data_ptr(), rep.data_ptr(), 0)
        torch.cuda.synchronize(dev)
        count = ctx.elem_count(0)
    reps = np.frombuffer(rep.cpu().numpy().tobytes(), dtype=np.uint32).reshape(P, 16)
    assert [int(r[14]) for r in reps] == [0, 1, 2]
    want, oreps = oracle.witness_batch(0, P, bytes(proofs), wl.targets, wl.trusteds, n, b"celestia", 100800)
    assert np.array_equal(out[:, :count].cpu().numpy().view(np.uint64), want)
    assert [r["precond"] for r in oreps] == [0, 1, 2] and all(r["all_ok"] for r in oreps)   # in-circuit: all enabled, same verdict


def test_sharded_entry_points_through_rccl(tmx, oracle):
    """BASELINE configs[3] / [4] through the C entry points (tmx_witness_batch_sharded_device, tmx_witness_validator_sharded_device) on the
    ranks available here: ONE, with a real RCCL communicator (tmx_comm_unique_id + tmx_comm_create with an id), so the grouped exchange is
    RCCL's; two ranks: tests/test_multi_gpu.py (needs two GPUs), the partition / reassembly rule on CPU: tests/test_sharding_gloo.py."""
    import torch
    from tendermintx_amd import sharding
    from tendermintx_amd.synth import Workload
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    # ---- validator-sharded: one proof at N = 512 (configs[4]), then three proofs at N = 32
    for n, P, nb in ((512, 1, 400), (32, 3, 29)):
        wl = Workload(0, n, P, nb, chain_id=b"celestia", seed=2024 + n, signed_permille=900)
        d = [torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev) for b in (wl.proofs, wl.targets, wl.trusteds)]
        with tmx.Context(n, b"celestia", max_batch=P) as ctx:
            assert ctx.comm_info() == (0, 1)
            ctx.comm_create(sharding.unique_id(), 0, 1)
            elems, rep = sharding.validator_sharded_skip(ctx, 0, d[0], d[1], d[2], n_proofs=P)
            torch.cuda.synchronize(dev)
            got = elems.cpu().numpy().view(np.uint64).reshape(P, -1)[:, :ctx.elem_count(0)]
            ctx.comm_destroy()
            assert ctx.comm_info() == (0, 1)
        want, oreps = oracle.witness_batch(0, P, wl.proofs, wl.targets, wl.trusteds, n, b"celestia", 100800)
        assert np.array_equal(got, want) and all(r["all_ok"] for r in oreps)
        assert bytes(rep.cpu().numpy()[:32]) == oreps[0]["header"]
    # ---- proof-sharded with the row exchange
    n, P = 16, 7
    wl = Workload(0, n, P, 13, chain_id=b"celestia", seed=77, signed_permille=900)
    d = [torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev) for b in (wl.proofs, wl.targets, wl.trusteds)]
    with tmx.Context(n, b"celestia", max_batch=P) as ctx:
        ctx.comm_create(sharding.unique_id(), 0, 1)
        out = torch.zeros((P, ctx.elem_stride(0)), dtype=torch.int64, device=dev)
        rep = torch.zeros(P * 64, dtype=torch.uint8, device=dev)
        sharding.proof_sharded_batch(ctx, 0, P, d[0], d[1], d[2], out, rep, gather=True)
        torch.cuda.synchronize(dev)
        want, _ = oracle.witness_batch(0, P, wl.proofs, wl.targets, wl.trusteds, n, b"celestia", 100800)
        assert np.array_equal(out[:, :ctx.elem_count(0)].cpu().numpy().view(np.uint64), want)
        with pytest.raises(tmx.TmxError):
            ctx.comm_create(None, 0, 2)          # a world of two needs the id


KNOBS = [
    {"TMX_DEDUP": "0"}, {"TMX_DEDUP": "2"}, {"TMX_KEY_CACHE": "0"}, {"TMX_KEY_CACHE": "0", "TMX_DEDUP": "2"}, {"TMX_KEY_CACHE_KEYS": "40"},
    {"TMX_WALK_PARTS": "1"}, {"TMX_WALK_PARTS": "1", "TMX_KEY_CACHE": "0"}, {"TMX_EXT_EVENTS": "0"}, {"TMX_LEAVES": "1"},
    {"TMX_LEAVES": "1", "TMX_SER_SPLIT": "0"}, {"TMX_SER_SPLIT": "0"}, {"TMX_SER_SPLIT": "0", "TMX_KEY_CACHE": "0"}, {"TMX_SCHEDULE": "warm"},
    {"TMX_SCHEDULE": "cold"}, {"TMX_SCHEDULE": "warm", "TMX_KEY_CACHE_KEYS": "40"}, {"TMX_SCHEDULE": "warm", "TMX_DEDUP": "0"},
    {"TMX_SCHEDULE": "cold", "TMX_LEAVES": "1", "TMX_WALK_PARTS": "1"},
    # round 4: the small path (two launches for <= 1536 lanes, TMX_TINY_MAX) and k_proof as role workgroups, off / forced / combined with the others
    {"TMX_TINY": "0"}, {"TMX_PROOF_ROLES": "0"}, {"TMX_TINY": "0", "TMX_PROOF_ROLES": "0"}, {"TMX_TINY": "1", "TMX_SCHEDULE": "cold"},
    {"TMX_TINY": "1", "TMX_KEY_CACHE": "0"}, {"TMX_TINY": "1", "TMX_EXT_EVENTS": "0"}, {"TMX_TINY": "1", "TMX_KEY_CACHE_KEYS": "40"},
    {"TMX_PHASE1_MAX": "0"}, {"TMX_PHASE1_MAX": "1000000", "TMX_TINY": "0"},
    # the warm schedule opening with the hash role (dedup + key pipeline on side2): off, forced onto a launch with new keys, with the
    # hash role as a kernel of its own, with record packets instead of completion signals
    {"TMX_HASH_FIRST": "0"}, {"TMX_HASH_FIRST": "1", "TMX_SCHEDULE": "warm"}, {"TMX_HASH_FIRST": "1", "TMX_PHASE1_MAX": "0"},
    {"TMX_HASH_FIRST": "1", "TMX_EXT_EVENTS": "0", "TMX_TINY": "0"},
    # round 5: the warm walk split by residency (resident lanes at once, new-key lanes behind the table build on the side stream): off; forced
    # onto launches with new keys (the mixed case is what the split is for: the batch below brings 100 new keys under a warm hint)
    {"TMX_WALK_SPLIT": "0", "TMX_SCHEDULE": "warm"}, {"TMX_SCHEDULE": "warm", "TMX_TINY": "0"}, {"TMX_SCHEDULE": "warm", "TMX_HASH_FIRST": "0", "TMX_KEY_CACHE_KEYS": "60"},
    {"TMX_SCHEDULE": "warm", "TMX_PHASE1_MAX": "0"}, {"TMX_SCHEDULE": "warm", "TMX_PHASE1_MAX": "0", "TMX_EXT_EVENTS": "0", "TMX_TINY": "0"},
    # round 5: lanes that did not sign take the context's precomputed record and the EdDSA kernels run over the dense list of the others (every
    # launch of > 2048 lanes whose chain the dedup opens: the cold calls of this test, the warm ones with TMX_HASH_FIRST=0): off; on in both schedules
    {"TMX_COMPACT": "0"}, {"TMX_COMPACT": "0", "TMX_HASH_FIRST": "0"}, {"TMX_HASH_FIRST": "0", "TMX_PHASE1_MAX": "0"}, {"TMX_HASH_FIRST": "0", "TMX_SCHEDULE": "cold"},
    # the cache epilogue in front of / behind the input sections of the low-priority stream; no validator-set cache
    {"TMX_EPI_LATE": "0", "TMX_SCHEDULE": "warm"}, {"TMX_SET_CACHE": "0"}, {"TMX_SET_CACHE": "0", "TMX_SCHEDULE": "warm", "TMX_TINY": "0"},
    # per-lane sections of the row span by span (k_serialize) instead of lane by lane (k_serialize_lanes, the default above 8 proofs)
    {"TMX_SER_LANES": "0"}, {"TMX_SER_LANES": "0", "TMX_SER_SPLIT": "0"}, {"TMX_SER_SPLIT": "0"},
    # round 6: the throughput regime's settings forced on a small batch (row-writer wave priority, input sections in front of the new-key
    # pipeline, the writers' workgroup cap), the verdict's tail on / off the caller's stream, the fused rows at this size (no carrier: the sweeper alone)
    {"TMX_WRITER_PRIO": "3"}, {"TMX_INPUTS_FIRST": "1"}, {"TMX_INPUTS_FIRST": "1", "TMX_WRITER_PRIO": "3", "TMX_FEW_WGS": "8192"},
    {"TMX_INPUTS_FIRST": "1", "TMX_SCHEDULE": "warm", "TMX_TINY": "0"}, {"TMX_TAIL_ASIDE_MIN": "0"}, {"TMX_TAIL_ASIDE_MIN": "1000000"},
    {"TMX_FUSED_ROWS": "4:2", "TMX_SCHEDULE": "warm", "TMX_PHASE1_MAX": "0"},
    # the capped row-writer launches grid-striding over (proof, block) as in round 5 instead of proof-major with the LUT words loaded once
    {"TMX_SER_ROWS": "0"}, {"TMX_SER_ROWS": "0", "TMX_FEW_WGS": "100"}, {"TMX_FEW_WGS": "100"}, {"TMX_FEW_WGS": "700", "TMX_INPUTS_FIRST": "1"},
    # verdict + D.5 + the seam spans as one launch of independent workgroups (k_verdict_tail_wide) / as three launches, alone and with D.1a
    # early / the tail never on the caller's stream
    {"TMX_TAIL_WIDE": "1"}, {"TMX_TAIL_WIDE": "0"}, {"TMX_TAIL_WIDE": "1", "TMX_P1_EARLY": "2"}, {"TMX_TAIL_WIDE": "0", "TMX_TAIL_ASIDE_MIN": "0", "TMX_TINY": "0"},
    {"TMX_TAIL_WIDE": "1", "TMX_TAIL_ASIDE_MIN": "0", "TMX_TINY": "0"},
    # the uncapped serializer calls as one launch from the first to the last selected section / one per run of adjacent sections
    {"TMX_TINY_MAX": "2048"}, {"TMX_TINY_MAX": "256"}, {"TMX_JOIN1": "0"}, {"TMX_JOIN1": "0", "TMX_TAIL_ASIDE_MIN": "0", "TMX_TINY": "0"},
    {"TMX_SER_ONE_LAUNCH": "100000", "TMX_TINY": "0"}, {"TMX_SER_ONE_LAUNCH": "0", "TMX_TINY": "0"}, {"TMX_SER_ONE_LAUNCH": "100000", "TMX_TAIL_ASIDE_MIN": "0"}]


@pytest.mark.parametrize("knobs", KNOBS, ids=lambda k: ",".join(f"{a}={b}" for a, b in k.items()))
def test_schedule_knobs_give_the_same_bits(tmx, oracle, monkeypatch, knobs):
    """Every knob the product keeps (api.cpp `Knobs`) changes a schedule -- table use, the key cache, launch splitting, the order of the
    serializer launches -- never a value: a repeated-validator-set batch at N = 128 (new-key tables built in parts, dummy lanes, a failing
    signature) is bit-exact vs the oracle under each of them, cold and then warm on the same context, followed by a single proof."""
    from tendermintx_amd.synth import Workload
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)      # read at context creation
    n, P = 128, 20                    # 2560 lanes: above the 2048 at which the table build is cut into parts
    wl = Workload(0, n, P, 100, chain_id=b"celestia", seed=4242, signed_permille=850)
    targets = bytearray(wl.targets)
    lane = next(l for l in range(n) if targets[l * 256 + 223] & 1)   # a lane of proof 0 that did sign
    targets[lane * 256 + 40] ^= 0x10  # corrupt its signature (R): the equation must fail on exactly that lane
    with tmx.Context(n, b"celestia", max_batch=P) as ctx:
        for _ in range(2):
            _, reps = _check_vs_oracle(tmx, oracle, 0, n, wl.proofs, bytes(targets), wl.trusteds, b"celestia", ctx=ctx)
            assert reps[0]["first_bad_sig"] == lane and not reps[0]["all_ok"] and all(r["first_bad_sig"] == -1 for r in reps[1:])
        _check_vs_oracle(tmx, oracle, 0, n, wl.proofs[2336:2 * 2336], bytes(targets[n * 256:2 * n * 256]), wl.trusteds[n * 48:2 * n * 48], b"celestia", ctx=ctx)
        # small launches: the proof with the failing lane alone (the finish's long way: R decoded, R + h*A formed), four proofs, and twelve
        # (1536 lanes: the small path too; with TMX_TINY=0 the classic graph with k_proof as role workgroups)
        for p0, p1 in ((0, 1), (0, 4), (3, 15)):
            _, reps = _check_vs_oracle(tmx, oracle, 0, n, wl.proofs[p0 * 2336:p1 * 2336], bytes(targets[p0 * n * 256:p1 * n * 256]),
                                       wl.trusteds[p0 * n * 48:p1 * n * 48], b"celestia", ctx=ctx)
            assert [r["first_bad_sig"] for r in reps] == ([lane] if p0 == 0 else [-1]) + [-1] * (p1 - p0 - 1)


@pytest.mark.parametrize("fused", ["0:0", "4:0", "0:3", "8:4", "40:40"])
def test_fused_rows_give_the_same_bits(tmx, oracle, monkeypatch, fused):
    """TMX_FUSED_ROWS=<b>:<w> (layout.h FusedRows): the input-only row spans of a warm batch above 16384 lanes as work items that
    the waves of s*B (b per table addition) and of the resident walk (w) claim between their additions, a capped claiming launch sweeping up
    the rest.  Whoever writes a span writes the same bytes: 140 proofs x 128 lanes (one failing signature), cold, warm, warm again with three
    proofs over new validator sets (new keys: the new-key walk does not carry) and warm once more -- all rows and reports vs the oracle.
    40:40 = the carriers claim more than there is.  (Measured slower than the launches of their own at 256 / 512 / 768 proofs: off by default,
    docs/experiments.md round 6.)"""
    from tendermintx_amd.synth import Workload
    monkeypatch.setenv("TMX_FUSED_ROWS", fused)
    n, P = 128, 140
    wl = Workload(0, n, P, 100, chain_id=b"celestia", seed=6161, signed_permille=880, n_sets=3)
    targets = bytearray(wl.targets)
    lane = next(l for l in range(n) if targets[l * 256 + 223] & 1)
    targets[lane * 256 + 40] ^= 0x10
    fresh = Workload(0, n, 3, 90, chain_id=b"celestia", seed=6262, signed_permille=950, n_sets=3)
    mix = (fresh.proofs + wl.proofs[3 * 2336:], fresh.targets + bytes(targets[3 * n * 256:]), fresh.trusteds + wl.trusteds[3 * n * 48:])
    with tmx.Context(n, b"celestia", max_batch=P) as ctx:
        for _ in range(2):
            _, reps = _check_vs_oracle(tmx, oracle, 0, n, wl.proofs, bytes(targets), wl.trusteds, b"celestia", ctx=ctx)
            assert reps[0]["first_bad_sig"] == lane and all(r["first_bad_sig"] == -1 for r in reps[1:])
        _check_vs_oracle(tmx, oracle, 0, n, *mix, b"celestia", ctx=ctx)
        _check_vs_oracle(tmx, oracle, 0, n, wl.proofs, bytes(targets), wl.trusteds, b"celestia", ctx=ctx, repeat=2)


@pytest.mark.parametrize("permille", [0, 1000, 500])
def test_compacted_launch_edges(tmx, oracle, monkeypatch, permille):
    """The dense list of lanes that signed (k_ed_dedup, kernels.h EdQuad.compact) at its edges: NOBODY signed in the whole batch (an empty
    list: every lane is the precomputed dummy record, every proof fails its threshold), EVERYBODY signed (the list is every lane), half of
    them -- 24 proofs x 128 lanes, cold and warm on one context with the dedup opening the chain (TMX_HASH_FIRST=0), bit-exact vs the oracle."""
    from tendermintx_amd.synth import Workload
    monkeypatch.setenv("TMX_HASH_FIRST", "0")
    n, P = 128, 24
    wl = Workload(0, n, P, 128 if permille == 1000 else 90, chain_id=b"celestia", seed=515 + permille, signed_permille=permille)
    with tmx.Context(n, b"celestia", max_batch=P) as ctx:
        _, reps = _check_vs_oracle(tmx, oracle, 0, n, wl.proofs, wl.targets, wl.trusteds, b"celestia", ctx=ctx, repeat=2)
        assert all(r["first_bad_sig"] == -1 for r in reps) and (permille != 0 or not any(r["all_ok"] for r in reps))


def test_many_distinct_keys_take_the_throughput_forms(tmx, oracle):
    """More than 8192 distinct keys in one launch: k_ed_keys decodes one key per thread and h*A runs in the quad form (the
    limb-parallel forms are for the few-keys / few-lanes chains).  Random 32-byte strings as keys (about half decode), random signatures:
    every lane's record against the oracle's trace of the same triple."""
    rng = np.random.default_rng(77)
    n_lanes = 8192 + 320
    pks = rng.integers(0, 256, (n_lanes, 32), dtype=np.uint8)
    sigs = rng.integers(0, 256, (n_lanes, 64), dtype=np.uint8)
    sigs[:, 63] &= 0x0f                                               # s < 2^252: canonical, so the lanes are not all rejected early
    msgs = rng.integers(0, 256, (n_lanes, 60), dtype=np.uint8)
    lanes = b"".join(_lane(pks[i].tobytes(), sigs[i].tobytes(), msgs[i].tobytes()) for i in range(n_lanes))
    with tmx.Context(128, b"celestia", max_batch=(n_lanes + 127) // 128) as ctx:
        got = ctx.eddsa_lanes(lanes)
        uniq, tables = ctx.last_dedup()
        assert uniq == n_lanes and not tables
        # the same lanes again: the keys the cache had room for now walk their tables, the rest keep the table-free form -- same bits
        again = ctx.eddsa_lanes(lanes)
        uniq2, tables2 = ctx.last_dedup()
        st = ctx.key_cache_stats()
        assert uniq2 == n_lanes and tables2 and 0 < st["last_hit_lanes"] < n_lanes
        assert np.array_equal(got, again)
    for i in list(range(0, n_lanes, 37)) + [n_lanes - 1]:
        want = _ed_record(oracle.eddsa_trace(pks[i].tobytes(), sigs[i].tobytes(), msgs[i].tobytes()))
        assert bytes(got[i]) == want, f"lane {i}"


def test_key_dedup_paths(tmx, oracle):
    """The EdDSA stage decodes every distinct public key once and walks per-key fixed-base tables instead of doubling when the key is
    resident in the key cache or repeats inside the launch (>= 8 lanes per key): every schedule must give the same bits.  Batches: one
    validator set repeated (fresh tables), every proof with its own validator set (table-free form; tables built for later), and a mix
    of resident and new keys."""
    from tendermintx_amd.synth import Workload
    n = 16

    def cat(wls):
        return b"".join(w.proofs for w in wls), b"".join(w.targets for w in wls), b"".join(w.trusteds for w in wls)

    same = Workload(0, n, 160, 16, chain_id=b"celestia", seed=1, signed_permille=1000)  # 2560 lanes (launches of <= 2048 lanes take the small path, which never waits for fresh tables)
    distinct = [Workload(0, n, 1, 16, chain_id=b"celestia", seed=100 + i, signed_permille=1000) for i in range(24)]
    mixed = [Workload(0, n, 12, 16, chain_id=b"celestia", seed=7, signed_permille=900)] + distinct[:6]
    with tmx.Context(n, b"celestia", max_batch=160) as ctx:
        _, reps = _check_vs_oracle(tmx, oracle, 0, n, same.proofs, same.targets, same.trusteds, b"celestia", ctx=ctx)
        assert all(r["all_ok"] for r in reps)
        uniq, tables = ctx.last_dedup()
        assert uniq == 16 and tables                      # 2560 lanes, 16 keys
        p, t, r = cat(distinct)
        _, reps = _check_vs_oracle(tmx, oracle, 0, n, p, t, r, b"celestia", ctx=ctx)
        assert all(x["all_ok"] for x in reps)
        uniq, tables = ctx.last_dedup()
        assert uniq == 24 * 16 and not tables              # every key new and seen once: table-free h*A
        p, t, r = cat(mixed)
        _, reps = _check_vs_oracle(tmx, oracle, 0, n, p, t, r, b"celestia", ctx=ctx)
        uniq, tables = ctx.last_dedup()
        assert uniq >= 16 + 6 * 16                         # (+ the dummy key of unsigned lanes)
        # the same batch again must not see stale keys / tables from the previous launches
        _, reps2 = _check_vs_oracle(tmx, oracle, 0, n, same.proofs, same.targets, same.trusteds, b"celestia", ctx=ctx)
        assert ctx.last_dedup() == (16, True)


@pytest.mark.parametrize("P", [640, 768, 1024])
def test_throughput_regime_all_rows(tmx, oracle, P):
    """The schedule of the throughput regime (round 6: from 512 proofs x 128 the row writers run at wave priority 3 with 1536 workgroups beside the
    chain; from 1024 proofs the input sections go in front of the new-key pipeline, 8192 workgroups, the leaves first): the bench workload at 640
    and 1024 proofs, cold, warm and warm with new keys in front -- EVERY row and report vs the oracle (4.4 GB of rows at 1024 proofs, compared
    128 proofs at a time)."""
    import torch
    from tendermintx_amd.synth import Workload, bench_workload
    n = 128
    wl = bench_workload("survey8d", n, P, seed=0x544D58 + P)
    fresh = Workload(0, n, 2, 90, chain_id=b"celestia", seed=31337 + P, signed_permille=950, n_sets=2)
    mixed = (fresh.proofs + wl.proofs[2 * 2336:], fresh.targets + wl.targets[2 * n * 256:], fresh.trusteds + wl.trusteds[2 * n * 48:])
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream(dev)
    with tmx.Context(n, b"celestia", max_batch=P) as ctx:
        stride, count = ctx.elem_stride(0), ctx.elem_count(0)
        out = torch.empty((P, stride), dtype=torch.int64, device=dev)
        rep = torch.zeros(P * 64, dtype=torch.uint8, device=dev)
        for run, (pr, tg, tr) in enumerate(((wl.proofs, wl.targets, wl.trusteds),) * 2 + (mixed, (wl.proofs, wl.targets, wl.trusteds))):
            d = [torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev) for b in (pr, tg, tr)]
            out.fill_(-1)
            rep.zero_()
            torch.cuda.synchronize(dev)
            ctx.witness_batch_device(0, P, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), out.data_ptr(), rep.data_ptr(), stream.cuda_stream)
            torch.cuda.synchronize(dev)
            reps = np.frombuffer(rep.cpu().numpy().tobytes(), dtype=np.uint32).reshape(P, 16)
            for p0 in range(0, P, 128):
                p1 = min(P, p0 + 128)
                want, oreps = oracle.witness_batch(0, p1 - p0, pr[p0 * 2336:p1 * 2336], tg[p0 * n * 256:p1 * n * 256], tr[p0 * n * 48:p1 * n * 48], n,
                                                   b"celestia", 100800, n_threads=os.cpu_count() or 8)
                got = out[p0:p1, :count].cpu().numpy().view(np.uint64)
                assert np.array_equal(got, want), (run, p0, np.argwhere(got != want)[:8].tolist())
                assert [int(r[8]) for r in reps[p0:p1]] == [int(o["all_ok"]) for o in oreps], (run, p0)
            assert int(out[:, count:].abs().sum().item()) == 0


def test_timed_workload_all_rows(tmx, oracle):
    """The exact batch bench.py times -- synth.bench_workload("survey8d", 128, 256): 100 validators in the 128 lanes, four validator sets,
    Bernoulli(0.9) signing re-drawn until > 2/3, rounds {0,0,0,3}, dummy-key lanes -- through the device entry point bench.py calls: all
    256 rows bit-exact vs the oracle, cold (401 distinct keys: the per-key tables are built and walked) and warm (every lane's key resident
    in the key cache: the schedule the headline is measured on), and `last_dedup` / the cache counters prove which path ran."""
    import torch
    from tendermintx_amd.synth import bench_workload
    n, P = 128, 256
    wl = bench_workload("survey8d", n, P, seed=0x544D58)
    want, oreps = oracle.witness_batch(0, P, wl.proofs, wl.targets, wl.trusteds, n, b"celestia", 100800, n_threads=os.cpu_count() or 8)
    dev = torch.device("cuda", 0)
    d = [torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev) for b in (wl.proofs, wl.targets, wl.trusteds)]
    stream = torch.cuda.current_stream(dev)
    with tmx.Context(n, b"celestia", max_batch=P) as ctx:
        stride, count = ctx.elem_stride(0), ctx.elem_count(0)
        for run in range(3):
            out = torch.full((P, stride), -1, dtype=torch.int64, device=dev)
            rep = torch.zeros(P * 64, dtype=torch.uint8, device=dev)
            ctx.witness_batch_device(0, P, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), out.data_ptr(), rep.data_ptr(), stream.cuda_stream)
            torch.cuda.synchronize(dev)
            got = out[:, :count].cpu().numpy().view(np.uint64)
            assert np.array_equal(got, want), (run, np.argwhere(got != want)[:8].tolist())
            assert int(out[:, count:].abs().sum().item()) == 0
            reps = np.frombuffer(rep.cpu().numpy().tobytes(), dtype=np.uint32).reshape(P, 16)
            assert all(int(r[8]) == 1 for r in reps) and all(o["all_ok"] for o in oreps)
            st = ctx.key_cache_stats()
            assert ctx.last_dedup() == (401, True)
            if run == 0:
                assert (st["last_new_keys"], st["last_built_keys"], st["last_hit_lanes"]) == (401, 401, 0)
            else:
                assert (st["last_new_keys"], st["last_hit_keys"], st["last_hit_lanes"]) == (0, 401, n * P)
