"""Level-2 trace rows (DESIGN.md "Level-2 trace rows"): the oracle's generator (affine arithmetic, one inversion per operation) against
its own constraint checker here, and the HIP rows (projective ladder, batched inversions, wave-cooperative stores) against both on the
GPU box: bit-exact vs the generator, and accepted row by row by the checker."""
import struct

import numpy as np
import pytest


def _case(cases, name):
    c = cases[name]
    return c["kind"], c["n"], bytes.fromhex(c["proof"]), bytes.fromhex(c["target"]), (bytes.fromhex(c["trusted"]) if c["trusted"] else None)


@pytest.mark.parametrize("name", ["skip_10000_10500_n4", "step_10500_n4", "skip_3000_3100_n4"])
def test_generated_rows_satisfy_every_constraint(oracle, cases, built_lib, name):
    kind, n, proof, targets, trusteds = _case(cases, name)
    tr = oracle.trace(kind, proof, targets, trusteds, n)
    assert tr.size == oracle.trace_elem_count(kind, n) == built_lib.tmx_trace_elem_count(kind, n)
    assert int(tr.max()) < 2**32
    assert oracle.trace_check(kind, proof, targets, trusteds, n, tr) == 0
    # the last row of every ladder is the Level-1 point: s*B / h*A of the lane (checked inside trace_check against tmxo_eddsa_trace_lane)
    rng = np.random.default_rng(3)
    for _ in range(300):   # any single-bit change of any element breaks a constraint
        i = int(rng.integers(0, tr.size))
        m = tr.copy()
        m[i] ^= np.uint64(1 << int(rng.integers(0, 33)))
        assert oracle.trace_check(kind, proof, targets, trusteds, n, m) != 0, i


@pytest.mark.parametrize("name", ["skip_10000_10500_n32", "step_10500_n4", "step_10500_n100"])
def test_tree_and_header_sections(oracle, cases, name):
    """T.5 / T.6: the traced hashes are the ones Level-1 reports -- the last path node of every proof against a real header is that
    header's hash (mocha-4 fixtures), the last tree slot's digest is the validators hash in the header -- and a flipped bit anywhere in the
    two sections is rejected."""
    import ctypes as C
    kind, n, proof, targets, trusteds = _case(cases, name)
    msgs, lens, dgs = (C.c_uint8 * (5 * 5 * 96))(), (C.c_uint32 * 25)(), (C.c_uint8 * (5 * 5 * 32))()
    nq = oracle.lib().tmxo_header_proof_messages(kind, proof, msgs, lens, dgs)
    assert nq == (4 if kind == 0 else 5)
    header = bytes.fromhex(cases[name]["header"])
    roots = [bytes(dgs[(5 * q + 4) * 32:(5 * q + 5) * 32]) for q in range(nq)]
    assert roots[0] == roots[1] == roots[2] == header                      # chain id, height, validators hash: proofs against the target header
    assert roots[3] == (proof[16:48] if kind == 0 else header)              # skip: the trusted header's hash (public input); step: last block id
    if kind == 1:
        assert roots[4] == proof[16:48]                                     # next validators hash: a leaf of the previous header
    assert all(lens[5 * q + h] == 65 for q in range(nq) for h in range(1, 5)) and lens[10] == 35
    tr = oracle.trace(kind, proof, targets, trusteds, n)
    sets = 2 if kind == 0 else 1
    tn, slots = 0, n
    while slots > 1:
        slots = (slots + 1) // 2
        tn += slots
    o5 = n * (2 * 256 * 65 + 2880 + sets * 576) + (n * n if kind == 0 else 0)
    assert tr.size == o5 + (sets * tn + nq * 5) * 1152
    # the digest of the target tree's last slot (its root when every lane is enabled) = IV + the last row of block 1
    iv = [0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19]
    def digest(rows):
        mid = [(iv[k] + int(rows[63 * 9 + 1 + k])) & 0xffffffff for k in range(8)]
        return b"".join(struct.pack(">I", (mid[k] + int(rows[127 * 9 + 1 + k])) & 0xffffffff) for k in range(8))
    nb = struct.unpack_from("<I", proof, 56)[0]
    if nb >= n:
        assert digest(tr[o5 + (tn - 1) * 1152:o5 + tn * 1152]) == bytes(msgs[(5 * 2) * 96 + 3:(5 * 2) * 96 + 35])   # = the validators hash leaf value
    assert digest(tr[o5 + (sets * tn + 4) * 1152:][:1152]) == header
    rng = np.random.default_rng(9)
    for _ in range(200):
        i = int(rng.integers(o5, tr.size))
        m = tr.copy()
        m[i] ^= np.uint64(1 << int(rng.integers(0, 33)))
        rc = oracle.trace_check(kind, proof, targets, trusteds, n, m)
        assert rc >= 5_000_000_000, (i, rc)


def _sha256_rows_digest(rows):
    """the digest a two-block SHA-256 row group ends in: IV + working variables after round 63 of block 0 (+ those of block 1, if used)"""
    iv = [0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19]
    st = [(iv[k] + int(rows[63 * 9 + 1 + k])) & 0xffffffff for k in range(8)]
    if rows[64 * 9:].any():
        st = [(st[k] + int(rows[127 * 9 + 1 + k])) & 0xffffffff for k in range(8)]
    return b"".join(struct.pack(">I", w) for w in st)


@pytest.mark.parametrize("entry_id", ["skip_baseline", "skip_nb3_of_4", "step_valhash_proof_root", "skip_chain_id_root"])
def test_tree_and_header_rows_against_hashlib(oracle, entry_id):
    """T.5 / T.6 against an INDEPENDENT restatement (tests/ASSERTS.md "Level-2"): the ledger's scenarios are built with hashlib + OpenSSL
    only; here every traced digest is recomputed with hashlib from the packed records -- the leaf bytes by the reference's marshalling
    rule (validator.rs:185-207, shared.rs:67-156), every PAIR of the fixed-shape tree hashed whether or not both children are enabled
    (validator.rs:248-251 selects afterwards), the proof paths from the header leaves (verify.rs:189-209, shared.rs:183-203) -- and
    compared with what the traced rows end in.  nb < N, a 35-byte validators-hash field and a 51-character chain id are among them."""
    import hashlib
    import ledger
    e = next(x for x in ledger.ENTRIES if x[0] == entry_id)
    sc = ledger.build(e[2], **e[3])
    kind, n, proof, targets, trusteds = sc["kind"], sc["n"], sc["proof"], sc["targets"], sc["trusteds"]
    tr = oracle.trace(kind, proof, targets, trusteds, n)
    assert oracle.trace_check(kind, proof, targets, trusteds, n, tr) == 0
    sha = lambda b: hashlib.sha256(b).digest()

    def varint9(v):
        s = [(v >> (7 * i)) & 0x7f for i in range(9)]
        last = max([i for i in range(9) if s[i]] or [0])
        return bytes(b | (0x80 if i < last else 0) for i, b in enumerate(s))

    def leaf(pk, power, vlen):
        return sha(b"\x00" + (b"\x0a\x22\x0a\x20" + pk + b"\x10" + varint9(power))[:min(vlen, 46)])

    sets = [[leaf(targets[256 * i:256 * i + 32], struct.unpack_from("<Q", targets, 256 * i + 224)[0], targets[256 * i + 222]) for i in range(n)]]
    nbs = [struct.unpack_from("<I", proof, 56)[0]]
    if kind == 0:
        sets.append([leaf(trusteds[48 * j:48 * j + 32], struct.unpack_from("<Q", trusteds, 48 * j + 32)[0], trusteds[48 * j + 40]) for j in range(n)])
        nbs.append(struct.unpack_from("<I", proof, 60)[0])
    tn, sz = 0, n
    while sz > 1:
        sz = (sz + 1) // 2
        tn += sz
    o5 = n * (2 * 256 * 65 + 2880 + len(sets) * 576) + (n * n if kind == 0 else 0)
    for s, (cur, nb) in enumerate(zip(sets, nbs)):
        en, slot = [i < nb for i in range(n)], 0
        while len(cur) > 1:
            nxt, nen = [], []
            for i in range((len(cur) + 1) // 2):
                rows = tr[o5 + (s * tn + slot) * 1152:o5 + (s * tn + slot + 1) * 1152]
                if 2 * i + 1 < len(cur):
                    h = sha(b"\x01" + cur[2 * i] + cur[2 * i + 1])
                    assert _sha256_rows_digest(rows) == h, (entry_id, s, slot)              # hashed whether enabled or not
                    nxt.append(h if en[2 * i] and en[2 * i + 1] else cur[2 * i])
                else:
                    assert not rows.any()                                                   # a promoted node: no hash, zero rows
                    nxt.append(cur[2 * i])
                nen.append(en[2 * i]); slot += 1
            cur, en = nxt, nen
    # T.6: chain id (leaf 1), height (2), validators hash (7) of the target header; X, Y as the kind says
    hdr = lambda off: [proof[off + 16 + 80 * i:off + 16 + 80 * i + min(proof[off + i], 79)] for i in range(14)]
    ha, hb = hdr(64), hdr(64 + 1136)

    def rfc_root(leaves):
        if len(leaves) == 1:
            return leaves[0]
        k = 1
        while 2 * k < len(leaves):
            k *= 2
        return sha(b"\x01" + rfc_root(leaves[:k]) + rfc_root(leaves[k:]))

    def aunts(leaves, idx):
        if len(leaves) == 1:
            return []
        k = 1
        while 2 * k < len(leaves):
            k *= 2
        return aunts(leaves[:k], idx) + [rfc_root(leaves[k:])] if idx < k else aunts(leaves[k:], idx - k) + [rfc_root(leaves[:k])]

    o6 = o5 + len(sets) * tn * 1152
    plan = [(ha, 1, 52), (ha, 2, None), (ha, 7, 34)] + ([(hb, 7, 34)] if kind == 0 else [(ha, 4, 72), (hb, 8, 34)])
    for q, (h, idx, fixed) in enumerate(plan):
        lh = [sha(b"\x00" + f) for f in h]
        field = h[idx]
        if idx == 2:    # the re-encoded height: 00 08 varint9(height), cut at 1 + the field's length (shared.rs:158-194)
            v, sft = 0, 0
            for b in field[1:11]:
                v |= (b & 0x7f) << sft; sft += 7
            msg = (b"\x00\x08" + varint9(v) + bytes(80))[:1 + len(field)]
        elif idx == 1:  # chain id resized to 52 bytes, hashed over 1 + the field's length (verify.rs:189-202)
            msg = (b"\x00" + field[:52].ljust(52, b"\0") + bytes(40))[:1 + len(field)]
        else:           # the leaf as the proof struct carries it: resized to 34 / 72 bytes
            msg = b"\x00" + field[:fixed].ljust(fixed, b"\0")
        cur = sha(msg)
        assert _sha256_rows_digest(tr[o6 + (5 * q) * 1152:][:1152]) == cur, (entry_id, q, "leaf")
        for k, aunt in enumerate(aunts(lh, idx)):
            cur = sha(b"\x01" + (aunt + cur if (idx >> k) & 1 else cur + aunt))
            assert _sha256_rows_digest(tr[o6 + (5 * q + 1 + k) * 1152:][:1152]) == cur, (entry_id, q, k)
        if msg == b"\x00" + field:
            assert cur == rfc_root(lh)      # a leaf that is the header's own: the proof ends in the header hash


def test_ladder_checker_rejects_a_consistent_trace_of_another_scalar(oracle):
    """Rows that satisfy every curve relation but belong to scalar k' != k fail the bit-composition constraint."""
    import ctypes as C
    L = oracle.lib()
    bx, by = (C.c_uint8 * 32)(), (C.c_uint8 * 32)()
    L.tmxo_base_point(bx, by)
    rows = np.zeros(256 * 65, dtype=np.uint64)
    k1, k2 = bytes(range(1, 33)), bytes(range(2, 34))
    L.tmxo_trace_ladder(k1, bytes(bx), bytes(by), rows.ctypes.data_as(C.POINTER(C.c_uint64)))
    last = rows[255 * 65 + 49:255 * 65 + 65]
    rx = b"".join(struct.pack("<I", int(w)) for w in last[:8]); ry = b"".join(struct.pack("<I", int(w)) for w in last[8:])
    L.tmxo_trace_ladder_check.restype = C.c_int
    assert L.tmxo_trace_ladder_check(rows.ctypes.data_as(C.POINTER(C.c_uint64)), k1, bytes(bx), bytes(by), rx, ry) == 0
    assert L.tmxo_trace_ladder_check(rows.ctypes.data_as(C.POINTER(C.c_uint64)), k2, bytes(bx), bytes(by), rx, ry) % 1000 == 2
    assert L.tmxo_trace_ladder_check(rows.ctypes.data_as(C.POINTER(C.c_uint64)), k1, bytes(bx), bytes(by), ry, rx) == 257008


def _gpu_trace(tmx, kind, n, proofs, targets, trusteds, sections=63):
    import torch
    P = len(proofs) // 2336
    dev = torch.device("cuda", 0)
    d = [torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev) if b else None for b in (proofs, targets, trusteds)]
    with tmx.Context(n, b"celestia", max_batch=P) as ctx:
        te = ctx.trace_elem_count(kind)
        out = torch.zeros((P, ctx.elem_stride(kind)), dtype=torch.int64, device=dev)
        rep = torch.zeros(P * 64, dtype=torch.uint8, device=dev)
        tr = torch.full((P, te), -1, dtype=torch.int64, device=dev)
        ctx.witness_batch_device(kind, P, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr() if d[2] is not None else None, out.data_ptr(), rep.data_ptr(), 0)
        ctx.trace_rows_device(kind, P, d[1].data_ptr(), d[2].data_ptr() if d[2] is not None else None, tr.data_ptr(), sections, 0)
        torch.cuda.synchronize(dev)
    return tr.cpu().numpy().view(np.uint64)


@pytest.mark.gpu
def test_hip_rows_equal_the_generator_and_pass_the_checker(built_lib, oracle, cases):
    import tendermintx_amd as tmx
    from tendermintx_amd.synth import Workload
    for name in ("skip_10000_10500_n4", "step_10500_n4"):
        kind, n, proof, targets, trusteds = _case(cases, name)
        got = _gpu_trace(tmx, kind, n, proof, targets, trusteds)
        assert np.array_equal(got[0], oracle.trace(kind, proof, targets, trusteds, n)), name
        assert oracle.trace_check(kind, proof, targets, trusteds, n, got[0]) == 0
    for kind, n, P, nb in ((0, 7, 5, 6), (1, 16, 3, 16), (0, 33, 2, 30)):
        wl = Workload(kind, n, P, nb, chain_id=b"celestia", seed=77 + n, signed_permille=800, rounds=(0, 2))
        targets = bytearray(wl.targets)
        targets[1 * 256:1 * 256 + 32] = (2).to_bytes(32, "little")            # proof 0, lane 1: undecodable public key -> zero ladders
        targets[2 * 256 + 40] ^= 0x10                                          # lane 2: corrupted R (decodes or not: both paths are legal)
        targets[(n + 3) * 256 + 64 + 31] |= 0xF0                               # proof 1, lane 3: s >= 2^252 (non-canonical, still a ladder)
        got = _gpu_trace(tmx, kind, n, wl.proofs, bytes(targets), wl.trusteds)
        for p in range(P):
            t = bytes(targets[p * n * 256:(p + 1) * n * 256])
            r = wl.trusteds[p * n * 48:(p + 1) * n * 48] if kind == 0 else None
            pr = wl.proofs[p * 2336:(p + 1) * 2336]
            assert np.array_equal(got[p], oracle.trace(kind, pr, t, r, n)), (kind, n, p)
            assert oracle.trace_check(kind, pr, t, r, n, got[p]) == 0


@pytest.mark.gpu
def test_hip_rows_at_n128(built_lib, oracle):
    """BASELINE configs[2] shape: two proofs at N = 128 (38 MB of rows each): bit-exact vs the generator, accepted by the checker; a section
    mask leaves the other sections untouched."""
    import tendermintx_amd as tmx
    from tendermintx_amd.synth import bench_workload
    n, P = 128, 2
    wl = bench_workload("survey8d", n, P, seed=5)
    got = _gpu_trace(tmx, 0, n, wl.proofs, wl.targets, wl.trusteds)
    for p in range(P):
        t, r = wl.targets[p * n * 256:(p + 1) * n * 256], wl.trusteds[p * n * 48:(p + 1) * n * 48]
        pr = wl.proofs[p * 2336:(p + 1) * 2336]
        assert oracle.trace_check(0, pr, t, r, n, got[p]) == 0
        assert np.array_equal(got[p], oracle.trace(0, pr, t, r, n))
    only = _gpu_trace(tmx, 0, n, wl.proofs, wl.targets, wl.trusteds, sections=2 | 8)
    lad = n * 2 * 256 * 65
    assert (only[:, :lad] == np.uint64(2**64 - 1)).all() and np.array_equal(only[:, lad:lad + n * 2880], got[:, lad:lad + n * 2880])
