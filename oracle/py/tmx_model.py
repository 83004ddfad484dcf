"""Pure-Python model of the skip/step value-level witness (Level-0 + Level-1), oracle side.

TEST INFRASTRUCTURE ONLY.  Small cases only (Python big ints).  It (1) converts the reference's
fixture JSON into the packed input records of include/tmx.h, (2) computes the full witness element
stream, and is used to pin oracle/c (the fast C restatement) and to generate tests/golden/.

Reference map:
  fixture -> lanes            circuits/input/conversion.rs:59-137 (target), :139-178 (trusted)
  skip inputs / hint order    circuits/input/mod.rs:425-523, circuits/skip.rs:85-100, circuits/variables.rs:91-105
  step inputs / hint order    circuits/input/mod.rs:316-423, circuits/step.rs:73-87, circuits/variables.rs:108-120
  gadget values               circuits/builder/verify.rs:136-564, validator.rs:72-253, voting.rs:29-110,
                              shared.rs:43-215
Element widths follow SURVEY.md App. B ([MEM] for third-party structs: order `proof` then `leaf` inside
MerkleInclusionProofVariable, `r` then `s` inside EDDSASignatureVariable -- parity unpinned there).
"""
import json
import os
import struct

import ed25519_model as ed
import tm_encoding as tm

MSG_MAX = 124          # consts.rs:29
VAL_BYTES_MAX = 46     # consts.rs:22
CHAIN_ID_PB_MAX = 52   # consts.rs:9
U64 = (1 << 64) - 1

REC_VALIDATOR = 256
REC_HASHFIELD = 48
REC_HEADER = 16 + 14 * 80
REC_PROOF = 64 + 2 * REC_HEADER

FLAG_SIGNED = 1
FLAG_PRESENT = 2

KIND_SKIP = 0
KIND_STEP = 1


# ------------------------------------------------------------------------------------------------ records
def pack_validator(pk, sig, msg, msg_len, power, vlen, flags):
    assert len(pk) == 32 and len(sig) == 64 and len(msg) <= MSG_MAX
    return struct.pack("<32s64s124sHBBQ24x", pk, sig, msg.ljust(MSG_MAX, b"\0"), msg_len, vlen, flags, power)


def unpack_validator(rec):
    pk, sig, msg, msg_len, vlen, flags, power = struct.unpack("<32s64s124sHBBQ24x", rec)
    return dict(pubkey=pk, sig=sig, msg=msg, msg_len=msg_len, vlen=vlen, flags=flags, power=power)


def pack_hashfield(pk, power, vlen, flags):
    return struct.pack("<32sQBB6x", pk, power, vlen, flags)


def unpack_hashfield(rec):
    pk, power, vlen, flags = struct.unpack("<32sQBB6x", rec)
    return dict(pubkey=pk, power=power, vlen=vlen, flags=flags)


def pack_header(leaves):
    assert len(leaves) == 14 and all(len(l) < 80 for l in leaves)
    return bytes(len(l) for l in leaves) + b"\0\0" + b"".join(l.ljust(80, b"\0") for l in leaves)


class _Leaves(list):
    raw_len = None


def unpack_header(rec):
    lens = rec[:14]
    # (a length byte above 79 is clamped -- the record carries 80 bytes per field, 79 is the longest leaf a two-block SHA-256 of
    # 00 | leaf can take; same rule as oracle/c and the kernels.  Found by tests/test_oracle_cross_fuzz.py: this model read on into the next field)
    leaves = _Leaves(rec[16 + 80 * i: 16 + 80 * i + min(lens[i], 79)] for i in range(14))
    leaves.raw_len = list(lens)   # the length bytes as given: the hint elements carry them unclamped (as oracle/c and the kernels do)
    return leaves


def pack_proof(block_a, block_b, hash32, round_, nb_a, nb_b, header_a, header_b):
    return struct.pack("<QQ32sQII", block_a, block_b, hash32, round_, nb_a, nb_b) + header_a + header_b


def unpack_proof(rec):
    a, b, h, r, na, nb = struct.unpack("<QQ32sQII", rec[:64])
    return dict(block_a=a, block_b=b, hash=h, round=r, nb_a=na, nb_b=nb,
                header_a=unpack_header(rec[64:64 + REC_HEADER]),
                header_b=unpack_header(rec[64 + REC_HEADER:64 + 2 * REC_HEADER]))


# ------------------------------------------------------------------------------------------------ fixture codec
class FixtureFetcher:
    """Fixture-mode twin of the reference's InputDataFetcher (circuits/input/mod.rs:188-282)."""

    def __init__(self, fixture_path):
        self.fixture_path = fixture_path

    def _path(self, height, name):
        return os.path.join(self.fixture_path, str(height), name)

    def signed_block(self, height):
        """SignedBlockResponse (tendermint_utils.rs:52-55, 97-112): {header, data, commit, validator_set} of one block."""
        with open(self._path(height, "signed_block.json")) as f:
            return json.load(f)["result"]

    def signed_header(self, height):
        if not os.path.exists(self._path(height, "commit.json")):
            sb = self.signed_block(height)
            return {"header": sb["header"], "commit": sb["commit"]}
        with open(self._path(height, "commit.json")) as f:
            return json.load(f)["result"]["signed_header"]

    def validators(self, height):
        if not os.path.exists(self._path(height, "validators_1.json")):
            return self.signed_block(height)["validator_set"]["validators"]
        out, page, so_far = [], 1, 0
        while True:
            with open(os.path.join(self.fixture_path, str(height), f"validators_{page}.json")) as f:
                r = json.load(f)["result"]
            out.extend(r["validators"])
            so_far += int(r["count"])
            if so_far >= int(r["total"]):
                return out
            page += 1


def _vinfo(v):
    pk = tm.b64(v["pub_key"]["value"])
    power = int(v["voting_power"])
    return pk, power, len(tm.validator_bytes(pk, power)), bytes.fromhex(v["address"])


def target_lanes(validators, signed_header, n_max):
    """get_validator_data_from_block (conversion.rs:59-137)."""
    commit = signed_header["commit"]
    chain_id = signed_header["header"]["chain_id"]
    bid = commit["block_id"]
    block_id = (bytes.fromhex(bid["hash"]), int(bid["parts"]["total"]), bytes.fromhex(bid["parts"]["hash"]))
    lanes = []
    for i, cs in enumerate(commit["signatures"]):
        pk, power, vlen, _ = _vinfo(validators[i])
        if cs["block_id_flag"] == 2:
            msg = tm.sign_bytes(chain_id, int(commit["height"]), int(commit["round"]), block_id, cs["timestamp"])
            assert len(msg) <= MSG_MAX
            sig = tm.b64(cs["signature"])
            lanes.append(pack_validator(pk, sig, msg, len(msg), power, vlen, FLAG_SIGNED | FLAG_PRESENT))
        else:
            lanes.append(pack_validator(pk, ed.DUMMY_SIGNATURE, b"", 32, power, vlen, FLAG_PRESENT))
    assert len(lanes) <= n_max
    while len(lanes) < n_max:
        lanes.append(pack_validator(ed.DUMMY_PUBLIC_KEY, ed.DUMMY_SIGNATURE, b"", 32, 0, VAL_BYTES_MAX, 0))
    return lanes


def trusted_lanes(validators, commit, n_max):
    """validator_hash_field_from_block (conversion.rs:139-178); Set::new sorts by (power desc, address asc)."""
    infos = sorted((_vinfo(v) for v in validators), key=lambda t: (-t[1], t[3]))
    lanes = []
    for i in range(len(commit["signatures"])):
        pk, power, vlen, _ = infos[i]
        lanes.append(pack_hashfield(pk, power, vlen, FLAG_PRESENT))
    while len(lanes) < n_max:
        lanes.append(pack_hashfield(ed.DUMMY_PUBLIC_KEY, 0, VAL_BYTES_MAX, 0))
    return lanes


def skip_inputs_from_fixtures(fetcher, trusted_block, target_block, n_max):
    """get_skip_inputs (input/mod.rs:425-523) up to the packed records.  Returns (proof_rec, target[], trusted[])."""
    tv = fetcher.validators(trusted_block)
    gv = fetcher.validators(target_block)
    assert len(tv) <= n_max and len(gv) <= n_max
    tsh = fetcher.signed_header(trusted_block)
    gsh = fetcher.signed_header(target_block)
    th = pack_header(tm.header_leaves(tsh["header"]))
    gh = pack_header(tm.header_leaves(gsh["header"]))
    trusted_hash = tm.root_from_leaf_hashes([tm.leaf_hash(l) for l in tm.header_leaves(tsh["header"])])
    proof = pack_proof(trusted_block, target_block, trusted_hash, int(gsh["commit"]["round"]), len(gv), len(tv), gh, th)
    return proof, target_lanes(gv, gsh, n_max), trusted_lanes(tv, tsh["commit"], n_max)


def step_inputs_from_fixtures(fetcher, prev_block, n_max):
    """get_step_inputs (input/mod.rs:316-423) up to the packed records."""
    psh = fetcher.signed_header(prev_block)
    nsh = fetcher.signed_header(prev_block + 1)
    nv = fetcher.validators(prev_block + 1)
    assert len(nv) <= n_max
    ph = pack_header(tm.header_leaves(psh["header"]))
    nh = pack_header(tm.header_leaves(nsh["header"]))
    prev_hash = tm.root_from_leaf_hashes([tm.leaf_hash(l) for l in tm.header_leaves(psh["header"])])
    proof = pack_proof(prev_block, prev_block + 1, prev_hash, int(nsh["commit"]["round"]), len(nv), 0, nh, ph)
    return proof, target_lanes(nv, nsh, n_max)


# ------------------------------------------------------------------------------------------------ element stream
class Elems:
    """Goldilocks element stream: every value below is < 2^32 so it is canonical as-is."""

    def __init__(self):
        self.v = []

    def byte(self, b):
        self.v.extend((b >> (7 - k)) & 1 for k in range(8))  # ByteVariable = 8 BE bits (validator.rs:75-77)

    def bytes(self, bs):
        for b in bs:
            self.byte(b)

    def u32(self, x):
        assert 0 <= x < (1 << 32)
        self.v.append(x)

    def u64(self, x):
        self.v.append(x & 0xFFFFFFFF)
        self.v.append((x >> 32) & 0xFFFFFFFF)

    def u256(self, x):
        for k in range(8):
            self.v.append((x >> (32 * k)) & 0xFFFFFFFF)

    def bool(self, b):
        self.v.append(1 if b else 0)


def tree_nodes_count(n):
    c = 0
    while n > 1:
        n = (n + 1) // 2
        c += n
    return c


def elem_count(kind, n):
    h = (1776 * n + 5320) if kind == KIND_SKIP else (1517 * n + 6919)
    return h + derived_count(kind, n)


DT = 368 + 256 + 512 + 8 + 80 + 7 + 4     # derived elements per target lane
DR = 368 + 256 + 2 + 4                    # derived elements per trusted lane
PROOF_D = 256 + 4 * 256                   # leaf hash + 4 path nodes


def derived_count(kind, n):
    if kind == KIND_SKIP:
        return n * DT + n * DR + 2 * tree_nodes_count(n) * 256 + (4 * PROOF_D + 88) + 34
    return n * DT + tree_nodes_count(n) * 256 + (5 * PROOF_D + 88) + 25


def path_bits(index, depth=4):
    """get_path_to_leaf (shared.rs:45-65): LSB first."""
    return [(index >> k) & 1 for k in range(depth)]


def header_tree(leaves):
    lh = [tm.leaf_hash(l) for l in leaves]
    root, proofs = tm.proofs_from_leaf_hashes(lh)
    return lh, root, proofs


def proof_walk(leaf_h, index, aunts):
    """Intermediate nodes of a depth-4 proof (last one = computed root)."""
    cur, nodes = leaf_h, []
    for bit, aunt in zip(path_bits(index), aunts):
        cur = tm.inner_hash(aunt, cur) if bit else tm.inner_hash(cur, aunt)
        nodes.append(cur)
    return nodes


def tally(powers, nb, in_group, num, den):
    """verify_voting_threshold (verify.rs:439-467) = get_total_voting_power (voting.rs:31-63) +
    is_voting_power_greater_than_threshold (voting.rs:66-109).  u64 wrap-around semantics kept."""
    n = len(powers)
    total, acc = 0, 0
    tot_prefix, acc_prefix = [], []
    no_overflow = True
    enabled = True
    for i in range(n):
        if i == nb:
            enabled = False
        val = powers[i] if enabled else 0
        t2 = (total + val) & U64
        if t2 < total:
            no_overflow = False
        total = t2
        tot_prefix.append(total)
    for i in range(n):
        val = powers[i] if in_group[i] else 0
        a2 = (acc + val) & U64
        if a2 < acc:
            no_overflow = False
        acc = a2
        acc_prefix.append(acc)
    scaled_acc = (acc * den) & U64
    if scaled_acc // den != acc:
        no_overflow = False
    scaled_total = (total * num) & U64
    if scaled_total // num != total:
        no_overflow = False
    return dict(total=total, acc=acc, scaled_acc=scaled_acc, scaled_total=scaled_total,
                gt=scaled_acc > scaled_total, tot_prefix=tot_prefix, acc_prefix=acc_prefix, no_overflow=no_overflow)


def sigdata_checks(msg, header_hash, height, round_, enabled, signed):
    """verify_validator_signature_data (validator.rs:80-153) + verify_hash_in_message (:155-183)."""
    off = 16 if round_ == 0 else 25
    hash_in_msg = msg[off:off + 32] == header_hash
    is_precommit = msg[1:3] == b"\x08\x02"
    height_ok = msg[4:12] == struct.pack("<Q", height)
    round_ok = True if round_ == 0 else (msg[13:21] == struct.pack("<Q", round_))
    valid = signed and enabled and hash_in_msg and is_precommit and height_ok and round_ok
    return hash_in_msg, is_precommit, height_ok, round_ok, (signed == valid)


def eddsa_lane(v):
    """curta_eddsa_verify_sigs_conditional value semantics (called at verify.rs:248-259): !signed lanes are
    evaluated on the dummy (pubkey, signature, 32-byte zero message)."""
    if v["flags"] & FLAG_SIGNED:
        return ed.verify_trace(v["pubkey"], v["sig"], v["msg"][:min(v["msg_len"], MSG_MAX)])
    return ed.verify_trace(ed.DUMMY_PUBLIC_KEY, ed.DUMMY_SIGNATURE, ed.DUMMY_MSG)


def _emit_validator_h(E, v):
    E.bytes(v["pubkey"])
    E.bytes(v["sig"][:32])
    E.u256(int.from_bytes(v["sig"][32:], "little"))
    E.bytes(v["msg"])
    E.u32(v["msg_len"])
    E.u64(v["power"])
    E.u32(v["vlen"])
    E.bool(v["flags"] & FLAG_SIGNED)


def _emit_hash_proof_h(E, aunts, leaf, leaf_size):
    for a in aunts:
        E.bytes(a)
    E.bytes(leaf[:leaf_size].ljust(leaf_size, b"\0"))


def _emit_chain_height_h(E, leaves, proofs, height_value):
    for a in proofs[1]:
        E.bytes(a)
    E.u32(leaves.raw_len[1] if getattr(leaves, "raw_len", None) else len(leaves[1]))
    E.bytes(leaves[1][:CHAIN_ID_PB_MAX].ljust(CHAIN_ID_PB_MAX, b"\0"))
    for a in proofs[2]:
        E.bytes(a)
    E.u32(leaves.raw_len[2] if getattr(leaves, "raw_len", None) else len(leaves[2]))
    E.u64(height_value)


def _height_from_leaf(leaf):
    """Inverse of `08 varint(height)`; the reference reads header.height.value() (input/mod.rs:481)."""
    x, s = 0, 0
    for b in leaf[1:11]:
        x |= (b & 0x7F) << s
        s += 7
    return x & U64


def _valset_derived(lanes):
    """marshal (validator.rs:185-207) + leaf hash (validator.rs:209-229) per lane."""
    out = []
    for v in lanes:
        m = b"\x0a\x22\x0a\x20" + v["pubkey"] + b"\x10" + tm.varint9(v["power"])
        out.append((m, tm.leaf_hash(m[:min(v["vlen"], VAL_BYTES_MAX)])))
    return out


def _emit_proof_d(E, leaf_hash, nodes):
    E.bytes(leaf_hash)
    for nd in nodes:
        E.bytes(nd)


def witness(kind, proof_rec, target_recs, trusted_recs, chain_id, skip_max):
    """Full element stream + report for one proof.  Returns (list[int], report dict)."""
    n = len(target_recs)
    p = unpack_proof(proof_rec)
    tgt = [unpack_validator(r) for r in target_recs]
    E = Elems()
    leaves_a = p["header_a"]
    lh_a, root_a, proofs_a = header_tree(leaves_a)
    leaves_b = p["header_b"]
    lh_b, root_b, proofs_b = header_tree(leaves_b)
    header = root_a                      # target_header / next_header (mod.rs:456-457, 403)
    height_a = _height_from_leaf(leaves_a[2])
    round_ = p["round"]
    nb = p["nb_a"]

    # ---------------- H: hint elements in variables.rs field order
    E.bytes(header)
    for v in tgt:
        _emit_validator_h(E, v)
    E.u32(nb)
    E.u64(round_)
    _emit_chain_height_h(E, leaves_a, proofs_a, height_a)
    _emit_hash_proof_h(E, proofs_a[7], leaves_a[7], 34)
    if kind == KIND_SKIP:
        trs = [unpack_hashfield(r) for r in trusted_recs]
        assert len(trs) == n
        E.u32(p["nb_b"])
        _emit_hash_proof_h(E, proofs_b[7], leaves_b[7], 34)
        for t in trs:
            E.bytes(t["pubkey"])
            E.u64(t["power"])
            E.u32(t["vlen"])
    else:
        _emit_hash_proof_h(E, proofs_a[4], leaves_a[4], 72)
        _emit_hash_proof_h(E, proofs_b[8], leaves_b[8], 34)
    assert len(E.v) == ((1776 * n + 5320) if kind == KIND_SKIP else (1517 * n + 6919))

    # ---------------- D: derived Level-1 values
    expected_height = p["block_b"]       # skip: target_block ; step: prev_block + 1 (verify.rs:476-477)
    signed = [bool(v["flags"] & FLAG_SIGNED) for v in tgt]
    tder = _valset_derived(tgt)
    tal_t = tally([v["power"] for v in tgt], nb, signed, 2, 3)
    all_eddsa, all_sigdata = True, True
    first_bad_sig = -1
    # marshal_int64_varint asserts bit 63 == 0 (shared.rs:80) for every voting power it marshals (validator.rs:200, both sets) and for
    # height_proof.height (shared.rs:178)
    varint_ok = all(v["power"] < (1 << 63) for v in tgt) and height_a < (1 << 63)
    # verify_non_negative_round (validator.rs:73-78; asserted per lane at :141 on the proof-wide round, signed or not)
    round_nonneg = (round_ >> 63) == 0
    lane_vals = []
    for i, v in enumerate(tgt):      # D.1a: the byte fields of every target lane
        tr = eddsa_lane(v)
        enabled = i < nb
        chk = sigdata_checks(v["msg"], header, expected_height, round_, enabled, signed[i])
        E.bytes(tder[i][0])
        E.bytes(tder[i][1])
        E.bytes(tr["digest"])
        lane_vals.append((tr, enabled, chk))
        if not tr["ok"]:
            all_eddsa = False
            if first_bad_sig < 0:
                first_bad_sig = i
        all_sigdata = all_sigdata and chk[4]
    for i, (tr, enabled, chk) in enumerate(lane_vals):      # D.1b: the word elements of every target lane
        E.u256(tr["h"])
        for name in ("A", "R", "sB", "hA", "sum"):
            pt = tr[name] if tr[name] is not None else (0, 0)
            E.u256(pt[0])
            E.u256(pt[1])
        E.bool(tr["ok"])
        E.bool(enabled)
        for c in chk:
            E.bool(c)
        E.u64(tal_t["tot_prefix"][i])
        E.u64(tal_t["acc_prefix"][i])
    layers_t, root_t = tm.fixed_shape_layers([d[1] for d in tder], nb)

    if kind == KIND_SKIP:
        nbt = p["nb_b"]
        rder = _valset_derived(trs)
        varint_ok = varint_ok and all(t["power"] < (1 << 63) for t in trs)
        # N x N match (verify.rs:398-418)
        matched = [any(signed[i] and tgt[i]["pubkey"] == trs[j]["pubkey"] for i in range(n)) for j in range(n)]
        tal_r = tally([t["power"] for t in trs], nbt, matched, 1, 3)
        for j, t in enumerate(trs):      # D.2a: byte fields
            E.bytes(rder[j][0])
            E.bytes(rder[j][1])
        for j, t in enumerate(trs):      # D.2b: word fields
            E.bool(j < nbt)
            E.bool(matched[j])
            E.u64(tal_r["tot_prefix"][j])
            E.u64(tal_r["acc_prefix"][j])
        layers_r, root_r = tm.fixed_shape_layers([d[1] for d in rder], nbt)
    for layer in layers_t:
        for nd in layer:
            E.bytes(nd)
    if kind == KIND_SKIP:
        for layer in layers_r:
            for nd in layer:
                E.bytes(nd)

    # header section
    # verify.rs:189-202: SHA-256 over 1 + enc_len bytes of 00 | chain_id[52] | zeros (the field resized to 52 bytes, mod.rs:476-478)
    cid_leaf_hash = tm.leaf_hash(leaves_a[1][:CHAIN_ID_PB_MAX].ljust(80, b"\0")[:len(leaves_a[1])])
    cid_nodes = proof_walk(cid_leaf_hash, 1, proofs_a[1])
    hl = b"\x00\x08" + tm.varint9(height_a)                        # shared.rs:158-167, 180-181
    h_leaf_hash = tm.leaf_hash(hl[1:].ljust(80, b"\0")[:len(leaves_a[2])])   # SHA256 over 1+len bytes of `00 08 varint9 00..`
    h_nodes = proof_walk(h_leaf_hash, 2, proofs_a[2])
    v_leaf_hash = tm.leaf_hash(leaves_a[7][:34].ljust(34, b"\0"))
    v_nodes = proof_walk(v_leaf_hash, 7, proofs_a[7])
    _emit_proof_d(E, cid_leaf_hash, cid_nodes)
    E.bytes(hl)
    _emit_proof_d(E, h_leaf_hash, h_nodes)
    _emit_proof_d(E, v_leaf_hash, v_nodes)
    chain_ok = leaves_a[1][:CHAIN_ID_PB_MAX].ljust(CHAIN_ID_PB_MAX, b"\0")[2:2 + len(chain_id)] == chain_id   # verify.rs:211-221
    checks = []
    if kind == KIND_SKIP:
        tv_leaf_hash = tm.leaf_hash(leaves_b[7][:34].ljust(34, b"\0"))
        tv_nodes = proof_walk(tv_leaf_hash, 7, proofs_b[7])
        _emit_proof_d(E, tv_leaf_hash, tv_nodes)
        for tl in (tal_t, tal_r):
            E.u64(tl["total"]); E.u64(tl["acc"]); E.u64(tl["scaled_acc"]); E.u64(tl["scaled_total"]); E.bool(tl["gt"])
        trusted_block, target_block = p["block_a"], p["block_b"]
        dist_gt = target_block > ((trusted_block + 1) & U64)                 # verify.rs:508-526
        dist_le = target_block <= ((trusted_block + skip_max) & U64)
        E.bool(dist_gt); E.bool(dist_le)
        checks = [
            tv_nodes[-1] == p["hash"],                                      # verify.rs:374-379
            root_r == leaves_b[7][:34].ljust(34, b"\0")[2:34],                   # verify.rs:382-389
            root_t == leaves_a[7][:34].ljust(34, b"\0")[2:34],                   # verify.rs:279-280
            v_nodes[-1] == header,                                          # verify.rs:283-286
            cid_nodes[-1] == header,                                        # verify.rs:205-209
            chain_ok,
            h_nodes[-1] == header,                                          # shared.rs:197-203
            height_a == expected_height,                                    # shared.rs:206
            all_sigdata,
            all_eddsa,
            tal_t["no_overflow"] and tal_r["no_overflow"],
            varint_ok,                                                      # shared.rs:80
            round_nonneg,                                                   # validator.rs:73-78
        ]
        for c in checks:
            E.bool(c)
        all_ok = all(checks) and tal_t["gt"] and tal_r["gt"] and dist_gt and dist_le
        E.bool(all_ok)
    else:
        lb_leaf = leaves_a[4][:72].ljust(72, b"\0")
        lb_leaf_hash = tm.leaf_hash(lb_leaf)
        lb_nodes = proof_walk(lb_leaf_hash, 4, proofs_a[4])
        _emit_proof_d(E, lb_leaf_hash, lb_nodes)
        nv_leaf = leaves_b[8][:34].ljust(34, b"\0")
        nv_leaf_hash = tm.leaf_hash(nv_leaf)
        nv_nodes = proof_walk(nv_leaf_hash, 8, proofs_b[8])
        _emit_proof_d(E, nv_leaf_hash, nv_nodes)
        tl = tal_t
        E.u64(tl["total"]); E.u64(tl["acc"]); E.u64(tl["scaled_acc"]); E.u64(tl["scaled_total"]); E.bool(tl["gt"])
        checks = [
            root_t == leaves_a[7][:34].ljust(34, b"\0")[2:34],
            v_nodes[-1] == header,
            cid_nodes[-1] == header,
            chain_ok,
            h_nodes[-1] == header,
            height_a == expected_height,
            all_sigdata,
            all_eddsa,
            tl["no_overflow"],
            varint_ok,
            lb_nodes[-1] == header,                                         # verify.rs:144-147
            lb_leaf[2:34] == p["hash"],                                     # verify.rs:150-153
            nv_nodes[-1] == p["hash"],                                      # verify.rs:166-170
            leaves_a[7][:34].ljust(34, b"\0")[2:34] == nv_leaf[2:34],            # verify.rs:173-177
            round_nonneg,                                                   # validator.rs:73-78
        ]
        for c in checks:
            E.bool(c)
        all_ok = all(checks) and tl["gt"]
        E.bool(all_ok)
    assert len(E.v) == elem_count(kind, n), (len(E.v), elem_count(kind, n))
    fail_mask = sum((0 if c else 1) << i for i, c in enumerate(checks))
    report = dict(header=header, all_ok=all_ok, fail_mask=fail_mask, first_bad_sig=first_bad_sig,
                  gt_target=tal_t["gt"], gt_trusted=(tal_r["gt"] if kind == KIND_SKIP else None))
    return E.v, report


# ------------------------------------------------------------------------------------------------ is_valid_skip (operator side)
def pack_addr(address20, has_address, power):
    return struct.pack("<20sB3xQ", address20, 1 if has_address else 0, power)


def skipcheck_records(start_validators, target_validators, target_commit):
    """JSON objects -> (start, target, sigs) lists of 32-byte records for is_valid_skip."""
    def vset(vals):
        infos = sorted(((bytes.fromhex(v["address"]), int(v["voting_power"])) for v in vals), key=lambda t: (-t[1], t[0]))
        return [pack_addr(a, True, p) for a, p in infos]
    sigs = []
    for cs in target_commit["signatures"]:
        if cs["block_id_flag"] != 1 and cs.get("validator_address"):
            sigs.append(pack_addr(bytes.fromhex(cs["validator_address"]), True, 0))
        else:
            sigs.append(pack_addr(bytes(20), False, 0))
    return vset(start_validators), vset(target_validators), sigs


def is_valid_skip(start, target, sigs):
    """reference circuits/input/tendermint_utils.rs:444-482 on records; Python floats are IEEE doubles like Rust's f64."""
    threshold = 1.0 / 3.0
    total = sum(struct.unpack_from("<Q", t, 24)[0] for t in target) & U64
    shared, idx = 0, 0
    while float(total) * threshold > float(shared) and idx < len(start):
        for t in target:
            if t[:20] == start[idx][:20]:
                for s in sigs:
                    if s[20] and s[:20] == t[:20]:
                        shared = (shared + struct.unpack_from("<Q", t, 24)[0]) & U64
                break
        idx += 1
    return float(total) * threshold <= float(shared), shared, total
