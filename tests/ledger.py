"""Assert ledger: one scenario per `assert*` of the reference's skip / step circuits, with the verdict the REFERENCE TEXT implies.

TEST INFRASTRUCTURE.  tests/ASSERTS.md is the human-readable table; this module is its executable form.  Every entry names the
reference assertion (file:line), builds an input that violates that assertion and -- wherever the input boundary allows it -- no other,
and states the expected report fields (`fail_mask` in the bit order of DESIGN.md "checks", the three verdict bools, `first_bad_sig`).
The expectations are written by hand from circuits/builder/{verify,validator,voting,shared}.rs, not computed by oracle/ or by the kernels:
the inputs are manufactured with hashlib + OpenSSL only (tendermintx_amd.synth helpers), so a shared misreading of the reference in
oracle/ and in the HIP path shows up here as a red line.  tests/test_assert_ledger.py runs every entry against oracle/c and oracle/py
(`-m "not gpu"`) and against the HIP path through the C ABI (`-m gpu`).
"""
import hashlib
import struct

from tendermintx_amd import synth as S

SKIP, STEP = 0, 1
SKIP_MAX = 100800

# fail_mask bits (DESIGN.md "checks")
SK_TRUSTED_ROOT, SK_TRUSTED_VALHASH, SK_VALHASH, SK_VALHASH_ROOT, SK_CID_ROOT, SK_CID_BYTES, SK_HEIGHT_ROOT, SK_HEIGHT, SK_SIGDATA, \
    SK_EDDSA, SK_OVERFLOW, SK_VARINT, SK_ROUND = (1 << k for k in range(13))
ST_VALHASH, ST_VALHASH_ROOT, ST_CID_ROOT, ST_CID_BYTES, ST_HEIGHT_ROOT, ST_HEIGHT, ST_SIGDATA, ST_EDDSA, ST_OVERFLOW, ST_VARINT, \
    ST_LBID_ROOT, ST_PREV_IN_LBID, ST_PREV_NVH_ROOT, ST_NVH_EQ, ST_ROUND = (1 << k for k in range(15))


def _key(tag, i):
    return S._Key(S._h(tag, 0x1ED6E7, i))


def _root_of_set(vals):
    return S._root([S._leaf(S._validator_bytes(pk, pw)) for pk, pw in vals])


def build(kind, n=4, nb=4, nbt=None, powers=(40_000_000, 30_000_000, 20_000_000, 10_000_000), signed=None, round_=0,
          block_a=2_000_000, dist=1000, hdr_height=None, chain_id=b"celestia", ctx_chain_id=None, msg_chain_id=None, lane=None,
          extra_signed_lanes=(), height_leaf=None, valhash_leaf_suffix=b"", trusted=None, trusted_extra=(), trusted_valhash_suffix=b"",
          lbid=None, lbid_hash=None, hdr_b_nvh=None, proof_hash=None, mutate=None):
    """One proof.  Everything that is signed is signed AFTER the overrides (a consistent scenario); `mutate(proof, targets, trusteds)`
    edits the packed records afterwards (a raw mutation).  lane = {i: {msg_height, msg_round, msg_type, msg_hash, sig_flip}}."""
    lane = lane or {}
    nbt = nb if nbt is None else nbt
    keys = [_key(b"ledger-key", i) for i in range(n + len(extra_signed_lanes))]
    powers = list(powers)[:nb]
    signed = [True] * nb if signed is None else list(signed)
    tgt = [(keys[i].pub, powers[i]) for i in range(nb)]
    tgt_root = _root_of_set(tgt)
    if trusted is None:  # the target set rotated by one (order differs: the N x N match must not depend on position)
        trusted = [tgt[(j + 1) % nb] for j in range(nb)][:nbt]
    tr_root = _root_of_set(trusted[:nbt])
    block_b = block_a + (dist if kind == SKIP else 1)
    hdr_height = block_b if hdr_height is None else hdr_height
    secs = 1_700_000_000
    tag = 7
    if kind == SKIP:
        hb = S._header_leaves(chain_id, block_a, secs, S._h(b"ledger-lb", block_a), tr_root, tr_root, tag)
    else:
        hb = S._header_leaves(chain_id, block_a, secs, S._h(b"ledger-lb", block_a), tr_root, tgt_root if hdr_b_nvh is None else hdr_b_nvh, tag)
    if trusted_valhash_suffix:
        hb[7] = hb[7] + trusted_valhash_suffix
    hash_b = S._root([S._leaf(l) for l in hb])
    last_hash = S._h(b"ledger-lb", block_b) if kind == SKIP else (hash_b if lbid_hash is None else lbid_hash)
    ha = S._header_leaves(chain_id, hdr_height, secs + 12, last_hash, tgt_root, tgt_root, tag)
    if lbid is not None:
        ha[4] = lbid(last_hash)
    if height_leaf is not None:
        ha[2] = height_leaf
    if valhash_leaf_suffix:
        ha[7] = ha[7] + valhash_leaf_suffix
    hash_a = S._root([S._leaf(l) for l in ha])
    psh = S._h(b"ledger-psh", block_b)
    mcid = chain_id[:13] if msg_chain_id is None else msg_chain_id
    dpk, dsig = S.dummy_lane()

    def signed_lane(i, key, power):
        o = lane.get(i, {})
        body = o.get("msg_type", b"\x08\x02") + b"\x11" + struct.pack("<Q", o.get("msg_height", block_b))
        r = o.get("msg_round", round_)
        if r:
            body += b"\x19" + struct.pack("<Q", r)
        bid = S._block_id(o.get("msg_hash", hash_a), psh)
        body += b"\x22" + S._varint(len(bid)) + bid
        ts = b"\x08" + S._varint(secs + 13) + b"\x10" + S._varint(1 + i * 7919)
        body += b"\x2a" + S._varint(len(ts)) + ts + b"\x32" + S._varint(len(mcid)) + mcid
        msg = S._varint(len(body)) + body
        assert len(msg) <= 124
        sig = bytearray(key.sign(msg))
        if "sig_flip" in o:
            sig[o["sig_flip"]] ^= 1
        return struct.pack("<32s64s124sHBBQ24x", key.pub, bytes(sig), msg.ljust(124, b"\0"), len(msg), len(S._validator_bytes(key.pub, power)), 3, power)

    lanes = []
    for i in range(n):
        if i < nb:
            if signed[i]:
                lanes.append(signed_lane(i, keys[i], powers[i]))
            else:  # present, did not sign: real key and power, dummy signature (conversion.rs:98-114)
                lanes.append(struct.pack("<32s64s124sHBBQ24x", keys[i].pub, dsig, bytes(124), 32, len(S._validator_bytes(keys[i].pub, powers[i])), 2, powers[i]))
        elif i - nb < len(extra_signed_lanes):  # a signed lane behind nb_enabled (the host never builds one; the circuit must reject it)
            lanes.append(signed_lane(i, keys[i], extra_signed_lanes[i - nb]))
        else:
            lanes.append(struct.pack("<32s64s124sHBBQ24x", dpk, dsig, bytes(124), 32, 46, 0, 0))
    trs = None
    if kind == SKIP:
        tl = []
        allt = list(trusted[:nbt]) + list(trusted_extra)
        for j in range(n):
            if j < len(allt):
                pk, pw = allt[j]
                tl.append(struct.pack("<32sQBB6x", pk, pw, min(46, len(S._validator_bytes(pk, pw & (2**63 - 1)))), 2))
            else:
                tl.append(struct.pack("<32sQBB6x", dpk, 0, 46, 0))
        trs = bytearray(b"".join(tl))
    proof = bytearray(struct.pack("<QQ32sQII", block_a, block_b, hash_b if proof_hash is None else proof_hash, round_, nb, nbt if kind == SKIP else 0)
                      + S._pack_header(ha) + S._pack_header(hb))
    targets = bytearray(b"".join(lanes))
    if mutate:
        mutate(proof, targets, trs)
    return dict(kind=kind, n=n, proof=bytes(proof), targets=bytes(targets), trusteds=bytes(trs) if trs is not None else None,
                chain_id=chain_id[:50] if ctx_chain_id is None else ctx_chain_id, skip_max=SKIP_MAX, header=hash_a, keys=keys)


def _exp(fail_mask=0, gt_target=True, gt_trusted=True, dist_ok=True, first_bad_sig=-1, kind=SKIP):
    ok = fail_mask == 0 and gt_target and (kind == STEP or (gt_trusted and dist_ok))
    return dict(all_ok=ok, fail_mask=fail_mask, gt_target=gt_target, gt_trusted=gt_trusted if kind == SKIP else False,
                dist_ok=dist_ok if kind == SKIP else False, first_bad_sig=first_bad_sig)


def _set_u64(buf, off, v):
    buf[off:off + 8] = struct.pack("<Q", v)


def _bump_target_power(lane_i):
    def f(proof, targets, trs):
        off = lane_i * 256 + 224
        _set_u64(targets, off, struct.unpack_from("<Q", targets, off)[0] + 1)
    return f


def _bump_trusted_power(j):
    def f(proof, targets, trs):
        off = j * 48 + 32
        _set_u64(trs, off, struct.unpack_from("<Q", trs, off)[0] + 1)
    return f


def _flip_proof_hash(proof, targets, trs):
    proof[16 + 5] ^= 0x40


def _bump_block_b(proof, targets, trs):
    _set_u64(proof, 8, struct.unpack_from("<Q", proof, 8)[0] + 1)


def _short_lbid(h):  # a 70-byte BlockID encoding (30-byte part-set hash): not the 72 bytes the proof type carries (mod.rs:311 would panic)
    return b"\x0a\x20" + h + b"\x12\x22\x08\x01\x12\x1e" + bytes(range(30))


def _nonminimal_height_leaf(h):  # 08 | varint(h) with one redundant continuation byte: decodes to h, is not what the circuit re-encodes
    v = S._varint(h)
    return b"\x08" + v[:-1] + bytes([v[-1] | 0x80, 0x00])


def _fresh_set(nb, powers):
    return [(_key(b"ledger-fresh", j).pub, powers[j]) for j in range(nb)]


P4 = (40_000_000, 30_000_000, 20_000_000, 10_000_000)
BIG = 2**63 - 1

# (id, reference assertion, kind, build kwargs, expected report fields) -- expectations derived from the reference text, see ASSERTS.md
ENTRIES = [
    # ---- baseline
    ("skip_baseline", "all of verify_skip (verify.rs:528-563) holds", SKIP, dict(), _exp()),
    ("skip_round3_baseline", "round != 0: hash at [25..57], round at [13..21] (validator.rs:125-141, 166-168)", SKIP, dict(round_=3), _exp()),
    ("skip_nb3_of_4", "enabled = idx < nb (verify.rs:306-313); dummy lanes behind nb", SKIP, dict(nb=3), _exp()),
    ("skip_unsigned_lane", "a present validator that did not sign: dummy signature, still > 2/3 (conversion.rs:98-114)", SKIP,
     dict(signed=[1, 1, 1, 0]), _exp()),
    # ---- verify_trusted_validators
    ("skip_trusted_proof_root", "verify.rs:379 assert_is_equal(header_from_validator_root_proof, trusted_header) (+ host mod.rs:450-455)",
     SKIP, dict(mutate=_flip_proof_hash), _exp(SK_TRUSTED_ROOT)),
    ("skip_trusted_valhash_len", "verify.rs:379: a trusted validators_hash field that is not 34 bytes (host: mod.rs:311 unwrap)", SKIP,
     dict(trusted_valhash_suffix=b"\x07"), _exp(SK_TRUSTED_ROOT)),
    ("skip_trusted_valhash", "verify.rs:389 assert_is_equal(computed_val_hash, expected_val_hash)", SKIP,
     dict(mutate=_bump_trusted_power(1)), _exp(SK_TRUSTED_VALHASH)),
    ("skip_trusted_third", "verify.rs:466 via :430-436: matched power * 3 > total (1/3 of the trusted set)", SKIP,
     dict(trusted=_fresh_set(4, P4)), _exp(gt_trusted=False)),
    ("skip_trusted_exact_third", "voting.rs:108 strict gt: matched = 1/3 exactly is not enough", SKIP,
     dict(powers=(10, 10, 10, 10), signed=[1, 1, 1, 1],
          trusted=[(_key(b"ledger-key", 0).pub, 10), (_key(b"ledger-fresh", 1).pub, 10), (_key(b"ledger-fresh", 2).pub, 10)], nbt=3),
     _exp(gt_trusted=False)),
    ("skip_trusted_match_needs_signed", "verify.rs:408-416: only SIGNED target validators mark a trusted validator", SKIP,
     dict(powers=(10, 10, 10, 10), signed=[0, 1, 1, 1],
          trusted=[(_key(b"ledger-key", 0).pub, 10), (_key(b"ledger-fresh", 1).pub, 10), (_key(b"ledger-fresh", 2).pub, 9)], nbt=3),
     _exp(gt_trusted=False)),
    # ---- verify_header
    ("skip_valhash_power", "verify.rs:280 assert_is_equal(extracted_hash, computed_validators_hash)", SKIP, dict(mutate=_bump_target_power(1)),
     _exp(SK_VALHASH)),
    ("skip_valhash_vlen", "verify.rs:280 through validator_byte_length (validator.rs:217-228: 1 + len bytes hashed)", SKIP,
     dict(mutate=lambda p, t, r: t.__setitem__(2 * 256 + 222, t[2 * 256 + 222] - 1)), _exp(SK_VALHASH)),
    ("skip_valhash_proof_root", "verify.rs:286 assert_is_equal(*header, header_from_validator_root_proof): validators_hash field of 35 bytes",
     SKIP, dict(valhash_leaf_suffix=b"\x01"), _exp(SK_VALHASH_ROOT)),
    ("skip_two_thirds", "verify.rs:466 via :289-303: signed power * 3 > total * 2", SKIP,
     dict(powers=(10, 10, 10, 10), signed=[1, 1, 0, 0]), _exp(gt_target=False)),
    ("skip_two_thirds_exact", "voting.rs:108 strict gt: exactly 2/3 (20 of 30) is not enough", SKIP,
     dict(nb=3, powers=(10, 10, 10), signed=[1, 1, 0]), _exp(gt_target=False)),
    ("skip_two_thirds_one_more", "voting.rs:108: 20 of 29 is enough (60 > 58)", SKIP, dict(nb=3, powers=(10, 10, 9), signed=[1, 1, 0]), _exp()),
    # ---- verify_validator_signature_data (validator.rs:143-152: signed == signed & enabled & hash & precommit & height & round)
    ("skip_signed_not_enabled", "validator.rs:152 (b): a signed lane at idx >= nb", SKIP, dict(nb=3, extra_signed_lanes=(5_000_000,)),
     _exp(SK_SIGDATA)),
    ("skip_hash_not_in_msg", "validator.rs:152 (c) / verify_hash_in_message :155-183", SKIP,
     dict(lane={1: dict(msg_hash=hashlib.sha256(b"other block").digest())}), _exp(SK_SIGDATA)),
    ("skip_prevote", "validator.rs:152 (d): type bytes 08 01 instead of 08 02 (:100-109)", SKIP, dict(lane={2: dict(msg_type=b"\x08\x01")}),
     _exp(SK_SIGDATA)),
    ("skip_msg_height", "validator.rs:152 (e): sfixed64 height at [4..12] (:111-123)", SKIP, dict(lane={0: dict(msg_height=2_001_001)}),
     _exp(SK_SIGDATA)),
    ("skip_msg_round", "validator.rs:152 (f): round != 0 and the message carries another round (:125-141)", SKIP,
     dict(round_=3, lane={3: dict(msg_round=2)}), _exp(SK_SIGDATA)),
    ("skip_msg_has_round_proof_has_none", "validator.rs:166-183: round == 0 selects [16..48]; a message that carries a round has the hash at 25",
     SKIP, dict(round_=0, lane={1: dict(msg_round=4)}), _exp(SK_SIGDATA)),
    ("skip_round_negative", "validator.rs:77 assert_is_equal(le_encoded_round[7].as_be_bits()[0], zero) (called at :141)", SKIP,
     dict(round_=2**63 + 5), _exp(SK_ROUND)),
    ("skip_round_negative_min", "validator.rs:77: round = 2^63", SKIP, dict(round_=2**63), _exp(SK_ROUND)),
    ("skip_round_negative_max", "validator.rs:77: round = 2^64 - 1", SKIP, dict(round_=2**64 - 1), _exp(SK_ROUND)),
    ("skip_round_max_positive", "validator.rs:77 holds for round = 2^63 - 1", SKIP, dict(round_=2**63 - 1), _exp()),
    # ---- EdDSA (curta_eddsa_verify_sigs_conditional, verify.rs:248-259; host twin conversion.rs:48-49)
    ("skip_bad_signature", "verify.rs:248-259 / conversion.rs:48-49: s corrupted on lane 2", SKIP, dict(lane={2: dict(sig_flip=40)}),
     _exp(SK_EDDSA, first_bad_sig=2)),
    ("skip_bad_signature_R", "verify.rs:248-259: R corrupted on lane 0", SKIP, dict(lane={0: dict(sig_flip=3)}), _exp(SK_EDDSA, first_bad_sig=0)),
    ("skip_unsigned_garbage_sig", "verify.rs:248-259 is conditional: an unsigned lane's signature bytes are not checked", SKIP,
     dict(signed=[1, 1, 1, 0], mutate=lambda p, t, r: t.__setitem__(slice(3 * 256 + 32, 3 * 256 + 96), bytes(range(64)))), _exp()),
    # ---- chain id / height (verify.rs:180-222, shared.rs:169-207)
    ("skip_chain_id_bytes", "verify.rs:221 assert_is_equal(extracted_chain_id, expected_chain_id)", SKIP, dict(ctx_chain_id=b"celestiA"),
     _exp(SK_CID_BYTES)),
    ("skip_chain_id_root", "verify.rs:210: a chain-id field longer than 52 bytes is truncated by mod.rs:476-478, its leaf is not the header's",
     SKIP, dict(chain_id=b"c" * 51, msg_chain_id=b"c" * 13), _exp(SK_CID_ROOT)),
    ("skip_height_root", "shared.rs:203: the height leaf is re-encoded from the value (:178-194); a non-minimal varint in the header differs",
     SKIP, dict(height_leaf=_nonminimal_height_leaf(2_001_000)), _exp(SK_HEIGHT_ROOT)),
    ("skip_height_value", "shared.rs:206 assert_is_equal(height_proof.height, expected_height)", SKIP, dict(hdr_height=2_001_001),
     _exp(SK_HEIGHT)),
    ("skip_target_block_off_by_one", "shared.rs:206 AND validator.rs:152 (e): verify_header passes target_block to both (verify.rs:547-557)",
     SKIP, dict(mutate=_bump_block_b), _exp(SK_HEIGHT | SK_SIGDATA)),
    # ---- voting.rs overflow asserts and the varint msb
    ("skip_total_overflow", "voting.rs:58 assert_is_equal(overflow, false) in get_total_voting_power (and :88, every lane signed)", SKIP,
     dict(powers=(BIG, BIG, BIG, 7), signed=[1, 1, 1, 1]), dict(fail_mask=SK_OVERFLOW, all_ok=False)),
    ("skip_acc_scaled_overflow", "voting.rs:98: accumulated * 3 wraps (no sum wraps: total = 3 * 2^61)", SKIP,
     dict(powers=(2**61, 2**61, 2**61, 5), signed=[1, 1, 1, 1]), dict(fail_mask=SK_OVERFLOW, all_ok=False)),
    ("skip_total_scaled_overflow", "voting.rs:105: total * 2 wraps (total = 2^63) -- and :98", SKIP,
     dict(powers=(2**62, 2**62, 3, 2), signed=[1, 1, 1, 1]), dict(fail_mask=SK_OVERFLOW, all_ok=False)),
    ("skip_trusted_acc_overflow", "voting.rs:88 in the trusted tally: in_group is not masked by enabled, matched lanes behind nb_trusted wrap the sum",
     SKIP, dict(trusted_extra=((_key(b"ledger-key", 0).pub, BIG), (_key(b"ledger-key", 1).pub, BIG), (_key(b"ledger-key", 2).pub, BIG)), nbt=1, n=4,
                trusted=[(_key(b"ledger-key", 3).pub, 10)]), dict(fail_mask=SK_OVERFLOW, all_ok=False)),
    ("skip_varint_msb_trusted", "shared.rs:80 assert_is_equal(value_bits[63], zero): marshalled for every lane, enabled or not (verify.rs:349-351)",
     SKIP, dict(nbt=3, trusted_extra=((_key(b"ledger-fresh", 9).pub, 2**63 + 5),)), _exp(SK_VARINT)),
    ("skip_varint_msb_height", "shared.rs:80 on height_proof.height (:178): bit 63 is dropped from the re-encoded leaf, so :203 fails with it",
     SKIP, dict(block_a=2**63 + 100, dist=1000), _exp(SK_VARINT | SK_HEIGHT_ROOT)),
    # ---- verify_skip_distance
    ("skip_dist_adjacent", "verify.rs:519 assert target > trusted + 1", SKIP, dict(dist=1), _exp(dist_ok=False)),
    ("skip_dist_two", "verify.rs:519 holds at trusted + 2", SKIP, dict(dist=2), _exp()),
    ("skip_dist_max", "verify.rs:525 holds at trusted + skip_max", SKIP, dict(dist=SKIP_MAX), _exp()),
    ("skip_dist_too_far", "verify.rs:525 assert target <= trusted + skip_max", SKIP, dict(dist=SKIP_MAX + 1), _exp(dist_ok=False)),
    # ---- step
    ("step_baseline", "all of verify_step (verify.rs:469-506) holds", STEP, dict(), _exp(kind=STEP)),
    ("step_round3_baseline", "round != 0", STEP, dict(round_=3), _exp(kind=STEP)),
    ("step_lbid_root", "verify.rs:148 assert_is_equal(header_from_last_block_id_proof, *header): a 70-byte last_block_id field", STEP,
     dict(lbid=_short_lbid), _exp(ST_LBID_ROOT, kind=STEP)),
    ("step_prev_in_lbid", "verify.rs:153 assert_is_equal(prev_header, extracted_prev_header_hash)", STEP,
     dict(lbid_hash=hashlib.sha256(b"not the previous header").digest()), _exp(ST_PREV_IN_LBID, kind=STEP)),
    ("step_prev_hash_public", "verify.rs:153 AND :169 both compare with the public prev_header_hash (+ host mod.rs:324-329)", STEP,
     dict(mutate=_flip_proof_hash), _exp(ST_PREV_IN_LBID | ST_PREV_NVH_ROOT, kind=STEP)),
    ("step_prev_nvh_root", "verify.rs:169 assert_is_equal(computed_prev_header_root, *prev_header)", STEP,
     dict(mutate=lambda p, t, r: p.__setitem__(64 + 1136 + 16 + 80 * 9 + 5, p[64 + 1136 + 16 + 80 * 9 + 5] ^ 1)), _exp(ST_PREV_NVH_ROOT, kind=STEP)),
    ("step_nvh_equal", "verify.rs:174-177 assert_is_equal(new_validators_hash, prev header's next_validators_hash)", STEP,
     dict(hdr_b_nvh=hashlib.sha256(b"another validator set").digest()), _exp(ST_NVH_EQ, kind=STEP)),
    ("step_valhash", "verify.rs:280", STEP, dict(mutate=_bump_target_power(0)), _exp(ST_VALHASH, kind=STEP)),
    ("step_valhash_proof_root", "verify.rs:286", STEP, dict(valhash_leaf_suffix=b"\x00\x01"), _exp(ST_VALHASH_ROOT, kind=STEP)),
    ("step_chain_id_bytes", "verify.rs:221", STEP, dict(ctx_chain_id=b"mocha-4"), _exp(ST_CID_BYTES, kind=STEP)),
    ("step_chain_id_root", "verify.rs:210", STEP, dict(chain_id=b"z" * 53, msg_chain_id=b"z"), _exp(ST_CID_ROOT, kind=STEP)),
    ("step_height_value", "shared.rs:206 with expected = prev + 1 (verify.rs:476-477)", STEP, dict(hdr_height=2_000_002),
     _exp(ST_HEIGHT, kind=STEP)),
    ("step_sigdata", "validator.rs:152", STEP, dict(lane={1: dict(msg_type=b"\x08\x20")}), _exp(ST_SIGDATA, kind=STEP)),
    ("step_bad_signature", "verify.rs:248-259", STEP, dict(lane={3: dict(sig_flip=63)}), _exp(ST_EDDSA, first_bad_sig=3, kind=STEP)),
    ("step_overflow", "voting.rs:58", STEP, dict(powers=(BIG, BIG, BIG, 1)), dict(fail_mask=ST_OVERFLOW, all_ok=False)),
    ("step_varint_msb", "shared.rs:80 on a disabled target lane (marshalled for every lane, verify.rs:349-351; not in any sum)", STEP,
     dict(nb=3, mutate=lambda p, t, r: _set_u64(t, 3 * 256 + 224, 2**63)), _exp(ST_VARINT, kind=STEP)),
    ("step_round_negative", "validator.rs:77", STEP, dict(round_=2**63 + 5), _exp(ST_ROUND, kind=STEP)),
    ("step_two_thirds", "verify.rs:466", STEP, dict(powers=(10, 10, 10, 10), signed=[1, 0, 1, 0]), _exp(gt_target=False, kind=STEP)),
]


def check(entry_id, report, expect):
    """Only the fields the entry states are compared (overflow entries leave the wrapped comparisons unspecified)."""
    for k, v in expect.items():
        assert report[k] == v, f"{entry_id}: {k} = {report[k]!r}, the reference text implies {v!r} (report {report})"
