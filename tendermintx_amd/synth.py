"""Synthetic skip / step workloads (bench + tests).  NOT part of the reference: BASELINE.json's configs ask for
"synthetic validators/signatures", so this module manufactures well-formed inputs of that shape on the host.

Independent of both the product kernels and the oracle: signatures come from OpenSSL (libcrypto EVP Ed25519 via
ctypes) and hashes from hashlib, so a batch that the GPU path accepts with all_ok == 1 has been cross-checked
against a third implementation.  Deterministic in `seed`.  Follows SURVEY.md §8(d): keys from
SHA256("tmx-key" | seed | i), mocha-like powers (1..30 M, descending), Bernoulli signing mask, trusted set = target
set rotated with ~10 % of the keys replaced, chain id / heights / rounds as given, per-lane timestamp nanos.
"""
import ctypes as C
import ctypes.util
import hashlib
import struct

_crypto = None


def _libcrypto():
    global _crypto
    if _crypto is None:
        name = ctypes.util.find_library("crypto")
        if not name:
            raise ImportError("tendermintx_amd.synth needs OpenSSL's libcrypto to sign synthetic votes")
        L = C.CDLL(name)
        L.EVP_PKEY_new_raw_private_key.restype = C.c_void_p
        L.EVP_PKEY_new_raw_private_key.argtypes = [C.c_int, C.c_void_p, C.c_char_p, C.c_size_t]
        L.EVP_PKEY_get_raw_public_key.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_size_t)]
        L.EVP_MD_CTX_new.restype = C.c_void_p
        L.EVP_MD_CTX_free.argtypes = [C.c_void_p]
        L.EVP_DigestSignInit.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.EVP_DigestSign.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_size_t), C.c_char_p, C.c_size_t]
        L.EVP_PKEY_free.argtypes = [C.c_void_p]
        _crypto = L
    return _crypto


EVP_PKEY_ED25519 = 1087


class _Key:
    def __init__(self, seed32):
        L = _libcrypto()
        self.pkey = L.EVP_PKEY_new_raw_private_key(EVP_PKEY_ED25519, None, seed32, 32)
        if not self.pkey:
            raise RuntimeError("EVP_PKEY_new_raw_private_key failed")
        buf, ln = C.create_string_buffer(32), C.c_size_t(32)
        L.EVP_PKEY_get_raw_public_key(self.pkey, buf, C.byref(ln))
        self.pub = buf.raw

    def sign(self, msg):
        L = _libcrypto()
        ctx = L.EVP_MD_CTX_new()
        try:
            if L.EVP_DigestSignInit(ctx, None, None, None, self.pkey) != 1:
                raise RuntimeError("EVP_DigestSignInit failed")
            sig, ln = C.create_string_buffer(64), C.c_size_t(64)
            if L.EVP_DigestSign(ctx, sig, C.byref(ln), msg, len(msg)) != 1:
                raise RuntimeError("EVP_DigestSign failed")
            return sig.raw
        finally:
            L.EVP_MD_CTX_free(ctx)


# dummy lane constants (plonky2x DUMMY_PUBLIC_KEY / DUMMY_SIGNATURE, see include/tmx.h and DESIGN.md): derived, not copied
_dummy = None


def dummy_lane():
    global _dummy
    if _dummy is None:
        k = _Key(bytes([1] * 32))
        _dummy = (k.pub, k.sign(bytes(32)))
    return _dummy


def _varint(n):
    out = bytearray()
    while n >= 0x80:
        out.append((n & 0x7F) | 0x80)
        n >>= 7
    out.append(n)
    return bytes(out)


def _h(tag, *ints):
    return hashlib.sha256(tag + b"".join(struct.pack("<Q", i) for i in ints)).digest()


def _leaf(b):
    return hashlib.sha256(b"\x00" + b).digest()


def _inner(l, r):
    return hashlib.sha256(b"\x01" + l + r).digest()


def _root(hs):
    if len(hs) == 1:
        return hs[0]
    k = 1
    while k * 2 < len(hs):
        k *= 2
    return _inner(_root(hs[:k]), _root(hs[k:]))


def _validator_bytes(pk, power):
    return b"\x0a\x22\x0a\x20" + pk + (b"\x10" + _varint(power) if power else b"")


def _block_id(hash32, psh32):
    return b"\x0a\x20" + hash32 + b"\x12\x24\x08\x01\x12\x20" + psh32


def _header_leaves(chain_id, height, secs, last_block_hash, valhash, next_valhash, tag):
    def hv(x):
        return b"\x0a\x20" + x
    return [b"\x08\x0b\x10\x01", b"\x0a" + _varint(len(chain_id)) + chain_id, b"\x08" + _varint(height),
            b"\x08" + _varint(secs) + b"\x10" + _varint(1 + (height * 7919) % 999999999),
            _block_id(last_block_hash, _h(b"tmx-lpsh", tag, height)), hv(_h(b"tmx-lc", tag, height)), hv(_h(b"tmx-data", tag, height)),
            hv(valhash), hv(next_valhash), hv(_h(b"tmx-cons", tag)), hv(_h(b"tmx-app", tag, height)), hv(_h(b"tmx-res", tag, height)),
            hv(_h(b"tmx-ev", tag, height)), b"\x0a\x14" + _h(b"tmx-prop", tag, height)[:20]]


def _pack_header(leaves):
    assert len(leaves) == 14 and all(len(l) < 80 for l in leaves)
    return bytes(len(l) for l in leaves) + b"\0\0" + b"".join(l.ljust(80, b"\0") for l in leaves)


def _sign_bytes(chain_id, height, round_, block_hash, psh, secs, nanos):
    body = b"\x08\x02\x11" + struct.pack("<Q", height)
    if round_:
        body += b"\x19" + struct.pack("<Q", round_)
    bid = _block_id(block_hash, psh)
    body += b"\x22" + _varint(len(bid)) + bid
    ts = b"\x08" + _varint(secs) + (b"\x10" + _varint(nanos) if nanos else b"")
    body += b"\x2a" + _varint(len(ts)) + ts + b"\x32" + _varint(len(chain_id)) + chain_id
    return _varint(len(body)) + body


class _ValidatorSet:
    """One synthetic validator set: keys, mocha-like powers (descending), its RFC-6962 root, and the trusted set derived from it
    (rotation by 3, every 10th validator replaced by a fresh key)."""

    def __init__(self, nb, seed):
        self.keys = [_Key(_h(b"tmx-key", seed, i)) for i in range(nb)]
        fresh = [_Key(_h(b"tmx-new", seed, i)) for i in range(nb)]
        step = max(1, 29_000_000 // nb)
        self.powers = [30_000_000 - i * step - (int.from_bytes(_h(b"tmx-pow", seed, i)[:4], "little") % min(step, 1000)) for i in range(nb)]
        self.root = _root([_leaf(_validator_bytes(self.keys[i].pub, self.powers[i])) for i in range(nb)])
        tr_idx = [(j + 3) % nb for j in range(nb)]
        self.tr_pk = [fresh[j].pub if j % 10 == 9 else self.keys[tr_idx[j]].pub for j in range(nb)]
        self.tr_pow = [self.powers[tr_idx[j]] for j in range(nb)]
        self.tr_root = _root([_leaf(_validator_bytes(self.tr_pk[j], self.tr_pow[j])) for j in range(nb)])


class Workload:
    """n_proofs independent skip (kind 0) or step (kind 1) inputs over `n_sets` synthetic validator sets (proof p uses set p % n_sets).

    signed_permille: every present validator signs with that probability; `ensure_two_thirds` re-draws a proof's mask (next nonce)
    until the signers hold more than 2/3 of the power, as SURVEY 8(d) prescribes for the measured workloads."""

    def __init__(self, kind, n_max, n_proofs, nb_validators=None, chain_id=b"celestia", seed=0x544D58, signed_permille=1000,
                 rounds=(0, 0, 0, 3), skip_distance=1000, n_sets=1, ensure_two_thirds=False):
        nb = n_max if nb_validators is None else nb_validators
        assert 1 <= nb <= n_max and len(chain_id) <= 13 and n_sets >= 1
        dpk, dsig = dummy_lane()
        sets = [_ValidatorSet(nb, seed if k == 0 else seed + 7919 * k) for k in range(min(n_sets, n_proofs))]
        proofs, targets, trusteds = [], [], []
        for p in range(n_proofs):
            vs = sets[p % len(sets)]
            keys, powers, tgt_root, tr_root, tr_pk, tr_pow = vs.keys, vs.powers, vs.root, vs.tr_root, vs.tr_pk, vs.tr_pow
            round_ = rounds[p % len(rounds)]
            block_a = 2_000_000 + 10 * p
            block_b = block_a + (skip_distance if kind == 0 else 1)
            secs = 1_700_000_000 + p
            if kind == 0:
                hb = _header_leaves(chain_id, block_a, secs, _h(b"tmx-lb", seed, block_a), tr_root, tr_root, seed)
                hash_b = _root([_leaf(l) for l in hb])
                ha = _header_leaves(chain_id, block_b, secs + 12 * skip_distance, _h(b"tmx-lb", seed, block_b), tgt_root, tgt_root, seed)
            else:
                hb = _header_leaves(chain_id, block_a, secs, _h(b"tmx-lb", seed, block_a), tr_root, tgt_root, seed)
                hash_b = _root([_leaf(l) for l in hb])
                ha = _header_leaves(chain_id, block_b, secs + 12, hash_b, tgt_root, tgt_root, seed)
            hash_a = _root([_leaf(l) for l in ha])
            psh = _h(b"tmx-psh", seed, block_b)
            proofs.append(struct.pack("<QQ32sQII", block_a, block_b, hash_b, round_, nb, nb if kind == 0 else 0) + _pack_header(ha) + _pack_header(hb))
            nonce = 0
            while True:
                if nonce == 0:
                    mask = [(int.from_bytes(_h(b"tmx-sgn", seed, p, i)[:4], "little") % 1000) < signed_permille for i in range(nb)]
                else:
                    mask = [(int.from_bytes(_h(b"tmx-sgn", seed, p, i, nonce)[:4], "little") % 1000) < signed_permille for i in range(nb)]
                if not ensure_two_thirds or 3 * sum(pw for pw, m in zip(powers, mask) if m) > 2 * sum(powers):
                    break
                nonce += 1
            lanes = []
            for i in range(n_max):
                if i < nb:
                    vlen = len(_validator_bytes(keys[i].pub, powers[i]))
                    if mask[i]:
                        msg = _sign_bytes(chain_id, block_b, round_, hash_a, psh, secs + 13, 1 + (i * 7919 + p * 104729) % 999_999_999)
                        assert len(msg) <= 124
                        lanes.append(struct.pack("<32s64s124sHBBQ24x", keys[i].pub, keys[i].sign(msg), msg.ljust(124, b"\0"), len(msg), vlen, 3, powers[i]))
                    else:  # BlockIDFlag absent or nil: the lane keeps key and power, carries the dummy signature (conversion.rs:98-114)
                        lanes.append(struct.pack("<32s64s124sHBBQ24x", keys[i].pub, dsig, bytes(124), 32, vlen, 2, powers[i]))
                else:
                    lanes.append(struct.pack("<32s64s124sHBBQ24x", dpk, dsig, bytes(124), 32, 46, 0, 0))
            targets.append(b"".join(lanes))
            if kind == 0:
                tl = []
                for j in range(n_max):
                    if j < nb:
                        tl.append(struct.pack("<32sQBB6x", tr_pk[j], tr_pow[j], len(_validator_bytes(tr_pk[j], tr_pow[j])), 2))
                    else:
                        tl.append(struct.pack("<32sQBB6x", dpk, 0, 46, 0))
                trusteds.append(b"".join(tl))
        self.kind, self.n_max, self.n_proofs, self.nb, self.chain_id, self.n_sets = kind, n_max, n_proofs, nb, chain_id, len(sets)
        self.proofs, self.targets = b"".join(proofs), b"".join(targets)
        self.trusteds = b"".join(trusteds) if kind == 0 else None
        self.describe = (f"{nb} validators in {n_max} lanes, {len(sets)} distinct validator set(s) per batch, "
                         f"{signed_permille / 10:.0f}% signing" + (" (re-drawn until > 2/3)" if ensure_two_thirds else "") +
                         f", rounds {{{','.join(str(r) for r in rounds)}}} cycling")


def bench_workload(name, n_max, n_proofs, seed=0x544D58):
    """The two measured skip workloads (bench.py, tools/profile_step.py).
    survey8d: SURVEY 8(d) -- 100 validators in the 128 lanes (Celestia-real; N of N at other sizes), four distinct validator sets per
              batch, Bernoulli(0.9) signing re-drawn until > 2/3 of the power signed, rounds {0,0,0,3}, trusted set = target set
              rotated with 10 % of the keys replaced.
    one_set:  the most deduplication-friendly batch (round 1's headline): ONE validator set for every proof, N of N, everybody signs."""
    if name == "survey8d":
        nb = 100 if n_max == 128 else n_max
        return Workload(0, n_max, n_proofs, nb, chain_id=b"celestia", seed=seed, signed_permille=900, rounds=(0, 0, 0, 3), n_sets=4,
                        ensure_two_thirds=True)
    if name == "one_set":
        return Workload(0, n_max, n_proofs, n_max, chain_id=b"celestia", seed=seed, signed_permille=1000, rounds=(0, 0, 0, 3))
    raise ValueError(name)
