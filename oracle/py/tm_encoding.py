"""Tendermint / CometBFT byte encodings on the skip/step hot path (pure Python, oracle side).

TEST INFRASTRUCTURE ONLY.  Restates, from the wire formats, what the reference obtains from the
un-vendored crates tendermint 0.33.2 / tendermint-proto 0.33.2 / prost 0.11.9 (Cargo.lock:4247-4278):
  * header -> 14 protobuf leaf encodings        reference circuits/input/tendermint_utils.rs:374-393
  * validator -> SimpleValidator bytes           reference circuits/input/conversion.rs:75 (hash_bytes)
                                                 and circuits/builder/validator.rs:185-207 (in-circuit marshal)
  * commit sig -> CanonicalVote sign-bytes       reference circuits/input/conversion.rs:33-39,
                                                 tendermint_utils.rs:404-441
  * RFC-6962 Merkle tree / proofs                reference circuits/input/tendermint_utils.rs:214-372
All of it is pinned by the reference's fixtures (block_id.hash, validators_hash, signatures).
"""
import base64
import calendar
import hashlib


def varint(n):
    """protobuf base-128 varint of a non-negative int."""
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def varint9(n):
    """Fixed 9-byte in-circuit varint (reference circuits/builder/shared.rs:67-156): septet i carries the
    continuation bit iff i < index of the last non-zero septet; trailing bytes are zero."""
    assert 0 <= n < (1 << 64)   # bit 63 is not part of any septet; the circuit asserts it is 0 (shared.rs:80) -- a check bit, not an exception
    septets = [(n >> (7 * i)) & 0x7F for i in range(9)]
    last = 0
    for i in range(9):
        if septets[i]:
            last = i
    return bytes(septets[i] | (0x80 if i < last else 0) for i in range(9))


def pb_bytes(field, b):
    return bytes([(field << 3) | 2]) + varint(len(b)) + b


def pb_varint(field, n):
    return bytes([(field << 3) | 0]) + varint(n)


def parse_time(ts):
    """RFC 3339 UTC timestamp 'YYYY-MM-DDTHH:MM:SS[.frac]Z' -> (seconds, nanos)."""
    assert ts.endswith("Z")
    body = ts[:-1]
    frac = "0"
    if "." in body:
        body, frac = body.split(".")
    date, clock = body.split("T")
    y, mo, d = (int(x) for x in date.split("-"))
    hh, mm, ss = (int(x) for x in clock.split(":"))
    secs = calendar.timegm((y, mo, d, hh, mm, ss, 0, 0, 0))
    nanos = int((frac + "000000000")[:9])
    return secs, nanos


def enc_timestamp(secs, nanos):
    out = b""
    if secs:
        out += pb_varint(1, secs & ((1 << 64) - 1))
    if nanos:
        out += pb_varint(2, nanos)
    return out


def enc_block_id(hash32, total, psh_hash32):
    """BlockID / CanonicalBlockID message body.  Empty hash => all-default => empty body."""
    psh = b""
    if total:
        psh += pb_varint(1, total)
    if psh_hash32:
        psh += pb_bytes(2, psh_hash32)
    out = b""
    if hash32:
        out += pb_bytes(1, hash32)
    # gogoproto non-nullable embedded message: always emitted by tendermint-rs' BlockId encoder when id present
    out += pb_bytes(2, psh)
    return out


def header_leaves(h):
    """14 leaf byte strings of a header JSON object (tendermint_utils.rs:374-393)."""
    def hx(s):
        return bytes.fromhex(s) if s else b""

    def wrap_bytes(b):  # google.protobuf.BytesValue; empty => empty encoding
        return pb_bytes(1, b) if b else b""

    ver = b""
    if int(h["version"].get("block", "0")):
        ver += pb_varint(1, int(h["version"]["block"]))
    if int(h["version"].get("app", "0") or 0):
        ver += pb_varint(2, int(h["version"]["app"]))
    secs, nanos = parse_time(h["time"])
    lb = h.get("last_block_id") or {}
    if lb.get("hash"):
        last_block_id = enc_block_id(hx(lb["hash"]), int(lb["parts"]["total"]), hx(lb["parts"]["hash"]))
    else:
        last_block_id = b""
    cid = h["chain_id"].encode()
    return [
        ver,
        pb_bytes(1, cid) if cid else b"",
        pb_varint(1, int(h["height"])) if int(h["height"]) else b"",
        enc_timestamp(secs, nanos),
        last_block_id,
        wrap_bytes(hx(h["last_commit_hash"])),
        wrap_bytes(hx(h["data_hash"])),
        wrap_bytes(hx(h["validators_hash"])),
        wrap_bytes(hx(h["next_validators_hash"])),
        wrap_bytes(hx(h["consensus_hash"])),
        wrap_bytes(hx(h["app_hash"])),
        wrap_bytes(hx(h["last_results_hash"])),
        wrap_bytes(hx(h["evidence_hash"])),
        wrap_bytes(hx(h["proposer_address"])),
    ]


def validator_bytes(pubkey32, power):
    """SimpleValidator: 0a 22 0a 20 pk 10 varint(power)  (power 0 => field omitted)."""
    out = pb_bytes(1, pb_bytes(1, pubkey32))
    if power:
        out += pb_varint(2, power)
    return out


def sign_bytes(chain_id, height, round_, block_id, ts):
    """Length-delimited CanonicalVote for a precommit.  block_id = (hash, total, psh_hash) or None (nil vote)."""
    body = pb_varint(1, 2)  # SIGNED_MSG_TYPE_PRECOMMIT
    if height:
        body += bytes([0x11]) + int(height).to_bytes(8, "little", signed=True)
    if round_:
        body += bytes([0x19]) + int(round_).to_bytes(8, "little", signed=True)
    if block_id is not None:
        body += pb_bytes(4, enc_block_id(*block_id))
    secs, nanos = parse_time(ts)
    body += pb_bytes(5, enc_timestamp(secs, nanos))
    if chain_id:
        body += pb_bytes(6, chain_id.encode())
    return varint(len(body)) + body


# ---------------------------------------------------------------- RFC 6962 Merkle (tendermint_utils.rs:214-372)
def leaf_hash(b):
    return hashlib.sha256(b"\x00" + b).digest()


def inner_hash(l, r):
    return hashlib.sha256(b"\x01" + l + r).digest()


def split_point(n):
    assert n >= 1
    k = 1 << (n.bit_length() - 1)
    return k >> 1 if k == n else k


def root_from_leaf_hashes(hs):
    if not hs:
        return hashlib.sha256(b"").digest()
    if len(hs) == 1:
        return hs[0]
    k = split_point(len(hs))
    return inner_hash(root_from_leaf_hashes(hs[:k]), root_from_leaf_hashes(hs[k:]))


def proofs_from_leaf_hashes(hs):
    """Returns (root, aunts[i]) with aunts ordered leaf->root as in Proof.aunts."""
    n = len(hs)
    if n == 1:
        return hs[0], [[]]
    k = split_point(n)
    lroot, laun = proofs_from_leaf_hashes(hs[:k])
    rroot, raun = proofs_from_leaf_hashes(hs[k:])
    root = inner_hash(lroot, rroot)
    return root, [a + [rroot] for a in laun] + [a + [lroot] for a in raun]


def root_from_proof(leaf_h, index_bits_lsb_first, aunts):
    """compute_hash_from_proof (tendermint_utils.rs:214-224) on an already-hashed leaf."""
    cur = leaf_h
    for bit, aunt in zip(index_bits_lsb_first, aunts):
        cur = inner_hash(aunt, cur) if bit else inner_hash(cur, aunt)
    return cur


def fixed_shape_layers(leaf_hashes, nb_enabled):
    """In-circuit root of the first nb_enabled leaves over a fixed array of N leaf hashes
    (get_root_from_hashed_leaves, called at validator.rs:248-251; plonky2x, absent).  Pairwise layers;
    node = both children enabled ? H(01|L|R) : L ; enabled(node) = enabled(L); an odd last node is
    promoted unchanged.  Returns (layers, root): layers[k] = list of 32-byte nodes of layer k+1."""
    nodes = list(leaf_hashes)
    en = [i < nb_enabled for i in range(len(nodes))]
    layers = []
    while len(nodes) > 1:
        nxt, nen = [], []
        for i in range(0, len(nodes), 2):
            if i + 1 < len(nodes):
                # the circuit hashes every pair and selects; value-wise identical
                h = inner_hash(nodes[i], nodes[i + 1])
                nxt.append(h if (en[i] and en[i + 1]) else nodes[i])
            else:
                nxt.append(nodes[i])
            nen.append(en[i])
        layers.append(nxt)
        nodes, en = nxt, nen
    return layers, nodes[0]


def b64(s):
    return base64.b64decode(s)
