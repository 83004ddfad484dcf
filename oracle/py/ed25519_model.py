"""Pure-Python big-int model of the Ed25519 arithmetic on the skip/step hot path.

TEST INFRASTRUCTURE ONLY (oracle).  Nothing in the product path imports this file.

The reference never implements this arithmetic itself: it calls
`curta_eddsa_verify_sigs_conditional` (reference circuits/builder/verify.rs:248-259), whose
implementation lives in the un-vendored dependency plonky2x @ succinctx v1.0.3 /
starkyx (Cargo.lock:3017-3019, 4037-4039), and the host pre-check
`Verifier::verify` (reference circuits/input/conversion.rs:48-49) lives in ed25519-consensus 2.1.0
(Cargo.lock:1108).  This file restates the *published* algorithm (RFC 8032 §5.1) that both follow:
    h = SHA512(R || A || M) mod l ;  accept iff  s*B == R + h*A   (cofactor-less, affine compare)
It is slow (Python ints) and is used to pin the C restatement in oracle/c/ and to make goldens.
"""
import hashlib

P = 2**255 - 19
L = 2**252 + 27742317777372353535851937790883648493
D = (-121665 * pow(121666, P - 2, P)) % P
SQRT_M1 = pow(2, (P - 1) // 4, P)


def _inv(x):
    return pow(x, P - 2, P)


BY = (4 * _inv(5)) % P


def recover_x(y, sign):
    """RFC 8032 §5.1.3 decoding, returns x or None.  y is reduced mod p first (non-canonical y accepted)."""
    y %= P
    u = (y * y - 1) % P
    v = (D * y * y + 1) % P
    # candidate root x = u v^3 (u v^7)^((p-5)/8)
    x = (u * pow(v, 3, P) * pow(u * pow(v, 7, P) % P, (P - 5) // 8, P)) % P
    vxx = (v * x * x) % P
    if vxx == u:
        pass
    elif vxx == (-u) % P:
        x = (x * SQRT_M1) % P
    else:
        return None
    if x == 0 and sign == 1:
        return None
    if (x & 1) != sign:
        x = P - x
    return x


BX = recover_x(BY, 0)
B = (BX, BY)
IDENT = (0, 1)


def decompress(b32):
    """32 little-endian bytes -> affine (x, y) or None."""
    v = int.from_bytes(b32, "little")
    sign = v >> 255
    y = v & ((1 << 255) - 1)
    x = recover_x(y, sign)
    if x is None:
        return None
    return (x, y % P)


def compress(pt):
    x, y = pt
    return (y | ((x & 1) << 255)).to_bytes(32, "little")


def add(p, q):
    """Complete affine twisted-Edwards addition (a = -1)."""
    x1, y1 = p
    x2, y2 = q
    t = D * x1 * x2 * y1 * y2 % P
    x3 = (x1 * y2 + x2 * y1) * _inv(1 + t) % P
    y3 = (y1 * y2 + x1 * x2) * _inv(1 - t) % P
    return (x3, y3)


def _ext_add(p, q):
    (X1, Y1, Z1, T1), (X2, Y2, Z2, T2) = p, q
    A = (Y1 - X1) * (Y2 - X2) % P
    Bv = (Y1 + X1) * (Y2 + X2) % P
    C = 2 * D * T1 * T2 % P
    Dv = 2 * Z1 * Z2 % P
    E, F, G, H = Bv - A, Dv - C, Dv + C, Bv + A
    return (E * F % P, G * H % P, F * G % P, E * H % P)


def scalarmult(k, pt):
    """k * pt by plain MSB-first double-and-add in extended coordinates; returns affine."""
    x, y = pt
    q = (x, y, 1, x * y % P)
    r = (0, 1, 1, 0)
    for i in reversed(range(k.bit_length())):
        r = _ext_add(r, r)
        if (k >> i) & 1:
            r = _ext_add(r, q)
    zi = _inv(r[2])
    return (r[0] * zi % P, r[1] * zi % P)


def hram(r32, a32, msg):
    dig = hashlib.sha512(r32 + a32 + msg).digest()
    return dig, int.from_bytes(dig, "little") % L


def verify_trace(pk32, sig64, msg):
    """Returns the Level-1 EdDSA values for one lane (all canonical):
    dict(digest, h, A, R, sB, hA, sum, ok).  Points are affine (x, y) ints; None on decode failure."""
    r32, s32 = sig64[:32], sig64[32:]
    s = int.from_bytes(s32, "little")
    dig, h = hram(r32, pk32, msg)
    A = decompress(pk32)
    R = decompress(r32)
    out = dict(digest=dig, h=h, s=s, A=A, R=R, sB=None, hA=None, sum=None, ok=False)
    if A is None or R is None:
        # one convention for the whole lane (oracle/c tmxo_eddsa_trace_lane, the kernels): if either point does not decode, NO point of
        # the lane is reported (all zero), not even the one that did decode.  (Found by the C-vs-model fuzz, round 2: this model used to
        # keep the decodable one.)
        out.update(A=None, R=None)
        return out
    sB = scalarmult(s, B)
    hA = scalarmult(h, A)
    sm = add(R, hA)
    out.update(sB=sB, hA=hA, sum=sm, ok=(sm == sB and s < L))
    return out


def keypair_from_seed(seed32):
    hsh = hashlib.sha512(seed32).digest()
    a = int.from_bytes(hsh[:32], "little")
    a &= (1 << 254) - 8
    a |= 1 << 254
    pk = compress(scalarmult(a, B))
    return a, hsh[32:], pk


def sign(seed32, msg):
    """RFC 8032 §5.1.6 deterministic signature."""
    a, prefix, pk = keypair_from_seed(seed32)
    r = int.from_bytes(hashlib.sha512(prefix + msg).digest(), "little") % L
    R = compress(scalarmult(r, B))
    _, h = hram(R, pk, msg)
    S = (r + h * a) % L
    return R + S.to_bytes(32, "little")


# Dummy lane constants.  The reference imports DUMMY_PUBLIC_KEY / DUMMY_SIGNATURE from plonky2x
# (conversion.rs:3-5, used at :100-101, :119-122, :168); their byte values are not in the repo.
# they are the RFC 8032 keypair of seed 01x32 signing the 32-byte zero message
# (consistent with message_byte_length = 32 for dummy lanes, conversion.rs:108-109).  Regenerated here.
DUMMY_SEED = bytes([1] * 32)
DUMMY_MSG = bytes(32)
DUMMY_MSG_LENGTH = 32
_, _, DUMMY_PUBLIC_KEY = keypair_from_seed(DUMMY_SEED)
DUMMY_SIGNATURE = sign(DUMMY_SEED, DUMMY_MSG)
