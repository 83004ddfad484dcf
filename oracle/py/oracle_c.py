"""ctypes binding of the C oracle (oracle/c/libtmx_oracle.so).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_CDIR = os.path.join(os.path.dirname(_HERE), "c")
_LIB = None


class Report(C.Structure):
    _fields_ = [("header", C.c_uint8 * 32), ("all_ok", C.c_uint32), ("fail_mask", C.c_uint32),
                ("first_bad_sig", C.c_int32), ("gt_target", C.c_uint32), ("gt_trusted", C.c_uint32),
                ("dist_ok", C.c_uint32), ("reserved", C.c_uint32 * 2)]


class EddsaTrace(C.Structure):
    _fields_ = [("digest", C.c_uint8 * 64), ("h", C.c_uint8 * 32), ("pt", (C.c_uint8 * 32) * 10),
                ("ok", C.c_uint32), ("decode_ok", C.c_uint32)]


def build():
    subprocess.check_call(["make", "-s", "-C", _CDIR])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_CDIR, "libtmx_oracle.so")
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
        _LIB.tmxo_elem_count.restype = C.c_size_t
        _LIB.tmxo_elem_count.argtypes = [C.c_int, C.c_size_t]
        _LIB.tmxo_tree_nodes.restype = C.c_size_t
        _LIB.tmxo_tree_nodes.argtypes = [C.c_size_t]
    return _LIB


def sha256(b):
    out = (C.c_uint8 * 32)()
    lib().tmxo_sha256(bytes(b), C.c_size_t(len(b)), out)
    return bytes(out)


def sha512(b):
    out = (C.c_uint8 * 64)()
    lib().tmxo_sha512(bytes(b), C.c_size_t(len(b)), out)
    return bytes(out)


def eddsa_trace(pk, sig, msg):
    tr = EddsaTrace()
    lib().tmxo_eddsa_trace_lane(bytes(pk), bytes(sig), bytes(msg), C.c_size_t(len(msg)), C.byref(tr))
    return dict(digest=bytes(tr.digest), h=bytes(tr.h), pt=[bytes(tr.pt[k]) for k in range(10)], ok=bool(tr.ok),
                decode_ok=bool(tr.decode_ok))


def pubkey(seed):
    out = (C.c_uint8 * 32)()
    lib().tmxo_ed25519_pubkey(bytes(seed), out)
    return bytes(out)


def sign(seed, msg):
    out = (C.c_uint8 * 64)()
    lib().tmxo_ed25519_sign(bytes(seed), bytes(msg), C.c_size_t(len(msg)), out)
    return bytes(out)


def dummy():
    pk, sig = (C.c_uint8 * 32)(), (C.c_uint8 * 64)()
    lib().tmxo_dummy(pk, sig)
    return bytes(pk), bytes(sig)


def sc_reduce512(b):
    out = (C.c_uint8 * 32)()
    lib().tmxo_sc_reduce512(bytes(b), out)
    return bytes(out)


def varint9(v):
    out = (C.c_uint8 * 9)()
    lib().tmxo_varint9(C.c_uint64(v), out)
    return bytes(out)


def marshal_validator(pk, power):
    out = (C.c_uint8 * 46)()
    lib().tmxo_marshal_validator(bytes(pk), C.c_uint64(power), out)
    return bytes(out)


def rfc6962_root(leaf_hashes):
    out = (C.c_uint8 * 32)()
    lib().tmxo_rfc6962_root(b"".join(leaf_hashes), C.c_size_t(len(leaf_hashes)), out)
    return bytes(out)


def fixed_shape_tree(leaf_hashes, nb):
    n = len(leaf_hashes)
    tn = lib().tmxo_tree_nodes(n)
    nodes = (C.c_uint8 * (32 * max(tn, 1)))()
    root = (C.c_uint8 * 32)()
    lib().tmxo_fixed_shape_tree(b"".join(leaf_hashes), C.c_size_t(n), C.c_size_t(nb), nodes, root)
    return bytes(nodes)[:32 * tn], bytes(root)


def tally(powers, nb, in_group, num, den):
    n = len(powers)
    pw = (C.c_uint64 * n)(*powers)
    ig = (C.c_uint8 * n)(*[1 if x else 0 for x in in_group])
    tp, ap, sc = (C.c_uint64 * n)(), (C.c_uint64 * n)(), (C.c_uint64 * 4)()
    no = C.c_int(1)
    gt = lib().tmxo_tally(pw, C.c_size_t(n), C.c_size_t(nb), ig, C.c_uint64(num), C.c_uint64(den), tp, ap, sc, C.byref(no))
    return dict(gt=bool(gt), tot_prefix=list(tp), acc_prefix=list(ap), total=sc[0], acc=sc[1], scaled_acc=sc[2],
                scaled_total=sc[3], no_overflow=bool(no.value))


def is_valid_skip(start, target, sigs):
    """start/target/sigs: bytes of 32-byte address records.  Returns (valid, shared_at_exit, total)."""
    sh, to = C.c_uint64(), C.c_uint64()
    v = lib().tmxo_is_valid_skip(bytes(start), C.c_uint32(len(start) // 32), bytes(target), C.c_uint32(len(target) // 32), bytes(sigs),
                                 C.c_uint32(len(sigs) // 32), C.byref(sh), C.byref(to))
    return bool(v), sh.value, to.value


def elem_count(kind, n):
    return lib().tmxo_elem_count(kind, n)


def _rep(r):
    return dict(header=bytes(r.header), all_ok=bool(r.all_ok), fail_mask=r.fail_mask, first_bad_sig=r.first_bad_sig,
                gt_target=bool(r.gt_target), gt_trusted=bool(r.gt_trusted), dist_ok=bool(r.dist_ok), precond=int(r.reserved[0]))


def witness(kind, proof_rec, target_recs, trusted_recs, chain_id, skip_max):
    """One proof.  *_recs: bytes (concatenated records).  Returns (np.uint64 array, report dict)."""
    n = len(target_recs) // 256
    out = np.zeros(elem_count(kind, n), dtype=np.uint64)
    rep = Report()
    rc = lib().tmxo_witness(kind, bytes(proof_rec), bytes(target_recs), bytes(trusted_recs) if trusted_recs else None,
                            C.c_uint32(n), bytes(chain_id), C.c_uint32(len(chain_id)), C.c_uint64(skip_max),
                            out.ctypes.data_as(C.POINTER(C.c_uint64)), C.byref(rep))
    if rc:
        raise RuntimeError(f"tmxo_witness rc={rc}")
    return out, _rep(rep)


def value_bytes(kind, n, with_derived):
    L = lib()
    L.tmxo_value_bytes.restype = C.c_size_t
    return int(L.tmxo_value_bytes(kind, C.c_size_t(n), 1 if with_derived else 0))


def witness_value(kind, proof_rec, target_recs, trusted_recs, chain_id, skip_max, with_derived=False):
    """One proof: the typed value of the hint (tmxo.h tmxo_*_value structs) as a np.uint8 array, and the report."""
    n = len(target_recs) // 256
    val = np.zeros(value_bytes(kind, n, with_derived) // 8, dtype=np.uint64)   # (8-byte aligned: the C side writes through struct pointers)
    rep = Report()
    rc = lib().tmxo_witness_value(kind, bytes(proof_rec), bytes(target_recs), bytes(trusted_recs) if trusted_recs else None,
                                  C.c_uint32(n), bytes(chain_id), C.c_uint32(len(chain_id)), C.c_uint64(skip_max), None, C.byref(rep),
                                  val.ctypes.data_as(C.c_void_p), 1 if with_derived else 0)
    if rc:
        raise RuntimeError(f"tmxo_witness_value rc={rc}")
    return val.view(np.uint8), _rep(rep)


def witness_batch(kind, n_proofs, proof_recs, target_recs, trusted_recs, n, chain_id, skip_max, n_threads=1, want_out=True):
    ec = elem_count(kind, n)
    out = np.zeros(ec * n_proofs, dtype=np.uint64) if want_out else None
    reps = (Report * n_proofs)()
    rc = lib().tmxo_witness_batch(kind, C.c_uint32(n_proofs), bytes(proof_recs), bytes(target_recs),
                                  bytes(trusted_recs) if trusted_recs else None, C.c_uint32(n), bytes(chain_id),
                                  C.c_uint32(len(chain_id)), C.c_uint64(skip_max),
                                  out.ctypes.data_as(C.POINTER(C.c_uint64)) if want_out else None, reps, C.c_uint32(n_threads))
    if rc:
        raise RuntimeError(f"tmxo_witness_batch rc={rc}")
    return (out.reshape(n_proofs, ec) if want_out else None), [_rep(r) for r in reps]


def witness_pool_seconds(kind, n_proofs, proof_recs, target_recs, trusted_recs, n, chain_id, skip_max, repeat, n_threads):
    """CPU baseline: persistent pthread pool, proofs dealt round-robin, compute only; returns wall seconds for n_proofs * repeat proofs."""
    L = lib()
    L.tmxo_witness_pool_seconds.restype = C.c_double
    t = L.tmxo_witness_pool_seconds(kind, C.c_uint32(n_proofs), bytes(proof_recs), bytes(target_recs), bytes(trusted_recs) if trusted_recs else None,
                                    C.c_uint32(n), bytes(chain_id), C.c_uint32(len(chain_id)), C.c_uint64(skip_max), C.c_uint32(repeat),
                                    C.c_uint32(n_threads))
    if t < 0:
        raise RuntimeError("tmxo_witness_pool_seconds failed")
    return float(t)


# ---- Level-2 trace rows (oracle/c/tmxo_trace.c)
def trace_elem_count(kind, n):
    L = lib()
    L.tmxo_trace_elem_count.restype = C.c_size_t
    L.tmxo_trace_elem_count.argtypes = [C.c_int, C.c_size_t]
    return int(L.tmxo_trace_elem_count(kind, n))


def trace(kind, proof_rec, target_recs, trusted_recs, n):
    """Level-2 trace block of one proof, generated with per-operation affine arithmetic (slow: ~20 ms per lane)."""
    out = np.zeros(trace_elem_count(kind, n), dtype=np.uint64)
    rc = lib().tmxo_trace(kind, bytes(proof_rec), bytes(target_recs), bytes(trusted_recs) if trusted_recs else None, C.c_uint32(n),
                          out.ctypes.data_as(C.POINTER(C.c_uint64)))
    if rc:
        raise RuntimeError(f"tmxo_trace rc={rc}")
    return out


def trace_check(kind, proof_rec, target_recs, trusted_recs, n, trace_rows):
    """Constraint checker: 0 if every row satisfies its recurrence and connects to the inputs / Level-1 values, else an error code
    (section * 10^9 + lane * 10^6 + detail)."""
    L = lib()
    L.tmxo_trace_check.restype = C.c_longlong
    a = np.ascontiguousarray(trace_rows, dtype=np.uint64)
    assert a.size == trace_elem_count(kind, n)
    return int(L.tmxo_trace_check(kind, bytes(proof_rec), bytes(target_recs), bytes(trusted_recs) if trusted_recs else None, C.c_uint32(n),
                                  a.ctypes.data_as(C.POINTER(C.c_uint64))))


# ---- Goldilocks NTT / coset LDE (oracle/c/tmxo_ntt.c)
GL_P = 2**64 - 2**32 + 1


PLONKY2_DOMAIN = (7277203076849721926, 14293326489335486720)   # recalled POWER_OF_TWO_GENERATOR, MULTIPLICATIVE_GROUP_GENERATOR
G7_DOMAIN = (0x185629DCDA58878C, 7)                             # g = 7 (Plonky3 / winterfell)


def ntt_set_domain(root_2_32, coset_shift):
    lib().tmxo_ntt_set_domain(C.c_uint64(root_2_32), C.c_uint64(coset_shift))


def gl_root(log_n):
    L = lib()
    L.tmxo_gl_root.restype = C.c_uint64
    L.tmxo_gl_root.argtypes = [C.c_uint32]
    return int(L.tmxo_gl_root(log_n))


def ntt(values, inverse=False):
    """values: 1-D np.uint64 of power-of-two length (one column) or 2-D [cols, n]; returns a new array, natural order."""
    a = np.ascontiguousarray(values, dtype=np.uint64).copy()
    cols = a.reshape(1, -1) if a.ndim == 1 else a
    n = cols.shape[1]
    log_n = n.bit_length() - 1
    assert 1 << log_n == n
    L = lib()
    L.tmxo_ntt.argtypes = [C.c_void_p, C.c_uint32, C.c_int]
    L.tmxo_ntt.restype = None
    for c in cols:
        L.tmxo_ntt(c.ctypes.data, log_n, 1 if inverse else 0)
    return a


def lde(values, log_blowup):
    a = np.ascontiguousarray(values, dtype=np.uint64)
    cols = a.reshape(1, -1) if a.ndim == 1 else a
    n = cols.shape[1]
    log_n = n.bit_length() - 1
    assert 1 << log_n == n
    out = np.zeros((cols.shape[0], n << log_blowup), dtype=np.uint64)
    L = lib()
    L.tmxo_lde.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32]
    L.tmxo_lde.restype = None
    for i, c in enumerate(cols):
        c = np.ascontiguousarray(c)
        L.tmxo_lde(c.ctypes.data, out[i].ctypes.data, log_n, log_blowup)
    return out[0] if a.ndim == 1 else out


# ---- Poseidon over Goldilocks + Merkle caps (oracle/c/tmxo_poseidon.c)
def poseidon_constants():
    rc, circ, diag = (C.c_uint64 * 360)(), (C.c_uint64 * 12)(), (C.c_uint64 * 12)()
    lib().tmxo_poseidon_get_constants(rc, circ, diag)
    return list(rc), list(circ), list(diag)


def poseidon_set_constants(rc=None, circ=None, diag=None):
    lib().tmxo_poseidon_set_constants((C.c_uint64 * 360)(*rc) if rc else None, (C.c_uint64 * 12)(*circ) if circ else None,
                                      (C.c_uint64 * 12)(*diag) if diag else None)


def poseidon_permute(states):
    a = np.ascontiguousarray(states, dtype=np.uint64).reshape(-1, 12).copy()
    L = lib()
    L.tmxo_poseidon_permute.argtypes = [C.c_void_p]
    for i in range(a.shape[0]):
        L.tmxo_poseidon_permute(a[i].ctypes.data)
    return a


def poseidon_merkle(cols, log_n, n_cols, cap_height):
    """cols: uint64 [n_cols << log_n] column-major.  Returns the levels (leaves .. cap) as one uint64 array of 4-element digests."""
    a = np.ascontiguousarray(cols, dtype=np.uint64)
    total = sum(1 << (log_n - k) for k in range(log_n - cap_height + 1))
    out = np.zeros(4 * total, dtype=np.uint64)
    L = lib()
    L.tmxo_poseidon_merkle.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
    rc = L.tmxo_poseidon_merkle(a.ctypes.data, log_n, n_cols, cap_height, out.ctypes.data)
    if rc:
        raise RuntimeError(f"tmxo_poseidon_merkle rc={rc}")
    return out.reshape(-1, 4)
