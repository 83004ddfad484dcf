"""Context + array-level entry points over the C ABI."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import KIND_SKIP, KIND_STEP, Config, Report, check


class Context:
    """One tmx_ctx: a HIP stream, device scratch for `max_batch` proofs and the serializer programs for a fixed
    (VALIDATOR_SET_SIZE_MAX, chain id, SKIP_MAX) -- the const generics / TendermintConfig of the reference's
    SkipCircuit<N, CHAIN_ID_SIZE_BYTES, C> (reference circuits/skip.rs:104-111, circuits/config.rs:3-8)."""

    def __init__(self, n_max, chain_id=b"celestia", skip_max=100800, device=0, max_batch=1):
        self._L = _lib.lib()
        self.n_max, self.chain_id, self.skip_max, self.max_batch = int(n_max), bytes(chain_id), int(skip_max), int(max_batch)
        cfg = Config()
        cfg.n_max = self.n_max
        cfg.chain_id_len = len(self.chain_id)
        for i, b in enumerate(self.chain_id[:52]):
            cfg.chain_id[i] = b
        cfg.skip_max = self.skip_max
        cfg.device = device
        cfg.max_batch = self.max_batch
        h = C.c_void_p()
        st = self._L.tmx_ctx_create(C.byref(cfg), C.byref(h))
        self._h = h
        if st != 0:
            try:
                check(st, h if h else None)
            finally:
                if h:
                    self._L.tmx_ctx_destroy(h)
                    self._h = None

    def close(self):
        if getattr(self, "_h", None):
            self._L.tmx_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def elem_count(self, kind):
        return int(self._L.tmx_elem_count(kind, self.n_max))

    def elem_stride(self, kind):
        return int(self._L.tmx_elem_stride(kind, self.n_max))

    def hint_elem_count(self, kind):
        return int(self._L.tmx_hint_elem_count(kind, self.n_max))

    # ---- host-buffer path
    def witness_batch(self, kind, proofs, targets, trusteds=None, want_elems=True, out=None):
        """proofs: bytes (n x 2336); targets: bytes (n x n_max x 256); trusteds: bytes (n x n_max x 48) for skip.
        out: optional preallocated np.uint64 array of n * elem_stride elements (e.g. a view of page-locked memory: the 1.1 GB of a
        256-proof batch come back at PCIe speed instead of through the runtime's pageable staging).
        Returns (np.uint64 [n, elem_count] or None, [report dict])."""
        n = len(proofs) // 2336
        assert len(proofs) == n * 2336 and len(targets) == n * self.n_max * 256
        if kind == KIND_SKIP:
            assert trusteds is not None and len(trusteds) == n * self.n_max * 48
        count, stride = self.elem_count(kind), self.elem_stride(kind)
        if want_elems and out is None:
            out = np.zeros(n * stride, dtype=np.uint64)
        elif want_elems:
            assert out.dtype == np.uint64 and out.size >= n * stride and out.flags["C_CONTIGUOUS"]
            out = out.reshape(-1)[:n * stride]
        else:
            out = None
        reps = (Report * n)()
        st = self._L.tmx_witness_batch(self._h, kind, n, bytes(proofs), bytes(targets), bytes(trusteds) if trusteds else None,
                                       out.ctypes.data if want_elems else None, out.size if want_elems else 0, reps)
        check(st, self._h)
        elems = out.reshape(n, stride)[:, :count] if want_elems else None
        return elems, [r.as_dict() for r in reps]

    def witness_batch_opts(self, kind, proofs, targets, trusteds=None, sections=_lib.SEC_ALL, fmt="u64", out=None):
        """tmx_witness_batch_opts: DENSE rows of the selected sections (_lib.SEC_HINT / SEC_DERIVED / SEC_ALL) as np.uint64 ("u64") or
        np.uint32 ("u32": every element of this witness is < 2^32).  Returns (array [n, row_elems], [report dict])."""
        n = len(proofs) // 2336
        assert len(proofs) == n * 2336 and len(targets) == n * self.n_max * 256
        if kind == KIND_SKIP:
            assert trusteds is not None and len(trusteds) == n * self.n_max * 48
        dt = np.uint32 if fmt == "u32" else np.uint64
        row = int(self._L.tmx_out_row_elems(kind, self.n_max, sections))
        if out is None:
            out = np.zeros(n * row, dtype=dt)
        else:
            assert out.flags["C_CONTIGUOUS"] and out.nbytes >= n * row * np.dtype(dt).itemsize
            out = out.reshape(-1).view(dt)[:n * row]
        reps = (Report * n)()
        st = self._L.tmx_witness_batch_opts(self._h, kind, n, bytes(proofs), bytes(targets), bytes(trusteds) if trusteds else None, sections,
                                            1 if fmt == "u32" else 0, out.ctypes.data, out.nbytes, reps)
        check(st, self._h)
        return out.reshape(n, row), [r.as_dict() for r in reps]

    def witness_batch_hint(self, kind, proofs, targets, trusteds=None, out=None, fmt="u32"):
        """Only the hint section H of every row (what the reference's hint writes to its output stream), narrowed to u32 by default."""
        return self.witness_batch_opts(kind, proofs, targets, trusteds, _lib.SEC_HINT, fmt, out)

    # ---- the typed value of the hint: SkipInputs<F> / StepInputs<F> field by field (include/tmx.h "TYPED VALUE")
    def value_layout(self, kind, sections=_lib.SEC_HINT):
        lay = _lib.ValueLayout()
        check(self._L.tmx_value_layout_of(kind, self.n_max, sections, C.byref(lay)), self._h)
        return lay

    def host_alloc(self, nbytes):
        """Page-locked host memory (tmx_host_alloc) as a np.uint8 array; release with host_free(array)."""
        p = self._L.tmx_host_alloc(self._h, nbytes)
        if not p:
            raise MemoryError(f"tmx_host_alloc({nbytes})")
        a = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(nbytes,))
        self._pinned = getattr(self, "_pinned", {})
        self._pinned[a.ctypes.data] = p
        return a

    def host_free(self, a):
        p = getattr(self, "_pinned", {}).pop(a.ctypes.data, None)
        if p:
            self._L.tmx_host_free(self._h, p)

    def inputs_value_batch(self, kind, proofs, targets, trusteds=None, sections=_lib.SEC_HINT, out=None):
        """tmx_inputs_value_batch: n proofs -> np.uint8 [n, layout.bytes] (one typed value per row) + the layout.  proofs / targets /
        trusteds: bytes, or np.uint8 arrays (e.g. views of host_alloc memory: nothing is copied on the way in)."""
        ptr = lambda b: b.ctypes.data if isinstance(b, np.ndarray) else bytes(b)
        size = lambda b: b.nbytes if isinstance(b, np.ndarray) else len(b)
        n = size(proofs) // 2336
        assert size(proofs) == n * 2336 and size(targets) == n * self.n_max * 256
        if kind == KIND_SKIP:
            assert trusteds is not None and size(trusteds) == n * self.n_max * 48
        lay = self.value_layout(kind, sections)
        if out is None:
            out = np.zeros(n * lay.bytes, dtype=np.uint8)
        else:
            assert out.dtype == np.uint8 and out.flags["C_CONTIGUOUS"] and out.size >= n * lay.bytes
            out = out.reshape(-1)[:n * lay.bytes]
        st = self._L.tmx_inputs_value_batch(self._h, kind, n, ptr(proofs), ptr(targets), ptr(trusteds) if trusteds is not None else None, sections,
                                            out.ctypes.data, out.nbytes)
        check(st, self._h)
        return out.reshape(n, lay.bytes), lay

    def inputs_value_batch_device(self, kind, n_proofs, d_proofs, d_targets, d_trusteds, d_out, sections=_lib.SEC_HINT, stream=None):
        check(self._L.tmx_inputs_value_batch_device(self._h, kind, n_proofs, d_proofs, d_targets, d_trusteds, sections, d_out, self._stream(stream)), self._h)

    def witness_batch_device_sections(self, kind, n_proofs, d_proofs, d_targets, d_trusteds, d_out, d_reports, sections, stream=None):
        check(self._L.tmx_witness_batch_device_sections(self._h, kind, n_proofs, d_proofs, d_targets, d_trusteds, d_out, d_reports,
                                                        self._stream(stream), sections), self._h)

    def trace_elem_count(self, kind):
        return int(self._L.tmx_trace_elem_count(kind, self.n_max))

    def trace_rows_device(self, kind, n_proofs, d_targets, d_trusteds, d_trace_out, sections=_lib.TRACE_ALL, stream=None):
        """Level-2 trace rows of the batch whose Level-1 witness this context computed last (same stream): tmx_trace_rows_device."""
        check(self._L.tmx_trace_rows_device(self._h, kind, n_proofs, d_targets, d_trusteds, d_trace_out, sections, self._stream(stream)), self._h)

    def valid_skip_batch(self, start, n_start, targets, n_targets, sigs, n_sigs):
        """is_valid_skip for len(n_targets) candidates.  start: bytes [n_max x 32]; targets, sigs: bytes [n_cand x n_max x 32].
        Returns (valid [bool], shared power [int], total power [int])."""
        nc = len(n_targets)
        assert len(targets) == nc * self.n_max * 32 == len(sigs)
        nt, ns = (C.c_uint32 * nc)(*n_targets), (C.c_uint32 * nc)(*n_sigs)
        valid, sh, to = (C.c_uint8 * nc)(), (C.c_uint64 * nc)(), (C.c_uint64 * nc)()
        check(self._L.tmx_valid_skip_batch(self._h, nc, bytes(start), n_start, bytes(targets), nt, bytes(sigs), ns, valid, sh, to), self._h)
        return [bool(v) for v in valid], list(sh), list(to)

    def eddsa_lanes(self, lanes):
        n = len(lanes) // 256
        out = np.zeros(n * _lib.ED_STRIDE, dtype=np.uint8)
        check(self._L.tmx_eddsa_lanes(self._h, n, bytes(lanes), out.ctypes.data), self._h)
        return out.reshape(n, _lib.ED_STRIDE)

    # ---- device-resident path (raw device pointers, e.g. torch tensors' data_ptr())
    def _stream(self, stream):
        """stream=None -> the context's own stream; an int (e.g. torch.cuda.current_stream().cuda_stream, 0 = the HIP default
        stream) is used exactly as given."""
        return self._L.tmx_ctx_stream(self._h) if stream is None else (stream or None)

    def witness_batch_device(self, kind, n_proofs, d_proofs, d_targets, d_trusteds, d_out, d_reports, stream=None):
        check(self._L.tmx_witness_batch_device(self._h, kind, n_proofs, d_proofs, d_targets, d_trusteds, d_out, d_reports,
                                               self._stream(stream)), self._h)

    def selftest_fe_invert(self, values):
        """values: ints; returns [(fermat_inverse, safegcd_inverse)] mod 2^255 - 19 as computed on the GPU (self-test hook)."""
        n = len(values)
        inp = np.zeros((n, 8), dtype=np.uint32)
        for i, v in enumerate(values):
            for k in range(8):
                inp[i, k] = (int(v) >> (32 * k)) & 0xFFFFFFFF
        out = np.zeros((n, 16), dtype=np.uint32)
        check(self._L.tmx_selftest_fe_invert(self._h, n, inp.ctypes.data, out.ctypes.data), self._h)
        conv = lambda w: sum(int(w[k]) << (32 * k) for k in range(8))
        return [(conv(out[i, :8]), conv(out[i, 8:])) for i in range(n)]

    def selftest_f16(self, words, doublings):
        """words: uint32 array (n, 128); returns uint32 (n, 256) -- see tmx_selftest_f16 (self-test hook)."""
        inp = np.ascontiguousarray(words, dtype=np.uint32)
        n = inp.shape[0]
        out = np.zeros((n, 256), dtype=np.uint32)
        check(self._L.tmx_selftest_f16(self._h, n, doublings, inp.ctypes.data, out.ctypes.data), self._h)
        return out

    # ---- Goldilocks NTT / coset LDE (device pointers; columns of 2**log_n u64, column c at element c << log_n)
    def ntt_set_domain(self, root_2_32, coset_shift):
        """Domain constants of the NTT / LDE: a primitive 2^32-th root of unity and the coset shift (default: plonky2's, as recalled)."""
        check(self._L.tmx_ntt_set_domain(self._h, C.c_uint64(root_2_32), C.c_uint64(coset_shift)), self._h)

    def ntt_device(self, log_n, n_cols, d_in, d_out, inverse=False, stream=None):
        check(self._L.tmx_ntt_goldilocks_device(self._h, log_n, n_cols, d_in, d_out, 1 if inverse else 0, self._stream(stream)), self._h)

    def lde_device(self, log_n, log_blowup, n_cols, d_in, d_out, stream=None):
        check(self._L.tmx_lde_goldilocks_device(self._h, log_n, log_blowup, n_cols, d_in, d_out, self._stream(stream)), self._h)

    # ---- Poseidon over Goldilocks + Merkle caps (tmx_poseidon_*)
    def poseidon_set_constants(self, round_constants=None, mds_circ=None, mds_diag=None):
        arr = lambda v, n: (C.c_uint64 * n)(*[int(x) for x in v]) if v is not None else None
        check(self._L.tmx_poseidon_set_constants(self._h, arr(round_constants, 360), arr(mds_circ, 12), arr(mds_diag, 12)), self._h)

    def poseidon_permute(self, states):
        a = np.ascontiguousarray(states, dtype=np.uint64).reshape(-1, 12)
        out = np.zeros_like(a)
        check(self._L.tmx_poseidon_permute(self._h, a.shape[0], a.ctypes.data, out.ctypes.data), self._h)
        return out

    def poseidon_merkle_digests(self, log_n, cap_height):
        return int(self._L.tmx_poseidon_merkle_digests(log_n, cap_height))

    def poseidon_merkle_device(self, log_n, n_cols, d_cols, cap_height, d_levels, stream=None):
        check(self._L.tmx_poseidon_merkle_device(self._h, log_n, n_cols, d_cols, cap_height, d_levels, self._stream(stream)), self._h)

    def eddsa_lanes_device(self, n_lanes, d_lanes, d_ed_out, stream=None):
        check(self._L.tmx_eddsa_lanes_device(self._h, n_lanes, d_lanes, d_ed_out, self._stream(stream)), self._h)

    def finish_batch_device(self, kind, n_proofs, d_proofs, d_targets, d_trusteds, d_ed, d_out, d_reports, stream=None):
        check(self._L.tmx_finish_batch_device(self._h, kind, n_proofs, d_proofs, d_targets, d_trusteds, d_ed, d_out, d_reports, self._stream(stream)),
              self._h)

    # ---- the commit pipeline on the device: section rows -> columns -> LDE -> Poseidon Merkle cap
    def trace_commit_shape(self, kind, section):
        lg, w = C.c_uint32(), C.c_uint32()
        check(self._L.tmx_trace_commit_shape(kind, self.n_max, section, C.byref(lg), C.byref(w)), self._h)
        return lg.value, w.value

    def trace_commit_device(self, kind, n_proofs, section, log_blowup, cap_height, d_trace_rows, d_cap, stream=None):
        check(self._L.tmx_trace_commit_device(self._h, kind, n_proofs, section, log_blowup, cap_height, d_trace_rows, d_cap, self._stream(stream)), self._h)

    def trace_commit_last_ms(self):
        ms = (C.c_float * 3)()
        check(self._L.tmx_trace_commit_last_ms(self._h, ms), self._h)
        return {"columns": ms[0], "lde": ms[1], "merkle": ms[2]}

    # ---- multi-GPU: the RCCL exchange behind the C ABI (include/tmx.h "multi-GPU")
    def comm_create(self, unique_id, rank, world):
        check(self._L.tmx_comm_create(self._h, bytes(unique_id) if unique_id is not None else None, rank, world), self._h)

    def comm_destroy(self):
        check(self._L.tmx_comm_destroy(self._h), self._h)

    def comm_abort(self):
        check(self._L.tmx_comm_abort(self._h), self._h)

    def comm_sync(self, stream=None, timeout_ms=0):
        """Bounded wait for `stream` after a sharded call: TmxError(-7) if a peer aborted / died or the timeout passed (include/tmx.h FAILURE CONTRACT)."""
        check(self._L.tmx_comm_sync(self._h, self._stream(stream), int(timeout_ms)), self._h)

    def comm_info(self):
        r, w = C.c_uint32(), C.c_uint32()
        check(self._L.tmx_comm_info(self._h, C.byref(r), C.byref(w)), self._h)
        return r.value, w.value

    def witness_batch_sharded_device(self, kind, n_total, d_proofs, d_targets, d_trusteds, d_out, d_reports, gather=False, stream=None):
        check(self._L.tmx_witness_batch_sharded_device(self._h, kind, n_total, d_proofs, d_targets, d_trusteds, d_out, d_reports,
                                                       1 if gather else 0, self._stream(stream)), self._h)

    def witness_validator_sharded_device(self, kind, n_proofs, d_proofs, d_targets, d_trusteds, d_out, d_reports, stream=None):
        check(self._L.tmx_witness_validator_sharded_device(self._h, kind, n_proofs, d_proofs, d_targets, d_trusteds, d_out, d_reports,
                                                           self._stream(stream)), self._h)

    def trace_rows_sharded_device(self, kind, n_total, d_targets, d_trusteds, d_trace_out, sections=_lib.TRACE_ALL, gather=False, stream=None):
        """Level-2 rows of this rank's proofs of a proof-sharded batch (after witness_batch_sharded_device of the same n_total); gather: all rows everywhere."""
        check(self._L.tmx_trace_rows_sharded_device(self._h, kind, n_total, d_targets, d_trusteds, d_trace_out, sections, 1 if gather else 0,
                                                    self._stream(stream)), self._h)

    def trace_rows_validator_sharded_device(self, kind, n_proofs, d_targets, d_trusteds, d_trace_out, sections=_lib.TRACE_ALL, stream=None):
        """Level-2 rows with the per-lane sections (ladders, SHA-512) lane-sharded and exchanged (after witness_validator_sharded_device)."""
        check(self._L.tmx_trace_rows_validator_sharded_device(self._h, kind, n_proofs, d_targets, d_trusteds, d_trace_out, sections, self._stream(stream)), self._h)

    def trace_commit_sharded_device(self, kind, n_total, section, log_blowup, cap_height, d_trace_rows, d_caps, stream=None):
        check(self._L.tmx_trace_commit_sharded_device(self._h, kind, n_total, section, log_blowup, cap_height, d_trace_rows, d_caps, self._stream(stream)), self._h)

    def last_kernel_ms(self):
        ms = (C.c_float * _lib.N_KERNELS)()
        check(self._L.tmx_last_kernel_ms(self._h, ms), self._h)
        return dict(zip(_lib.KERNEL_NAMES, (float(x) for x in ms)))

    def kernel_ms_mean(self, last_k):
        """Mean HIP-event duration per kernel over the last `last_k` enqueued batches (blocks until they finished)."""
        ms = (C.c_float * _lib.N_KERNELS)()
        check(self._L.tmx_kernel_ms_mean(self._h, last_k, ms), self._h)
        return dict(zip(_lib.KERNEL_NAMES, (float(x) for x in ms)))

    def last_dedup(self):
        """(distinct public keys, table path used) of the last EdDSA launch."""
        u, t = C.c_uint32(), C.c_uint32()
        check(self._L.tmx_last_dedup(self._h, C.byref(u), C.byref(t)), self._h)
        return int(u.value), bool(t.value)

    # ---- persistent per-key table cache (tmx_key_cache_*)
    def key_cache_stats(self):
        info = _lib.KeyCacheInfo()
        check(self._L.tmx_key_cache_stats(self._h, C.byref(info)), self._h)
        return info.as_dict()

    def set_cache_stats(self):
        """The validator-set cache (tmx_set_cache_stats): sets resident / served from the cache / computed / inserted / evicted (LRU), capacity."""
        out = (C.c_uint32 * 8)()
        check(self._L.tmx_set_cache_stats(self._h, out), self._h)
        return dict(zip(("resident", "served", "computed", "inserted", "evicted", "capacity"), (int(x) for x in out)))

    def key_cache_flush(self):
        check(self._L.tmx_key_cache_flush(self._h), self._h)

    def key_cache_config(self, enabled=True, max_keys=0):
        """enabled=False: tables do not survive a call (every call cold); max_keys != 0: new capacity (flushes)."""
        check(self._L.tmx_key_cache_config(self._h, 1 if enabled else 0, int(max_keys)), self._h)

    def sync(self):
        check(self._L.tmx_sync(self._h), self._h)
