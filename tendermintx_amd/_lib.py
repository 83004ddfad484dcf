"""ctypes binding of libtmx.so (include/tmx.h).  Fails loudly when the library or a HIP runtime is missing:
there is no CPU fallback in this package."""
import ctypes as C
import ctypes.util
import importlib.util
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TMX_LIB") or os.path.join(_HERE, "libtmx.so")

KIND_SKIP, KIND_STEP = 0, 1
FLAG_SIGNED, FLAG_PRESENT = 1, 2
N_KERNELS = 4
KERNEL_NAMES = ("k_eddsa", "k_proof", "k_verdict", "k_serialize")
ED_STRIDE = 448
SEC_HINT, SEC_DERIVED, SEC_ALL = 1, 2, 3   # TMX_SEC_*
TRACE_LADDERS, TRACE_SHA512, TRACE_SHA256, TRACE_MATCH, TRACE_TREE, TRACE_HEADER, TRACE_ALL = 1, 2, 4, 8, 16, 32, 63   # TMX_TRACE_*


class ValidatorRec(C.Structure):
    _fields_ = [("pubkey", C.c_uint8 * 32), ("signature", C.c_uint8 * 64), ("message", C.c_uint8 * 124),
                ("message_byte_length", C.c_uint16), ("validator_byte_length", C.c_uint8), ("flags", C.c_uint8),
                ("voting_power", C.c_uint64), ("pad", C.c_uint8 * 24)]


class HashFieldRec(C.Structure):
    _fields_ = [("pubkey", C.c_uint8 * 32), ("voting_power", C.c_uint64), ("validator_byte_length", C.c_uint8),
                ("flags", C.c_uint8), ("pad", C.c_uint8 * 6)]


class HeaderRec(C.Structure):
    _fields_ = [("leaf_len", C.c_uint8 * 14), ("pad", C.c_uint8 * 2), ("leaf", (C.c_uint8 * 80) * 14)]


class ProofRec(C.Structure):
    _fields_ = [("block_a", C.c_uint64), ("block_b", C.c_uint64), ("hash", C.c_uint8 * 32), ("round", C.c_uint64),
                ("nb_a", C.c_uint32), ("nb_b", C.c_uint32), ("header_a", HeaderRec), ("header_b", HeaderRec)]


class Report(C.Structure):
    _fields_ = [("header", C.c_uint8 * 32), ("all_ok", C.c_uint32), ("fail_mask", C.c_uint32),
                ("first_bad_sig", C.c_int32), ("gt_target", C.c_uint32), ("gt_trusted", C.c_uint32),
                ("dist_ok", C.c_uint32), ("precond", C.c_uint32), ("reserved", C.c_uint32)]

    def as_dict(self):
        return dict(header=bytes(self.header), all_ok=bool(self.all_ok), fail_mask=self.fail_mask,
                    first_bad_sig=self.first_bad_sig, gt_target=bool(self.gt_target),
                    gt_trusted=bool(self.gt_trusted), dist_ok=bool(self.dist_ok), precond=int(self.precond))


class KeyCacheInfo(C.Structure):
    _fields_ = [("capacity_keys", C.c_uint32), ("resident_keys", C.c_uint32), ("enabled", C.c_uint32), ("epoch", C.c_uint32),
                ("bytes_per_key", C.c_uint64), ("last_new_keys", C.c_uint32), ("last_hit_keys", C.c_uint32),
                ("last_hit_lanes", C.c_uint32), ("last_built_keys", C.c_uint32), ("hit_lanes", C.c_uint64), ("miss_lanes", C.c_uint64),
                ("built_keys", C.c_uint64), ("evicted_keys", C.c_uint64), ("evictions", C.c_uint64), ("launches", C.c_uint64)]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


class AddrRec(C.Structure):
    _fields_ = [("address", C.c_uint8 * 20), ("has_address", C.c_uint8), ("pad", C.c_uint8 * 3), ("voting_power", C.c_uint64)]


class Config(C.Structure):
    _fields_ = [("n_max", C.c_uint32), ("chain_id_len", C.c_uint32), ("chain_id", C.c_uint8 * 52),
                ("skip_max", C.c_uint64), ("device", C.c_int32), ("max_batch", C.c_uint32)]


# ---- the typed value of the hint (include/tmx.h "TYPED VALUE"): SkipInputs<F> / StepInputs<F> field by field
class ValidatorValue(C.Structure):
    _fields_ = [("pubkey", C.c_uint8 * 32), ("sig_r", C.c_uint8 * 32), ("sig_s", C.c_uint8 * 32), ("message", C.c_uint8 * 124),
                ("message_byte_length", C.c_uint32), ("voting_power", C.c_uint64), ("validator_byte_length", C.c_uint32), ("signed_", C.c_uint32)]


class HashFieldValue(C.Structure):
    _fields_ = [("pubkey", C.c_uint8 * 32), ("voting_power", C.c_uint64), ("validator_byte_length", C.c_uint32), ("pad", C.c_uint32)]


class ChainIdProofValue(C.Structure):
    _fields_ = [("proof", (C.c_uint8 * 32) * 4), ("enc_chain_id_byte_length", C.c_uint32), ("chain_id", C.c_uint8 * 52), ("pad", C.c_uint8 * 8)]


class HeightProofValue(C.Structure):
    _fields_ = [("proof", (C.c_uint8 * 32) * 4), ("enc_height_byte_length", C.c_uint32), ("pad", C.c_uint32), ("height", C.c_uint64)]


class HashInclusionProofValue(C.Structure):
    _fields_ = [("proof", (C.c_uint8 * 32) * 4), ("leaf", C.c_uint8 * 34), ("pad", C.c_uint8 * 14)]


class BlockIdInclusionProofValue(C.Structure):
    _fields_ = [("proof", (C.c_uint8 * 32) * 4), ("leaf", C.c_uint8 * 72), ("pad", C.c_uint8 * 8)]


class SkipInputsFixed(C.Structure):
    _fields_ = [("target_header", C.c_uint8 * 32), ("trusted_header", C.c_uint8 * 32), ("round", C.c_uint64),
                ("nb_target_validators", C.c_uint32), ("nb_trusted_validators", C.c_uint32),
                ("target_block_chain_id_proof", ChainIdProofValue), ("target_block_height_proof", HeightProofValue),
                ("target_block_validators_hash_proof", HashInclusionProofValue), ("trusted_block_validators_hash_proof", HashInclusionProofValue),
                ("report", Report)]


class StepInputsFixed(C.Structure):
    _fields_ = [("next_header", C.c_uint8 * 32), ("round", C.c_uint64), ("nb_validators", C.c_uint32), ("pad", C.c_uint32),
                ("next_block_chain_id_proof", ChainIdProofValue), ("next_block_height_proof", HeightProofValue),
                ("next_block_validators_hash_proof", HashInclusionProofValue), ("next_block_last_block_id_proof", BlockIdInclusionProofValue),
                ("prev_block_next_validators_hash_proof", HashInclusionProofValue), ("report", Report)]


class TargetLaneDerived(C.Structure):
    _fields_ = [("sha512_digest", C.c_uint8 * 64), ("h", C.c_uint8 * 32), ("points", (C.c_uint8 * 32) * 10), ("eddsa_ok", C.c_uint32),
                ("decode_ok", C.c_uint32), ("pad0", C.c_uint8 * 24), ("marshalled", C.c_uint8 * 46), ("pad1", C.c_uint8 * 2),
                ("leaf_hash", C.c_uint8 * 32), ("flags", C.c_uint8 * 6), ("pad2", C.c_uint8 * 2), ("total_prefix", C.c_uint64),
                ("signed_prefix", C.c_uint64), ("pad3", C.c_uint8 * 8)]


class TrustedLaneDerived(C.Structure):
    _fields_ = [("marshalled", C.c_uint8 * 46), ("pad1", C.c_uint8 * 2), ("leaf_hash", C.c_uint8 * 32), ("flags", C.c_uint8 * 2),
                ("pad2", C.c_uint8 * 6), ("total_prefix", C.c_uint64), ("matched_prefix", C.c_uint64), ("pad3", C.c_uint8 * 8)]


class ProofDerived(C.Structure):
    _fields_ = [("proofs", ((C.c_uint8 * 32) * 5) * 5), ("height_leaf", C.c_uint8 * 11), ("pad0", C.c_uint8 * 5), ("tally_target", C.c_uint64 * 4),
                ("tally_trusted", C.c_uint64 * 4), ("verdicts", C.c_uint32 * 4), ("checks", C.c_uint32 * 16), ("all_ok", C.c_uint32),
                ("pad1", C.c_uint32), ("height", C.c_uint64)]


class ValueLayout(C.Structure):
    _fields_ = [("bytes", C.c_uint64), ("fixed_bytes", C.c_uint32), ("off_validators", C.c_uint32), ("off_hashfields", C.c_uint32),
                ("off_target_lanes", C.c_uint32), ("off_trusted_lanes", C.c_uint32), ("off_nodes_target", C.c_uint32),
                ("off_nodes_trusted", C.c_uint32), ("off_proof_derived", C.c_uint32), ("tree_nodes", C.c_uint32), ("reserved", C.c_uint32)]


assert (C.sizeof(ValidatorValue), C.sizeof(HashFieldValue), C.sizeof(SkipInputsFixed), C.sizeof(StepInputsFixed)) == (240, 48, 832, 1008)
assert (C.sizeof(TargetLaneDerived), C.sizeof(TrustedLaneDerived), C.sizeof(ProofDerived)) == (560, 112, 976)
assert C.sizeof(ValidatorRec) == 256 and C.sizeof(HashFieldRec) == 48 and C.sizeof(ProofRec) == 2336
assert C.sizeof(Report) == 64 and C.sizeof(AddrRec) == 32

_lib = None
_hip = None


class TmxError(RuntimeError):
    def __init__(self, status, msg=""):
        self.status = status
        super().__init__(f"libtmx status {status}: {msg}")


def _hip_runtime_candidates():
    # One HIP runtime per process: if PyTorch is (or will be) in this process, libtmx must run on PyTorch's bundled
    # libamdhip64 so that device pointers, streams and events are shared.
    cands = []
    spec = importlib.util.find_spec("torch") if "torch" in sys.modules or os.environ.get("TMX_HIP_FROM_TORCH", "1") == "1" else None
    if spec and spec.origin:
        cands.append(os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so"))
    cands.append("/opt/rocm/lib/libamdhip64.so")
    found = ctypes.util.find_library("amdhip64")
    if found:
        cands.append(found)
    return cands


def load_hip_runtime():
    global _hip
    if _hip is not None:
        return _hip
    errors = []
    for path in _hip_runtime_candidates():
        if os.path.sep in path and not os.path.exists(path):
            continue
        try:
            _hip = C.CDLL(path, mode=C.RTLD_GLOBAL)
            return _hip
        except OSError as e:  # keep looking, report all at the end
            errors.append(f"{path}: {e}")
    raise ImportError("tendermintx_amd needs a HIP runtime (libamdhip64.so); tried: " + "; ".join(errors or ["<none found>"]))


def lib():
    """The loaded libtmx.so.  Raises ImportError (never falls back to a CPU path) if it cannot be loaded."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(make -C tendermintx_amd/csrc); tendermintx_amd has no CPU fallback")
    load_hip_runtime()
    L = C.CDLL(LIB_PATH)
    L.tmx_version.restype = C.c_uint32
    L.tmx_status_str.restype = C.c_char_p
    L.tmx_status_str.argtypes = [C.c_int32]
    L.tmx_last_error.restype = C.c_char_p
    L.tmx_last_error.argtypes = [C.c_void_p]
    L.tmx_ctx_create.argtypes = [C.POINTER(Config), C.POINTER(C.c_void_p)]
    L.tmx_ctx_destroy.argtypes = [C.c_void_p]
    L.tmx_ctx_destroy.restype = None
    for f in (L.tmx_elem_count, L.tmx_elem_stride, L.tmx_hint_elem_count):
        f.restype = C.c_uint64
        f.argtypes = [C.c_int32, C.c_uint32]
    L.tmx_witness_batch.argtypes = [C.c_void_p, C.c_int32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_uint64, C.c_void_p]
    L.tmx_out_row_elems.restype = C.c_uint64
    L.tmx_out_row_elems.argtypes = [C.c_int32, C.c_uint32, C.c_uint32]
    L.tmx_witness_batch_opts.argtypes = [C.c_void_p, C.c_int32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32,
                                         C.c_void_p, C.c_uint64, C.c_void_p]
    L.tmx_value_layout_of.argtypes = [C.c_int32, C.c_uint32, C.c_uint32, C.POINTER(ValueLayout)]
    L.tmx_inputs_value_batch.argtypes = [C.c_void_p, C.c_int32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64]
    L.tmx_inputs_value_batch_device.argtypes = [C.c_void_p, C.c_int32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    L.tmx_host_alloc.restype = C.c_void_p
    L.tmx_host_alloc.argtypes = [C.c_void_p, C.c_uint64]
    L.tmx_host_free.restype = None
    L.tmx_host_free.argtypes = [C.c_void_p, C.c_void_p]
    L.tmx_witness_batch_device_sections.argtypes = [C.c_void_p, C.c_int32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                    C.c_void_p, C.c_void_p, C.c_uint32]
    L.tmx_trace_elem_count.restype = C.c_uint64
    L.tmx_trace_elem_count.argtypes = [C.c_int32, C.c_uint32]
    L.tmx_trace_rows_device.argtypes = [C.c_void_p, C.c_int32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
    L.tmx_witness_batch_device.argtypes = [C.c_void_p, C.c_int32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_void_p]
    L.tmx_ntt_set_domain.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64]
    L.tmx_ntt_goldilocks_device.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
    L.tmx_lde_goldilocks_device.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    L.tmx_poseidon_set_constants.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.tmx_poseidon_constants_injected.argtypes = [C.c_void_p]
    L.tmx_poseidon_merkle_digests.restype = C.c_uint64
    L.tmx_poseidon_merkle_digests.argtypes = [C.c_uint32, C.c_uint32]
    L.tmx_poseidon_merkle_device.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    L.tmx_poseidon_permute.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    L.tmx_selftest_fe_invert.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    L.tmx_selftest_f16.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
    L.tmx_eddsa_lanes_device.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    L.tmx_finish_batch_device.argtypes = [C.c_void_p, C.c_int32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_void_p]
    L.tmx_trace_commit_shape.argtypes = [C.c_int32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.tmx_trace_commit_device.argtypes = [C.c_void_p, C.c_int32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    L.tmx_trace_commit_last_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
    L.tmx_shard_range.restype = None
    L.tmx_shard_range.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.tmx_comm_unique_id.argtypes = [C.c_char_p]
    L.tmx_comm_create.argtypes = [C.c_void_p, C.c_char_p, C.c_uint32, C.c_uint32]
    L.tmx_comm_destroy.argtypes = [C.c_void_p]
    L.tmx_comm_info.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    try:
        L.tmx_comm_abort.argtypes = [C.c_void_p]
        L.tmx_comm_sync.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
    except AttributeError:   # only an older build named by $TMX_LIB (tools/ab_lib.py compares library builds): the in-tree library has both
        if not os.environ.get("TMX_LIB"):
            raise
    L.tmx_witness_batch_sharded_device.argtypes = [C.c_void_p, C.c_int32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                   C.c_uint32, C.c_void_p]
    L.tmx_witness_validator_sharded_device.argtypes = [C.c_void_p, C.c_int32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                       C.c_void_p]
    L.tmx_trace_rows_sharded_device.argtypes = [C.c_void_p, C.c_int32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
    L.tmx_trace_rows_validator_sharded_device.argtypes = [C.c_void_p, C.c_int32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
    L.tmx_trace_commit_sharded_device.argtypes = [C.c_void_p, C.c_int32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    L.tmx_last_kernel_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
    L.tmx_kernel_ms_mean.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_float)]
    L.tmx_ctx_stream.restype = C.c_void_p
    L.tmx_ctx_stream.argtypes = [C.c_void_p]
    L.tmx_sync.argtypes = [C.c_void_p]
    L.tmx_last_dedup.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.tmx_key_cache_stats.argtypes = [C.c_void_p, C.POINTER(KeyCacheInfo)]
    L.tmx_key_cache_flush.argtypes = [C.c_void_p]
    L.tmx_key_cache_config.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
    L.tmx_set_cache_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
    L.tmx_eddsa_lanes.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    L.tmx_skip_inputs_from_json.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_uint32, C.c_uint64,
                                            C.c_char_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
    L.tmx_step_inputs_from_json.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_uint32, C.c_uint64, C.c_char_p,
                                            C.c_void_p, C.c_void_p]
    L.tmx_valid_skip_batch.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_void_p]
    L.tmx_skipcheck_inputs_from_json.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_uint32, C.c_void_p, C.POINTER(C.c_uint32), C.c_void_p,
                                                 C.POINTER(C.c_uint32), C.c_void_p, C.POINTER(C.c_uint32)]
    L.tmx_pack_skip_input.argtypes = [C.c_uint64, C.c_char_p, C.c_uint64, C.c_char_p]
    L.tmx_pack_skip_input.restype = None
    L.tmx_unpack_skip_input.argtypes = [C.c_char_p, C.POINTER(C.c_uint64), C.c_char_p, C.POINTER(C.c_uint64)]
    L.tmx_unpack_skip_input.restype = None
    L.tmx_pack_step_input.argtypes = [C.c_uint64, C.c_char_p, C.c_char_p]
    L.tmx_pack_step_input.restype = None
    L.tmx_unpack_step_input.argtypes = [C.c_char_p, C.POINTER(C.c_uint64), C.c_char_p]
    L.tmx_unpack_step_input.restype = None
    _lib = L
    return L


def check(status, ctx=None):
    if status != 0:
        L = lib()
        msg = L.tmx_status_str(status).decode()
        if ctx:
            detail = L.tmx_last_error(ctx).decode()
            if detail:
                msg += " -- " + detail
        raise TmxError(status, msg)
