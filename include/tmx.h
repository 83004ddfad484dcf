/* libtmx -- MI355X-native witness generator for the TendermintX skip / step circuits.  C ABI.
 *
 * This is the drop-in boundary for ONE path of succinctlabs/tendermintx: the value-level witness that the
 * off-chain-input hints produce and the gadget tree consumes.  Reference interfaces replaced:
 *
 *   SkipOffchainInputs::hint   reference circuits/skip.rs:64-102   (reads U64, Bytes32, U64; writes VerifySkipVariable<N>)
 *   StepOffchainInputs::hint   reference circuits/step.rs:56-89    (reads U64, Bytes32;      writes VerifyStepVariable<N>)
 *   InputDataFetcher::get_skip_inputs / get_step_inputs   reference circuits/input/mod.rs:425-523, 316-423
 *   get_validator_data_from_block / validator_hash_field_from_block   reference circuits/input/conversion.rs:59-178
 *   verify_skip / verify_step gadget values   reference circuits/builder/verify.rs:469-563 (+ validator.rs, voting.rs, shared.rs)
 *
 * The Rust host keeps `impl Circuit for SkipCircuit / StepCircuit` (skip.rs:113-143, step.rs:100-127) and
 * bin/skip.rs / bin/step.rs unchanged; only the hint body calls into this library (INTEGRATION.md shows the
 * FFI stub).  All compute below runs in hand-written HIP kernels for gfx950; there is no CPU fallback: every
 * entry point fails with TMX_ERR_HIP if no device is usable.
 *
 * Conventions: every function returns 0 on success or a negative tmx_status; nothing throws across the ABI;
 * the caller owns all buffers passed in; the library never keeps caller pointers after returning; a tmx_ctx is
 * not thread-safe (use one per thread).  Contexts of one device share three internal side streams (the GPU runs four hardware queues:
 * DESIGN.md section 3) and each context owns ONE set of scratch buffers and join events, reused by every call: consecutive
 * tmx_witness_batch_device / tmx_finish_batch_device calls on one context are ordered by the library itself (a call on another stream
 * waits for the end of the previous one); the other *_device entry points (EdDSA lanes, trace rows, NTT) must be stream-ordered by the
 * caller -- the same stream, or an explicit dependency.
 * The host-buffer entry points block until their results are in host memory, so they are ordered by construction.
 */
#ifndef TMX_H
#define TMX_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TMX_KIND_SKIP 0
#define TMX_KIND_STEP 1

#define TMX_N_MAX_LIMIT 512 /* largest VALIDATOR_SET_SIZE_MAX a context accepts (BASELINE config 5) */

/* reference circuits/consts.rs:9-29 */
#define TMX_VALIDATOR_MESSAGE_BYTES_LENGTH_MAX 124
#define TMX_VALIDATOR_BYTE_LENGTH_MAX 46
#define TMX_PROTOBUF_CHAIN_ID_SIZE_BYTES 52
#define TMX_HEADER_PROOF_DEPTH 4

#define TMX_FLAG_SIGNED 1u  /* CommitSig::is_commit() -- conversion.rs:79 */
#define TMX_FLAG_PRESENT 2u /* lane index < commit.signatures.len() -- conversion.rs:70 vs :118 */

typedef enum {
  TMX_OK = 0,
  TMX_ERR_BAD_ARG = -1,
  TMX_ERR_SET_TOO_LARGE = -2, /* reference input/mod.rs:439-444, 338-342: validator set larger than N */
  TMX_ERR_HIP = -3,           /* no device / HIP runtime error (message in tmx_last_error) */
  TMX_ERR_CAPACITY = -4,      /* output buffer or context batch capacity too small */
  TMX_ERR_PARSE = -5,         /* malformed fixture / RPC JSON */
  TMX_ERR_MSG_TOO_LONG = -6,  /* sign-bytes longer than 124 B -- conversion.rs:52 `try_into().unwrap()` */
  TMX_ERR_RCCL = -7           /* RCCL not loadable / a collective failed (message in tmx_last_error) -- SURVEY 8(b) error convention */
} tmx_status;

/* One lane of `target_block_validators` (ValidatorType, reference circuits/variables.rs:69-79) before
 * field-element expansion.  256 B so that a wavefront's lanes read 16-B aligned, coalescable records. */
typedef struct {
  uint8_t pubkey[32];
  uint8_t signature[64]; /* R || s, s little-endian (conversion.rs:43-46) */
  uint8_t message[TMX_VALIDATOR_MESSAGE_BYTES_LENGTH_MAX]; /* sign-bytes, zero padded (conversion.rs:37-39) */
  uint16_t message_byte_length;
  uint8_t validator_byte_length;
  uint8_t flags; /* TMX_FLAG_* */
  uint64_t voting_power;
  uint8_t pad[24];
} tmx_validator_rec;

/* One lane of `trusted_header_validator_hash_fields` (ValidatorHashField, variables.rs:82-88). */
typedef struct {
  uint8_t pubkey[32];
  uint64_t voting_power;
  uint8_t validator_byte_length;
  uint8_t flags;
  uint8_t pad[6];
} tmx_hashfield_rec;

/* The 14 protobuf-encoded header fields (reference circuits/input/tendermint_utils.rs:374-393), each < 80 B. */
typedef struct {
  uint8_t leaf_len[14];
  uint8_t pad[2];
  uint8_t leaf[14][80];
} tmx_header_rec;

/* Per-proof fixed inputs.  skip: block_a = trusted_block, block_b = target_block, hash = trusted_header_hash,
 * header_a = target header, header_b = trusted header, nb_a / nb_b = target / trusted validator counts.
 * step: block_a = prev_block_number, block_b = prev + 1, hash = prev_header_hash, header_a = next header,
 * header_b = prev header, nb_a = next validator count, nb_b = 0. */
typedef struct {
  uint64_t block_a;
  uint64_t block_b;
  uint8_t hash[32];
  uint64_t round; /* commit.round of header_a's block */
  uint32_t nb_a;
  uint32_t nb_b;
  tmx_header_rec header_a;
  tmx_header_rec header_b;
} tmx_proof_rec;

/* Level-0 output + verdicts, one per proof.  fail_mask bit order: DESIGN.md "checks". */
typedef struct {
  uint8_t header[32]; /* target_header (skip.rs:132) / next_header (step.rs:116) */
  uint32_t all_ok;
  uint32_t fail_mask;
  int32_t first_bad_sig; /* first lane whose EdDSA equation fails (conversion.rs:48-49 panic site), or -1 */
  uint32_t gt_target;    /* signed power * 3 > total * 2   (verify.rs:289-303) */
  uint32_t gt_trusted;   /* matched power * 3 > total * 1  (verify.rs:428-436), skip only */
  uint32_t dist_ok;      /* verify_skip_distance (verify.rs:508-526), skip only */
  uint32_t precond;      /* host preconditions of the reference that the circuit itself does not assert: bit 0 nb_a > n_max,
                            bit 1 nb_b > n_max (input/mod.rs:439-444, 338-342 panic there; in-circuit `idx == nb` never fires and
                            every lane counts as enabled).  The host entry points refuse such input with TMX_ERR_SET_TOO_LARGE; the
                            device entry points cannot look at device memory before enqueueing and report it here instead. */
  uint32_t reserved;
} tmx_report;

typedef struct {
  uint32_t n_max;        /* VALIDATOR_SET_SIZE_MAX (const generic of SkipCircuit / StepCircuit), 1..512 */
  uint32_t chain_id_len; /* CHAIN_ID_SIZE_BYTES */
  uint8_t chain_id[TMX_PROTOBUF_CHAIN_ID_SIZE_BYTES]; /* TendermintConfig::CHAIN_ID_BYTES (config.rs:6) */
  uint64_t skip_max;     /* TendermintConfig::SKIP_MAX (config.rs:7) */
  int32_t device;        /* HIP device ordinal */
  uint32_t max_batch;    /* proofs per call this context preallocates scratch for (>= 1) */
} tmx_config;

typedef struct tmx_ctx tmx_ctx;

/* names of the kernels timed by tmx_last_kernel_ms, in launch order */
#define TMX_N_KERNELS 4
#define TMX_K_EDDSA 0     /* per-validator Ed25519 (k_ed_dedup, k_ed_keys, k_ed_tab_x, k_ed_phase1, k_ed_mul_x, k_ed_fin): SHA-512, decode, s*B, h*A, R+hA, affine */
#define TMX_K_PROOF 1     /* k_proof on the context's side stream, concurrent with the EdDSA kernels: marshal + SHA-256 leaves +
                             Merkle trees + header proofs + NxN match + tallies */
#define TMX_K_VERDICT 2   /* join: wait for k_proof, merge the per-lane EdDSA verdicts (k_verdict) */
#define TMX_K_SERIALIZE 3 /* Goldilocks element fill */

uint32_t tmx_version(void);
const char* tmx_status_str(int32_t status);

int32_t tmx_ctx_create(const tmx_config* cfg, tmx_ctx** out);
void tmx_ctx_destroy(tmx_ctx* ctx);
const char* tmx_last_error(const tmx_ctx* ctx);

/* number of Goldilocks elements of one witness, and the row stride used in batched output (a multiple of 16 elements: rows start on a
 * 128-byte line; the pad elements are written as zeros) */
uint64_t tmx_elem_count(int32_t kind, uint32_t n_max);
uint64_t tmx_elem_stride(int32_t kind, uint32_t n_max);
/* offset and length of the hint section H (= VerifySkipVariable<N> / VerifyStepVariable<N> elements) in a row */
uint64_t tmx_hint_elem_count(int32_t kind, uint32_t n_max);

/* ---- host-buffer entry points: what the Rust hint binds.  out_elems receives tmx_elem_count() elements. */
int32_t tmx_skip_witness(tmx_ctx* ctx, const tmx_proof_rec* proof, const tmx_validator_rec* target /*[n_max]*/,
                         const tmx_hashfield_rec* trusted /*[n_max]*/, uint64_t* out_elems, uint64_t cap_elems,
                         tmx_report* report);
int32_t tmx_step_witness(tmx_ctx* ctx, const tmx_proof_rec* proof, const tmx_validator_rec* target /*[n_max]*/,
                         uint64_t* out_elems, uint64_t cap_elems, tmx_report* report);
/* n_proofs independent proofs; rows of tmx_elem_stride() elements; out_elems may be NULL (reports only) */
int32_t tmx_witness_batch(tmx_ctx* ctx, int32_t kind, uint32_t n_proofs, const tmx_proof_rec* proofs,
                          const tmx_validator_rec* targets /*[n_proofs][n_max]*/,
                          const tmx_hashfield_rec* trusteds /*[n_proofs][n_max], NULL for step*/, uint64_t* out_elems,
                          uint64_t cap_elems, tmx_report* reports);

/* ---- host-buffer entry point with a section selection and a transfer format.  A hint body only needs H (43 % of a skip row at
 * N = 128: what SkipOffchainInputs::hint writes to its output stream, skip.rs:85-100), and every element of this witness is < 2^32
 * (a bit, a byte, a u32 limb), so a PCIe-bound caller can fetch it as u32 and widen with F::from_canonical_u32 on its side.
 * out receives n_proofs DENSE rows of tmx_out_row_elems(kind, n_max, sections) elements of 8 (TMX_OUT_U64) or 4 (TMX_OUT_U32) bytes;
 * with both sections selected a row is H then D.  Sections that are not selected are not serialized on the device either. */
#define TMX_SEC_HINT 1u
#define TMX_SEC_DERIVED 2u
#define TMX_SEC_ALL 3u
#define TMX_OUT_U64 0u
#define TMX_OUT_U32 1u
uint64_t tmx_out_row_elems(int32_t kind, uint32_t n_max, uint32_t sections);
int32_t tmx_witness_batch_opts(tmx_ctx* ctx, int32_t kind, uint32_t n_proofs, const tmx_proof_rec* proofs,
                               const tmx_validator_rec* targets, const tmx_hashfield_rec* trusteds /*NULL for step*/, uint32_t sections,
                               uint32_t format, void* out, uint64_t cap_bytes, tmx_report* reports);

/* ---- the TYPED VALUE of the hint: what the reference's hint bodies actually hold before plonky2x expands it into field elements.
 * SkipOffchainInputs::hint builds a `VerifySkipStruct` literal from a `SkipInputs<F>` (reference circuits/skip.rs:85-98, circuits/input/mod.rs:60-74)
 * and hands it to `write_value` (skip.rs:100), which does the bit / limb expansion itself; StepOffchainInputs::hint the same with `StepInputs<F>`
 * (step.rs:75-87, input/mod.rs:45-58).  These structs mirror those two types FIELD BY FIELD -- bytes as bytes, u64 as u64 -- so a hint body
 * assigns them by name (rust-shim/skip_hint.rs.example) and no knowledge of the element order inside plonky2x's variable types is needed:
 * 38 KB per N = 128 skip proof instead of the 1.86 MB of u64 elements (0.93 MB as u32) of the expanded row.
 * One proof's value = fixed part | tmx_validator_value[n_max] | tmx_hashfield_value[n_max] (skip only)
 *                     [| derived values, with TMX_SEC_DERIVED: see tmx_value_layout], every part 16-byte aligned, little-endian. */
typedef struct {             /* ValidatorType<F>: the value type of ValidatorVariable, reference circuits/variables.rs:69-79; built by
                                get_validator_data_from_block, circuits/input/conversion.rs:59-137 */
  uint8_t pubkey[32];        /* CompressedEdwardsY */
  uint8_t sig_r[32];         /* EDDSASignatureVariableValue.r (compressed point) */
  uint8_t sig_s[32];         /* EDDSASignatureVariableValue.s: U256, little-endian */
  uint8_t message[TMX_VALIDATOR_MESSAGE_BYTES_LENGTH_MAX];
  uint32_t message_byte_length;
  uint64_t voting_power;
  uint32_t validator_byte_length;
  uint32_t signed_;          /* bool `signed` */
} tmx_validator_value;       /* 240 B */
typedef struct {             /* ValidatorHashField<F>: variables.rs:82-88; validator_hash_field_from_block, conversion.rs:139-178 */
  uint8_t pubkey[32];
  uint64_t voting_power;
  uint32_t validator_byte_length;
  uint32_t pad;
} tmx_hashfield_value;       /* 48 B */
typedef struct {             /* ChainIdProofValueType<F>: variables.rs:35-41; input/mod.rs:471-482 */
  uint8_t proof[TMX_HEADER_PROOF_DEPTH][32];
  uint32_t enc_chain_id_byte_length;
  uint8_t chain_id[TMX_PROTOBUF_CHAIN_ID_SIZE_BYTES]; /* the encoded field resized to 52 bytes */
  uint8_t pad[8];
} tmx_chain_id_proof_value;  /* 192 B */
typedef struct {             /* HeightProofValueType<F>: variables.rs:49-55; input/mod.rs:484-493 */
  uint8_t proof[TMX_HEADER_PROOF_DEPTH][32];
  uint32_t enc_height_byte_length;
  uint32_t pad;
  uint64_t height;
} tmx_height_proof_value;    /* 144 B */
typedef struct {             /* InclusionProof<HEADER_PROOF_DEPTH, PROTOBUF_HASH_SIZE_BYTES, F>: input/mod.rs:303-314 */
  uint8_t proof[TMX_HEADER_PROOF_DEPTH][32];
  uint8_t leaf[34];
  uint8_t pad[14];
} tmx_hash_inclusion_proof_value;     /* 176 B */
typedef struct {             /* InclusionProof<HEADER_PROOF_DEPTH, PROTOBUF_BLOCK_ID_SIZE_BYTES, F> */
  uint8_t proof[TMX_HEADER_PROOF_DEPTH][32];
  uint8_t leaf[72];
  uint8_t pad[8];
} tmx_block_id_inclusion_proof_value; /* 208 B */
typedef struct {             /* SkipInputs<F> (input/mod.rs:60-74) without its two Vecs, which follow as arrays; + the verdicts */
  uint8_t target_header[32];
  uint8_t trusted_header[32];
  uint64_t round;
  uint32_t nb_target_validators;
  uint32_t nb_trusted_validators;
  tmx_chain_id_proof_value target_block_chain_id_proof;
  tmx_height_proof_value target_block_height_proof;
  tmx_hash_inclusion_proof_value target_block_validators_hash_proof;
  tmx_hash_inclusion_proof_value trusted_block_validators_hash_proof;
  tmx_report report;
} tmx_skip_inputs_fixed;     /* 832 B */
typedef struct {             /* StepInputs<F> (input/mod.rs:45-58) without its Vec; + the verdicts */
  uint8_t next_header[32];
  uint64_t round;
  uint32_t nb_validators;
  uint32_t pad;
  tmx_chain_id_proof_value next_block_chain_id_proof;
  tmx_height_proof_value next_block_height_proof;
  tmx_hash_inclusion_proof_value next_block_validators_hash_proof;
  tmx_block_id_inclusion_proof_value next_block_last_block_id_proof;
  tmx_hash_inclusion_proof_value prev_block_next_validators_hash_proof;
  tmx_report report;
} tmx_step_inputs_fixed;     /* 1008 B */
/* Derived Level-1 values (section D of the row, DESIGN.md "Witness layout") in packed form -- optional (TMX_SEC_DERIVED): what a
 * replacement for Curta's result hints would consume.  Pad bytes are written as zeros. */
typedef struct {             /* D.1a + D.1b of one target lane */
  uint8_t sha512_digest[64]; /* SHA-512(R | A | M) of the lane's effective triple */
  uint8_t h[32];             /* digest mod l, little-endian */
  uint8_t points[10][32];    /* A.x A.y R.x R.y sB.x sB.y hA.x hA.y (R+hA).x (R+hA).y: canonical, little-endian */
  uint32_t eddsa_ok;
  uint32_t decode_ok;
  uint8_t pad0[24];
  uint8_t marshalled[TMX_VALIDATOR_BYTE_LENGTH_MAX];
  uint8_t pad1[2];
  uint8_t leaf_hash[32];
  uint8_t flags[6];          /* enabled, hash_in_msg, is_precommit, height_ok, round_ok, sigdata_ok */
  uint8_t pad2[2];
  uint64_t total_prefix;     /* running sum of enabled powers (voting.rs:31-63) */
  uint64_t signed_prefix;    /* running sum of signed powers (voting.rs:79-89) */
  uint8_t pad3[8];
} tmx_target_lane_derived;   /* 560 B */
typedef struct {             /* D.2a + D.2b of one trusted lane (skip) */
  uint8_t marshalled[TMX_VALIDATOR_BYTE_LENGTH_MAX];
  uint8_t pad1[2];
  uint8_t leaf_hash[32];
  uint8_t flags[2];          /* enabled, matched (verify.rs:398-418) */
  uint8_t pad2[6];
  uint64_t total_prefix;
  uint64_t matched_prefix;
  uint8_t pad3[8];
} tmx_trusted_lane_derived;  /* 112 B */
typedef struct {             /* D.5 / D.6 of one proof */
  uint8_t proofs[5][5][32];  /* chain id, height, validators hash, X, Y (step only): leaf hash then the four path nodes */
  uint8_t height_leaf[11];   /* 00 08 varint9(height) */
  uint8_t pad0[5];
  uint64_t tally_target[4];  /* total, acc, acc * 3, total * 2 */
  uint64_t tally_trusted[4]; /* skip only */
  uint32_t verdicts[4];      /* gt_target, gt_trusted, dist_gt, dist_le (the last three: skip only) */
  uint32_t checks[16];       /* 13 (skip) / 15 (step) check words, tmx_report.fail_mask bit order */
  uint32_t all_ok;
  uint32_t pad1;
  uint64_t height;
} tmx_proof_derived;         /* 976 B */
typedef struct {
  uint64_t bytes;              /* of one proof's value: proof p of a batch starts at p * bytes */
  uint32_t fixed_bytes;        /* sizeof(tmx_skip_inputs_fixed) / sizeof(tmx_step_inputs_fixed), at offset 0 */
  uint32_t off_validators;     /* tmx_validator_value[n_max] */
  uint32_t off_hashfields;     /* tmx_hashfield_value[n_max]; 0 for step */
  uint32_t off_target_lanes;   /* tmx_target_lane_derived[n_max]; this and the following: 0 without TMX_SEC_DERIVED */
  uint32_t off_trusted_lanes;  /* tmx_trusted_lane_derived[n_max]; 0 for step */
  uint32_t off_nodes_target;   /* [tree_nodes][32]: every node of the fixed-shape validator tree, layer by layer (D.3) */
  uint32_t off_nodes_trusted;  /* (D.4) 0 for step */
  uint32_t off_proof_derived;  /* tmx_proof_derived */
  uint32_t tree_nodes;
  uint32_t reserved;
} tmx_value_layout;
/* sections: TMX_SEC_HINT (the reference's hint value) or TMX_SEC_ALL (+ the derived values); TMX_ERR_BAD_ARG otherwise */
int32_t tmx_value_layout_of(int32_t kind, uint32_t n_max, uint32_t sections, tmx_value_layout* out);
/* Host buffers in, host buffer out: `out` receives n_proofs values of layout.bytes each.  No element row is produced on the device at all
 * (the serializer does not run).  Fastest with page-locked buffers (tmx_host_alloc): the copies are then direct DMA, and a page-locked
 * `out` is written by the GPU itself. */
int32_t tmx_inputs_value_batch(tmx_ctx* ctx, int32_t kind, uint32_t n_proofs, const tmx_proof_rec* proofs, const tmx_validator_rec* targets,
                               const tmx_hashfield_rec* trusteds /*NULL for step*/, uint32_t sections, void* out, uint64_t cap_bytes);
/* the single-proof forms a hint body binds (reference circuits/skip.rs:64-102, circuits/step.rs:56-89) */
int32_t tmx_skip_inputs_value(tmx_ctx* ctx, const tmx_proof_rec* proof, const tmx_validator_rec* target /*[n_max]*/,
                              const tmx_hashfield_rec* trusted /*[n_max]*/, uint32_t sections, void* out, uint64_t cap_bytes);
int32_t tmx_step_inputs_value(tmx_ctx* ctx, const tmx_proof_rec* proof, const tmx_validator_rec* target /*[n_max]*/, uint32_t sections,
                              void* out, uint64_t cap_bytes);
/* Device pointers in and out, asynchronous on hip_stream like tmx_witness_batch_device (d_out may also be the device address of mapped
 * page-locked host memory). */
int32_t tmx_inputs_value_batch_device(tmx_ctx* ctx, int32_t kind, uint32_t n_proofs, const void* d_proofs, const void* d_targets,
                                      const void* d_trusteds, uint32_t sections, void* d_out, void* hip_stream);
/* page-locked host memory for the host entry points (hipHostMalloc: mapped, portable).  NULL on failure. */
void* tmx_host_alloc(tmx_ctx* ctx, uint64_t bytes);
void tmx_host_free(tmx_ctx* ctx, void* p);

/* ---- device-resident entry point: inputs already in HBM, outputs stay in HBM.  All pointers are device
 * pointers of the context's device; `hip_stream` is the hipStream_t to enqueue on, used exactly as passed (NULL = the HIP
 * default stream; tmx_ctx_stream() = the context's own stream).  Asynchronous: returns after enqueueing. */
void* tmx_ctx_stream(tmx_ctx* ctx);
int32_t tmx_witness_batch_device(tmx_ctx* ctx, int32_t kind, uint32_t n_proofs, const void* d_proofs, const void* d_targets,
                                 const void* d_trusteds, void* d_out_elems, void* d_reports, void* hip_stream);
/* the same with a section selection: rows keep tmx_elem_stride(), sections that are not selected are left unwritten */
int32_t tmx_witness_batch_device_sections(tmx_ctx* ctx, int32_t kind, uint32_t n_proofs, const void* d_proofs, const void* d_targets,
                                          const void* d_trusteds, void* d_out_elems, void* d_reports, void* hip_stream, uint32_t sections);
/* The two halves of the call above, for the validator-sharded single-proof mode (BASELINE config 5): each GPU runs
 * k_eddsa on its slice of lanes, the 448-B lane records are exchanged (one RCCL all-gather), then k_proof + k_serialize
 * run on the reassembled records.  d_ed_out / d_ed: TMX ED records, 448 B per lane (layout at tmx_eddsa_lanes). */
int32_t tmx_eddsa_lanes_device(tmx_ctx* ctx, uint32_t n_lanes, const void* d_lanes, void* d_ed_out, void* hip_stream);
int32_t tmx_finish_batch_device(tmx_ctx* ctx, int32_t kind, uint32_t n_proofs, const void* d_proofs, const void* d_targets,
                                const void* d_trusteds, const void* d_ed, void* d_out_elems, void* d_reports, void* hip_stream);
/* HIP-event times (ms) of the kernels of the LAST tmx_witness_batch_device / host call; blocks until they finished */
int32_t tmx_last_kernel_ms(tmx_ctx* ctx, float ms[TMX_N_KERNELS]);
/* mean over the last `last_k` enqueued batches (each batch keeps its own event set, ring of 128): lets a caller time
 * a whole region without synchronising inside it */
int32_t tmx_kernel_ms_mean(tmx_ctx* ctx, uint32_t last_k, float ms[TMX_N_KERNELS]);
int32_t tmx_sync(tmx_ctx* ctx);
/* EdDSA stage bookkeeping of the last launch: number of distinct (effective) public keys among its lanes and whether h*A of any lane
 * walked a per-key table (a key resident in the key cache, or a new key with >= 8 lanes per key on average; TMX_DEDUP=0|1|2 forces
 * never / automatic / whenever a table fits).  Blocks. */
int32_t tmx_last_dedup(tmx_ctx* ctx, uint32_t* n_unique, uint32_t* used_tables);

/* ---- multi-GPU: one process per GPU, the exchange step behind this ABI so that the host the reference actually has -- the
 * Rust process of bin/skip.rs, reached only through SkipOffchainInputs::hint (reference circuits/skip.rs:64-102) -- can shard without
 * any Python.  RCCL is resolved at run time from the host process (no DT_NEEDED, like the HIP runtime): $TMX_RCCL_LIB if set (then it is
 * the only candidate), else an already loaded librccl, else librccl.so.1 / librccl.so by the loader's search path.  A context-less call
 * (tmx_comm_unique_id) leaves its reason in tmx_last_error(NULL), per thread.  Bootstrap as with NCCL: ONE rank calls
 * tmx_comm_unique_id, hands the 128 bytes to the others by whatever channel the host has (a file, its RPC, MPI ...), then every rank calls
 * tmx_comm_create(ctx, id, rank, world) on a context of ITS device.  world = 1 needs no id and loads nothing.
 *   tmx_shard_range                       contiguous [lo, hi) of n_items for `rank` of `world`; sizes differ by at most one
 *   tmx_witness_batch_sharded_device      BASELINE configs[3]: n_total independent proofs, every rank holds all input records and an output
 *                                         buffer for all rows; rank r computes rows [lo_r, hi_r) in place; gather != 0 then makes every row
 *                                         (and report) resident on every rank -- ONE exchange in place, no padding, no staging copy: an
 *                                         ncclAllGather when n_total divides by the world (256 proofs over 2 / 4 / 8 ranks), else a group of
 *                                         broadcasts, each rank the root of its own slice.  gather = 0: no data-path collective at all.
 *   tmx_witness_validator_sharded_device  BASELINE configs[4]: the n_proofs * n_max validator lanes split across the ranks for the EdDSA stage,
 *                                         ONE grouped exchange of the 448-byte lane records, then every rank finishes every proof
 *                                         (tmx_finish_batch_device on the reassembled records): full rows + reports on every rank.
 * Both are asynchronous on hip_stream like tmx_witness_batch_device; n_total / the lanes may be smaller than the world (empty shards).
 *
 * FAILURE CONTRACT of the sharded calls (round 6).  A collective is entered by every rank or by none.  Argument errors (TMX_ERR_BAD_ARG) are
 * assumed to be the same on every rank and are returned before anything is enqueued.  A failure that is LOCAL to one rank between the start of
 * a sharded call and its exchange -- its shard exceeds ITS context's max_batch, a launch fails on ITS device -- must not leave the peers waiting
 * in the collective: the failing rank aborts its communicator (ncclCommAbort) and returns TMX_ERR_RCCL (tmx_last_error names the local cause);
 * the peers' exchange then fails instead of hanging -- at the collective call itself, or asynchronously: a host that needs a bounded wait
 * synchronises with tmx_comm_sync(ctx, stream, timeout_ms) instead of hipStreamSynchronize; it polls the stream and ncclCommGetAsyncError, and on
 * an asynchronous error or the timeout aborts this rank's communicator too and returns TMX_ERR_RCCL.  After TMX_ERR_RCCL from any sharded call
 * or from tmx_comm_sync the context refuses further sharded calls (TMX_ERR_RCCL) until tmx_comm_create is called again -- on EVERY rank, with
 * a fresh id; the non-sharded entry points and the caches of the context are unaffected.  The buffers of a failed call hold nothing defined.
 *   tmx_comm_abort   what a host calls on its healthy contexts when IT learns (by its own channel) that a peer process died
 *   tmx_comm_sync    timeout_ms = 0: no timeout (asynchronous errors still end the wait) */
#define TMX_UNIQUE_ID_BYTES 128
void tmx_shard_range(uint64_t n_items, uint32_t rank, uint32_t world, uint64_t* lo, uint64_t* hi);
int32_t tmx_comm_unique_id(uint8_t out[TMX_UNIQUE_ID_BYTES]);
int32_t tmx_comm_create(tmx_ctx* ctx, const uint8_t unique_id[TMX_UNIQUE_ID_BYTES] /* NULL iff world == 1 */, uint32_t rank, uint32_t world);
int32_t tmx_comm_destroy(tmx_ctx* ctx);
int32_t tmx_comm_info(const tmx_ctx* ctx, uint32_t* rank, uint32_t* world);  /* (0, 1) before tmx_comm_create */
int32_t tmx_comm_abort(tmx_ctx* ctx);
int32_t tmx_comm_sync(tmx_ctx* ctx, void* hip_stream, uint32_t timeout_ms);
int32_t tmx_witness_batch_sharded_device(tmx_ctx* ctx, int32_t kind, uint32_t n_total, const void* d_proofs, const void* d_targets,
                                         const void* d_trusteds, void* d_out_elems, void* d_reports, uint32_t gather, void* hip_stream);
int32_t tmx_witness_validator_sharded_device(tmx_ctx* ctx, int32_t kind, uint32_t n_proofs, const void* d_proofs, const void* d_targets,
                                             const void* d_trusteds, void* d_out_elems, void* d_reports, void* hip_stream);

/* The Level-2 trace rows across the ranks (the one payload of this path big enough for xGMI to matter: 41 MB per proof at N = 128, 136 MB at
 * N = 512).  Same conventions as above: d_trace_out holds the rows of ALL proofs (tmx_trace_elem_count() u64 each), asynchronous on hip_stream.
 *   tmx_trace_rows_sharded_device            after tmx_witness_batch_sharded_device of the same n_total: rank r writes the rows of its proofs
 *                                            [lo_r, hi_r) in place; gather != 0: one exchange (ncclAllGather when the shards are equal) leaves
 *                                            every proof's rows on every rank.
 *   tmx_trace_rows_validator_sharded_device  after tmx_witness_validator_sharded_device of the same n_proofs: the per-lane sections (ladders,
 *                                            SHA-512 rounds: 93 % of the rows) are computed for this rank's lanes only and exchanged lane slab
 *                                            by lane slab (one proof: one all-gather per section); the small per-proof sections (leaf / tree /
 *                                            header SHA-256, N x N) are computed on every rank.  All rows on every rank afterwards.
 *   tmx_trace_commit_sharded_device          tmx_trace_commit_device over this rank's own proofs (no row crosses a link), its cap into slot
 *                                            `rank` of d_caps[world][4 << cap_height], then ONE all-gather of the caps. */
int32_t tmx_trace_rows_sharded_device(tmx_ctx* ctx, int32_t kind, uint32_t n_total, const void* d_targets, const void* d_trusteds, void* d_trace_out,
                                      uint32_t sections, uint32_t gather, void* hip_stream);
int32_t tmx_trace_rows_validator_sharded_device(tmx_ctx* ctx, int32_t kind, uint32_t n_proofs, const void* d_targets, const void* d_trusteds,
                                                void* d_trace_out, uint32_t sections, void* hip_stream);
int32_t tmx_trace_commit_sharded_device(tmx_ctx* ctx, int32_t kind, uint32_t n_total, uint32_t section, uint32_t log_blowup, uint32_t cap_height,
                                        const void* d_trace_rows, uint64_t* d_caps /*[world][4 << cap_height]*/, void* hip_stream);

/* ---- persistent per-key table cache.  h*A of a lane is 32 additions from a 655-KB window table of its public key instead of 252
 * doublings + 64 additions; the context keeps those tables in a content-addressed cache in HBM (key = the 32 public-key bytes, all 32
 * compared on a hit), so that a validator set that was seen by an earlier call -- a light client re-verifies the same, slowly changing set
 * for days (reference bin/tendermintx.rs:171) -- skips the decode -> doubling chain -> table build entirely.  New keys are inserted by the
 * call that first sees them (a single-proof call builds their tables off its critical path, for the next call); when the cache runs full
 * the least recently used keys are evicted.  Exact group arithmetic on every path: the witness is bit-identical with the cache on, off,
 * cold or warm.  Default capacity: 1024 .. 8192 keys by max_batch * n_max (TMX_KEY_CACHE_KEYS overrides; TMX_KEY_CACHE=0 disables).
 * All three calls block until the context's work in flight is done. */
typedef struct {
  uint32_t capacity_keys, resident_keys, enabled, epoch;
  uint64_t bytes_per_key;
  uint32_t last_new_keys, last_hit_keys, last_hit_lanes, last_built_keys; /* the last EdDSA launch */
  uint64_t hit_lanes, miss_lanes, built_keys, evicted_keys, evictions, launches; /* since the cache was created / flushed */
} tmx_key_cache_info;
int32_t tmx_key_cache_stats(tmx_ctx* ctx, tmx_key_cache_info* out);
int32_t tmx_key_cache_flush(tmx_ctx* ctx);                                      /* forget every key */
int32_t tmx_key_cache_config(tmx_ctx* ctx, uint32_t enabled, uint32_t max_keys); /* max_keys = 0 keeps the capacity; a new capacity flushes */
/* The validator-set cache of a context (round 5): marshalled validators, leaf hashes and every node of the fixed-shape tree
 * (reference circuits/builder/validator.rs:185-252) depend on the validator set alone -- (pubkey, voting power, validator_byte_length) of every lane and
 * the number of enabled lanes -- not on the proof, and a light client re-verifies the same slowly changing sets.  A batch's k_proof looks both
 * sets of a proof up by a 64-bit fingerprint, compares EVERY key byte, and copies the cached values instead of hashing (15 SHA-256
 * compressions off its chain); sets it had to compute are inserted.  256 sets per context, least-recently-used eviction at launch granularity
 * (round 6): a hit or an insert stamps the set with the launch's number, and a one-workgroup kernel behind every k_proof launch keeps an eighth of the slots free by
 * evicting the sets used longest ago -- a prover that lives for months (reference bin/tendermintx.rs:171) keeps the sets it is verifying now, not
 * the first 256 it ever saw.  tmx_key_cache_flush empties it too; TMX_SET_CACHE=0 disables; TMX_SET_CACHE_SETS=<4..256> = capacity (tests).
 * Bit-identical by construction.  out: [0] sets resident [1] sets served from the cache [2] sets computed [3] sets inserted [4] sets evicted
 * (totals since creation / the last flush) [5] capacity [6..7] reserved (0). */
int32_t tmx_set_cache_stats(tmx_ctx* ctx, uint32_t out[8]);

/* ---- Level-2 trace rows (SURVEY 8a "Level-2", 8f rank 2): the row-level execution trace behind the Level-1 values -- what the reference
 * produces inside Curta's trace generators for `curta_eddsa_verify_sigs_conditional` (reference circuits/builder/verify.rs:248-259) and
 * `curta_sha256_variable` (validator.rs:228).  Those sources are absent, so the row layout is this build's own specification (DESIGN.md
 * "Level-2 trace rows"), validated row by row by the constraint checker under oracle/c -- NOT claimed equal to Curta's columns.
 * Per proof tmx_trace_elem_count() elements (u64, every value < 2^32):
 *   ladders   lane i, ladder k (0: s*B, 1: h*A), 256 rows x 65: bit | acc | dbl = 2 acc | add = dbl + P | nxt = bit ? add : dbl,
 *             points as canonical affine (x, y) in eight little-endian u32 limbs each; acc_0 = (0, 1), acc_{r+1} = nxt_r, nxt_255 = k * P
 *   SHA-512   lane i, block b < 2, 80 rounds x 18: W_t and a..h after the round (64-bit words as lo, hi)
 *   SHA-256   validator leaf hashes of the target (and, for skip, trusted) set: 64 rounds x 9
 *   N x N     skip: signed[i] & (target pubkey i == trusted pubkey j)
 *   tree      set s, node slot of the fixed-shape validator tree (Level-1 order): 2 blocks x 64 rounds x 9 of SHA-256(01 | L | R) over the
 *             slot's two children as Level-1 holds them (every pair is hashed, then selected: validator.rs:248-251); promoted slots are zero
 *   header    the header proofs in Level-1 order (chain id, height, validators hash, X, Y), each the leaf hash and the four path-node
 *             hashes: 2 blocks x 64 x 9 each (verify.rs:189-209, shared.rs:183-203)
 * tmx_trace_rows_device reads the Level-1 lane records the context holds: call it after tmx_witness_batch_device of the SAME batch, on the
 * same stream.  d_trace_out: n_proofs * tmx_trace_elem_count() u64.  41 MB per proof at N = 128: this launch is HBM-write work. */
#define TMX_TRACE_LADDERS 1u
#define TMX_TRACE_SHA512 2u
#define TMX_TRACE_SHA256 4u
#define TMX_TRACE_MATCH 8u
#define TMX_TRACE_TREE 16u
#define TMX_TRACE_HEADER 32u
#define TMX_TRACE_ALL 63u
uint64_t tmx_trace_elem_count(int32_t kind, uint32_t n_max);
int32_t tmx_trace_rows_device(tmx_ctx* ctx, int32_t kind, uint32_t n_proofs, const void* d_targets, const void* d_trusteds, void* d_trace_out,
                              uint32_t sections, void* hip_stream);

/* ---- the commit pipeline on the device: what the reference's `prove` does with the trace in ONE process -- witness ->
 * trace -> low-degree extension -> Merkle commit (reference circuits/skip.rs:119-133, through plonky2x / starkyx / plonky2, absent here) --
 * chained on the GPU so that only the cap leaves it: the rows of ONE section of every proof of the batch (d_trace_rows: what
 * tmx_trace_rows_device wrote; row-major, tmx_trace_elem_count() elements per proof) become n_proofs * width columns of 2^log_rows
 * elements (zero rows behind a proof's own; a tiled transpose), every column is extended to the coset shift <omega_(2^log_rows << log_blowup)>
 * (tmx_lde_goldilocks_device), and the Merkle tree over the extended rows (leaf = row across ALL columns of the batch, hash_no_pad; inner
 * nodes two_to_one: tmx_poseidon_merkle_device) is reduced to its cap: d_cap receives 4 << cap_height u64.
 * section: ONE of TMX_TRACE_LADDERS / SHA512 / SHA256 / TREE / HEADER (the N x N match bits are not a row table).
 * Same caveats as its stages: own row layout, natural (not bit-reversed) row order, no salt, Poseidon constants as set on the context --
 * parity is pinned against the CPU chain under oracle/c (tmxo_trace -> tmxo_ntt -> tmxo_poseidon), not against plonky2.
 * Scratch (columns, extended columns, tree levels) is owned by the context and grows on demand: (1 + 3 * 2^log_blowup) * n_proofs * width *
 * 2^log_rows * 8 B -- 30 GB for the SHA-512 section of 256 proofs at N = 128, 8 x blow-up.  Asynchronous on hip_stream. */
int32_t tmx_trace_commit_shape(int32_t kind, uint32_t n_max, uint32_t section, uint32_t* log_rows, uint32_t* width);
int32_t tmx_trace_commit_device(tmx_ctx* ctx, int32_t kind, uint32_t n_proofs, uint32_t section, uint32_t log_blowup, uint32_t cap_height,
                                const void* d_trace_rows, uint64_t* d_cap, void* hip_stream);
/* HIP-event times (ms) of the stages of the LAST tmx_trace_commit_device: columns, LDE, Merkle; blocks until they finished */
int32_t tmx_trace_commit_last_ms(tmx_ctx* ctx, float ms[3]);

/* ---- per-lane Level-1 EdDSA values only (unit-test / profiling hook of the dominant kernel).
 * d_out: 448 B per lane = digest[64] | h[32] | A.x A.y R.x R.y sB.x sB.y hA.x hA.y sum.x sum.y [10][32] | ok u32 |
 * decode_ok u32 | pad.  Host variant copies in/out. */
int32_t tmx_eddsa_lanes(tmx_ctx* ctx, uint32_t n_lanes, const tmx_validator_rec* lanes, uint8_t* out /*[n_lanes][448]*/);

/* ---- input codec: the reference's on-disk fixture / CometBFT RPC JSON -> packed records
 * (InputDataFetcher fixture mode, reference circuits/input/mod.rs:188-282; conversion.rs:59-178;
 *  tendermint_utils.rs:374-441).  Pure host code, no hashing: the trusted header hash is NOT computed here. */
int32_t tmx_skip_inputs_from_json(const char* trusted_commit_json, const char* trusted_validators_json,
                                  const char* target_commit_json, const char* target_validators_json, uint32_t n_max,
                                  uint64_t trusted_block, const uint8_t trusted_header_hash[32], uint64_t target_block,
                                  tmx_proof_rec* proof, tmx_validator_rec* target, tmx_hashfield_rec* trusted);
int32_t tmx_step_inputs_from_json(const char* prev_commit_json, const char* next_commit_json, const char* next_validators_json,
                                  uint32_t n_max, uint64_t prev_block, const uint8_t prev_header_hash[32],
                                  tmx_proof_rec* proof, tmx_validator_rec* target);

/* ---- the caller of the path: `is_valid_skip` (reference circuits/input/tendermint_utils.rs:444-482) for many
 * candidate target blocks at once -- what `find_block_to_request` (reference circuits/input/mod.rs:160-186) asks block by block.
 * For candidate c:  shared = sum over start validators found (by address) in target set c of
 *                            power_in_target * #(commit signatures of c carrying that address, commit or nil votes alike),
 *                   valid  = f64(total power of target set c) * (1/3) <= f64(shared)      (IEEE double, as the reference computes it). */
typedef struct {
  uint8_t address[20];
  uint8_t has_address; /* signatures: CommitSig::validator_address().is_some() (flags 2 and 3); validators: 1 */
  uint8_t pad[3];
  uint64_t voting_power; /* validators only */
} tmx_addr_rec; /* 32 B */
/* start[n_start]; per candidate: targets[c][n_max] (n_targets[c] used), sigs[c][n_max] (n_sigs[c] used).  Outputs per candidate. */
int32_t tmx_valid_skip_batch(tmx_ctx* ctx, uint32_t n_candidates, const tmx_addr_rec* start, uint32_t n_start,
                             const tmx_addr_rec* targets, const uint32_t* n_targets, const tmx_addr_rec* sigs, const uint32_t* n_sigs,
                             uint8_t* valid, uint64_t* shared_power, uint64_t* total_power);
/* codec for it: `/validators` JSON of the start block, `/validators` + `/commit` JSON of one candidate -> records (each array n_max long) */
int32_t tmx_skipcheck_inputs_from_json(const char* start_validators_json, const char* target_validators_json, const char* target_commit_json,
                                       uint32_t n_max, tmx_addr_rec* start, uint32_t* n_start, tmx_addr_rec* target, uint32_t* n_target,
                                       tmx_addr_rec* sigs, uint32_t* n_sigs);

/* ---- public I/O packing: abi.encodePacked(uint64,bytes32,uint64) / (uint64,bytes32)
 * (reference contracts/src/TendermintX.sol:104-108, circuits/skip.rs:120-122, circuits/step.rs:107-108) */
void tmx_pack_skip_input(uint64_t trusted_block, const uint8_t trusted_header_hash[32], uint64_t target_block, uint8_t out[48]);
void tmx_unpack_skip_input(const uint8_t in[48], uint64_t* trusted_block, uint8_t trusted_header_hash[32], uint64_t* target_block);
void tmx_pack_step_input(uint64_t prev_block, const uint8_t prev_header_hash[32], uint8_t out[40]);
void tmx_unpack_step_input(const uint8_t in[40], uint64_t* prev_block, uint8_t prev_header_hash[32]);

/* ---- Goldilocks NTT / coset low-degree extension (the step after the witness fill of a plonky2-style prover).
 * Own statement of the published definitions (plonky2_field is not in the reference tree; parity pinned against the CPU restatement
 * under oracle/c):  p = 2^64 - 2^32 + 1, omega_N = root^(2^32 / N) for a primitive 2^32-th root of unity `root`; forward
 * X[j] = sum_i x[i] omega_N^(ij), inverse with N^-1, natural order in and out, any u64 input taken mod p, canonical outputs.
 * Domain constants: by default the ones recalled from plonky2's GoldilocksField -- POWER_OF_TWO_GENERATOR = 7277203076849721926 and coset
 * shift MULTIPLICATIVE_GROUP_GENERATOR = 14293326489335486720 (the first is the second to the power (p-1)/2^32: checked in
 * tests/test_ntt_oracle.py; plonky2's source is absent, so "recalled" stays the word).  tmx_ntt_set_domain selects another convention,
 * e.g. g = 7 with root 0x185629dcda58878c (Plonky3 / winterfell).
 * n_cols columns of 2^log_n elements, column c at element c << log_n; device pointers, asynchronous on hip_stream (used exactly as
 * passed).  In place (d_out == d_in) is allowed.
 * tmx_lde_goldilocks_device: evaluations on <omega_N> -> evaluations on the coset shift <omega_M>, M = N << log_blowup (interpolate,
 * scale coefficient i by shift^i, zero-pad, evaluate); d_out holds n_cols << (log_n + log_blowup) elements. */
int32_t tmx_ntt_set_domain(tmx_ctx* ctx, uint64_t root_2_32, uint64_t coset_shift);
#define TMX_NTT_MAX_LOG 22
int32_t tmx_ntt_goldilocks_device(tmx_ctx* ctx, uint32_t log_n, uint32_t n_cols, const uint64_t* d_in, uint64_t* d_out,
                                  int32_t inverse, void* hip_stream);
int32_t tmx_lde_goldilocks_device(tmx_ctx* ctx, uint32_t log_n, uint32_t log_blowup, uint32_t n_cols, const uint64_t* d_in,
                                  uint64_t* d_out, void* hip_stream);

/* ---- Poseidon over Goldilocks + Merkle caps ("commit primitives": what a plonky2-style prover does with the LDE'd trace
 * columns -- the reference reaches it through plonky2x `prove`, reference circuits/skip.rs:119-133; plonky2 0.2.0 is an un-vendored
 * dependency, Cargo.lock:2957-2982).  Width 12 (rate 8, capacity 4), S-box x^7, 4 + 22 + 4 rounds, MDS = circulant + diagonal:
 * new[r] = sum_i circ[i] old[(i + r) mod 12] + diag[r] old[r]; hash_no_pad = overwrite-mode sponge; leaf of a row of C columns = the row
 * itself (zero padded) if C <= 4, else hash_no_pad(row); two_to_one(l, r) = permute(l | r | 0000)[0..4).
 * PARITY UNPINNED: plonky2's 360 round constants are not in the reference tree and cannot be recalled, so the default round constants are
 * the Poseidon paper's own Grain-LFSR stream for these parameters (field 1, sbox 0, n 64, t 12, R_F 8, R_P 22) -- NOT plonky2's table; the
 * default MDS is the circulant recalled from plonky2 (17 15 41 16 2 28 13 13 39 18 34 20, diagonal 8 0 .. 0).  tmx_poseidon_set_constants
 * injects the real tables (any argument NULL keeps the current one; values are taken mod p).  Checked against the CPU oracle and
 * an independent Python model; the algebraic self-checks (bijection through the inverse permutation, MDS invertible) are in tests/.
 * tmx_poseidon_merkle_device: n_cols columns of 2^log_n u64 (column c at element c << log_n, e.g. the output of tmx_lde_goldilocks_device),
 * device pointers, asynchronous on hip_stream.  d_levels receives tmx_poseidon_merkle_digests() digests of 4 u64: the 2^log_n leaf digests,
 * then each level above them down to the cap (the last 2^cap_height digests). */
int32_t tmx_poseidon_set_constants(tmx_ctx* ctx, const uint64_t* round_constants /*[360]*/, const uint64_t* mds_circ /*[12]*/,
                                   const uint64_t* mds_diag /*[12]*/);
/* 1 once tmx_poseidon_set_constants has been given round constants, 0 while the context still hashes with the DEFAULT ones -- the Poseidon
 * paper's Grain stream, which can never equal plonky2's table (a seeded ChaCha stream): a cap computed with them is self-consistent and
 * oracle-checked but is NOT the reference prover's commitment.  A host that wants drop-in caps must inject plonky2's ALL_ROUND_CONSTANTS,
 * hand over the extended rows in plonky2's order (it commits bit-reversed, optionally salted LDE rows; this API hashes the columns as given,
 * natural order, no salt) and should assert this returns 1. */
int32_t tmx_poseidon_constants_injected(const tmx_ctx* ctx);
uint64_t tmx_poseidon_merkle_digests(uint32_t log_n, uint32_t cap_height);
int32_t tmx_poseidon_merkle_device(tmx_ctx* ctx, uint32_t log_n, uint32_t n_cols, const uint64_t* d_cols, uint32_t cap_height, uint64_t* d_levels,
                                   void* hip_stream);
/* n permutations of caller-provided states (host buffers, 12 u64 each, blocking): test hook and micro-benchmark */
int32_t tmx_poseidon_permute(tmx_ctx* ctx, uint32_t n, const uint64_t* states_in, uint64_t* states_out);

/* Self-test hook: k_ed_fin inverts with Bernstein-Yang division steps (inv25519.hpp); this runs that inversion and the Fermat chain
 * on n caller-provided values (eight little-endian words each, taken mod 2^255 - 19) and returns both results per value:
 * out_words[16 i .. 16 i + 7] = Fermat, out_words[16 i + 8 .. 16 i + 15] = division steps.  Host buffers, blocking. */
int32_t tmx_selftest_fe_invert(tmx_ctx* ctx, uint32_t n, const uint32_t* in_words, uint32_t* out_words);

/* Self test of the limb-parallel field arithmetic used by the table chain: per item 128 words in (A, B: 4 rows x 16 limbs), 256 words
 * out (A*B | 2^doublings * A as a point | A in ten limbs | B's ten-limb words in 16-bit limbs).  Test hook, not part of the path. */
int32_t tmx_selftest_f16(tmx_ctx* ctx, uint32_t n, uint32_t doublings, const uint32_t* in_words, uint32_t* out_words);

#ifdef __cplusplus
}
#endif
#endif
