/* ORACLE -- TEST INFRASTRUCTURE ONLY.
 * CPU restatement (plain C) of the TendermintX skip/step value-level witness path.  It is the checker the
 * HIP path is compared against (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg) and is never
 * linked, imported or executed by the product library (tendermintx_amd/csrc).
 *
 * The reference is Rust with un-vendored dependencies and cannot be built here (no cargo/rustc): see
 * DESIGN.md "Oracle".  Pinning: reference fixtures + the five CI known-answer tables + RFC 8032 vectors,
 * via tests/golden and oracle/py (pure-Python big-int model).
 */
#ifndef TMXO_H
#define TMXO_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TMXO_KIND_SKIP 0
#define TMXO_KIND_STEP 1
#define TMXO_REC_VALIDATOR 256
#define TMXO_REC_HASHFIELD 48
#define TMXO_REC_HEADER 1136
#define TMXO_REC_PROOF 2336
#define TMXO_FLAG_SIGNED 1
#define TMXO_FLAG_PRESENT 2

typedef struct {
  uint8_t header[32];   /* Level-0 output: target_header / next_header */
  uint32_t all_ok;
  uint32_t fail_mask;   /* bit i = check i failed (order: DESIGN.md "checks") */
  int32_t first_bad_sig;
  uint32_t gt_target;   /* 2/3 verdict on the target set */
  uint32_t gt_trusted;  /* 1/3 verdict on the trusted set (skip only) */
  uint32_t dist_ok;     /* both skip-distance bounds hold (skip only) */
  uint32_t reserved[2];
} tmxo_report;

/* Level-1 EdDSA values of one lane, all canonical little-endian */
typedef struct {
  uint8_t digest[64];
  uint8_t h[32];
  uint8_t pt[10][32]; /* A.x A.y R.x R.y sB.x sB.y hA.x hA.y sum.x sum.y */
  uint32_t ok;
  uint32_t decode_ok;
} tmxo_eddsa_trace;

void tmxo_sha256(const uint8_t* msg, size_t len, uint8_t out[32]);
void tmxo_sha512(const uint8_t* msg, size_t len, uint8_t out[64]);
void tmxo_sha512_3(const uint8_t* p0, size_t l0, const uint8_t* p1, size_t l1, const uint8_t* p2, size_t l2, uint8_t out[64]);

void tmxo_eddsa_trace_lane(const uint8_t pk[32], const uint8_t sig[64], const uint8_t* msg, size_t len, tmxo_eddsa_trace* out);
void tmxo_ed25519_pubkey(const uint8_t seed[32], uint8_t pk[32]);
void tmxo_ed25519_sign(const uint8_t seed[32], const uint8_t* msg, size_t len, uint8_t sig[64]);
void tmxo_sc_reduce512(const uint8_t in[64], uint8_t out[32]);
void tmxo_dummy(uint8_t pk[32], uint8_t sig[64]);

void tmxo_varint9(uint64_t v, uint8_t out[9]);
void tmxo_marshal_validator(const uint8_t pk[32], uint64_t power, uint8_t out[46]);
void tmxo_leaf_hash(const uint8_t* b, size_t len, uint8_t out[32]);
void tmxo_inner_hash(const uint8_t l[32], const uint8_t r[32], uint8_t out[32]);
/* RFC-6962 root over n already-hashed leaves (n >= 1) */
void tmxo_rfc6962_root(const uint8_t* leaf_hashes, size_t n, uint8_t out[32]);
/* fixed-shape in-circuit tree: writes all layer nodes (tree_nodes(n) x 32 B), returns root */
size_t tmxo_tree_nodes(size_t n);
void tmxo_fixed_shape_tree(const uint8_t* leaf_hashes, size_t n, size_t nb_enabled, uint8_t* nodes_out, uint8_t root[32]);
/* threshold: returns gt; fills totals; *no_overflow cleared on any wrap */
int tmxo_tally(const uint64_t* powers, size_t n, size_t nb, const uint8_t* in_group, uint64_t num, uint64_t den,
               uint64_t* tot_prefix, uint64_t* acc_prefix, uint64_t scal[4], int* no_overflow);

int tmxo_is_valid_skip(const uint8_t* start, uint32_t n_start, const uint8_t* target, uint32_t n_target, const uint8_t* sigs, uint32_t n_sigs,
                       uint64_t* shared, uint64_t* total);

/* ---- the typed value of the hint: SkipInputs<F> / StepInputs<F> of the reference (circuits/input/mod.rs:45-74) field by field, as the
 * hint bodies hold it before `write_value` expands it (circuits/skip.rs:85-100, circuits/step.rs:75-87), + the derived Level-1 values in
 * packed form.  The oracle's own statement of the layout the product declares in include/tmx.h (tests compare the two byte for byte, and
 * expand this value back into the H elements of tmxo_witness).  All little-endian, pad bytes zero. */
typedef struct { uint8_t pubkey[32], sig_r[32], sig_s[32], message[124]; uint32_t message_byte_length; uint64_t voting_power; uint32_t validator_byte_length, signed_; } tmxo_validator_value;   /* variables.rs:69-79 */
typedef struct { uint8_t pubkey[32]; uint64_t voting_power; uint32_t validator_byte_length, pad; } tmxo_hashfield_value;                                                             /* variables.rs:82-88 */
typedef struct { uint8_t proof[4][32]; uint32_t enc_chain_id_byte_length; uint8_t chain_id[52]; uint8_t pad[8]; } tmxo_chain_id_proof_value;                                         /* variables.rs:35-41 */
typedef struct { uint8_t proof[4][32]; uint32_t enc_height_byte_length, pad; uint64_t height; } tmxo_height_proof_value;                                                             /* variables.rs:49-55 */
typedef struct { uint8_t proof[4][32]; uint8_t leaf[34]; uint8_t pad[14]; } tmxo_hash_inclusion_proof_value;                                                                         /* input/mod.rs:303-314 */
typedef struct { uint8_t proof[4][32]; uint8_t leaf[72]; uint8_t pad[8]; } tmxo_block_id_inclusion_proof_value;
typedef struct {   /* input/mod.rs:60-74 */
  uint8_t target_header[32], trusted_header[32]; uint64_t round; uint32_t nb_target_validators, nb_trusted_validators;
  tmxo_chain_id_proof_value target_block_chain_id_proof; tmxo_height_proof_value target_block_height_proof;
  tmxo_hash_inclusion_proof_value target_block_validators_hash_proof, trusted_block_validators_hash_proof;
  tmxo_report report;
} tmxo_skip_inputs_fixed;
typedef struct {   /* input/mod.rs:45-58 */
  uint8_t next_header[32]; uint64_t round; uint32_t nb_validators, pad;
  tmxo_chain_id_proof_value next_block_chain_id_proof; tmxo_height_proof_value next_block_height_proof;
  tmxo_hash_inclusion_proof_value next_block_validators_hash_proof; tmxo_block_id_inclusion_proof_value next_block_last_block_id_proof;
  tmxo_hash_inclusion_proof_value prev_block_next_validators_hash_proof;
  tmxo_report report;
} tmxo_step_inputs_fixed;
typedef struct {
  uint8_t sha512_digest[64], h[32], points[10][32]; uint32_t eddsa_ok, decode_ok; uint8_t pad0[24];
  uint8_t marshalled[46], pad1[2], leaf_hash[32], flags[6], pad2[2]; uint64_t total_prefix, signed_prefix; uint8_t pad3[8];
} tmxo_target_lane_derived;
typedef struct { uint8_t marshalled[46], pad1[2], leaf_hash[32], flags[2], pad2[6]; uint64_t total_prefix, matched_prefix; uint8_t pad3[8]; } tmxo_trusted_lane_derived;
typedef struct {
  uint8_t proofs[5][5][32], height_leaf[11], pad0[5]; uint64_t tally_target[4], tally_trusted[4]; uint32_t verdicts[4], checks[16], all_ok, pad1; uint64_t height;
} tmxo_proof_derived;
/* bytes of one proof's value: fixed | validators[n] | hashfields[n] (skip) [| target lanes | trusted lanes (skip) | target nodes | trusted
 * nodes (skip) | proof derived, when with_derived] */
size_t tmxo_value_bytes(int kind, size_t n, int with_derived);
/* tmxo_witness with the typed value as a second output (either output may be NULL) */
int tmxo_witness_value(int kind, const uint8_t* proof_rec, const uint8_t* target_recs, const uint8_t* trusted_recs, uint32_t n,
                       const uint8_t* chain_id, uint32_t chain_id_len, uint64_t skip_max, uint64_t* out, tmxo_report* rep, uint8_t* value, int with_derived);

size_t tmxo_elem_count(int kind, size_t n);
/* one proof; out must hold tmxo_elem_count(kind, n) elements.  trusted_recs ignored for step. returns 0 / <0 */
int tmxo_witness(int kind, const uint8_t* proof_rec, const uint8_t* target_recs, const uint8_t* trusted_recs, uint32_t n,
                 const uint8_t* chain_id, uint32_t chain_id_len, uint64_t skip_max, uint64_t* out, tmxo_report* rep);
/* batch of independent proofs split over n_threads host threads (pthreads); out may be NULL (compute only) */
int tmxo_witness_batch(int kind, uint32_t n_proofs, const uint8_t* proof_recs, const uint8_t* target_recs,
                       const uint8_t* trusted_recs, uint32_t n, const uint8_t* chain_id, uint32_t chain_id_len,
                       uint64_t skip_max, uint64_t* out, tmxo_report* reps, uint32_t n_threads);

/* CPU baseline: persistent pool of n_threads workers over a virtual batch of n_proofs * repeat proofs, compute only; wall seconds */
double tmxo_witness_pool_seconds(int kind, uint32_t n_proofs, const uint8_t* proof_recs, const uint8_t* target_recs, const uint8_t* trusted_recs,
                                 uint32_t n, const uint8_t* chain_id, uint32_t chain_id_len, uint64_t skip_max, uint32_t repeat, uint32_t n_threads);

/* ---- Level-2 trace rows (tmxo_trace.c, DESIGN.md "Level-2 trace rows"): generator with independent affine arithmetic, and the
 * constraint checker (0 = every row satisfies its recurrence and connects to the inputs / Level-1 values; else an error code) */
size_t tmxo_trace_elem_count(int kind, size_t n);
int tmxo_trace(int kind, const uint8_t* proof_rec, const uint8_t* target_recs, const uint8_t* trusted_recs, uint32_t n, uint64_t* out);
long long tmxo_trace_check(int kind, const uint8_t* proof_rec, const uint8_t* target_recs, const uint8_t* trusted_recs, uint32_t n, const uint64_t* trace);
int tmxo_header_proof_messages(int kind, const uint8_t* proof_rec, uint8_t msgs[5][5][96], uint32_t lens[5][5], uint8_t digests[5][5][32]);
void tmxo_trace_ladder(const uint8_t k[32], const uint8_t px[32], const uint8_t py[32], uint64_t* rows);
int tmxo_trace_ladder_check(const uint64_t* rows, const uint8_t k[32], const uint8_t px[32], const uint8_t py[32], const uint8_t rx[32],
                            const uint8_t ry[32]);
void tmxo_base_point(uint8_t x[32], uint8_t y[32]);
void tmxo_trace_sha512(const uint8_t* msg, size_t len, uint64_t* rows);
int tmxo_trace_sha512_check(const uint64_t* rows, const uint8_t* msg, size_t len, const uint8_t digest[64]);
void tmxo_trace_sha256_1(const uint8_t* msg, size_t len, uint64_t* rows);
int tmxo_trace_sha256_1_check(const uint64_t* rows, const uint8_t* msg, size_t len, const uint8_t digest[32]);
void tmxo_trace_sha256_2(const uint8_t* msg, size_t len, uint64_t* rows);
int tmxo_trace_sha256_2_check(const uint64_t* rows, const uint8_t* msg, size_t len, const uint8_t digest[32]);

/* ---- Goldilocks NTT / coset LDE (tmxo_ntt.c; parity unpinned against plonky2, see the file header) */
uint64_t tmxo_gl_pow(uint64_t b, uint64_t e);
uint64_t tmxo_gl_root(uint32_t log_n);
void tmxo_ntt_set_domain(uint64_t root_2_32, uint64_t coset_shift);
void tmxo_ntt(uint64_t* x, uint32_t log_n, int inverse);
void tmxo_lde(const uint64_t* in, uint64_t* out, uint32_t log_n, uint32_t log_blowup);

/* ---- Poseidon over Goldilocks + Merkle caps (tmxo_poseidon.c; SURVEY 8(f) rank 2 "commit primitives"; constants injectable, the defaults
 * are NOT plonky2's: see the file header -- parity unpinned) */
void tmxo_poseidon_grain_constants(uint64_t* out, uint32_t count);
void tmxo_poseidon_set_constants(const uint64_t* rc, const uint64_t* circ, const uint64_t* diag);
void tmxo_poseidon_get_constants(uint64_t* rc, uint64_t* circ, uint64_t* diag);
void tmxo_poseidon_permute(uint64_t state[12]);
void tmxo_poseidon_hash_no_pad(const uint64_t* in, size_t n, uint64_t out[4]);
void tmxo_poseidon_two_to_one(const uint64_t l[4], const uint64_t r[4], uint64_t out[4]);
int tmxo_poseidon_merkle(const uint64_t* cols, uint32_t log_n, uint32_t n_cols, uint32_t cap_height, uint64_t* levels);

#ifdef __cplusplus
}
#endif
#endif
