/* ORACLE -- TEST INFRASTRUCTURE ONLY (CPU restatement; never linked into the product library).
 *
 * Value-level witness of the skip / step circuits: hint elements (H) + every derived Level-1 value (D),
 * appended sequentially as Goldilocks elements (all values < 2^32, hence canonical).
 *
 * Reference map (file:line under /root/reference):
 *   H order, skip     circuits/variables.rs:91-105, circuits/skip.rs:85-100
 *   H order, step     circuits/variables.rs:108-120, circuits/step.rs:73-87
 *   verify_skip       circuits/builder/verify.rs:528-563      verify_step      verify.rs:469-506
 *   verify_header     verify.rs:224-334                        trusted match    verify.rs:361-437
 *   marshal / leaf / set hash   circuits/builder/validator.rs:185-252, shared.rs:67-156
 *   sig-data checks   validator.rs:80-183                      tally            circuits/builder/voting.rs:31-109
 *   chain id / height verify.rs:180-222, shared.rs:169-207     header proofs    circuits/input/tendermint_utils.rs:214-393
 * Layout of D and of the check bits: DESIGN.md "Witness layout".
 */
#define _POSIX_C_SOURCE 200809L /* pthread barriers, clock_gettime (the CPU-baseline pool) */
#include "tmxo.h"
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ------------------------------------------------------------------ encodings + Merkle */
void tmxo_varint9(uint64_t v, uint8_t out[9]) {
  /* shared.rs:67-156: nine septets; continuation bit iff index < last non-zero septet; trailing zeros */
  int last = 0;
  for (int i = 0; i < 9; i++) if ((v >> (7 * i)) & 0x7f) last = i;
  for (int i = 0; i < 9; i++) out[i] = (uint8_t)(((v >> (7 * i)) & 0x7f) | (i < last ? 0x80 : 0));
}
void tmxo_marshal_validator(const uint8_t pk[32], uint64_t power, uint8_t out[46]) {
  /* validator.rs:185-207: 0a 22 0a 20 pk 10 varint9 */
  out[0] = 0x0a; out[1] = 0x22; out[2] = 0x0a; out[3] = 0x20;
  memcpy(out + 4, pk, 32);
  out[36] = 0x10;
  tmxo_varint9(power, out + 37);
}
void tmxo_leaf_hash(const uint8_t* b, size_t len, uint8_t out[32]) {
  uint8_t buf[1 + 128];
  buf[0] = 0x00; memcpy(buf + 1, b, len);
  tmxo_sha256(buf, len + 1, out);
}
void tmxo_inner_hash(const uint8_t l[32], const uint8_t r[32], uint8_t out[32]) {
  uint8_t buf[65];
  buf[0] = 0x01; memcpy(buf + 1, l, 32); memcpy(buf + 33, r, 32);
  tmxo_sha256(buf, 65, out);
}
static size_t split_point(size_t n) { /* tendermint_utils.rs:338-349 */
  size_t k = 1;
  while (k * 2 < n) k *= 2;
  return k;
}
void tmxo_rfc6962_root(const uint8_t* lh, size_t n, uint8_t out[32]) {
  if (n == 1) { memcpy(out, lh, 32); return; }
  size_t k = split_point(n);
  uint8_t l[32], r[32];
  tmxo_rfc6962_root(lh, k, l); tmxo_rfc6962_root(lh + 32 * k, n - k, r);
  tmxo_inner_hash(l, r, out);
}
/* aunts (leaf->root) of leaf `index` among n hashed leaves; returns depth */
static int rfc6962_aunts(const uint8_t* lh, size_t n, size_t index, uint8_t aunts[][32]) {
  if (n == 1) return 0;
  size_t k = split_point(n);
  int d;
  if (index < k) { d = rfc6962_aunts(lh, k, index, aunts); tmxo_rfc6962_root(lh + 32 * k, n - k, aunts[d]); }
  else { d = rfc6962_aunts(lh + 32 * k, n - k, index - k, aunts); tmxo_rfc6962_root(lh, k, aunts[d]); }
  return d + 1;
}
size_t tmxo_tree_nodes(size_t n) { size_t c = 0; while (n > 1) { n = (n + 1) / 2; c += n; } return c; }

void tmxo_fixed_shape_tree(const uint8_t* leaf_hashes, size_t n, size_t nb, uint8_t* nodes_out, uint8_t root[32]) {
  /* get_root_from_hashed_leaves value semantics (call: validator.rs:248-251): see oracle/py/tm_encoding.py */
  uint8_t* cur = (uint8_t*)malloc(32 * n); uint8_t* en = (uint8_t*)malloc(n);
  memcpy(cur, leaf_hashes, 32 * n);
  for (size_t i = 0; i < n; i++) en[i] = i < nb;
  size_t sz = n;
  while (sz > 1) {
    size_t nx = (sz + 1) / 2;
    for (size_t i = 0; i < nx; i++) {
      uint8_t nd[32];
      if (2 * i + 1 < sz && en[2 * i] && en[2 * i + 1]) tmxo_inner_hash(cur + 64 * i, cur + 64 * i + 32, nd);
      else memcpy(nd, cur + 64 * i, 32);
      en[i] = en[2 * i];
      memcpy(cur + 32 * i, nd, 32);
      if (nodes_out) { memcpy(nodes_out, nd, 32); nodes_out += 32; }
    }
    sz = nx;
  }
  memcpy(root, cur, 32);
  free(cur); free(en);
}

int tmxo_tally(const uint64_t* powers, size_t n, size_t nb, const uint8_t* in_group, uint64_t num, uint64_t den,
               uint64_t* tot_prefix, uint64_t* acc_prefix, uint64_t scal[4], int* no_overflow) {
  uint64_t total = 0, acc = 0;
  int enabled = 1;
  for (size_t i = 0; i < n; i++) {          /* voting.rs:31-63 */
    if (i == nb) enabled = 0;
    uint64_t t2 = total + (enabled ? powers[i] : 0);
    if (t2 < total) *no_overflow = 0;
    total = t2;
    if (tot_prefix) tot_prefix[i] = total;
  }
  for (size_t i = 0; i < n; i++) {          /* voting.rs:79-89 */
    uint64_t a2 = acc + (in_group[i] ? powers[i] : 0);
    if (a2 < acc) *no_overflow = 0;
    acc = a2;
    if (acc_prefix) acc_prefix[i] = acc;
  }
  uint64_t sa = acc * den, st = total * num; /* voting.rs:91-105: wrapping mul checked by division */
  if (sa / den != acc) *no_overflow = 0;
  if (st / num != total) *no_overflow = 0;
  scal[0] = total; scal[1] = acc; scal[2] = sa; scal[3] = st;
  return sa > st;                            /* voting.rs:108 */
}

/* ------------------------------------------------------------------ element stream */
typedef struct { uint64_t* p; size_t n; } es;
static void e_byte(es* e, uint8_t b) { for (int k = 7; k >= 0; k--) { if (e->p) e->p[e->n] = (b >> k) & 1; e->n++; } }
static void e_bytes(es* e, const uint8_t* b, size_t len) { for (size_t i = 0; i < len; i++) e_byte(e, b[i]); }
static void e_u32(es* e, uint64_t v) { if (e->p) e->p[e->n] = v & 0xffffffffu; e->n++; }
static void e_u64(es* e, uint64_t v) { e_u32(e, v); e_u32(e, v >> 32); }
static void e_u256le(es* e, const uint8_t b[32]) {
  for (int k = 0; k < 8; k++) e_u32(e, (uint64_t)b[4 * k] | ((uint64_t)b[4 * k + 1] << 8) | ((uint64_t)b[4 * k + 2] << 16) | ((uint64_t)b[4 * k + 3] << 24));
}
static void e_bool(es* e, int b) { e_u32(e, b ? 1 : 0); }

static uint64_t rd64(const uint8_t* p) { uint64_t v = 0; for (int k = 7; k >= 0; k--) v = (v << 8) | p[k]; return v; }
static uint32_t rd32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

typedef struct {
  uint8_t len[14];
  const uint8_t* leaf[14];
  uint8_t lh[14][32];
  uint8_t root[32];
} hdr;

static void hdr_load(hdr* h, const uint8_t* rec) {
  /* len[] keeps the raw length byte (it is a hint element); hashing and copies use at most the 79 bytes a slot can hold */
  for (int i = 0; i < 14; i++) { h->len[i] = rec[i]; h->leaf[i] = rec + 16 + 80 * i; tmxo_leaf_hash(h->leaf[i], rec[i] > 79 ? 79 : rec[i], h->lh[i]); }
  tmxo_rfc6962_root(&h->lh[0][0], 14, h->root);
}
/* walk a depth-4 proof (tendermint_utils.rs:214-224; path bits LSB-first per shared.rs:45-65) */
static void proof_walk(const uint8_t leaf_hash[32], int index, uint8_t aunts[4][32], uint8_t nodes[4][32]) {
  uint8_t cur[32]; memcpy(cur, leaf_hash, 32);
  for (int k = 0; k < 4; k++) {
    if ((index >> k) & 1) tmxo_inner_hash(aunts[k], cur, nodes[k]); else tmxo_inner_hash(cur, aunts[k], nodes[k]);
    memcpy(cur, nodes[k], 32);
  }
}
static void emit_proof_d(es* e, const uint8_t lh[32], uint8_t nodes[4][32]) { e_bytes(e, lh, 32); for (int k = 0; k < 4; k++) e_bytes(e, nodes[k], 32); }

static uint64_t height_from_leaf(const uint8_t* leaf, int len) {
  /* inverse of `08 varint(height)`; a varint has at most 10 bytes, anything longer is ignored */
  uint64_t x = 0; int s = 0;
  for (int i = 1; i < len && i <= 10; i++) { x |= (uint64_t)(leaf[i] & 0x7f) << s; s += 7; }
  return x;
}

#define DT 1235
#define DR 630
#define PROOF_D 1280
size_t tmxo_elem_count(int kind, size_t n) {
  if (kind == TMXO_KIND_SKIP) return 1776 * n + 5320 + n * DT + n * DR + 2 * tmxo_tree_nodes(n) * 256 + (4 * PROOF_D + 88) + 34;
  return 1517 * n + 6919 + n * DT + tmxo_tree_nodes(n) * 256 + (5 * PROOF_D + 88) + 25;
}

/* where the parts of a proof's typed value start (tmxo.h) */
typedef struct { size_t fixed, validators, hashfields, lane_t, lane_r, nodes_t, nodes_r, proof_d, bytes; } vlay;
static vlay value_lay(int kind, size_t n, int with_derived) {
  vlay L; size_t o = 0; const int skip = kind == TMXO_KIND_SKIP; const size_t tn = tmxo_tree_nodes(n);
  L.fixed = o; o += skip ? sizeof(tmxo_skip_inputs_fixed) : sizeof(tmxo_step_inputs_fixed);
  L.validators = o; o += n * sizeof(tmxo_validator_value);
  L.hashfields = o; o += skip ? n * sizeof(tmxo_hashfield_value) : 0;
  L.lane_t = o; o += with_derived ? n * sizeof(tmxo_target_lane_derived) : 0;
  L.lane_r = o; o += with_derived && skip ? n * sizeof(tmxo_trusted_lane_derived) : 0;
  L.nodes_t = o; o += with_derived ? 32 * tn : 0;
  L.nodes_r = o; o += with_derived && skip ? 32 * tn : 0;
  L.proof_d = o; o += with_derived ? sizeof(tmxo_proof_derived) : 0;
  L.bytes = o;
  return L;
}
size_t tmxo_value_bytes(int kind, size_t n, int with_derived) { return value_lay(kind, n, with_derived).bytes; }
static void put_proof(uint8_t dst[4][32], uint8_t aunts[4][32]) { memcpy(dst, aunts, 128); }

int tmxo_witness(int kind, const uint8_t* prec, const uint8_t* trec, const uint8_t* rrec, uint32_t n,
                 const uint8_t* chain_id, uint32_t chain_id_len, uint64_t skip_max, uint64_t* out, tmxo_report* rep) {
  return tmxo_witness_value(kind, prec, trec, rrec, n, chain_id, chain_id_len, skip_max, out, rep, NULL, 0);
}

int tmxo_witness_value(int kind, const uint8_t* prec, const uint8_t* trec, const uint8_t* rrec, uint32_t n,
                       const uint8_t* chain_id, uint32_t chain_id_len, uint64_t skip_max, uint64_t* out, tmxo_report* rep_out, uint8_t* value, int with_derived) {
  if (n == 0 || n > 4096) return -1;
  tmxo_report rep_local;
  tmxo_report* rep = rep_out ? rep_out : &rep_local;   /* (the typed value carries the report too) */
  const vlay VL = value_lay(kind, n, with_derived);
  if (value) memset(value, 0, VL.bytes);
  tmxo_skip_inputs_fixed* vsk = value && kind == TMXO_KIND_SKIP ? (tmxo_skip_inputs_fixed*)(value + VL.fixed) : NULL;
  tmxo_step_inputs_fixed* vst = value && kind == TMXO_KIND_STEP ? (tmxo_step_inputs_fixed*)(value + VL.fixed) : NULL;
  tmxo_target_lane_derived* vlt = value && with_derived ? (tmxo_target_lane_derived*)(value + VL.lane_t) : NULL;
  tmxo_trusted_lane_derived* vlr = value && with_derived && kind == TMXO_KIND_SKIP ? (tmxo_trusted_lane_derived*)(value + VL.lane_r) : NULL;
  tmxo_proof_derived* vpd = value && with_derived ? (tmxo_proof_derived*)(value + VL.proof_d) : NULL;
  es E = {out, 0};
  uint64_t block_a = rd64(prec), block_b = rd64(prec + 8), round = rd64(prec + 48);
  const uint8_t* pub_hash = prec + 16;
  uint32_t nb = rd32(prec + 56), nbt = rd32(prec + 60);
  hdr ha, hb;
  hdr_load(&ha, prec + 64); hdr_load(&hb, prec + 64 + TMXO_REC_HEADER);
  const uint8_t* header = ha.root;
  uint64_t height_a = height_from_leaf(ha.leaf[2], ha.len[2]);
  uint8_t dummy_pk[32], dummy_sig[64], zero_msg[32] = {0};
  tmxo_dummy(dummy_pk, dummy_sig);

  uint8_t a_cid[4][32], a_h[4][32], a_v[4][32], a_tv[4][32], a_lb[4][32], a_nv[4][32];
  rfc6962_aunts(&ha.lh[0][0], 14, 1, a_cid); rfc6962_aunts(&ha.lh[0][0], 14, 2, a_h); rfc6962_aunts(&ha.lh[0][0], 14, 7, a_v);
  uint8_t leaf34[34], leaf72[72], leafb[34];
  if (value) {   /* the two Vecs of SkipInputs / StepInputs: conversion.rs:59-178 */
    tmxo_validator_value* vv = (tmxo_validator_value*)(value + VL.validators);
    for (uint32_t i = 0; i < n; i++) {
      const uint8_t* v = trec + (size_t)TMXO_REC_VALIDATOR * i;
      memcpy(vv[i].pubkey, v, 32); memcpy(vv[i].sig_r, v + 32, 32); memcpy(vv[i].sig_s, v + 64, 32); memcpy(vv[i].message, v + 96, 124);
      vv[i].message_byte_length = (uint32_t)v[220] | ((uint32_t)v[221] << 8); vv[i].voting_power = rd64(v + 224);
      vv[i].validator_byte_length = v[222]; vv[i].signed_ = v[223] & TMXO_FLAG_SIGNED;
    }
    if (kind == TMXO_KIND_SKIP) {
      tmxo_hashfield_value* hv = (tmxo_hashfield_value*)(value + VL.hashfields);
      for (uint32_t j = 0; j < n; j++) { const uint8_t* t = rrec + (size_t)TMXO_REC_HASHFIELD * j; memcpy(hv[j].pubkey, t, 32); hv[j].voting_power = rd64(t + 32); hv[j].validator_byte_length = t[40]; }
    }
  }

  /* ---------------- H */
  e_bytes(&E, header, 32);
  for (uint32_t i = 0; i < n; i++) {
    const uint8_t* v = trec + (size_t)TMXO_REC_VALIDATOR * i;
    e_bytes(&E, v, 32); e_bytes(&E, v + 32, 32); e_u256le(&E, v + 64); e_bytes(&E, v + 96, 124);
    e_u32(&E, (uint64_t)v[220] | ((uint64_t)v[221] << 8)); e_u64(&E, rd64(v + 224)); e_u32(&E, v[222]); e_bool(&E, v[223] & TMXO_FLAG_SIGNED);
  }
  e_u32(&E, nb); e_u64(&E, round);
  for (int k = 0; k < 4; k++) e_bytes(&E, a_cid[k], 32);
  e_u32(&E, ha.len[1]);
  { uint8_t cid52[52] = {0}; memcpy(cid52, ha.leaf[1], ha.len[1] < 52 ? ha.len[1] : 52); e_bytes(&E, cid52, 52); }
  for (int k = 0; k < 4; k++) e_bytes(&E, a_h[k], 32);
  e_u32(&E, ha.len[2]); e_u64(&E, height_a);
  memset(leaf34, 0, 34); memcpy(leaf34, ha.leaf[7], ha.len[7] < 34 ? ha.len[7] : 34);
  for (int k = 0; k < 4; k++) e_bytes(&E, a_v[k], 32);
  e_bytes(&E, leaf34, 34);
  if (value) {   /* input/mod.rs:471-499 (skip), 370-398 (step): the three proofs both kinds carry */
    tmxo_chain_id_proof_value* cp = vsk ? &vsk->target_block_chain_id_proof : &vst->next_block_chain_id_proof;
    tmxo_height_proof_value* hp = vsk ? &vsk->target_block_height_proof : &vst->next_block_height_proof;
    tmxo_hash_inclusion_proof_value* vp = vsk ? &vsk->target_block_validators_hash_proof : &vst->next_block_validators_hash_proof;
    put_proof(cp->proof, a_cid); cp->enc_chain_id_byte_length = ha.len[1]; memcpy(cp->chain_id, ha.leaf[1], ha.len[1] < 52 ? ha.len[1] : 52);
    put_proof(hp->proof, a_h); hp->enc_height_byte_length = ha.len[2]; hp->height = height_a;
    put_proof(vp->proof, a_v); memcpy(vp->leaf, leaf34, 34);
    if (vsk) { memcpy(vsk->target_header, header, 32); memcpy(vsk->trusted_header, pub_hash, 32); vsk->round = round; vsk->nb_target_validators = nb; vsk->nb_trusted_validators = nbt; }
    else { memcpy(vst->next_header, header, 32); vst->round = round; vst->nb_validators = nb; }
  }
  if (kind == TMXO_KIND_SKIP) {
    rfc6962_aunts(&hb.lh[0][0], 14, 7, a_tv);
    memset(leafb, 0, 34); memcpy(leafb, hb.leaf[7], hb.len[7] < 34 ? hb.len[7] : 34);
    e_u32(&E, nbt);
    for (int k = 0; k < 4; k++) e_bytes(&E, a_tv[k], 32);
    e_bytes(&E, leafb, 34);
    if (vsk) { put_proof(vsk->trusted_block_validators_hash_proof.proof, a_tv); memcpy(vsk->trusted_block_validators_hash_proof.leaf, leafb, 34); }
    for (uint32_t j = 0; j < n; j++) { const uint8_t* t = rrec + (size_t)TMXO_REC_HASHFIELD * j; e_bytes(&E, t, 32); e_u64(&E, rd64(t + 32)); e_u32(&E, t[40]); }
  } else {
    rfc6962_aunts(&ha.lh[0][0], 14, 4, a_lb); rfc6962_aunts(&hb.lh[0][0], 14, 8, a_nv);
    memset(leaf72, 0, 72); memcpy(leaf72, ha.leaf[4], ha.len[4] < 72 ? ha.len[4] : 72);
    memset(leafb, 0, 34); memcpy(leafb, hb.leaf[8], hb.len[8] < 34 ? hb.len[8] : 34);
    for (int k = 0; k < 4; k++) e_bytes(&E, a_lb[k], 32);
    e_bytes(&E, leaf72, 72);
    for (int k = 0; k < 4; k++) e_bytes(&E, a_nv[k], 32);
    e_bytes(&E, leafb, 34);
    if (vst) {
      put_proof(vst->next_block_last_block_id_proof.proof, a_lb); memcpy(vst->next_block_last_block_id_proof.leaf, leaf72, 72);
      put_proof(vst->prev_block_next_validators_hash_proof.proof, a_nv); memcpy(vst->prev_block_next_validators_hash_proof.leaf, leafb, 34);
    }
  }

  /* ---------------- D */
  uint64_t expected_height = block_b;
  uint64_t* powers = (uint64_t*)malloc(8 * n); uint8_t* signedv = (uint8_t*)malloc(n);
  uint64_t* totp = (uint64_t*)malloc(8 * n); uint64_t* accp = (uint64_t*)malloc(8 * n);
  uint8_t* leaves = (uint8_t*)malloc(32 * n); uint8_t* nodes = (uint8_t*)malloc(32 * (tmxo_tree_nodes(n) + 1));
  int no_overflow = 1, varint_ok = 1;
  for (uint32_t i = 0; i < n; i++) { const uint8_t* v = trec + (size_t)TMXO_REC_VALIDATOR * i; powers[i] = rd64(v + 224); signedv[i] = v[223] & TMXO_FLAG_SIGNED; if (powers[i] >> 63) varint_ok = 0; }
  if (height_a >> 63) varint_ok = 0;   /* verify_block_height marshals height_proof.height through the same varint gadget (shared.rs:178, :80) */
  /* verify_non_negative_round (validator.rs:73-78), asserted once per lane inside verify_validator_signature_data (:141) on the
   * proof-wide round whether or not the lane signed: bit 7 of the most significant byte of LE64(round) must be 0 */
  const int round_nonneg = (round >> 63) == 0;
  uint64_t scal_t[4], scal_r[4] = {0, 0, 0, 0};
  int gt_t = tmxo_tally(powers, n, nb, signedv, 2, 3, totp, accp, scal_t, &no_overflow), gt_r = 0;
  int all_eddsa = 1, all_sigdata = 1; int32_t first_bad = -1;
  tmxo_eddsa_trace* trs = (tmxo_eddsa_trace*)malloc(sizeof(tmxo_eddsa_trace) * n);
  uint8_t* lflags = (uint8_t*)malloc(6 * n);
  /* D.1a: the byte fields of every target lane (marshalled validator, leaf hash, SHA-512 digest) */
  for (uint32_t i = 0; i < n; i++) {
    const uint8_t* v = trec + (size_t)TMXO_REC_VALIDATOR * i;
    uint8_t m[46], lh[32];
    tmxo_marshal_validator(v, powers[i], m);
    tmxo_leaf_hash(m, v[222] > 46 ? 46 : v[222], lh);    /* validator.rs:209-229: 1 + vlen bytes (vlen <= 46 by type) */
    memcpy(leaves + 32 * i, lh, 32);
    tmxo_eddsa_trace* tr = &trs[i];
    size_t mlen = (size_t)v[220] | ((size_t)v[221] << 8);
    if (mlen > 124) mlen = 124;                              /* message buffer is 124 bytes (consts.rs:29) */
    if (signedv[i]) tmxo_eddsa_trace_lane(v, v + 32, v + 96, mlen, tr);
    else tmxo_eddsa_trace_lane(dummy_pk, dummy_sig, zero_msg, 32, tr);   /* conditional substitution (verify.rs:248-259) */
    const uint8_t* msg = v + 96;
    int enabled = i < nb;
    int off = round == 0 ? 16 : 25;
    int hash_in_msg = memcmp(msg + off, header, 32) == 0;                   /* validator.rs:155-183 */
    int is_precommit = msg[1] == 8 && msg[2] == 2;                          /* validator.rs:100-109 */
    int height_ok = rd64(msg + 4) == expected_height;                       /* validator.rs:111-123 */
    int round_ok = round == 0 ? 1 : (rd64(msg + 13) == round);              /* validator.rs:125-141 */
    int valid = signedv[i] && enabled && hash_in_msg && is_precommit && height_ok && round_ok;
    int sigdata_ok = (signedv[i] != 0) == valid;                            /* validator.rs:143-152 */
    uint8_t* f = lflags + 6 * i;
    f[0] = (uint8_t)enabled; f[1] = (uint8_t)hash_in_msg; f[2] = (uint8_t)is_precommit; f[3] = (uint8_t)height_ok; f[4] = (uint8_t)round_ok; f[5] = (uint8_t)sigdata_ok;
    e_bytes(&E, m, 46); e_bytes(&E, lh, 32); e_bytes(&E, tr->digest, 64);
    if (vlt) {
      tmxo_target_lane_derived* d = &vlt[i];
      memcpy(d->sha512_digest, tr->digest, 64); memcpy(d->h, tr->h, 32); memcpy(d->points, tr->pt, 320); d->eddsa_ok = tr->ok; d->decode_ok = tr->decode_ok;
      memcpy(d->marshalled, m, 46); memcpy(d->leaf_hash, lh, 32); memcpy(d->flags, f, 6); d->total_prefix = totp[i]; d->signed_prefix = accp[i];
    }
    if (!tr->ok) { all_eddsa = 0; if (first_bad < 0) first_bad = (int32_t)i; }
    if (!sigdata_ok) all_sigdata = 0;
  }
  /* D.1b: the word elements of every target lane (h, the ten coordinates, the EdDSA verdict, the six flags, the two prefix sums) */
  for (uint32_t i = 0; i < n; i++) {
    const tmxo_eddsa_trace* tr = &trs[i];
    e_u256le(&E, tr->h);
    for (int k = 0; k < 10; k++) e_u256le(&E, tr->pt[k]);
    e_bool(&E, tr->ok);
    for (int k = 0; k < 6; k++) e_bool(&E, lflags[6 * i + k]);
    e_u64(&E, totp[i]); e_u64(&E, accp[i]);
  }
  free(trs); free(lflags);
  uint8_t root_t[32], root_r[32];
  tmxo_fixed_shape_tree(leaves, n, nb, nodes, root_t);
  size_t tn = tmxo_tree_nodes(n);
  uint8_t* nodes_r = NULL;
  if (kind == TMXO_KIND_SKIP) {
    uint8_t* matched = (uint8_t*)malloc(n); uint64_t* rp = (uint64_t*)malloc(8 * n);
    uint8_t* rleaves = (uint8_t*)malloc(32 * n); uint8_t* rm = (uint8_t*)malloc(46 * n);
    nodes_r = (uint8_t*)malloc(32 * (tn + 1));
    for (uint32_t j = 0; j < n; j++) {
      const uint8_t* t = rrec + (size_t)TMXO_REC_HASHFIELD * j;
      rp[j] = rd64(t + 32); if (rp[j] >> 63) varint_ok = 0;
      matched[j] = 0;
      for (uint32_t i = 0; i < n; i++)                                  /* verify.rs:398-418 */
        if (signedv[i] && memcmp(trec + (size_t)TMXO_REC_VALIDATOR * i, t, 32) == 0) matched[j] = 1;
      tmxo_marshal_validator(t, rp[j], rm + 46 * j);
      tmxo_leaf_hash(rm + 46 * j, t[40] > 46 ? 46 : t[40], rleaves + 32 * j);
    }
    gt_r = tmxo_tally(rp, n, nbt, matched, 1, 3, totp, accp, scal_r, &no_overflow);
    for (uint32_t j = 0; j < n; j++) { e_bytes(&E, rm + 46 * j, 46); e_bytes(&E, rleaves + 32 * j, 32); }                    /* D.2a: byte fields */
    for (uint32_t j = 0; j < n; j++) { e_bool(&E, j < nbt); e_bool(&E, matched[j]); e_u64(&E, totp[j]); e_u64(&E, accp[j]); }  /* D.2b: word fields */
    if (vlr)
      for (uint32_t j = 0; j < n; j++) {
        memcpy(vlr[j].marshalled, rm + 46 * j, 46); memcpy(vlr[j].leaf_hash, rleaves + 32 * j, 32);
        vlr[j].flags[0] = j < nbt; vlr[j].flags[1] = matched[j]; vlr[j].total_prefix = totp[j]; vlr[j].matched_prefix = accp[j];
      }
    tmxo_fixed_shape_tree(rleaves, n, nbt, nodes_r, root_r);
    free(matched); free(rp); free(rleaves); free(rm);
  }
  e_bytes(&E, nodes, 32 * tn);
  if (kind == TMXO_KIND_SKIP) e_bytes(&E, nodes_r, 32 * tn);
  if (value && with_derived) { memcpy(value + VL.nodes_t, nodes, 32 * tn); if (kind == TMXO_KIND_SKIP) memcpy(value + VL.nodes_r, nodes_r, 32 * tn); }

  uint8_t n_cid[4][32], n_h[4][32], n_v[4][32], n_x[4][32], n_y[4][32], hl[96] = {0}, hlh[32], vlh[32], xlh[32], ylh[32];
  /* verify.rs:189-202: SHA-256 over 1 + enc_len bytes of 00 | chain_id[52] | zeros, where chain_id is the encoded field resized to
   * 52 bytes (input/mod.rs:476-478 -- a longer field is truncated there, and its leaf is then not the header's) */
  uint8_t cid_ext[81] = {0}, cid_lh[32];
  memcpy(cid_ext + 1, ha.leaf[1], ha.len[1] < 52 ? ha.len[1] : 52);
  tmxo_sha256(cid_ext, (size_t)(ha.len[1] > 79 ? 79 : ha.len[1]) + 1, cid_lh);
  proof_walk(cid_lh, 1, a_cid, n_cid);                                    /* verify.rs:203-209 */
  hl[0] = 0x00; hl[1] = 0x08; tmxo_varint9(height_a, hl + 2);             /* shared.rs:158-167 */
  tmxo_leaf_hash(hl + 1, ha.len[2] > 79 ? 79 : ha.len[2], hlh);                                  /* shared.rs:183-194: 1 + enc_len bytes */
  proof_walk(hlh, 2, a_h, n_h);
  tmxo_leaf_hash(leaf34, 34, vlh); proof_walk(vlh, 7, a_v, n_v);
  emit_proof_d(&E, cid_lh, n_cid);
  e_bytes(&E, hl, 11); emit_proof_d(&E, hlh, n_h);
  emit_proof_d(&E, vlh, n_v);
  uint8_t cid52[52] = {0}; memcpy(cid52, ha.leaf[1], ha.len[1] < 52 ? ha.len[1] : 52);
  int chain_ok = chain_id_len <= 50 && memcmp(cid52 + 2, chain_id, chain_id_len) == 0;   /* verify.rs:211-221 */
  int checks[16]; int nchk = 0; int all_ok = 1;
  if (kind == TMXO_KIND_SKIP) {
    tmxo_leaf_hash(leafb, 34, xlh); proof_walk(xlh, 7, a_tv, n_x);
    emit_proof_d(&E, xlh, n_x);
    for (int k = 0; k < 4; k++) e_u64(&E, scal_t[k]);
    e_bool(&E, gt_t);
    for (int k = 0; k < 4; k++) e_u64(&E, scal_r[k]);
    e_bool(&E, gt_r);
    int dist_gt = block_b > block_a + 1, dist_le = block_b <= block_a + skip_max;     /* verify.rs:508-526 */
    e_bool(&E, dist_gt); e_bool(&E, dist_le);
    checks[nchk++] = memcmp(n_x[3], pub_hash, 32) == 0;     /* verify.rs:374-379 */
    checks[nchk++] = memcmp(root_r, leafb + 2, 32) == 0;    /* verify.rs:382-389 */
    checks[nchk++] = memcmp(root_t, leaf34 + 2, 32) == 0;   /* verify.rs:279-280 */
    checks[nchk++] = memcmp(n_v[3], header, 32) == 0;       /* verify.rs:283-286 */
    checks[nchk++] = memcmp(n_cid[3], header, 32) == 0;     /* verify.rs:205-209 */
    checks[nchk++] = chain_ok;
    checks[nchk++] = memcmp(n_h[3], header, 32) == 0;       /* shared.rs:197-203 */
    checks[nchk++] = height_a == expected_height;           /* shared.rs:206 */
    checks[nchk++] = all_sigdata; checks[nchk++] = all_eddsa; checks[nchk++] = no_overflow; checks[nchk++] = varint_ok;
    checks[nchk++] = round_nonneg;                          /* validator.rs:73-78 */
    for (int k = 0; k < nchk; k++) { e_bool(&E, checks[k]); all_ok = all_ok && checks[k]; }
    all_ok = all_ok && gt_t && gt_r && dist_gt && dist_le;
    e_bool(&E, all_ok);
    if (rep) rep->dist_ok = (uint32_t)(dist_gt && dist_le);
  } else {
    tmxo_leaf_hash(leaf72, 72, xlh); proof_walk(xlh, 4, a_lb, n_x);
    emit_proof_d(&E, xlh, n_x);
    tmxo_leaf_hash(leafb, 34, ylh); proof_walk(ylh, 8, a_nv, n_y);
    emit_proof_d(&E, ylh, n_y);
    for (int k = 0; k < 4; k++) e_u64(&E, scal_t[k]);
    e_bool(&E, gt_t);
    checks[nchk++] = memcmp(root_t, leaf34 + 2, 32) == 0;
    checks[nchk++] = memcmp(n_v[3], header, 32) == 0;
    checks[nchk++] = memcmp(n_cid[3], header, 32) == 0;
    checks[nchk++] = chain_ok;
    checks[nchk++] = memcmp(n_h[3], header, 32) == 0;
    checks[nchk++] = height_a == expected_height;
    checks[nchk++] = all_sigdata; checks[nchk++] = all_eddsa; checks[nchk++] = no_overflow; checks[nchk++] = varint_ok;
    checks[nchk++] = memcmp(n_x[3], header, 32) == 0;       /* verify.rs:144-147 */
    checks[nchk++] = memcmp(leaf72 + 2, pub_hash, 32) == 0; /* verify.rs:150-153 */
    checks[nchk++] = memcmp(n_y[3], pub_hash, 32) == 0;     /* verify.rs:166-170 */
    checks[nchk++] = memcmp(leaf34 + 2, leafb + 2, 32) == 0;/* verify.rs:173-177 */
    checks[nchk++] = round_nonneg;                          /* validator.rs:73-78 */
    for (int k = 0; k < nchk; k++) { e_bool(&E, checks[k]); all_ok = all_ok && checks[k]; }
    all_ok = all_ok && gt_t;
    e_bool(&E, all_ok);
    if (rep) rep->dist_ok = 0;
  }
  {
    memcpy(rep->header, header, 32);
    rep->all_ok = (uint32_t)all_ok; rep->fail_mask = 0;
    for (int k = 0; k < nchk; k++) if (!checks[k]) rep->fail_mask |= 1u << k;
    rep->first_bad_sig = first_bad; rep->gt_target = (uint32_t)gt_t; rep->gt_trusted = (uint32_t)gt_r;
    rep->reserved[0] = (nb > n ? 1u : 0u) | (kind == TMXO_KIND_SKIP && nbt > n ? 2u : 0u);   /* precond: input/mod.rs:439-444, 338-342 */
    rep->reserved[1] = 0;
  }
  if (vpd) {
    uint8_t (*nd[5])[32] = {n_cid, n_h, n_v, n_x, n_y};
    const uint8_t* lhs[5] = {cid_lh, hlh, vlh, xlh, ylh};
    for (int q = 0; q < (kind == TMXO_KIND_SKIP ? 4 : 5); q++) { memcpy(vpd->proofs[q][0], lhs[q], 32); memcpy(vpd->proofs[q][1], nd[q], 128); }
    memcpy(vpd->height_leaf, hl, 11);
    for (int k = 0; k < 4; k++) { vpd->tally_target[k] = scal_t[k]; vpd->tally_trusted[k] = scal_r[k]; }
    vpd->verdicts[0] = (uint32_t)gt_t;
    if (kind == TMXO_KIND_SKIP) { vpd->verdicts[1] = (uint32_t)gt_r; vpd->verdicts[2] = block_b > block_a + 1; vpd->verdicts[3] = block_b <= block_a + skip_max; }
    for (int k = 0; k < nchk; k++) vpd->checks[k] = (uint32_t)checks[k];
    vpd->all_ok = (uint32_t)all_ok; vpd->height = height_a;
  }
  if (vsk) vsk->report = *rep;
  if (vst) vst->report = *rep;
  free(powers); free(signedv); free(totp); free(accp); free(leaves); free(nodes); free(nodes_r);
  return E.n == tmxo_elem_count(kind, n) ? 0 : -2;
}

/* The SHA-256 messages behind the header proofs of one proof record, in the order Level-1 emits the proofs (chain id, height, validators
 * hash, X = trusted next-validators hash (skip) / last block id (step), Y = next validators hash (step)): per proof the leaf message
 * (00 | leaf as the proof struct carries it, cut at 1 + enc_len) and the four path-node messages (01 | left | right, path bits LSB-first),
 * exactly the values tmxo_witness hashes above.  Level-2 section T.6 traces them.  Returns the number of proofs (4 / 5). */
int tmxo_header_proof_messages(int kind, const uint8_t* prec, uint8_t msgs[5][5][96], uint32_t lens[5][5], uint8_t digests[5][5][32]) {
  hdr ha, hb;
  hdr_load(&ha, prec + 64); hdr_load(&hb, prec + 64 + TMXO_REC_HEADER);
  const uint64_t height_a = height_from_leaf(ha.leaf[2], ha.len[2]);
  const int nq = kind == TMXO_KIND_SKIP ? 4 : 5;
  const int index[5] = {1, 2, 7, kind == TMXO_KIND_SKIP ? 7 : 4, 8};
  const hdr* from[5] = {&ha, &ha, &ha, kind == TMXO_KIND_SKIP ? &hb : &ha, &hb};
  memset(msgs, 0, 5 * 5 * 96); memset(lens, 0, sizeof(uint32_t) * 25); memset(digests, 0, 5 * 5 * 32);
  for (int q = 0; q < nq; q++) {
    uint8_t* m = msgs[q][0];
    const hdr* h = from[q];
    const int f = index[q];
    if (q == 0) { memcpy(m + 1, h->leaf[1], h->len[1] < 52 ? h->len[1] : 52); lens[q][0] = (uint32_t)(h->len[1] > 79 ? 79 : h->len[1]) + 1; }
    else if (q == 1) { m[1] = 0x08; tmxo_varint9(height_a, m + 2); lens[q][0] = (uint32_t)(h->len[2] > 79 ? 79 : h->len[2]) + 1; }
    else { const size_t w = (kind == TMXO_KIND_STEP && q == 3) ? 72 : 34; memcpy(m + 1, h->leaf[f], h->len[f] < w ? h->len[f] : w); lens[q][0] = (uint32_t)w + 1; }
    tmxo_sha256(m, lens[q][0], digests[q][0]);
    uint8_t aunts[4][32];
    rfc6962_aunts(&h->lh[0][0], 14, (size_t)f, aunts);
    for (int k = 0; k < 4; k++) {
      uint8_t* nm = msgs[q][1 + k];
      nm[0] = 0x01;
      if ((f >> k) & 1) { memcpy(nm + 1, aunts[k], 32); memcpy(nm + 33, digests[q][k], 32); }
      else { memcpy(nm + 1, digests[q][k], 32); memcpy(nm + 33, aunts[k], 32); }
      lens[q][1 + k] = 65;
      tmxo_sha256(nm, 65, digests[q][1 + k]);
    }
  }
  return nq;
}

/* is_valid_skip (reference circuits/input/tendermint_utils.rs:444-482), loop for loop, on 32-byte address records
 * (address[20], has_address, pad[3], voting_power).  Returns the predicate; *shared / *total as the reference accumulates them
 * (including its early exit, so *shared may be a partial sum when the result is true). */
int tmxo_is_valid_skip(const uint8_t* start, uint32_t n_start, const uint8_t* target, uint32_t n_target, const uint8_t* sigs, uint32_t n_sigs,
                       uint64_t* shared, uint64_t* total) {
  const double threshold = 1.0 / 3.0;
  uint64_t shared_voting_power = 0, total_power = 0;
  for (uint32_t j = 0; j < n_target; j++) total_power += rd64(target + 32 * j + 24);
  uint32_t idx = 0;
  while ((double)total_power * threshold > (double)shared_voting_power && idx < n_start) {
    for (uint32_t j = 0; j < n_target; j++) {
      if (memcmp(start + 32 * idx, target + 32 * j, 20) == 0) {            /* target_validator_set.validator(address) */
        for (uint32_t k = 0; k < n_sigs; k++)
          if (sigs[32 * k + 20] && memcmp(sigs + 32 * k, target + 32 * j, 20) == 0) shared_voting_power += rd64(target + 32 * j + 24);
        break;
      }
    }
    idx++;
  }
  if (shared) *shared = shared_voting_power;
  if (total) *total = total_power;
  return (double)total_power * threshold <= (double)shared_voting_power;
}

typedef struct {
  int kind; uint32_t lo, hi, n; const uint8_t *p, *t, *r, *cid; uint32_t cid_len; uint64_t skip_max; uint64_t* out; tmxo_report* reps; int rc;
} job;
static void* worker(void* arg) {
  job* j = (job*)arg;
  size_t ec = tmxo_elem_count(j->kind, j->n);
  for (uint32_t i = j->lo; i < j->hi; i++) {
    int rc = tmxo_witness(j->kind, j->p + (size_t)TMXO_REC_PROOF * i, j->t + (size_t)TMXO_REC_VALIDATOR * j->n * i,
                          j->r ? j->r + (size_t)TMXO_REC_HASHFIELD * j->n * i : NULL, j->n, j->cid, j->cid_len, j->skip_max,
                          j->out ? j->out + ec * i : NULL, j->reps ? j->reps + i : NULL);
    if (rc) j->rc = rc;
  }
  return NULL;
}
int tmxo_witness_batch(int kind, uint32_t n_proofs, const uint8_t* proof_recs, const uint8_t* target_recs,
                       const uint8_t* trusted_recs, uint32_t n, const uint8_t* chain_id, uint32_t chain_id_len,
                       uint64_t skip_max, uint64_t* out, tmxo_report* reps, uint32_t n_threads) {
  if (n_threads < 1) n_threads = 1;
  if (n_threads > n_proofs) n_threads = n_proofs ? n_proofs : 1;
  pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * n_threads); job* jobs = (job*)malloc(sizeof(job) * n_threads);
  int rc = 0;
  for (uint32_t k = 0; k < n_threads; k++) {
    jobs[k] = (job){kind, (uint32_t)((uint64_t)n_proofs * k / n_threads), (uint32_t)((uint64_t)n_proofs * (k + 1) / n_threads), n,
                    proof_recs, target_recs, trusted_recs, chain_id, chain_id_len, skip_max, out, reps, 0};
    if (n_threads == 1) worker(&jobs[k]); else pthread_create(&th[k], NULL, worker, &jobs[k]);
  }
  for (uint32_t k = 0; k < n_threads; k++) { if (n_threads > 1) pthread_join(th[k], NULL); if (jobs[k].rc) rc = jobs[k].rc; }
  free(th); free(jobs);
  return rc;
}

/* CPU baseline with a persistent pool (bench.py `cpu_baseline.all_cores`): n_threads workers are created once, meet at a barrier, and
 * worker k then computes proofs k, k + T, k + 2T, ... of a virtual batch of n_proofs * repeat proofs (record index modulo n_proofs),
 * compute only (no element output).  Returns the wall seconds from the barrier to the last worker's end; thread creation is outside. */
typedef struct {
  int kind; uint32_t k, T, n_proofs, total, n; const uint8_t *p, *t, *r, *cid; uint32_t cid_len; uint64_t skip_max; pthread_barrier_t* bar; int rc;
} pjob;
static void* pool_worker(void* arg) {
  pjob* j = (pjob*)arg;
  pthread_barrier_wait(j->bar);
  for (uint32_t v = j->k; v < j->total; v += j->T) {
    const uint32_t i = v % j->n_proofs;
    int rc = tmxo_witness(j->kind, j->p + (size_t)TMXO_REC_PROOF * i, j->t + (size_t)TMXO_REC_VALIDATOR * j->n * i,
                          j->r ? j->r + (size_t)TMXO_REC_HASHFIELD * j->n * i : NULL, j->n, j->cid, j->cid_len, j->skip_max, NULL, NULL);
    if (rc) j->rc = rc;
  }
  return NULL;
}
double tmxo_witness_pool_seconds(int kind, uint32_t n_proofs, const uint8_t* proof_recs, const uint8_t* target_recs, const uint8_t* trusted_recs,
                                 uint32_t n, const uint8_t* chain_id, uint32_t chain_id_len, uint64_t skip_max, uint32_t repeat, uint32_t n_threads) {
  if (n_threads < 1) n_threads = 1;
  pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * n_threads); pjob* jobs = (pjob*)malloc(sizeof(pjob) * n_threads);
  pthread_barrier_t bar;
  pthread_barrier_init(&bar, NULL, n_threads + 1);
  for (uint32_t k = 0; k < n_threads; k++) {
    jobs[k] = (pjob){kind, k, n_threads, n_proofs, n_proofs * repeat, n, proof_recs, target_recs, trusted_recs, chain_id, chain_id_len, skip_max, &bar, 0};
    pthread_create(&th[k], NULL, pool_worker, &jobs[k]);
  }
  struct timespec a, b;
  pthread_barrier_wait(&bar);
  clock_gettime(CLOCK_MONOTONIC, &a);
  int rc = 0;
  for (uint32_t k = 0; k < n_threads; k++) { pthread_join(th[k], NULL); if (jobs[k].rc) rc = jobs[k].rc; }
  clock_gettime(CLOCK_MONOTONIC, &b);
  pthread_barrier_destroy(&bar);
  free(th); free(jobs);
  return rc ? -1.0 : (double)(b.tv_sec - a.tv_sec) + 1e-9 * (double)(b.tv_nsec - a.tv_nsec);
}
