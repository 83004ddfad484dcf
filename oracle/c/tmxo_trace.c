/* ORACLE -- TEST INFRASTRUCTURE ONLY (CPU restatement; never linked into the product library).
 *
 * Level-2 trace rows of one skip / step proof (DESIGN.md "Level-2 trace rows"): this build's own row specification for what the
 * reference generates inside plonky2x / Curta (`curta_eddsa_verify_sigs_conditional` at reference circuits/builder/verify.rs:248-259,
 * `curta_sha256_variable` at validator.rs:228) -- absent sources, so the layout is NOT claimed equal to the reference's AIR columns;
 * it is validated by the constraint checker below, which ties every row to the input records and to the Level-1 values.
 *
 * Sections of a proof's trace block (elements, every one < 2^32):
 *   T.1 ladders     lane i, ladder k in {0: s*B, 1: h*A}, row r in 0..255, 65 elements             offset ((2 i + k) 256 + r) 65
 *   T.2 SHA-512     lane i, block b in {0, 1}, round t in 0..79, 18 elements                        N * 33280 + ((2 i + b) 80 + t) 18
 *   T.3 SHA-256     set s (0 target, 1 trusted: skip only), lane i, round t in 0..63, 9 elements    N * 36160 + ((s N + i) 64 + t) 9
 *   T.4 N x N bits  skip only: m[i][j] = signed[i] & (target pubkey i == trusted pubkey j)          after T.3, i * N + j
 *   T.5 tree nodes  set s, node slot q of the fixed-shape validator tree in Level-1 order (level by level, tmxo_tree_nodes(N) slots):
 *                   2 blocks x 64 rounds x 9 of SHA-256(01 | L | R) over the two children AS LEVEL-1 HOLDS THEM (the circuit hashes
 *                   every pair and then selects `both enabled ? hash : L`, validator.rs:248-251); a promoted slot (odd last node,
 *                   no right sibling) is zero                                                          after T.4, (s TN + q) 1152
 *   T.6 header      proof q (chain id, height, validators hash, X, Y: 4 for skip, 5 for step), hash h in {leaf, path node 0..3}:
 *                   2 blocks x 64 x 9 (an unused second block of a leaf hash is zero)                  after T.5, (5 q + h) 1152
 * Lanes that did not sign are traced on the dummy triple, like Level-1 (verify.rs:248-259 is conditional).  A lane whose A or R does
 * not decode has all-zero ladders (Level-1 reports zero points there as well). */
#include "tmxo.h"
#include <stdlib.h>
#include <string.h>

size_t tmxo_trace_elem_count(int kind, size_t n) {
  const size_t sets = kind == TMXO_KIND_SKIP ? 2 : 1;
  return n * (2 * 256 * 65 + 2 * 80 * 18 + sets * 64 * 9) + (kind == TMXO_KIND_SKIP ? n * n : 0) + sets * tmxo_tree_nodes(n) * 1152 +
         (kind == TMXO_KIND_SKIP ? 4 : 5) * 5 * 1152;
}

typedef struct { const uint8_t *pk, *sig, *msg; size_t mlen; } triple;
static triple effective(const uint8_t* v, const uint8_t* dpk, const uint8_t* dsig, const uint8_t* zero) {
  triple t;
  if (v[223] & TMXO_FLAG_SIGNED) { t.pk = v; t.sig = v + 32; t.msg = v + 96; t.mlen = (size_t)v[220] | ((size_t)v[221] << 8); if (t.mlen > 124) t.mlen = 124; }
  else { t.pk = dpk; t.sig = dsig; t.msg = zero; t.mlen = 32; }
  return t;
}
static size_t leaf_message(const uint8_t* pk, uint64_t power, uint32_t vlen, uint8_t out[48]) { /* 00 | marshalled[0..vlen] */
  uint8_t m[46];
  tmxo_marshal_validator(pk, power, m);
  if (vlen > 46) vlen = 46;
  out[0] = 0; memcpy(out + 1, m, vlen);
  return 1 + vlen;
}
static uint64_t rd64le(const uint8_t* p) { uint64_t v = 0; for (int k = 7; k >= 0; k--) v = (v << 8) | p[k]; return v; }

/* generate (check == 0: writes `trace`) or check (check != 0: reads it; returns 0 or section * 10^9 + lane * 10^6 + detail) */
static uint32_t rd32le(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
static int all_zero(const uint64_t* r, size_t k) { for (size_t i = 0; i < k; i++) if (r[i]) return 0; return 1; }

/* T.5 of one set: the children of every node slot come from the Level-1 tree (tmxo_fixed_shape_tree) */
static long long tree_rows(const uint8_t* leaves, uint32_t n, uint32_t nb, uint64_t* rows, int check, long long code) {
  const size_t tn = tmxo_tree_nodes(n);
  uint8_t* nodes = (uint8_t*)malloc(32 * (tn + 1)); uint8_t root[32];
  tmxo_fixed_shape_tree(leaves, n, nb, nodes, root);
  const uint8_t* cur = leaves; size_t sz = n, slot = 0; long long rc = 0;
  while (sz > 1 && !rc) {
    const size_t nx = (sz + 1) / 2;
    for (size_t i = 0; i < nx && !rc; i++, slot++) {
      uint64_t* r = rows + slot * 1152;
      if (2 * i + 1 < sz) {
        uint8_t m[65], dg[32];
        m[0] = 0x01; memcpy(m + 1, cur + 64 * i, 64);
        if (!check) tmxo_trace_sha256_2(m, 65, r);
        else { tmxo_inner_hash(cur + 64 * i, cur + 64 * i + 32, dg); const int e = tmxo_trace_sha256_2_check(r, m, 65, dg); if (e) rc = code + 100000LL * (long long)slot + e; }
      } else if (!check) memset(r, 0, sizeof(uint64_t) * 1152);
      else if (!all_zero(r, 1152)) rc = code + 100000LL * (long long)slot + 99999;
    }
    cur = nodes + 32 * (slot - nx); sz = nx;
  }
  free(nodes);
  return rc;
}

static long long run(int kind, const uint8_t* prec, const uint8_t* trec, const uint8_t* rrec, uint32_t n, uint64_t* trace, int check) {
  uint8_t dpk[32], dsig[64], zero[32] = {0}, bx[32], by[32];
  tmxo_dummy(dpk, dsig);
  tmxo_base_point(bx, by);
  const size_t o2 = (size_t)n * 33280, o3 = (size_t)n * 36160, sets = kind == TMXO_KIND_SKIP ? 2 : 1, o4 = o3 + sets * n * 576;
  const size_t tn = tmxo_tree_nodes(n), o5 = o4 + (kind == TMXO_KIND_SKIP ? (size_t)n * n : 0), o6 = o5 + sets * tn * 1152;
  uint64_t* tmp = (uint64_t*)malloc(sizeof(uint64_t) * 256 * 65);
  uint8_t* leaves = (uint8_t*)malloc(32 * (size_t)n * 2);
  long long rc = 0;
  for (uint32_t i = 0; i < n && !rc; i++) {
    const uint8_t* v = trec + (size_t)TMXO_REC_VALIDATOR * i;
    const triple t = effective(v, dpk, dsig, zero);
    tmxo_eddsa_trace tr;
    tmxo_eddsa_trace_lane(t.pk, t.sig, t.msg, t.mlen, &tr);
    for (int k = 0; k < 2 && !rc; k++) {
      uint64_t* rows = trace + ((size_t)(2 * i + k) * 256) * 65;
      const uint8_t* scalar = k ? tr.h : t.sig + 32;
      const uint8_t* px = k ? tr.pt[0] : bx; const uint8_t* py = k ? tr.pt[1] : by;
      if (!tr.decode_ok) {
        if (!check) memset(rows, 0, sizeof(uint64_t) * 256 * 65);
        else for (int e = 0; e < 256 * 65; e++) if (rows[e]) { rc = 1000000000LL + 1000000LL * i + 999; break; }
      } else if (!check) tmxo_trace_ladder(scalar, px, py, rows);
      else { int e = tmxo_trace_ladder_check(rows, scalar, px, py, tr.pt[k ? 6 : 4], tr.pt[k ? 7 : 5]); if (e) rc = 1000000000LL + 1000000LL * i + e + (k ? 500000 : 0); }
    }
    uint8_t hm[64 + 124];
    memcpy(hm, t.sig, 32); memcpy(hm + 32, t.pk, 32); memcpy(hm + 64, t.msg, t.mlen);
    uint64_t* srows = trace + o2 + (size_t)i * 2880;
    if (!rc) {
      if (!check) tmxo_trace_sha512(hm, 64 + t.mlen, srows);
      else { int e = tmxo_trace_sha512_check(srows, hm, 64 + t.mlen, tr.digest); if (e) rc = 2000000000LL + 1000000LL * i + e; }
    }
    for (size_t s = 0; s < sets && !rc; s++) {
      const uint8_t* rec = s ? rrec + (size_t)TMXO_REC_HASHFIELD * i : v;
      uint8_t lm[48], dg[32];
      const size_t ll = leaf_message(rec, rd64le(rec + (s ? 32 : 224)), rec[s ? 40 : 222], lm);
      tmxo_sha256(lm, ll, leaves + 32 * (s * n + i));
      uint64_t* lrows = trace + o3 + ((size_t)(s * n + i) * 64) * 9;
      if (!check) tmxo_trace_sha256_1(lm, ll, lrows);
      else { tmxo_sha256(lm, ll, dg); int e = tmxo_trace_sha256_1_check(lrows, lm, ll, dg); if (e) rc = 3000000000LL + 1000000LL * i + e + (s ? 500000 : 0); }
    }
    if (kind == TMXO_KIND_SKIP && !rc)
      for (uint32_t j = 0; j < n; j++) {
        const uint64_t m = (v[223] & TMXO_FLAG_SIGNED) && memcmp(v, rrec + (size_t)TMXO_REC_HASHFIELD * j, 32) == 0;
        if (!check) trace[o4 + (size_t)i * n + j] = m;
        else if (trace[o4 + (size_t)i * n + j] != m) { rc = 4000000000LL + 1000000LL * i + j; break; }
      }
  }
  for (size_t s = 0; s < sets && !rc; s++)
    rc = tree_rows(leaves + 32 * s * n, n, rd32le(prec + (s ? 60 : 56)), trace + o5 + s * tn * 1152, check, 5000000000LL + (s ? 500000000LL : 0));
  if (!rc) {
    uint32_t lens[5][5]; uint8_t dgs[5][5][32];
    uint8_t (*mm)[5][96] = (uint8_t (*)[5][96])malloc(5 * 5 * 96);
    const int nq = tmxo_header_proof_messages(kind, prec, mm, lens, dgs);
    for (int q = 0; q < nq && !rc; q++)
      for (int h = 0; h < 5 && !rc; h++) {
        uint64_t* r = trace + o6 + (size_t)(5 * q + h) * 1152;
        if (!check) tmxo_trace_sha256_2(mm[q][h], lens[q][h], r);
        else { const int e = tmxo_trace_sha256_2_check(r, mm[q][h], lens[q][h], dgs[q][h]); if (e) rc = 6000000000LL + 100000LL * (5 * q + h) + e; }
      }
    free(mm);
  }
  free(tmp); free(leaves);
  return rc;
}
int tmxo_trace(int kind, const uint8_t* proof_rec, const uint8_t* target_recs, const uint8_t* trusted_recs, uint32_t n, uint64_t* out) {
  return run(kind, proof_rec, target_recs, trusted_recs, n, out, 0) ? -1 : 0;
}
long long tmxo_trace_check(int kind, const uint8_t* proof_rec, const uint8_t* target_recs, const uint8_t* trusted_recs, uint32_t n, const uint64_t* trace) {
  return run(kind, proof_rec, target_recs, trusted_recs, n, (uint64_t*)trace, 1);
}
