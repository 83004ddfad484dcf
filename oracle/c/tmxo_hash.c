/* ORACLE -- TEST INFRASTRUCTURE ONLY (CPU restatement; never linked into the product library).
 *
 * SHA-256 (FIPS 180-4 §6.2) and SHA-512 (§6.4).  The reference reaches these through
 *   curta_sha256_variable        reference circuits/builder/validator.rs:228, verify.rs:202, shared.rs:194
 *   sha2::Sha256 (leaf/inner)    reference circuits/input/tendermint_utils.rs:351-372
 *   SHA-512 inside EdDSA         reference circuits/builder/verify.rs:248-259 (plonky2x/curta, absent)
 * Pinned by tests/test_oracle_kat.py against hashlib and the FIPS "abc" vectors.
 */
#include "tmxo.h"
#include <string.h>

static const uint32_t K256[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
    0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
    0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
    0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
    0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
    0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};

static uint32_t ror32(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }

static void sha256_block(uint32_t st[8], const uint8_t blk[64]) {
  uint32_t w[64];
  for (int i = 0; i < 16; i++)
    w[i] = ((uint32_t)blk[4 * i] << 24) | ((uint32_t)blk[4 * i + 1] << 16) | ((uint32_t)blk[4 * i + 2] << 8) | blk[4 * i + 3];
  for (int i = 16; i < 64; i++) {
    uint32_t s0 = ror32(w[i - 15], 7) ^ ror32(w[i - 15], 18) ^ (w[i - 15] >> 3);
    uint32_t s1 = ror32(w[i - 2], 17) ^ ror32(w[i - 2], 19) ^ (w[i - 2] >> 10);
    w[i] = w[i - 16] + s0 + w[i - 7] + s1;
  }
  uint32_t a = st[0], b = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
  for (int i = 0; i < 64; i++) {
    uint32_t S1 = ror32(e, 6) ^ ror32(e, 11) ^ ror32(e, 25);
    uint32_t ch = (e & f) ^ (~e & g);
    uint32_t t1 = h + S1 + ch + K256[i] + w[i];
    uint32_t S0 = ror32(a, 2) ^ ror32(a, 13) ^ ror32(a, 22);
    uint32_t mj = (a & b) ^ (a & c) ^ (b & c);
    uint32_t t2 = S0 + mj;
    h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
  }
  st[0] += a; st[1] += b; st[2] += c; st[3] += d; st[4] += e; st[5] += f; st[6] += g; st[7] += h;
}

void tmxo_sha256(const uint8_t* msg, size_t len, uint8_t out[32]) {
  uint32_t st[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
  size_t full = len / 64;
  for (size_t i = 0; i < full; i++) sha256_block(st, msg + 64 * i);
  uint8_t tail[128];
  size_t rem = len - 64 * full;
  memset(tail, 0, sizeof tail);
  memcpy(tail, msg + 64 * full, rem);
  tail[rem] = 0x80;
  size_t tl = (rem + 9 <= 64) ? 64 : 128;
  uint64_t bits = (uint64_t)len * 8;
  for (int i = 0; i < 8; i++) tail[tl - 1 - i] = (uint8_t)(bits >> (8 * i));
  sha256_block(st, tail);
  if (tl == 128) sha256_block(st, tail + 64);
  for (int i = 0; i < 8; i++) {
    out[4 * i] = (uint8_t)(st[i] >> 24); out[4 * i + 1] = (uint8_t)(st[i] >> 16);
    out[4 * i + 2] = (uint8_t)(st[i] >> 8); out[4 * i + 3] = (uint8_t)st[i];
  }
}

static const uint64_t K512[80] = {
    0x428a2f98d728ae22ULL, 0x7137449123ef65cdULL, 0xb5c0fbcfec4d3b2fULL, 0xe9b5dba58189dbbcULL, 0x3956c25bf348b538ULL,
    0x59f111f1b605d019ULL, 0x923f82a4af194f9bULL, 0xab1c5ed5da6d8118ULL, 0xd807aa98a3030242ULL, 0x12835b0145706fbeULL,
    0x243185be4ee4b28cULL, 0x550c7dc3d5ffb4e2ULL, 0x72be5d74f27b896fULL, 0x80deb1fe3b1696b1ULL, 0x9bdc06a725c71235ULL,
    0xc19bf174cf692694ULL, 0xe49b69c19ef14ad2ULL, 0xefbe4786384f25e3ULL, 0x0fc19dc68b8cd5b5ULL, 0x240ca1cc77ac9c65ULL,
    0x2de92c6f592b0275ULL, 0x4a7484aa6ea6e483ULL, 0x5cb0a9dcbd41fbd4ULL, 0x76f988da831153b5ULL, 0x983e5152ee66dfabULL,
    0xa831c66d2db43210ULL, 0xb00327c898fb213fULL, 0xbf597fc7beef0ee4ULL, 0xc6e00bf33da88fc2ULL, 0xd5a79147930aa725ULL,
    0x06ca6351e003826fULL, 0x142929670a0e6e70ULL, 0x27b70a8546d22ffcULL, 0x2e1b21385c26c926ULL, 0x4d2c6dfc5ac42aedULL,
    0x53380d139d95b3dfULL, 0x650a73548baf63deULL, 0x766a0abb3c77b2a8ULL, 0x81c2c92e47edaee6ULL, 0x92722c851482353bULL,
    0xa2bfe8a14cf10364ULL, 0xa81a664bbc423001ULL, 0xc24b8b70d0f89791ULL, 0xc76c51a30654be30ULL, 0xd192e819d6ef5218ULL,
    0xd69906245565a910ULL, 0xf40e35855771202aULL, 0x106aa07032bbd1b8ULL, 0x19a4c116b8d2d0c8ULL, 0x1e376c085141ab53ULL,
    0x2748774cdf8eeb99ULL, 0x34b0bcb5e19b48a8ULL, 0x391c0cb3c5c95a63ULL, 0x4ed8aa4ae3418acbULL, 0x5b9cca4f7763e373ULL,
    0x682e6ff3d6b2b8a3ULL, 0x748f82ee5defb2fcULL, 0x78a5636f43172f60ULL, 0x84c87814a1f0ab72ULL, 0x8cc702081a6439ecULL,
    0x90befffa23631e28ULL, 0xa4506cebde82bde9ULL, 0xbef9a3f7b2c67915ULL, 0xc67178f2e372532bULL, 0xca273eceea26619cULL,
    0xd186b8c721c0c207ULL, 0xeada7dd6cde0eb1eULL, 0xf57d4f7fee6ed178ULL, 0x06f067aa72176fbaULL, 0x0a637dc5a2c898a6ULL,
    0x113f9804bef90daeULL, 0x1b710b35131c471bULL, 0x28db77f523047d84ULL, 0x32caab7b40c72493ULL, 0x3c9ebe0a15c9bebcULL,
    0x431d67c49c100d4cULL, 0x4cc5d4becb3e42b6ULL, 0x597f299cfc657e2aULL, 0x5fcb6fab3ad6faecULL, 0x6c44198c4a475817ULL};

static uint64_t ror64(uint64_t x, int n) { return (x >> n) | (x << (64 - n)); }

static void sha512_block(uint64_t st[8], const uint8_t blk[128]) {
  uint64_t w[80];
  for (int i = 0; i < 16; i++) {
    uint64_t v = 0;
    for (int k = 0; k < 8; k++) v = (v << 8) | blk[8 * i + k];
    w[i] = v;
  }
  for (int i = 16; i < 80; i++) {
    uint64_t s0 = ror64(w[i - 15], 1) ^ ror64(w[i - 15], 8) ^ (w[i - 15] >> 7);
    uint64_t s1 = ror64(w[i - 2], 19) ^ ror64(w[i - 2], 61) ^ (w[i - 2] >> 6);
    w[i] = w[i - 16] + s0 + w[i - 7] + s1;
  }
  uint64_t a = st[0], b = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
  for (int i = 0; i < 80; i++) {
    uint64_t S1 = ror64(e, 14) ^ ror64(e, 18) ^ ror64(e, 41);
    uint64_t ch = (e & f) ^ (~e & g);
    uint64_t t1 = h + S1 + ch + K512[i] + w[i];
    uint64_t S0 = ror64(a, 28) ^ ror64(a, 34) ^ ror64(a, 39);
    uint64_t mj = (a & b) ^ (a & c) ^ (b & c);
    uint64_t t2 = S0 + mj;
    h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
  }
  st[0] += a; st[1] += b; st[2] += c; st[3] += d; st[4] += e; st[5] += f; st[6] += g; st[7] += h;
}

void tmxo_sha512_3(const uint8_t* p0, size_t l0, const uint8_t* p1, size_t l1, const uint8_t* p2, size_t l2, uint8_t out[64]) {
  /* hash of the concatenation p0|p1|p2, total <= 4 blocks is all this path needs; generic anyway */
  uint64_t st[8] = {0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL, 0xa54ff53a5f1d36f1ULL,
                    0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL, 0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL};
  uint8_t buf[128];
  size_t fill = 0, total = l0 + l1 + l2;
  const uint8_t* ps[3] = {p0, p1, p2};
  size_t ls[3] = {l0, l1, l2};
  for (int k = 0; k < 3; k++)
    for (size_t i = 0; i < ls[k]; i++) {
      buf[fill++] = ps[k][i];
      if (fill == 128) { sha512_block(st, buf); fill = 0; }
    }
  buf[fill++] = 0x80;
  if (fill > 112) { memset(buf + fill, 0, 128 - fill); sha512_block(st, buf); fill = 0; }
  memset(buf + fill, 0, 128 - fill);
  uint64_t bits = (uint64_t)total * 8;
  for (int i = 0; i < 8; i++) buf[127 - i] = (uint8_t)(bits >> (8 * i));
  sha512_block(st, buf);
  for (int i = 0; i < 8; i++)
    for (int k = 0; k < 8; k++) out[8 * i + k] = (uint8_t)(st[i] >> (56 - 8 * k));
}

void tmxo_sha512(const uint8_t* msg, size_t len, uint8_t out[64]) { tmxo_sha512_3(msg, len, 0, 0, 0, 0, out); }

/* ================================================================================================ Level-2 SHA round rows
 * TEST INFRASTRUCTURE.  DESIGN.md "Level-2 trace rows".  SHA-512 (the EdDSA hash, at most two blocks): per block 80 rows of 18 elements
 * [W_t lo, W_t hi, a lo, a hi, ..., h lo, h hi] = the schedule word and the eight working variables AFTER round t, 32-bit limbs, low limb
 * first; the rows of an unused second block are zero.  SHA-256 (validator leaf hash, one block): 64 rows of 9 elements [W_t, a .. h]. */
static const uint64_t IV512[8] = {0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL, 0xa54ff53a5f1d36f1ULL,
                                  0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL, 0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL};
static const uint32_t IV256[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};

static size_t pad512(const uint8_t* msg, size_t len, uint8_t buf[256]) { /* returns the number of blocks (1 or 2), len <= 239 */
  const size_t nb = (len + 17 > 128) ? 2 : 1;
  memset(buf, 0, 256); memcpy(buf, msg, len); buf[len] = 0x80;
  const uint64_t bits = (uint64_t)len * 8;
  for (int i = 0; i < 8; i++) buf[128 * nb - 1 - i] = (uint8_t)(bits >> (8 * i));
  return nb;
}
static void put64(uint64_t* o, uint64_t v) { o[0] = v & 0xffffffffu; o[1] = v >> 32; }
static int get64(const uint64_t* p, uint64_t* v) { if ((p[0] >> 32) || (p[1] >> 32)) return 0; *v = p[0] | (p[1] << 32); return 1; }

void tmxo_trace_sha512(const uint8_t* msg, size_t len, uint64_t* rows) {
  uint8_t buf[256];
  const size_t nb = pad512(msg, len, buf);
  memset(rows, 0, sizeof(uint64_t) * 2 * 80 * 18);
  uint64_t st[8];
  memcpy(st, IV512, sizeof st);
  for (size_t b = 0; b < nb; b++) {
    uint64_t w[80], v[8];
    for (int i = 0; i < 16; i++) { w[i] = 0; for (int k = 0; k < 8; k++) w[i] = (w[i] << 8) | buf[128 * b + 8 * i + k]; }
    for (int i = 16; i < 80; i++)
      w[i] = w[i - 16] + (ror64(w[i - 15], 1) ^ ror64(w[i - 15], 8) ^ (w[i - 15] >> 7)) + w[i - 7] + (ror64(w[i - 2], 19) ^ ror64(w[i - 2], 61) ^ (w[i - 2] >> 6));
    memcpy(v, st, sizeof v);
    for (int t = 0; t < 80; t++) {
      const uint64_t t1 = v[7] + (ror64(v[4], 14) ^ ror64(v[4], 18) ^ ror64(v[4], 41)) + ((v[4] & v[5]) ^ (~v[4] & v[6])) + K512[t] + w[t];
      const uint64_t t2 = (ror64(v[0], 28) ^ ror64(v[0], 34) ^ ror64(v[0], 39)) + ((v[0] & v[1]) ^ (v[0] & v[2]) ^ (v[1] & v[2]));
      v[7] = v[6]; v[6] = v[5]; v[5] = v[4]; v[4] = v[3] + t1; v[3] = v[2]; v[2] = v[1]; v[1] = v[0]; v[0] = t1 + t2;
      uint64_t* row = rows + (b * 80 + t) * 18;
      put64(row, w[t]);
      for (int k = 0; k < 8; k++) put64(row + 2 + 2 * k, v[k]);
    }
    for (int k = 0; k < 8; k++) st[k] += v[k];
  }
}
/* 0 = holds; else 100 * (block * 80 + round + 1) + constraint number */
int tmxo_trace_sha512_check(const uint64_t* rows, const uint8_t* msg, size_t len, const uint8_t digest[64]) {
  uint8_t buf[256];
  const size_t nb = pad512(msg, len, buf);
  uint64_t st[8], prev[8], w[80];
  memcpy(st, IV512, sizeof st);
  for (size_t b = 0; b < 2; b++) {
    if (b >= nb) { for (int i = 0; i < 80 * 18; i++) if (rows[b * 80 * 18 + i]) return 100 * (int)(b * 80 + 1) + 9; continue; }   /* unused block: zeros */
    memcpy(prev, st, sizeof prev);
    for (int t = 0; t < 80; t++) {
      const uint64_t* row = rows + (b * 80 + t) * 18;
      const int e = 100 * (int)(b * 80 + t + 1);
      uint64_t v[8];
      if (!get64(row, &w[t])) return e + 1;                                                        /* u32 limbs */
      for (int k = 0; k < 8; k++) if (!get64(row + 2 + 2 * k, &v[k])) return e + 1;
      uint64_t want;
      if (t < 16) { want = 0; for (int k = 0; k < 8; k++) want = (want << 8) | buf[128 * b + 8 * t + k]; }   /* the padded message */
      else want = w[t - 16] + (ror64(w[t - 15], 1) ^ ror64(w[t - 15], 8) ^ (w[t - 15] >> 7)) + w[t - 7] + (ror64(w[t - 2], 19) ^ ror64(w[t - 2], 61) ^ (w[t - 2] >> 6));
      if (w[t] != want) return e + 2;
      const uint64_t t1 = prev[7] + (ror64(prev[4], 14) ^ ror64(prev[4], 18) ^ ror64(prev[4], 41)) + ((prev[4] & prev[5]) ^ (~prev[4] & prev[6])) + K512[t] + w[t];
      const uint64_t t2 = (ror64(prev[0], 28) ^ ror64(prev[0], 34) ^ ror64(prev[0], 39)) + ((prev[0] & prev[1]) ^ (prev[0] & prev[2]) ^ (prev[1] & prev[2]));
      if (v[0] != t1 + t2 || v[1] != prev[0] || v[2] != prev[1] || v[3] != prev[2] || v[4] != prev[3] + t1 || v[5] != prev[4] || v[6] != prev[5] ||
          v[7] != prev[6]) return e + 3;                                                             /* the round function */
      memcpy(prev, v, sizeof prev);
    }
    for (int k = 0; k < 8; k++) st[k] += prev[k];
  }
  for (int i = 0; i < 8; i++) for (int k = 0; k < 8; k++) if (digest[8 * i + k] != (uint8_t)(st[i] >> (56 - 8 * k))) return 4;   /* = the Level-1 digest */
  return 0;
}

void tmxo_trace_sha256_1(const uint8_t* msg, size_t len, uint64_t* rows) { /* one block: len <= 55 */
  uint8_t buf[64];
  memset(buf, 0, 64); memcpy(buf, msg, len); buf[len] = 0x80;
  const uint64_t bits = (uint64_t)len * 8;
  for (int i = 0; i < 8; i++) buf[63 - i] = (uint8_t)(bits >> (8 * i));
  uint32_t w[64], v[8];
  for (int i = 0; i < 16; i++) w[i] = ((uint32_t)buf[4 * i] << 24) | ((uint32_t)buf[4 * i + 1] << 16) | ((uint32_t)buf[4 * i + 2] << 8) | buf[4 * i + 3];
  for (int i = 16; i < 64; i++)
    w[i] = w[i - 16] + (ror32(w[i - 15], 7) ^ ror32(w[i - 15], 18) ^ (w[i - 15] >> 3)) + w[i - 7] + (ror32(w[i - 2], 17) ^ ror32(w[i - 2], 19) ^ (w[i - 2] >> 10));
  memcpy(v, IV256, sizeof v);
  for (int t = 0; t < 64; t++) {
    const uint32_t t1 = v[7] + (ror32(v[4], 6) ^ ror32(v[4], 11) ^ ror32(v[4], 25)) + ((v[4] & v[5]) ^ (~v[4] & v[6])) + K256[t] + w[t];
    const uint32_t t2 = (ror32(v[0], 2) ^ ror32(v[0], 13) ^ ror32(v[0], 22)) + ((v[0] & v[1]) ^ (v[0] & v[2]) ^ (v[1] & v[2]));
    v[7] = v[6]; v[6] = v[5]; v[5] = v[4]; v[4] = v[3] + t1; v[3] = v[2]; v[2] = v[1]; v[1] = v[0]; v[0] = t1 + t2;
    rows[9 * t] = w[t];
    for (int k = 0; k < 8; k++) rows[9 * t + 1 + k] = v[k];
  }
}
int tmxo_trace_sha256_1_check(const uint64_t* rows, const uint8_t* msg, size_t len, const uint8_t digest[32]) {
  uint8_t buf[64];
  memset(buf, 0, 64); memcpy(buf, msg, len); buf[len] = 0x80;
  const uint64_t bits = (uint64_t)len * 8;
  for (int i = 0; i < 8; i++) buf[63 - i] = (uint8_t)(bits >> (8 * i));
  uint32_t w[64], prev[8];
  memcpy(prev, IV256, sizeof prev);
  for (int t = 0; t < 64; t++) {
    const uint64_t* row = rows + 9 * t;
    const int e = 100 * (t + 1);
    for (int k = 0; k < 9; k++) if (row[k] >> 32) return e + 1;
    w[t] = (uint32_t)row[0];
    uint32_t want;
    if (t < 16) want = ((uint32_t)buf[4 * t] << 24) | ((uint32_t)buf[4 * t + 1] << 16) | ((uint32_t)buf[4 * t + 2] << 8) | buf[4 * t + 3];
    else want = w[t - 16] + (ror32(w[t - 15], 7) ^ ror32(w[t - 15], 18) ^ (w[t - 15] >> 3)) + w[t - 7] + (ror32(w[t - 2], 17) ^ ror32(w[t - 2], 19) ^ (w[t - 2] >> 10));
    if (w[t] != want) return e + 2;
    const uint32_t t1 = prev[7] + (ror32(prev[4], 6) ^ ror32(prev[4], 11) ^ ror32(prev[4], 25)) + ((prev[4] & prev[5]) ^ (~prev[4] & prev[6])) + K256[t] + w[t];
    const uint32_t t2 = (ror32(prev[0], 2) ^ ror32(prev[0], 13) ^ ror32(prev[0], 22)) + ((prev[0] & prev[1]) ^ (prev[0] & prev[2]) ^ (prev[1] & prev[2]));
    const uint32_t v[8] = {t1 + t2, prev[0], prev[1], prev[2], prev[3] + t1, prev[4], prev[5], prev[6]};
    for (int k = 0; k < 8; k++) if ((uint32_t)row[1 + k] != v[k]) return e + 3;
    memcpy(prev, v, sizeof prev);
  }
  for (int i = 0; i < 8; i++) {
    const uint32_t s = IV256[i] + prev[i];
    if (digest[4 * i] != (uint8_t)(s >> 24) || digest[4 * i + 1] != (uint8_t)(s >> 16) || digest[4 * i + 2] != (uint8_t)(s >> 8) || digest[4 * i + 3] != (uint8_t)s) return 4;
  }
  return 0;
}

/* SHA-256 of a message of at most 119 bytes (two blocks): 2 x 64 rows of 9 elements [W_t, a .. h after round t]; the second block starts
 * from the chaining value IV + (a .. h of row 63); the rows of an unused second block are zero.  Inner tree nodes (01 | L | R: 65 bytes,
 * always two blocks) and the header-proof leaves and path nodes. */
static size_t pad256(const uint8_t* msg, size_t len, uint8_t buf[128]) {
  const size_t nb = (len + 9 > 64) ? 2 : 1;
  memset(buf, 0, 128); memcpy(buf, msg, len); buf[len] = 0x80;
  const uint64_t bits = (uint64_t)len * 8;
  for (int i = 0; i < 8; i++) buf[64 * nb - 1 - i] = (uint8_t)(bits >> (8 * i));
  return nb;
}
void tmxo_trace_sha256_2(const uint8_t* msg, size_t len, uint64_t* rows) {
  uint8_t buf[128];
  const size_t nb = pad256(msg, len, buf);
  memset(rows, 0, sizeof(uint64_t) * 2 * 64 * 9);
  uint32_t st[8];
  memcpy(st, IV256, sizeof st);
  for (size_t b = 0; b < nb; b++) {
    uint32_t w[64], v[8];
    for (int i = 0; i < 16; i++) { const uint8_t* q = buf + 64 * b + 4 * i; w[i] = ((uint32_t)q[0] << 24) | ((uint32_t)q[1] << 16) | ((uint32_t)q[2] << 8) | q[3]; }
    for (int i = 16; i < 64; i++)
      w[i] = w[i - 16] + (ror32(w[i - 15], 7) ^ ror32(w[i - 15], 18) ^ (w[i - 15] >> 3)) + w[i - 7] + (ror32(w[i - 2], 17) ^ ror32(w[i - 2], 19) ^ (w[i - 2] >> 10));
    memcpy(v, st, sizeof v);
    for (int t = 0; t < 64; t++) {
      const uint32_t t1 = v[7] + (ror32(v[4], 6) ^ ror32(v[4], 11) ^ ror32(v[4], 25)) + ((v[4] & v[5]) ^ (~v[4] & v[6])) + K256[t] + w[t];
      const uint32_t t2 = (ror32(v[0], 2) ^ ror32(v[0], 13) ^ ror32(v[0], 22)) + ((v[0] & v[1]) ^ (v[0] & v[2]) ^ (v[1] & v[2]));
      v[7] = v[6]; v[6] = v[5]; v[5] = v[4]; v[4] = v[3] + t1; v[3] = v[2]; v[2] = v[1]; v[1] = v[0]; v[0] = t1 + t2;
      uint64_t* row = rows + (b * 64 + t) * 9;
      row[0] = w[t];
      for (int k = 0; k < 8; k++) row[1 + k] = v[k];
    }
    for (int k = 0; k < 8; k++) st[k] += v[k];
  }
}
/* 0 = holds; else 100 * (block * 64 + round + 1) + constraint number; 4 = the digest */
int tmxo_trace_sha256_2_check(const uint64_t* rows, const uint8_t* msg, size_t len, const uint8_t digest[32]) {
  uint8_t buf[128];
  const size_t nb = pad256(msg, len, buf);
  uint32_t st[8], prev[8], w[64];
  memcpy(st, IV256, sizeof st);
  for (size_t b = 0; b < 2; b++) {
    if (b >= nb) { for (int i = 0; i < 64 * 9; i++) if (rows[b * 64 * 9 + i]) return 100 * (int)(b * 64 + 1) + 9; continue; }
    memcpy(prev, st, sizeof prev);
    for (int t = 0; t < 64; t++) {
      const uint64_t* row = rows + (b * 64 + t) * 9;
      const int e = 100 * (int)(b * 64 + t + 1);
      for (int k = 0; k < 9; k++) if (row[k] >> 32) return e + 1;
      w[t] = (uint32_t)row[0];
      uint32_t want;
      if (t < 16) { const uint8_t* q = buf + 64 * b + 4 * t; want = ((uint32_t)q[0] << 24) | ((uint32_t)q[1] << 16) | ((uint32_t)q[2] << 8) | q[3]; }
      else want = w[t - 16] + (ror32(w[t - 15], 7) ^ ror32(w[t - 15], 18) ^ (w[t - 15] >> 3)) + w[t - 7] + (ror32(w[t - 2], 17) ^ ror32(w[t - 2], 19) ^ (w[t - 2] >> 10));
      if (w[t] != want) return e + 2;
      const uint32_t t1 = prev[7] + (ror32(prev[4], 6) ^ ror32(prev[4], 11) ^ ror32(prev[4], 25)) + ((prev[4] & prev[5]) ^ (~prev[4] & prev[6])) + K256[t] + w[t];
      const uint32_t t2 = (ror32(prev[0], 2) ^ ror32(prev[0], 13) ^ ror32(prev[0], 22)) + ((prev[0] & prev[1]) ^ (prev[0] & prev[2]) ^ (prev[1] & prev[2]));
      const uint32_t v[8] = {t1 + t2, prev[0], prev[1], prev[2], prev[3] + t1, prev[4], prev[5], prev[6]};
      for (int k = 0; k < 8; k++) if ((uint32_t)row[1 + k] != v[k]) return e + 3;
      memcpy(prev, v, sizeof prev);
    }
    for (int k = 0; k < 8; k++) st[k] += prev[k];
  }
  for (int i = 0; i < 8; i++)
    if (digest[4 * i] != (uint8_t)(st[i] >> 24) || digest[4 * i + 1] != (uint8_t)(st[i] >> 16) || digest[4 * i + 2] != (uint8_t)(st[i] >> 8) || digest[4 * i + 3] != (uint8_t)st[i]) return 4;
  return 0;
}
