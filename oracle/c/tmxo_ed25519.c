/* ORACLE -- TEST INFRASTRUCTURE ONLY (CPU restatement; never linked into the product library).
 *
 * Ed25519 value semantics of `curta_eddsa_verify_sigs_conditional` (call site: reference
 * circuits/builder/verify.rs:248-259; implementation in plonky2x@succinctx v1.0.3 / starkyx -- absent,
 * Cargo.lock:3017-3019, 4037-4039) and of the host check `Verifier::verify` (reference
 * circuits/input/conversion.rs:48-49; ed25519-consensus 2.1.0 -- absent).  Restated from RFC 8032 §5.1:
 *     h = SHA512(R || A || M) mod l,  A = decode(pk), R = decode(sig[0..32]),  accept iff s*B == R + h*A.
 * Representation deliberately differs from the HIP path (5 x 51-bit limbs + unsigned __int128 here,
 * 10 x 25.5-bit limbs there) so that the two implementations are independent.
 * Pinned by: RFC 8032 §7.1 vectors, every signature in the reference fixtures, oracle/py big-int model.
 */
#include "tmxo.h"
#include <pthread.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef uint64_t fe[5];
#define M51 ((1ULL << 51) - 1)

static void fe_copy(fe o, const fe a) { for (int i = 0; i < 5; i++) o[i] = a[i]; }
static void fe_set(fe o, uint64_t v) { o[0] = v; o[1] = o[2] = o[3] = o[4] = 0; }

static void fe_carry(fe h) {
  uint64_t c;
  for (int rep = 0; rep < 2; rep++) {
    c = h[0] >> 51; h[0] &= M51; h[1] += c;
    c = h[1] >> 51; h[1] &= M51; h[2] += c;
    c = h[2] >> 51; h[2] &= M51; h[3] += c;
    c = h[3] >> 51; h[3] &= M51; h[4] += c;
    c = h[4] >> 51; h[4] &= M51; h[0] += 19 * c;
  }
}
static void fe_add(fe o, const fe a, const fe b) { for (int i = 0; i < 5; i++) o[i] = a[i] + b[i]; fe_carry(o); }
static void fe_sub(fe o, const fe a, const fe b) {
  /* a + 4p - b keeps every limb non-negative for carried inputs */
  o[0] = a[0] + 4 * (M51 - 18) - b[0];
  for (int i = 1; i < 5; i++) o[i] = a[i] + 4 * M51 - b[i];
  fe_carry(o);
}
static void fe_neg(fe o, const fe a) { fe z; fe_set(z, 0); fe_sub(o, z, a); }

static void fe_reduce_wide(fe o, u128 t0, u128 t1, u128 t2, u128 t3, u128 t4) {
  uint64_t r0, r1, r2, r3, r4, c;
  t1 += (uint64_t)(t0 >> 51); r0 = (uint64_t)t0 & M51;
  t2 += (uint64_t)(t1 >> 51); r1 = (uint64_t)t1 & M51;
  t3 += (uint64_t)(t2 >> 51); r2 = (uint64_t)t2 & M51;
  t4 += (uint64_t)(t3 >> 51); r3 = (uint64_t)t3 & M51;
  c = (uint64_t)(t4 >> 51); r4 = (uint64_t)t4 & M51;
  r0 += 19 * c; c = r0 >> 51; r0 &= M51; r1 += c;
  c = r1 >> 51; r1 &= M51; r2 += c;
  o[0] = r0; o[1] = r1; o[2] = r2; o[3] = r3; o[4] = r4;
}
static void fe_mul(fe o, const fe a, const fe b) {
  uint64_t a0 = a[0], a1 = a[1], a2 = a[2], a3 = a[3], a4 = a[4];
  uint64_t b0 = b[0], b1 = b[1], b2 = b[2], b3 = b[3], b4 = b[4];
  uint64_t b1_19 = 19 * b1, b2_19 = 19 * b2, b3_19 = 19 * b3, b4_19 = 19 * b4;
  u128 t0 = (u128)a0 * b0 + (u128)a1 * b4_19 + (u128)a2 * b3_19 + (u128)a3 * b2_19 + (u128)a4 * b1_19;
  u128 t1 = (u128)a0 * b1 + (u128)a1 * b0 + (u128)a2 * b4_19 + (u128)a3 * b3_19 + (u128)a4 * b2_19;
  u128 t2 = (u128)a0 * b2 + (u128)a1 * b1 + (u128)a2 * b0 + (u128)a3 * b4_19 + (u128)a4 * b3_19;
  u128 t3 = (u128)a0 * b3 + (u128)a1 * b2 + (u128)a2 * b1 + (u128)a3 * b0 + (u128)a4 * b4_19;
  u128 t4 = (u128)a0 * b4 + (u128)a1 * b3 + (u128)a2 * b2 + (u128)a3 * b1 + (u128)a4 * b0;
  fe_reduce_wide(o, t0, t1, t2, t3, t4);
}
static void fe_sq(fe o, const fe a) {
  uint64_t a0 = a[0], a1 = a[1], a2 = a[2], a3 = a[3], a4 = a[4];
  uint64_t d0 = 2 * a0, d1 = 2 * a1, d2 = 2 * a2, a3_19 = 19 * a3, a4_19 = 19 * a4;
  u128 t0 = (u128)a0 * a0 + (u128)d1 * a4_19 + (u128)d2 * a3_19;
  u128 t1 = (u128)d0 * a1 + (u128)d2 * a4_19 + (u128)a3 * a3_19;
  u128 t2 = (u128)d0 * a2 + (u128)a1 * a1 + (u128)(2 * a3) * a4_19;
  u128 t3 = (u128)d0 * a3 + (u128)d1 * a2 + (u128)a4 * a4_19;
  u128 t4 = (u128)d0 * a4 + (u128)d1 * a3 + (u128)a2 * a2;
  fe_reduce_wide(o, t0, t1, t2, t3, t4);
}
static void fe_sqn(fe o, const fe a, int n) { fe_sq(o, a); for (int i = 1; i < n; i++) fe_sq(o, o); }

static void fe_frombytes(fe o, const uint8_t s[32]) {
  uint64_t w[4];
  for (int i = 0; i < 4; i++) { w[i] = 0; for (int k = 7; k >= 0; k--) w[i] = (w[i] << 8) | s[8 * i + k]; }
  w[3] &= 0x7fffffffffffffffULL; /* bit 255 is the sign bit of the encoding */
  o[0] = w[0] & M51;
  o[1] = ((w[0] >> 51) | (w[1] << 13)) & M51;
  o[2] = ((w[1] >> 38) | (w[2] << 26)) & M51;
  o[3] = ((w[2] >> 25) | (w[3] << 39)) & M51;
  o[4] = (w[3] >> 12) & M51;
}
static void fe_tobytes(uint8_t s[32], const fe a) {
  fe t; fe_copy(t, a); fe_carry(t);
  /* canonical: add 19, see if it overflows 2^255, then conditionally subtract p */
  uint64_t q = (t[0] + 19) >> 51;
  q = (t[1] + q) >> 51; q = (t[2] + q) >> 51; q = (t[3] + q) >> 51; q = (t[4] + q) >> 51;
  t[0] += 19 * q;
  uint64_t c;
  c = t[0] >> 51; t[0] &= M51; t[1] += c;
  c = t[1] >> 51; t[1] &= M51; t[2] += c;
  c = t[2] >> 51; t[2] &= M51; t[3] += c;
  c = t[3] >> 51; t[3] &= M51; t[4] += c;
  t[4] &= M51;
  uint64_t w[4];
  w[0] = t[0] | (t[1] << 51);
  w[1] = (t[1] >> 13) | (t[2] << 38);
  w[2] = (t[2] >> 26) | (t[3] << 25);
  w[3] = (t[3] >> 39) | (t[4] << 12);
  for (int i = 0; i < 4; i++) for (int k = 0; k < 8; k++) s[8 * i + k] = (uint8_t)(w[i] >> (8 * k));
}
static int fe_iszero(const fe a) { uint8_t s[32]; fe_tobytes(s, a); uint8_t r = 0; for (int i = 0; i < 32; i++) r |= s[i]; return r == 0; }

static int fe_isodd(const fe a) { uint8_t s[32]; fe_tobytes(s, a); return s[0] & 1; }

/* z^(2^250 - 1) by the classic addition chain; also returns z^11 */
static void fe_pow2_250_1(fe out, fe z11, const fe z) {
  fe z2, z9, t, z5, z10, z20, z50, z100;
  fe_sq(z2, z);
  fe_sqn(t, z2, 2);           /* z^8 */
  fe_mul(z9, t, z);
  fe_mul(z11, z9, z2);
  fe_sq(t, z11);              /* z^22 */
  fe_mul(z5, t, z9);          /* 2^5 - 1 */
  fe_sqn(t, z5, 5); fe_mul(z10, t, z5);
  fe_sqn(t, z10, 10); fe_mul(z20, t, z10);
  fe_sqn(t, z20, 20); fe_mul(t, t, z20);      /* 2^40 - 1 */
  fe_sqn(t, t, 10); fe_mul(z50, t, z10);
  fe_sqn(t, z50, 50); fe_mul(z100, t, z50);
  fe_sqn(t, z100, 100); fe_mul(t, t, z100);   /* 2^200 - 1 */
  fe_sqn(t, t, 50); fe_mul(out, t, z50);      /* 2^250 - 1 */
}
static void fe_invert(fe o, const fe z) { fe t, z11; fe_pow2_250_1(t, z11, z); fe_sqn(t, t, 5); fe_mul(o, t, z11); }
static void fe_pow22523(fe o, const fe z) { fe t, z11; fe_pow2_250_1(t, z11, z); fe_sqn(t, t, 2); fe_mul(o, t, z); }

/* curve constants, little-endian canonical bytes */
static const uint8_t D_BYTES[32] = {0xa3, 0x78, 0x59, 0x13, 0xca, 0x4d, 0xeb, 0x75, 0xab, 0xd8, 0x41, 0x41, 0x4d, 0x0a, 0x70, 0x00,
                                    0x98, 0xe8, 0x79, 0x77, 0x79, 0x40, 0xc7, 0x8c, 0x73, 0xfe, 0x6f, 0x2b, 0xee, 0x6c, 0x03, 0x52};
static const uint8_t SQRTM1_BYTES[32] = {0xb0, 0xa0, 0x0e, 0x4a, 0x27, 0x1b, 0xee, 0xc4, 0x78, 0xe4, 0x2f, 0xad, 0x06, 0x18, 0x43, 0x2f,
                                         0xa7, 0xd7, 0xfb, 0x3d, 0x99, 0x00, 0x4d, 0x2b, 0x0b, 0xdf, 0xc1, 0x4f, 0x80, 0x24, 0x83, 0x2b};
static const uint8_t BY_BYTES[32] = {0x58, 0x66, 0x66, 0x66, 0x66, 0x66, 0x66, 0x66, 0x66, 0x66, 0x66, 0x66, 0x66, 0x66, 0x66, 0x66,
                                     0x66, 0x66, 0x66, 0x66, 0x66, 0x66, 0x66, 0x66, 0x66, 0x66, 0x66, 0x66, 0x66, 0x66, 0x66, 0x66};

typedef struct { fe X, Y, Z, T; } ge;

static fe FE_D, FE_2D, FE_SQRTM1;
static ge GE_B;
static ge BTAB[64][16]; /* BTAB[i][j] = j * 16^i * B */
static pthread_once_t once = PTHREAD_ONCE_INIT;

static void ge_ident(ge* r) { fe_set(r->X, 0); fe_set(r->Y, 1); fe_set(r->Z, 1); fe_set(r->T, 0); }

/* unified addition, twisted Edwards a = -1, extended coordinates (Hisil-Wong-Carter-Dawson 2008 §3.1) */
static void ge_add(ge* r, const ge* p, const ge* q) {
  fe a, b, c, d, e, f, g, h, t;
  fe_sub(a, p->Y, p->X); fe_sub(t, q->Y, q->X); fe_mul(a, a, t);
  fe_add(b, p->Y, p->X); fe_add(t, q->Y, q->X); fe_mul(b, b, t);
  fe_mul(c, p->T, q->T); fe_mul(c, c, FE_2D);
  fe_mul(d, p->Z, q->Z); fe_add(d, d, d);
  fe_sub(e, b, a); fe_sub(f, d, c); fe_add(g, d, c); fe_add(h, b, a);
  fe_mul(r->X, e, f); fe_mul(r->Y, g, h); fe_mul(r->Z, f, g); fe_mul(r->T, e, h);
}
static void ge_dbl(ge* r, const ge* p) {
  fe a, b, c, e, f, g, h, t;
  fe_sq(a, p->X); fe_sq(b, p->Y); fe_sq(c, p->Z); fe_add(c, c, c);
  fe_add(h, a, b);                       /* H = A + B   (with a=-1: D = -A) */
  fe_add(t, p->X, p->Y); fe_sq(t, t); fe_sub(e, h, t);   /* E = H - (X+Y)^2 */
  fe_sub(g, a, b);                       /* G = A - B */
  fe_add(f, c, g);                       /* F = C + G */
  fe_mul(r->X, e, f); fe_mul(r->Y, g, h); fe_mul(r->Z, f, g); fe_mul(r->T, e, h);
}

static int ge_decompress(ge* r, const uint8_t s[32]) {
  fe y, u, v, v3, x, vxx, chk;
  fe one; fe_set(one, 1);
  fe_frombytes(y, s);
  fe_sq(u, y); fe_mul(v, u, FE_D); fe_sub(u, u, one); fe_add(v, v, one);
  fe_sq(v3, v); fe_mul(v3, v3, v);
  fe_sq(x, v3); fe_mul(x, x, v); fe_mul(x, x, u);   /* u v^7 */
  fe_pow22523(x, x);
  fe_mul(x, x, v3); fe_mul(x, x, u);               /* u v^3 (u v^7)^((p-5)/8) */
  fe_sq(vxx, x); fe_mul(vxx, vxx, v);
  fe_sub(chk, vxx, u);
  if (!fe_iszero(chk)) {
    fe_add(chk, vxx, u);
    if (!fe_iszero(chk)) return 0;
    fe_mul(x, x, FE_SQRTM1);
  }
  int sign = s[31] >> 7;
  if (fe_iszero(x) && sign) return 0;
  if (fe_isodd(x) != sign) fe_neg(x, x);
  fe_copy(r->X, x); fe_copy(r->Y, y); fe_set(r->Z, 1); fe_mul(r->T, x, y);
  return 1;
}

static void init_tables(void) {
  fe_frombytes(FE_D, D_BYTES);
  fe_add(FE_2D, FE_D, FE_D);
  fe_frombytes(FE_SQRTM1, SQRTM1_BYTES);
  ge_decompress(&GE_B, BY_BYTES);
  ge base = GE_B;
  for (int i = 0; i < 64; i++) {
    ge_ident(&BTAB[i][0]);
    for (int j = 1; j < 16; j++) ge_add(&BTAB[i][j], &BTAB[i][j - 1], &base);
    ge_add(&base, &BTAB[i][15], &base); /* 16 * base */
  }
}

/* r = k * B, k = 32 LE bytes (any 256-bit value) */
static void ge_scalarmult_base(ge* r, const uint8_t k[32]) {
  ge_ident(r);
  for (int i = 0; i < 64; i++) {
    int nib = (k[i / 2] >> (4 * (i & 1))) & 15;
    if (nib) ge_add(r, r, &BTAB[i][nib]);
  }
}
/* r = k * P, fixed 4-bit windows MSB first */
static void ge_scalarmult(ge* r, const uint8_t k[32], const ge* p) {
  ge tab[16];
  ge_ident(&tab[0]);
  tab[1] = *p;
  for (int j = 2; j < 16; j++) ge_add(&tab[j], &tab[j - 1], p);
  ge_ident(r);
  for (int i = 63; i >= 0; i--) {
    ge_dbl(r, r); ge_dbl(r, r); ge_dbl(r, r); ge_dbl(r, r);
    int nib = (k[i / 2] >> (4 * (i & 1))) & 15;
    if (nib) ge_add(r, r, &tab[nib]);
  }
}
static void ge_affine_bytes(const ge* p, uint8_t x[32], uint8_t y[32]) {
  fe zi, t;
  fe_invert(zi, p->Z);
  fe_mul(t, p->X, zi); fe_tobytes(x, t);
  fe_mul(t, p->Y, zi); fe_tobytes(y, t);
}
/* affine bytes of k points with one inversion (Montgomery's trick) */
static void ge_affine_bytes_batch(const ge* const* ps, int k, uint8_t (*xy)[32]) {
  fe pre[8], inv, t;
  fe_copy(pre[0], ps[0]->Z);
  for (int i = 1; i < k; i++) fe_mul(pre[i], pre[i - 1], ps[i]->Z);
  fe_invert(inv, pre[k - 1]);
  for (int i = k - 1; i >= 0; i--) {
    fe zi;
    if (i) { fe_mul(zi, inv, pre[i - 1]); fe_mul(inv, inv, ps[i]->Z); } else fe_copy(zi, inv);
    fe_mul(t, ps[i]->X, zi); fe_tobytes(xy[2 * i], t);
    fe_mul(t, ps[i]->Y, zi); fe_tobytes(xy[2 * i + 1], t);
  }
}

/* ---------------------------------------------------------------- scalars mod l */
static const uint64_t L64[4] = {0x5812631a5cf5d3edULL, 0x14def9dea2f79cd6ULL, 0, 0x1000000000000000ULL};

/* x (n little-endian 64-bit words) mod l by bitwise shift-and-subtract: slow, obviously correct */
static void sc_mod_l(uint64_t r[4], const uint64_t* x, int nwords) {
  uint64_t a[4] = {0, 0, 0, 0};
  for (int bit = nwords * 64 - 1; bit >= 0; bit--) {
    /* a = 2a + bit ; a < 2l < 2^254 so no overflow */
    uint64_t in = (x[bit / 64] >> (bit % 64)) & 1;
    a[3] = (a[3] << 1) | (a[2] >> 63); a[2] = (a[2] << 1) | (a[1] >> 63); a[1] = (a[1] << 1) | (a[0] >> 63); a[0] = (a[0] << 1) | in;
    /* if a >= l then a -= l */
    uint64_t d[4]; uint64_t borrow = 0;
    for (int i = 0; i < 4; i++) {
      u128 t = (u128)a[i] - L64[i] - borrow;
      d[i] = (uint64_t)t; borrow = (uint64_t)(t >> 64) & 1;
    }
    if (!borrow) { a[0] = d[0]; a[1] = d[1]; a[2] = d[2]; a[3] = d[3]; }
  }
  for (int i = 0; i < 4; i++) r[i] = a[i];
}
static void load_words(uint64_t* w, const uint8_t* s, int nwords) {
  for (int i = 0; i < nwords; i++) { w[i] = 0; for (int k = 7; k >= 0; k--) w[i] = (w[i] << 8) | s[8 * i + k]; }
}
static void store_words(uint8_t* s, const uint64_t* w, int nwords) {
  for (int i = 0; i < nwords; i++) for (int k = 0; k < 8; k++) s[8 * i + k] = (uint8_t)(w[i] >> (8 * k));
}
void tmxo_sc_reduce512(const uint8_t in[64], uint8_t out[32]) {
  uint64_t x[8], r[4];
  load_words(x, in, 8); sc_mod_l(r, x, 8); store_words(out, r, 4);
}
/* out = (a*b + c) mod l */
static void sc_muladd(uint8_t out[32], const uint8_t a[32], const uint8_t b[32], const uint8_t c[32]) {
  uint64_t x[4], y[4], z[4], p[9] = {0};
  load_words(x, a, 4); load_words(y, b, 4); load_words(z, c, 4);
  for (int i = 0; i < 4; i++) {
    uint64_t carry = 0;
    for (int j = 0; j < 4; j++) {
      u128 t = (u128)x[i] * y[j] + p[i + j] + carry;
      p[i + j] = (uint64_t)t; carry = (uint64_t)(t >> 64);
    }
    p[i + 4] += carry;
  }
  uint64_t carry = 0;
  for (int i = 0; i < 9; i++) {
    u128 t = (u128)p[i] + (i < 4 ? z[i] : 0) + carry;
    p[i] = (uint64_t)t; carry = (uint64_t)(t >> 64);
  }
  uint64_t r[4];
  sc_mod_l(r, p, 9); store_words(out, r, 4);
}
static int sc_is_canonical(const uint8_t s[32]) {
  uint64_t w[4]; load_words(w, s, 4);
  for (int i = 3; i >= 0; i--) { if (w[i] < L64[i]) return 1; if (w[i] > L64[i]) return 0; }
  return 0;
}

/* ---------------------------------------------------------------- public entry points */
void tmxo_eddsa_trace_lane(const uint8_t pk[32], const uint8_t sig[64], const uint8_t* msg, size_t len, tmxo_eddsa_trace* out) {
  pthread_once(&once, init_tables);
  memset(out, 0, sizeof *out);
  tmxo_sha512_3(sig, 32, pk, 32, msg, len, out->digest);
  tmxo_sc_reduce512(out->digest, out->h);
  ge A, R, sB, hA, sum;
  int okA = ge_decompress(&A, pk), okR = ge_decompress(&R, sig);
  out->decode_ok = (uint32_t)(okA && okR);
  if (!out->decode_ok) return; /* points stay zero, ok = 0 (same convention as oracle/py) */
  ge_scalarmult_base(&sB, sig + 32);
  ge_scalarmult(&hA, out->h, &A);
  ge_add(&sum, &R, &hA);
  const ge* ps[5] = {&A, &R, &sB, &hA, &sum};
  ge_affine_bytes_batch(ps, 5, out->pt);
  out->ok = (memcmp(out->pt[4], out->pt[8], 64) == 0) && sc_is_canonical(sig + 32);
}

static void clamp_expand(const uint8_t seed[32], uint8_t a[32], uint8_t prefix[32]) {
  uint8_t h[64];
  tmxo_sha512(seed, 32, h);
  memcpy(a, h, 32); memcpy(prefix, h + 32, 32);
  a[0] &= 248; a[31] &= 127; a[31] |= 64;
}
static void ge_compress(uint8_t s[32], const ge* p) {
  uint8_t x[32];
  ge_affine_bytes(p, x, s);
  s[31] |= (uint8_t)((x[0] & 1) << 7);
}
void tmxo_ed25519_pubkey(const uint8_t seed[32], uint8_t pk[32]) {
  pthread_once(&once, init_tables);
  uint8_t a[32], prefix[32]; ge A;
  clamp_expand(seed, a, prefix);
  ge_scalarmult_base(&A, a); ge_compress(pk, &A);
}
/* RFC 8032 §5.1.6 -- used only to manufacture synthetic test inputs */
void tmxo_ed25519_sign(const uint8_t seed[32], const uint8_t* msg, size_t len, uint8_t sig[64]) {
  pthread_once(&once, init_tables);
  uint8_t a[32], prefix[32], pk[32], rd[64], r[32], hd[64], h[32]; ge A, Rp;
  clamp_expand(seed, a, prefix);
  ge_scalarmult_base(&A, a); ge_compress(pk, &A);
  tmxo_sha512_3(prefix, 32, msg, len, 0, 0, rd);
  tmxo_sc_reduce512(rd, r);
  ge_scalarmult_base(&Rp, r); ge_compress(sig, &Rp);
  tmxo_sha512_3(sig, 32, pk, 32, msg, len, hd);
  tmxo_sc_reduce512(hd, h);
  sc_muladd(sig + 32, h, a, r);
}
void tmxo_dummy(uint8_t pk[32], uint8_t sig[64]) {
  /* DUMMY_PUBLIC_KEY / DUMMY_SIGNATURE (reference conversion.rs:3-5): keypair of seed 01x32 signing 00x32 */
  uint8_t seed[32], msg[32];
  memset(seed, 1, 32); memset(msg, 0, 32);
  tmxo_ed25519_pubkey(seed, pk); tmxo_ed25519_sign(seed, msg, 32, sig);
}

/* ================================================================================================ Level-2 ladder rows
 * TEST INFRASTRUCTURE.  DESIGN.md "Level-2 trace rows" (this build's own specification: the reference's Curta AIR is not observable).
 * One ladder = 256 rows of 65 elements for k * P, most significant bit first:
 *   [0] bit b_r = bit (255 - r) of k | [1..16] acc_r | [17..32] dbl_r = 2 acc_r | [33..48] add_r = dbl_r + P | [49..64] nxt_r = b_r ? add_r : dbl_r
 * every point as canonical affine (x, y), eight little-endian u32 limbs each; acc_0 = (0, 1), acc_{r+1} = nxt_r, nxt_255 = k * P.
 * Generator: affine twisted-Edwards formulas with one field inversion per operation (slow and independent of the HIP path, which keeps
 * projective coordinates and inverts in batches).  Checker: the same laws cross-multiplied, no inversion. */
static void fe_words(uint64_t out[8], const fe a) {
  uint8_t b[32]; fe_tobytes(b, a);
  for (int k = 0; k < 8; k++) out[k] = (uint64_t)b[4 * k] | ((uint64_t)b[4 * k + 1] << 8) | ((uint64_t)b[4 * k + 2] << 16) | ((uint64_t)b[4 * k + 3] << 24);
}
static int fe_from_limbs_checked(fe o, const uint64_t w[8]) { /* 0 if a limb is not a u32 or the value is not canonical */
  uint8_t b[32], c[32];
  for (int k = 0; k < 8; k++) { if (w[k] >> 32) return 0; for (int j = 0; j < 4; j++) b[4 * k + j] = (uint8_t)(w[k] >> (8 * j)); }
  if (b[31] & 0x80) return 0;
  fe_frombytes(o, b); fe_tobytes(c, o);
  return memcmp(b, c, 32) == 0;
}
/* affine addition law of -x^2 + y^2 = 1 + d x^2 y^2 (complete: d is not a square) */
static void aff_add(fe x3, fe y3, const fe x1, const fe y1, const fe x2, const fe y2) {
  fe a, b, c, t, one, den;
  fe_set(one, 1);
  fe_mul(a, x1, y2); fe_mul(b, y1, x2); fe_mul(c, a, b); fe_mul(c, c, FE_D);      /* c = d x1 x2 y1 y2 */
  fe_add(t, a, b); fe_add(den, one, c); fe_invert(den, den); fe_mul(x3, t, den);
  fe_mul(a, y1, y2); fe_mul(b, x1, x2); fe_add(t, a, b); fe_sub(den, one, c); fe_invert(den, den); fe_mul(y3, t, den);
}
void tmxo_trace_ladder(const uint8_t k[32], const uint8_t px[32], const uint8_t py[32], uint64_t* rows /* 256 * 65 */) {
  pthread_once(&once, init_tables);
  fe x, y, Px, Py, dx, dy, ax, ay;
  fe_set(x, 0); fe_set(y, 1);
  fe_frombytes(Px, px); fe_frombytes(Py, py);
  for (int r = 0; r < 256; r++) {
    uint64_t* row = rows + 65 * r;
    const int bit = (k[(255 - r) / 8] >> ((255 - r) & 7)) & 1;
    row[0] = (uint64_t)bit;
    fe_words(row + 1, x); fe_words(row + 9, y);
    aff_add(dx, dy, x, y, x, y);
    fe_words(row + 17, dx); fe_words(row + 25, dy);
    aff_add(ax, ay, dx, dy, Px, Py);
    fe_words(row + 33, ax); fe_words(row + 41, ay);
    if (bit) { fe_copy(x, ax); fe_copy(y, ay); } else { fe_copy(x, dx); fe_copy(y, dy); }
    fe_words(row + 49, x); fe_words(row + 57, y);
  }
}
/* x3 (1 + d x1 x2 y1 y2) == x1 y2 + y1 x2  and  y3 (1 - d x1 x2 y1 y2) == y1 y2 + x1 x2 */
static int aff_add_holds(const fe x3, const fe y3, const fe x1, const fe y1, const fe x2, const fe y2) {
  fe a, b, c, l, r, one;
  fe_set(one, 1);
  fe_mul(a, x1, y2); fe_mul(b, y1, x2); fe_mul(c, a, b); fe_mul(c, c, FE_D);
  fe_add(r, a, b); fe_add(l, one, c); fe_mul(l, l, x3); fe_sub(l, l, r);
  if (!fe_iszero(l)) return 0;
  fe_mul(a, y1, y2); fe_mul(b, x1, x2); fe_add(r, a, b); fe_sub(l, one, c); fe_mul(l, l, y3); fe_sub(l, l, r);
  return fe_iszero(l);
}
/* 0 = every constraint holds; otherwise 1000 * (row + 1) + the number of the violated constraint */
int tmxo_trace_ladder_check(const uint64_t* rows, const uint8_t k[32], const uint8_t px[32], const uint8_t py[32], const uint8_t rx[32],
                            const uint8_t ry[32]) {
  pthread_once(&once, init_tables);
  fe P[2], acc[2], dbl[2], add[2], nxt[2], prev[2];
  fe_frombytes(P[0], px); fe_frombytes(P[1], py);
  fe_set(prev[0], 0); fe_set(prev[1], 1);
  for (int r = 0; r < 256; r++) {
    const uint64_t* row = rows + 65 * r;
    const int e = 1000 * (r + 1);
    if (row[0] > 1) return e + 1;
    if (row[0] != (uint64_t)((k[(255 - r) / 8] >> ((255 - r) & 7)) & 1)) return e + 2;            /* the bits compose the scalar */
    if (!fe_from_limbs_checked(acc[0], row + 1) || !fe_from_limbs_checked(acc[1], row + 9) || !fe_from_limbs_checked(dbl[0], row + 17) ||
        !fe_from_limbs_checked(dbl[1], row + 25) || !fe_from_limbs_checked(add[0], row + 33) || !fe_from_limbs_checked(add[1], row + 41) ||
        !fe_from_limbs_checked(nxt[0], row + 49) || !fe_from_limbs_checked(nxt[1], row + 57)) return e + 3;   /* canonical u32 limbs */
    fe t;
    fe_sub(t, acc[0], prev[0]); if (!fe_iszero(t)) return e + 4;                                    /* acc_0 = O, acc_r = nxt_{r-1} */
    fe_sub(t, acc[1], prev[1]); if (!fe_iszero(t)) return e + 4;
    if (!aff_add_holds(dbl[0], dbl[1], acc[0], acc[1], acc[0], acc[1])) return e + 5;              /* dbl = 2 acc */
    if (!aff_add_holds(add[0], add[1], dbl[0], dbl[1], P[0], P[1])) return e + 6;                  /* add = dbl + P */
    const fe* sel = row[0] ? add : dbl;
    fe_sub(t, nxt[0], sel[0]); if (!fe_iszero(t)) return e + 7;                                    /* nxt = bit ? add : dbl */
    fe_sub(t, nxt[1], sel[1]); if (!fe_iszero(t)) return e + 7;
    fe_copy(prev[0], nxt[0]); fe_copy(prev[1], nxt[1]);
  }
  fe R[2], t;
  fe_frombytes(R[0], rx); fe_frombytes(R[1], ry);
  fe_sub(t, prev[0], R[0]); if (!fe_iszero(t)) return 257000 + 8;                                  /* nxt_255 = the Level-1 point */
  fe_sub(t, prev[1], R[1]); if (!fe_iszero(t)) return 257000 + 8;
  return 0;
}
void tmxo_base_point(uint8_t x[32], uint8_t y[32]) {
  pthread_once(&once, init_tables);
  ge_affine_bytes(&GE_B, x, y);
}
