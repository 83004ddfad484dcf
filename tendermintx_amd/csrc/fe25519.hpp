// GF(2^255 - 19) arithmetic for gfx950 wavefronts: one field element per lane, ten signed limbs in
// radix 2^25.5 (26,25,26,25,... bits) held in VGPRs, products accumulated with v_mad_i64_i32.
//
// Why this schedule (tools/microbench/valu_rates.hip on MI355X): v_mad_*64_*32 issues at half rate
// (4 cycles / wave64), 64-bit shifts and adds cost two full-rate ops.  A 10-limb schoolbook product is
// 100 mads (400 cycles) + ~110 cycles of carries; saturated 8x32 limbs need a carry-out per mad and end up
// level, 5x51 needs 4 mads per limb product.  Signed limbs let add/sub stay carry-free between products:
// a product tolerates operands up to 3.3x the carried bound.
//
// Replaces (value semantics only) the field layer under `curta_eddsa_verify_sigs_conditional`
// (reference circuits/builder/verify.rs:248-259; plonky2x / starkyx, un-vendored).
//
// Bounds contract ("c" = carried): c-limbs satisfy |even| <= 1.01*2^25, |odd| <= 1.01*2^24.
//   fe_mul / fe_sq : inputs up to 3.3x c-bound each, output c.
//   fe_add / fe_sub: limb-wise, no carry: bound(out) = bound(a) + bound(b).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tmx {

struct fe {
  int32_t v[10];
};

#define TMX_DEV __device__ __forceinline__
// A/B switch: -DTMX_FE_NOINLINE turns the field products into real functions (smaller code).  Measured on MI355X
// (bench.py, 256 x 128): 22 % slower for the quad path, 55 % slower for the one-lane path (call ABI spills) -> inlined.
#ifdef TMX_FE_NOINLINE
#define TMX_FE_FN __device__ __noinline__
#else
#define TMX_FE_FN __device__ __forceinline__
#endif

TMX_DEV fe fe_zero() {
  fe r;
#pragma unroll
  for (int i = 0; i < 10; i++) r.v[i] = 0;
  return r;
}
TMX_DEV fe fe_one() {
  fe r = fe_zero();
  r.v[0] = 1;
  return r;
}
TMX_DEV fe fe_add(const fe& a, const fe& b) {
  fe r;
#pragma unroll
  for (int i = 0; i < 10; i++) r.v[i] = a.v[i] + b.v[i];
  return r;
}
TMX_DEV fe fe_sub(const fe& a, const fe& b) {
  fe r;
#pragma unroll
  for (int i = 0; i < 10; i++) r.v[i] = a.v[i] - b.v[i];
  return r;
}
TMX_DEV fe fe_neg(const fe& a) {
  fe r;
#pragma unroll
  for (int i = 0; i < 10; i++) r.v[i] = -a.v[i];
  return r;
}
// r = c ? b : a   (lane-wise select, no divergence)
TMX_DEV fe fe_select(const fe& a, const fe& b, bool c) {
  fe r;
#pragma unroll
  for (int i = 0; i < 10; i++) r.v[i] = c ? b.v[i] : a.v[i];
  return r;
}

// Signed carry chain over ten 64-bit column sums -> carried limbs.  Two interleaved chains (limbs 0..4 and 5..9) halve the
// dependent depth: a lone wave per SIMD spends most of a 255-squaring exponentiation waiting on this chain.
//   (0,5) (1,6) (2,7) (3,8) (4,9: limb 9 wraps into limb 0 times 19) then (0,5) once more
TMX_DEV void fe_carry_step(int64_t h[10], int i) {
  const int bits = (i & 1) ? 25 : 26;
  int64_t c = (h[i] + ((int64_t)1 << (bits - 1))) >> bits;
  h[i] -= c << bits;
  if (i < 9) h[i + 1] += c; else h[0] += 19 * c;
}
TMX_DEV fe fe_carry_wide(int64_t h[10]) {
#pragma unroll
  for (int i = 0; i < 5; i++) { fe_carry_step(h, i); fe_carry_step(h, i + 5); }
  fe_carry_step(h, 0);
  fe_carry_step(h, 5);
  fe r;
#pragma unroll
  for (int i = 0; i < 10; i++) r.v[i] = (int32_t)h[i];
  return r;
}

TMX_FE_FN fe fe_mul(const fe f, const fe g) {
  int32_t g19[10], f2[10];
#pragma unroll
  for (int i = 0; i < 10; i++) {
    g19[i] = 19 * g.v[i];
    f2[i] = 2 * f.v[i];
  }
  int64_t h[10];
#pragma unroll
  for (int k = 0; k < 10; k++) h[k] = 0;
#pragma unroll
  for (int i = 0; i < 10; i++) {
#pragma unroll
    for (int j = 0; j < 10; j++) {
      const int k = i + j;
      const bool wrap = k >= 10;
      const bool both_odd = (i & 1) && (j & 1);
      const int32_t fi = both_odd ? f2[i] : f.v[i];
      const int32_t gj = wrap ? g19[j] : g.v[j];
      h[wrap ? k - 10 : k] += (int64_t)fi * gj;
    }
  }
  return fe_carry_wide(h);
}

template <bool DOUBLE_IT>
TMX_DEV fe fe_sq_impl(const fe& f) {
  int32_t f2[10], f19[10], f38[10];
#pragma unroll
  for (int i = 0; i < 10; i++) {
    f2[i] = 2 * f.v[i];
    f19[i] = 19 * f.v[i];
    f38[i] = 38 * f.v[i];  // only odd indices are used (|odd| <= 3.3*2^24 -> fits)
  }
  int64_t h[10];
#pragma unroll
  for (int k = 0; k < 10; k++) h[k] = 0;
#pragma unroll
  for (int i = 0; i < 10; i++) {
#pragma unroll
    for (int j = i; j < 10; j++) {
      const int k = i + j;
      const bool wrap = k >= 10;
      const bool both_odd = (i & 1) && (j & 1);
      const int32_t left = (i == j) ? f.v[i] : f2[i];
      const int32_t right = both_odd ? (wrap ? f38[j] : f2[j]) : (wrap ? f19[j] : f.v[j]);
      h[wrap ? k - 10 : k] += (int64_t)left * right;
    }
  }
  if (DOUBLE_IT) {
#pragma unroll
    for (int k = 0; k < 10; k++) h[k] += h[k];
  }
  return fe_carry_wide(h);
}
TMX_FE_FN fe fe_sq(const fe f) { return fe_sq_impl<false>(f); }
TMX_FE_FN fe fe_sq2(const fe f) { return fe_sq_impl<true>(f); }  // 2 f^2, carried

// n successive squarings
TMX_DEV fe fe_sqn(fe x, int n) {
  for (int i = 0; i < n; i++) x = fe_sq(x);
  return x;
}

// 32-bit signed carry pass: any lazy sum (<= ~30x) back to the carried bound
TMX_DEV fe fe_carry32(const fe& a) {
  fe r = a;
#pragma unroll
  for (int i = 0; i < 10; i++) {
    const int bits = (i & 1) ? 25 : 26;
    int32_t c = (r.v[i] + (1 << (bits - 1))) >> bits;
    r.v[i] -= c << bits;
    if (i < 9) r.v[i + 1] += c; else r.v[0] += 19 * c;
  }
  int32_t c = (r.v[0] + (1 << 25)) >> 26;
  r.v[0] -= c << 26;
  r.v[1] += c;
  return r;
}

// limbs from 32 little-endian bytes given as eight u32 words; bit 255 ignored
TMX_DEV fe fe_from_words(const uint32_t w[8]) {
  fe r;
  // limb i starts at bit ceil(25.5 * i): 0,26,51,77,102,128,153,179,204,230
  const int start[10] = {0, 26, 51, 77, 102, 128, 153, 179, 204, 230};
#pragma unroll
  for (int i = 0; i < 10; i++) {
    const int s = start[i], bits = (i & 1) ? 25 : 26;
    const int wi = s >> 5, sh = s & 31;
    uint64_t two = (uint64_t)w[wi] | ((wi + 1 < 8) ? ((uint64_t)w[wi + 1] << 32) : 0);
    uint32_t val = (uint32_t)(two >> sh) & ((1u << bits) - 1);
    if (i == 9) val &= (1u << 25) - 1;  // drops bit 255
    r.v[i] = (int32_t)val;
  }
  return r;
}

// canonical value (0 <= x < p) as eight little-endian u32 words.  Input: any lazy sum up to ~8x carried.
TMX_DEV void fe_to_words(const fe& a, uint32_t w[8]) {
  int32_t h[10];
#pragma unroll
  for (int i = 0; i < 10; i++) h[i] = a.v[i];
  // bring the limbs back to the carried bound first (32-bit signed carry pass)
#pragma unroll
  for (int i = 0; i < 10; i++) {
    const int bits = (i & 1) ? 25 : 26;
    int32_t c = (h[i] + (1 << (bits - 1))) >> bits;
    h[i] -= c << bits;
    if (i < 9) h[i + 1] += c; else h[0] += 19 * c;
  }
  // q = floor((h + 19) / 2^255) computed limb by limb, then h += 19 q and propagate floor carries
  int32_t q = (19 * h[9] + (1 << 24)) >> 25;
#pragma unroll
  for (int i = 0; i < 10; i++) q = (h[i] + q) >> ((i & 1) ? 25 : 26);
  h[0] += 19 * q;
#pragma unroll
  for (int i = 0; i < 10; i++) {
    const int bits = (i & 1) ? 25 : 26;
    int32_t c = h[i] >> bits;
    h[i] -= c << bits;
    if (i < 9) h[i + 1] += c;  // the carry out of limb 9 is the discarded 2^255 multiple
  }
  const int start[10] = {0, 26, 51, 77, 102, 128, 153, 179, 204, 230};
  uint64_t acc[8];
#pragma unroll
  for (int k = 0; k < 8; k++) acc[k] = 0;
#pragma unroll
  for (int i = 0; i < 10; i++) {
    const int wi = start[i] >> 5, sh = start[i] & 31;
    uint64_t v = (uint64_t)(uint32_t)h[i] << sh;
    acc[wi] |= v & 0xffffffffu;
    if (wi + 1 < 8) acc[wi + 1] |= v >> 32;
  }
#pragma unroll
  for (int k = 0; k < 8; k++) w[k] = (uint32_t)acc[k];
}

TMX_DEV bool words_is_zero(const uint32_t w[8]) {
  uint32_t o = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) o |= w[k];
  return o == 0;
}
TMX_DEV bool fe_is_zero(const fe& a) {
  uint32_t w[8];
  fe_to_words(a, w);
  return words_is_zero(w);
}
TMX_DEV bool fe_is_odd(const fe& a) {
  uint32_t w[8];
  fe_to_words(a, w);
  return w[0] & 1;
}

// z^(2^250 - 1) and z^11 (shared prefix of inversion and the square-root exponent)
TMX_DEV fe fe_pow2_250_1(const fe& z, fe& z11) {
  fe z2 = fe_sq(z);
  fe z9 = fe_mul(fe_sqn(z2, 2), z);
  z11 = fe_mul(z9, z2);
  fe z5 = fe_mul(fe_sq(z11), z9);                // 2^5 - 1
  fe z10 = fe_mul(fe_sqn(z5, 5), z5);            // 2^10 - 1
  fe z20 = fe_mul(fe_sqn(z10, 10), z10);         // 2^20 - 1
  fe z40 = fe_mul(fe_sqn(z20, 20), z20);         // 2^40 - 1
  fe z50 = fe_mul(fe_sqn(z40, 10), z10);         // 2^50 - 1
  fe z100 = fe_mul(fe_sqn(z50, 50), z50);        // 2^100 - 1
  fe z200 = fe_mul(fe_sqn(z100, 100), z100);     // 2^200 - 1
  return fe_mul(fe_sqn(z200, 50), z50);          // 2^250 - 1
}
TMX_DEV fe fe_invert(const fe& z) {  // z^(p-2) = z^(2^255 - 21)
  fe z11;
  fe t = fe_pow2_250_1(z, z11);
  return fe_mul(fe_sqn(t, 5), z11);
}
TMX_DEV fe fe_pow_p58(const fe& z) {  // z^((p-5)/8) = z^(2^252 - 3)
  fe z11;
  fe t = fe_pow2_250_1(z, z11);
  return fe_mul(fe_sqn(t, 2), z);
}

// ---- curve constants as limbs (constexpr-evaluated from their little-endian byte strings)
struct bytes32 {
  uint8_t b[32];
};
constexpr fe fe_const(const bytes32& s) {
  fe r{};
  const int start[10] = {0, 26, 51, 77, 102, 128, 153, 179, 204, 230};
  for (int i = 0; i < 10; i++) {
    const int bits = (i & 1) ? 25 : 26;
    uint32_t val = 0;
    for (int b = 0; b < bits; b++) {
      const int pos = start[i] + b;
      if (pos < 255) val |= (uint32_t)((s.b[pos >> 3] >> (pos & 7)) & 1) << b;
    }
    r.v[i] = (int32_t)val;
  }
  return r;
}
// d = -121665/121666, 2d, sqrt(-1) mod p  (RFC 8032 §5.1)
constexpr bytes32 K_D_BYTES = {{0xa3, 0x78, 0x59, 0x13, 0xca, 0x4d, 0xeb, 0x75, 0xab, 0xd8, 0x41, 0x41, 0x4d, 0x0a, 0x70, 0x00,
                                0x98, 0xe8, 0x79, 0x77, 0x79, 0x40, 0xc7, 0x8c, 0x73, 0xfe, 0x6f, 0x2b, 0xee, 0x6c, 0x03, 0x52}};
constexpr bytes32 K_2D_BYTES = {{0x59, 0xf1, 0xb2, 0x26, 0x94, 0x9b, 0xd6, 0xeb, 0x56, 0xb1, 0x83, 0x82, 0x9a, 0x14, 0xe0, 0x00,
                                 0x30, 0xd1, 0xf3, 0xee, 0xf2, 0x80, 0x8e, 0x19, 0xe7, 0xfc, 0xdf, 0x56, 0xdc, 0xd9, 0x06, 0x24}};
constexpr bytes32 K_SQRTM1_BYTES = {{0xb0, 0xa0, 0x0e, 0x4a, 0x27, 0x1b, 0xee, 0xc4, 0x78, 0xe4, 0x2f, 0xad, 0x06, 0x18, 0x43, 0x2f,
                                     0xa7, 0xd7, 0xfb, 0x3d, 0x99, 0x00, 0x4d, 0x2b, 0x0b, 0xdf, 0xc1, 0x4f, 0x80, 0x24, 0x83, 0x2b}};
constexpr bytes32 K_BY_BYTES = {{0x58, 0x66, 0x66, 0x66, 0x66, 0x66, 0x66, 0x66, 0x66, 0x66, 0x66, 0x66, 0x66, 0x66, 0x66, 0x66,
                                 0x66, 0x66, 0x66, 0x66, 0x66, 0x66, 0x66, 0x66, 0x66, 0x66, 0x66, 0x66, 0x66, 0x66, 0x66, 0x66}};
constexpr fe K_D = fe_const(K_D_BYTES);
constexpr fe K_2D = fe_const(K_2D_BYTES);
constexpr fe K_SQRTM1 = fe_const(K_SQRTM1_BYTES);

}  // namespace tmx
