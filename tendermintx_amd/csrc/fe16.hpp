// Limb-parallel arithmetic mod p = 2^255 - 19 for the latency-bound chains (key decoding, the 252 doublings behind the per-key
// tables): ONE field element = sixteen 16-bit limbs in the sixteen lanes of a DPP row, so a wave holds four elements -- the four
// coordinates of one point -- in a single VGPR and a multiplication is ~60 instructions instead of the ~250 of the limbs-in-registers
// form (a lone wave issues one VALU instruction per 4-5 cycles whatever it is, so chain latency = instruction count).
//
// Bounds: "carried" = every limb <= F16_C (2^16 + 2280); mul() takes limbs up to 441505 (a * 38 must stay below 2^24) and returns
// carried limbs; add() is plain, sub() adds a multiple of p whose limbs dominate the subtrahend's.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tmx {
namespace f16 {

constexpr uint32_t F16_C = 65536 + 2280;

template <int S>
__device__ __forceinline__ uint32_t ror(uint32_t v) {  // lane k of a row <- lane (k - S) mod 16
  // (update_dpp with a zero "old": the form the compiler folds into the consuming v_mul_u32_u24 as a DPP operand)
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x120 + S, 0xf, 0xf, false);
}
template <int S>
__device__ __forceinline__ uint32_t bcast(uint32_t v) {  // every lane of a row <- lane S (row_newbcast)
  return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x150 + S, 0xf, 0xf, false);
}

struct Ctx {
  uint32_t fac[16];  // fac[s] = 38 on the lanes a rotation by s wraps around (2^256 = 38 mod p), else 1
  uint32_t b4, b8;   // this lane's limb of 4p / 8p, balanced so that every limb is 131070 / 262140 (limb 0: 130996 / 261992)
  uint32_t k, row;
};
__device__ __forceinline__ Ctx make_ctx(int tid) {
  Ctx c;
  c.k = tid & 15; c.row = (tid >> 4) & 3;
#pragma unroll
  for (int s = 0; s < 16; s++) c.fac[s] = c.k < (uint32_t)s ? 38u : 1u;
  c.b4 = c.k == 0 ? 130996u : 131070u;
  c.b8 = c.k == 0 ? 261992u : 262140u;
  return c;
}

template <int S>
__device__ __forceinline__ void mul_step(uint64_t& acc, uint32_t a, uint32_t b, const Ctx& c) {
  acc += (uint64_t)__umul24(ror<S>(a), c.fac[S]) * bcast<S>(b);  // lane k: a[k - S] * b[S], times 38 where k < S
}
// two carry rounds: 2^46 -> carried
__device__ __forceinline__ uint32_t carry_wide(uint64_t acc, const Ctx& c) {
  const uint32_t hi = (uint32_t)(acc >> 16);
  uint32_t r = (uint32_t)acc & 0xffffu;
  r += __umul24(ror<1>(hi & 0xffffu), c.fac[1]) + __umul24(ror<2>(hi >> 16), c.fac[2]);
  const uint32_t c2 = r >> 16;
  return (r & 0xffffu) + __umul24(ror<1>(c2), c.fac[1]);
}
// one carry round for a sum of a few loose limbs (< 2^24)
__device__ __forceinline__ uint32_t carry(uint32_t r, const Ctx& c) { return (r & 0xffffu) + __umul24(ror<1>(r >> 16), c.fac[1]); }

__device__ __forceinline__ uint32_t mul(uint32_t a, uint32_t b, const Ctx& c) {
  uint64_t acc = (uint64_t)a * bcast<0>(b);
  mul_step<1>(acc, a, b, c); mul_step<2>(acc, a, b, c); mul_step<3>(acc, a, b, c); mul_step<4>(acc, a, b, c);
  mul_step<5>(acc, a, b, c); mul_step<6>(acc, a, b, c); mul_step<7>(acc, a, b, c); mul_step<8>(acc, a, b, c);
  mul_step<9>(acc, a, b, c); mul_step<10>(acc, a, b, c); mul_step<11>(acc, a, b, c); mul_step<12>(acc, a, b, c);
  mul_step<13>(acc, a, b, c); mul_step<14>(acc, a, b, c); mul_step<15>(acc, a, b, c);
  return carry_wide(acc, c);
}

// each row's element in all four rows (v_permlane16_swap / v_permlane32_swap: gfx950)
struct Rows { uint32_t r0, r1, r2, r3; };
__device__ __forceinline__ Rows rows(uint32_t v) {
  const auto p = __builtin_amdgcn_permlane16_swap(v, v, false, false);        // (r0 r0 r2 r2), (r1 r1 r3 r3)
  const auto e = __builtin_amdgcn_permlane32_swap(p[0], p[0], false, false);  // (r0 x 4), (r2 x 4)
  const auto o = __builtin_amdgcn_permlane32_swap(p[1], p[1], false, false);  // (r1 x 4), (r3 x 4)
  return {e[0], o[0], e[1], o[1]};
}
__device__ __forceinline__ uint32_t row_pick(const Ctx& c, uint32_t v0, uint32_t v1, uint32_t v2, uint32_t v3) {
  uint32_t r = v0;  // (three selects on values that already exist: no branches)
  r = c.row == 1 ? v1 : r;
  r = c.row == 2 ? v2 : r;
  r = c.row == 3 ? v3 : r;
  return r;
}

// Point doubling on rows (X, Y, Z, X+Y) -> Q = rows (X3, Y3, Z3, T3) and the next (X3, Y3, Z3, X3+Y3); ref10's ge_p2_dbl + p1p1
// conversion: e = (X+Y)^2 - XX - YY, h = YY + XX, g = YY - XX, f = 2ZZ - g; X3 = e f, Y3 = h g, Z3 = g f, T3 = e h.
__device__ __forceinline__ uint32_t dbl(uint32_t V, uint32_t& Q, const Ctx& c) {
  const Rows s = rows(mul(V, V, c));  // XX, YY, ZZ, (X+Y)^2: carried
  const uint32_t h = s.r0 + s.r1;
  const uint32_t e = s.r3 + c.b8 - h;                // h <= 2 C < limbs of 8p
  const uint32_t g = s.r1 + c.b4 - s.r0;             // <= C + 131070
  const uint32_t f = s.r2 + s.r2 + s.r0 + c.b4 - s.r1;  // <= 3 C + 131070 = 334518
  Q = mul(row_pick(c, e, h, g, e), row_pick(c, f, g, f, h), c);
  const auto p = __builtin_amdgcn_permlane16_swap(Q, Q, false, false);  // (X X Z Z), (Y Y T T)
  const uint32_t xy = p[0] + p[1];                                      // (X+Y, X+Y, Z+T, Z+T)
  const auto t = __builtin_amdgcn_permlane32_swap(xy, xy, false, false);
  return c.row == 3 ? t[0] : Q;
}

// radix-2^25.5 limbs (signed, |limb| < 2^26: what the quad kernels store) -> this lane's 16-bit limb (< 2^17; lane 15 keeps the bits
// above 2^256).  p10 = the ten limbs of this row's coordinate.
__device__ __forceinline__ uint32_t from_limbs10(const int32_t* __restrict__ p10, const Ctx& c) {
  uint32_t out = 0;
  const int lo = 16 * (int)c.k;
#pragma unroll
  for (int j = 0; j < 10; j++) {
    const int off = (51 * j + 1) / 2;                                              // ceil(25.5 j)
    const uint32_t twop = j == 0 ? 0x7ffffdau : ((j & 1) ? 0x3fffffeu : 0x7fffffeu);  // + 2p: every limb positive, < 2^28
    const uint32_t f = (uint32_t)(p10[j] + (int32_t)twop);
    const int sh = off - lo;  // bit position of the limb relative to this lane's window
    uint32_t v = 0;
    if (sh >= 0 && sh < 16) v = f << sh;
    else if (sh < 0 && sh > -28) v = f >> (-sh);
    out += c.k == 15 ? v : (v & 0xffffu);
  }
  return out;
}
// sixteen loose limbs (< 2^18 each, all in registers) -> ten radix-2^25.5 limbs, non-negative, in the usual bounds
__device__ __forceinline__ void to_limbs10(const uint32_t l[16], int32_t out[10]) {
#pragma unroll
  for (int j = 0; j < 10; j++) {
    const int off = (51 * j + 1) / 2, nxt = (51 * (j + 1) + 1) / 2;
    uint32_t acc = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
      const int b0 = 16 * k;                        // l[k] covers bits [b0, b0 + 18)
      const int from = b0 > off ? b0 : off;         // first bit of the overlap with [off, nxt)
      const int to = (j == 9) ? b0 + 18 : (b0 + 18 < nxt ? b0 + 18 : nxt);
      if (to > from) {
        uint32_t v = l[k] >> (from - b0);
        if (to - from < 18 - (from - b0)) v &= (1u << (to - from)) - 1u;
        acc += v << (from - off);
      }
    }
    out[j] = (int32_t)acc;
  }
  const uint32_t top = (uint32_t)out[9] >> 25;  // 2^255 = 19
  out[9] &= 0x1ffffff;
  out[0] += (int32_t)(19u * top);
}

// constants as 16-bit limbs
__device__ __forceinline__ uint32_t const_d(uint32_t k) {
  constexpr uint16_t t[16] = {0x78a3, 0x1359, 0x4dca, 0x75eb, 0xd8ab, 0x4141, 0x0a4d, 0x0070, 0xe898, 0x7779, 0x4079, 0x8cc7, 0xfe73, 0x2b6f, 0x6cee, 0x5203};
  return t[k];
}
__device__ __forceinline__ uint32_t const_sqrtm1(uint32_t k) {
  constexpr uint16_t t[16] = {0xa0b0, 0x4a0e, 0x1b27, 0xc4ee, 0xe478, 0xad2f, 0x1806, 0x2f43, 0xd7a7, 0x3dfb, 0x0099, 0x2b4d, 0xdf0b, 0x4fc1, 0x2480, 0x2b83};
  return t[k];
}
// ---- ONE element held in all four rows (e.g. the product of the Z's a finish inverts): the four rows share the sixteen partial products
// of a limb -- row r takes a[k - s] b[s] for s = 4r .. 4r + 3 -- and the row sums are added across the rows; the result is again in all
// four rows.  ~38 instructions instead of 61: a 254-squaring inversion chain is a third shorter.  Inputs carried (<= F16_C).
__device__ __forceinline__ uint32_t mul1(uint32_t a, uint32_t b, const Ctx& c) {
  // a_r[k] = a[k - 4r] (times 38 where that index wraps), b_r[k] = b[k + 4r]: DPP row rotations applied to one row each (row_mask)
  int ar = (int)a, br = (int)b;
  ar = __builtin_amdgcn_update_dpp(ar, (int)a, 0x120 + 4, 0x2, 0xf, false);
  ar = __builtin_amdgcn_update_dpp(ar, (int)a, 0x120 + 8, 0x4, 0xf, false);
  ar = __builtin_amdgcn_update_dpp(ar, (int)a, 0x120 + 12, 0x8, 0xf, false);
  br = __builtin_amdgcn_update_dpp(br, (int)b, 0x120 + 12, 0x2, 0xf, false);
  br = __builtin_amdgcn_update_dpp(br, (int)b, 0x120 + 8, 0x4, 0xf, false);
  br = __builtin_amdgcn_update_dpp(br, (int)b, 0x120 + 4, 0x8, 0xf, false);
  const uint32_t a4 = __umul24((uint32_t)ar, c.k < 4 * c.row ? 38u : 1u), b4 = (uint32_t)br;
  // (a wrapped index is wrapped by exactly one of the two rotations, so the factors 38 never meet: a4 fac[j] < 2^24 as in mul())
  uint64_t acc = (uint64_t)a4 * bcast<0>(b4);
  mul_step<1>(acc, a4, b4, c); mul_step<2>(acc, a4, b4, c); mul_step<3>(acc, a4, b4, c);
  const uint32_t r = carry_wide(acc, c);  // this row's share, carried
  const auto p = __builtin_amdgcn_permlane16_swap(r, r, false, false);  // (r0 r0 r2 r2), (r1 r1 r3 r3)
  const uint32_t s2 = p[0] + p[1];
  const auto e = __builtin_amdgcn_permlane32_swap(s2, s2, false, false);  // (r0 + r1) x 4, (r2 + r3) x 4
  return carry(e[0] + e[1], c);
}
template <bool ONE>
__device__ __forceinline__ uint32_t mulx(uint32_t a, uint32_t b, const Ctx& c) {
  if constexpr (ONE) return mul1(a, b, c);
  else return mul(a, b, c);
}
template <bool ONE = false>
__device__ __forceinline__ uint32_t sqn(uint32_t a, int n, const Ctx& c) {
#pragma unroll 1
  for (int i = 0; i < n; i++) a = mulx<ONE>(a, a, c);
  return a;
}
// z^((p-5)/8) = z^(2^252 - 3): ref10's fe_pow22523 addition chain (252 squarings, 11 multiplications)
template <bool ONE = false>
__device__ __forceinline__ uint32_t pow_p58(uint32_t z, const Ctx& c) {
  uint32_t t0 = mulx<ONE>(z, z, c);                  // 2
  uint32_t t1 = mulx<ONE>(z, sqn<ONE>(t0, 2, c), c);      // 9
  t0 = mulx<ONE>(t0, t1, c);                         // 11
  t0 = mulx<ONE>(t1, mulx<ONE>(t0, t0, c), c);             // 31 = 2^5 - 1
  t0 = mulx<ONE>(sqn<ONE>(t0, 5, c), t0, c);              // 2^10 - 1
  t1 = mulx<ONE>(sqn<ONE>(t0, 10, c), t0, c);             // 2^20 - 1
  t1 = mulx<ONE>(sqn<ONE>(t1, 20, c), t1, c);             // 2^40 - 1
  t0 = mulx<ONE>(sqn<ONE>(t1, 10, c), t0, c);             // 2^50 - 1
  t1 = mulx<ONE>(sqn<ONE>(t0, 50, c), t0, c);             // 2^100 - 1
  t1 = mulx<ONE>(sqn<ONE>(t1, 100, c), t1, c);            // 2^200 - 1
  t0 = mulx<ONE>(sqn<ONE>(t1, 50, c), t0, c);             // 2^250 - 1
  return mulx<ONE>(sqn<ONE>(t0, 2, c), z, c);             // 2^252 - 3
}

__device__ __forceinline__ uint32_t const_2d(uint32_t k) {
  constexpr uint16_t t[16] = {0xf159, 0x26b2, 0x9b94, 0xebd6, 0xb156, 0x8283, 0x149a, 0x00e0, 0xd130, 0xeef3, 0x80f2, 0x198e, 0xfce7, 0x56df, 0xd9dc, 0x2406};
  return t[k];
}
// rows (X, Y, Z, T) -> rows (Y-X, Y+X, T, Z): the left operand of an addition's first level (limbs of P up to 3 * 2^16)
__device__ __forceinline__ uint32_t add_left(uint32_t P, const Ctx& c) {
  const Rows s = rows(P);
  return row_pick(c, s.r1 + c.b4 - s.r0, s.r1 + s.r0, s.r3, s.r2);
}
// cached form of a point, rows (Y-X, Y+X, 2dT, Z), carried
__device__ __forceinline__ uint32_t to_cached(uint32_t P, const Ctx& c) {
  const uint32_t one = c.k == 0 ? 1u : 0u;
  return mul(add_left(P, c), c.row == 2 ? const_2d(c.k) : one, c);
}
__device__ __forceinline__ uint32_t neg_cached(uint32_t q, const Ctx& c) {  // (Y+X, Y-X, -2dT, Z)
  const auto p = __builtin_amdgcn_permlane16_swap(q, q, false, false);   // (r0 r0 r2 r2), (r1 r1 r3 r3)
  return row_pick(c, p[1], p[0], c.b4 - q, q);
}
// ref10 ge_add: rows (X, Y, Z, T) + cached rows -> rows (X3, Y3, Z3, T3)
__device__ __forceinline__ uint32_t add_cached(uint32_t P, uint32_t q, const Ctx& c) {
  const Rows m = rows(mul(add_left(P, c), q, c));  // A, B, C, ZZ
  const uint32_t E = m.r1 + c.b4 - m.r0, H = m.r1 + m.r0, zz2 = m.r3 + m.r3;
  const uint32_t F = zz2 + c.b4 - m.r2, G = zz2 + m.r2;
  return mul(row_pick(c, E, G, F, E), row_pick(c, F, H, G, H), c);
}
// rows (X, Y, Z, T) -> (X, Y, Z, X+Y), what dbl() takes
__device__ __forceinline__ uint32_t with_xy(uint32_t Q, const Ctx& c) {
  const auto p = __builtin_amdgcn_permlane16_swap(Q, Q, false, false);
  const uint32_t xy = p[0] + p[1];
  const auto t = __builtin_amdgcn_permlane32_swap(xy, xy, false, false);
  return c.row == 3 ? t[0] : Q;
}

// z^(p-2) = z^(2^255 - 21): ref10's fe_invert chain (254 squarings, 11 multiplications)
template <bool ONE = false>
__device__ __forceinline__ uint32_t invert(uint32_t z, const Ctx& c) {
  uint32_t t0 = mulx<ONE>(z, z, c);                  // 2
  uint32_t t1 = mulx<ONE>(z, sqn<ONE>(t0, 2, c), c);      // 9
  t0 = mulx<ONE>(t0, t1, c);                         // 11
  t1 = mulx<ONE>(t1, mulx<ONE>(t0, t0, c), c);             // 31 = 2^5 - 1
  t1 = mulx<ONE>(sqn<ONE>(t1, 5, c), t1, c);              // 2^10 - 1
  uint32_t t2 = mulx<ONE>(sqn<ONE>(t1, 10, c), t1, c);    // 2^20 - 1
  t2 = mulx<ONE>(sqn<ONE>(t2, 20, c), t2, c);             // 2^40 - 1
  t1 = mulx<ONE>(sqn<ONE>(t2, 10, c), t1, c);             // 2^50 - 1
  t2 = mulx<ONE>(sqn<ONE>(t1, 50, c), t1, c);             // 2^100 - 1
  t2 = mulx<ONE>(sqn<ONE>(t2, 100, c), t2, c);            // 2^200 - 1
  t1 = mulx<ONE>(sqn<ONE>(t2, 50, c), t1, c);             // 2^250 - 1
  return mulx<ONE>(sqn<ONE>(t1, 5, c), t0, c);            // 2^255 - 32 + 11
}

}  // namespace f16
}  // namespace tmx
