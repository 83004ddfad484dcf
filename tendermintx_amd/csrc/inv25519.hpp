// Inversion mod p = 2^255 - 19 by Bernstein-Yang "safegcd" division steps (constant time: 20 batches of 30 divsteps; the same
// schedule as the 32-bit modular inverse of libsecp256k1, whose bound of 590 divsteps covers every 256-bit modulus).
// One inversion is ~13 k VALU instructions instead of the ~34 k of the 254-squaring Fermat chain (fe_invert) -- the chain that a wave
// of k_ed_fin executes once for its 64 lanes on the step's critical path.  Same function: the inverse is unique, 0 maps to 0.
// Values: nine signed 30-bit limbs; every transition matrix entry fits 31 bits, accumulations are v_mad_i64_i32.
#pragma once
#include "fe25519.hpp"

namespace tmx {

struct s30 {
  int32_t v[9];
};
constexpr int32_t S30_M = (1 << 30) - 1;
constexpr uint32_t P25519_INV30 = 0x179435e5u;  // p^-1 mod 2^30 (p = -19 mod 2^30); limbs of p: {-19, 0 x 7, 2^15}

// 30 division steps on the low words; returns the new zeta and the transition matrix t = (u v; q r), entries in (-2^30, 2^30]
TMX_DEV int32_t bys_divsteps30(int32_t zeta, uint32_t f, uint32_t g, int32_t t[4]) {
  uint32_t u = 1, v = 0, q = 0, r = 1;
#pragma unroll 6
  for (int i = 0; i < 30; i++) {
    uint32_t c1 = (uint32_t)(zeta >> 31);  // zeta < 0
    const uint32_t c2 = 0u - (g & 1u);     // g odd
    const uint32_t x = (f ^ c1) - c1, y = (u ^ c1) - c1, z = (v ^ c1) - c1;  // (f, u, v) negated where zeta < 0
    g += x & c2; q += y & c2; r += z & c2;
    c1 &= c2;
    zeta = (int32_t)((uint32_t)zeta ^ c1) - 1;
    f += g & c1; u += q & c1; v += r & c1;
    g >>= 1; u <<= 1; v <<= 1;
  }
  t[0] = (int32_t)u; t[1] = (int32_t)v; t[2] = (int32_t)q; t[3] = (int32_t)r;
  return zeta;
}
// (d, e) <- t (d, e) / 2^30 mod p   (multiples of p are added so that the division is exact)
TMX_DEV void bys_update_de(s30& d, s30& e, const int32_t t[4]) {
  const int32_t u = t[0], v = t[1], q = t[2], r = t[3];
  const int32_t sd = d.v[8] >> 31, se = e.v[8] >> 31;
  int32_t md = (u & sd) + (v & se), me = (q & sd) + (r & se);
  int64_t cd = (int64_t)u * d.v[0] + (int64_t)v * e.v[0];
  int64_t ce = (int64_t)q * d.v[0] + (int64_t)r * e.v[0];
  md -= (int32_t)((P25519_INV30 * (uint32_t)cd + (uint32_t)md) & (uint32_t)S30_M);
  me -= (int32_t)((P25519_INV30 * (uint32_t)ce + (uint32_t)me) & (uint32_t)S30_M);
  cd += (int64_t)(-19) * md;
  ce += (int64_t)(-19) * me;
  cd >>= 30; ce >>= 30;
#pragma unroll
  for (int i = 1; i < 9; i++) {
    cd += (int64_t)u * d.v[i] + (int64_t)v * e.v[i];
    ce += (int64_t)q * d.v[i] + (int64_t)r * e.v[i];
    if (i == 8) { cd += (int64_t)md << 15; ce += (int64_t)me << 15; }  // limb 8 of p is 2^15, limbs 1..7 are 0
    d.v[i - 1] = (int32_t)cd & S30_M; cd >>= 30;
    e.v[i - 1] = (int32_t)ce & S30_M; ce >>= 30;
  }
  d.v[8] = (int32_t)cd; e.v[8] = (int32_t)ce;
}
// (f, g) <- t (f, g) / 2^30   (exact)
TMX_DEV void bys_update_fg(s30& f, s30& g, const int32_t t[4]) {
  const int32_t u = t[0], v = t[1], q = t[2], r = t[3];
  int64_t cf = (int64_t)u * f.v[0] + (int64_t)v * g.v[0];
  int64_t cg = (int64_t)q * f.v[0] + (int64_t)r * g.v[0];
  cf >>= 30; cg >>= 30;
#pragma unroll
  for (int i = 1; i < 9; i++) {
    cf += (int64_t)u * f.v[i] + (int64_t)v * g.v[i];
    cg += (int64_t)q * f.v[i] + (int64_t)r * g.v[i];
    f.v[i - 1] = (int32_t)cf & S30_M; cf >>= 30;
    g.v[i - 1] = (int32_t)cg & S30_M; cg >>= 30;
  }
  f.v[8] = (int32_t)cf; g.v[8] = (int32_t)cg;
}
// r in (-2p, p) -> sign(f) * r mod p in [0, p), limbs in [0, 2^30)
TMX_DEV void bys_normalize(s30& r, int32_t sign) {
  const int32_t PM[9] = {-19, 0, 0, 0, 0, 0, 0, 0, 1 << 15};
  int32_t cond_add = r.v[8] >> 31;
#pragma unroll
  for (int i = 0; i < 9; i++) r.v[i] += PM[i] & cond_add;
  const int32_t cond_negate = sign >> 31;
#pragma unroll
  for (int i = 0; i < 9; i++) r.v[i] = (r.v[i] ^ cond_negate) - cond_negate;
#pragma unroll
  for (int i = 0; i < 8; i++) { r.v[i + 1] += r.v[i] >> 30; r.v[i] &= S30_M; }
  cond_add = r.v[8] >> 31;
#pragma unroll
  for (int i = 0; i < 9; i++) r.v[i] += PM[i] & cond_add;
#pragma unroll
  for (int i = 0; i < 8; i++) { r.v[i + 1] += r.v[i] >> 30; r.v[i] &= S30_M; }
}

TMX_DEV fe fe_invert_safegcd(const fe& z) {
  uint32_t w[8];
  fe_to_words(z, w);  // canonical, < p
  s30 d, e, f, g;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    const int s = 30 * i, wi = s >> 5, sh = s & 31;
    uint64_t two = (uint64_t)w[wi] | ((wi + 1 < 8) ? ((uint64_t)w[wi + 1] << 32) : 0);
    g.v[i] = (int32_t)((uint32_t)(two >> sh) & (uint32_t)S30_M);
    d.v[i] = 0; e.v[i] = i == 0 ? 1 : 0;
    f.v[i] = i == 0 ? -19 : (i == 8 ? (1 << 15) : 0);
  }
  int32_t zeta = -1;
#pragma unroll 1
  for (int it = 0; it < 20; it++) {
    int32_t t[4];
    zeta = bys_divsteps30(zeta, (uint32_t)f.v[0], (uint32_t)g.v[0], t);
    bys_update_de(d, e, t);
    bys_update_fg(f, g, t);
  }
  bys_normalize(d, f.v[8]);  // g = 0, f = +-1 (f = +-p and d = 0 for z = 0)
  uint32_t o[8];
#pragma unroll
  for (int k = 0; k < 8; k++) {  // bits [32k, 32k + 32) of sum d_i 2^(30 i)
    const int lo = 32 * k, i0 = lo / 30, sh = lo - 30 * i0;
    uint64_t acc = (uint64_t)(uint32_t)d.v[i0] >> sh;
    int have = 30 - sh;
    if (i0 + 1 < 9) { acc |= (uint64_t)(uint32_t)d.v[i0 + 1] << have; have += 30; }
    if (have < 32 && i0 + 2 < 9) acc |= (uint64_t)(uint32_t)d.v[i0 + 2] << have;
    o[k] = (uint32_t)acc;
  }
  return fe_carry32(fe_from_words(o));
}

}  // namespace tmx
