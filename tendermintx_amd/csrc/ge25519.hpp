// Twisted-Edwards (a = -1) group operations on edwards25519, one point per lane.
//
// Value semantics of the curve layer under `curta_eddsa_verify_sigs_conditional`
// (reference circuits/builder/verify.rs:248-259; plonky2x/starkyx, un-vendored): decode A and R,
// compute s*B and h*A, R + h*A, and hand back canonical affine coordinates.
//
// Formulas: Hisil-Wong-Carter-Dawson 2008 extended coordinates (unified add, dedicated doubling), with the
// usual split into a "completed" result (two fractions E/G... here (E,F,G,H) with x = E/G', see below)
// so that a doubling that feeds another doubling skips the T product.
#pragma once
#include "fe25519.hpp"

namespace tmx {

struct ge_ext {  // x = X/Z, y = Y/Z, T = XY/Z
  fe X, Y, Z, T;
};
struct ge_proj {  // x = X/Z, y = Y/Z
  fe X, Y, Z;
};
struct ge_comp {  // completed: x = E/G, y = H/F   ->  X = E*F, Y = G*H, Z = F*G, T = E*H
  fe E, F, G, H;
};
struct ge_cached {  // addend form: (Y+X, Y-X, Z, 2dT)
  fe YpX, YmX, Z, T2d;
};
struct ge_affc {  // affine addend (Z = 1): (y+x, y-x, 2dxy)
  fe ypx, ymx, xy2d;
};

TMX_DEV ge_ext ge_identity() {
  ge_ext r;
  r.X = fe_zero(); r.Y = fe_one(); r.Z = fe_one(); r.T = fe_zero();
  return r;
}
TMX_DEV ge_proj comp_to_proj(const ge_comp& c) {
  ge_proj r;
  r.X = fe_mul(c.E, c.F); r.Y = fe_mul(c.G, c.H); r.Z = fe_mul(c.F, c.G);
  return r;
}
TMX_DEV ge_ext comp_to_ext(const ge_comp& c) {
  ge_ext r;
  r.X = fe_mul(c.E, c.F); r.Y = fe_mul(c.G, c.H); r.Z = fe_mul(c.F, c.G); r.T = fe_mul(c.E, c.H);
  return r;
}
TMX_DEV ge_proj ext_to_proj(const ge_ext& p) {
  ge_proj r;
  r.X = p.X; r.Y = p.Y; r.Z = p.Z;
  return r;
}
TMX_DEV ge_cached ext_to_cached(const ge_ext& p) {
  ge_cached r;
  r.YpX = fe_add(p.Y, p.X); r.YmX = fe_sub(p.Y, p.X); r.Z = p.Z; r.T2d = fe_mul(p.T, K_2D);
  return r;  // YpX / YmX stay 2x-lazy: still legal product operands (<= 3.3x)
}

// 2*(X:Y:Z): A = X^2, B = Y^2, C = 2Z^2; H = A+B, E = H-(X+Y)^2, G = A-B, F = C+G
// (all four are the negatives of the textbook dbl-2008-hwcd values, which leaves every product unchanged)
TMX_DEV ge_comp ge_double(const ge_proj& p) {
  fe a = fe_sq(p.X), b = fe_sq(p.Y);
  fe c = fe_sq2(p.Z);  // carried, so that F below stays within the 3.3x product bound
  fe s = fe_sq(fe_add(p.X, p.Y));
  ge_comp r;
  r.H = fe_add(a, b);
  r.E = fe_sub(r.H, s);
  r.G = fe_sub(a, b);
  r.F = fe_add(c, r.G);  // 1x + 2x = 3x
  return r;
}

// p + q, q in cached form (add-2008-hwcd-3)
TMX_DEV ge_comp ge_add_cached(const ge_ext& p, const ge_cached& q) {
  fe a = fe_mul(fe_sub(p.Y, p.X), q.YmX);
  fe b = fe_mul(fe_add(p.Y, p.X), q.YpX);
  fe c = fe_mul(p.T, q.T2d);
  fe d = fe_mul(p.Z, q.Z);
  fe d2 = fe_add(d, d);
  ge_comp r;
  r.E = fe_sub(b, a); r.F = fe_sub(d2, c); r.G = fe_add(d2, c); r.H = fe_add(b, a);
  return r;
}
// p - q
TMX_DEV ge_comp ge_sub_cached(const ge_ext& p, const ge_cached& q) {
  fe a = fe_mul(fe_sub(p.Y, p.X), q.YpX);
  fe b = fe_mul(fe_add(p.Y, p.X), q.YmX);
  fe c = fe_mul(p.T, q.T2d);
  fe d = fe_mul(p.Z, q.Z);
  fe d2 = fe_add(d, d);
  ge_comp r;
  r.E = fe_sub(b, a); r.F = fe_add(d2, c); r.G = fe_sub(d2, c); r.H = fe_add(b, a);
  return r;
}
// p + q, q affine addend
TMX_DEV ge_comp ge_add_affc(const ge_ext& p, const ge_affc& q) {
  fe a = fe_mul(fe_sub(p.Y, p.X), q.ymx);
  fe b = fe_mul(fe_add(p.Y, p.X), q.ypx);
  fe c = fe_mul(p.T, q.xy2d);
  fe d2 = fe_add(p.Z, p.Z);
  ge_comp r;
  r.E = fe_sub(b, a); r.F = fe_sub(d2, c); r.G = fe_add(d2, c); r.H = fe_add(b, a);
  return r;
}
TMX_DEV ge_comp ge_sub_affc(const ge_ext& p, const ge_affc& q) {
  fe a = fe_mul(fe_sub(p.Y, p.X), q.ypx);
  fe b = fe_mul(fe_add(p.Y, p.X), q.ymx);
  fe c = fe_mul(p.T, q.xy2d);
  fe d2 = fe_add(p.Z, p.Z);
  ge_comp r;
  r.E = fe_sub(b, a); r.F = fe_add(d2, c); r.G = fe_sub(d2, c); r.H = fe_add(b, a);
  return r;
}

// RFC 8032 §5.1.3 decoding of 32 little-endian bytes (as 8 words).  y is taken mod p (bit 255 = sign of x).
// Returns false if no square root exists or x = 0 with the sign bit set.
TMX_DEV bool ge_decode(const uint32_t w[8], ge_ext& out) {
  fe y = fe_carry32(fe_from_words(w));  // unsigned 26/25-bit chunks -> signed carried limbs
  fe yy = fe_sq(y);
  fe u = fe_sub(yy, fe_one());                 // y^2 - 1
  fe v = fe_add(fe_mul(yy, K_D), fe_one());    // d y^2 + 1
  fe v3 = fe_mul(fe_sq(v), v);
  fe uv7 = fe_mul(fe_mul(fe_sq(v3), v), u);
  fe x = fe_mul(fe_mul(fe_pow_p58(uv7), v3), u);   // u v^3 (u v^7)^((p-5)/8)
  fe vxx = fe_mul(fe_sq(x), v);
  bool ok_direct = fe_is_zero(fe_sub(vxx, u));
  bool ok_flipped = fe_is_zero(fe_add(vxx, u));
  fe xi = fe_mul(x, K_SQRTM1);
  x = fe_select(x, xi, !ok_direct);
  bool sign = (w[7] >> 31) & 1;
  uint32_t xw[8];
  fe_to_words(x, xw);
  bool x_zero = words_is_zero(xw);
  bool x_odd = xw[0] & 1;
  x = fe_select(x, fe_neg(x), x_odd != sign);
  out.X = x; out.Y = y; out.Z = fe_one(); out.T = fe_mul(x, y);
  return (ok_direct || ok_flipped) && !(x_zero && sign);
}

}  // namespace tmx
