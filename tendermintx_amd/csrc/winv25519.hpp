// One field inversion per WAVE: Montgomery's trick across the 64 lanes, the single inversion in the limb-parallel form (fe16.hpp).
// Shared by the EdDSA finish (kernels.hip: k_ed_fin) and the Level-2 ladder rows (trace.hip: k_trace_ladder_pass2).
#pragma once
#include "fe16.hpp"
#include "fe25519.hpp"

namespace tmx {

__device__ __forceinline__ fe f16_row_to_fe(const uint32_t* row16) {
  uint32_t l16[16];
#pragma unroll
  for (int l = 0; l < 16; l++) l16[l] = row16[l];
  fe r;
  f16::to_limbs10(l16, r.v);
  return fe_carry32(r);
}
// Montgomery's trick ACROSS THE LANES OF A WAVE.  Every thread of k_ed_fin needs 1 / (Z_sB Z_hA Z_D) of its own lane; round 4 ran 64
// independent Bernstein-Yang inversions per wave (21 k of the kernel's 26 k instructions, all on the step's critical chain).  Here the
// wave shares ONE: inclusive prefix and suffix products of the lanes' values by log-depth scans (six steps each, a ten-limb product per
// step), the product of all 64 turned into the limb-parallel form (sixteen 16-bit limbs across a DPP row, the same element in all four
// rows) and inverted by the Fermat chain with f16::mul1 -- 265 products of ~38 instructions with all 64 lanes working on the one element
// -- then 1 / z_i = (1 / T) (z_0 .. z_i-1) (z_i+1 .. z_63): ~13 k instructions on the chain instead of ~21 k.
// z must be non-zero in EVERY lane (a zero would poison the wave): callers substitute 1 for lanes whose value is not a product of Z
// coordinates of points on the curve (dead lanes, undecodable keys -- their outputs are zeros anyway); the Z of a point on the curve is
// never zero (the addition law is complete).
__device__ __forceinline__ fe wave_batch_invert(const fe& z, uint32_t tid, int32_t* s_t, uint32_t* s_inv) {
  fe P = z, S = z;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    fe a, b;
#pragma unroll
    for (int i = 0; i < 10; i++) { a.v[i] = __shfl_up(P.v[i], off, 64); b.v[i] = __shfl_down(S.v[i], off, 64); }
    const fe pa = fe_mul(a, P), sb = fe_mul(S, b);
    P = fe_select(P, pa, tid >= (uint32_t)off);          // (every lane multiplies, the lanes without a partner keep theirs: no divergence)
    S = fe_select(S, sb, tid + (uint32_t)off < 64u);
  }
  fe E, X;  // exclusive prefix z_0 .. z_i-1 and exclusive suffix z_i+1 .. z_63
#pragma unroll
  for (int i = 0; i < 10; i++) { E.v[i] = __shfl_up(P.v[i], 1, 64); X.v[i] = __shfl_down(S.v[i], 1, 64); }
  E = fe_select(E, fe_one(), tid == 0);
  X = fe_select(X, fe_one(), tid == 63);
  if (tid == 63) {
#pragma unroll
    for (int i = 0; i < 10; i++) s_t[i] = P.v[i];
  }
  __syncthreads();
  const f16::Ctx c = f16::make_ctx((int)tid);
  const uint32_t t16 = f16::carry(f16::from_limbs10(s_t, c), c);  // the same element in all four rows, carried
  s_inv[tid] = f16::invert<true>(t16, c);
  __syncthreads();
  const fe inv_t = f16_row_to_fe(s_inv);
  return fe_mul(fe_mul(inv_t, E), X);
}
}  // namespace tmx
