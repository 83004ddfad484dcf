// Poseidon over Goldilocks and the Merkle-cap commitment of LDE'd columns ("commit primitives": what a
// plonky2-style prover does with the trace columns after the LDE -- the reference reaches it through plonky2x `prove`, reference
// circuits/skip.rs:119-133; plonky2 itself is absent from the reference tree, Cargo.lock:2957-2982).
// Definition (include/tmx.h; restated on the CPU by the test oracle): width 12, x^7, 4 + 22 + 4 rounds, circulant-plus-diagonal MDS, overwrite-mode sponge of rate 8,
// two_to_one, hash_or_noop leaves.  Round constants and MDS rows are DATA (injected through tmx_poseidon_set_constants; the defaults are
// the Poseidon paper's Grain-LFSR stream, not plonky2's table: parity unpinned), read with scalar loads -- wave-uniform addresses.
//
// One permutation per thread, the twelve state elements in 24 VGPRs as arbitrary 64-bit representatives of their classes ("lazy": only the
// stored digests are canonical).  This is pure 64-bit integer VALU work -- no MFMA (nothing is a dense contraction: the MDS layer is a
// 12 x 12 product by 6-bit constants), HBM traffic is 8 B in / 0.5 B out per hashed element -- so its roof is VALU issue:
//   S-box x^7 = 4 field products (4 v_mad_u64_u32 + ~14 for the reduction each), 118 S-boxes per permutation;
//   MDS layer with small entries: the sums  sum c_i lo(s_i)  and  sum c_i hi(s_i)  over the 32-bit halves fit 64 bits, so a row is
//   24 v_mad_u64_u32 and ONE reduction (2^32 (h_lo + 2^32 h_hi) = 2^32 h_lo + (2^32 - 1) h_hi) instead of 12 field products.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "goldilocks.hpp"
#include "poseidon.h"

namespace tmx {

struct PosConsts {
  const uint64_t* rc;    // [POS_ROUNDS][12]
  const uint64_t* circ;  // [12]
  const uint64_t* diag;  // [12]
  const uint64_t* x;     // the whole buffer (poseidon.h: POS_X_* tables of the merged partial rounds)
};
__device__ __forceinline__ PosConsts pos_consts(const uint64_t* c) { return PosConsts{c, c + POS_ROUNDS * POS_T, c + POS_ROUNDS * POS_T + POS_T, c}; }

__device__ __forceinline__ uint64_t pos_sbox(uint64_t x) {
  const uint64_t x2 = gl_mul_lazy(x, x), x3 = gl_mul_lazy(x2, x), x4 = gl_mul_lazy(x2, x2);
  return gl_mul_lazy(x3, x4);
}
// lo + 2^32 hi for lo, hi < 2^62: any representative.  2^32 hi = 2^32 h_lo + 2^64 h_hi = 2^32 h_lo + (2^32 - 1) h_hi: the second term
// rides on ONE v_mad_u64_u32 with lo as its addend (< 2^30 2^32 + 2^62: no carry), the first is an add into the high word whose carry
// (2^64 = 2^32 - 1) is folded back once -- after a carry the sum is < 2^63, so it cannot wrap again.
__device__ __forceinline__ uint64_t pos_fold(uint64_t lo, uint64_t hi) {
  const uint32_t h_lo = (uint32_t)hi, h_hi = (uint32_t)(hi >> 32);
  const uint64_t t = (uint64_t)h_hi * 0xffffffffu + lo;
  uint32_t n_hi;
  const bool carry = __builtin_uadd_overflow((uint32_t)(t >> 32), h_lo, &n_hi);
  const uint64_t s = ((uint64_t)n_hi << 32) | (uint32_t)t;
  return s + (carry ? GL_EPS : 0ull);
}
// MDS layer.  SMALL (every entry < 2^16): the diagonal entry joins the circulant's entry 0 on the scalar unit, and the NEXT round's
// constants enter as one more multiply-add per accumulator (rc_lo * inj, rc_hi * inj with inj = 1 in a VGPR; 0 behind the last round) --
// two instructions per element instead of a 64-bit modular add (five).  Bounds: 13 products < 2^17 2^32 and one < 2^32: < 2^53.
template <bool SMALL>
__device__ __forceinline__ void pos_mds(uint64_t (&s)[12], const PosConsts& K, const uint64_t* __restrict__ rc_next, bool more, uint32_t inj) {
  uint64_t o[12];
  if (SMALL) {
    uint32_t lo[12], hi[12];
#pragma unroll
    for (int i = 0; i < 12; i++) { lo[i] = (uint32_t)s[i]; hi[i] = (uint32_t)(s[i] >> 32); }
#pragma unroll
    for (int r = 0; r < 12; r++) {
      const uint64_t rcn = rc_next[r];
      uint64_t al = (uint64_t)(uint32_t)rcn * inj, ah = (uint64_t)(uint32_t)(rcn >> 32) * inj;
#pragma unroll
      for (int i = 0; i < 12; i++) {
        const uint32_t c = (uint32_t)K.circ[i] + (i == 0 ? (uint32_t)K.diag[r] : 0u);
        al += (uint64_t)c * lo[(i + r) % 12];
        ah += (uint64_t)c * hi[(i + r) % 12];
      }
      o[r] = pos_fold(al, ah);
    }
  } else {
#pragma unroll
    for (int r = 0; r < 12; r++) {
      uint64_t acc = gl_mul(s[r], K.diag[r]);
#pragma unroll
      for (int i = 0; i < 12; i++) acc = gl_add(acc, gl_mul(s[(i + r) % 12], K.circ[i]));
      o[r] = more ? gl_add_lazy(acc, rc_next[r]) : acc;  // (uniform)
    }
  }
#pragma unroll
  for (int i = 0; i < 12; i++) s[i] = o[i];
}
// Three partial rounds at once (poseidon.h: POS_MODE_MERGE3).  s holds u = the state with its round constants added; on return the same
// three rounds later.  428 multiply-adds instead of 3 x 324.  The coefficients are scalar loads; the compiler barriers keep them row by row
// (hoisted to the top of the group, the 200 dwords do not fit the scalar registers and were spilled into VGPR lanes: 700 v_writelane /
// v_readlane per group).
__device__ __forceinline__ void pos_group3(uint64_t (&s)[12], const PosConsts& K, uint32_t g, uint32_t vone) {
  const uint64_t* G = K.x + POS_X_GROUPS + g * POS_X_GROUP_WORDS;
  const uint32_t* U = reinterpret_cast<const uint32_t*>(K.x + POS_X_U32);
  s[0] = pos_sbox(s[0]);
  uint32_t lo[12], hi[12];
#pragma unroll
  for (int i = 0; i < 12; i++) { lo[i] = (uint32_t)s[i]; hi[i] = (uint32_t)(s[i] >> 32); }
  uint64_t al, ah;
  // y1 = M[0,:] v + K1
  al = (uint64_t)(uint32_t)G[0] * vone; ah = (uint64_t)(uint32_t)(G[0] >> 32) * vone;
#pragma unroll
  for (int i = 0; i < 12; i++) { const uint32_t c = U[POS_U_R1 + i]; al += (uint64_t)c * lo[i]; ah += (uint64_t)c * hi[i]; }
  const uint64_t y1 = pos_fold(al, ah);
  const uint64_t d1 = gl_sub_lazy(pos_sbox(y1), gl_canon(y1));
  const uint32_t d1lo = (uint32_t)d1, d1hi = (uint32_t)(d1 >> 32);
  asm volatile("" ::: "memory");
  // y2 = M^2[0,:] v + d1 M[0,0] + K2
  al = (uint64_t)(uint32_t)G[1] * vone; ah = (uint64_t)(uint32_t)(G[1] >> 32) * vone;
#pragma unroll
  for (int i = 0; i < 12; i++) { const uint32_t c = U[POS_U_R2 + i]; al += (uint64_t)c * lo[i]; ah += (uint64_t)c * hi[i]; }
  { const uint32_t c = U[POS_U_C1]; al += (uint64_t)c * d1lo; ah += (uint64_t)c * d1hi; }
  const uint64_t y2 = pos_fold(al, ah);
  const uint64_t d2 = gl_sub_lazy(pos_sbox(y2), gl_canon(y2));
  const uint32_t d2lo = (uint32_t)d2, d2hi = (uint32_t)(d2 >> 32);
  // u' = M^3 v + d1 M^2[:,0] + d2 M[:,0] + K3: the coefficients of row r + 1 are requested before the arithmetic of row r and waited for
  // behind it (the empty asm with "s" operands is where the compiler puts the s_waitcnt: scalar loads return out of order, so a wait in
  // front of the arithmetic would wait for the next row's too)
  typedef uint32_t pos_u32x4 __attribute__((ext_vector_type(4)));
  struct Row { pos_u32x4 a, b, c; uint32_t c2, c1; uint64_t k3; };
  auto load_row = [&](int r) {
    Row w;
    const pos_u32x4* m = reinterpret_cast<const pos_u32x4*>(U + POS_U_M3 + 12 * r);
    w.a = m[0]; w.b = m[1]; w.c = m[2]; w.c2 = U[POS_U_C2 + r]; w.c1 = U[POS_U_C1 + r]; w.k3 = G[2 + r];
    return w;
  };
  auto settle = [&](const Row& w) {
    asm volatile("" ::"s"(w.a.x), "s"(w.a.y), "s"(w.a.z), "s"(w.a.w), "s"(w.b.x), "s"(w.b.y), "s"(w.b.z), "s"(w.b.w), "s"(w.c.x), "s"(w.c.y), "s"(w.c.z),
                 "s"(w.c.w), "s"(w.c2), "s"(w.c1), "s"(w.k3)
                 : "memory");
  };
  Row nxt = load_row(0);
  settle(nxt);
#pragma unroll
  for (int r = 0; r < 12; r++) {
    const Row w = nxt;
    if (r + 1 < 12) nxt = load_row(r + 1);
    const uint32_t cf[12] = {w.a.x, w.a.y, w.a.z, w.a.w, w.b.x, w.b.y, w.b.z, w.b.w, w.c.x, w.c.y, w.c.z, w.c.w};
    al = (uint64_t)(uint32_t)w.k3 * vone; ah = (uint64_t)(uint32_t)(w.k3 >> 32) * vone;
#pragma unroll
    for (int i = 0; i < 12; i++) { al += (uint64_t)cf[i] * lo[i]; ah += (uint64_t)cf[i] * hi[i]; }
    al += (uint64_t)w.c2 * d1lo; ah += (uint64_t)w.c2 * d1hi;
    al += (uint64_t)w.c1 * d2lo; ah += (uint64_t)w.c1 * d2hi;
    s[r] = pos_fold(al, ah);
    if (r + 1 < 12) settle(nxt); else asm volatile("" ::: "memory");
  }
}
template <int MODE>
__device__ __forceinline__ void pos_permute(uint64_t (&s)[12], const PosConsts& K) {
  constexpr bool SMALL = MODE != POS_MODE_GENERAL;
  uint32_t vone;
  asm volatile("v_mov_b32 %0, 1" : "=v"(vone));  // (opaque to the optimizer: rc * 1 has to stay a multiply-add)
#pragma unroll
  for (int i = 0; i < 12; i++) s[i] = gl_add_lazy(s[i], K.rc[i]);
#pragma unroll 1
  for (int r = 0; r < (int)POS_ROUNDS; r++) {
    if (MODE == POS_MODE_MERGE3 && r == (int)POS_MERGE_FIRST) {  // rounds 5 .. 25 in seven groups of three
#pragma unroll 1
      for (uint32_t g = 0; g < POS_MERGE_GROUPS; g++) pos_group3(s, K, g, vone);
      r = (int)(POS_RF / 2 + POS_RP) - 1;
      continue;
    }
    if (r < (int)POS_RF / 2 || r >= (int)(POS_RF / 2 + POS_RP)) {  // (uniform: a scalar branch)
#pragma unroll
      for (int i = 0; i < 12; i++) s[i] = pos_sbox(s[i]);
    } else {
      s[0] = pos_sbox(s[0]);
    }
    const bool more = r + 1 < (int)POS_ROUNDS;
    pos_mds<SMALL>(s, K, K.rc + (more ? r + 1 : r) * 12, more, more ? vone : 0u);
  }
}

template <int MODE>
__global__ __launch_bounds__(256) void k_poseidon_permute(const uint64_t* __restrict__ consts, uint32_t n, const uint64_t* __restrict__ in,
                                                          uint64_t* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const PosConsts K = pos_consts(consts);
  uint64_t s[12];
#pragma unroll
  for (int k = 0; k < 12; k++) s[k] = in[(size_t)i * 12 + k];
  pos_permute<MODE>(s, K);
#pragma unroll
  for (int k = 0; k < 12; k++) out[(size_t)i * 12 + k] = gl_canon(s[k]);
}

// leaf digest of row r: the row's n_cols values (column c at cols[(c << log_n) + r]: consecutive threads read consecutive addresses of
// every column), absorbed eight at a time in overwrite mode; rows of at most four values are their own digest (hash_or_noop)
template <int MODE>
__global__ __launch_bounds__(256) void k_poseidon_leaves(const uint64_t* __restrict__ consts, uint32_t log_n, uint32_t n_cols,
                                                         const uint64_t* __restrict__ cols, uint64_t* __restrict__ digests) {
  const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >> log_n) return;
  const PosConsts K = pos_consts(consts);
  uint64_t s[12];
#pragma unroll
  for (int k = 0; k < 12; k++) s[k] = 0;
  if (n_cols <= 4) {
#pragma unroll
    for (uint32_t c = 0; c < 4; c++) s[c] = c < n_cols ? gl_canon(cols[((uint64_t)c << log_n) + r]) : 0;  // (inputs are taken mod p: any u64 < 2 p)
  } else {
    for (uint32_t c0 = 0; c0 < n_cols; c0 += 8) {
#pragma unroll
      for (uint32_t k = 0; k < 8; k++)
        if (c0 + k < n_cols) s[k] = cols[((uint64_t)(c0 + k) << log_n) + r];
      pos_permute<MODE>(s, K);
    }
  }
  typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
  u64x2 a, b;
  a.x = gl_canon(s[0]); a.y = gl_canon(s[1]); b.x = gl_canon(s[2]); b.y = gl_canon(s[3]);
  u64x2* o = reinterpret_cast<u64x2*>(digests + 4 * r);
  o[0] = a; o[1] = b;
}

template <int MODE>
__global__ __launch_bounds__(256) void k_poseidon_level(const uint64_t* __restrict__ consts, uint64_t n_out, const uint64_t* __restrict__ in,
                                                        uint64_t* __restrict__ out) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_out) return;
  const PosConsts K = pos_consts(consts);
  typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
  const u64x2* p = reinterpret_cast<const u64x2*>(in + 8 * i);
  const u64x2 v0 = p[0], v1 = p[1], v2 = p[2], v3 = p[3];
  uint64_t s[12] = {v0.x, v0.y, v1.x, v1.y, v2.x, v2.y, v3.x, v3.y, 0, 0, 0, 0};
  pos_permute<MODE>(s, K);
  u64x2 a, b;
  a.x = gl_canon(s[0]); a.y = gl_canon(s[1]); b.x = gl_canon(s[2]); b.y = gl_canon(s[3]);
  u64x2* o = reinterpret_cast<u64x2*>(out + 4 * i);
  o[0] = a; o[1] = b;
}

static inline hipStream_t S_(void* s) { return reinterpret_cast<hipStream_t>(s); }

int launch_poseidon_permute(const void* d_consts, int mode, uint32_t n, const void* d_in, void* d_out, void* stream) {
  if (n == 0) return 0;
  const dim3 grid((n + 255) / 256);
  if (mode == POS_MODE_MERGE3)
    hipLaunchKernelGGL(k_poseidon_permute<POS_MODE_MERGE3>, grid, dim3(256), 0, S_(stream), reinterpret_cast<const uint64_t*>(d_consts), n,
                       reinterpret_cast<const uint64_t*>(d_in), reinterpret_cast<uint64_t*>(d_out));
  else if (mode == POS_MODE_SMALL)
    hipLaunchKernelGGL(k_poseidon_permute<POS_MODE_SMALL>, grid, dim3(256), 0, S_(stream), reinterpret_cast<const uint64_t*>(d_consts), n,
                       reinterpret_cast<const uint64_t*>(d_in), reinterpret_cast<uint64_t*>(d_out));
  else
    hipLaunchKernelGGL(k_poseidon_permute<POS_MODE_GENERAL>, grid, dim3(256), 0, S_(stream), reinterpret_cast<const uint64_t*>(d_consts), n,
                       reinterpret_cast<const uint64_t*>(d_in), reinterpret_cast<uint64_t*>(d_out));
  return (int)hipGetLastError();
}
int launch_poseidon_leaves(const void* d_consts, int mode, uint32_t log_n, uint32_t n_cols, const void* d_cols, void* d_digests, void* stream) {
  const uint64_t n = 1ull << log_n;
  const dim3 grid((uint32_t)((n + 255) / 256));
  if (mode == POS_MODE_MERGE3)
    hipLaunchKernelGGL(k_poseidon_leaves<POS_MODE_MERGE3>, grid, dim3(256), 0, S_(stream), reinterpret_cast<const uint64_t*>(d_consts), log_n, n_cols,
                       reinterpret_cast<const uint64_t*>(d_cols), reinterpret_cast<uint64_t*>(d_digests));
  else if (mode == POS_MODE_SMALL)
    hipLaunchKernelGGL(k_poseidon_leaves<POS_MODE_SMALL>, grid, dim3(256), 0, S_(stream), reinterpret_cast<const uint64_t*>(d_consts), log_n, n_cols,
                       reinterpret_cast<const uint64_t*>(d_cols), reinterpret_cast<uint64_t*>(d_digests));
  else
    hipLaunchKernelGGL(k_poseidon_leaves<POS_MODE_GENERAL>, grid, dim3(256), 0, S_(stream), reinterpret_cast<const uint64_t*>(d_consts), log_n, n_cols,
                       reinterpret_cast<const uint64_t*>(d_cols), reinterpret_cast<uint64_t*>(d_digests));
  return (int)hipGetLastError();
}
int launch_poseidon_level(const void* d_consts, int mode, uint64_t n_out, const void* d_in, void* d_out, void* stream) {
  if (n_out == 0) return 0;
  const dim3 grid((uint32_t)((n_out + 255) / 256));
  if (mode == POS_MODE_MERGE3)
    hipLaunchKernelGGL(k_poseidon_level<POS_MODE_MERGE3>, grid, dim3(256), 0, S_(stream), reinterpret_cast<const uint64_t*>(d_consts), n_out,
                       reinterpret_cast<const uint64_t*>(d_in), reinterpret_cast<uint64_t*>(d_out));
  else if (mode == POS_MODE_SMALL)
    hipLaunchKernelGGL(k_poseidon_level<POS_MODE_SMALL>, grid, dim3(256), 0, S_(stream), reinterpret_cast<const uint64_t*>(d_consts), n_out,
                       reinterpret_cast<const uint64_t*>(d_in), reinterpret_cast<uint64_t*>(d_out));
  else
    hipLaunchKernelGGL(k_poseidon_level<POS_MODE_GENERAL>, grid, dim3(256), 0, S_(stream), reinterpret_cast<const uint64_t*>(d_consts), n_out,
                       reinterpret_cast<const uint64_t*>(d_in), reinterpret_cast<uint64_t*>(d_out));
  return (int)hipGetLastError();
}

}  // namespace tmx
