// Goldilocks NTT / coset low-degree extension on gfx950 (the step after the witness fill of a plonky2-style prover).
// Definitions (the CPU checker under oracle/ restates them independently; plonky2 itself is not in the reference tree -- parity unpinned):
//   p = 2^64 - 2^32 + 1, omega_N = root^(2^32/N) for the context's 2^32-th root of unity (plonky2's by default, tmx_ntt_set_domain);  X[j] = sum_i x[i] omega_N^(ij), natural order in and out.
//
// One kernel template does every pass: a workgroup loads a tile of T sub-transforms of length L = 2^LOG_L into LDS (T L = 2^12 .. 2^14
// elements, 16 per thread), runs the decimation-in-frequency stages there FOUR AT A TIME as radix-16 steps in registers (every inner
// twiddle of a 16-point transform is a power of two in this field: shifts, not products; 15 general products per 32 butterflies) and
// stores the tile -- so a column of up to 2^11 elements costs one read and one write of HBM, and longer ones (to 2^22) two of each by
// the four-step split N = N1 N2:
//   pass A: N2 strided transforms of length N1 (a tile = T adjacent n2, so that every load is a run of T consecutive elements),
//           multiplied by omega_N^(n2 k1) on the way out, same [k1][n2] layout;
//   pass B: N1 contiguous transforms of length N2, stored transposed (X[k1 + N1 k2]; a tile = T adjacent k1 rows).
// LOG_L and (for the usual tile sizes) T are template parameters: every LDS offset is an immediate, the load / store phases are 16
// unrolled accesses per thread with all loads in flight, and the LDS layout is skewed per sub-transform against bank conflicts.
// Twiddles come from one table of omega_N^i, i < N/2, per transform size (built once per context, omega^(i + N/2) = -omega^i).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "goldilocks.hpp"
#include "ntt.h"

namespace tmx {

// W[i] = omega_N^i, i < N/2 (at least one entry)
__global__ __launch_bounds__(256) void k_ntt_table(uint64_t* __restrict__ W, uint32_t log_n, uint64_t root_2_32) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, half = log_n ? (1ull << (log_n - 1)) : 1ull;
  if (i >= half) return;
  uint64_t w = root_2_32;  // primitive 2^32-th root of unity of the configured domain (api.cpp: plonky2's by default)
  for (uint32_t k = log_n; k < 32; k++) w = gl_mul(w, w);
  W[i] = gl_pow(w, i);
}

// omega_N^idx (inverse: omega_N^-idx), idx < N, from the half table
__device__ __forceinline__ uint64_t twiddle(const uint64_t* __restrict__ W, uint32_t log_n, uint64_t idx, bool inverse) {
  if (idx == 0) return 1;
  const uint64_t n = 1ull << log_n, half = n >> 1;
  if (inverse) idx = n - idx;
  return idx < half ? W[idx] : gl_neg(W[idx - half]);
}

// x 2^S mod p for a compile-time 0 < S < 96: the 160-bit shifted value lo + h0 2^64 + h1 2^96 + h2 2^128 folds with 2^64 = 2^32 - 1,
// 2^96 = -1, 2^128 = -2^32 into  lo + (h0 << 32) - (h0 + h1 + (h2 << 32));  about 16 instructions against 26 for a general product.
// Every root of unity of order <= 64 is a power of two here (2^96 = -1): the inner twiddles of a radix-16 stage group are all of this kind.
template <int S>
__device__ __forceinline__ uint64_t gl_mul_pow2(uint64_t x) {
  static_assert(S > 0 && S < 96, "shift");
  constexpr int q = S / 32, r = S % 32;
  const uint32_t x0 = (uint32_t)x, x1 = (uint32_t)(x >> 32);
  uint32_t w[5] = {0, 0, 0, 0, 0};
  if (r == 0) { w[q] = x0; w[q + 1] = x1; }
  else { w[q] = x0 << r; w[q + 1] = (x1 << r) | (x0 >> (32 - r)); w[q + 2] = x1 >> (32 - r); }
  const uint64_t lo = ((uint64_t)w[1] << 32) | w[0];
  const uint64_t b = (uint64_t)w[2] + w[3] + ((uint64_t)w[4] << 32);  // < 2^63 + 2^33
  unsigned long long t, res;
  const bool borrow = __builtin_usubll_overflow(lo, b, &t);
  t -= borrow ? GL_EPS : 0ull;  // the wrap added 2^64 = p + EPS; t >= 2^64 - b stays non-negative
  const bool carry = __builtin_uaddll_overflow(t, (uint64_t)w[2] << 32, &res);
  res += carry ? GL_EPS : 0ull;  // after a carry res < h0 << 32 <= 2^64 - 2^32: cannot wrap again
  return res;  // (any representative: see the lazy forms above)
}
template <int S>
__device__ __forceinline__ uint64_t gl_mul_pow2_or_id(uint64_t x) {
  if constexpr (S == 0) return x; else return gl_mul_pow2<S>(x);
}

// LDS index with one pad element per 16: the last stage group reads 16 consecutive elements per thread (stride 128 B between threads,
// a 32-way bank conflict without the skew)
__device__ __forceinline__ uint32_t lds_pad(uint32_t i) { return i + (i >> 4); }

// One radix-2^R decimation-in-frequency step (the stages of lengths 2^ll .. 2^(ll-R+1)) on the 2^R elements base + (k << sh) of a thread:
//   y_m[j] = (sum_i x[j + i L/E] omega_E^(i m)) omega_L^(j m),   E = 2^R, L = 2^ll, j < L/E, stored where the radix-2 network leaves it
//   (block bitrev_R(m) of the L/E-element blocks).
// omega_E of the context's domain is (2^(192/E))^k for an odd k; the elements are LOADED in the order kinv i (kinv k = 1 mod 16), after
// which the inner network runs with omega_E = 2^(192/E): every inner twiddle is a compile-time shift, and only the E - 1 outer
// twiddles omega_L^(j m) are general products (15 per 32 butterflies instead of 32).  One LDS round trip and one barrier per group.
template <int R, int ST, int K>
__device__ __forceinline__ void dft_stage(uint64_t (&x)[1 << R]) {
  constexpr int E = 1 << R;
  if constexpr (K < E) {
    constexpr int hb = R - 1 - ST;
    if constexpr ((K & (1 << hb)) == 0) {
      constexpr int k2 = K | (1 << hb), pos = K & ((1 << hb) - 1);
      constexpr int S = (96 * pos) >> hb;  // omega_(2 h)^pos = 2^(96 pos / h), h = 2^hb
      const uint64_t a = x[K], c = gl_canon_rare(x[k2]);
      x[K] = gl_add_lazy(a, c);
      x[k2] = gl_mul_pow2_or_id<S>(gl_sub_lazy(a, c));
    }
    dft_stage<R, ST, K + 1>(x);
  }
}
template <int R, int ST = 0>
__device__ __forceinline__ void dft_pow2(uint64_t (&x)[1 << R]) {
  if constexpr (ST < R) {
    dft_stage<R, ST, 0>(x);
    dft_pow2<R, ST + 1>(x);
  }
}
__device__ __forceinline__ constexpr uint32_t pad_off(uint32_t d) { return d + (d >> 4); }
// Outer-twiddle tables of a tile, one per stage group that has any (every group but the last): group at length 2^LL with radix 2^R holds
// omega_(2^LL)^(j m) at [(m - 1) << (LL - R) | j], j < 2^(LL - R), m = 1 .. 2^R - 1; the groups follow each other (fewer than 2^LOG_L words).
template <int LOG_L>
__device__ __forceinline__ constexpr int group_radix(int ll) { return (ll == LOG_L && LOG_L % 4 != 0) ? LOG_L % 4 : 4; }
template <int LOG_L, int LL>
__device__ __forceinline__ constexpr uint32_t tw_group_offset() {
  uint32_t off = 0;
  for (int ll = LOG_L; ll > LL;) {
    const int r = group_radix<LOG_L>(ll);
    off += ((1u << r) - 1u) << (ll - r);
    ll -= r;
  }
  return off;
}
template <int LOG_L, int LL>
__device__ __forceinline__ void tw_build(uint64_t* __restrict__ tw, const uint64_t* __restrict__ W, uint32_t log_n, bool inv) {
  if constexpr (LL >= 1) {
    constexpr int R = group_radix<LOG_L>(LL), sh = LL - R;
    if constexpr (sh != 0) {
      constexpr uint32_t cnt = ((1u << R) - 1u) << sh;
      uint64_t* tg = tw + tw_group_offset<LOG_L, LL>();
      for (uint32_t i = threadIdx.x; i < cnt; i += blockDim.x) {
        const uint32_t m = (i >> sh) + 1, j = i & ((1u << sh) - 1u);
        tg[i] = twiddle(W, log_n, (uint64_t)(j * m) << (log_n - LL), inv);
      }
      tw_build<LOG_L, LL - R>(tw, W, log_n, inv);
    }
  }
}
template <int R, int LOG_L, int LL, uint32_t SS>
__device__ __forceinline__ void ntt_stage_group(uint64_t* __restrict__ s, const uint64_t* __restrict__ tw, uint32_t tile, uint32_t kinv) {
  constexpr int E = 1 << R, sh = LL - R;
  const uint32_t n_blocks = tile >> R;
  for (uint32_t b = threadIdx.x; b < n_blocks; b += blockDim.x) {
    const uint32_t low = b & ((1u << sh) - 1u), base = ((b >> sh) << LL) | low;
    // base and the element offsets k << sh have no bit in common: lds_pad(base + d) = lds_pad(base) + pad_off(d), so that the
    // stores take compile-time offsets and the (permuted) loads a wave-uniform one
    uint64_t* p = SS ? s + (base >> LOG_L) * SS + lds_pad(base & ((1u << LOG_L) - 1u)) : s + lds_pad(base);
    uint64_t x[E];
#pragma unroll
    for (int k = 0; k < E; k++) x[k] = p[pad_off(((kinv * (uint32_t)k) & (E - 1)) << sh)];
    dft_pow2<R>(x);
    if constexpr (sh != 0) {  // (the last group of a sub-transform has j = 0 only)
      // outer twiddles omega_2^LL^(low m) from this group's table [m - 1][low]: consecutive lanes read consecutive words
      const uint64_t* tg = tw + tw_group_offset<LOG_L, LL>() + low;
#pragma unroll
      for (int k = 1; k < E; k++) {
        const uint32_t m = __builtin_bitreverse32((uint32_t)k) >> (32 - R);  // register k holds y_bitrev(k)
        x[k] = gl_mul_lazy(x[k], tg[(m - 1) << sh]);
      }
    }
#pragma unroll
    for (int k = 0; k < E; k++) p[pad_off((uint32_t)k << sh)] = x[k];
  }
}
// the radix-16 groups below the first one
template <int LOG_L, int LL, uint32_t SS>
__device__ __forceinline__ void ntt_groups16(uint64_t* __restrict__ s, const uint64_t* __restrict__ tw, uint32_t tile, uint32_t kinv) {
  if constexpr (LL >= 4) {
    ntt_stage_group<4, LOG_L, LL, SS>(s, tw, tile, kinv);
    __syncthreads();
    ntt_groups16<LOG_L, LL - 4, SS>(s, tw, tile, kinv);
  }
}

// Which (t, j) of the tile a thread moves in the load / store phases when the tile shape is a compile-time constant: element i of
// thread tid is (t0(tid) + dt(i), j0(tid) + dj(i)) with compile-time dt / dj whose bits never meet those of the thread's part, so that
// LDS addresses are one runtime base + immediates and global addresses one base + wave-uniform steps.  CONTIG: j is the unit-stride
// dimension of global memory (lanes follow j), otherwise t is (the strided passes of the four-step split).
template <int LOG_L, int LOG_T, bool CONTIG>
struct TileMap {
  static constexpr uint32_t L = 1u << LOG_L, T = 1u << LOG_T, TH = 1u << (LOG_L + LOG_T - 4);  // TH threads, 16 elements each
  static constexpr uint32_t Q = (CONTIG && TH < L) ? L / TH : 1u;
  static __device__ __forceinline__ uint32_t t_of(uint32_t tid) { return CONTIG ? (TH >= L ? tid >> LOG_L : 0u) : (tid & (T - 1)); }
  static __device__ __forceinline__ uint32_t j_of(uint32_t tid) { return CONTIG ? (TH >= L ? (tid & (L - 1)) : tid) : (tid >> LOG_T); }
  static constexpr uint32_t dt(uint32_t i) { return CONTIG ? (TH >= L ? i * (TH >> LOG_L) : i / Q) : 0u; }
  static constexpr uint32_t dj(uint32_t i) { return CONTIG ? (TH >= L ? 0u : (i % Q) * TH) : i * (L >> 4); }
};
// LDS layout of a tile with a compile-time shape: sub-transform t at t * sub_stride, element j of it at lds_pad(j).  The skew makes the T
// lanes that differ only in t (the unit-stride dimension of the strided passes) fall into different banks: with T = 8 and 8 consecutive
// j per wave, 8 t + j covers every 8-byte bank pair exactly twice.
template <int LOG_L, int LOG_T>
__device__ __forceinline__ constexpr uint32_t sub_stride() {
  return (1u << LOG_L) + ((1u << LOG_L) >> 4) + (LOG_T >= 1 && LOG_T <= 6 ? (64u >> LOG_T) : 0u);
}
constexpr uint32_t brev_bits(uint32_t v, int bits) {
  uint32_t r = 0;
  for (int b = 0; b < bits; b++) r |= ((v >> b) & 1u) << (bits - 1 - b);
  return r;
}

template <int LOG_L, int LOG_T, bool CONTIG, bool FULL>
__device__ __forceinline__ void tile_load(uint64_t* __restrict__ s, const uint64_t* __restrict__ src, uint32_t t0, uint32_t n_sub, uint32_t ts,
                                          uint32_t js, uint32_t j_nz) {
  using M = TileMap<LOG_L, LOG_T, CONTIG>;
  constexpr uint32_t SS = sub_stride<LOG_L, LOG_T>();
  const uint32_t tl = M::t_of(threadIdx.x), j0 = M::j_of(threadIdx.x);
  const uint64_t* p = src + ((uint64_t)(t0 + tl) * ts + (uint64_t)j0 * js);
  uint64_t v[16];
#pragma unroll
  for (int i = 0; i < 16; i++) {
    const uint64_t step = (uint64_t)M::dt(i) * ts + (uint64_t)M::dj(i) * js;  // wave-uniform
    v[i] = ((FULL || t0 + tl + M::dt(i) < n_sub) && j0 + M::dj(i) < j_nz) ? p[step] : 0;  // (j_nz = 2^32 - 1 when every input is read)
  }
  uint64_t* q = s + tl * SS + lds_pad(j0);
#pragma unroll
  for (int i = 0; i < 16; i++) q[M::dt(i) * SS + pad_off(M::dj(i))] = v[i];
}
// Output k of a sub-transform sits at bitrev(k).  Strided passes (t is the unit-stride dimension, so any k may go to any thread): a thread
// takes the 16 CONSECUTIVE words 16 r .. 16 r + 15 of its sub-transform, i.e. the outputs k = bitrev(16 r + i) = (bitrev4(i) << (LOG_L - 4)) |
// bitrev(r) -- lanes with consecutive k would read words L/2, L/4, ... apart, all in a few banks.  Contiguous passes: k follows the lanes.
template <int LOG_L, int LOG_T, bool CONTIG, bool FULL>
__device__ __forceinline__ void tile_store(const uint64_t* __restrict__ s, uint64_t* __restrict__ dst, const uint64_t* __restrict__ TM,
                                           const NttPass& P, uint32_t t0, uint32_t n_sub, uint32_t ts, uint32_t js) {
  using M = TileMap<LOG_L, LOG_T, CONTIG>;
  constexpr uint32_t SS = sub_stride<LOG_L, LOG_T>();
  const uint32_t tl = M::t_of(threadIdx.x), r0 = M::j_of(threadIdx.x);
  const bool post = P.post_twiddle != 0, scaled = P.scale != 1;
  constexpr int HB = LOG_L > 4 ? LOG_L - 4 : 1;  // bits of r0 in the strided mapping
  const uint32_t k0 = CONTIG ? r0 : (LOG_L > 4 ? (__brev(r0) >> (32 - HB)) : 0u);
  const uint32_t kr0 = __brev(r0) >> (32 - LOG_L);  // (contiguous mapping) bitrev(k0 + dj) = bitrev(k0) + bitrev(dj): no common bits
  const uint64_t* q = s + tl * SS + (CONTIG ? lds_pad(kr0) : 17u * r0);
  // the four-step twiddles omega_N^(n2 k1) come from a matrix in the layout of this pass's output column: same offsets, same coalescing
  const uint64_t in_col = (uint64_t)(t0 + tl) * ts + (uint64_t)k0 * js;
  const uint64_t tts = P.tm_t_stride;  // (the four-step matrix: = ts, same offsets as the output; a vector indexed by k alone: 0)
  const uint64_t* tm = TM + ((uint64_t)(t0 + tl) * tts + (uint64_t)k0 * js);
  uint64_t v[16], w[16];
  uint32_t dk[16];
#pragma unroll
  for (int i = 0; i < 16; i++) {
    dk[i] = CONTIG ? M::dj(i) : (brev_bits(i, 4) << (LOG_L - 4));
    const uint64_t step = (uint64_t)M::dt(i) * ts + (uint64_t)dk[i] * js;
    v[i] = q[CONTIG ? M::dt(i) * SS + pad_off(brev_bits(M::dj(i), LOG_L)) : (uint32_t)i];
    w[i] = (post && (FULL || t0 + tl + M::dt(i) < n_sub)) ? tm[(uint64_t)M::dt(i) * tts + (uint64_t)dk[i] * js] : 1;
  }
  uint64_t* o = dst + in_col;
#pragma unroll
  for (int i = 0; i < 16; i++) {
    uint64_t r = v[i];
    if (post) r = gl_mul_lazy(r, w[i]);
    if (scaled) r = gl_mul_lazy(r, P.scale);
    r = gl_canon_rare(r);
    const uint64_t step = (uint64_t)M::dt(i) * ts + (uint64_t)dk[i] * js;
    if (FULL || t0 + tl + M::dt(i) < n_sub) o[step] = r;
  }
}

// LOG_T >= 0: the tile holds 2^LOG_T sub-transforms (launch with 2^(LOG_L + LOG_T - 4) threads); LOG_T < 0: P.log_t of them (any tile
// of up to 2^14 elements, launch with max(64, tile / 16) threads)
template <int LOG_L, int LOG_T>
__global__ __launch_bounds__(1024) void k_ntt_tile(NttPass P, const uint64_t* __restrict__ in, uint64_t* __restrict__ out,
                                                  const uint64_t* __restrict__ W, const uint64_t* __restrict__ M) {
  extern __shared__ uint64_t lds[];
  constexpr uint32_t L = 1u << LOG_L;
  constexpr bool FIXED = LOG_T >= 0;
  constexpr uint32_t SS = FIXED ? sub_stride<LOG_L, (FIXED ? LOG_T : 0)>() : 0u;  // 0: flat layout lds_pad(t L + j)
  uint64_t* tw = lds;     // the outer-twiddle tables of the stage groups (tw_build), fewer than L words
  uint64_t* s = lds + L;  // the padded tile
  const uint32_t log_t = FIXED ? (uint32_t)LOG_T : P.log_t;
  const uint32_t T = 1u << log_t, tile = L << log_t;
  const uint32_t col = blockIdx.x / P.tiles_per_col, tix = blockIdx.x % P.tiles_per_col;
  const uint32_t t0 = tix * T;  // first sub-transform of this tile within the column
  const uint64_t* src = in + (size_t)col * P.col_stride_in;
  uint64_t* dst = out + (size_t)col * P.col_stride_out;
  const bool inv = P.inverse != 0;
  const uint32_t n_sub = (uint32_t)P.n_sub;
  const bool full = t0 + T <= n_sub;
  tw_build<LOG_L, LOG_L>(tw, W, P.log_n, inv);
  // load: consecutive threads follow the unit-stride dimension (every stride of a pass fits 32 bits: columns have at most 2^22 elements).
  // A thread moves 16 elements; the unrolled loops keep its 16 loads in flight together.
  {
    const uint32_t ts = (uint32_t)P.t_stride_in, js = (uint32_t)P.j_stride_in;
    const bool contiguous = js == 1;
    const uint32_t j_nz = P.j_nonzero ? P.j_nonzero : 0xffffffffu;
    if constexpr (FIXED) {
      constexpr int LT = FIXED ? LOG_T : 0;
      if (contiguous) { if (full) tile_load<LOG_L, LT, true, true>(s, src, t0, n_sub, ts, js, j_nz); else tile_load<LOG_L, LT, true, false>(s, src, t0, n_sub, ts, js, j_nz); }
      else { if (full) tile_load<LOG_L, LT, false, true>(s, src, t0, n_sub, ts, js, j_nz); else tile_load<LOG_L, LT, false, false>(s, src, t0, n_sub, ts, js, j_nz); }
    } else {
      uint64_t v[16];
#pragma unroll
      for (int i = 0; i < 16; i++) {
        const uint32_t e = threadIdx.x + i * blockDim.x;
        uint32_t t, j;
        if (contiguous) { j = e & (L - 1); t = e >> LOG_L; } else { t = e & (T - 1); j = e >> log_t; }
        const bool live = e < tile && t0 + t < n_sub && j < j_nz;
        const uint64_t off = (uint64_t)(t0 + t) * ts + (uint64_t)j * js;
        v[i] = live ? src[off] : 0;
      }
#pragma unroll
      for (int i = 0; i < 16; i++) {
        const uint32_t e = threadIdx.x + i * blockDim.x;
        uint32_t t, j;
        if (contiguous) { j = e & (L - 1); t = e >> LOG_L; } else { t = e & (T - 1); j = e >> log_t; }
        if (e < tile) s[lds_pad(t * L + j)] = v[i];
      }
    }
  }
  __syncthreads();
  // decimation in frequency, natural order in, bit-reversed order out: the LOG_L mod 4 leading stages as one group, then radix-16 groups
  constexpr int R0 = LOG_L % 4;
  if constexpr (R0 != 0) {
    ntt_stage_group<R0, LOG_L, LOG_L, SS>(s, tw, tile, P.kinv);
    __syncthreads();
  }
  ntt_groups16<LOG_L, LOG_L - R0, SS>(s, tw, tile, P.kinv);
  // store, with the four-step twiddle / the inverse scale
  {
    const uint32_t ts = (uint32_t)P.t_stride_out, js = (uint32_t)P.j_stride_out;
    const bool contiguous = js == 1;
    if constexpr (FIXED) {
      constexpr int LT = FIXED ? LOG_T : 0;
      if (contiguous) { if (full) tile_store<LOG_L, LT, true, true>(s, dst, M, P, t0, n_sub, ts, js); else tile_store<LOG_L, LT, true, false>(s, dst, M, P, t0, n_sub, ts, js); }
      else { if (full) tile_store<LOG_L, LT, false, true>(s, dst, M, P, t0, n_sub, ts, js); else tile_store<LOG_L, LT, false, false>(s, dst, M, P, t0, n_sub, ts, js); }
    } else {
      const bool post = P.post_twiddle != 0, scaled = P.scale != 1;
      uint64_t v[16], w[16];
#pragma unroll
      for (int i = 0; i < 16; i++) {
        const uint32_t e = threadIdx.x + i * blockDim.x;
        uint32_t t, k;
        if (contiguous) { k = e & (L - 1); t = e >> LOG_L; } else { t = e & (T - 1); k = e >> log_t; }
        const bool live = e < tile && t0 + t < n_sub;
        const uint32_t kr = LOG_L ? (__brev(k) >> (32 - (LOG_L ? LOG_L : 1))) : 0u;
        v[i] = live ? s[lds_pad(t * L + kr)] : 0;
        w[i] = (post && live) ? M[(uint64_t)(t0 + t) * P.tm_t_stride + (uint64_t)k * js] : 1;  // omega_N^(n2 k1), in the layout of the output column
      }
#pragma unroll
      for (int i = 0; i < 16; i++) {
        const uint32_t e = threadIdx.x + i * blockDim.x;
        uint32_t t, k;
        if (contiguous) { k = e & (L - 1); t = e >> LOG_L; } else { t = e & (T - 1); k = e >> log_t; }
        const bool live = e < tile && t0 + t < n_sub;
        uint64_t r = v[i];
        if (post) r = gl_mul_lazy(r, w[i]);
        if (scaled) r = gl_mul_lazy(r, P.scale);
        r = gl_canon_rare(r);
        if (live) dst[(uint64_t)(t0 + t) * ts + (uint64_t)k * js] = r;
      }
    }
  }
}

// coefficient scaling of the coset LDE: c_i <- c_i g^i for i < n, zero for n <= i < m (per column of stride m)
__global__ __launch_bounds__(256) void k_lde_expand(uint64_t* __restrict__ buf, uint32_t log_n, uint32_t log_m, uint64_t total, uint64_t shift) {
  const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const uint64_t i = e & ((1ull << log_m) - 1);
  if (i >> log_n) { buf[e] = 0; return; }
  buf[e] = gl_mul(buf[e], gl_pow(shift, i));
}

// S[i] = scale * shift^i: what the last pass of the inverse transform of a coset LDE multiplies coefficient i by
__global__ __launch_bounds__(256) void k_lde_scale_table(uint64_t* __restrict__ S, uint32_t log_n, uint64_t shift, uint64_t scale) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >> log_n) return;
  S[i] = gl_mul(scale, gl_pow(shift, i));
}
int launch_lde_scale_table(void* d_s, uint32_t log_n, uint64_t shift, uint64_t scale, void* stream) {
  const uint64_t n = 1ull << log_n;
  hipLaunchKernelGGL(k_lde_scale_table, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     reinterpret_cast<uint64_t*>(d_s), log_n, shift, scale);
  return (int)hipGetLastError();
}
int launch_ntt_table(void* d_w, uint32_t log_n, uint64_t root_2_32, void* stream) {
  const uint64_t half = log_n ? (1ull << (log_n - 1)) : 1ull;
  hipLaunchKernelGGL(k_ntt_table, dim3((unsigned)((half + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     reinterpret_cast<uint64_t*>(d_w), log_n, root_2_32);
  return (int)hipGetLastError();
}
// M[k1 N2 + n2] = omega_N^(+- n2 k1): the twiddles between the two passes of the four-step split, in the layout pass A writes
__global__ __launch_bounds__(256) void k_ntt_matrix(uint64_t* __restrict__ M, const uint64_t* __restrict__ W, uint32_t log_n, uint32_t log_n2,
                                                    uint32_t inverse) {
  const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >> log_n) return;
  const uint64_t k1 = e >> log_n2, n2 = e & ((1ull << log_n2) - 1);
  M[e] = twiddle(W, log_n, n2 * k1, inverse != 0);
}
int launch_ntt_matrix(void* d_m, const void* d_w, uint32_t log_n, uint32_t log_n2, bool inverse, void* stream) {
  const uint64_t n = 1ull << log_n;
  hipLaunchKernelGGL(k_ntt_matrix, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     reinterpret_cast<uint64_t*>(d_m), reinterpret_cast<const uint64_t*>(d_w), log_n, log_n2, inverse ? 1u : 0u);
  return (int)hipGetLastError();
}
int launch_ntt_pass(const NttPass& P, uint32_t n_cols, const void* d_in, void* d_out, const void* d_w, const void* d_m, void* stream) {
  const size_t tile = ((size_t)1 << P.log_l) << P.log_t;
  // the outer twiddles (fewer than L words) + the padded tile (+ 64 words for the per-sub-transform skew of the fixed-shape kernels)
  const size_t lds = 8 * (tile + (tile >> 4) + 1 + 64) + ((size_t)8 << P.log_l);
  const unsigned threads = tile >= 1024 ? (unsigned)(tile >> 4) : 64u;  // 16 elements per thread (tiles of up to 2^14 elements)
#define TMX_NTT_LAUNCH(LL, LT)                                                                                                        \
  {                                                                                                                                  \
    static size_t lds_set = 0;                                                                                                       \
    if (lds > 65536 && lds > lds_set) {                                                                                              \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ntt_tile<LL, LT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
      if (e != hipSuccess) return (int)e;                                                                                            \
      lds_set = lds;                                                                                                                 \
    }                                                                                                                                \
    hipLaunchKernelGGL((k_ntt_tile<LL, LT>), dim3(n_cols * P.tiles_per_col), dim3(threads), lds, reinterpret_cast<hipStream_t>(stream), P, \
                       reinterpret_cast<const uint64_t*>(d_in), reinterpret_cast<uint64_t*>(d_out), reinterpret_cast<const uint64_t*>(d_w), \
                       reinterpret_cast<const uint64_t*>(d_m));                                                                        \
  }
  // tiles of 2^12 .. 2^14 elements take the kernels with a compile-time shape, anything else (few columns) the generic one
#define TMX_NTT_CASE(LL)                                                                                                              \
  case LL:                                                                                                                           \
    if (P.log_l + P.log_t == 12) TMX_NTT_LAUNCH(LL, 12 - LL)                                                                         \
    else if (P.log_l + P.log_t == 13) TMX_NTT_LAUNCH(LL, 13 - LL)                                                                    \
    else if (P.log_l + P.log_t == 14) TMX_NTT_LAUNCH(LL, 14 - LL)                                                                    \
    else TMX_NTT_LAUNCH(LL, -1)                                                                                                      \
    break;
#define TMX_NTT_SMALL(LL) case LL: TMX_NTT_LAUNCH(LL, -1) break;  // (sub-transforms shorter than a thread's 16 elements)
  switch (P.log_l) {
    TMX_NTT_SMALL(0) TMX_NTT_SMALL(1) TMX_NTT_SMALL(2) TMX_NTT_SMALL(3) TMX_NTT_CASE(4) TMX_NTT_CASE(5) TMX_NTT_CASE(6)
    TMX_NTT_CASE(7) TMX_NTT_CASE(8) TMX_NTT_CASE(9) TMX_NTT_CASE(10) TMX_NTT_CASE(11)
    default: return (int)hipErrorInvalidValue;
  }
#undef TMX_NTT_SMALL
#undef TMX_NTT_LAUNCH
#undef TMX_NTT_CASE
  return (int)hipGetLastError();
}
int launch_lde_expand(void* d_buf, uint32_t log_n, uint32_t log_m, uint32_t n_cols, uint64_t shift, void* stream) {
  const uint64_t total = (uint64_t)n_cols << log_m;
  hipLaunchKernelGGL(k_lde_expand, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     reinterpret_cast<uint64_t*>(d_buf), log_n, log_m, total, shift);
  return (int)hipGetLastError();
}

}  // namespace tmx
