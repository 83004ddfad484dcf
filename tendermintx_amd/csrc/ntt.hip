// Goldilocks NTT / coset low-degree extension on gfx950 (the step after the witness fill of a plonky2-style prover).
// Definitions (the CPU checker under oracle/ restates them independently; plonky2 itself is not in the reference tree -- parity unpinned):
//   p = 2^64 - 2^32 + 1, omega_N = root^(2^32/N) for the context's 2^32-th root of unity (plonky2's by default, tmx_ntt_set_domain);  X[j] = sum_i x[i] omega_N^(ij), natural order in and out.
//
// One kernel does every pass: a workgroup loads a tile of T sub-transforms of length L = 2^log_l into LDS (T L <= 4096 elements =
// 32 KB), runs the log_l radix-2 decimation-in-frequency stages there and stores the tile -- so a column of up to 2^11 elements
// costs one read and one write of HBM, and longer ones (to 2^22) two of each by the four-step split N = N1 N2:
//   pass A: N2 strided transforms of length N1 (a tile = T adjacent n2, so that every load is a run of T consecutive elements),
//           multiplied by omega_N^(n2 k1) on the way out, same [k1][n2] layout;
//   pass B: N1 contiguous transforms of length N2, stored transposed (X[k1 + N1 k2]; a tile = T adjacent k1 rows).
// Twiddles come from one table of omega_N^i, i < N/2, per transform size (built once per context, omega^(i + N/2) = -omega^i).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "ntt.h"

namespace tmx {

constexpr uint64_t GL_P = 0xffffffff00000001ull, GL_EPS = 0xffffffffull;

__device__ __forceinline__ uint64_t gl_canon(uint64_t x) { return x >= GL_P ? x - GL_P : x; }
__device__ __forceinline__ uint64_t gl_add(uint64_t a, uint64_t b) {  // a, b < p
  unsigned long long s;
  const bool carry = __builtin_uaddll_overflow(a, b, &s);
  s += carry ? GL_EPS : 0ull;  // 2^64 = 2^32 - 1 (mod p); cannot wrap again and stays below p
  return gl_canon(s);
}
__device__ __forceinline__ uint64_t gl_sub(uint64_t a, uint64_t b) {
  unsigned long long d;
  const bool borrow = __builtin_usubll_overflow(a, b, &d);
  d += borrow ? GL_P : 0ull;
  return d;
}
__device__ __forceinline__ uint64_t gl_neg(uint64_t a) { return a ? GL_P - a : 0; }
__device__ __forceinline__ uint64_t gl_mul(uint64_t a, uint64_t b) {
  // 128-bit product from four 32 x 32 -> 64 multiply-adds (v_mad_u64_u32), no addend can overflow:
  //   p00 = a0 b0;  p01 = a0 b1 + hi(p00);  p10 = a1 b0 + lo(p01);  p11 = a1 b1 + hi(p01) + hi(p10)
  const uint32_t a0 = (uint32_t)a, a1 = (uint32_t)(a >> 32), b0 = (uint32_t)b, b1 = (uint32_t)(b >> 32);
  const uint64_t p00 = (uint64_t)a0 * b0;
  const uint64_t p01 = (uint64_t)a0 * b1 + (p00 >> 32);
  const uint64_t p10 = (uint64_t)a1 * b0 + (uint32_t)p01;
  const uint64_t hi = (uint64_t)a1 * b1 + ((p01 >> 32) + (p10 >> 32));
  const uint64_t lo = (p10 << 32) | (uint32_t)p00;
  const uint64_t hi_hi = hi >> 32, hi_lo = hi & GL_EPS;  // x = lo + 2^64 hi_lo + 2^96 hi_hi = lo + (2^32 - 1) hi_lo - hi_hi
  unsigned long long t0, r;
  const bool borrow = __builtin_usubll_overflow(lo, hi_hi, &t0);
  t0 -= borrow ? GL_EPS : 0ull;  // the wrap added 2^64 = p + EPS
  const uint64_t t1 = (hi_lo << 32) - hi_lo;
  const bool carry = __builtin_uaddll_overflow(t0, t1, &r);
  r += carry ? GL_EPS : 0ull;
  return gl_canon(r);
}
__device__ __forceinline__ uint64_t gl_pow(uint64_t b, uint64_t e) {
  uint64_t r = 1;
  while (e) {
    if (e & 1) r = gl_mul(r, b);
    b = gl_mul(b, b);
    e >>= 1;
  }
  return r;
}

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

// LDS index with one pad element per 16: the last stage group reads 16 consecutive elements per thread (stride 128 B between threads,
// a 32-way bank conflict without the skew)
__device__ __forceinline__ uint32_t lds_pad(uint32_t i) { return i + (i >> 4); }

// R consecutive radix-2 DIF stages (lengths 2^ll .. 2^(ll-R+1)) on 2^R elements per thread held in registers: one LDS round trip and
// one barrier per group instead of per stage
template <int R>
__device__ __forceinline__ void ntt_stage_group(uint64_t* __restrict__ s, const uint64_t* __restrict__ tw, uint32_t log_l, uint32_t ll,
                                                uint32_t tile) {
  constexpr int E = 1 << R;
  const uint32_t sh = ll - R, n_blocks = tile >> R;
  for (uint32_t b = threadIdx.x; b < n_blocks; b += blockDim.x) {
    const uint32_t low = b & ((1u << sh) - 1u), base = ((b >> sh) << ll) | low;  // the thread's elements: base + (k << sh)
    uint64_t x[E];
#pragma unroll
    for (int k = 0; k < E; k++) x[k] = s[lds_pad(base + ((uint32_t)k << sh))];
#pragma unroll
    for (int st = 0; st < R; st++) {
      const int hb = R - 1 - st;            // the bit of k that this stage pairs
      const uint32_t wsh = log_l - (ll - st);  // twiddle index = (index mod half) << wsh
#pragma unroll
      for (int k = 0; k < E; k++) {
        if (k & (1 << hb)) continue;
        const int k2 = k | (1 << hb);
        const uint32_t pos = low + ((uint32_t)(k & ((1 << hb) - 1)) << sh);
        const uint64_t w = tw[pos << wsh];
        const uint64_t a = x[k], c = x[k2];
        x[k] = gl_add(a, c);
        x[k2] = gl_mul(gl_sub(a, c), w);
      }
    }
#pragma unroll
    for (int k = 0; k < E; k++) s[lds_pad(base + ((uint32_t)k << sh))] = x[k];
  }
}

__global__ __launch_bounds__(256) void k_ntt_tile(NttPass P, const uint64_t* __restrict__ in, uint64_t* __restrict__ out,
                                                  const uint64_t* __restrict__ W) {
  extern __shared__ uint64_t s[];
  const uint32_t L = 1u << P.log_l, T = 1u << P.log_t, tile = L * T;
  uint64_t* tw = s + lds_pad(tile) + 1;  // (behind the padded tile) omega_L^j, j < L/2: every stage twiddle of the tile (omega_len^pos = omega_L^(pos L / len))
  const uint32_t col = blockIdx.x / P.tiles_per_col, tix = blockIdx.x % P.tiles_per_col;
  const uint64_t t0 = (uint64_t)tix * T;  // first sub-transform of this tile within the column
  const uint64_t* src = in + (size_t)col * P.col_stride_in;
  uint64_t* dst = out + (size_t)col * P.col_stride_out;
  const bool inv = P.inverse != 0;
  for (uint32_t j = threadIdx.x; j < (L >> 1); j += blockDim.x) tw[j] = twiddle(W, P.log_n, (uint64_t)j << (P.log_n - P.log_l), inv);
  // load: consecutive threads follow the unit-stride dimension
  for (uint32_t e = threadIdx.x; e < tile; e += blockDim.x) {
    uint32_t t, j;
    if (P.j_stride_in == 1) { j = e & (L - 1); t = e >> P.log_l; } else { t = e & (T - 1); j = e >> P.log_t; }
    const bool live = t0 + t < P.n_sub;
    s[lds_pad(t * L + j)] = live ? gl_canon(src[(t0 + t) * P.t_stride_in + (uint64_t)j * P.j_stride_in]) : 0;
  }
  __syncthreads();
  // decimation in frequency: natural order in, bit-reversed order out; four stages per LDS round trip, the remainder in one group
  for (uint32_t ll = P.log_l; ll >= 1;) {
    const uint32_t r = ll >= 4 ? 4u : ll;
    if (r == 4) ntt_stage_group<4>(s, tw, P.log_l, ll, tile);
    else if (r == 3) ntt_stage_group<3>(s, tw, P.log_l, ll, tile);
    else if (r == 2) ntt_stage_group<2>(s, tw, P.log_l, ll, tile);
    else ntt_stage_group<1>(s, tw, P.log_l, ll, tile);
    ll -= r;
    __syncthreads();
  }
  // store (output index k of a sub-transform sits at bitrev(k)), with the four-step twiddle / the inverse scale
  for (uint32_t e = threadIdx.x; e < tile; e += blockDim.x) {
    uint32_t t, k;
    if (P.j_stride_out == 1) { k = e & (L - 1); t = e >> P.log_l; } else { t = e & (T - 1); k = e >> P.log_t; }
    if (t0 + t >= P.n_sub) continue;
    const uint32_t kr = P.log_l ? (__brev(k) >> (32 - P.log_l)) : 0u;
    uint64_t v = s[lds_pad(t * L + kr)];
    if (P.post_twiddle) v = gl_mul(v, twiddle(W, P.log_n, (t0 + t) * (uint64_t)k, inv));  // omega_N^(n2 k1)
    if (P.scale != 1) v = gl_mul(v, P.scale);
    dst[(t0 + t) * P.t_stride_out + (uint64_t)k * P.j_stride_out] = v;
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

int launch_ntt_table(void* d_w, uint32_t log_n, uint64_t root_2_32, void* stream) {
  const uint64_t half = log_n ? (1ull << (log_n - 1)) : 1ull;
  hipLaunchKernelGGL(k_ntt_table, dim3((unsigned)((half + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     reinterpret_cast<uint64_t*>(d_w), log_n, root_2_32);
  return (int)hipGetLastError();
}
int launch_ntt_pass(const NttPass& P, uint32_t n_cols, const void* d_in, void* d_out, const void* d_w, void* stream) {
  const size_t tile = ((size_t)1 << P.log_l) << P.log_t;
  const size_t lds = 8 * (tile + (tile >> 4) + 1) + ((size_t)8 << P.log_l) / 2;  // padded tile (lds_pad) + the stage twiddles
  hipLaunchKernelGGL(k_ntt_tile, dim3(n_cols * P.tiles_per_col), dim3(256), lds, reinterpret_cast<hipStream_t>(stream), P,
                     reinterpret_cast<const uint64_t*>(d_in), reinterpret_cast<uint64_t*>(d_out), reinterpret_cast<const uint64_t*>(d_w));
  return (int)hipGetLastError();
}
int launch_lde_expand(void* d_buf, uint32_t log_n, uint32_t log_m, uint32_t n_cols, uint64_t shift, void* stream) {
  const uint64_t total = (uint64_t)n_cols << log_m;
  hipLaunchKernelGGL(k_lde_expand, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     reinterpret_cast<uint64_t*>(d_buf), log_n, log_m, total, shift);
  return (int)hipGetLastError();
}

}  // namespace tmx
