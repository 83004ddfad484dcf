// Scalars modulo the group order l = 2^252 + 27742317777372353535851937790883648493, one per lane.
// h = SHA-512(R || A || M) mod l for the EdDSA equation evaluated at reference
// circuits/builder/verify.rs:248-259 (plonky2x, un-vendored); RFC 8032 §5.1.7.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tmx {

// l as eight little-endian 32-bit words
__device__ __constant__ const uint32_t K_L[8] = {0x5cf5d3edu, 0x5812631au, 0xa2f79cd6u, 0x14def9deu, 0u, 0u, 0u, 0x10000000u};

// conditional subtraction r -= l if r >= l (branch-free)
__device__ __forceinline__ void sc_sub_l_if_ge(uint32_t r[8]) {
  uint32_t d[8];
  uint32_t borrow = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    uint64_t t = (uint64_t)r[i] - K_L[i] - borrow;
    d[i] = (uint32_t)t;
    borrow = (uint32_t)(t >> 63);
  }
#pragma unroll
  for (int i = 0; i < 8; i++) r[i] = borrow ? r[i] : d[i];
}

__device__ __forceinline__ bool sc_is_canonical(const uint32_t s[8]) {
  uint32_t borrow = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    uint64_t t = (uint64_t)s[i] - K_L[i] - borrow;
    borrow = (uint32_t)(t >> 63);
  }
  return borrow != 0;  // s < l
}

// x (nw words, value < 2^(32 nw)) mod l by bitwise shift-and-subtract, MSB first.
template <int NW>
__device__ __forceinline__ void sc_reduce_bits(const uint32_t* x, uint32_t out[8]) {
  uint32_t a[8];
#pragma unroll
  for (int i = 0; i < 8; i++) a[i] = 0;
  for (int wi = NW - 1; wi >= 0; wi--) {
    uint32_t word = x[wi];
    for (int b = 31; b >= 0; b--) {
      // a = 2a + bit   (a < l < 2^253, so 2a+1 < 2^254: no overflow)
#pragma unroll
      for (int i = 7; i > 0; i--) a[i] = (a[i] << 1) | (a[i - 1] >> 31);
      a[0] = (a[0] << 1) | ((word >> b) & 1);
      sc_sub_l_if_ge(a);
    }
  }
#pragma unroll
  for (int i = 0; i < 8; i++) out[i] = a[i];
}

// out = x mod l for a 512-bit x (16 LE words): radix-2^21 folding.  l = 2^252 + c and 252 = 12 * 21, so limb i >= 12 folds
// into limbs i-12 .. i-7 with the signed radix-2^21 digits of -c (derived in DESIGN.md §3; c = l - 2^252):
//   2^252 = -c = 666643 + 470296*2^21 + 654183*2^42 - 997805*2^63 + 136657*2^84 - 683901*2^105  (mod l)
// ~72 multiply-adds + carries instead of 512 shift-subtract rounds; every intermediate stays below 2^56 (checked exhaustively on
// the magnitudes by the model in tests/test_oracle_kat.py::test_sc_fold_model).
__device__ __forceinline__ void sc_reduce512(const uint32_t x[16], uint32_t out[8]) {
  int64_t s[24];
#pragma unroll
  for (int i = 0; i < 24; i++) {
    const int bit = 21 * i, w = bit >> 5, sh = bit & 31;
    uint64_t two = (uint64_t)x[w] | ((w + 1 < 16) ? ((uint64_t)x[w + 1] << 32) : 0ull);
    uint64_t v = two >> sh;
    s[i] = (int64_t)(i < 23 ? (v & 0x1fffffull) : v);  // limb 23 keeps the top 29 bits
  }
  const int64_t M0 = 666643, M1 = 470296, M2 = 654183, M3 = -997805, M4 = 136657, M5 = -683901;
  auto fold = [&](int i) {
    s[i - 12] += s[i] * M0; s[i - 11] += s[i] * M1; s[i - 10] += s[i] * M2;
    s[i - 9] += s[i] * M3;  s[i - 8] += s[i] * M4;  s[i - 7] += s[i] * M5;
    s[i] = 0;
  };
#pragma unroll
  for (int i = 23; i >= 18; i--) fold(i);
#pragma unroll
  for (int i = 6; i < 17; i++) { int64_t c = (s[i] + (1 << 20)) >> 21; s[i] -= c << 21; s[i + 1] += c; }
#pragma unroll
  for (int i = 17; i >= 12; i--) fold(i);
#pragma unroll
  for (int i = 0; i < 12; i++) { int64_t c = (s[i] + (1 << 20)) >> 21; s[i] -= c << 21; s[i + 1] += c; }
  fold(12);
#pragma unroll
  for (int i = 0; i < 12; i++) { int64_t c = s[i] >> 21; s[i] -= c << 21; s[i + 1] += c; }
  fold(12);
#pragma unroll
  for (int i = 0; i < 12; i++) { int64_t c = s[i] >> 21; s[i] -= c << 21; s[i + 1] += c; }
  // limbs 0..11 are now in [0, 2^21) and s[12] in {0, 1}: value < 2^253.  Pack to eight words and subtract l while >= l.
  uint64_t acc[8];
#pragma unroll
  for (int k = 0; k < 8; k++) acc[k] = 0;
#pragma unroll
  for (int i = 0; i < 13; i++) {
    const int bit = 21 * i, w = bit >> 5, sh = bit & 31;
    uint64_t v = (uint64_t)s[i] << sh;
    if (w < 8) acc[w] |= v & 0xffffffffull;
    if (w + 1 < 8) acc[w + 1] |= v >> 32;
  }
#pragma unroll
  for (int k = 0; k < 8; k++) out[k] = (uint32_t)acc[k];
  sc_sub_l_if_ge(out);
  sc_sub_l_if_ge(out);
}

}  // namespace tmx
