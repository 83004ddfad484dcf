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

// out = (a * b + c) mod l, all eight-word little-endian (used by the synthetic-workload signer only)
__device__ __forceinline__ void sc_muladd(const uint32_t a[8], const uint32_t b[8], const uint32_t c[8], uint32_t out[8]) {
  uint32_t p[17];
#pragma unroll
  for (int i = 0; i < 17; i++) p[i] = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    uint32_t carry = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      uint64_t t = (uint64_t)a[i] * b[j] + p[i + j] + carry;
      p[i + j] = (uint32_t)t;
      carry = (uint32_t)(t >> 32);
    }
    p[i + 8] = carry;
  }
  uint32_t carry = 0;
#pragma unroll
  for (int i = 0; i < 17; i++) {
    uint64_t t = (uint64_t)p[i] + (i < 8 ? c[i] : 0u) + carry;
    p[i] = (uint32_t)t;
    carry = (uint32_t)(t >> 32);
  }
  sc_reduce_bits<17>(p, out);
}

}  // namespace tmx
