// Goldilocks field arithmetic for the HIP kernels (ntt.hip, poseidon.hip): p = 2^64 - 2^32 + 1, 2^64 = 2^32 - 1 (mod p), 2^96 = -1.
// 64-bit integer math only (v_mad_u64_u32 products, add / sub with carry): nothing here is a dense contraction, no MFMA.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace tmx {

constexpr uint64_t GL_P = 0xffffffff00000001ull, GL_EPS = 0xffffffffull;

__device__ __forceinline__ uint64_t gl_canon(uint64_t x) { return x >= GL_P ? x - GL_P : x; }
// The same for a value that is "any representative" out of the lazy forms below: x >= p needs the high word to be all ones, once in 2^32
// values -- one compare and a wave-wide branch instead of a compare, two selects and a 64-bit subtract in every butterfly.
__device__ __forceinline__ uint64_t gl_canon_rare(uint64_t x) {
  if (__builtin_amdgcn_ballot_w64(x >= GL_P)) {
    asm volatile("; gl_canon_rare: x >= p");  // (keeps the block a branch target: if-converted, it is the plain form again)
    x = gl_canon(x);
  }
  return x;
}
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
// "Lazy" forms for the transform kernels: values are any 64-bit representative of their class; only the second operand of a butterfly is
// made canonical (both corrections below rely on b < p), and the final store canonicalizes.  (Saves the closing compare-and-subtract of
// every product, shift and load: 9 % of the kernel's instructions.)
__device__ __forceinline__ uint64_t gl_add_lazy(uint64_t a, uint64_t b) {  // any a, b < p -> any
  unsigned long long s;
  const bool carry = __builtin_uaddll_overflow(a, b, &s);
  return s + (carry ? GL_EPS : 0ull);  // after a carry s <= p - 2: cannot wrap again
}
__device__ __forceinline__ uint64_t gl_sub_lazy(uint64_t a, uint64_t b) {  // any a, b < p -> any
  unsigned long long d;
  const bool borrow = __builtin_usubll_overflow(a, b, &d);
  return d - (borrow ? GL_EPS : 0ull);  // after a borrow d >= 2^64 - (p - 1) > EPS: cannot wrap again
}
__device__ __forceinline__ uint64_t gl_mul_lazy(uint64_t a, uint64_t b) {  // any a, b -> any
  // 128-bit product: four independent 32 x 32 -> 64 products (v_mad_u64_u32, no addend) summed by 32-bit carry chains.  (Chaining the
  // products through the multiply-add's 64-bit addend -- p01 = a0 b1 + hi(p00) ... -- costs two v_mov per addend: a high half has to
  // move to the even register of an aligned pair whose odd register is zero; this form has no move at all: 22 instead of 25 VALU
  // instructions per product, profiles/r04_poseidon_isa.txt.)
  const uint32_t a0 = (uint32_t)a, a1 = (uint32_t)(a >> 32), b0 = (uint32_t)b, b1 = (uint32_t)(b >> 32);
  const uint64_t p00 = (uint64_t)a0 * b0, p01 = (uint64_t)a0 * b1, p10 = (uint64_t)a1 * b0, p11 = (uint64_t)a1 * b1;
  unsigned c1, c2, c3, c4, c5;
  const uint32_t mid_lo = __builtin_addc((uint32_t)p01, (uint32_t)p10, 0u, &c1);
  const uint32_t mid_hi = __builtin_addc((uint32_t)(p01 >> 32), (uint32_t)(p10 >> 32), c1, &c2);  // p01 + p10 = mid_lo + 2^32 mid_hi + 2^64 c2
  const uint32_t l_hi = __builtin_addc((uint32_t)(p00 >> 32), mid_lo, 0u, &c3);
  const uint32_t hi_lo = __builtin_addc((uint32_t)p11, mid_hi, c3, &c4);
  const uint32_t hi_hi = __builtin_addc((uint32_t)(p11 >> 32), c2, c4, &c5);  // (the product is < 2^128: c5 = 0)
  const uint64_t lo = ((uint64_t)l_hi << 32) | (uint32_t)p00;
  // x = lo + 2^64 hi_lo + 2^96 hi_hi = lo + (2^32 - 1) hi_lo - hi_hi
  unsigned long long t0, r;
  const bool borrow = __builtin_usubll_overflow(lo, hi_hi, &t0);
  // lo < hi_hi < 2^32 happens once in 2^32 products: the correction sits behind a wave-wide branch (one scalar branch instead of three
  // vector instructions in every product)
  if (__builtin_amdgcn_ballot_w64(borrow)) {
    asm volatile("; gl_mul_lazy: rare borrow");  // (keeps the block a branch target: if-converted, it costs more than the plain form)
    t0 -= borrow ? GL_EPS : 0ull;                // the wrap added 2^64 = p + EPS
  }
  const uint64_t t1 = (uint64_t)hi_lo * 0xffffffffu;
  const bool carry = __builtin_uaddll_overflow(t0, t1, &r);
  r += carry ? GL_EPS : 0ull;
  return r;
}
__device__ __forceinline__ uint64_t gl_mul(uint64_t a, uint64_t b) { return gl_canon(gl_mul_lazy(a, b)); }
__device__ __forceinline__ uint64_t gl_pow(uint64_t b, uint64_t e) {
  uint64_t r = 1;
  while (e) {
    if (e & 1) r = gl_mul(r, b);
    b = gl_mul(b, b);
    e >>= 1;
  }
  return r;
}

}  // namespace tmx
