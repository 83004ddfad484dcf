// SHA-256 / SHA-512 (FIPS 180-4) for one message per lane.
// SHA-256: validator leaves, RFC-6962 inner nodes, header leaves and proofs
//   (reference circuits/builder/validator.rs:209-252, verify.rs:180-222, shared.rs:169-207,
//    circuits/input/tendermint_utils.rs:351-393).
// SHA-512: EdDSA challenge hash (reference circuits/builder/verify.rs:248-259 via plonky2x).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tmx {

__device__ __constant__ const uint32_t K_SHA256[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
    0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
    0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
    0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
    0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
    0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};

__device__ __constant__ const uint64_t K_SHA512[80] = {
    0x428a2f98d728ae22ULL, 0x7137449123ef65cdULL, 0xb5c0fbcfec4d3b2fULL, 0xe9b5dba58189dbbcULL, 0x3956c25bf348b538ULL,
    0x59f111f1b605d019ULL, 0x923f82a4af194f9bULL, 0xab1c5ed5da6d8118ULL, 0xd807aa98a3030242ULL, 0x12835b0145706fbeULL,
    0x243185be4ee4b28cULL, 0x550c7dc3d5ffb4e2ULL, 0x72be5d74f27b896fULL, 0x80deb1fe3b1696b1ULL, 0x9bdc06a725c71235ULL,
    0xc19bf174cf692694ULL, 0xe49b69c19ef14ad2ULL, 0xefbe4786384f25e3ULL, 0x0fc19dc68b8cd5b5ULL, 0x240ca1cc77ac9c65ULL,
    0x2de92c6f592b0275ULL, 0x4a7484aa6ea6e483ULL, 0x5cb0a9dcbd41fbd4ULL, 0x76f988da831153b5ULL, 0x983e5152ee66dfabULL,
    0xa831c66d2db43210ULL, 0xb00327c898fb213fULL, 0xbf597fc7beef0ee4ULL, 0xc6e00bf33da88fc2ULL, 0xd5a79147930aa725ULL,
    0x06ca6351e003826fULL, 0x142929670a0e6e70ULL, 0x27b70a8546d22ffcULL, 0x2e1b21385c26c926ULL, 0x4d2c6dfc5ac42aedULL,
    0x53380d139d95b3dfULL, 0x650a73548baf63deULL, 0x766a0abb3c77b2a8ULL, 0x81c2c92e47edaee6ULL, 0x92722c851482353bULL,
    0xa2bfe8a14cf10364ULL, 0xa81a664bbc423001ULL, 0xc24b8b70d0f89791ULL, 0xc76c51a30654be30ULL, 0xd192e819d6ef5218ULL,
    0xd69906245565a910ULL, 0xf40e35855771202aULL, 0x106aa07032bbd1b8ULL, 0x19a4c116b8d2d0c8ULL, 0x1e376c085141ab53ULL,
    0x2748774cdf8eeb99ULL, 0x34b0bcb5e19b48a8ULL, 0x391c0cb3c5c95a63ULL, 0x4ed8aa4ae3418acbULL, 0x5b9cca4f7763e373ULL,
    0x682e6ff3d6b2b8a3ULL, 0x748f82ee5defb2fcULL, 0x78a5636f43172f60ULL, 0x84c87814a1f0ab72ULL, 0x8cc702081a6439ecULL,
    0x90befffa23631e28ULL, 0xa4506cebde82bde9ULL, 0xbef9a3f7b2c67915ULL, 0xc67178f2e372532bULL, 0xca273eceea26619cULL,
    0xd186b8c721c0c207ULL, 0xeada7dd6cde0eb1eULL, 0xf57d4f7fee6ed178ULL, 0x06f067aa72176fbaULL, 0x0a637dc5a2c898a6ULL,
    0x113f9804bef90daeULL, 0x1b710b35131c471bULL, 0x28db77f523047d84ULL, 0x32caab7b40c72493ULL, 0x3c9ebe0a15c9bebcULL,
    0x431d67c49c100d4cULL, 0x4cc5d4becb3e42b6ULL, 0x597f299cfc657e2aULL, 0x5fcb6fab3ad6faecULL, 0x6c44198c4a475817ULL};

__device__ __forceinline__ uint32_t rotr32(uint32_t x, int n) { return __builtin_rotateright32(x, n); }
__device__ __forceinline__ uint64_t rotr64(uint64_t x, int n) { return __builtin_rotateright64(x, n); }

// one compression; w[16] holds the big-endian message words and is clobbered (rolling schedule)
__device__ __forceinline__ void sha256_compress(uint32_t st[8], uint32_t w[16]) {
  uint32_t a = st[0], b = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
#pragma unroll
  for (int i = 0; i < 64; i++) {
    if (i >= 16) {
      uint32_t w15 = w[(i + 1) & 15], w2 = w[(i + 14) & 15];
      uint32_t s0 = rotr32(w15, 7) ^ rotr32(w15, 18) ^ (w15 >> 3);
      uint32_t s1 = rotr32(w2, 17) ^ rotr32(w2, 19) ^ (w2 >> 10);
      w[i & 15] = w[i & 15] + s0 + w[(i + 9) & 15] + s1;
    }
    uint32_t t1 = h + (rotr32(e, 6) ^ rotr32(e, 11) ^ rotr32(e, 25)) + ((e & f) ^ (~e & g)) + K_SHA256[i] + w[i & 15];
    uint32_t t2 = (rotr32(a, 2) ^ rotr32(a, 13) ^ rotr32(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
    h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
  }
  st[0] += a; st[1] += b; st[2] += c; st[3] += d; st[4] += e; st[5] += f; st[6] += g; st[7] += h;
}
__device__ __forceinline__ void sha256_init(uint32_t st[8]) {
  st[0] = 0x6a09e667; st[1] = 0xbb67ae85; st[2] = 0x3c6ef372; st[3] = 0xa54ff53a;
  st[4] = 0x510e527f; st[5] = 0x9b05688c; st[6] = 0x1f83d9ab; st[7] = 0x5be0cd19;
}

// Byte-addressed message of up to 119 bytes (two blocks).  get(i) returns message byte i (i < len).
// Digest returned as eight big-endian words (word k = digest bytes 4k..4k+3).
template <typename GetByte>
__device__ __forceinline__ void sha256_short(GetByte get, uint32_t len, uint32_t dig[8]) {
  sha256_init(dig);
  const uint32_t nblk = (len + 9 <= 64) ? 1 : 2;
  for (uint32_t blk = 0; blk < nblk; blk++) {
    uint32_t w[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
      uint32_t word = 0;
#pragma unroll
      for (int b = 0; b < 4; b++) {
        uint32_t pos = blk * 64 + 4 * k + b;
        uint32_t byte = pos < len ? (uint32_t)get(pos) : (pos == len ? 0x80u : 0u);
        word = (word << 8) | byte;
      }
      w[k] = word;
    }
    if (blk == nblk - 1) w[15] = len * 8;  // bit length (< 2^32), w[14] already 0
    sha256_compress(dig, w);
  }
}

// RFC-6962 leaf of a short field held in registers: SHA-256(0x00 || first len bytes of f), f = up to 80 bytes as 20 little-endian
// words whose bytes from the end of the data on are zero; len <= 79.  Fully unrolled, register-indexed only.
__device__ __forceinline__ void sha256_leaf80(const uint32_t f[20], uint32_t len, uint32_t dig[8]) {
  const uint32_t p = len + 1;  // position of the 0x80 marker in the message 00 | data
  uint32_t m[32];
#pragma unroll
  for (int j = 0; j < 32; j++) {
    const uint32_t prev = (j >= 1 && j <= 20) ? f[j - 1] & 0xff000000u : 0u;
    const uint32_t cur = j < 20 ? __builtin_bswap32(f[j]) >> 8 : 0u;
    m[j] = prev | cur;
    m[j] |= ((uint32_t)j == (p >> 2)) ? (0x80u << (24 - 8 * (p & 3))) : 0u;
  }
  const bool two = p + 9 > 64;
  sha256_init(dig);
  uint32_t w[16];
#pragma unroll
  for (int k = 0; k < 16; k++) w[k] = m[k];
  if (!two) w[15] = p * 8;
  sha256_compress(dig, w);
  if (two) {
#pragma unroll
    for (int k = 0; k < 16; k++) w[k] = m[16 + k];
    w[15] = p * 8;
    sha256_compress(dig, w);
  }
}

// RFC-6962 inner node: SHA-256(0x01 || L || R), L and R as eight big-endian words each (65 bytes, 2 blocks)
__device__ __forceinline__ void sha256_inner(const uint32_t l[8], const uint32_t r[8], uint32_t dig[8]) {
  uint32_t w[16];
  sha256_init(dig);
  // bytes: 01 l0..l31 r0..r30 | r31 80 00.. len
  w[0] = 0x01000000u | (l[0] >> 8);
#pragma unroll
  for (int k = 1; k < 8; k++) w[k] = (l[k - 1] << 24) | (l[k] >> 8);
  w[8] = (l[7] << 24) | (r[0] >> 8);
#pragma unroll
  for (int k = 1; k < 8; k++) w[8 + k] = (r[k - 1] << 24) | (r[k] >> 8);
  sha256_compress(dig, w);
  w[0] = (r[7] << 24) | 0x00800000u;
#pragma unroll
  for (int k = 1; k < 15; k++) w[k] = 0;
  w[15] = 65 * 8;
  sha256_compress(dig, w);
}

__device__ __forceinline__ void sha512_compress(uint64_t st[8], uint64_t w[16]) {
  uint64_t a = st[0], b = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
#pragma unroll
  for (int i = 0; i < 80; i++) {
    if (i >= 16) {
      uint64_t w15 = w[(i + 1) & 15], w2 = w[(i + 14) & 15];
      uint64_t s0 = rotr64(w15, 1) ^ rotr64(w15, 8) ^ (w15 >> 7);
      uint64_t s1 = rotr64(w2, 19) ^ rotr64(w2, 61) ^ (w2 >> 6);
      w[i & 15] = w[i & 15] + s0 + w[(i + 9) & 15] + s1;
    }
    uint64_t t1 = h + (rotr64(e, 14) ^ rotr64(e, 18) ^ rotr64(e, 41)) + ((e & f) ^ (~e & g)) + K_SHA512[i] + w[i & 15];
    uint64_t t2 = (rotr64(a, 28) ^ rotr64(a, 34) ^ rotr64(a, 39)) + ((a & b) ^ (a & c) ^ (b & c));
    h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
  }
  st[0] += a; st[1] += b; st[2] += c; st[3] += d; st[4] += e; st[5] += f; st[6] += g; st[7] += h;
}
__device__ __forceinline__ void sha512_init(uint64_t st[8]) {
  st[0] = 0x6a09e667f3bcc908ULL; st[1] = 0xbb67ae8584caa73bULL; st[2] = 0x3c6ef372fe94f82bULL; st[3] = 0xa54ff53a5f1d36f1ULL;
  st[4] = 0x510e527fade682d1ULL; st[5] = 0x9b05688c2b3e6c1fULL; st[6] = 0x1f83d9abfb41bd6bULL; st[7] = 0x5be0cd19137e2179ULL;
}
// message of up to 239 bytes (two blocks) addressed through get(i)
template <typename GetByte>
__device__ __forceinline__ void sha512_short(GetByte get, uint32_t len, uint64_t dig[8]) {
  sha512_init(dig);
  const uint32_t nblk = (len + 17 <= 128) ? 1 : 2;
  for (uint32_t blk = 0; blk < nblk; blk++) {
    uint64_t w[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
      uint64_t word = 0;
#pragma unroll
      for (int b = 0; b < 8; b++) {
        uint32_t pos = blk * 128 + 8 * k + b;
        uint32_t byte = pos < len ? (uint32_t)get(pos) : (pos == len ? 0x80u : 0u);
        word = (word << 8) | byte;
      }
      w[k] = word;
    }
    if (blk == nblk - 1) w[15] = (uint64_t)len * 8;
    sha512_compress(dig, w);
  }
}

}  // namespace tmx
