/* ORACLE -- TEST INFRASTRUCTURE ONLY (see tmxo.h).
 * Goldilocks NTT / coset low-degree extension, the primitive SURVEY 8(f) rank 2 names as the step after the witness fill of a
 * plonky2-style prover.  plonky2 itself is an un-vendored dependency (Cargo.lock: plonky2_field 0.2.0 @ mir-protocol/plonky2#4f8e6315),
 * so this restates the published definitions -- PARITY UNPINNED against plonky2's own code:
 *   p = 2^64 - 2^32 + 1; domain = (primitive 2^32-th root of unity, coset shift), by default the constants recalled from plonky2's
 *   GoldilocksField (POWER_OF_TWO_GENERATOR 7277203076849721926 = MULTIPLICATIVE_GROUP_GENERATOR 14293326489335486720 ^ ((p-1)/2^32));
 *   tmxo_ntt_set_domain selects another one, e.g. g = 7 with 0x185629dcda58878c (Plonky3 / winterfell),
 *   forward  X[j] = sum_i x[i] omega_N^(ij)   (natural order in and out),   inverse  x[i] = N^-1 sum_j X[j] omega_N^(-ij),
 *   coset LDE: coefficients c = INTT_N(x); y = NTT_M(c_i g^i, zero padded), M = N 2^b  (evaluations on the coset g <omega_M>).
 * Arithmetic with unsigned __int128 and %, textbook iterative radix-2 with a bit-reversal permutation: nothing shared with the HIP path. */
#include <stdlib.h>
#include <string.h>

#include "tmxo.h"

#define GL_P 0xffffffff00000001ull

static uint64_t gl_add(uint64_t a, uint64_t b) { return (uint64_t)(((unsigned __int128)a + b) % GL_P); }
static uint64_t gl_sub(uint64_t a, uint64_t b) { return (uint64_t)(((unsigned __int128)a + GL_P - b) % GL_P); }
static uint64_t gl_mul(uint64_t a, uint64_t b) { return (uint64_t)(((unsigned __int128)a * b) % GL_P); }
uint64_t tmxo_gl_pow(uint64_t b, uint64_t e) {
  uint64_t r = 1;
  b %= GL_P;
  while (e) {
    if (e & 1) r = gl_mul(r, b);
    b = gl_mul(b, b);
    e >>= 1;
  }
  return r;
}
static uint64_t g_root = 7277203076849721926ull, g_shift = 14293326489335486720ull;
void tmxo_ntt_set_domain(uint64_t root_2_32, uint64_t coset_shift) { g_root = root_2_32; g_shift = coset_shift; }
uint64_t tmxo_gl_root(uint32_t log_n) { return tmxo_gl_pow(g_root, 1ull << (32 - log_n)); }

/* in place, natural order in and out; values are taken mod p */
void tmxo_ntt(uint64_t* x, uint32_t log_n, int inverse) {
  const uint64_t n = 1ull << log_n;
  for (uint64_t i = 0; i < n; i++) {
    x[i] %= GL_P;
    uint64_t r = 0;
    for (uint32_t b = 0; b < log_n; b++) r |= ((i >> b) & 1) << (log_n - 1 - b);
    if (r > i) { uint64_t t = x[i] % GL_P; x[i] = x[r] % GL_P; x[r] = t; }
  }
  uint64_t w_n = tmxo_gl_root(log_n);
  if (inverse) w_n = tmxo_gl_pow(w_n, GL_P - 2);
  for (uint32_t s = 1; s <= log_n; s++) {
    const uint64_t len = 1ull << s, half = len >> 1;
    const uint64_t w_len = tmxo_gl_pow(w_n, n >> s);
    for (uint64_t k = 0; k < n; k += len) {
      uint64_t w = 1;
      for (uint64_t j = 0; j < half; j++) {
        const uint64_t u = x[k + j], v = gl_mul(x[k + j + half], w);
        x[k + j] = gl_add(u, v);
        x[k + j + half] = gl_sub(u, v);
        w = gl_mul(w, w_len);
      }
    }
  }
  if (inverse) {
    const uint64_t n_inv = tmxo_gl_pow(n % GL_P, GL_P - 2);
    for (uint64_t i = 0; i < n; i++) x[i] = gl_mul(x[i], n_inv);
  }
}

/* out[0 .. n 2^b) = evaluations on the coset shift <omega_(n 2^b)> of the polynomial with evaluations in[0 .. n) on <omega_n> */
void tmxo_lde(const uint64_t* in, uint64_t* out, uint32_t log_n, uint32_t log_blowup) {
  const uint64_t n = 1ull << log_n, m = n << log_blowup;
  memset(out, 0, m * sizeof(uint64_t));
  memcpy(out, in, n * sizeof(uint64_t));
  tmxo_ntt(out, log_n, 1);
  uint64_t s = 1;
  for (uint64_t i = 0; i < n; i++) {
    out[i] = gl_mul(out[i], s);
    s = gl_mul(s, g_shift);
  }
  tmxo_ntt(out, log_n + log_blowup, 0);
}
