// Launch wrappers of ntt.hip (host side: plain C++, no HIP headers needed)
#pragma once
#include <cstdint>

namespace tmx {

struct NttPass {
  uint32_t log_n;        // size of the whole transform (selects the twiddle table omega_N)
  uint32_t log_l;        // length of the sub-transforms of this pass
  uint32_t log_t;        // sub-transforms per tile (tile = 2^(log_l + log_t) <= 4096 elements)
  uint32_t tiles_per_col;
  uint64_t n_sub;        // sub-transforms per column
  uint64_t col_stride_in, t_stride_in, j_stride_in;     // element strides: column, sub-transform, index inside it
  uint64_t col_stride_out, t_stride_out, j_stride_out;
  uint32_t inverse;
  uint32_t post_twiddle;  // multiply output k of sub-transform t by omega_N^(t k) (four-step pass A)
  uint64_t scale;         // multiply every output by this (N^-1 on the last pass of an inverse transform), 1 = none
  uint32_t kinv;          // omega_16 of this pass's direction = (2^12)^k; kinv k = 1 (mod 16): load order of the radix-16 groups
  uint32_t j_nonzero;     // 0: every input is read; else inputs with index j >= j_nonzero inside a sub-transform are ZERO and are not read
                          // (the forward transform of a zero-padded column: coset LDE)
  uint64_t tm_t_stride;   // stride of the post-twiddle table along t (= t_stride_out for the four-step matrix; 0 for a vector indexed by k only)
};

int launch_ntt_table(void* d_w, uint32_t log_n, uint64_t root_2_32, void* stream);
int launch_ntt_matrix(void* d_m, const void* d_w, uint32_t log_n, uint32_t log_n2, bool inverse, void* stream);  // the pass-A twiddles
int launch_ntt_pass(const NttPass& P, uint32_t n_cols, const void* d_in, void* d_out, const void* d_w, const void* d_m, void* stream);
int launch_lde_expand(void* d_buf, uint32_t log_n, uint32_t log_m, uint32_t n_cols, uint64_t shift, void* stream);
// S[i] = scale * shift^i, i < 2^log_n (the coefficient scaling of a coset LDE with the inverse transform's 1 / N folded in)
int launch_lde_scale_table(void* d_s, uint32_t log_n, uint64_t shift, uint64_t scale, void* stream);

}  // namespace tmx
