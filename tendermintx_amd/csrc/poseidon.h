// Launch wrappers of poseidon.hip (host side: plain C++, no HIP headers needed)
#pragma once
#include <cstdint>

namespace tmx {

constexpr uint32_t POS_T = 12, POS_RF = 8, POS_RP = 22, POS_ROUNDS = POS_RF + POS_RP;
constexpr uint32_t POS_CONST_WORDS = POS_ROUNDS * POS_T + 2 * POS_T;  // round constants | MDS circulant row | MDS diagonal (u64 each)

// d_consts: POS_CONST_WORDS u64 in device memory (canonical values).  mds_small: every MDS entry < 2^16 (the 32-bit-limb MDS layer).
int launch_poseidon_permute(const void* d_consts, bool mds_small, uint32_t n, const void* d_in, void* d_out, void* stream);
// leaf digests of the 2^log_n rows of n_cols column-major columns -> d_digests[2^log_n][4]
int launch_poseidon_leaves(const void* d_consts, bool mds_small, uint32_t log_n, uint32_t n_cols, const void* d_cols, void* d_digests, void* stream);
// one tree level: out[i] = two_to_one(in[2 i], in[2 i + 1]), i < n_out
int launch_poseidon_level(const void* d_consts, bool mds_small, uint64_t n_out, const void* d_in, void* d_out, void* stream);

}  // namespace tmx
