// Launch wrappers of poseidon.hip (host side: plain C++, no HIP headers needed)
#pragma once
#include <cstdint>

namespace tmx {

constexpr uint32_t POS_T = 12, POS_RF = 8, POS_RP = 22, POS_ROUNDS = POS_RF + POS_RP;
constexpr uint32_t POS_CONST_WORDS = POS_ROUNDS * POS_T + 2 * POS_T;  // round constants | MDS circulant row | MDS diagonal (u64 each)

// Partial rounds in groups of three (POS_MODE_MERGE3): with M the MDS matrix over the integers and u the state with its round constants
// added, three partial rounds are  u' = M^3 v + d1 M^2[:,0] + d2 M[:,0] + K3  with v = u, v0 = u0^7, and the two inner S-box inputs
//   y1 = M[0,:] v + K1,   y2 = M^2[0,:] v + d1 M[0,0] + K2,   d_j = y_j^7 - y_j
// -- ONE dense layer (by M^3) and two row products instead of three dense layers, as long as every entry of M, M^2, M^3 stays below 2^26
// (plonky2's circulant: 2^24.1), so that the 32-bit-limb form of the layer still applies.  The tables sit behind the constants in the
// device buffer (u64 words, built on the host when the constants are uploaded):
//   per group g (rounds 5 + 3g ...): K1, K2, K3[12] (u64)  |  as u32: row 0 of M | row 0 of M^2 | column 0 of M^2 | column 0 of M | M^3 (row-major)
// Round 4 (the first partial round) runs alone, then seven groups cover rounds 5 .. 25.
constexpr uint32_t POS_MERGE_GROUPS = 7, POS_MERGE_FIRST = POS_RF / 2 + 1;
constexpr uint32_t POS_X_GROUPS = POS_CONST_WORDS, POS_X_GROUP_WORDS = 2 + POS_T, POS_X_U32 = POS_X_GROUPS + POS_MERGE_GROUPS * POS_X_GROUP_WORDS;  // (u64 words)
constexpr uint32_t POS_U_R1 = 0, POS_U_R2 = POS_T, POS_U_C2 = 2 * POS_T, POS_U_C1 = 3 * POS_T, POS_U_M3 = 4 * POS_T, POS_U_WORDS = 4 * POS_T + POS_T * POS_T;  // (u32 words)
constexpr uint32_t POS_CONST_WORDS_EXT = POS_X_U32 + POS_U_WORDS / 2;
static_assert(POS_MERGE_FIRST + 3 * POS_MERGE_GROUPS == POS_RF / 2 + POS_RP, "the groups end with the last partial round");
enum : int { POS_MODE_GENERAL = 0, POS_MODE_SMALL = 1, POS_MODE_MERGE3 = 2 };

// d_consts: POS_CONST_WORDS_EXT u64 in device memory (canonical values).  mode: POS_MODE_SMALL = every MDS entry < 2^16 (the 32-bit-limb
// MDS layer), POS_MODE_MERGE3 = that and the merged partial rounds, POS_MODE_GENERAL = field products.
int launch_poseidon_permute(const void* d_consts, int mode, uint32_t n, const void* d_in, void* d_out, void* stream);
// leaf digests of the 2^log_n rows of n_cols column-major columns -> d_digests[2^log_n][4]
int launch_poseidon_leaves(const void* d_consts, int mode, uint32_t log_n, uint32_t n_cols, const void* d_cols, void* d_digests, void* stream);
// one tree level: out[i] = two_to_one(in[2 i], in[2 i + 1]), i < n_out
int launch_poseidon_level(const void* d_consts, int mode, uint64_t n_out, const void* d_in, void* d_out, void* stream);

}  // namespace tmx
