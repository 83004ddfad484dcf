// Level-2 trace rows (DESIGN.md "Level-2 trace rows", SURVEY 8a "Level-2" / 8f rank 2): the row-level execution trace of the two scalar
// multiplications, of the EdDSA SHA-512 and of the validator leaf SHA-256 of every lane, plus the N x N match bits -- what the reference
// generates inside Curta's AIR trace generators (`curta_eddsa_verify_sigs_conditional` at reference circuits/builder/verify.rs:248-259,
// `curta_sha256_variable` at validator.rs:228; sources absent, so the layout is this build's own specification, validated by the
// constraint checker under oracle/c).  266 KB of ladder rows per lane: this is the part of the witness that is written, not computed
// -- 9.8 GB per 256-proof batch at N = 128.
//
//   k_trace_ladder_pass1 / _pass2   the double-and-add chain in extended coordinates (one thread per ladder), then the inversions, canonical
//                    limbs and stores with one thread per (ladder, eight rows): see the comment at the kernels
//   k_trace_sha512   one thread per lane, four rounds per flush        k_trace_sha256   one thread per (set, lane)
//   k_trace_match    one thread per (i, j)
//   k_trace_sha256x2 one thread per two-block SHA-256: the inner nodes of the validator trees and the header-proof hashes
#include "trace.h"

#include <hip/hip_runtime.h>

#include <cstdlib>

#include "ge25519.hpp"
#include "inv25519.hpp"
#include "sha2.hpp"

namespace tmx {
namespace {

__device__ __forceinline__ uint32_t ld32(const uint8_t* p) { return *reinterpret_cast<const uint32_t*>(p); }
__device__ __forceinline__ uint64_t ld64(const uint8_t* p) { return *reinterpret_cast<const uint64_t*>(p); }

// plonky2x DUMMY_PUBLIC_KEY / DUMMY_SIGNATURE (kernels.hip has the same words; pinned by tests/test_oracle_kat.py::test_dummy_constants)
__device__ __constant__ const uint32_t T_DUMMY_PK[8] = {0xdde3888au, 0x95f10974u, 0x2ddb52fdu, 0x725dba3cu, 0xbf0967cau, 0x1b12941du, 0x018874f3u, 0x5c6f0fb4u};
__device__ __constant__ const uint32_t T_DUMMY_SIG[16] = {0x9e681437u, 0x11c27854u, 0xa49ded06u, 0x899e5855u, 0xf0bb77bbu, 0x3f50499fu, 0x5b4aa285u, 0x8a063530u,
                                                          0x79162901u, 0x91c62ef9u, 0xd203669bu, 0x37ad87a8u, 0x7e2d48fcu, 0x07bfb2a9u, 0x5a704399u, 0x078c2196u};
// the base point B = (x, 4/5), canonical little-endian words (RFC 8032 section 5.1)
__device__ __constant__ const uint32_t T_BX[8] = {0x8f25d51au, 0xc9562d60u, 0x9525a7b2u, 0x692cc760u, 0xfdd6dc5cu, 0xc0a4e231u, 0xcd6e53feu, 0x216936d3u};
__device__ __constant__ const uint32_t T_BY[8] = {0x66666658u, 0x66666666u, 0x66666666u, 0x66666666u, 0x66666666u, 0x66666666u, 0x66666666u, 0x66666666u};

constexpr int CH = 8;  // ladder rows per inversion batch (2 CH points)

// flush `nv` staged values of every thread's current row group: value e of thread j goes to out[base[j] + off + e]
template <int NV>
__device__ __forceinline__ void coop_flush(const uint32_t (*stage)[NV + 1], const uint64_t* s_base, const uint8_t* s_live, uint64_t off, uint64_t* __restrict__ out) {
  __syncthreads();
  const uint32_t tid = threadIdx.x;
#pragma unroll 1
  for (int j = 0; j < 64; j++) {
    if (!s_live[j]) continue;
    uint64_t* dst = out + s_base[j] + off;
    for (uint32_t e = tid; e < (uint32_t)NV; e += 64) __builtin_nontemporal_store((uint64_t)stage[j][e], dst + e);
  }
  __syncthreads();
}

// ---- ladders, in two passes so that the expensive half (inversions, canonical limbs, the stores) is not a 256-step chain:
//   pass 1  one thread per (lane, ladder): the double-and-add chain in extended coordinates; (X:Y:Z) of dbl_r and add_r of every row go
//           to a scratch buffer (240 B per row, laid out [block][row][word][thread] so that every store / load instruction is coalesced)
//   pass 2  one thread per (ladder, batch of eight rows), a wave = the same 64 ladders as in pass 1: the 16 points of the batch plus the
//           accumulator it starts from made affine together (Montgomery's trick around one safegcd inversion), rows staged in LDS and
//           written by the whole wave (every store instruction covers 512 contiguous bytes of ONE ladder).
// 32 x the threads of the one-pass form for the part that is 60 % of the instructions.  Measured per 256-proof batch at N = 128: one pass
// 8.3 ms (7.2 ms at 32 proofs: a pure latency chain); two passes 2.75 + 4.25 ms (1.9 ms at 32 proofs).
struct LadderIn {
  uint32_t sc[8];  // the scalar
  ge_affc P;       // the point that is added
  bool decoded;
};
__device__ __forceinline__ LadderIn ladder_inputs(uint32_t lane, uint32_t k, const uint8_t* __restrict__ in_target, const uint8_t* __restrict__ ed,
                                                  uint32_t ed_stride, bool want_point) {
  LadderIn L;
  const uint8_t* rec = in_target + (size_t)lane * VR_STRIDE;
  const uint8_t* er = ed + (size_t)lane * ed_stride;
  const bool is_signed = rec[VR_OFF_FLAGS] & 1;
  L.decoded = ld32(er + ED_OFF_DECODE_OK) != 0;
  uint32_t pw[16];
#pragma unroll
  for (int w = 0; w < 8; w++) {
    L.sc[w] = k ? ld32(er + ED_OFF_H + 4 * w) : (is_signed ? ld32(rec + VR_OFF_SIG + 32 + 4 * w) : T_DUMMY_SIG[8 + w]);
    pw[w] = (k && L.decoded) ? ld32(er + ED_OFF_PTS + 4 * w) : T_BX[w];           // A.x | B.x (an undecodable lane computes on B, stores zeros)
    pw[8 + w] = (k && L.decoded) ? ld32(er + ED_OFF_PTS + 32 + 4 * w) : T_BY[w];  // A.y | B.y
  }
  if (want_point) {
    const fe x = fe_from_words(pw), y = fe_from_words(pw + 8);
    L.P.ypx = fe_add(y, x); L.P.ymx = fe_sub(y, x); L.P.xy2d = fe_mul(fe_mul(x, y), K_2D);
  }
  return L;
}
__device__ __forceinline__ uint32_t scalar_bit(const uint32_t sc[8], int r) {  // bit 255 - r
  const int b = 255 - r;
  uint32_t word = sc[0];
#pragma unroll
  for (int w = 1; w < 8; w++) word = (b >> 5) == w ? sc[w] : word;
  return (word >> (b & 31)) & 1u;
}
constexpr uint32_t TR_PT_WORDS = 60;  // per row in the scratch buffer: dbl (X, Y, Z), add (X, Y, Z), ten limbs each

__global__ __launch_bounds__(64) void k_trace_ladder_pass1(uint32_t n_lanes, const uint8_t* __restrict__ in_target, const uint8_t* __restrict__ ed,
                                                           uint32_t ed_stride, int32_t* __restrict__ pts, uint32_t row0, uint32_t row1) {
  const uint32_t t = threadIdx.x, id = blockIdx.x * 64u + t;
  const bool live = id < 2u * n_lanes;
  const LadderIn L = ladder_inputs(live ? id >> 1 : 0u, id & 1u, in_target, ed, ed_stride, true);
  int32_t* o = pts + (size_t)blockIdx.x * TR_LADDER_ROWS * TR_PT_WORDS * 64u + t;
  ge_proj acc = ext_to_proj(ge_identity());
  if (row0) {  // a later segment of the chain: the accumulator is row0 - 1's nxt, already in the scratch buffer
    const int32_t* prev = o + (size_t)(row0 - 1) * TR_PT_WORDS * 64u + (size_t)scalar_bit(L.sc, (int)row0 - 1) * 30u * 64u;
#pragma unroll
    for (int l = 0; l < 10; l++) { acc.X.v[l] = prev[l * 64]; acc.Y.v[l] = prev[(10 + l) * 64]; acc.Z.v[l] = prev[(20 + l) * 64]; }
  }
#pragma unroll 1
  for (int r = (int)row0; r < (int)row1; r++) {
    const uint32_t bit = scalar_bit(L.sc, r);
    const ge_ext d = comp_to_ext(ge_double(acc));
    const ge_proj a = comp_to_proj(ge_add_affc(d, L.P));  // (the sum's T is never needed: the next step doubles from (X : Y : Z))
    int32_t* row = o + (size_t)r * TR_PT_WORDS * 64u;
#pragma unroll
    for (int l = 0; l < 10; l++) {
      row[(0 + l) * 64] = d.X.v[l]; row[(10 + l) * 64] = d.Y.v[l]; row[(20 + l) * 64] = d.Z.v[l];
      row[(30 + l) * 64] = a.X.v[l]; row[(40 + l) * 64] = a.Y.v[l]; row[(50 + l) * 64] = a.Z.v[l];
    }
    acc.X = fe_select(d.X, a.X, bit); acc.Y = fe_select(d.Y, a.Y, bit); acc.Z = fe_select(d.Z, a.Z, bit);
  }
}

// Everything below is unrolled with compile-time indices: no indexed private array (the first version kept X, Y, Z, prefix products and the
// affine words of a batch in 4 KB of scratch per thread -- 8 GB of spill traffic per 256-proof batch, twice the rows themselves).
// Montgomery's trick with SUFFIX products so that the points come out in row order: total = Z_0 ... Z_16, then for i = 0, 1, ...:
// 1 / Z_i = inv_i * suf_{i+1} with inv_i = 1 / suf_i, inv_{i+1} = inv_i * Z_i.  Only suf_4, suf_8, suf_12, suf_16 are kept (40 VGPRs);
// the three in between are recomputed per group of four from the Z's (re-read from the scratch buffer: coalesced, mostly L2).
// One safegcd inversion serves GCH batches (the inversion was 35 % of this pass's instructions with one per batch): a first backward
// sweep over the Z's of all GCH batches leaves the suffix products at the batch boundaries in LDS, the inversion of the total gives
// 1 / (everything from batch 0 on), and every batch then runs the schedule above with "everything behind this batch" folded into its
// suffix products; the running inverse ends a batch as the inverse of what is left, i.e. where the next batch starts.
template <int GCH>
__global__ __launch_bounds__(64) void k_trace_ladder_pass2(uint32_t n_lanes, uint32_t n, const uint8_t* __restrict__ in_target, const uint8_t* __restrict__ ed,
                                                           uint32_t ed_stride, const int32_t* __restrict__ pts, uint64_t* __restrict__ out,
                                                           uint64_t proof_stride, uint32_t chunk0) {
  __shared__ uint32_t stage[64][TR_LADDER_ROW + 1];
  __shared__ uint64_t s_base[64];
  __shared__ uint8_t s_live[64];
  __shared__ int32_t s_behind[GCH > 1 ? GCH - 1 : 1][10][64];  // [q - 1]: product of the Z's of batches q .. GCH - 1 of this thread
  const uint32_t t = threadIdx.x, id = blockIdx.x * 64u + t, c_first = chunk0 + blockIdx.y * GCH;  // batch c = rows CH c .. CH c + CH - 1
  const bool live = id < 2u * n_lanes;
  const uint32_t lane = live ? id >> 1 : 0u, k = id & 1u;
  {
    const uint32_t p = lane / n, i = lane - p * n;
    s_base[t] = (uint64_t)p * proof_stride + (uint64_t)((2u * i + k) * TR_LADDER_ROWS) * TR_LADDER_ROW;
    s_live[t] = live ? 1 : 0;
  }
  const LadderIn L = ladder_inputs(lane, k, in_target, ed, ed_stride, false);
  const int32_t* src = pts + (size_t)blockIdx.x * TR_LADDER_ROWS * TR_PT_WORDS * 64u + t;
  static_assert(CH == 8, "the unrolled schedule below is written for batches of eight rows (17 points)");
  // point i of batch c: 0 = the accumulator the batch starts from (row CH c - 1's nxt; the identity for c = 0); 1 + 2j / 2 + 2j = dbl / add of row j
  auto coord = [&](uint32_t c, uint32_t start_bit, int i, int which_coord) -> fe {  // which_coord: 0 X, 1 Y, 2 Z
    fe v;
    if (i == 0 && c == 0) { v = which_coord == 0 ? fe_zero() : fe_one(); return v; }
    const int r = i == 0 ? (int)(CH * c) - 1 : (int)(CH * c) + (i - 1) / 2;
    const uint32_t which = i == 0 ? start_bit : (uint32_t)((i - 1) & 1);
    const int32_t* row = src + (size_t)r * TR_PT_WORDS * 64u + (size_t)(which * 30u + (uint32_t)which_coord * 10u) * 64u;
#pragma unroll
    for (int l = 0; l < 10; l++) v.v[l] = row[l * 64];
    return v;
  };
  auto start_bit_of = [&](uint32_t c) -> uint32_t { return c ? scalar_bit(L.sc, (int)(CH * c) - 1) : 0u; };
  // sweep 1: the product of every Z of the GCH batches, suffixes at the batch boundaries to LDS
  fe inv;
  {
    fe tot = fe_one();
#pragma unroll 1
    for (int q = GCH - 1; q >= 0; q--) {
      const uint32_t c = c_first + (uint32_t)q, sb = start_bit_of(c);
#pragma unroll
      for (int i = 16; i >= 0; i--) tot = (q == GCH - 1 && i == 16) ? coord(c, sb, 16, 2) : fe_mul(coord(c, sb, i, 2), tot);
      if (q > 0) {
#pragma unroll
        for (int l = 0; l < 10; l++) s_behind[q - 1][l][t] = tot.v[l];
      }
    }
    inv = fe_invert_safegcd(tot);
  }
  const uint32_t z = L.decoded ? 0xffffffffu : 0u;  // an undecodable lane: all-zero rows (Level-1 reports zero points there too)
  uint32_t accw[16], dblw[16];
#pragma unroll 1
  for (int q = 0; q < GCH; q++) {
    const uint32_t c = c_first + (uint32_t)q, start_bit = start_bit_of(c);
    const bool last = q == GCH - 1;
    fe behind = fe_one();  // product of everything behind this batch
    if (!last) {
#pragma unroll
      for (int l = 0; l < 10; l++) behind.v[l] = s_behind[last ? 0 : q][l][t];
    }
    fe ck[4];  // suf_4, suf_8, suf_12, suf_16 (each times `behind`)
    {
      fe tot = last ? coord(c, start_bit, 16, 2) : fe_mul(coord(c, start_bit, 16, 2), behind);
      ck[3] = tot;
#pragma unroll
      for (int i = 15; i >= 1; i--) {
        tot = fe_mul(coord(c, start_bit, i, 2), tot);
        if (i == 12) ck[2] = tot;
        if (i == 8) ck[1] = tot;
        if (i == 4) ck[0] = tot;
      }
    }
    auto affine = [&](int i, const fe& zinv, uint32_t w[16]) {
      fe_to_words(fe_mul(coord(c, start_bit, i, 0), zinv), w);
      fe_to_words(fe_mul(coord(c, start_bit, i, 1), zinv), w + 8);
    };
    auto emit_point = [&](int i, const fe& zinv) {  // points arrive in order: start, dbl_0, add_0, dbl_1, ...
      uint32_t w[16];
      affine(i, zinv, w);
      if (i == 0) {
#pragma unroll
        for (int qq = 0; qq < 16; qq++) accw[qq] = w[qq];
      } else if (i & 1) {
#pragma unroll
        for (int qq = 0; qq < 16; qq++) dblw[qq] = w[qq];
      } else {
        const int j = (i - 2) / 2;
        const uint32_t bit = scalar_bit(L.sc, (int)(CH * c) + j);
        stage[t][0] = bit & z;
#pragma unroll
        for (int qq = 0; qq < 16; qq++) {
          const uint32_t nx = bit ? w[qq] : dblw[qq];
          stage[t][1 + qq] = accw[qq] & z;
          stage[t][17 + qq] = dblw[qq] & z;
          stage[t][33 + qq] = w[qq] & z;
          stage[t][49 + qq] = nx & z;
          accw[qq] = nx;
        }
        coop_flush<TR_LADDER_ROW>(stage, s_base, s_live, (uint64_t)(CH * c + j) * TR_LADDER_ROW, out);
      }
    };
#pragma unroll
    for (int g = 0; g < 4; g++) {
      const fe z3 = coord(c, start_bit, 4 * g + 3, 2), z2 = coord(c, start_bit, 4 * g + 2, 2), z1 = coord(c, start_bit, 4 * g + 1, 2),
               z0 = coord(c, start_bit, 4 * g, 2);
      const fe s3 = fe_mul(z3, ck[g]), s2 = fe_mul(z2, s3), s1 = fe_mul(z1, s2);
      emit_point(4 * g, fe_mul(inv, s1)); inv = fe_mul(inv, z0);
      emit_point(4 * g + 1, fe_mul(inv, s2)); inv = fe_mul(inv, z1);
      emit_point(4 * g + 2, fe_mul(inv, s3)); inv = fe_mul(inv, z2);
      emit_point(4 * g + 3, fe_mul(inv, ck[g])); inv = fe_mul(inv, z3);
    }
    emit_point(16, last ? inv : fe_mul(inv, behind));
    if (!last) inv = fe_mul(inv, coord(c, start_bit, 16, 2));
  }
}

// SHA-512(R | A | M) of the effective triple of one lane (at most two blocks), 18 values per round
__global__ __launch_bounds__(64) void k_trace_sha512(uint32_t n_lanes, uint32_t n, const uint8_t* __restrict__ in_target, uint64_t* __restrict__ out,
                                                     uint64_t proof_stride) {
  constexpr int RPF = 4, NV = RPF * TR_SHA512_ROW;  // rounds per flush
  __shared__ uint32_t stage[64][NV + 1];
  __shared__ uint64_t s_base[64];
  __shared__ uint8_t s_live[64];
  const uint32_t t = threadIdx.x, id = blockIdx.x * 64u + t;
  const bool live = id < n_lanes;
  const uint32_t lane = live ? id : 0u;
  {
    const uint32_t p = lane / n, i = lane - p * n;
    s_base[t] = (uint64_t)p * proof_stride + (uint64_t)n * (2u * TR_LADDER_ROWS * TR_LADDER_ROW) + (uint64_t)i * (2u * 80u * TR_SHA512_ROW);
    s_live[t] = live ? 1 : 0;
  }
  const uint8_t* rec = in_target + (size_t)lane * VR_STRIDE;
  const bool is_signed = rec[VR_OFF_FLAGS] & 1;
  uint32_t mlen = is_signed ? (uint32_t)rec[VR_OFF_MLEN] | ((uint32_t)rec[VR_OFF_MLEN + 1] << 8) : 32u;
  if (mlen > 124u) mlen = 124u;
  const uint32_t total = 64u + mlen, nblk = total + 17u > 128u ? 2u : 1u;
  auto byte_at = [&](uint32_t pos) -> uint32_t {  // the padded message
    if (pos < 32u) return ((is_signed ? ld32(rec + VR_OFF_SIG + (pos & ~3u)) : T_DUMMY_SIG[pos >> 2]) >> (8u * (pos & 3u))) & 0xffu;
    if (pos < 64u) return ((is_signed ? ld32(rec + VR_OFF_PK + ((pos - 32u) & ~3u)) : T_DUMMY_PK[(pos - 32u) >> 2]) >> (8u * (pos & 3u))) & 0xffu;
    if (pos < total) return is_signed ? (uint32_t)rec[VR_OFF_MSG + (pos - 64u)] : 0u;
    return pos == total ? 0x80u : 0u;
  };
  uint64_t st[8];
  sha512_init(st);
#pragma unroll 1
  for (uint32_t blk = 0; blk < 2; blk++) {
    uint64_t w[16], v[8];
#pragma unroll 1
    for (int q = 0; q < 16; q++) {
      uint64_t word = 0;
      for (int b = 0; b < 8; b++) word = (word << 8) | byte_at(blk * 128u + 8u * q + b);
      w[q] = word;
    }
    if (blk == nblk - 1u) w[15] = (uint64_t)total * 8u;
#pragma unroll
    for (int q = 0; q < 8; q++) v[q] = st[q];
    const bool used = blk < nblk;
#pragma unroll 1
    for (int g = 0; g < 80 / RPF; g++) {
#pragma unroll
      for (int u = 0; u < RPF; u++) {
        const int i = g * RPF + u;
        uint64_t wt;
        {  // rolling schedule; the index is dynamic (the loop over g is not unrolled), so go through a select chain
          uint64_t w16 = w[0], w15 = w[0], w7 = w[0], w2 = w[0];
#pragma unroll
          for (int q = 0; q < 16; q++) {
            w16 = ((i) & 15) == q ? w[q] : w16; w15 = ((i + 1) & 15) == q ? w[q] : w15;
            w7 = ((i + 9) & 15) == q ? w[q] : w7; w2 = ((i + 14) & 15) == q ? w[q] : w2;
          }
          wt = w16;
          if (i >= 16) {
            wt = w16 + (rotr64(w15, 1) ^ rotr64(w15, 8) ^ (w15 >> 7)) + w7 + (rotr64(w2, 19) ^ rotr64(w2, 61) ^ (w2 >> 6));
#pragma unroll
            for (int q = 0; q < 16; q++) w[q] = (i & 15) == q ? wt : w[q];
          }
        }
        const uint64_t t1 = v[7] + (rotr64(v[4], 14) ^ rotr64(v[4], 18) ^ rotr64(v[4], 41)) + ((v[4] & v[5]) ^ (~v[4] & v[6])) + K_SHA512[i] + wt;
        const uint64_t t2 = (rotr64(v[0], 28) ^ rotr64(v[0], 34) ^ rotr64(v[0], 39)) + ((v[0] & v[1]) ^ (v[0] & v[2]) ^ (v[1] & v[2]));
        v[7] = v[6]; v[6] = v[5]; v[5] = v[4]; v[4] = v[3] + t1; v[3] = v[2]; v[2] = v[1]; v[1] = v[0]; v[0] = t1 + t2;
        uint32_t* o = stage[t] + u * TR_SHA512_ROW;
        o[0] = used ? (uint32_t)wt : 0u; o[1] = used ? (uint32_t)(wt >> 32) : 0u;
#pragma unroll
        for (int q = 0; q < 8; q++) { o[2 + 2 * q] = used ? (uint32_t)v[q] : 0u; o[3 + 2 * q] = used ? (uint32_t)(v[q] >> 32) : 0u; }
      }
      coop_flush<NV>(stage, s_base, s_live, (uint64_t)(blk * 80u + g * RPF) * TR_SHA512_ROW, out);
    }
#pragma unroll
    for (int q = 0; q < 8; q++) st[q] += v[q];
  }
}

// SHA-256(00 | marshalled validator[0 .. vlen]) of one lane of the target or trusted set: one block, 9 values per round
__global__ __launch_bounds__(64) void k_trace_sha256(uint32_t kind, uint32_t n_lanes, uint32_t n, const uint8_t* __restrict__ in_target,
                                                     const uint8_t* __restrict__ in_trusted, uint64_t* __restrict__ out, uint64_t proof_stride) {
  constexpr int RPF = 4, NV = RPF * TR_SHA256_ROW;
  __shared__ uint32_t stage[64][NV + 1];
  __shared__ uint64_t s_base[64];
  __shared__ uint8_t s_live[64];
  const uint32_t t = threadIdx.x, id = blockIdx.x * 64u + t, sets = kind == 0 ? 2u : 1u;
  const bool live = id < sets * n_lanes;
  const uint32_t set = live && id >= n_lanes ? 1u : 0u, lane = live ? id - set * n_lanes : 0u;
  {
    const uint32_t p = lane / n, i = lane - p * n;
    s_base[t] = (uint64_t)p * proof_stride + (uint64_t)n * (2u * TR_LADDER_ROWS * TR_LADDER_ROW + 2u * 80u * TR_SHA512_ROW) +
                (uint64_t)(set * n + i) * (64u * TR_SHA256_ROW);
    s_live[t] = live ? 1 : 0;
  }
  const uint8_t* rec = set ? in_trusted + (size_t)lane * HR_STRIDE : in_target + (size_t)lane * VR_STRIDE;
  const uint64_t power = ld64(rec + (set ? HR_OFF_POWER : VR_OFF_POWER));
  uint32_t vlen = rec[set ? HR_OFF_VLEN : VR_OFF_VLEN];
  if (vlen > 46u) vlen = 46u;
  int last = 0;
#pragma unroll
  for (int s = 0; s < 9; s++) if ((power >> (7 * s)) & 0x7f) last = s;
  const uint32_t len = 1u + vlen;
  auto byte_at = [&](uint32_t pos) -> uint32_t {  // 00 | 0a 22 0a 20 | pk | 10 | varint9, cut at len, padded
    if (pos >= len) return pos == len ? 0x80u : 0u;
    if (pos == 0u) return 0u;
    const uint32_t m = pos - 1u;
    if (m < 4u) return (m & 1u) ? (m == 1u ? 0x22u : 0x20u) : 0x0au;
    if (m < 36u) return (ld32(rec + ((m - 4u) & ~3u)) >> (8u * ((m - 4u) & 3u))) & 0xffu;
    if (m == 36u) return 0x10u;
    const uint32_t s = m - 37u;
    return (uint32_t)((power >> (7u * s)) & 0x7fu) | ((int)s < last ? 0x80u : 0u);
  };
  uint32_t w[16], v[8];
#pragma unroll 1
  for (int q = 0; q < 16; q++) {
    uint32_t word = 0;
    for (int b = 0; b < 4; b++) word = (word << 8) | byte_at(4u * q + b);
    w[q] = word;
  }
  w[15] = len * 8u;
  sha256_init(v);
#pragma unroll 1
  for (int g = 0; g < 64 / RPF; g++) {
#pragma unroll
    for (int u = 0; u < RPF; u++) {
      const int i = g * RPF + u;
      uint32_t w16 = w[0], w15 = w[0], w7 = w[0], w2 = w[0];
#pragma unroll
      for (int q = 0; q < 16; q++) {
        w16 = ((i) & 15) == q ? w[q] : w16; w15 = ((i + 1) & 15) == q ? w[q] : w15;
        w7 = ((i + 9) & 15) == q ? w[q] : w7; w2 = ((i + 14) & 15) == q ? w[q] : w2;
      }
      uint32_t wt = w16;
      if (i >= 16) {
        wt = w16 + (rotr32(w15, 7) ^ rotr32(w15, 18) ^ (w15 >> 3)) + w7 + (rotr32(w2, 17) ^ rotr32(w2, 19) ^ (w2 >> 10));
#pragma unroll
        for (int q = 0; q < 16; q++) w[q] = (i & 15) == q ? wt : w[q];
      }
      const uint32_t t1 = v[7] + (rotr32(v[4], 6) ^ rotr32(v[4], 11) ^ rotr32(v[4], 25)) + ((v[4] & v[5]) ^ (~v[4] & v[6])) + K_SHA256[i] + wt;
      const uint32_t t2 = (rotr32(v[0], 2) ^ rotr32(v[0], 13) ^ rotr32(v[0], 22)) + ((v[0] & v[1]) ^ (v[0] & v[2]) ^ (v[1] & v[2]));
      v[7] = v[6]; v[6] = v[5]; v[5] = v[4]; v[4] = v[3] + t1; v[3] = v[2]; v[2] = v[1]; v[1] = v[0]; v[0] = t1 + t2;
      uint32_t* o = stage[t] + u * TR_SHA256_ROW;
      o[0] = wt;
#pragma unroll
      for (int q = 0; q < 8; q++) o[1 + q] = v[q];
    }
    coop_flush<NV>(stage, s_base, s_live, (uint64_t)(g * RPF) * TR_SHA256_ROW, out);
  }
}

// T.5 / T.6: SHA-256 of a message of at most two blocks, one thread per hash, over values Level-1 holds in the context's scratch
//   T.5  item (set, node slot): 01 | L | R over the two children of the slot in the fixed-shape validator tree -- leaf hashes below the
//        first level, Level-1 nodes above -- hashed for every pair whether or not both are enabled (the circuit selects afterwards,
//        reference validator.rs:248-251); a promoted slot has zero rows
//   T.6  item (proof q, hash h): h = 0 the leaf hash 00 | leaf (as the proof struct carries it), h = 1..4 the path nodes 01 | left | right
//        (verify.rs:189-209, shared.rs:183-203, tendermint_utils.rs:214-224)
// Items of a proof are contiguous in the output: T.5 (sets x tree_nodes), then T.6 (4 or 5 proofs x 5), 1152 elements each.
constexpr uint32_t TR_SHA256_2 = 2u * 64u * TR_SHA256_ROW;
__global__ __launch_bounds__(64) void k_trace_sha256x2(uint32_t kind, uint32_t n_proofs, uint32_t n, uint32_t tn, uint32_t sections,
                                                       const uint8_t* __restrict__ lt, uint32_t lt_stride, const uint8_t* __restrict__ lr,
                                                       const uint8_t* __restrict__ nodes_t, const uint8_t* __restrict__ nodes_r,
                                                       const uint8_t* __restrict__ pf, uint64_t* __restrict__ out, uint64_t proof_stride, uint64_t sec_off) {
  constexpr int RPF = 8, NV = RPF * TR_SHA256_ROW;
  __shared__ uint32_t stage[64][NV + 1];
  __shared__ uint32_t msg[64][33];
  __shared__ uint64_t s_base[64];
  __shared__ uint8_t s_live[64];
  const uint32_t t = threadIdx.x, id = blockIdx.x * 64u + t, sets = kind == 0 ? 2u : 1u, nq = kind == 0 ? 4u : 5u;
  const uint32_t per_proof = sets * tn + nq * 5u;
  const uint32_t p = id / per_proof, it = id - p * per_proof;
  const bool tree_item = it < sets * tn;
  const bool live = p < n_proofs && (sections & (tree_item ? 16u : 32u));
  s_base[t] = (uint64_t)p * proof_stride + sec_off + (uint64_t)it * TR_SHA256_2;
  s_live[t] = live ? 1 : 0;
  uint8_t* mb = reinterpret_cast<uint8_t*>(msg[t]);
#pragma unroll 1
  for (int q = 0; q < 32; q++) msg[t][q] = 0u;
  uint32_t len = 0;  // 0: zero rows
  if (live && tree_item) {
    const uint32_t set = it >= tn ? 1u : 0u;
    uint32_t rem = it - set * tn, sz = n, level = 0, first = 0, prev_first = 0;  // first: slot of the level's first node
    for (;;) {
      const uint32_t nx = (sz + 1u) / 2u;
      if (rem < nx) break;
      rem -= nx; prev_first = first; first += nx; sz = nx; level++;
    }
    if (2u * rem + 1u < sz) {
      const uint8_t* l;
      uint32_t step;
      if (level == 0) {
        step = set ? LANE_STRIDE : lt_stride;
        l = (set ? lr : lt) + (size_t)(p * n + 2u * rem) * step + LN_OFF_LEAF;
      } else {
        step = 32u;
        l = (set ? nodes_r : nodes_t) + ((size_t)p * tn + prev_first + 2u * rem) * 32u;
      }
      mb[0] = 0x01;
      for (uint32_t b = 0; b < 32u; b++) { mb[1u + b] = l[b]; mb[33u + b] = l[step + b]; }
      len = 65u;
    }
  } else if (live) {
    const uint32_t j = it - sets * tn, q = j / 5u, h = j - q * 5u;
    const uint8_t* r = pf + (size_t)p * PF_STRIDE;
    if (h == 0) {
      if (q == 0) {
        for (uint32_t b = 0; b < 52u; b++) mb[1u + b] = r[PF_OFF_CID52 + b];
        len = min(ld32(r + PF_OFF_CIDLEN), 79u) + 1u;
      } else if (q == 1) {
        for (uint32_t b = 0; b < 11u; b++) mb[b] = r[PF_OFF_HLEAF + b];
        len = min(ld32(r + PF_OFF_HLEN), 79u) + 1u;
      } else {
        const uint32_t off = q == 2 ? PF_OFF_LEAFV : (q == 3 ? PF_OFF_LEAFX : PF_OFF_LEAFY), w = (kind == 1 && q == 3) ? 72u : 34u;
        for (uint32_t b = 0; b < w; b++) mb[1u + b] = r[off + b];
        len = w + 1u;
      }
    } else {
      const uint32_t k = h - 1u, index = q == 0 ? 1u : (q == 1 ? 2u : (q == 2 ? 7u : (q == 3 ? (kind == 0 ? 7u : 4u) : 8u)));
      const uint8_t* cur = r + PF_OFF_PROOFD + q * 160u + k * 32u;
      const uint8_t* aunt = r + PF_OFF_AUNTS + q * 128u + k * 32u;
      const bool right = (index >> k) & 1u;  // the running hash is the right child
      mb[0] = 0x01;
      for (uint32_t b = 0; b < 32u; b++) { mb[1u + b] = right ? aunt[b] : cur[b]; mb[33u + b] = right ? cur[b] : aunt[b]; }
      len = 65u;
    }
  }
  const uint32_t n_blocks = len == 0 ? 0u : (len + 9u > 64u ? 2u : 1u);
  if (len) {
    mb[len] = 0x80;
    msg[t][16u * n_blocks - 1u] = __builtin_bswap32(len * 8u);
  }
  uint32_t st[8];
  sha256_init(st);
#pragma unroll 1
  for (uint32_t blk = 0; blk < 2u; blk++) {
    const bool used = blk < n_blocks;
    uint32_t w[16], v[8];
#pragma unroll
    for (int q = 0; q < 16; q++) w[q] = __builtin_bswap32(msg[t][16u * blk + q]);
#pragma unroll
    for (int q = 0; q < 8; q++) v[q] = st[q];
#pragma unroll 1
    for (int g = 0; g < 64 / RPF; g++) {
#pragma unroll
      for (int u = 0; u < RPF; u++) {
        const int i = g * RPF + u;
        uint32_t w16 = w[0], w15 = w[0], w7 = w[0], w2 = w[0];
#pragma unroll
        for (int q = 0; q < 16; q++) {
          w16 = ((i) & 15) == q ? w[q] : w16; w15 = ((i + 1) & 15) == q ? w[q] : w15;
          w7 = ((i + 9) & 15) == q ? w[q] : w7; w2 = ((i + 14) & 15) == q ? w[q] : w2;
        }
        uint32_t wt = w16;
        if (i >= 16) {
          wt = w16 + (rotr32(w15, 7) ^ rotr32(w15, 18) ^ (w15 >> 3)) + w7 + (rotr32(w2, 17) ^ rotr32(w2, 19) ^ (w2 >> 10));
#pragma unroll
          for (int q = 0; q < 16; q++) w[q] = (i & 15) == q ? wt : w[q];
        }
        const uint32_t t1 = v[7] + (rotr32(v[4], 6) ^ rotr32(v[4], 11) ^ rotr32(v[4], 25)) + ((v[4] & v[5]) ^ (~v[4] & v[6])) + K_SHA256[i] + wt;
        const uint32_t t2 = (rotr32(v[0], 2) ^ rotr32(v[0], 13) ^ rotr32(v[0], 22)) + ((v[0] & v[1]) ^ (v[0] & v[2]) ^ (v[1] & v[2]));
        v[7] = v[6]; v[6] = v[5]; v[5] = v[4]; v[4] = v[3] + t1; v[3] = v[2]; v[2] = v[1]; v[1] = v[0]; v[0] = t1 + t2;
        uint32_t* o = stage[t] + u * TR_SHA256_ROW;
        o[0] = used ? wt : 0u;
#pragma unroll
        for (int q = 0; q < 8; q++) o[1 + q] = used ? v[q] : 0u;
      }
      coop_flush<NV>(stage, s_base, s_live, (uint64_t)(blk * 64u + g * RPF) * TR_SHA256_ROW, out);
    }
#pragma unroll
    for (int q = 0; q < 8; q++) st[q] += v[q];
  }
}

// skip: m[i][j] = signed[i] and target pubkey i == trusted pubkey j (verify.rs:398-418, every pair)
__global__ __launch_bounds__(256) void k_trace_match(uint32_t n_proofs, uint32_t n, const uint8_t* __restrict__ in_target, const uint8_t* __restrict__ in_trusted,
                                                     uint64_t* __restrict__ out, uint64_t proof_stride, uint64_t sec_off) {
  const uint32_t p = blockIdx.y, e = blockIdx.x * 256u + threadIdx.x;
  if (p >= n_proofs || e >= n * n) return;
  const uint32_t i = e / n, j = e - i * n;
  const uint8_t* a = in_target + ((size_t)p * n + i) * VR_STRIDE;
  const uint8_t* b = in_trusted + ((size_t)p * n + j) * HR_STRIDE;
  uint32_t d = 0;
#pragma unroll
  for (int w = 0; w < 8; w++) d |= ld32(a + 4 * w) ^ ld32(b + 4 * w);
  out[(size_t)p * proof_stride + sec_off + e] = (d == 0 && (a[VR_OFF_FLAGS] & 1)) ? 1ull : 0ull;
}

}  // namespace

static inline hipStream_t S_(void* s) { return reinterpret_cast<hipStream_t>(s); }

static uint32_t tree_slots(uint32_t n) {  // nodes of the fixed-shape tree, every level (promoted odd nodes included)
  uint32_t c = 0;
  while (n > 1) { n = (n + 1) / 2; c += n; }
  return c;
}
static uint64_t trace_off_tree(uint32_t kind, uint32_t n) {
  const uint64_t sets = kind == 0 ? 2 : 1;
  return (uint64_t)n * (2ull * TR_LADDER_ROWS * TR_LADDER_ROW + 2ull * 80 * TR_SHA512_ROW + sets * 64 * TR_SHA256_ROW) + (kind == 0 ? (uint64_t)n * n : 0);
}
// one section of a proof's trace block as a row-major matrix: offset, rows, row width (elements).  The N x N match bits have no row
// structure of their own (false: not a committed table).
bool trace_section_geom(uint32_t kind, uint32_t n, uint32_t section, uint64_t* off, uint64_t* rows, uint32_t* width) {
  const uint64_t sets = kind == 0 ? 2 : 1, tn = tree_slots(n);
  const uint64_t o_sha512 = (uint64_t)n * 2ull * TR_LADDER_ROWS * TR_LADDER_ROW, o_sha256 = o_sha512 + (uint64_t)n * 2ull * 80 * TR_SHA512_ROW;
  switch (section) {
    case 1u: *off = 0; *rows = 2ull * n * TR_LADDER_ROWS; *width = TR_LADDER_ROW; return true;
    case 2u: *off = o_sha512; *rows = 2ull * n * 80; *width = TR_SHA512_ROW; return true;
    case 4u: *off = o_sha256; *rows = sets * n * 64; *width = TR_SHA256_ROW; return true;
    case 16u: *off = trace_off_tree(kind, n); *rows = sets * tn * 128; *width = TR_SHA256_ROW; return tn != 0;
    case 32u: *off = trace_off_tree(kind, n) + sets * tn * TR_SHA256_2; *rows = (kind == 0 ? 4ull : 5ull) * 5 * 128; *width = TR_SHA256_ROW; return true;
    default: return false;
  }
}

// Row-major rows of one section of every proof -> column-major columns for the LDE: column (p, c) = element c of every row of proof p,
// zero-padded to 2^log_n rows, at cols[(p * width + c) << log_n].  A workgroup moves 64 rows: one coalesced read of 64 * width
// consecutive elements, staged in LDS, width runs of 64 consecutive elements (512 B) out.
__global__ __launch_bounds__(256) void k_trace_to_columns(const uint64_t* __restrict__ trace, uint64_t proof_stride, uint64_t sec_off, uint64_t rows,
                                                          uint32_t width, uint32_t log_n, uint64_t* __restrict__ cols) {
  extern __shared__ uint64_t s_tile[];  // [width][65] (one pad word per column: conflict-free column reads)
  const uint32_t p = blockIdx.y, t = threadIdx.x;
  const uint64_t r0 = (uint64_t)blockIdx.x * 64;
  const uint64_t* src = trace + (size_t)p * proof_stride + sec_off + r0 * width;
  const uint32_t n_el = 64 * width;
  for (uint32_t e = t; e < n_el; e += 256) {
    const uint32_t r = e / width, c = e - r * width;
    s_tile[c * 65 + r] = r0 + r < rows ? src[e] : 0ull;
  }
  __syncthreads();
  uint64_t* dst = cols + (((size_t)p * width) << log_n) + r0;
  for (uint32_t e = t; e < n_el; e += 256) {
    const uint32_t c = e >> 6, r = e & 63u;
    dst[((size_t)c << log_n) + r] = s_tile[c * 65 + r];
  }
}
int launch_trace_to_columns(const void* d_trace, uint64_t proof_stride, uint64_t sec_off, uint64_t rows, uint32_t width, uint32_t log_n, uint32_t n_proofs,
                            void* d_cols, void* stream) {
  if (n_proofs == 0) return 0;
  hipLaunchKernelGGL(k_trace_to_columns, dim3((uint32_t)(((uint64_t)1 << log_n) / 64), n_proofs), dim3(256), (size_t)width * 65 * 8, S_(stream),
                     reinterpret_cast<const uint64_t*>(d_trace), proof_stride, sec_off, rows, width, log_n, reinterpret_cast<uint64_t*>(d_cols));
  return (int)hipGetLastError();
}

uint64_t trace_elems(uint32_t kind, uint32_t n) {
  return trace_off_tree(kind, n) + ((kind == 0 ? 2ull : 1ull) * tree_slots(n) + (kind == 0 ? 4ull : 5ull) * 5) * TR_SHA256_2;
}

size_t trace_tmp_bytes(uint32_t n, uint32_t n_proofs) { return (size_t)((2ull * n_proofs * n + 63) / 64) * 64 * TR_LADDER_ROWS * TR_PT_WORDS * 4; }

// rows [row0, row1) of every ladder: the chain (pass 1) and, once it is done, the affine rows (pass 2); row0 / row1 multiples of eight
int launch_trace_ladder_pass1(uint32_t n, uint32_t n_proofs, const void* d_target, const void* d_ed, uint32_t ed_stride, void* d_tmp, uint32_t row0,
                              uint32_t row1, void* stream) {
  if (n_proofs == 0) return 0;
  const uint32_t lanes = n_proofs * n;
  hipLaunchKernelGGL(k_trace_ladder_pass1, dim3((2 * lanes + 63) / 64), dim3(64), 0, S_(stream), lanes, reinterpret_cast<const uint8_t*>(d_target),
                     reinterpret_cast<const uint8_t*>(d_ed), ed_stride, reinterpret_cast<int32_t*>(d_tmp), row0, row1);
  return (int)hipGetLastError();
}
int launch_trace_ladder_pass2(uint32_t kind, uint32_t n, uint32_t n_proofs, const void* d_target, const void* d_ed, uint32_t ed_stride, const void* d_tmp,
                              void* d_out, uint32_t row0, uint32_t row1, void* stream) {
  if (n_proofs == 0) return 0;
  const uint32_t lanes = n_proofs * n, chunks = (row1 - row0) / CH;
  // batches per inversion.  Measured per 256-proof batch (ladders): 1 -> 6.80 ms, 2 -> 6.58, 4 -> 8.25, 8 -> 7.47: the inversion is 35 % of the
  // pass's instructions, but the pass is latency-bound at two waves per SIMD -- a thread that does four batches in a row loses more to the
  // longer chain and the thinner launch than it saves.  TMX_TRACE_GCH overrides.
#define TMX_TRACE_P2(G)                                                                                                                        \
  hipLaunchKernelGGL((k_trace_ladder_pass2<G>), dim3((2 * lanes + 63) / 64, chunks / G), dim3(64), 0, S_(stream), lanes, n,                       \
                     reinterpret_cast<const uint8_t*>(d_target), reinterpret_cast<const uint8_t*>(d_ed), ed_stride,                             \
                     reinterpret_cast<const int32_t*>(d_tmp), reinterpret_cast<uint64_t*>(d_out), trace_elems(kind, n), row0 / CH)
  static const int g_env = std::getenv("TMX_TRACE_GCH") ? std::atoi(std::getenv("TMX_TRACE_GCH")) : 2;
  if (g_env >= 8 && chunks % 8 == 0) TMX_TRACE_P2(8);
  else if (g_env >= 4 && chunks % 4 == 0) TMX_TRACE_P2(4);
  else if (g_env >= 2 && chunks % 2 == 0) TMX_TRACE_P2(2);
  else TMX_TRACE_P2(1);
#undef TMX_TRACE_P2
  return (int)hipGetLastError();
}

// the other sections (bits 1, 2, 3 of `sections`)
int launch_trace_rest(uint32_t kind, uint32_t n, uint32_t n_proofs, const void* d_target, const void* d_trusted, const TraceLevel1& L1, void* d_out,
                      uint32_t sections, void* stream) {
  if (n_proofs == 0) return 0;
  const uint32_t lanes = n_proofs * n;
  const uint64_t stride = trace_elems(kind, n);
  uint64_t* out = reinterpret_cast<uint64_t*>(d_out);
  const uint8_t* tg = reinterpret_cast<const uint8_t*>(d_target);
  const uint8_t* tr = reinterpret_cast<const uint8_t*>(d_trusted);
  if (sections & 2u) hipLaunchKernelGGL(k_trace_sha512, dim3((lanes + 63) / 64), dim3(64), 0, S_(stream), lanes, n, tg, out, stride);
  if (sections & 4u)
    hipLaunchKernelGGL(k_trace_sha256, dim3(((kind == 0 ? 2 : 1) * lanes + 63) / 64), dim3(64), 0, S_(stream), kind, lanes, n, tg, tr, out, stride);
  if ((sections & 8u) && kind == 0) {
    const uint64_t off = (uint64_t)n * (2ull * TR_LADDER_ROWS * TR_LADDER_ROW + 2ull * 80 * TR_SHA512_ROW + 2ull * 64 * TR_SHA256_ROW);
    hipLaunchKernelGGL(k_trace_match, dim3((n * n + 255) / 256, n_proofs), dim3(256), 0, S_(stream), n_proofs, n, tg, tr, out, stride, off);
  }
  if (sections & (16u | 32u)) {
    const uint32_t tn = tree_slots(n), items = n_proofs * ((kind == 0 ? 2u : 1u) * tn + (kind == 0 ? 4u : 5u) * 5u);
    hipLaunchKernelGGL(k_trace_sha256x2, dim3((items + 63) / 64), dim3(64), 0, S_(stream), kind, n_proofs, n, tn, sections,
                       reinterpret_cast<const uint8_t*>(L1.d_lt), L1.lt_stride, reinterpret_cast<const uint8_t*>(L1.d_lr),
                       reinterpret_cast<const uint8_t*>(L1.d_nodes_t), reinterpret_cast<const uint8_t*>(L1.d_nodes_r), reinterpret_cast<const uint8_t*>(L1.d_pf), out,
                       stride, trace_off_tree(kind, n));
  }
  return (int)hipGetLastError();
}

}  // namespace tmx
