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
#include "winv25519.hpp"
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


// flush NV (even) staged values of every thread's current row group: value e of thread j goes to out[base[j] + off + e].  One thread's group
// per store instruction, 16 bytes per lane (lanes 0 .. NV / 2 - 1; a group starts at any multiple of 8 bytes: wide global stores need dword
// alignment only); the group's address and whether it exists come from the owning lane's registers as SGPRs (v_readlane with a constant
// lane) -- the first form read both from LDS in front of every store and needed two store instructions per group of 72.
typedef uint64_t tr_u64x2 __attribute__((ext_vector_type(2)));
typedef tr_u64x2 __attribute__((aligned(8))) tr_u64x2_a8;
template <int NV>
__device__ __forceinline__ void coop_flush(const uint32_t (*stage)[NV + 1], const uint64_t* s_base, const uint8_t* s_live, uint64_t off, uint64_t* __restrict__ out) {
  static_assert(NV % 2 == 0 && NV <= 128, "two values per lane");
  __syncthreads();
  const uint32_t tid = threadIdx.x;
  const uint64_t my_dst = s_base[tid] + off;
  const uint32_t dst_lo = (uint32_t)my_dst, dst_hi = (uint32_t)(my_dst >> 32), live = s_live[tid];
  const uint32_t e = 2u * tid < (uint32_t)NV ? 2u * tid : 0u;  // (lanes beyond the group read its first pair and store nothing)
#pragma unroll 8
  for (int j = 0; j < 64; j++) {
    if (!__builtin_amdgcn_readlane((int)live, j)) continue;
    uint64_t* dst = out + ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)dst_lo, j) | ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)dst_hi, j) << 32));
    tr_u64x2 v;
    v.x = stage[j][e]; v.y = stage[j][e + 1];
    if (2u * tid < (uint32_t)NV) __builtin_nontemporal_store(v, reinterpret_cast<tr_u64x2_a8*>(dst + e));
  }
  __syncthreads();
}

// ---- ladders, in two passes so that the expensive half (inversions, canonical limbs, the stores) is not a 256-step chain:
//   pass 1  one thread per (lane, ladder): the double-and-add chain in extended coordinates.  (X:Y:Z) of dbl_r and add_r of every row go to
//           a scratch buffer together with the PREFIX PRODUCT of all Z's up to that point (P_2r = Z(dbl_0) Z(add_0) ... Z(dbl_r),
//           P_2r+1 = P_2r Z(add_r): two more products per row, off the point chain); P_2r+1 is stored with them -- 280 B per row, every
//           store / load instruction a contiguous run of 16 bytes per lane (layout below).
//   pass 2  one thread per (ladder, R rows), a wave = the same 64 ladders as in pass 1: the inverse of P at the thread's last point (ONE
//           inversion per wave: winv25519.hpp), then Montgomery's trick walked BACKWARDS -- 1 / Z_k = inv(P_k) P_k-1, inv(P_k-1) = inv(P_k) Z_k: two products per
//           point and nothing recomputed (the first version swept the Z's of its rows three times for the suffix products it could not keep:
//           14 products per row against 8) --, the affine words staged in LDS and written by the whole wave.
// What a thread writes per row is the 65 CONTIGUOUS elements  dbl_r | add_r | nxt_r | bit_r+1 | acc_r+1  (acc_r+1 = nxt_r: elements 17 ... 64
// of row r and 0 ... 16 of row r + 1), so that no thread needs the affine form of the point its rows start from; the thread with row 0 adds
// the ladder's first 17 elements (bit_0, the identity), the one with row 255 stops after nxt.
// Measured per 256-proof batch at N = 128 (8.7 GB of rows): one pass 8.3 ms; two passes with suffix products 6.1 ms (round 3); this form
// 4.95 ms (round 4; 4.6 ms with the wave-shared inversion and without P(dbl) in the scratch rows, round 5) -- pass 2 alone 0.52 ms per 64-row segment without its stores, 0.8 - 0.9 with them: what is left is the write stream of 520-byte
// pieces 130 KB apart (DESIGN.md "The writer, measured").
// (the scalar as a VECTOR: a wave-uniform dynamic index into a vector is a register select; into an array it became a scratch load per row,
// and every scratch load waits for ALL memory operations of the wave -- vmcnt counts in order -- i.e. for the stores of the row before)
typedef uint32_t tr_u32x8 __attribute__((ext_vector_type(8)));
struct LadderIn {
  tr_u32x8 sc;  // the scalar
  ge_affc P;    // the point that is added
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
__device__ __forceinline__ uint32_t scalar_bit(const tr_u32x8& sc, int r) {  // bit 255 - r
  const int b = 255 - r;
  return (sc[(b >> 5) & 7] >> (b & 31)) & 1u;
}
// The scratch buffer: per block of 64 ladders and per row SEVEN field elements -- dbl (X, Y, Z), add (X, Y, Z), P(add) --, each as
// 640 words [part][thread][words of the part] with parts of 4, 4 and 2 limbs: a wave moves one element with three instructions (16, 16 and
// 8 bytes per lane, every one of them a contiguous run: 21 loads or stores per row instead of 70 of one word -- a wave may have 63 in flight).
// (P(dbl_r) = P(add_r-1) Z(dbl_r) is one product of two values pass 2 loads anyway: round 4 stored it too -- 320 B per row, 12.5 % more
// scratch traffic in both directions.)
constexpr uint32_t TR_FE_WORDS = 640, TR_PT_FES = 7, TR_ROW_WORDS = TR_FE_WORDS * TR_PT_FES;
enum : uint32_t { TR_DX = 0, TR_DY = 1, TR_DZ = 2, TR_AX = 3, TR_AY = 4, TR_AZ = 5, TR_PA = 6 };
typedef int32_t tr_i32x4 __attribute__((ext_vector_type(4)));
typedef int32_t tr_i32x2 __attribute__((ext_vector_type(2)));
// `blk` = the block's part of the buffer; t = the thread
__device__ __forceinline__ const int32_t* pt_at(const int32_t* __restrict__ blk, int row, uint32_t which) {
  return blk + ((size_t)row * TR_PT_FES + which) * TR_FE_WORDS;
}
__device__ __forceinline__ fe pt_load(const int32_t* __restrict__ p, uint32_t t) {
  const tr_i32x4 a = *reinterpret_cast<const tr_i32x4*>(p + 4 * t), b = *reinterpret_cast<const tr_i32x4*>(p + 256 + 4 * t);
  const tr_i32x2 c = *reinterpret_cast<const tr_i32x2*>(p + 512 + 2 * t);
  fe v;
  v.v[0] = a.x; v.v[1] = a.y; v.v[2] = a.z; v.v[3] = a.w; v.v[4] = b.x; v.v[5] = b.y; v.v[6] = b.z; v.v[7] = b.w; v.v[8] = c.x; v.v[9] = c.y;
  return v;
}
__device__ __forceinline__ void pt_store(int32_t* __restrict__ p, uint32_t t, const fe& v) {
  tr_i32x4 a, b;
  tr_i32x2 c;
  a.x = v.v[0]; a.y = v.v[1]; a.z = v.v[2]; a.w = v.v[3]; b.x = v.v[4]; b.y = v.v[5]; b.z = v.v[6]; b.w = v.v[7]; c.x = v.v[8]; c.y = v.v[9];
  *reinterpret_cast<tr_i32x4*>(p + 4 * t) = a;
  *reinterpret_cast<tr_i32x4*>(p + 256 + 4 * t) = b;
  *reinterpret_cast<tr_i32x2*>(p + 512 + 2 * t) = c;
}

__global__ __launch_bounds__(64) void k_trace_ladder_pass1(uint32_t lane0, uint32_t n_lanes, const uint8_t* __restrict__ in_target, const uint8_t* __restrict__ ed,
                                                           uint32_t ed_stride, int32_t* __restrict__ pts, uint32_t row0, uint32_t row1) {
  // (lanes [lane0, lane0 + n_lanes) of the batch: a rank of a validator-sharded launch traces only its own -- api.cpp tmx_trace_rows_validator_sharded_device)
  const uint32_t t = threadIdx.x, id = blockIdx.x * 64u + t;
  const bool live = id < 2u * n_lanes;
  const LadderIn L = ladder_inputs(lane0 + (live ? id >> 1 : 0u), id & 1u, in_target, ed, ed_stride, true);
  int32_t* blk = pts + (size_t)blockIdx.x * TR_LADDER_ROWS * TR_ROW_WORDS;
  ge_proj acc = ext_to_proj(ge_identity());
  fe prod = fe_one();
  __builtin_amdgcn_s_setprio(3);  // the chain is the latency of the whole section; pass 2 of the previous segment runs beside it as throughput work
  if (row0) {  // a later segment of the chain: the accumulator is row0 - 1's nxt, the running product its P(add) -- both in the scratch buffer
    const uint32_t w = scalar_bit(L.sc, (int)row0 - 1) ? TR_AX : TR_DX;
    acc.X = pt_load(pt_at(blk, (int)row0 - 1, w), t); acc.Y = pt_load(pt_at(blk, (int)row0 - 1, w + 1), t); acc.Z = pt_load(pt_at(blk, (int)row0 - 1, w + 2), t);
    prod = pt_load(pt_at(blk, (int)row0 - 1, TR_PA), t);
  }
#pragma unroll 1
  for (int r = (int)row0; r < (int)row1; r++) {
    const uint32_t bit = scalar_bit(L.sc, r);
    const ge_ext d = comp_to_ext(ge_double(acc));
    const ge_proj a = comp_to_proj(ge_add_affc(d, L.P));  // (the sum's T is never needed: the next step doubles from (X : Y : Z))
    const fe pd = fe_mul(prod, d.Z);
    prod = fe_mul(pd, a.Z);
    int32_t* row = blk + (size_t)r * TR_ROW_WORDS;
    pt_store(row + TR_DX * TR_FE_WORDS, t, d.X); pt_store(row + TR_DY * TR_FE_WORDS, t, d.Y); pt_store(row + TR_DZ * TR_FE_WORDS, t, d.Z);
    pt_store(row + TR_AX * TR_FE_WORDS, t, a.X); pt_store(row + TR_AY * TR_FE_WORDS, t, a.Y); pt_store(row + TR_AZ * TR_FE_WORDS, t, a.Z);
    pt_store(row + TR_PA * TR_FE_WORDS, t, prod);
    acc.X = fe_select(d.X, a.X, bit); acc.Y = fe_select(d.Y, a.Y, bit); acc.Z = fe_select(d.Z, a.Z, bit);
  }
}

// The stores of pass 2.  Row r of a ladder yields the 65 contiguous elements  dbl_r | add_r | nxt_r | bit_r+1 | acc_r+1  (its "unit", at element
// U = 65 r + 17 of the ladder: nxt = acc of the next row = bit_r ? add : dbl).  520 bytes at an arbitrary multiple of 8: written unit by unit,
// every unit left two partially written 64-byte lines for the unit beside it to complete tens of microseconds later -- measured, the same
// kernel with 512-byte aligned pseudo-rows ran 1.4x faster (0.71 vs 1.01 ms per segment; without any store 0.52).  So a flush writes WHOLE
// LINES: from the first line boundary in the unit to the first line boundary behind it, i.e. without the unit's first d = -U mod 8 elements
// (the row below writes them) and with the first c elements of the unit above (its dbl words, kept from the previous flush): 64 elements,
// 72 when the unit itself starts on a boundary.  Only the first and the last unit of a thread's rows end inside a line.
//   staged words of a thread: [0, 16) dbl, [16, 32) add, [32] bit_r+1, [33, 40) / [40, 47) the first seven dbl words of the odd / even rows
//   (a row leaves its own there while the flush reads those of the row above); stride 47 (odd)
//   element k of the span -> staged word through a 4 x 72-byte table in LDS (by bit_r: nxt and acc are add or dbl; by the parity of r)
// One ladder per store instruction: 64 lanes x 8 bytes; what belongs to the ladder (address, bit, bounds of its span) is read from the owning
// lane's registers into SGPRs (v_readlane with a constant lane) -- the address is scalar base + lane offset, nothing waits for LDS but the data.
constexpr uint32_t TR_STAGE = 47, TR_ST_BIT = 32, TR_ST_CARRY = 33, TR_SPAN_MAX = 72;
// element k of (unit of row r | carried head of the unit of row r + 1) -> staged word;  sel = bit_r | (r & 1) << 1
__device__ __forceinline__ uint32_t span_word(uint32_t k, uint32_t sel) {
  const uint32_t up = (sel & 1u) ? 16u : 0u, above = (sel & 2u) ? 7u : 0u;  // (row r + 1 is even where r is odd: the second carry area)
  return k < 32 ? k : (k < 48 ? k - 32 + up : (k == 48 ? TR_ST_BIT : (k < 65 ? k - 49 + up : TR_ST_CARRY + above + (k - 65))));
}
// my_info: bit_r | (r & 1) << 1 | live << 2 | k_lo << 3 (3 bits) | (k_hi - k_lo) << 6;  my_dst: element index of unit element k_lo in `out`
__device__ __forceinline__ void ladder_flush(const uint32_t (*stage)[TR_STAGE], const uint8_t* lut, uint64_t my_dst, uint32_t my_info, uint64_t* __restrict__ out) {
  __syncthreads();
  const uint32_t e = threadIdx.x;
  const uint32_t dst_lo = (uint32_t)my_dst, dst_hi = (uint32_t)(my_dst >> 32);
  const char* sb = reinterpret_cast<const char*>(&stage[0][0]);
  // The usual flush: every ladder of the wave has the same alignment and a span of exactly 64 elements -- then a lane's staged word depends
  // on the ladder's bit only, and the two candidates are looked up once per row instead of once per ladder (no table read, no exec mask
  // and so nothing between the LDS reads of consecutive ladders: they overlap).  Any other ladder takes the general path below.
  const uint32_t ref = (uint32_t)__builtin_amdgcn_readlane((int)my_info, 0);
  const uint32_t fast_key = ((ref & 4u) && (ref >> 6) == 64u) ? (ref & ~1u) : 0xffffffffu;
  const uint32_t kk = ((ref >> 3) & 7u) + ((ref >> 1) & 1u) * (2u * TR_SPAN_MAX);
  const uint32_t a0 = 4u * lut[kk + e], a1 = 4u * lut[kk + TR_SPAN_MAX + e];
#pragma unroll 8
  for (int j = 0; j < 64; j++) {
    const uint32_t b = (uint32_t)__builtin_amdgcn_readlane((int)my_info, j);
    uint64_t* dst = out + ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)dst_lo, j) | ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)dst_hi, j) << 32));
    if ((b & ~1u) == fast_key) {
      const uint32_t v = *reinterpret_cast<const uint32_t*>(sb + (size_t)j * (TR_STAGE * 4u) + ((b & 1u) ? a1 : a0));
      __builtin_nontemporal_store((uint64_t)v, dst + e);
      continue;
    }
    if (!(b & 4u)) continue;  // (ladders that do not exist are at the end of the last block)
    const uint32_t k0 = ((b >> 3) & 7u) + (b & 3u) * TR_SPAN_MAX, n = b >> 6;  // (scalar)
    const uint32_t w = lut[k0 + e];
    const uint32_t v = *reinterpret_cast<const uint32_t*>(sb + (size_t)j * (TR_STAGE * 4u) + 4u * w);
    if (e < n) __builtin_nontemporal_store((uint64_t)v, dst + e);
    if (n > 64u) {  // (a unit that starts on a line boundary, or the lowest row of a thread: up to eight more elements)
      const uint32_t w2 = lut[k0 + 64u + (e & 7u)];
      const uint32_t v2 = *reinterpret_cast<const uint32_t*>(sb + (size_t)j * (TR_STAGE * 4u) + 4u * w2);
      if (e < n - 64u) __builtin_nontemporal_store((uint64_t)v2, dst + 64 + e);
    }
  }
  __syncthreads();
}

template <int R>
__global__ __launch_bounds__(64) void k_trace_ladder_pass2(uint32_t lane0, uint32_t n_lanes, uint32_t n, const uint8_t* __restrict__ in_target, const uint8_t* __restrict__ ed,
                                                           uint32_t ed_stride, const int32_t* __restrict__ pts, uint64_t* __restrict__ out,
                                                           uint64_t proof_stride, uint32_t row0) {
  __shared__ uint32_t stage[64][TR_STAGE];
  __shared__ uint8_t lut[4 * TR_SPAN_MAX + 16];
  __shared__ int32_t s_t[10];
  __shared__ uint32_t s_inv[64];
  const uint32_t t = threadIdx.x, id = blockIdx.x * 64u + t;
  const int r_first = (int)(row0 + blockIdx.y * R), r_end = r_first + R;  // this thread's rows
  const bool live = id < 2u * n_lanes;
  const uint32_t lane = lane0 + (live ? id >> 1 : 0u), k = id & 1u;
  const uint64_t base = (uint64_t)(lane / n) * proof_stride + (uint64_t)((2u * (lane % n) + k) * TR_LADDER_ROWS) * TR_LADDER_ROW;
  for (uint32_t i = t; i < 4 * TR_SPAN_MAX + 16; i += 64) lut[i] = i < 4 * TR_SPAN_MAX ? (uint8_t)span_word(i % TR_SPAN_MAX, i / TR_SPAN_MAX) : (uint8_t)0;
  const LadderIn L = ladder_inputs(lane, k, in_target, ed, ed_stride, false);
  const int32_t* blk = pts + (size_t)blockIdx.x * TR_LADDER_ROWS * TR_ROW_WORDS;
  const uint32_t z = L.decoded ? 0xffffffffu : 0u;  // an undecodable lane: all-zero rows (Level-1 reports zero points there too)
  if (r_first == 0 && live) {  // the head of the ladder: bit_0 | acc_0 = the identity (0, 1)
    uint64_t* dst = out + base;
    dst[0] = scalar_bit(L.sc, 0) & z;
#pragma unroll
    for (int q = 0; q < 16; q++) dst[1 + q] = q == 8 ? (1u & z) : 0u;
  }
  // 1 / P(add of the thread's last row): ONE inversion per wave (winv25519.hpp: Montgomery's trick across the lanes + a cooperative Fermat chain,
  // ~13 k instructions) instead of 64 Bernstein-Yang inversions side by side (~21 k).  P is a product of Z coordinates of points on the curve: never zero.
  fe inv = wave_batch_invert(pt_load(pt_at(blk, r_end - 1, TR_PA), t), t, s_t, s_inv);
  uint32_t bit_above = r_end < (int)TR_LADDER_ROWS ? scalar_bit(L.sc, r_end) : 0u;
  const uint32_t out_align = (uint32_t)(reinterpret_cast<uintptr_t>(out) >> 3);  // (lines are 64 bytes of the ADDRESS, not of the element index)
  // A point's operands are requested together (the first form loaded each where it used it: six exposed memory latencies per row), and
  // those of the NEXT row's first point before this row's stores go out: vmcnt counts loads and stores in order, so a load issued behind the
  // 65 stores of a flush cannot be waited for without waiting for the stores -- the first point of a row is made affine while they drain.
  // pp = P(add of the row below) (row 0: the empty product), dz = Z(dbl): P(dbl_r) = pp dz is formed here, not stored by pass 1.
  fe az = pt_load(pt_at(blk, r_end - 1, TR_AZ), t), ax = pt_load(pt_at(blk, r_end - 1, TR_AX), t), ay = pt_load(pt_at(blk, r_end - 1, TR_AY), t);
  fe pp = r_end - 1 > 0 ? pt_load(pt_at(blk, r_end - 2, TR_PA), t) : fe_one(), dz = pt_load(pt_at(blk, r_end - 1, TR_DZ), t);
#pragma unroll 1
  for (int r = r_end - 1; r >= r_first; r--) {
    {  // add_r: 1 / Z = inv(P_add) P_dbl, then inv(P_dbl) = inv(P_add) Z
      uint32_t o[16];
      const fe zinv = fe_mul(inv, fe_mul(pp, dz));
      inv = fe_mul(inv, az);
      fe_to_words(fe_mul(ax, zinv), o);
      fe_to_words(fe_mul(ay, zinv), o + 8);
#pragma unroll
      for (int q = 0; q < 16; q++) stage[t][16 + q] = o[q] & z;
    }
    {  // dbl_r: 1 / Z = inv(P_dbl) P_add(r - 1)   (row 0: the empty product)
      uint32_t o[16];
      const fe zinv = r > 0 ? fe_mul(inv, pp) : inv;
      if (r > r_first) inv = fe_mul(inv, dz);
      fe_to_words(fe_mul(pt_load(pt_at(blk, r, TR_DX), t), zinv), o);
      fe_to_words(fe_mul(pt_load(pt_at(blk, r, TR_DY), t), zinv), o + 8);
#pragma unroll
      for (int q = 0; q < 16; q++) stage[t][q] = o[q] & z;
#pragma unroll
      for (int q = 0; q < 7; q++) stage[t][TR_ST_CARRY + ((r & 1) ? 0 : 7) + q] = o[q] & z;  // (for the flush of the row below)
    }
    if (r > r_first) {
      az = pt_load(pt_at(blk, r - 1, TR_AZ), t); ax = pt_load(pt_at(blk, r - 1, TR_AX), t); ay = pt_load(pt_at(blk, r - 1, TR_AY), t);
      pp = r - 1 > 0 ? pt_load(pt_at(blk, r - 2, TR_PA), t) : fe_one(); dz = pt_load(pt_at(blk, r - 1, TR_DZ), t);
    }
    const uint32_t bit = scalar_bit(L.sc, r);
    stage[t][TR_ST_BIT] = bit_above & z;  // (an undecodable lane stores zero bits too)
    // the span of this flush, in elements of the unit: [k_lo, k_hi)
    const uint64_t U = base + (uint64_t)r * TR_LADDER_ROW + 17u;
    const uint32_t a = ((uint32_t)U + out_align) & 7u, c = (8u - ((a + 1u) & 7u)) & 7u;
    const bool top = r == r_end - 1, last = r + 1 == (int)TR_LADDER_ROWS;
    const uint32_t k_lo = r == r_first ? 0u : (8u - a) & 7u, k_hi = last ? 48u : (top ? 65u : 65u + c);
    ladder_flush(stage, lut, U + k_lo, (bit & z & 1u) | (uint32_t)(r & 1) << 1 | (live ? 4u : 0u) | k_lo << 3 | (k_hi - k_lo) << 6, out);
    bit_above = bit;
  }
}

// SHA-512(R | A | M) of the effective triple of one lane (at most two blocks), 18 values per round
__global__ __launch_bounds__(64) void k_trace_sha512(uint32_t lane0, uint32_t n_lanes, uint32_t n, const uint8_t* __restrict__ in_target, uint64_t* __restrict__ out,
                                                     uint64_t proof_stride) {
  constexpr int RPF = 4, NV = RPF * TR_SHA512_ROW;  // rounds per flush
  __shared__ uint32_t stage[64][NV + 1];
  __shared__ uint64_t s_base[64];
  __shared__ uint8_t s_live[64];
  const uint32_t t = threadIdx.x, id = blockIdx.x * 64u + t;
  const bool live = id < n_lanes;
  const uint32_t lane = lane0 + (live ? id : 0u);
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
    // (sixteen rounds = four flushes per trip: the schedule index i mod 16 is then a constant and the select chains below fold away)
#pragma unroll 1
    for (int g4 = 0; g4 < 80 / RPF; g4 += 16 / RPF) {
#pragma unroll
     for (int g = g4; g < g4 + 16 / RPF; g++) {
#pragma unroll
      for (int u = 0; u < RPF; u++) {
        const int i = g * RPF + u;
        uint64_t wt;
        {  // rolling schedule, written for any i: a select chain that is constant-folded with i mod 16 known
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

size_t trace_tmp_bytes(uint32_t n, uint32_t n_proofs) { return (size_t)((2ull * n_proofs * n + 63) / 64) * TR_LADDER_ROWS * TR_ROW_WORDS * 4; }

// rows [row0, row1) of every ladder: the chain (pass 1) and, once it is done, the affine rows (pass 2); row0 / row1 multiples of eight
int launch_trace_ladder_pass1(uint32_t n, uint32_t n_proofs, const void* d_target, const void* d_ed, uint32_t ed_stride, void* d_tmp, uint32_t row0,
                              uint32_t row1, void* stream, uint32_t lane0, uint32_t lane_count) {
  const uint32_t lanes = lane_count == 0xffffffffu ? n_proofs * n : lane_count;
  if (lanes == 0) return 0;
  hipLaunchKernelGGL(k_trace_ladder_pass1, dim3((2 * lanes + 63) / 64), dim3(64), 0, S_(stream), lane0, lanes, reinterpret_cast<const uint8_t*>(d_target),
                     reinterpret_cast<const uint8_t*>(d_ed), ed_stride, reinterpret_cast<int32_t*>(d_tmp), row0, row1);
  return (int)hipGetLastError();
}
int launch_trace_ladder_pass2(uint32_t kind, uint32_t n, uint32_t n_proofs, const void* d_target, const void* d_ed, uint32_t ed_stride, const void* d_tmp,
                              void* d_out, uint32_t row0, uint32_t row1, void* stream, uint32_t lane0, uint32_t lane_count) {
  const uint32_t lanes = lane_count == 0xffffffffu ? n_proofs * n : lane_count, rows = row1 - row0;
  if (lanes == 0) return 0;
  // rows per inversion (= per thread): the inversion is ~13 k instructions, a row ~3 k; fewer rows = more, shorter threads.  TMX_TRACE_ROWS overrides.
#define TMX_TRACE_P2(RR)                                                                                                                       \
  hipLaunchKernelGGL((k_trace_ladder_pass2<RR>), dim3((2 * lanes + 63) / 64, rows / RR), dim3(64), 0, S_(stream), lane0, lanes, n,                      \
                     reinterpret_cast<const uint8_t*>(d_target), reinterpret_cast<const uint8_t*>(d_ed), ed_stride,                             \
                     reinterpret_cast<const int32_t*>(d_tmp), reinterpret_cast<uint64_t*>(d_out), trace_elems(kind, n), row0)
  static const int r_env = std::getenv("TMX_TRACE_ROWS") ? std::atoi(std::getenv("TMX_TRACE_ROWS")) : 16;
  if (r_env >= 64 && rows % 64 == 0) TMX_TRACE_P2(64);
  else if (r_env >= 32 && rows % 32 == 0) TMX_TRACE_P2(32);
  else if (r_env >= 16 && rows % 16 == 0) TMX_TRACE_P2(16);
  else TMX_TRACE_P2(8);
#undef TMX_TRACE_P2
  return (int)hipGetLastError();
}

// the other sections (bits 1, 2, 3 of `sections`)
int launch_trace_rest(uint32_t kind, uint32_t n, uint32_t n_proofs, const void* d_target, const void* d_trusted, const TraceLevel1& L1, void* d_out,
                      uint32_t sections, void* stream, uint32_t lane0, uint32_t lane_count) {
  if (n_proofs == 0) return 0;
  const uint32_t lanes = n_proofs * n;
  const uint32_t l512 = lane_count == 0xffffffffu ? lanes : lane_count;  // (only the per-lane SHA-512 section takes a lane range)
  const uint64_t stride = trace_elems(kind, n);
  uint64_t* out = reinterpret_cast<uint64_t*>(d_out);
  const uint8_t* tg = reinterpret_cast<const uint8_t*>(d_target);
  const uint8_t* tr = reinterpret_cast<const uint8_t*>(d_trusted);
  if ((sections & 2u) && l512) hipLaunchKernelGGL(k_trace_sha512, dim3((l512 + 63) / 64), dim3(64), 0, S_(stream), lane0, l512, n, tg, out, stride);
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
