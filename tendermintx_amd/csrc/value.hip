// k_pack_value: the typed value of the hint (include/tmx.h "TYPED VALUE") -- SkipInputs<F> / StepInputs<F> of the reference
// (circuits/input/mod.rs:45-74) field by field, bytes as bytes and u64 as u64, plus (optionally) the derived Level-1 values in packed
// form -- gathered from the input records and from the Level-1 records the batch's kernels left in the context.
//
// Pure data movement: 38 KB per N = 128 skip proof (134 KB with the derived values) against the 1.86 MB of the expanded u64 row, so
// the launch is a few microseconds even for 256 proofs; what matters is that every store is a whole 16-byte chunk (the destination may
// be mapped page-locked host memory: posted PCIe writes, no read-modify-write) and that pad bytes are written as zeros (the value is
// compared byte for byte with the oracle's).  One thread = one 16-byte chunk of one part of one proof's value.
#include <hip/hip_runtime.h>

#include "value.h"

namespace tmx {

ValueLayout value_layout(uint32_t kind, uint32_t n, uint32_t tree_nodes, uint32_t sections) {
  ValueLayout L = {};
  L.kind = kind; L.n = n; L.tree_nodes = tree_nodes; L.sections = sections;
  const bool skip = kind == 0, derived = (sections & 2u) != 0;
  uint32_t len[VP_COUNT] = {};
  len[VP_FIXED] = skip ? VAL_FIXED_SKIP : VAL_FIXED_STEP;
  len[VP_VALIDATORS] = n * VAL_VALIDATOR;
  len[VP_HASHFIELDS] = skip ? n * VAL_HASHFIELD : 0;
  if (derived) {
    len[VP_LANE_T] = n * VAL_LANE_T;
    len[VP_LANE_R] = skip ? n * VAL_LANE_R : 0;
    len[VP_NODES_T] = tree_nodes * 32;
    len[VP_NODES_R] = skip ? tree_nodes * 32 : 0;
    len[VP_PROOF_D] = VAL_PROOF_D;
  }
  uint32_t o = 0;
  for (uint32_t p = 0; p < VP_COUNT; p++) { L.off[p] = o; o += len[p]; }
  L.off[VP_COUNT] = o;
  return L;
}

namespace {

__device__ __forceinline__ uint4 ld16(const uint8_t* p) { return *reinterpret_cast<const uint4*>(p); }
__device__ __forceinline__ uint32_t ld4(const uint8_t* p) { return *reinterpret_cast<const uint32_t*>(p); }

__global__ __launch_bounds__(256) void k_pack_value(const ValueLayout L, const ValueSources S, const uint16_t* __restrict__ fixed_lut, uint8_t* __restrict__ out,
                                                    const uint32_t proof0) {
  const uint32_t proof = proof0 + blockIdx.y;
  const uint32_t byte = (blockIdx.x * 256u + threadIdx.x) * 16u;
  if (byte >= L.off[VP_COUNT]) return;
  uint32_t part = 0;
#pragma unroll
  for (uint32_t p = 1; p < VP_COUNT; p++) part += byte >= L.off[p] ? 1u : 0u;  // (offsets are non-decreasing; empty parts share an offset with their successor)
  const uint32_t rel = byte - L.off[part];
  const uint32_t n = L.n;
  uint4 v = make_uint4(0, 0, 0, 0);
  switch (part) {
    case VP_FIXED: {
      const uint8_t* base[3] = {S.pf + (size_t)proof * PF_STRIDE, S.proofs + (size_t)proof * PR_STRIDE, S.reports + (size_t)proof * 64};
      uint32_t w[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        uint32_t acc = 0;
#pragma unroll
        for (int b = 0; b < 4; b++) {
          const uint32_t e = fixed_lut[rel + 4 * k + b];
          const uint32_t x = e == VAL_LUT_ZERO ? 0u : (uint32_t)base[e >> 12][e & 0xfffu];
          acc |= x << (8 * b);
        }
        w[k] = acc;
      }
      v = make_uint4(w[0], w[1], w[2], w[3]);
      break;
    }
    case VP_VALIDATORS: {  // tmx_validator_rec (256 B) -> tmx_validator_value (240 B): bytes 0 .. 219 are the same in both
      const uint32_t lane = rel / VAL_VALIDATOR, c = (rel - lane * VAL_VALIDATOR) >> 4;
      const uint8_t* r = S.targets + ((size_t)proof * n + lane) * VR_STRIDE;
      if (c < 13) v = ld16(r + 16 * c);
      else if (c == 13) { v = ld16(r + 208); v.w &= 0xffffu; }                                     // message[112..124) | message_byte_length
      else { const uint32_t w55 = ld4(r + 220); v = make_uint4(ld4(r + VR_OFF_POWER), ld4(r + VR_OFF_POWER + 4), (w55 >> 16) & 0xffu, (w55 >> 24) & 1u); }
      break;
    }
    case VP_HASHFIELDS: {
      const uint32_t lane = rel / VAL_HASHFIELD, c = (rel - lane * VAL_HASHFIELD) >> 4;
      const uint8_t* r = S.trusteds + ((size_t)proof * n + lane) * HR_STRIDE;
      if (c < 2) v = ld16(r + 16 * c);
      else v = make_uint4(ld4(r + HR_OFF_POWER), ld4(r + HR_OFF_POWER + 4), ld4(r + HR_OFF_VLEN) & 0xffu, 0u);
      break;
    }
    case VP_LANE_T: {  // the context's per-target-lane record: ED part (448 B) | LT part (112 B)
      const uint32_t lane = rel / TL_STRIDE, c = (rel - lane * TL_STRIDE) >> 4;
      const uint8_t* r = S.tl + ((size_t)proof * n + lane) * TL_STRIDE;
      if (c < 26) v = ld16(r + 16 * c);                                                            // digest, h, ten coordinates
      else if (c == 26) { v = ld16(r + 416); v.z = 0; v.w = 0; }                                   // eddsa_ok, decode_ok
      else if (c == 27) {}
      else {
        const uint32_t j = c - 28;
        const uint8_t* t = r + TL_OFF_LT;
        v = ld16(t + 16 * j);
        if (j == 2) v.w &= 0xffffu;                                                                // marshalled[44..46) | pad
        else if (j == 5) v.y &= 0xffffu;                                                           // flags[4..6) | pad
        else if (j == 6) { v.z = 0; v.w = 0; }
      }
      break;
    }
    case VP_LANE_R: {
      const uint32_t lane = rel / LANE_STRIDE, j = (rel - lane * LANE_STRIDE) >> 4;
      const uint8_t* t = S.lr + ((size_t)proof * n + lane) * LANE_STRIDE;
      v = ld16(t + 16 * j);
      if (j == 2) v.w &= 0xffffu;
      else if (j == 5) { v.x &= 0xffffu; v.y = 0; }                                                // flags[0..2) | pad
      else if (j == 6) { v.z = 0; v.w = 0; }
      break;
    }
    case VP_NODES_T: v = ld16(S.nodes_t + (size_t)proof * L.tree_nodes * 32 + rel); break;
    case VP_NODES_R: v = ld16(S.nodes_r + (size_t)proof * L.tree_nodes * 32 + rel); break;
    default: {  // VP_PROOF_D: bytes [PF_OFF_PROOFD, PF_OFF_PROOFD + 976) of the per-proof record, fields the kind does not have as zeros
      const uint8_t* r = S.pf + (size_t)proof * PF_STRIDE + PF_OFF_PROOFD;
      const bool skip = L.kind == 0;
      const uint32_t c = rel >> 4;
      v = ld16(r + rel);
      if (c < 50) { if (skip && c >= 40) v = make_uint4(0, 0, 0, 0); }                             // proofs[q][5][32]: q = 4 is step's
      else if (c == 50) { v.z &= 0x00ffffffu; v.w = 0; }                                           // height_leaf[11] | pad
      else if (c < 53) {}                                                                          // tally_target
      else if (c < 55) { if (!skip) v = make_uint4(0, 0, 0, 0); }                                  // tally_trusted
      else if (c == 55) { if (!skip) { v.y = 0; v.z = 0; v.w = 0; } }                              // verdicts
      else if (c < 60) {                                                                           // checks[16]: 13 / 15 are defined
        const uint32_t k0 = (c - 56) * 4, nchk = skip ? 13u : 15u;
        if (k0 + 0 >= nchk) v.x = 0;
        if (k0 + 1 >= nchk) v.y = 0;
        if (k0 + 2 >= nchk) v.z = 0;
        if (k0 + 3 >= nchk) v.w = 0;
      } else v.y = 0;                                                                              // all_ok | pad | height
      break;
    }
  }
  *reinterpret_cast<uint4*>(out + (size_t)proof * L.off[VP_COUNT] + byte) = v;
}

}  // namespace

int launch_pack_value(const ValueLayout& L, const ValueSources& src, const void* d_fixed_lut, uint32_t n_proofs, void* d_out, void* stream) {
  if (n_proofs == 0) return 0;
  const uint32_t chunks = L.off[VP_COUNT] / 16;
  // (grid.y holds at most 65535: a context with a small n_max can have a larger max_batch -- chunked over proofs like k_serialize)
  for (uint32_t proof0 = 0; proof0 < n_proofs; proof0 += 65535u) {
    const uint32_t np = n_proofs - proof0 < 65535u ? n_proofs - proof0 : 65535u;
    hipLaunchKernelGGL(k_pack_value, dim3((chunks + 255) / 256, np), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), L, src,
                       reinterpret_cast<const uint16_t*>(d_fixed_lut), reinterpret_cast<uint8_t*>(d_out), proof0);
    const int e = (int)hipGetLastError();
    if (e) return e;
  }
  return 0;
}

}  // namespace tmx
