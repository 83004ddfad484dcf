// The typed value of the hint (include/tmx.h "TYPED VALUE"): layout arithmetic + the launch wrapper of k_pack_value (value.hip).
// Plain C++ so that api.cpp needs no device code.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include "layout.h"

namespace tmx {

// sizes of the public structs (static_assert'ed against include/tmx.h in api.cpp)
constexpr uint32_t VAL_VALIDATOR = 240, VAL_HASHFIELD = 48, VAL_FIXED_SKIP = 832, VAL_FIXED_STEP = 1008, VAL_LANE_T = TL_STRIDE, VAL_LANE_R = LANE_STRIDE,
                   VAL_PROOF_D = 976;
constexpr uint32_t VAL_FIXED_MAX = VAL_FIXED_STEP;

// One proof's value = a handful of PARTS laid end to end, each a multiple of 16 bytes: k_pack_value gives every thread one 16-byte
// chunk of one part.
enum : uint32_t { VP_FIXED = 0, VP_VALIDATORS, VP_HASHFIELDS, VP_LANE_T, VP_LANE_R, VP_NODES_T, VP_NODES_R, VP_PROOF_D, VP_COUNT };
struct ValueLayout {
  uint32_t kind, n, tree_nodes, sections;
  uint32_t off[VP_COUNT + 1];  // byte offset of part p inside a proof's value; off[p + 1] - off[p] = its length (0: absent); off[VP_COUNT] = bytes per proof
};
ValueLayout value_layout(uint32_t kind, uint32_t n, uint32_t tree_nodes, uint32_t sections);

// fixed part: one u16 per output byte, (source << 12) | byte offset inside that source's per-proof record; 0xffff = a zero byte
enum : uint32_t { VSRC_PF = 0, VSRC_PROOF = 1, VSRC_REPORT = 2 };
constexpr uint16_t VAL_LUT_ZERO = 0xffffu;

struct ValueSources {
  const uint8_t *proofs, *targets, *trusteds;  // the input records
  const uint8_t *tl, *lr, *pf, *nodes_t, *nodes_r, *reports;  // the context's Level-1 records of the batch
};
// d_fixed_lut: VAL_FIXED_MAX u16 entries for this kind.  d_out: n_proofs values of L.off[VP_COUNT] bytes (device memory or mapped host memory)
int launch_pack_value(const ValueLayout& L, const ValueSources& src, const void* d_fixed_lut, uint32_t n_proofs, void* d_out, void* stream);

}  // namespace tmx
