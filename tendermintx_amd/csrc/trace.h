// Launch wrapper of the Level-2 trace-row kernels (trace.hip); plain C++ so that api.cpp needs no device code.
#pragma once
#include <stdint.h>

#include "layout.h"

namespace tmx {

// elements of one proof's trace block (DESIGN.md "Level-2 trace rows")
uint64_t trace_elems(uint32_t kind, uint32_t n);
// Ladder rows [row0, row1) (row1 - row0 a multiple of eight) of every lane: pass 1 = the double-and-add chain + running Z products into d_tmp
// (trace_tmp_bytes; a segment with row0 > 0 continues from the buffer), pass 2 = affine rows out of it.  d_ed: the Level-1 EdDSA lane
// records of the SAME batch (h, A, decode flag).  Each returns a hipError_t value.
size_t trace_tmp_bytes(uint32_t n, uint32_t n_proofs);
int launch_trace_ladder_pass1(uint32_t n, uint32_t n_proofs, const void* d_target, const void* d_ed, uint32_t ed_stride, void* d_tmp, uint32_t row0,
                              uint32_t row1, void* stream, uint32_t lane0 = 0, uint32_t lane_count = 0xffffffffu);
int launch_trace_ladder_pass2(uint32_t kind, uint32_t n, uint32_t n_proofs, const void* d_target, const void* d_ed, uint32_t ed_stride, const void* d_tmp,
                              void* d_out, uint32_t row0, uint32_t row1, void* stream, uint32_t lane0 = 0, uint32_t lane_count = 0xffffffffu);
// (lane0 / lane_count: only lanes [lane0, lane0 + lane_count) of the batch -- the lane-sharded form; inputs and outputs stay indexed by the absolute lane)
// Level-1 values of the SAME batch the tree / header sections hash over (the context's scratch): per-lane derived records of the target set
// (d_lt: LT part inside TL, lt_stride) and of the trusted set (d_lr), the tree nodes of both sets, the per-proof derived records
struct TraceLevel1 {
  const void *d_lt, *d_lr, *d_nodes_t, *d_nodes_r, *d_pf;
  uint32_t lt_stride;
};
// sections: bit 1 SHA-512 rounds, 2 leaf SHA-256 rounds, 3 N x N match bits, 4 inner tree nodes, 5 header-proof hashes
int launch_trace_rest(uint32_t kind, uint32_t n, uint32_t n_proofs, const void* d_target, const void* d_trusted, const TraceLevel1& L1, void* d_out,
                      uint32_t sections, void* stream, uint32_t lane0 = 0, uint32_t lane_count = 0xffffffffu);

// the commit pipeline's first stage (api.cpp: tmx_trace_commit_device): one section as a row-major matrix, and its columns for the LDE
bool trace_section_geom(uint32_t kind, uint32_t n, uint32_t section, uint64_t* off, uint64_t* rows, uint32_t* width);
int launch_trace_to_columns(const void* d_trace, uint64_t proof_stride, uint64_t sec_off, uint64_t rows, uint32_t width, uint32_t log_n, uint32_t n_proofs,
                            void* d_cols, void* stream);

}  // namespace tmx
