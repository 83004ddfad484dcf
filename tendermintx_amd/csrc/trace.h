// Launch wrapper of the Level-2 trace-row kernels (trace.hip); plain C++ so that api.cpp needs no device code.
#pragma once
#include <stdint.h>

#include "layout.h"

namespace tmx {

// elements of one proof's trace block (DESIGN.md "Level-2 trace rows")
uint64_t trace_elems(uint32_t kind, uint32_t n);
// sections: bit 0 ladders, 1 SHA-512 rounds, 2 leaf SHA-256 rounds, 3 N x N match bits.  d_ed: the Level-1 EdDSA lane records of the
// SAME batch (h, A, decode flag).  Returns a hipError_t value.
// d_tmp: trace_tmp_bytes(n, n_proofs) bytes of scratch (the projective points between the two ladder passes), needed with section bit 0
size_t trace_tmp_bytes(uint32_t n, uint32_t n_proofs);
int launch_trace(uint32_t kind, uint32_t n, uint32_t n_proofs, const void* d_target, const void* d_trusted, const void* d_ed, uint32_t ed_stride, void* d_out,
                 void* d_tmp, uint32_t sections, void* stream);

}  // namespace tmx
