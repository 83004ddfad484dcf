// Launch wrappers of the HIP kernels (kernels.hip); plain C++ so that api.cpp needs no device code.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include "layout.h"

namespace tmx {

size_t base_table_bytes();
// each returns a hipError_t value (0 = success); all launches are asynchronous on `stream` (hipStream_t)
int launch_init_base(void* d_table, void* stream);
int launch_eddsa(uint32_t n_lanes, const void* d_target, void* d_ed, const void* d_table, void* stream);
int launch_proof(const ProofParams& P, uint32_t n_proofs, const void* d_proofs, const void* d_target, const void* d_trusted, const void* d_ed,
                 void* d_lt, void* d_lr, void* d_pf, void* d_nodes_t, void* d_nodes_r, void* d_reports, void* stream);
int launch_serialize(const SerializeProgram& S, const SerializeSources& src, const void* d_lut, uint32_t n_proofs, void* d_out, void* stream);

}  // namespace tmx
