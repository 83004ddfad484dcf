// Launch wrappers of the HIP kernels (kernels.hip); plain C++ so that api.cpp needs no device code.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include "layout.h"

namespace tmx {

size_t base_table_bytes();
// each returns a hipError_t value (0 = success); all launches are asynchronous on `stream` (hipStream_t)
int launch_init_base(void* d_table, void* stream);
int launch_eddsa(uint32_t n_lanes, const void* d_target, void* d_ed, uint32_t ed_stride, const void* d_table, void* stream);
// quad-parallel EdDSA path (k_ed_pre -> k_ed_mul -> k_ed_fin)
size_t quad_table_bytes();
size_t pre_bytes_per_lane();
size_t mulout_bytes_per_lane();
int launch_init_base_quad(const void* d_table, void* d_qtable, void* stream);
int launch_eddsa_quad(uint32_t n_lanes, const void* d_target, void* d_ed, uint32_t ed_stride, const void* d_qtable, void* d_pre, void* d_mulout,
                      void* stream);
int launch_proof(const ProofParams& P, uint32_t n_proofs, const void* d_proofs, const void* d_target, const void* d_trusted, void* d_lt,
                 uint32_t lt_stride, void* d_lr, void* d_pf, void* d_nodes_t, void* d_nodes_r, void* d_reports, void* stream);
int launch_verdict(uint32_t kind, uint32_t n, uint32_t n_proofs, const void* d_ed, uint32_t ed_stride, void* d_pf, void* d_reports, void* stream);
int launch_serialize(const SerializeProgram& S, const SerializeSources& src, const void* d_lut, const void* d_wave_sec, uint32_t n_proofs,
                     void* d_out, void* stream);

}  // namespace tmx
