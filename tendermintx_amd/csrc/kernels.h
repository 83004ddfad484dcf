// Launch wrappers of the HIP kernels (kernels.hip); plain C++ so that api.cpp needs no device code.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include "layout.h"

namespace tmx {

size_t base_table_bytes(uint32_t w_bits);
// each returns a hipError_t value (0 = success); all launches are asynchronous on `stream` (hipStream_t)
int launch_init_base(void* d_table, uint32_t w_bits, void* stream);
int launch_selftest_invert(uint32_t n, const void* d_in, void* d_out, void* stream);
int launch_selftest_f16(uint32_t n, uint32_t doublings, const void* d_in, void* d_out, void* stream);
int launch_eddsa(uint32_t n_lanes, const void* d_target, void* d_ed, uint32_t ed_stride, const void* d_table, void* stream);
// quad-parallel EdDSA path.  Three launch groups so that api.cpp can run the key pipeline on a side stream:
//   keys pipeline (dedup -> decode distinct keys -> optional per-key tables)  ||  phase 1 (decode R, SHA-512 mod l, s*B)
//   then h*A + finish
struct EdQuad {
  uint32_t n_lanes;
  const void* d_target;
  void* d_ed;
  uint32_t ed_stride;
  const void* d_qtable;
  uint32_t base_w;   // window width of the fixed-base table of B (4, 8 or 10 bits)
  void *d_pre, *d_mulout;
  void* d_hash;
  uint32_t hash_mask;
  void *d_cnt, *d_cnt_next, *d_owner_of, *d_uid_of_owner, *d_owners, *d_keyrec, *d_anchors, *d_keytab;
  uint32_t key_w;    // window width of the per-key tables (4 or 6 bits)
  uint32_t key_cap;  // keys the table buffers can hold
  uint32_t mode;     // 0 never build tables, 1 automatic (>= 8 lanes per key), 2 whenever they fit
  uint32_t mul16;    // 1: h*A of small launches without tables in the limb-parallel form (one wave per lane)
  uint32_t keys16;   // 1: keys are decoded in the limb-parallel form when there are few enough of them
  uint32_t anchor16; // 1: the anchor chain runs in the limb-parallel form (one wave per key), 0: one quad per key
  uint32_t mul_split; // quads per lane in the table walk: 1, 2, 4, or 0 = by launch size (TMX_MUL_SPLIT)
  void* fin_done;    // event attached to the k_ed_fin dispatch as its completion signal (no separate record packet), or null
};
size_t quad_table_bytes(uint32_t w_bits);
size_t pre_bytes_per_lane();
size_t mulout_bytes_per_lane();
size_t key_bytes_per_key();
size_t anchor_bytes_per_key(uint32_t key_w);
size_t keytab_bytes_per_key(uint32_t key_w);
int launch_init_base_quad(const void* d_table, void* d_qtable, uint32_t w_bits, void* stream);
// (`done`: an event signalled by the dispatch itself when the kernel completes -- saves the record packet behind it; may be null)
int launch_ed_dedup(const EdQuad& Q, void* stream, void* done = nullptr);
// direct_n != 0: no dedup ran, lanes 0 .. direct_n-1 are their own keys (limb-parallel form only: direct_n <= 8192, Q.keys16 set)
int launch_ed_keys(const EdQuad& Q, void* stream, void* done = nullptr, uint32_t direct_n = 0);
int launch_ed_tab_anchor(const EdQuad& Q, uint32_t part, uint32_t parts, void* stream, void* done = nullptr);
int launch_ed_tab_mult(const EdQuad& Q, uint32_t part, uint32_t parts, void* stream, void* done = nullptr);
// fuse_fin (launches of <= MUL16_MAX_LANES lanes with Q.mul16 only): k_ed_fin's work in the same kernel, Q.fin_done on its dispatch
int launch_ed_mul_direct(const EdQuad& Q, void* stream, bool fuse_fin = false);
// roles: 0 = both (SHA-512 + mod l, then s*B), 1 = the hash role only, 2 = s*B only
int launch_ed_phase1(const EdQuad& Q, void* stream, void* done = nullptr, int roles = 0);
// (lane0 .. lane_end - 1: the lanes of this launch; lane_end = 0: to the end)
int launch_ed_mul_tab(const EdQuad& Q, uint32_t part, uint32_t parts, void* stream, uint32_t lane0 = 0, uint32_t lane_end = 0);
int launch_ed_fin(const EdQuad& Q, void* stream, uint32_t lane0 = 0, uint32_t lane_end = 0);
int launch_proof(const ProofParams& P, uint32_t n_proofs, const void* d_proofs, const void* d_target, const void* d_trusted, void* d_lt,
                 uint32_t lt_stride, void* d_lr, void* d_pf, void* d_nodes_t, void* d_nodes_r, void* d_reports, void* stream);
// marshalled validators + leaf hashes of both sets as a launch of its own; k_proof then reads them (ProofParams::leaves_done)
int launch_leaves(uint32_t kind, uint32_t n, uint32_t n_proofs, const void* d_target, const void* d_trusted, void* d_lt, uint32_t lt_stride, void* d_lr,
                  void* stream, void* done = nullptr);
int launch_valid_skip(uint32_t n_cand, uint32_t n_max, const void* d_start, uint32_t n_start, const void* d_targets, const void* d_nt, const void* d_sigs,
                      const void* d_ns, void* d_valid, void* d_shared, void* d_total, void* stream);
int launch_verdict(uint32_t kind, uint32_t n, uint32_t n_proofs, const void* d_ed, uint32_t ed_stride, void* d_pf, void* d_reports, void* stream,
                   void* started = nullptr, void* done = nullptr);
// sec_mask: bit s = section s, bit 31 = waves straddling a section boundary / the row end
int launch_serialize(const SerializeProgram& S, const SerializeSources& src, const void* d_lut, const void* d_wave_sec, const void* d_seam_waves,
                     uint32_t n_seams, uint32_t n_proofs, void* d_out, uint32_t sec_mask, void* stream, uint32_t max_wgs = 0, uint32_t proof0 = 0);
// (max_wgs != 0: at most that many workgroups walk the spans -- launches that run beside the EdDSA latency chain must not fill every wave slot)
// packs elements [first, first + row_elems) of every row densely into d_out as u64 or u32 (transfer formats of the host entry point)
int launch_pack_rows(const void* d_rows, void* d_out, uint32_t elem_stride, uint32_t first, uint32_t row_elems, uint32_t n_proofs, bool as_u32,
                     void* stream);

}  // namespace tmx
