// Launch wrappers of the HIP kernels (kernels.hip); plain C++ so that api.cpp needs no device code.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include "layout.h"

namespace tmx {

// persistent per-key table cache of a context (DESIGN.md "Key cache"): device pointers + geometry, passed to the kernels by value
struct KeyCache {
  uint32_t cap;        // slots: decoded key record + window table (655 KB at the default 8-bit windows) each
  uint32_t hash_mask;  // open-addressing table of (hash_mask + 1) slot ids, >= 4 * cap
  uint32_t new_cap;    // tables one launch builds at most (anchor scratch)
  uint32_t persist;    // 1: keys stay resident across launches; 0: every launch starts from an empty cache (same code path)
  uint32_t* d_hash;    // [hash_mask + 1] slot id or 0xffffffff
  uint32_t* d_pk;      // [cap][8] the 32 key bytes of a resident slot (all compared on a probe)
  uint32_t* d_used;    // [cap] epoch of the last launch that used the slot; 0 = free
  uint32_t* d_free;    // [cap] free list: d_free[free_head .. n_free) are free slots
  uint32_t* d_state;   // KC_STATE_WORDS words, indices below
  uint32_t* hint;      // page-locked host memory the epilogue writes for the NEXT enqueue to look at: [0] launches committed, [1] new keys of the last
};
enum : uint32_t { KC_FREE_HEAD = 0, KC_N_FREE = 1, KC_EPOCH = 2, KC_RESIDENT = 3, KC_LAST_NEW = 4, KC_LAST_HIT_KEYS = 5, KC_LAST_HIT_LANES = 6,
                  KC_LAST_BUILT = 7, KC_LAST_USE_NEW = 8, KC_TOTALS = 16 /* u64 counters from this word on */, KC_STATE_WORDS = 32 };
enum : uint32_t { KC_TOT_HIT_LANES = 0, KC_TOT_MISS_LANES = 1, KC_TOT_BUILT = 2, KC_TOT_EVICTED = 3, KC_TOT_GC_RUNS = 4, KC_TOT_LAUNCHES = 5 };

size_t base_table_bytes();
// each returns a hipError_t value (0 = success); all launches are asynchronous on `stream` (hipStream_t)
int launch_init_base(void* d_table, void* stream);
int launch_selftest_invert(uint32_t n, const void* d_in, void* d_out, void* stream);
int launch_selftest_f16(uint32_t n, uint32_t doublings, const void* d_in, void* d_out, void* stream);
// EdDSA stage.  Launch groups so that api.cpp can run the key pipeline on a side stream:
//   key pipeline (dedup -> decode new keys -> their tables -> cache epilogue)  ||  phase 1 (SHA-512 mod l, s*B)
//   then h*A (table walk for lanes whose key has a table, table-free form for the others) + finish
struct EdQuad {
  uint32_t n_lanes;
  const void* d_target;
  void* d_ed;
  uint32_t ed_stride;
  const void* d_qtable;
  void *d_pre, *d_mulout;
  void* d_hash;       // the launch's own dedup hash table (lanes whose key is not resident)
  uint32_t hash_mask;
  void *d_cnt, *d_cnt_next, *d_owner_of, *d_slot_of_owner, *d_slot_of_uid, *d_owners, *d_keyrec, *d_anchors, *d_keytab;
  KeyCache kc;
  uint32_t mode;      // 0 never use tables, 1 automatic, 2 whenever they fit (TMX_DEDUP)
  uint32_t use_new;   // 1: the walk is enqueued behind the table build, lanes of new keys may use their fresh tables
  uint32_t warm;      // 1: the launch expects (nearly) all of its keys to be resident: small grids for the new-key kernels (they loop)
  RowOut row;         // where the finish writes D.1b straight into the witness rows (rows = null: lane records only)
  void* fin_done;     // event attached to the k_ed_fin dispatch as its completion signal (no separate record packet), or null
  // Compacted launch: lanes that did not sign receive the context's precomputed dummy record in k_ed_dedup, the lanes that signed are listed
  // densely in d_live (their count in cnt[4]) and the hash / s*B / walk / finish kernels run over that list.
  uint32_t compact = 0;
  void* d_live = nullptr;
  const void* d_dummy_ed = nullptr;
  // Fused rows (layout.h FusedRows): the input-only row spans carried by k_ed_base / the resident k_ed_mul_tab; fused.ctr == null: off
  FusedRows fused = {};
  SerializeProgram fused_prog = {};  // the program resolved for this batch (resolve_serialize_program)
};
size_t quad_table_bytes();
size_t pre_bytes_per_lane();
size_t mulout_bytes_per_lane();
size_t key_bytes_per_key();
size_t anchor_bytes_per_key();
size_t keytab_bytes_per_key();
int launch_init_base_quad(const void* d_table, void* d_qtable, void* stream);
int launch_kc_reset(const KeyCache& kc, void* stream);
// (`done`: an event signalled by the dispatch itself when the kernel completes -- saves the record packet behind it; may be null)
int launch_ed_dedup(const EdQuad& Q, void* stream, void* done = nullptr);
int launch_ed_keys(const EdQuad& Q, void* stream, void* done = nullptr);
int launch_ed_tab_anchor(const EdQuad& Q, uint32_t part, uint32_t parts, void* stream, void* done = nullptr);
int launch_ed_tab_mult(const EdQuad& Q, uint32_t part, uint32_t parts, void* stream, void* done = nullptr);
int launch_kc_epilogue(const EdQuad& Q, void* stream);
// fuse_fin (launches of <= 2048 lanes): k_ed_fin's work for the lanes this kernel multiplies, in the same kernel
int launch_ed_mul_direct(const EdQuad& Q, void* stream, bool fuse_fin = false, void* done = nullptr);
int launch_ed_phase1(const EdQuad& Q, void* stream, void* done = nullptr);
int launch_ed_hash(const EdQuad& Q, void* stream, void* done = nullptr);
// which: 0 every lane whose key has a table, 1 only lanes of resident keys, 2 only lanes of keys whose table this launch builds
int launch_ed_mul_tab(const EdQuad& Q, uint32_t part, uint32_t parts, void* stream, uint32_t which = 0);
int launch_ed_base(const EdQuad& Q, void* stream, void* done = nullptr);  // s*B alone (the hash role ran as k_ed_hash)
// which: 0 every lane, 1 only lanes of resident keys (cache hits), 2 only the others (the split warm schedule finishes them on the side stream)
int launch_ed_fin(const EdQuad& Q, void* stream, bool fused_direct = false, uint32_t which = 0);
int launch_proof(const ProofParams& P, uint32_t n_proofs, const void* d_proofs, const void* d_target, const void* d_trusted, void* d_lt,
                 uint32_t lt_stride, void* d_lr, void* d_pf, void* d_nodes_t, void* d_nodes_r, void* d_reports, void* stream, void* started = nullptr,
                 void* done = nullptr, const SetCache& SC = SetCache{});
constexpr uint32_t KC_EPILOGUE_CLEARS_UP_TO = 1u << 16;  // launch hash tables up to this many words are cleared by k_kc_epilogue itself
// marshalled validators + leaf hashes of both sets as a launch of its own; k_proof then reads them (ProofParams::leaves_done)
int launch_leaves(uint32_t kind, uint32_t n, uint32_t n_proofs, const void* d_target, const void* d_trusted, void* d_lt, uint32_t lt_stride, void* d_lr,
                  void* stream, void* done = nullptr);
int launch_valid_skip(uint32_t n_cand, uint32_t n_max, const void* d_start, uint32_t n_start, const void* d_targets, const void* d_nt, const void* d_sigs,
                      const void* d_ns, void* d_valid, void* d_shared, void* d_total, void* stream);
// (row.rows != null: d_ed is the ED part of the unified lane records, and the ten k_proof-derived D.1b elements of every lane are written)
int launch_verdict(uint32_t kind, uint32_t n, uint32_t n_proofs, const void* d_ed, uint32_t ed_stride, void* d_pf, void* d_reports, const RowOut& row,
                   void* stream, void* started = nullptr, void* done = nullptr);
// a few proofs: k_verdict, the serialization of the sections in sec_mask (bit 31: the seam spans) element by element and the seams in ONE launch
int launch_verdict_tail(uint32_t kind, uint32_t n, uint32_t n_proofs, const void* d_ed, uint32_t ed_stride, void* d_pf, void* d_reports, const RowOut& row,
                        const SerializeProgram& S, const SerializeSources& src, const void* d_lut, const void* d_seam_waves, uint32_t n_seams, void* d_out,
                        uint32_t sec_mask, void* stream, void* started = nullptr, void* done = nullptr);
// any batch size: k_verdict, the spans of the sections in sec_mask in front of the first element that depends on the verdict, the seam spans and
// the dependent end of the row in ONE launch of independent workgroups (k_verdict_tail_wide)
int launch_verdict_tail_wide(uint32_t kind, uint32_t n, uint32_t n_proofs, const void* d_ed, uint32_t ed_stride, void* d_pf, void* d_reports,
                             const RowOut& row, const SerializeProgram& S, const SerializeSources& src, const void* d_lut, const void* d_wave_sec,
                             const void* d_seam_waves, uint32_t n_seams, void* d_out, uint32_t sec_mask, uint32_t tail_dep_elem, void* stream,
                             void* started = nullptr, void* done = nullptr);
// LRU eviction of the validator-set cache (layout.h SetCache): one workgroup behind a k_proof launch on its stream
int launch_setc_evict(const SetCache& SC, uint32_t epoch, void* stream);
SerializeProgram resolve_serialize_program(const SerializeProgram& S, const SerializeSources& src);
// the sweeper of the fused rows: at most max_wgs workgroups claim the spans the carriers have not taken; zeroes *d_zero_ctr (the other parity's counter)
int launch_serialize_claim(const SerializeProgram& S_resolved, const FusedRows& F, void* d_zero_ctr, void* stream, uint32_t max_wgs);
// sec_mask: bit s = section s, bit 31 = waves straddling a section boundary / the row end
int launch_serialize(const SerializeProgram& S, const SerializeSources& src, const void* d_lut, const void* d_wave_sec, const void* d_seam_waves,
                     uint32_t n_seams, uint32_t n_proofs, void* d_out, uint32_t sec_mask, void* stream, uint32_t max_wgs = 0, uint32_t proof0 = 0);
// (max_wgs != 0: at most that many workgroups walk the spans -- launches that run beside the EdDSA latency chain must not fill every wave slot)
// packs elements [first, first + row_elems) of every row densely into d_out as u64 or u32 (transfer formats of the host entry point)
int launch_pack_rows(const void* d_rows, void* d_out, uint32_t elem_stride, uint32_t first, uint32_t row_elems, uint32_t n_proofs, bool as_u32,
                     void* stream);


// ---- the small-launch path (tiny.hpp): a whole batch of <= TINY_MAX_LANES validator lanes as TWO launches on the caller's stream
constexpr uint32_t TINY_MAX_LANES = 2048;
struct TinyLaunch {
  // EdDSA lanes
  uint32_t n_lanes;
  const void* d_target;
  void* d_ed;
  uint32_t ed_stride;
  const void *d_qtable, *d_keytab, *d_keyrec;
  KeyCache kc;
  RowOut row;
  void* d_tiny;    // tiny_counter_words(max_batch) zeroed words owned by the context
  void* d_shadow;  // TINY_MAX_LANES x 256 B: the lanes' key bytes + flags for the key pipeline that runs behind the launch
  // proofs (n_proofs = 0: the lanes only -- tmx_eddsa_lanes_device)
  uint32_t n_proofs;
  ProofParams P;
  const void *d_proofs, *d_trusted;
  void* d_lt;
  uint32_t lt_stride;
  void *d_lr, *d_pf, *d_nodes_t, *d_nodes_r, *d_reports;
  // witness rows (d_out = null: reports only)
  const SerializeProgram* S;
  const SerializeSources* src;
  const void *d_lut, *d_wave_sec, *d_seams;
  uint32_t n_seams;
  void* d_out;
  uint32_t mask_inputs;  // sections k_tiny's input role writes
  uint32_t mask_after;   // sections k_tiny_tail's span roles write
  uint32_t mask_tail;    // sections that carry the verdict (k_tiny_tail: span roles up to tail_dep_elem, the per-proof final role from there)
  uint32_t tail_dep_elem;  // first element of a row that depends on the final checks
};
size_t tiny_counter_words(uint32_t max_proofs);
// k_proof as four (step: three) workgroups per proof, the last of them to finish running the checks (tiny.hpp: k_proof_roles)
int launch_proof_roles(const ProofParams& P, uint32_t n_proofs, const void* d_proofs, const void* d_target, const void* d_trusted, void* d_lt,
                       uint32_t lt_stride, void* d_lr, void* d_pf, void* d_nodes_t, void* d_nodes_r, void* d_reports, void* d_tiny, void* stream,
                       void* started = nullptr, void* done = nullptr);
int launch_tiny(const TinyLaunch& T, void* stream, void* started = nullptr, void* done = nullptr);
int launch_tiny_tail(const TinyLaunch& T, void* stream, void* started = nullptr, void* done = nullptr);

}  // namespace tmx
