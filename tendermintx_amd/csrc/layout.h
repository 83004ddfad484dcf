// Device-side buffer layouts shared by the kernels (kernels.hip) and the host API (api.cpp).
// Host structs of the C ABI are in include/tmx.h; everything here is internal to libtmx.
#pragma once
#include <stdint.h>

namespace tmx {

constexpr uint32_t TMX_N_LIMIT = 512;  // largest VALIDATOR_SET_SIZE_MAX (LDS sizing of k_proof)

// ---- compact per-lane / per-proof result buffers written by k_eddsa / k_proof, read by k_serialize
constexpr uint32_t ED_STRIDE = 448;  // digest[64] h[32] pts[10][32] ok u32 decode_ok u32 pad
constexpr uint32_t ED_OFF_DIGEST = 0, ED_OFF_H = 64, ED_OFF_PTS = 96, ED_OFF_OK = 416, ED_OFF_DECODE_OK = 420;

// k_eddsa and k_proof write into ONE record per target lane (TL = ED part | LT part) so that the serializer reads every
// per-lane section from a single source: base pointer and stride are then wave-uniform (SGPRs).
constexpr uint32_t TL_STRIDE = 560, TL_OFF_ED = 0, TL_OFF_LT = 448;
constexpr uint32_t LANE_STRIDE = 112;  // derived per-lane part (LT inside TL; LR = the trusted set's own buffer)
constexpr uint32_t LN_OFF_MARSHAL = 0, LN_OFF_LEAF = 48, LN_OFF_FLAGS = 80, LN_OFF_TOT = 88, LN_OFF_ACC = 96;
// target flags bytes: [0] enabled [1] hash_in_msg [2] is_precommit [3] height_ok [4] round_ok [5] sigdata_ok
// trusted flags bytes: [0] enabled [1] matched

constexpr uint32_t PF_STRIDE = 1920;  // per-proof derived record
constexpr uint32_t PF_OFF_HEADER = 0;
constexpr uint32_t PF_OFF_AUNTS = 32;     // 5 x [4][32]: chain-id, height, validators-hash, X, Y
constexpr uint32_t PF_OFF_PROOFD = 672;   // 5 x (leaf_hash[32] + nodes[4][32]) same order
constexpr uint32_t PF_OFF_HLEAF = 1472;   // 00 08 varint9(height)  (11 B)
constexpr uint32_t PF_OFF_TALLY_T = 1488; // u64 total, acc, scaled_acc, scaled_total
constexpr uint32_t PF_OFF_TALLY_R = 1520;
constexpr uint32_t PF_OFF_VERDICTS = 1552;  // u32 gt_t, gt_r, dist_gt, dist_le
constexpr uint32_t PF_OFF_CHECKS = 1568;    // u32 x 16
constexpr uint32_t PF_OFF_ALLOK = 1632;     // u32
constexpr uint32_t PF_OFF_HEIGHT = 1640;    // u64 header_a height
constexpr uint32_t PF_OFF_CIDLEN = 1648;    // u32 enc chain id length
constexpr uint32_t PF_OFF_HLEN = 1652;      // u32 enc height length
constexpr uint32_t PF_OFF_CID52 = 1664;     // chain-id leaf zero-padded to 52
constexpr uint32_t PF_OFF_LEAFV = 1728;     // header_a leaf 7 padded to 34
constexpr uint32_t PF_OFF_LEAFX = 1776;     // skip: header_b leaf 7 (34) ; step: header_a leaf 4 (72)
constexpr uint32_t PF_OFF_LEAFY = 1856;     // step: header_b leaf 8 (34)
constexpr uint32_t PF_OFF_NB_A = 1656, PF_OFF_NB_B = 1660, PF_OFF_ROUND = 1896;  // copies of proof-record fields (hint elements)

// ---- input record offsets (include/tmx.h structs, as raw bytes)
constexpr uint32_t VR_STRIDE = 256, VR_OFF_PK = 0, VR_OFF_SIG = 32, VR_OFF_MSG = 96, VR_OFF_MLEN = 220, VR_OFF_VLEN = 222,
                   VR_OFF_FLAGS = 223, VR_OFF_POWER = 224;
constexpr uint32_t HR_STRIDE = 48, HR_OFF_PK = 0, HR_OFF_POWER = 32, HR_OFF_VLEN = 40, HR_OFF_FLAGS = 41;
constexpr uint32_t HDR_SIZE = 1136, PR_STRIDE = 2336, PR_OFF_BLOCK_A = 0, PR_OFF_BLOCK_B = 8, PR_OFF_HASH = 16, PR_OFF_ROUND = 48,
                   PR_OFF_NB_A = 56, PR_OFF_NB_B = 60, PR_OFF_HDR_A = 64, PR_OFF_HDR_B = 64 + HDR_SIZE;

// ---- serializer program: the element stream is a list of sections; fixed and per-lane sections are driven by
// look-up tables with one u32 per element (lut_field below); every section reads ONE source buffer
enum : uint32_t { SRC_TARGET = 0, SRC_TRUSTED = 1, SRC_TL = 2, SRC_LR = 3, SRC_PF = 4, SRC_COUNT = 5 };
// One entry describes a bit-field of a 4-byte-aligned dword of the source record (every multi-byte field of the records is 4-aligned):
//   left shift [31:24] | right shift [23:16] | aligned byte offset within the record [15:0]      field = (dword << lsh) >> rsh
// (a field of `width` bits at bit `pos`: lsh = 32 - pos - width, rsh = 32 - width; the whole dword: 0, 0).  Byte-aligned sub-fields of the
// entry are operand selects (SDWA) of the two shifts and of the address add, so an element is one aligned dword load + two shifts with no
// instruction spent on decoding the entry and no branch on the field type.  (Rounds 1-3: width | first bit | 22-bit offset and a
// v_bfe_u32: two shifts for its operands and a select for the whole-dword case -- five instructions per element instead of two.)
enum : uint32_t { W_BIT = 1, W_U8 = 8, W_U16 = 16, W_U32 = 0 };
constexpr uint32_t LUT_OFF_MASK = 0xffffu;
constexpr uint32_t lut_field(uint32_t width, uint32_t byte_off, uint32_t bit_in_byte) {
  return ((32u - (8u * (byte_off & 3u) + bit_in_byte) - (width ? width : 32u)) << 24) | ((32u - (width ? width : 32u)) << 16) | (byte_off & ~3u);
}
static_assert(TL_STRIDE <= LUT_OFF_MASK && PF_STRIDE <= LUT_OFF_MASK && VR_STRIDE <= LUT_OFF_MASK && PR_STRIDE <= LUT_OFF_MASK && HR_STRIDE <= LUT_OFF_MASK &&
                  LANE_STRIDE <= LUT_OFF_MASK,
              "record offsets are 16-bit in the LUT (and record strides 24-bit multiplicands in serialize_pair)");

enum : uint32_t { SEC_LUT = 0, SEC_LINEAR_T = 1, SEC_LINEAR_R = 2 };
struct Section {
  uint32_t elem_start;  // first element of the section within a proof row
  uint32_t lane_elems;  // elements per lane (whole section if n_lanes == 1)
  uint32_t n_lanes;     // 1 for fixed sections
  uint32_t lut_off;     // index into the LUT array (SEC_LUT)
  uint32_t kind;
  uint32_t src;         // SRC_* buffer this section reads
  uint32_t magic;       // ceil(2^32 / lane_elems): lane = mulhi(rel, magic), exact for rel * lane_elems < 2^32
  // resolved per launch (launch_serialize) so that the kernel does no buffer selection:
  uint32_t rec_stride;  // bytes per source record
  uint32_t rec_mul;     // record = proof * rec_mul + lane   (lanes-per-proof for per-lane buffers, 1 for per-proof ones)
  uint32_t pad;
  const uint8_t* base;  // source buffer
};
#ifndef TMX_SER_SPAN
#define TMX_SER_SPAN 256  // elements one serializer wave expands (SerializeProgram::span): TMX_SER_SPAN / 128 coalesced 16-byte stores per thread
#endif
constexpr uint32_t SER_SPAN_ELEMS = TMX_SER_SPAN;
constexpr int MAX_SECTIONS = 12;
struct SerializeProgram {
  Section sec[MAX_SECTIONS];
  uint32_t n_sections;
  uint32_t elem_count;   // elements per proof
  uint32_t elem_stride;  // row stride (even)
  uint32_t n;            // lanes per proof
  uint32_t tree_nodes;   // nodes per validator tree
  uint32_t lane_fast;    // 1: a span that lies inside ONE lane of a per-lane section takes the scalar-lane path (serialize_span; TMX_SER_LANES=0: never)
  uint32_t span;         // elements one wave serializes (128, 256 or 512): SPAN/128 coalesced 16-byte stores per thread, all loads in flight together
  uint32_t wave_prio;    // s_setprio of the row-writer waves (0: none)
  uint32_t rows_major;   // 1: capped launches walk the proofs per span block with the position's LUT words loaded once (k_serialize_rows); 0: k_serialize_few (TMX_SER_ROWS=0)
  uint32_t one_launch_max;  // up to that many proofs an uncapped launch_serialize is ONE launch from the first to the last selected section (the blocks of unselected sections in between exit at once) instead of one per run of adjacent sections
};
// Fused rows (round 6): the row spans that only expand input records (H.2, H.4: 42 % of a skip row) are work items of ONE claim counter; the
// throughput-bound EdDSA kernels (s*B, the table walk) take `per_base` / `per_walk` of them per table addition -- their stores ride between the
// additions of waves that are on the chip anyway -- and a capped sweeper launch on the low-priority stream takes whatever is left.  Every span is
// written by exactly one claimer; which one decides nothing but the schedule.  ctr == null: off.
struct FusedRows {
  uint32_t* ctr;           // this launch's claim counter: item w = span (first_span + w % n_spans) of proof (w / n_spans)
  const uint32_t* lut;
  const uint8_t* wave_sec;
  uint64_t* out;
  uint32_t sec_mask, first_span, n_spans, n_proofs;
  uint32_t per_base, per_walk, pad0, pad1;
};
struct SerializeSources {
  const uint8_t* base[SRC_COUNT];
  const uint8_t* nodes_t;
  const uint8_t* nodes_r;
};

// ---- D.1b, the word fields of the per-target-lane derived values (DESIGN.md "Witness layout"): written straight into the witness rows by the
// EdDSA finish (the first D1B_ED_ELEMS elements of a lane: h, ten coordinates, eddsa_ok) and by k_verdict (the other ten: k_proof's flags
// and prefix sums) when the batch comes through run_eddsa; by the serializer when the lane records come from the caller
constexpr uint32_t D1B_LANE_ELEMS = 99, D1B_ED_ELEMS = 89;
struct RowOut {
  uint64_t* rows;        // first element of the batch's first row, or null: no direct row writes
  uint32_t elem_stride;  // elements between rows
  uint32_t n;            // lanes per proof
  uint32_t d1b_start;    // first element of the D.1b section within a row
};

// ---- Level-2 trace rows (trace.hip): row widths in elements
constexpr uint32_t TR_LADDER_ROW = 65, TR_LADDER_ROWS = 256, TR_SHA512_ROW = 18, TR_SHA256_ROW = 9;

// The validator-set cache of a context (round 5).  marshal + leaf hash of every lane and the fixed-shape tree above them depend on the set
// alone -- (pubkey, voting power, validator_byte_length) of every lane and the number of enabled lanes -- not on the proof: a light client
// re-verifies the same slowly changing sets (reference bin/tendermintx.rs:171), and the 256 proofs of a batch share a handful.  Content
// addressed like the key cache: a 64-bit fingerprint finds a slot, ALL key bytes are compared before its values are used.
//   table[tab_mask + 1]   0 = empty, (slot + 1) = valid, (slot + 1) | SETC_PENDING = claimed and being written
//   state[8]              [0] next never-used slot  [1] sets served from the cache  [2] sets computed  [3] sets inserted
//                         [4] entries on the free list (signed: a pop that finds none takes it below zero for a moment)  [5] free
//                         [6] sets evicted  [7] free
//   freelist[cap]         slots given back by an eviction (behind state[]: same allocation)
//   slot                  u64 fingerprint | u32 nb | u32 varint-msb failures | u32 epoch of the last launch that used it (0: free) | pad to 32 |
//                         root[32] | keys n x 48 B (pubkey, power, vlen) | per-lane values n x 80 B (marshalled validator 48, leaf hash 32) |
//                         tree nodes tree_nodes x 32 B
// LRU at launch granularity (round 6; like the key cache's): a hit or an insert stamps the slot with the launch's epoch; k_setc_evict -- one
// workgroup enqueued behind every k_proof launch on its stream, so nobody else touches the cache while it runs --
// keeps an eighth of the slots free: it evicts the least recently used, rebuilds the table from the surviving slots' fingerprints and puts the
// evicted slots on the free list.  Within a launch slots are never rewritten.  tmx_key_cache_flush empties the cache.
constexpr uint32_t SETC_PENDING = 0x80000000u, SETC_HDR = 64, SETC_KEY = 48, SETC_VAL = 80, SETC_OFF_EPOCH = 16, SETC_STATE_WORDS = 8, SETC_MAX_SLOTS = 256;
struct SetCache {
  uint32_t* table;  // null: no cache (every set is computed)
  uint32_t* state;
  uint8_t* slots;
  uint32_t cap, tab_mask, slot_bytes, pad;
};
constexpr uint32_t setcache_slot_bytes(uint32_t n, uint32_t tree_nodes) { return SETC_HDR + n * (SETC_KEY + SETC_VAL) + tree_nodes * 32; }

struct ProofParams {
  uint32_t kind, n, tree_nodes, chain_id_len;
  uint32_t setc_epoch;  // this launch's stamp for the validator-set cache (never 0)
  uint32_t pad1_;
  uint32_t pad2_;
  uint32_t leaves_done;  // 1: k_leaves has written the marshalled validators and leaf hashes of this batch
  uint64_t skip_max;
  uint8_t chain_id[52];
};

}  // namespace tmx
