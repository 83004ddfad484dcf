// libtmx C ABI (include/tmx.h): context, device scratch, serializer program, launch sequence.
//
// Host side of the drop-in boundary: what SkipOffchainInputs::hint / StepOffchainInputs::hint
// (reference circuits/skip.rs:64-102, circuits/step.rs:56-89) do after the RPC/fixture fetch, expressed as
//   records -> [k_eddsa] -> [k_proof] -> [k_serialize] -> Goldilocks elements (+ tmx_report)
// on one HIP stream.  No CPU compute path exists here: without a usable HIP device every call fails.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "kernels.h"
#include "ntt.h"
#include "poseidon.h"
#include "trace.h"
#include "tmx.h"
#include "value.h"

using namespace tmx;

// why the last context-less call of this thread failed (tmx_comm_unique_id has no context to carry the text): tmx_last_error(NULL)
static thread_local std::string g_tls_err;

static_assert(sizeof(tmx_validator_rec) == VR_STRIDE, "validator record layout");
static_assert(sizeof(tmx_hashfield_rec) == HR_STRIDE, "hash-field record layout");
static_assert(sizeof(tmx_header_rec) == HDR_SIZE, "header record layout");
static_assert(sizeof(tmx_proof_rec) == PR_STRIDE, "proof record layout");
static_assert(sizeof(tmx_report) == 64, "report layout");
static_assert(offsetof(tmx_validator_rec, message_byte_length) == VR_OFF_MLEN, "mlen offset");
static_assert(offsetof(tmx_validator_rec, voting_power) == VR_OFF_POWER, "power offset");
static_assert(offsetof(tmx_hashfield_rec, validator_byte_length) == HR_OFF_VLEN, "vlen offset");
static_assert(offsetof(tmx_proof_rec, header_b) == PR_OFF_HDR_B, "header_b offset");
static_assert(TMX_N_MAX_LIMIT == TMX_N_LIMIT, "n_max limit");
static_assert(sizeof(tmx_validator_value) == VAL_VALIDATOR && sizeof(tmx_hashfield_value) == VAL_HASHFIELD, "typed value: lane structs");
static_assert(sizeof(tmx_skip_inputs_fixed) == VAL_FIXED_SKIP && sizeof(tmx_step_inputs_fixed) == VAL_FIXED_STEP, "typed value: fixed parts");
static_assert(sizeof(tmx_target_lane_derived) == VAL_LANE_T && sizeof(tmx_trusted_lane_derived) == VAL_LANE_R && sizeof(tmx_proof_derived) == VAL_PROOF_D,
              "typed value: derived structs");
static_assert(offsetof(tmx_validator_value, message_byte_length) == 220 && offsetof(tmx_validator_value, voting_power) == 224 &&
                  offsetof(tmx_validator_value, validator_byte_length) == 232 && offsetof(tmx_validator_value, signed_) == 236,
              "tmx_validator_value: k_pack_value writes these offsets");
static_assert(offsetof(tmx_target_lane_derived, eddsa_ok) == ED_OFF_OK && offsetof(tmx_target_lane_derived, marshalled) == TL_OFF_LT + LN_OFF_MARSHAL &&
                  offsetof(tmx_target_lane_derived, leaf_hash) == TL_OFF_LT + LN_OFF_LEAF && offsetof(tmx_target_lane_derived, flags) == TL_OFF_LT + LN_OFF_FLAGS &&
                  offsetof(tmx_target_lane_derived, total_prefix) == TL_OFF_LT + LN_OFF_TOT && offsetof(tmx_target_lane_derived, signed_prefix) == TL_OFF_LT + LN_OFF_ACC,
              "tmx_target_lane_derived mirrors the context's lane record");
static_assert(offsetof(tmx_trusted_lane_derived, flags) == LN_OFF_FLAGS && offsetof(tmx_trusted_lane_derived, matched_prefix) == LN_OFF_ACC, "tmx_trusted_lane_derived");
static_assert(offsetof(tmx_proof_derived, height_leaf) == PF_OFF_HLEAF - PF_OFF_PROOFD && offsetof(tmx_proof_derived, tally_target) == PF_OFF_TALLY_T - PF_OFF_PROOFD &&
                  offsetof(tmx_proof_derived, tally_trusted) == PF_OFF_TALLY_R - PF_OFF_PROOFD && offsetof(tmx_proof_derived, verdicts) == PF_OFF_VERDICTS - PF_OFF_PROOFD &&
                  offsetof(tmx_proof_derived, checks) == PF_OFF_CHECKS - PF_OFF_PROOFD && offsetof(tmx_proof_derived, all_ok) == PF_OFF_ALLOK - PF_OFF_PROOFD &&
                  offsetof(tmx_proof_derived, height) == PF_OFF_HEIGHT - PF_OFF_PROOFD,
              "tmx_proof_derived mirrors the context's per-proof record");

// ------------------------------------------------------------------------------------------------ element layout
// Declarative description of the witness row: hint section H in the field order of VerifySkipVariable<N> /
// VerifyStepVariable<N> (reference circuits/variables.rs:91-120) with the element widths of the plonky2x
// variable types (ByteVariable = 8 big-endian bits, U32/Variable/Bool = 1, U64 = 2 LE limbs, U256 = 8 LE limbs),
// followed by the derived section D (DESIGN.md "Witness layout").
namespace {

struct LutBuilder {  // offsets are relative to the section's single source record
  std::vector<uint32_t> v;
  void bytes(uint32_t off, uint32_t n) {  // ByteVariable: 8 elements per byte, most significant bit first
    for (uint32_t i = 0; i < n; i++)
      for (uint32_t k = 0; k < 8; k++) v.push_back(lut_field(W_BIT, off + i, 7 - k));
  }
  void u8(uint32_t off) { v.push_back(lut_field(W_U8, off, 0)); }
  void u16(uint32_t off) { v.push_back(lut_field(W_U16, off, 0)); }  // 2-aligned inside its dword
  void u32(uint32_t off) { v.push_back(lut_field(W_U32, off, 0)); }  // 4-aligned
  void u64(uint32_t off) { u32(off); u32(off + 4); }
  void u256(uint32_t off) { for (uint32_t k = 0; k < 8; k++) u32(off + 4 * k); }
  void flag0(uint32_t off) { v.push_back(lut_field(W_BIT, off, 0)); }
};

uint32_t tree_nodes(uint32_t n) {
  uint32_t c = 0;
  while (n > 1) { n = (n + 1) / 2; c += n; }
  return c;
}

// MerkleInclusionProofVariable<4, LEAF>: proof[4] (Bytes32 each) then leaf bytes  (variables.rs:58-62)
void emit_inclusion_proof(LutBuilder& L, int q, uint32_t leaf_src_off, uint32_t leaf_size) {
  L.bytes(PF_OFF_AUNTS + 128 * q, 128);
  L.bytes(leaf_src_off, leaf_size);
}
// derived: leaf hash + the four path nodes of proof q
void emit_proof_d(LutBuilder& L, int q) { L.bytes(PF_OFF_PROOFD + 160 * q, 160); }

struct Program {
  SerializeProgram sp;
  std::vector<uint32_t> lut;
  uint32_t mask_inputs = 0, mask_proof = 0, mask_final = 0, mask_tail = 0, mask_p1 = 0, mask_leaves = 0;  // sections by readiness: inputs only / after k_proof / after EdDSA / after k_verdict
  std::vector<uint32_t> seam_waves;  // indices of the spans that straddle a boundary (0xff entries of wave_sec)
  std::vector<uint8_t> wave_sec;  // per sp.span-element span of a row: its section, or 0xff if it straddles a boundary / the row end
  uint32_t hint_elems;
  uint32_t d1b_start = 0;  // first element of D.1b in a row (the EdDSA finish writes that section directly: RowOut)
  uint32_t mask_hint = 0, mask_derived = 0;  // sections of H / of D (tmx_witness_batch_opts: a caller may ask for one of them only)
  uint32_t tail_dep_elem = 0;  // first element of a row whose value is written by the final checks (verdicts, check words, all_ok): the small-launch tail
};

Program build_program(int kind, uint32_t n) {
  Program P;
  std::memset(&P.sp, 0, sizeof P.sp);
  LutBuilder L;
  const bool skip = kind == TMX_KIND_SKIP;
  const uint32_t tn = tree_nodes(n);
  uint32_t elem = 0;
  // ready: 0 = needs only the input records, 1 = needs k_proof, 2 = needs the EdDSA kernels (and k_proof), 3 = needs k_verdict,
  // 4 = needs the leaves (k_leaves, or k_proof) and phase 1 (not k_ed_fin), 5 = needs only the leaves
  auto add_section = [&](uint32_t lane_elems, uint32_t n_lanes, uint32_t lut_off, uint32_t kind_, uint32_t src, int ready) {
    (ready == 0 ? P.mask_inputs : ready == 1 ? P.mask_proof : ready == 2 ? P.mask_final : ready == 4 ? P.mask_p1 : ready == 5 ? P.mask_leaves : P.mask_tail) |=
        1u << P.sp.n_sections;
    Section& s = P.sp.sec[P.sp.n_sections++];
    s.elem_start = elem; s.lane_elems = lane_elems; s.n_lanes = n_lanes; s.lut_off = lut_off; s.kind = kind_; s.src = src;
    s.magic = (uint32_t)((0x100000000ull + lane_elems - 1) / lane_elems);  // lane = mulhi(rel, magic); rel * lane_elems < 2^32 here
    s.rec_stride = 0; s.rec_mul = 0; s.pad = 0; s.base = nullptr;
    elem += lane_elems * n_lanes;
  };
  uint32_t mark;

  // H.1 target_header / next_header : Bytes32
  mark = (uint32_t)L.v.size();
  L.bytes(PF_OFF_HEADER, 32);
  add_section((uint32_t)L.v.size() - mark, 1, mark, SEC_LUT, SRC_PF, 1);

  // H.2 validators[N] : ValidatorVariable (variables.rs:69-79) = pubkey, signature{r, s}, message[124],
  //     message_byte_length, voting_power, validator_byte_length, signed
  mark = (uint32_t)L.v.size();
  L.bytes(VR_OFF_PK, 32);
  L.bytes(VR_OFF_SIG, 32);
  L.u256(VR_OFF_SIG + 32);
  L.bytes(VR_OFF_MSG, 124);
  L.u16(VR_OFF_MLEN);
  L.u64(VR_OFF_POWER);
  L.u8(VR_OFF_VLEN);
  L.flag0(VR_OFF_FLAGS);
  add_section((uint32_t)L.v.size() - mark, n, mark, SEC_LUT, SRC_TARGET, 0);

  // H.3 nb_validators, round, ChainIdProofVariable, HeightProofVariable, validators-hash proof, then
  //     skip: trusted nb + trusted validators-hash proof ; step: last_block_id proof + prev next_validators_hash proof
  mark = (uint32_t)L.v.size();
  L.u32(PF_OFF_NB_A);
  L.u64(PF_OFF_ROUND);
  L.bytes(PF_OFF_AUNTS + 128 * 0, 128);   // chain_id_proof.proof
  L.u32(PF_OFF_CIDLEN);                   // enc_chain_id_byte_length
  L.bytes(PF_OFF_CID52, 52);              // chain_id
  L.bytes(PF_OFF_AUNTS + 128 * 1, 128);   // height_proof.proof
  L.u32(PF_OFF_HLEN);                     // enc_height_byte_length
  L.u64(PF_OFF_HEIGHT);                   // height
  emit_inclusion_proof(L, 2, PF_OFF_LEAFV, 34);
  if (skip) {
    L.u32(PF_OFF_NB_B);
    emit_inclusion_proof(L, 3, PF_OFF_LEAFX, 34);
  } else {
    emit_inclusion_proof(L, 3, PF_OFF_LEAFX, 72);
    emit_inclusion_proof(L, 4, PF_OFF_LEAFY, 34);
  }
  add_section((uint32_t)L.v.size() - mark, 1, mark, SEC_LUT, SRC_PF, 1);

  // H.4 skip: trusted_header_validator_hash_fields[N] : ValidatorHashFieldVariable (variables.rs:82-88)
  if (skip) {
    mark = (uint32_t)L.v.size();
    L.bytes(HR_OFF_PK, 32);
    L.u64(HR_OFF_POWER);
    L.u8(HR_OFF_VLEN);
    add_section((uint32_t)L.v.size() - mark, n, mark, SEC_LUT, SRC_TRUSTED, 0);
  }
  P.hint_elems = elem;

  // D.1a per target lane: the byte fields (marshalled validator, leaf hash: k_proof; SHA-512 digest: phase 1).  Their own section so that
  // they are written while the table walk runs; only D.1b waits for k_ed_fin.
  mark = (uint32_t)L.v.size();
  L.bytes(TL_OFF_LT + LN_OFF_MARSHAL, 46);
  L.bytes(TL_OFF_LT + LN_OFF_LEAF, 32);
  L.bytes(TL_OFF_ED + ED_OFF_DIGEST, 64);
  add_section((uint32_t)L.v.size() - mark, n, mark, SEC_LUT, SRC_TL, 4);
  // D.1b per target lane: h, the ten coordinates, the EdDSA verdict, the six flags, the two prefix sums
  mark = (uint32_t)L.v.size();
  L.u256(TL_OFF_ED + ED_OFF_H);
  for (int p = 0; p < 10; p++) L.u256(TL_OFF_ED + ED_OFF_PTS + 32 * p);
  L.u32(TL_OFF_ED + ED_OFF_OK);
  for (int f = 0; f < 6; f++) L.u8(TL_OFF_LT + LN_OFF_FLAGS + f);
  L.u64(TL_OFF_LT + LN_OFF_TOT);
  L.u64(TL_OFF_LT + LN_OFF_ACC);
  P.d1b_start = elem;
  if ((uint32_t)L.v.size() - mark != D1B_LANE_ELEMS) P.d1b_start = 0xffffffffu;  // (caught by tmx_ctx_create: layout.h and this list must agree)
  add_section((uint32_t)L.v.size() - mark, n, mark, SEC_LUT, SRC_TL, 2);

  // D.2a / D.2b per trusted lane: byte fields (marshalled validator, leaf hash: ready with the leaves), word fields (flags, prefix sums: k_proof)
  if (skip) {
    mark = (uint32_t)L.v.size();
    L.bytes(LN_OFF_MARSHAL, 46);
    L.bytes(LN_OFF_LEAF, 32);
    add_section((uint32_t)L.v.size() - mark, n, mark, SEC_LUT, SRC_LR, 5);
    mark = (uint32_t)L.v.size();
    L.u8(LN_OFF_FLAGS);
    L.u8(LN_OFF_FLAGS + 1);
    L.u64(LN_OFF_TOT);
    L.u64(LN_OFF_ACC);
    add_section((uint32_t)L.v.size() - mark, n, mark, SEC_LUT, SRC_LR, 1);
  }
  // D.3 / D.4 tree nodes
  if (tn) {
    add_section(tn * 256, 1, 0, SEC_LINEAR_T, SRC_PF, 1);
    if (skip) add_section(tn * 256, 1, 0, SEC_LINEAR_R, SRC_PF, 1);
  }
  // D.5 header proofs, tallies, checks, verdict
  mark = (uint32_t)L.v.size();
  emit_proof_d(L, 0);
  L.bytes(PF_OFF_HLEAF, 11);
  emit_proof_d(L, 1);
  emit_proof_d(L, 2);
  emit_proof_d(L, 3);
  if (!skip) emit_proof_d(L, 4);
  for (int k = 0; k < 4; k++) L.u64(PF_OFF_TALLY_T + 8 * k);
  P.tail_dep_elem = elem + ((uint32_t)L.v.size() - mark);
  L.u32(PF_OFF_VERDICTS);
  if (skip) {
    for (int k = 0; k < 4; k++) L.u64(PF_OFF_TALLY_R + 8 * k);
    L.u32(PF_OFF_VERDICTS + 4);
    L.u32(PF_OFF_VERDICTS + 8);
    L.u32(PF_OFF_VERDICTS + 12);
  }
  const int n_checks = skip ? 13 : 15;
  for (int k = 0; k < n_checks; k++) L.u32(PF_OFF_CHECKS + 4 * k);
  L.u32(PF_OFF_ALLOK);
  add_section((uint32_t)L.v.size() - mark, 1, mark, SEC_LUT, SRC_PF, 3);
  P.mask_tail |= 1u << 31;  // boundary waves are written last, when every source is ready
  for (uint32_t k = 0; k < P.sp.n_sections; k++) (P.sp.sec[k].elem_start < P.hint_elems ? P.mask_hint : P.mask_derived) |= 1u << k;

  P.sp.elem_count = elem;
  P.sp.elem_stride = (elem + 15) & ~15u;  // rows start on a 128-byte line
  P.sp.n = n;
  P.sp.tree_nodes = tn;
  P.lut = std::move(L.v);
  const uint32_t span = SER_SPAN_ELEMS;  // elements per wave (layout.h TMX_SER_SPAN): 128 and 512 measured slower on MI355X (round 1; round 6: docs/experiments.md)
  P.sp.span = span;
  for (uint32_t w = 0; w * span < P.sp.elem_stride; w++) {
    const uint32_t first = w * span, last = first + span - 1;
    uint8_t sec = 0xff;
    if (last < P.sp.elem_count)
      for (uint32_t s = 0; s < P.sp.n_sections; s++) {
        const uint32_t lo = P.sp.sec[s].elem_start, hi = lo + P.sp.sec[s].lane_elems * P.sp.sec[s].n_lanes;
        if (first >= lo && last < hi) sec = (uint8_t)s;
      }
    if (sec == 0xff) P.seam_waves.push_back(w);
    P.wave_sec.push_back(sec);
  }
  return P;
}

}  // namespace

// ------------------------------------------------------------------------------------------------ context
constexpr int EV_RING_DECL = 128;
constexpr uint64_t THROUGHPUT_LANES = 81920;  // proofs x validators from which run_batch schedules for throughput (see beside_chain_wgs)
// Schedule knobs, read ONCE at context creation (getenv is not safe against a concurrent setenv, and the enqueue path is
// latency-critical).  Each selects a schedule, never a value (tests/test_gpu_parity.py::test_schedule_knobs_give_the_same_bits); the A/B
// harness that measured the alternatives no longer in the product lives in tools/ (DESIGN.md appendix "measured and dropped").
struct Knobs {
  uint32_t dedup_mode = 1;   // TMX_DEDUP=0|1|2: never use per-key tables / automatic / whenever they fit
  bool key_cache = true;     // TMX_KEY_CACHE=0: the per-key tables do not survive a call (every call cold)
  uint32_t key_cache_keys = 0;  // TMX_KEY_CACHE_KEYS: capacity of the cache in keys (0: by max_batch * n_max)
  bool ser_split = true;     // TMX_SER_SPLIT=0: one k_serialize launch at the end of the step (times the kernel on its own)
  int leaves = -1;           // TMX_LEAVES=0|1: leaf hashes as a launch of their own in front of k_proof (default: from 131072 lanes)
  int walk_parts = -1;       // TMX_WALK_PARTS=0|1: the table walk follows the table build part by part (default: from 65536 lanes)
  bool ext_events = true;    // TMX_EXT_EVENTS=0: record packets instead of completion signals on the chain kernels
  int warm_schedule = -1;    // TMX_SCHEDULE=warm|cold: the EdDSA schedule for resident / new keys (default: by what the last launch saw)
  int phase1_max = -1;       // TMX_PHASE1_MAX=<lanes>: up to that many lanes the warm schedule runs s*B as a role of the hash launch (default 16384: 128 proofs at N = 128)
  bool join1 = true;         // TMX_JOIN1=0: the small form's tail behind two wait packets (k_proof, the new-key lanes' finish) instead of one joined on side2
  long tiny_max = -1;        // TMX_TINY_MAX=<lanes>: the two-launch small path up to that many lanes (default 1536, at most TINY_MAX_LANES = 2048)
  int one_launch_max = -1;   // TMX_SER_ONE_LAUNCH=<proofs>: up to that many proofs the uncapped serializer calls are one launch each (default 8)
  bool proof_roles = true;   // TMX_PROOF_ROLES=0: k_proof as one workgroup per proof (the round-3 kernel) instead of four role workgroups
  int tail_wide = 1;         // TMX_TAIL_WIDE=0: k_verdict -> D.5 -> the seam spans as three launches / one workgroup per proof (rounds 2 - 5) instead of ONE launch of independent workgroups (k_verdict_tail_wide)
  int p1_early = -1;         // TMX_P1_EARLY=0|1|2: D.1a behind k_proof's sections on the side stream (1: on a capped grid, 2: full grid) instead of behind k_ed_fin (default: capped, from 131072 lanes)
  int hash_first = -1;       // TMX_HASH_FIRST=0|1: warm schedule with the hash role in front of the dedup (which moves to side2); default: see run_eddsa
  bool walk_split = true;    // TMX_WALK_SPLIT=0: the warm schedule's table walk as ONE launch behind the table build (the round-4 form) instead of resident lanes at once + new-key lanes behind the build
  bool compact = true;       // TMX_COMPACT=0: every lane through the EdDSA kernels (round 4), also the ones that did not sign
  bool set_cache = true;     // TMX_SET_CACHE=0: k_proof computes the leaves and the tree of every validator set of every proof (round 4)
  bool epi_late = true;      // TMX_EPI_LATE=0: the cache epilogue of a split warm batch in front of the input sections on side3 (the first round-5 form)
  bool ser_lanes = true;     // TMX_SER_LANES=0: no scalar-lane path for the spans that lie inside one lane of a per-lane section (serialize_span)
  int fused_base = -1, fused_walk = -1;  // TMX_FUSED_ROWS=<b>[,<w>]: input-only row spans a wave of s*B / of the resident walk takes per table addition
                                         // (fused rows, layout.h FusedRows); 0,0 = off: the sections as capped launches of their own (round 5); default: see run_batch
  long inputs_first_min = -1; // TMX_INPUTS_FIRST=<lanes>: from that many lanes on the split warm schedule enqueues the input sections IN FRONT of the new-key pipeline on the low-priority stream (0: never)
  bool ser_rows = true;      // TMX_SER_ROWS=0: the capped row-writer launches grid-stride over (proof, block) (k_serialize_few, round 5) instead of proof-major (k_serialize_rows)
  int writer_prio = -1;      // TMX_WRITER_PRIO=<0..3>: s_setprio of the row-writer waves (default: by size, run_batch)
  int tail_aside_min = -1;   // TMX_TAIL_ASIDE_MIN=<lanes>: from how many lanes on the verdict + its sections leave the caller's stream (default 28672)
  int few_wgs = 0;           // TMX_FEW_WGS=<n>: workgroups of the serializer launches beside the chain (A/B; 0: by size, run_batch)
  int tiny = -1;             // TMX_TINY=0|1: never / always (also under a forced TMX_SCHEDULE) take the two-launch small path for <= TINY_MAX_LANES lanes
};
static Knobs read_knobs() {
  Knobs k;
  const char* v;
  if ((v = std::getenv("TMX_DEDUP")) && v[0] >= '0' && v[0] <= '2') k.dedup_mode = (uint32_t)(v[0] - '0');
  k.key_cache = !((v = std::getenv("TMX_KEY_CACHE")) && v[0] == '0');
  if ((v = std::getenv("TMX_KEY_CACHE_KEYS")) && std::atoll(v) > 0) k.key_cache_keys = (uint32_t)std::min<long long>(std::atoll(v), 1 << 20);
  k.ser_split = !((v = std::getenv("TMX_SER_SPLIT")) && v[0] == '0');
  k.leaves = (v = std::getenv("TMX_LEAVES")) ? (v[0] != '0' ? 1 : 0) : -1;
  k.walk_parts = (v = std::getenv("TMX_WALK_PARTS")) ? (v[0] == '1' ? 1 : 0) : -1;
  k.ext_events = !((v = std::getenv("TMX_EXT_EVENTS")) && v[0] == '0');
  k.warm_schedule = (v = std::getenv("TMX_SCHEDULE")) ? (v[0] == 'w' ? 1 : 0) : -1;
  k.tiny = (v = std::getenv("TMX_TINY")) ? (v[0] != '0' ? 1 : 0) : -1;
  k.walk_split = !((v = std::getenv("TMX_WALK_SPLIT")) && v[0] == '0');
  k.proof_roles = !((v = std::getenv("TMX_PROOF_ROLES")) && v[0] == '0');
  if ((v = std::getenv("TMX_SER_ONE_LAUNCH"))) k.one_launch_max = std::atoi(v);
  if ((v = std::getenv("TMX_TINY_MAX"))) k.tiny_max = std::atol(v);
  if ((v = std::getenv("TMX_JOIN1"))) k.join1 = v[0] != '0';
  if ((v = std::getenv("TMX_PHASE1_MAX"))) k.phase1_max = std::atoi(v);
  k.hash_first = (v = std::getenv("TMX_HASH_FIRST")) ? (v[0] != '0' ? 1 : 0) : -1;
  if ((v = std::getenv("TMX_TAIL_WIDE"))) k.tail_wide = v[0] != '0' ? 1 : 0;
  k.p1_early = (v = std::getenv("TMX_P1_EARLY")) && v[0] >= '0' && v[0] <= '2' ? v[0] - '0' : -1;
  k.compact = !((v = std::getenv("TMX_COMPACT")) && v[0] == '0');
  k.set_cache = !((v = std::getenv("TMX_SET_CACHE")) && v[0] == '0');
  k.epi_late = !((v = std::getenv("TMX_EPI_LATE")) && v[0] == '0');
  if ((v = std::getenv("TMX_FEW_WGS"))) k.few_wgs = std::atoi(v);
  if ((v = std::getenv("TMX_TAIL_ASIDE_MIN"))) k.tail_aside_min = std::atoi(v);
  if ((v = std::getenv("TMX_WRITER_PRIO"))) k.writer_prio = std::atoi(v);
  k.ser_rows = !((v = std::getenv("TMX_SER_ROWS")) && v[0] == '0');
  if ((v = std::getenv("TMX_INPUTS_FIRST"))) k.inputs_first_min = std::atol(v);
  if ((v = std::getenv("TMX_FUSED_ROWS"))) {
    k.fused_base = std::atoi(v);
    const char* comma = std::strpbrk(v, ",:");
    k.fused_walk = comma ? std::atoi(comma + 1) : 0;
    if (k.fused_base < 0) k.fused_base = 0;
    if (k.fused_walk < 0) k.fused_walk = 0;
  }
  if ((v = std::getenv("TMX_SER_LANES"))) k.ser_lanes = std::atoi(v) != 0;
  return k;
}

struct tmx_ctx {
  tmx_config cfg;
  Knobs knobs;
  std::string err;
  hipStream_t stream = nullptr;
  hipStream_t side = nullptr;  // k_proof runs here, concurrently with the EdDSA kernels of the caller's stream
  hipEvent_t ev_join = nullptr;
  hipEvent_t ev_side[EV_RING_DECL][4] = {};
  hipEvent_t ev_hash = nullptr;  // phase 1 is done: the SHA-512 digest and h of every lane are in the lane records
  bool ev_hash_recorded = false;
  hipEvent_t ev_leaves = nullptr;
  hipEvent_t ev_tail = nullptr, ev_fork2 = nullptr, ev_hash_clean = nullptr, ev_keys = nullptr;
  bool fin_done_attached = false;
  void* fin_done = nullptr;  // set by run_batch around the EdDSA producer: the event k_ed_fin signals
  RowOut row = {};           // set by run_batch around the EdDSA producer: where the finish writes D.1b into the rows (or null)
  FusedRows fused = {};      // set by run_batch around the EdDSA producer: the input-only row spans its throughput kernels carry (ctr null: none)
  SerializeProgram fused_prog = {};
  uint32_t* d_span_ctr = nullptr;  // two claim counters (a fused launch uses one, its sweeper zeroes the other for the next)
  uint32_t span_parity = 0;
  bool span_ctr_dirty = false;     // a fused launch failed half-way: both counters are zeroed before the next one
  hipEvent_t ev_direct = nullptr;  // the table-free lanes of a launch are done (side2)
  volatile uint32_t* h_hint = nullptr;  // page-locked, written by k_kc_epilogue: [0] launches committed, [1] new keys of the last one
  hipEvent_t ev_part[4] = {};
  bool have_streams = false;
  // ring of HIP-event sets: one set (TMX_N_KERNELS + 1 events) per enqueued batch, so that kernel durations can be
  // averaged over a whole timed region afterwards without synchronising inside it
  static constexpr int EV_RING = EV_RING_DECL;
  hipEvent_t ev[EV_RING][4] = {};
  uint64_t n_calls = 0;
  hipStream_t last_stream = nullptr;  // stream and end event of the previous batch (cross-stream callers are ordered behind it)
  bool last_stream_valid = false;
  hipEvent_t ev_done = nullptr;
  hipEvent_t ev_value = nullptr;  // end of the last typed-value batch (its k_pack_value launch)
  hipEvent_t ev_base = nullptr;   // s*B of the batch being enqueued is done (the split warm schedule: its own launch on side2)
  // the EdDSA schedule of the batch being enqueued, decided ONCE (the hint it looks at lives in host memory the device writes)
  struct EdPlan { bool valid, tiny, warm, hash_first, sb_with_hash, split; } plan = {};
  bool epilogue_pending = false;  // (TMX_EPI_LATE) the epilogue of the batch being enqueued goes behind the input sections on side3
  EdQuad pending_q;
  int32_t last_kind = -1;       // kind and size of the last Level-1 batch (tmx_trace_rows_device reads its lane records)
  uint32_t last_n_proofs = 0;
  Program prog[2];
  void* d_lut[2] = {nullptr, nullptr};
  void* d_wave_sec[2] = {nullptr, nullptr};
  void* d_seams[2] = {nullptr, nullptr};
  void* d_table = nullptr;
  void *d_qtable = nullptr, *d_pre = nullptr, *d_mulout = nullptr;
  // EdDSA stage: the launch's own dedup structures, the key records (cache slots, then one per lane for keys without a slot), the
  // anchor scratch of the tables being built, and the persistent key cache
  // the lanes that did not sign: one precomputed record for all of them (dummy_record), the dense list of the others per launch
  void *d_live = nullptr, *d_dummy_ed = nullptr, *d_dummy_in = nullptr;
  bool dummy_ready = false;
  SetCache setc = {};   // the validator-set cache (layout.h); table == nullptr: off (TMX_SET_CACHE=0)
  uint32_t setc_epoch = 1;  // LRU stamp of the next k_proof launch (never 0: a slot's 0 means free)
  void *d_hash = nullptr, *d_cnt = nullptr, *d_owner_of = nullptr, *d_slot_of_owner = nullptr, *d_slot_of_uid = nullptr, *d_owners = nullptr,
       *d_keyrec = nullptr, *d_anchors = nullptr, *d_keytab = nullptr;
  KeyCache kc = {};
  uint32_t hash_mask = 0;
  hipStream_t side2 = nullptr;  // new-key pipeline, concurrent with phase 1
  hipStream_t side3 = nullptr;  // early serialization of the input-only sections
  hipEvent_t ev_join3 = nullptr;
  uint32_t parity = 0;  // which of the two counter sets this launch uses
  void* d_commit = nullptr;    // scratch of tmx_trace_commit_device: columns | extended columns | tree levels (grows on demand)
  size_t commit_bytes = 0;
  hipEvent_t ev_commit[4] = {};
  void* comm = nullptr;        // ncclComm_t of this context's device (tmx_comm_create), or null
  bool comm_aborted = false;   // a local failure (or a peer's) in front of a collective aborted the communicator: tmx_comm_create again
  uint32_t comm_rank = 0, comm_world = 1;
  void* d_tiny = nullptr;      // counters of the small-launch path (kernels.h: tiny_counter_words), zero between launches
  void* d_shadow = nullptr;    // key bytes + flags of the lanes of a small launch (TINY_MAX_LANES records): what its key pipeline reads
  bool slot_tiny[EV_RING_DECL] = {};  // which event sets of the ring belong to small launches (their four events mark other points)
  uint64_t last_lanes = 0;
  // scratch sized for cfg.max_batch proofs
  void *d_ed = nullptr, *d_tl = nullptr, *d_lr = nullptr, *d_pf = nullptr, *d_nodes_t = nullptr, *d_nodes_r = nullptr, *d_reports = nullptr;
  // staging for the host-buffer entry points
  void *d_in_proofs = nullptr, *d_in_targets = nullptr, *d_in_trusteds = nullptr, *d_out = nullptr;
  uint64_t d_out_elems = 0;
  hipEvent_t ev_trace[17] = {};  // ladder segments done (up to 16) + [16] the side stream's pass 2 done
  hipEvent_t ev_trace_rest[2] = {};  // the other sections beside the ladders: fork, join
  void* d_trace_tmp = nullptr;  // projective ladder points between the two passes of the Level-2 ladder kernels (allocated on first use)
  void* d_pack = nullptr;  // dense / narrowed rows for tmx_witness_batch_opts (allocated on first use)
  void* d_val_lut[2] = {nullptr, nullptr};  // fixed-part gather tables of the typed value (value.h), per kind
  void* d_value = nullptr;  // device staging of tmx_inputs_value_batch when `out` is not page-locked (allocated on first use)
  uint64_t d_value_bytes = 0;
  uint64_t d_pack_bytes = 0;
  uint32_t sections = 3;  // TMX_SEC_* of the batch being enqueued
  // Goldilocks NTT: twiddle tables per transform size (built on first use), scratch for the four-step split / LDE
  void* d_ntt_w[TMX_NTT_MAX_LOG + 1] = {};
  void* d_ntt_m[2][TMX_NTT_MAX_LOG + 1] = {};  // four-step twiddle matrices omega_N^(+- n2 k1) (forward, inverse), built on first use
  void* d_lde_s[TMX_NTT_MAX_LOG + 1] = {};  // coset-LDE scale vectors S[i] = shift^i / N per column length (built on first use, dropped with the domain)
  void* d_ntt_tmp = nullptr;
  size_t ntt_tmp_bytes = 0;
  // NTT domain: primitive 2^32-th root of unity and coset shift.  Default: the constants recalled from plonky2's GoldilocksField
  // (POWER_OF_TWO_GENERATOR, MULTIPLICATIVE_GROUP_GENERATOR; self-consistent: the first is the second to the (p-1)/2^32).
  uint64_t ntt_root = 7277203076849721926ull, ntt_shift = 14293326489335486720ull;
  // Poseidon: round constants | MDS circulant | MDS diagonal, host copy and device copy (uploaded on first use / on change)
  std::vector<uint64_t> pos_consts;
  void* d_pos_consts = nullptr;
  bool pos_dirty = true;
  int pos_mode = POS_MODE_SMALL;  // poseidon.h POS_MODE_*: chosen from the MDS entries when the tables are uploaded
  bool pos_rc_injected = false;  // round constants came from tmx_poseidon_set_constants (the defaults are NOT plonky2's table)
};

static int32_t fail(tmx_ctx* c, int32_t st, const std::string& msg) {
  if (c) c->err = msg;
  return st;
}
#define HIPCK(c, call)                                                                                   \
  do {                                                                                                    \
    hipError_t e_ = (call);                                                                               \
    if (e_ != hipSuccess) return fail(c, TMX_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
  } while (0)

static ProofParams proof_params(const tmx_ctx* c, int32_t kind, bool leaves_done) {
  ProofParams P;
  std::memset(&P, 0, sizeof P);
  P.leaves_done = leaves_done ? 1u : 0u;
  P.kind = (uint32_t)kind; P.n = c->cfg.n_max; P.tree_nodes = tree_nodes(c->cfg.n_max); P.chain_id_len = c->cfg.chain_id_len;
  P.skip_max = c->cfg.skip_max;
  P.setc_epoch = c->setc_epoch;
  std::memcpy(P.chain_id, c->cfg.chain_id, sizeof P.chain_id);
  return P;
}

#ifndef TMX_FUSED_BASE_DEFAULT
#define TMX_FUSED_BASE_DEFAULT 0  // (spans per table addition of s*B / of the resident walk; set by measurement: docs/experiments.md round 6)
#define TMX_FUSED_WALK_DEFAULT 0
#endif
#ifndef TMX_HASH_FIRST_MAX
#define TMX_HASH_FIRST_MAX 16384u  // measured: 64 / 128 proofs at N = 128 -3 %, 256 +2 %, 1024 +6 % (the dedup beside the hash role on a full chip)
#endif
// which EdDSA schedule a launch of n_lanes takes (run_eddsa has the graphs).  in_batch: the launch is the EdDSA stage of run_batch, which
// can put s*B first on its low-priority stream.
static tmx_ctx::EdPlan ed_plan(const tmx_ctx* c, uint32_t n_lanes, bool in_batch) {
  const Knobs& K = c->knobs;
  tmx_ctx::EdPlan P = {};
  P.valid = true;
  P.tiny = n_lanes != 0 && n_lanes <= 512;
  P.warm = K.warm_schedule >= 0 ? K.warm_schedule != 0
                                : (c->kc.persist && K.dedup_mode != 0 && c->h_hint && c->h_hint[0] != 0 && c->h_hint[1] == 0);
  P.hash_first = P.warm && !P.tiny && n_lanes != 0 && (K.hash_first >= 0 ? K.hash_first != 0 : n_lanes <= TMX_HASH_FIRST_MAX);
  P.sb_with_hash = P.tiny || (K.phase1_max >= 0 ? n_lanes <= (uint32_t)K.phase1_max : n_lanes <= 16384);
  // the warm schedule split by residency (run_eddsa_split): only as the EdDSA stage of a batch (run_batch reorders its low-priority stream for it)
  P.split = in_batch && K.walk_split && P.warm && !P.tiny && n_lanes != 0 && K.dedup_mode != 0 && c->kc.cap != 0;
  return P;
}

// Launch sequence of one batch.  Caller's stream s: [ev0] EdDSA kernels [ev1] k_serialize of the EdDSA-dependent section [ev3]
//                               side:   (after ev0) [side0] k_proof [side1], then the sections that only need it -> ev_join
//                               side3:  (after ev0) the sections that only expand the input records -> ev_join3
//                               side2:  the new-key pipeline of the EdDSA stage; the tail: k_verdict + the sections that carry it -> ev_tail
// k_proof does not depend on the EdDSA results, so it overlaps with them; `ed_producer` enqueues whatever fills the ED part of
// c->d_tl on s (the EdDSA kernels, or a strided copy of caller-provided lane records).
template <typename EdProducer>
static int32_t run_batch(tmx_ctx* c, int32_t kind, uint32_t n_proofs, const void* d_proofs, const void* d_targets, const void* d_trusteds,
                         void* d_out_elems, void* d_reports, hipStream_t s, bool eddsa_writes_rows, EdProducer ed_producer) {
  const uint32_t n = c->cfg.n_max;
  const Knobs& K = c->knobs;
  uint8_t* tl = reinterpret_cast<uint8_t*>(c->d_tl);
  void* reports = d_reports ? d_reports : c->d_reports;
  const uint64_t slot = c->n_calls % tmx_ctx::EV_RING;
  hipEvent_t* ev = c->ev[slot];
  hipEvent_t* evs = c->ev_side[slot];
  SerializeSources src;
  std::memset(&src, 0, sizeof src);
  src.base[SRC_TARGET] = (const uint8_t*)d_targets; src.base[SRC_TRUSTED] = (const uint8_t*)d_trusteds;
  src.base[SRC_TL] = tl; src.base[SRC_LR] = (const uint8_t*)c->d_lr; src.base[SRC_PF] = (const uint8_t*)c->d_pf;
  src.nodes_t = (const uint8_t*)c->d_nodes_t; src.nodes_r = (const uint8_t*)c->d_nodes_r;
  Program& prog = c->prog[kind];
  // Throughput regime (round 6, profiles/r06_throughput_regime_ab.txt): from 512 proofs x 128 on a step is a VALU-bound chain plus an HBM-bound tail
  // that overlap badly -- whatever shortens one lengthens the other.  What moves the sum: the row writers' short waves at wave priority 3 (they
  // issue a dozen instructions between memory round trips: first in line they keep the store path fed, and the walk loses < 10 % of its slots) and,
  // from 1024 proofs on, the input sections in front of the new-key pipeline so that the stores start with the step: 512 / 1024 / 2048 proofs
  // 0.764 -> 0.741, 1.479 -> 1.422, 3.12 -> 2.81 ms.  (256 proofs: +1 % / +4.6 %: the latency regime keeps the round-5 form.)
  const uint64_t lanes_w = (uint64_t)n_proofs * c->cfg.n_max;
  prog.sp.wave_prio = K.writer_prio >= 0 ? (uint32_t)K.writer_prio : (lanes_w >= 65536 ? 3u : 0u);
  prog.sp.rows_major = K.ser_rows ? 1u : 0u;
  // (round 6: in the small form of the tail -- below 28 672 lanes, see tail_aside -- k_proof's sections + D.1a are ONE launch instead of four: 80 / 128 /
  // 208 proofs x 128 -1.6 / -4.5 / -2.6 %, N = 512 x 32 proofs -5.6 %, 20 ... 48 proofs level; at 256 proofs +2 %: profiles/r06_one_launch_ab.txt)
  const bool tail_aside_early = K.ser_split && (uint64_t)n_proofs * n >= (K.tail_aside_min >= 0 ? (uint64_t)K.tail_aside_min : 28672u);
  prog.sp.one_launch_max = K.one_launch_max >= 0 ? (uint32_t)K.one_launch_max : (tail_aside_early ? 8u : n_proofs);
  auto serialize = [&](uint32_t mask, hipStream_t on, uint32_t max_wgs = 0) -> int32_t {
    if (!d_out_elems) return TMX_OK;
    // (sections the caller did not ask for are not written; the seam spans are few and always written)
    mask &= ((c->sections & TMX_SEC_HINT) ? prog.mask_hint : 0u) | ((c->sections & TMX_SEC_DERIVED) ? prog.mask_derived : 0u) | (1u << 31);
    int r = launch_serialize(prog.sp, src, c->d_lut[kind], c->d_wave_sec[kind], c->d_seams[kind], (uint32_t)prog.seam_waves.size(), n_proofs,
                             d_out_elems, mask, on, max_wgs);
    if (r) return fail(c, TMX_ERR_HIP, std::string("k_serialize launch: ") + hipGetErrorString((hipError_t)r));
    return TMX_OK;
  };
  // (every record / wait is a packet the command processor handles in order: the caller's stream carries the critical path, so
  // events are shared where they mark the same point and joins are chained through the side streams)
  // One set of scratch buffers and join events per context: a call on another stream starts behind the previous call's end
  // (a no-op in the usual case of one stream per context)
  if (c->last_stream_valid && c->last_stream != s) HIPCK(c, hipStreamWaitEvent(s, c->ev_done, 0));
  HIPCK(c, hipEventRecord(ev[0], s));
  HIPCK(c, hipStreamWaitEvent(c->side, ev[0], 0));
  int32_t st0 = TMX_OK;
  int rc = 0;
  // (round 5, once the compaction and the validator-set cache had shortened the chain and the step's end was the row writes: 2048 workgroups at
  // 256 proofs 0.377 - 0.387 vs 0.395 - 0.410 ms with 1024; 512 / 768: 0.455 / 0.432; 3072 / 4096 / 8192: 0.385 / 0.381 / 0.393; no difference
  // at 64, 128, 512 and 1024 proofs)
  const uint64_t lanes_bc = (uint64_t)n_proofs * n;
  // (round 6, profiles/r06_writer_cap_sweep.txt: from 1024 proofs on -- where the leaves go first and the step is the sum of a VALU-bound chain
  // and an HBM-bound tail -- 8192 workgroups: 1024 / 1280 / 1536 proofs 1.557 -> 1.455, 2.008 -> 1.890, 2.358 -> 2.145 ms, 2048 proofs flat;
  // 512 to 1023 proofs 1536 instead of 1024: 640 / 768 / 896 proofs 1.11 -> 0.98, 1.164 -> 1.116, 1.332 -> 1.315 ms, 512 proofs unchanged)
  // (round 6, with the proof-major writer k_serialize_rows, profiles/r06_writer_cap_sweep2.txt: persistent writers hold their wave slots for
  // the whole launch and the chain's kernels wait for a slot; 32768 short-lived workgroups instead of 8192: 1024 / 1536 / 2048 proofs
  // 1.42 -> 1.28, 2.06 -> 1.91, 2.81 -> 2.59 ms; 65536: the same; 131072 and uncapped: +6 ... +10 %.  The whole throughput regime -- this cap,
  // the input sections first, the leaves first, D.1a early -- from THROUGHPUT_LANES on (first 98304: 768 / 896 proofs 1.08 -> 1.05, 1.25 -> 1.20 ms; then 81920: 672 / 704 / 736 proofs 0.894 -> 0.854, 0.943 -> 0.876, 1.011 -> 0.946, 640 level);
  // 512 / 640 proofs within +-2 % either way and left as they were)
  const bool throughput = lanes_bc >= THROUGHPUT_LANES;
  const uint32_t beside_chain_wgs = K.few_wgs > 0 ? (uint32_t)K.few_wgs : (throughput ? 32768u : (lanes_bc >= 65536 ? 1536u : (lanes_bc > 16384 ? 2048u : 1024u)));  // (warm key cache, 256 proofs: 2048 / 4096 / uncapped +10 / +4 / +2 %)
  // Leaves first (TMX_LEAVES=1|0, default by size): marshalled validators + leaf hashes as a 10-us launch of their own in front of k_proof,
  // so that the byte fields of the two per-lane derived sections (D.2a: the leaves; D.1a: the leaves + phase 1 -- 42 % of a skip row) are
  // written by the low-priority stream behind the input sections instead of behind k_proof (which ends at ~300 us inside a step) / k_ed_fin.
  // Measured at N = 128: -5 % step at 1024 proofs (on top of the -6 % of D.1a behind k_proof's sections), but +1.5 % at 256, +4.5 % at 512,
  // +8 % at 64: below ~1000 proofs every extra concurrent launch stretches the EdDSA chain by more than the tail it removes.
  const bool leaves_first = K.ser_split && d_out_elems && (K.leaves >= 0 ? K.leaves != 0 : throughput);
  // side: k_proof first -- for a single proof it IS the critical path, and every API call in front of its launch is latency --, its two
  // timing events on the dispatch itself; then the sections that only need its results
  const bool xp = K.ext_events && !leaves_first;
  if (!xp) HIPCK(c, hipEventRecord(evs[0], c->side));
  if (leaves_first) {
    rc = launch_leaves((uint32_t)kind, n, n_proofs, d_targets, d_trusteds, tl + TL_OFF_LT, TL_STRIDE, c->d_lr, c->side, K.ext_events ? c->ev_leaves : nullptr);
    if (rc) return fail(c, TMX_ERR_HIP, std::string("k_leaves launch: ") + hipGetErrorString((hipError_t)rc));
    if (!K.ext_events) HIPCK(c, hipEventRecord(c->ev_leaves, c->side));
  }
  // k_proof as four role workgroups per proof where its latency is the call's (a handful of proofs, or no EdDSA stage beside it: the
  // validator-sharded finish); as one workgroup per proof in a batch, where 2048 role waves held up the EdDSA chain (256 proofs: 0.60
  // vs 0.41 ms) and k_proof is hidden behind it anyway (32 / 64 proofs: +-1 %)
  const uint64_t lanes_all = (uint64_t)n_proofs * n;
  const bool roles = K.proof_roles && c->d_tiny && (lanes_all <= 2048 || (!eddsa_writes_rows && lanes_all <= 16384));
  if (!roles && c->setc.table && ++c->setc_epoch == 0) c->setc_epoch = 1;  // (ages are differences mod 2^32: a wrap is harmless)
  rc = roles
           ? launch_proof_roles(proof_params(c, kind, leaves_first), n_proofs, d_proofs, d_targets, d_trusteds, tl + TL_OFF_LT, TL_STRIDE, c->d_lr, c->d_pf,
                                c->d_nodes_t, c->d_nodes_r, reports, c->d_tiny, c->side, xp ? evs[0] : nullptr, xp ? evs[1] : nullptr)
           : launch_proof(proof_params(c, kind, leaves_first), n_proofs, d_proofs, d_targets, d_trusteds, tl + TL_OFF_LT, TL_STRIDE, c->d_lr, c->d_pf,
                          c->d_nodes_t, c->d_nodes_r, reports, c->side, xp ? evs[0] : nullptr, xp ? evs[1] : nullptr, c->setc);
  if (rc) return fail(c, TMX_ERR_HIP, std::string("k_proof launch: ") + hipGetErrorString((hipError_t)rc));
  if (!xp) HIPCK(c, hipEventRecord(evs[1], c->side));
  // side3: the sections that are a pure expansion of the input records (42 % of a skip row) -- HBM is idle while EdDSA runs.
  // Launches that run beside the EdDSA latency chain are walked by a fixed number of workgroups (k_serialize_few): a memory-bound grid of
  // half a million short waves keeps every wave slot of the chip taken, and the chain's workgroups then wait for a slot whatever their
  // priority (a pure store stream beside the EdDSA stage: k_ed_keys 50 -> 103 us, the stage 0.42 -> 0.72 ms).  Four workgroups per CU
  // still write at the rate these sections need: step -3 % at 256 proofs x 128 (1024 workgroups; 512: the EdDSA stage -48 us but the
  // sections end after it), -4.5 % at 512, -3 % at 1024 (1536), -1 % at 64, +-0 at 32.
  // (the split warm schedule puts the new-key pipeline on side3 IN FRONT of these sections -- usually three empty launches; with new keys
  // the pipeline must not queue behind 200 us of trickling stores -- so the sections are enqueued behind the EdDSA stage then)
  c->plan = eddsa_writes_rows ? ed_plan(c, (uint32_t)lanes_all, true) : tmx_ctx::EdPlan{};
  // Fused rows (round 6; layout.h FusedRows): in the split warm schedule above 16384 lanes -- s*B and the resident walk are launches of their
  // own there, throughput-bound and on the chip from ~40 us on -- the input-only sections are not a launch that queues beside the chain but
  // work items the waves of those kernels claim between their table additions; the capped launch on side3 sweeps up what they leave.
  FusedRows fused = {};
  {
    const uint32_t in_mask = prog.mask_inputs & (((c->sections & TMX_SEC_HINT) ? prog.mask_hint : 0u) | ((c->sections & TMX_SEC_DERIVED) ? prog.mask_derived : 0u));
    const int fb = K.fused_base >= 0 ? K.fused_base : TMX_FUSED_BASE_DEFAULT, fw = K.fused_walk >= 0 ? K.fused_walk : TMX_FUSED_WALK_DEFAULT;
    if (c->plan.split && !c->plan.sb_with_hash && K.ser_split && d_out_elems && in_mask && (fb | fw) && c->d_span_ctr) {
      uint32_t lo = 0xffffffffu, hi = 0;
      for (uint32_t k = 0; k < prog.sp.n_sections; k++) {
        if (!((in_mask >> k) & 1u)) continue;
        const Section& sc = prog.sp.sec[k];
        lo = std::min(lo, sc.elem_start);
        hi = std::max(hi, sc.elem_start + sc.lane_elems * sc.n_lanes);
      }
      const uint32_t first = lo / prog.sp.span, last = (hi + prog.sp.span - 1) / prog.sp.span;
      if (prog.sp.span == SER_SPAN_ELEMS && last > first && (uint64_t)(last - first) * n_proofs < 0x7fffffffull) {
        if (c->span_ctr_dirty) { HIPCK(c, hipMemsetAsync(c->d_span_ctr, 0, 256, s)); c->span_ctr_dirty = false; }
        fused.ctr = c->d_span_ctr + 32 * c->span_parity;  // (the two counters on lines of their own)
        fused.lut = reinterpret_cast<const uint32_t*>(c->d_lut[kind]); fused.wave_sec = reinterpret_cast<const uint8_t*>(c->d_wave_sec[kind]);
        fused.out = reinterpret_cast<uint64_t*>(d_out_elems); fused.sec_mask = in_mask; fused.first_span = first; fused.n_spans = last - first;
        fused.n_proofs = n_proofs; fused.per_base = (uint32_t)fb; fused.per_walk = (uint32_t)fw;
        c->fused_prog = resolve_serialize_program(prog.sp, src);
      }
    }
  }
  auto side3_inputs = [&]() -> int32_t {
    HIPCK(c, hipStreamWaitEvent(c->side3, ev[0], 0));
    if (fused.ctr) {
      int r = launch_serialize_claim(c->fused_prog, fused, c->d_span_ctr + 32 * (c->span_parity ^ 1u), c->side3, beside_chain_wgs);
      if (r) c->span_ctr_dirty = true;
      if (r) return fail(c, TMX_ERR_HIP, std::string("k_serialize_claim launch: ") + hipGetErrorString((hipError_t)r));
      c->span_parity ^= 1u;
    } else if (K.ser_split && (st0 = serialize(prog.mask_inputs, c->side3, beside_chain_wgs))) return st0;
    if (leaves_first) {
      HIPCK(c, hipStreamWaitEvent(c->side3, c->ev_leaves, 0));
      if ((st0 = serialize(prog.mask_leaves, c->side3, beside_chain_wgs))) return st0;
    }
    HIPCK(c, hipEventRecord(c->ev_join3, c->side3));
    return TMX_OK;
  };
  const bool inputs_first = K.inputs_first_min >= 0 ? (K.inputs_first_min > 0 && lanes_bc >= (uint64_t)K.inputs_first_min) : throughput;
  const bool defer3 = c->plan.split && !inputs_first;
  if (!defer3 && (st0 = side3_inputs())) return st0;

  // ev[1] rides on the k_ed_fin dispatch itself (TMX_EXT_EVENTS=0: a record packet behind it)
  c->fin_done = K.ext_events ? ev[1] : nullptr;
  c->fin_done_attached = false;
  c->ev_hash_recorded = false;
  // D.1b (the word fields of the per-target-lane derived values: h, the ten coordinates, the verdict) goes straight from the EdDSA
  // finish into the rows when the EdDSA kernels run here (and k_verdict adds the lane's ten k_proof-derived elements): no serializer
  // launch behind the last kernel of the chain.  A producer that copies caller-provided lane records leaves the section to the serializer.
  RowOut row = {};
  if (eddsa_writes_rows && d_out_elems && (c->sections & TMX_SEC_DERIVED)) {
    row.rows = reinterpret_cast<uint64_t*>(d_out_elems); row.elem_stride = prog.sp.elem_stride; row.n = n; row.d1b_start = prog.d1b_start;
  }
  const uint32_t mask_final = row.rows ? 0u : prog.mask_final;
  c->row = row;
  c->fused = fused;
  int32_t st = ed_producer(s);
  c->fused = FusedRows{};
  if (fused.ctr && st) c->span_ctr_dirty = true;  // (the sweeper that zeroes the next counter was never enqueued)
  const bool fin_split = c->plan.split;  // (the lanes of new keys are finished on side2: whatever reads every lane's verdict on s waits for ev_direct)
  c->plan.valid = false;
  c->fin_done = nullptr;
  c->row = RowOut{};
  // the late epilogue (TMX_EPI_LATE): owed to the NEXT launch once the producer has enqueued the dedup -- on the error paths too, or that launch
  // would probe a dirty hash table behind an event that was never re-recorded
  auto late_epilogue = [&]() -> int32_t {
    if (!c->epilogue_pending) return TMX_OK;
    c->epilogue_pending = false;
    if ((size_t)c->hash_mask + 1 > KC_EPILOGUE_CLEARS_UP_TO) HIPCK(c, hipMemsetAsync(c->d_hash, 0xff, ((size_t)c->hash_mask + 1) * 4, c->side3));
    if ((rc = launch_kc_epilogue(c->pending_q, c->side3))) return fail(c, TMX_ERR_HIP, std::string("k_kc_epilogue launch: ") + hipGetErrorString((hipError_t)rc));
    HIPCK(c, hipEventRecord(c->ev_hash_clean, c->side3));
    HIPCK(c, hipEventRecord(c->ev_join3, c->side3));
    return TMX_OK;
  };
  if (st) { (void)late_epilogue(); return st; }
  if (defer3 && (st0 = side3_inputs())) { (void)late_epilogue(); return st0; }
  if ((st0 = late_epilogue())) return st0;
  if (leaves_first) {  // side3, behind the input sections and D.2a: D.1a as soon as phase 1 is done; ev_join3 moves behind it
    if (!c->ev_hash_recorded) HIPCK(c, hipEventRecord(c->ev_hash, s));  // (a producer without a phase-1 event: everything it enqueued)
    HIPCK(c, hipStreamWaitEvent(c->side3, c->ev_hash, 0));
    if ((st0 = serialize(prog.mask_p1, c->side3, beside_chain_wgs))) return st0;
    HIPCK(c, hipEventRecord(c->ev_join3, c->side3));
  }
  // (a small batch is pure latency: its tail stays on s, two cross-stream hops cost more than the overlap gains)
  // (round 6: the threshold moved from 4096 to 10240 lanes -- 32 / 48 / 64 proofs x 128: 0.210 -> 0.199, 0.222 -> 0.214, 0.259 -> 0.252 ms with the
  // one-launch tail on s; 96 proofs: 0.2625 aside vs 0.2715 on s.  profiles/r06_small_tail_ab.txt)
  // (round 6, with k_verdict_tail_wide as the one launch -- profiles/r06_tail_wide_ab.txt: 32 / 48 / 64 proofs 0.209 -> 0.204, 0.233 -> 0.224, 0.244 -> 0.225;
  // the one-launch form on s up to 28 672 lanes instead of 10 240: 80 / 96 / 160 / 208 proofs x 128 0.269 -> 0.257, 0.284 -> 0.273, 0.309 -> 0.300, 0.326 -> 0.318,
  // 224 / 240 level, 256 +1.4 %; N = 32: 512 / 768 proofs -3.7 / -1.8 %, 1024 +5 %; N = 512: 48 proofs -4 %, 32 / 64 level)
  const bool tail_aside = tail_aside_early;
  const bool small_tail = K.ser_split && !tail_aside;  // D.1a goes into the SAME launch as k_proof's sections (one launch, behind the hash event)
  if (small_tail && !leaves_first) {
    if (!c->ev_hash_recorded) HIPCK(c, hipEventRecord(c->ev_hash, s));  // (a producer without a phase-1 event: everything it enqueued)
    HIPCK(c, hipStreamWaitEvent(c->side, c->ev_hash, 0));
  }
  st0 = K.ser_split ? serialize(prog.mask_proof | (leaves_first ? 0u : prog.mask_leaves) | (small_tail && !leaves_first ? prog.mask_p1 : 0u), c->side) : TMX_OK;
  if (st0) return st0;
  // the validator-set cache's LRU: one workgroup behind k_proof and its sections on this stream (the next k_proof is behind it), off every chain
  if (!roles && c->setc.table && (rc = launch_setc_evict(c->setc, c->setc_epoch, c->side)))
    return fail(c, TMX_ERR_HIP, std::string("k_setc_evict launch: ") + hipGetErrorString((hipError_t)rc));
  // D.1a (the byte fields of the per-target-lane derived values: a quarter of the row) needs k_proof and phase 1, not k_ed_fin: behind
  // k_proof's sections on the side stream, i.e. while the table walk and the finish run.  Measured (round 2): -6.3 % step at 1024
  // proofs x 128; +-0.5 % at 256 and 64, +1 ... +2 % at 512 proofs (and +7 % at 256 with a warm key cache): on from 131072 lanes
  const bool p1_early = leaves_first || small_tail ||
                        (K.ser_split && c->ev_hash_recorded && (K.p1_early >= 0 ? K.p1_early != 0 : throughput));
  if (p1_early && !leaves_first && !small_tail) {
    HIPCK(c, hipStreamWaitEvent(c->side, c->ev_hash, 0));
    if ((st0 = serialize(prog.mask_p1, c->side, K.p1_early == 2 ? 0u : beside_chain_wgs))) return st0;
  }

  HIPCK(c, hipStreamWaitEvent(c->side, c->ev_join3, 0));  // ev_join = both low-priority streams done
  HIPCK(c, hipEventRecord(c->ev_join, c->side));
  if (!c->fin_done_attached) HIPCK(c, hipEventRecord(ev[1], s));
  if (small_tail) {
    // D.1a (needs the leaves and the hash role, both long done) goes with k_proof's sections on the side stream; what follows the join of
    // k_proof and the EdDSA finish on s is ONE launch: verdict + the sections that carry it + the seam spans (k_verdict_tail)
    if (fin_split && K.join1) {
      // ONE wait packet between the finish and the tail instead of two: side2 -- idle behind the new-key lanes' finish -- waits for k_proof and
      // records ev_direct again, which then stands for both
      HIPCK(c, hipStreamWaitEvent(c->side2, evs[1], 0));
      HIPCK(c, hipEventRecord(c->ev_direct, c->side2));
      HIPCK(c, hipStreamWaitEvent(s, c->ev_direct, 0));
    } else {
      HIPCK(c, hipStreamWaitEvent(s, evs[1], 0));
      if (fin_split) HIPCK(c, hipStreamWaitEvent(s, c->ev_direct, 0));
    }
    const bool xv = K.ext_events && n_proofs != 0;  // the verdict's two timing events ride on its dispatch
    if (!xv) HIPCK(c, hipEventRecord(evs[2], s));
    const bool with_rows = d_out_elems != nullptr;
    uint32_t tail_mask = (mask_final | prog.mask_tail) & (((c->sections & TMX_SEC_HINT) ? prog.mask_hint : 0u) | ((c->sections & TMX_SEC_DERIVED) ? prog.mask_derived : 0u) | (1u << 31));
    if (K.tail_wide && !(tail_mask & mask_final))  // (k_verdict_tail: one 1024-thread workgroup per proof expands D.5 and the seam spans element by element -- 21 us at 32 proofs, 82 at 256)
      rc = launch_verdict_tail_wide((uint32_t)kind, n, n_proofs, tl + TL_OFF_ED, TL_STRIDE, c->d_pf, reports, row, prog.sp, src, c->d_lut[kind], c->d_wave_sec[kind],
                                    c->d_seams[kind], (uint32_t)prog.seam_waves.size(), with_rows ? d_out_elems : nullptr, tail_mask, prog.tail_dep_elem, s,
                                    xv ? evs[2] : nullptr, xv ? evs[3] : nullptr);
    else
    rc = launch_verdict_tail((uint32_t)kind, n, n_proofs, tl + TL_OFF_ED, TL_STRIDE, c->d_pf, reports, row, prog.sp, src, c->d_lut[kind], c->d_seams[kind],
                             (uint32_t)prog.seam_waves.size(), with_rows ? d_out_elems : nullptr, tail_mask, s, xv ? evs[2] : nullptr, xv ? evs[3] : nullptr);
    if (rc) return fail(c, TMX_ERR_HIP, std::string("k_verdict_tail launch: ") + hipGetErrorString((hipError_t)rc));
    if (!xv) HIPCK(c, hipEventRecord(evs[3], s));
    HIPCK(c, hipStreamWaitEvent(s, c->ev_join, 0));
  } else if (K.ser_split) {
    // tail: the per-lane derived section (a quarter of the row) only needs k_ed_fin + k_proof, so it is written on s while the
    // verdict and the few sections that carry it go through the high-priority side stream.  (ev[2] = ev[1] here: every packet
    // between k_ed_fin and the serializer is latency on the critical path.)
    // (the wait that is over first goes first: k_proof ends before the finish, and a wait packet behind the finish's is ~5 us of the tail's start)
    if (K.join1) {
      HIPCK(c, hipStreamWaitEvent(c->side2, evs[1], 0));  // k_proof itself, not the serializer launches queued behind it
      HIPCK(c, hipStreamWaitEvent(c->side2, ev[1], 0));
    } else {
    HIPCK(c, hipStreamWaitEvent(c->side2, ev[1], 0));
    HIPCK(c, hipStreamWaitEvent(c->side2, evs[1], 0));  // k_proof itself, not the serializer launches queued behind it
    }
    if (K.tail_wide && n_proofs) {
      const bool xv = K.ext_events;
      if (!xv) HIPCK(c, hipEventRecord(evs[2], c->side2));
      const uint32_t tail_mask = prog.mask_tail & (((c->sections & TMX_SEC_HINT) ? prog.mask_hint : 0u) | ((c->sections & TMX_SEC_DERIVED) ? prog.mask_derived : 0u) | (1u << 31));
      rc = launch_verdict_tail_wide((uint32_t)kind, n, n_proofs, tl + TL_OFF_ED, TL_STRIDE, c->d_pf, reports, row, prog.sp, src, c->d_lut[kind], c->d_wave_sec[kind],
                                    c->d_seams[kind], (uint32_t)prog.seam_waves.size(), d_out_elems, tail_mask, prog.tail_dep_elem, c->side2,
                                    xv ? evs[2] : nullptr, xv ? evs[3] : nullptr);
      if (rc) return fail(c, TMX_ERR_HIP, std::string("k_verdict_tail_wide launch: ") + hipGetErrorString((hipError_t)rc));
      if (!xv) HIPCK(c, hipEventRecord(evs[3], c->side2));
    } else {
    HIPCK(c, hipEventRecord(evs[2], c->side2));
    rc = launch_verdict((uint32_t)kind, n, n_proofs, tl + TL_OFF_ED, TL_STRIDE, c->d_pf, reports, row, c->side2);
    if (rc) return fail(c, TMX_ERR_HIP, std::string("k_verdict launch: ") + hipGetErrorString((hipError_t)rc));
    HIPCK(c, hipEventRecord(evs[3], c->side2));
    if ((st0 = serialize(prog.mask_tail, c->side2))) return st0;
    }
    HIPCK(c, hipStreamWaitEvent(c->side2, c->ev_join, 0));  // ev_tail = every side stream done
    HIPCK(c, hipEventRecord(c->ev_tail, c->side2));
    if (mask_final | (p1_early ? 0u : prog.mask_p1)) {
      HIPCK(c, hipStreamWaitEvent(s, evs[1], 0));
      if ((st0 = serialize(mask_final | (p1_early ? 0u : prog.mask_p1), s))) return st0;
    }
    HIPCK(c, hipStreamWaitEvent(s, c->ev_tail, 0));
  } else {
    HIPCK(c, hipStreamWaitEvent(s, c->ev_join, 0));
    if (fin_split) HIPCK(c, hipStreamWaitEvent(s, c->ev_direct, 0));
    HIPCK(c, hipEventRecord(evs[2], s));
    rc = launch_verdict((uint32_t)kind, n, n_proofs, tl + TL_OFF_ED, TL_STRIDE, c->d_pf, reports, row, s);
    if (rc) return fail(c, TMX_ERR_HIP, std::string("k_verdict launch: ") + hipGetErrorString((hipError_t)rc));
    HIPCK(c, hipEventRecord(evs[3], s));
    HIPCK(c, hipEventRecord(ev[2], s));
    if ((st0 = serialize(prog.mask_inputs | prog.mask_proof | prog.mask_leaves | mask_final | prog.mask_p1 | prog.mask_tail, s))) return st0;
  }
  HIPCK(c, hipEventRecord(ev[3], s));
  c->slot_tiny[slot] = false;
  c->last_stream = s; c->last_stream_valid = true;
  c->ev_done = ev[3];
  c->last_kind = kind; c->last_n_proofs = n_proofs;
  c->n_calls++;
  return TMX_OK;
}

// ---- the small-launch path (kernels.h TinyLaunch, tiny.hpp): <= TINY_MAX_LANES validator lanes.
// Two launches on the caller's stream, no cross-stream hand-off in front of any of them: k_tiny (every role that needs only the input
// records) and k_tiny_tail (checks, verdict, the sections that needed k_tiny).  A lane whose key is not resident is computed table-free
// inside its workgroup.  The key cache is only READ by k_tiny; its bookkeeping is the classic key pipeline, unchanged, enqueued on the
// high-priority side stream BEHIND k_tiny over the context's shadow copy of the lanes' keys (k_ed_dedup -> k_ed_keys -> tables of the new
// keys -> k_kc_epilogue): hit / miss counters, LRU stamps, insertion, eviction and the schedule hint are exactly what a classic launch
// leaves, and nothing reads the caller's buffers once the caller's stream is done.  TMX_TINY=0 never takes this path.
static bool use_tiny(const tmx_ctx* c, uint64_t n_lanes) {
  const Knobs& K = c->knobs;
  // (round 6: up to 1536 lanes instead of TINY_MAX_LANES = 2048 -- with the validator-set cache, the one-launch tail and the one-launch sections the
  // classic graph needs 0.18 - 0.205 ms at 13 ... 16 proofs x 128 where the two launches need 0.207 ... 0.225 (a workgroup per lane: their time grows
  // with the lanes); N = 64: 28 proofs 0.216 -> 0.201; N = 512: 4 proofs 0.51 -> 0.44; N = 32 loses 3 - 5 % between 1536 and 2048 lanes.
  // profiles/r06_small_batches_ab.txt)
  // (small validator sets keep the old bound: N = 32 x 56 / 64 proofs 0.222 / 0.219 on the small path against 0.228 / 0.230)
  const uint64_t tiny_max = K.tiny_max >= 0 ? std::min<uint64_t>((uint64_t)K.tiny_max, TINY_MAX_LANES) : (c->cfg.n_max <= 32 ? TINY_MAX_LANES : 1536u);
  if (K.tiny == 0 || n_lanes == 0 || n_lanes > tiny_max || !c->d_tiny || !c->d_shadow) return false;
  if (c->kc.cap == 0 || K.dedup_mode == 0) return false;  // (TMX_DEDUP=0: no per-key tables at all -- the classic table-free kernels)
  if (K.warm_schedule >= 0 && K.tiny < 0) return false;   // a forced schedule names one of the classic graphs
  return true;
}
static void tiny_common(tmx_ctx* c, TinyLaunch& T, uint32_t n_lanes, const void* d_lanes, void* d_ed, uint32_t ed_stride) {
  std::memset(&T, 0, sizeof T);
  T.n_lanes = n_lanes; T.d_target = d_lanes; T.d_ed = d_ed; T.ed_stride = ed_stride;
  T.d_qtable = c->d_qtable; T.d_keytab = c->d_keytab; T.d_keyrec = c->d_keyrec; T.kc = c->kc; T.d_tiny = c->d_tiny; T.d_shadow = c->d_shadow;
  c->last_lanes = n_lanes;
}
// the ordering a small launch needs in front of it: the previous batch if it ran on another stream, and the tail of the key pipeline
// the previous launch left on the side stream (it may still be inserting keys into the cache this launch probes)
static int32_t tiny_order(tmx_ctx* c, hipStream_t s) {
  if (c->last_stream_valid && c->last_stream != s) HIPCK(c, hipStreamWaitEvent(s, c->ev_done, 0));
  // (an isolated call finds the previous key pipeline long finished: a host-side query instead of a wait packet in front of the first launch)
  if (hipEventQuery(c->ev_hash_clean) != hipSuccess) {
    (void)hipGetLastError();
    HIPCK(c, hipStreamWaitEvent(s, c->ev_hash_clean, 0));
  }
  return TMX_OK;
}
// the cache bookkeeping of a small launch: the classic key pipeline on side2, behind k_tiny (event `after`), over the shadow records
static int32_t tiny_key_pipeline(tmx_ctx* c, uint32_t n_lanes, hipEvent_t after) {
  const Knobs& K = c->knobs;
  EdQuad Q;
  std::memset(&Q, 0, sizeof Q);
  Q.n_lanes = n_lanes; Q.d_target = c->d_shadow; Q.d_qtable = c->d_qtable; Q.d_pre = c->d_pre; Q.d_mulout = c->d_mulout;
  Q.d_hash = c->d_hash; Q.hash_mask = c->hash_mask; Q.d_owner_of = c->d_owner_of; Q.d_slot_of_owner = c->d_slot_of_owner;
  Q.d_slot_of_uid = c->d_slot_of_uid; Q.d_owners = c->d_owners; Q.d_keyrec = c->d_keyrec; Q.d_anchors = c->d_anchors; Q.d_keytab = c->d_keytab;
  Q.kc = c->kc; Q.mode = K.dedup_mode;
  Q.d_cnt = reinterpret_cast<uint32_t*>(c->d_cnt) + 8 * c->parity;
  Q.d_cnt_next = reinterpret_cast<uint32_t*>(c->d_cnt) + 8 * (c->parity ^ 1);
  c->parity ^= 1;
  Q.use_new = 0;  // nobody waits for the tables of this launch: they are for the next call
  Q.warm = (c->kc.persist && c->h_hint && c->h_hint[0] != 0 && c->h_hint[1] == 0) ? 1u : 0u;  // (grid sizes of the new-key kernels only)
  HIPCK(c, hipStreamWaitEvent(c->side2, after, 0));
  int rc = launch_ed_dedup(Q, c->side2);
  if (!rc) rc = launch_ed_keys(Q, c->side2);
  if (!rc) rc = launch_ed_tab_anchor(Q, 0, 1, c->side2);
  if (!rc) rc = launch_ed_tab_mult(Q, 0, 1, c->side2);
  if (rc) return fail(c, TMX_ERR_HIP, std::string("key pipeline launch: ") + hipGetErrorString((hipError_t)rc));
  if ((size_t)c->hash_mask + 1 > KC_EPILOGUE_CLEARS_UP_TO) HIPCK(c, hipMemsetAsync(c->d_hash, 0xff, ((size_t)c->hash_mask + 1) * 4, c->side2));
  rc = launch_kc_epilogue(Q, c->side2);
  if (rc) return fail(c, TMX_ERR_HIP, std::string("k_kc_epilogue launch: ") + hipGetErrorString((hipError_t)rc));
  HIPCK(c, hipEventRecord(c->ev_hash_clean, c->side2));
  return TMX_OK;
}
static int32_t run_tiny_lanes(tmx_ctx* c, uint32_t n_lanes, const void* d_lanes, void* d_ed, uint32_t ed_stride, hipStream_t s) {
  int32_t st = tiny_order(c, s);
  if (st) return st;
  TinyLaunch T;
  tiny_common(c, T, n_lanes, d_lanes, d_ed, ed_stride);
  const bool x = c->knobs.ext_events;
  int rc = launch_tiny(T, s, nullptr, x ? c->ev_fork2 : nullptr);
  if (rc) return fail(c, TMX_ERR_HIP, std::string("k_tiny launch: ") + hipGetErrorString((hipError_t)rc));
  if (!x) HIPCK(c, hipEventRecord(c->ev_fork2, s));
  return tiny_key_pipeline(c, n_lanes, c->ev_fork2);
}
static int32_t run_tiny(tmx_ctx* c, int32_t kind, uint32_t n_proofs, const void* d_proofs, const void* d_targets, const void* d_trusteds,
                        void* d_out_elems, void* d_reports, hipStream_t s) {
  const uint32_t n = c->cfg.n_max;
  const Knobs& K = c->knobs;
  int32_t st = tiny_order(c, s);
  if (st) return st;
  uint8_t* tl = reinterpret_cast<uint8_t*>(c->d_tl);
  const uint64_t slot = c->n_calls % tmx_ctx::EV_RING;
  hipEvent_t* ev = c->ev[slot];
  const Program& prog = c->prog[kind];
  SerializeSources src;
  std::memset(&src, 0, sizeof src);
  src.base[SRC_TARGET] = (const uint8_t*)d_targets; src.base[SRC_TRUSTED] = (const uint8_t*)d_trusteds;
  src.base[SRC_TL] = tl; src.base[SRC_LR] = (const uint8_t*)c->d_lr; src.base[SRC_PF] = (const uint8_t*)c->d_pf;
  src.nodes_t = (const uint8_t*)c->d_nodes_t; src.nodes_r = (const uint8_t*)c->d_nodes_r;
  TinyLaunch T;
  tiny_common(c, T, n_proofs * n, d_targets, tl + TL_OFF_ED, TL_STRIDE);
  T.n_proofs = n_proofs; T.P = proof_params(c, kind, false); T.d_proofs = d_proofs; T.d_trusted = d_trusteds;
  T.d_lt = tl + TL_OFF_LT; T.lt_stride = TL_STRIDE; T.d_lr = c->d_lr; T.d_pf = c->d_pf; T.d_nodes_t = c->d_nodes_t; T.d_nodes_r = c->d_nodes_r;
  T.d_reports = d_reports ? d_reports : c->d_reports;
  T.S = &prog.sp; T.src = &src; T.d_lut = c->d_lut[kind]; T.d_wave_sec = c->d_wave_sec[kind]; T.d_seams = c->d_seams[kind];
  T.n_seams = (uint32_t)prog.seam_waves.size(); T.d_out = d_out_elems;
  // sections the caller asked for (the seam spans are few and always written); D.1b goes straight from the finish into the rows
  const uint32_t want = ((c->sections & TMX_SEC_HINT) ? prog.mask_hint : 0u) | ((c->sections & TMX_SEC_DERIVED) ? prog.mask_derived : 0u) | (1u << 31);
  if (d_out_elems && (c->sections & TMX_SEC_DERIVED)) {
    T.row.rows = reinterpret_cast<uint64_t*>(d_out_elems); T.row.elem_stride = prog.sp.elem_stride; T.row.n = n; T.row.d1b_start = prog.d1b_start;
  }
  T.mask_inputs = prog.mask_inputs & want;
  T.mask_after = (prog.mask_proof | prog.mask_leaves | prog.mask_p1 | (T.row.rows ? 0u : prog.mask_final)) & want;
  T.mask_tail = prog.mask_tail & want;
  T.tail_dep_elem = prog.tail_dep_elem;
  const bool x = K.ext_events;
  if (!x) HIPCK(c, hipEventRecord(ev[0], s));
  int rc = launch_tiny(T, s, x ? ev[0] : nullptr, x ? ev[1] : nullptr);
  if (rc) return fail(c, TMX_ERR_HIP, std::string("k_tiny launch: ") + hipGetErrorString((hipError_t)rc));
  if (!x) { HIPCK(c, hipEventRecord(ev[1], s)); HIPCK(c, hipEventRecord(ev[2], s)); }
  rc = launch_tiny_tail(T, s, x ? ev[2] : nullptr, x ? ev[3] : nullptr);
  if (rc) return fail(c, TMX_ERR_HIP, std::string("k_tiny_tail launch: ") + hipGetErrorString((hipError_t)rc));
  if (!x) HIPCK(c, hipEventRecord(ev[3], s));
  // (enqueued after both launches of the caller's stream: the key pipeline is nobody's critical path)
  if ((st = tiny_key_pipeline(c, n_proofs * n, ev[1]))) return st;
  c->slot_tiny[slot] = true;
  c->last_stream = s; c->last_stream_valid = true;
  c->ev_done = ev[3];
  c->last_kind = kind; c->last_n_proofs = n_proofs;
  c->n_calls++;
  return TMX_OK;
}

static int32_t check_batch_args(tmx_ctx* c, int32_t kind, uint32_t n_proofs, const void* d_proofs, const void* d_targets, const void* d_trusteds) {
  if (!c || (kind != TMX_KIND_SKIP && kind != TMX_KIND_STEP) || !d_proofs || !d_targets) return TMX_ERR_BAD_ARG;
  if (kind == TMX_KIND_SKIP && !d_trusteds) return fail(c, TMX_ERR_BAD_ARG, "skip needs the trusted hash fields");
  if (n_proofs > c->cfg.max_batch) return fail(c, TMX_ERR_CAPACITY, "n_proofs exceeds the context's max_batch");
  return TMX_OK;
}

// EdDSA stage: begins and ends on stream s.  The new-key pipeline (decode -> doubling chain -> window tables -> cache epilogue) runs on
// the high-priority stream side2; what s does beside it depends on what the launch EXPECTS (the epilogue of the last finished launch left
// a hint in page-locked memory: did it see new keys?) -- either schedule is correct for any input, the expectation only picks the faster:
//   cold (new keys expected):  s: dedup -> phase 1 (hash | s*B) -> [keys] table-free lanes -> [tables, part by part] walk -> finish
//   warm (keys resident):      s: dedup -> hash -> [tables] walk -> [s*B and the table-free lanes, on side2] -> finish
//                              -- the walk waits for the hash role only, nothing empty is launched on s, the new-key kernels are
//                              small grids that find nothing to do long before the hash role ends, and s*B (which only the finish
//                              needs) fills the machine beside the 512 latency-bound waves of the hash role
//   tiny (<= 512 lanes):       s: dedup -> phase 1 -> walk of the resident keys -> [table-free lanes + their finish, on side2] -> finish;
//                              never waits for tables: the tables of new keys are built on side2 for the next call
static int run_eddsa(tmx_ctx* c, uint32_t n_lanes, const void* d_lanes, void* d_ed, uint32_t ed_stride, hipStream_t s) {
  const Knobs& K = c->knobs;
  EdQuad Q;
  Q.n_lanes = n_lanes; Q.d_target = d_lanes; Q.d_ed = d_ed; Q.ed_stride = ed_stride; Q.d_qtable = c->d_qtable; Q.d_pre = c->d_pre;
  Q.d_mulout = c->d_mulout; Q.d_hash = c->d_hash; Q.hash_mask = c->hash_mask; Q.d_owner_of = c->d_owner_of;
  Q.d_slot_of_owner = c->d_slot_of_owner; Q.d_slot_of_uid = c->d_slot_of_uid; Q.d_owners = c->d_owners; Q.d_keyrec = c->d_keyrec;
  Q.d_anchors = c->d_anchors; Q.d_keytab = c->d_keytab;
  Q.kc = c->kc; Q.mode = K.dedup_mode;
  Q.fin_done = c->fin_done; c->fin_done = nullptr;
  Q.row = c->row;
  Q.fused = c->fused; Q.fused_prog = c->fused_prog;
  c->last_lanes = n_lanes;
  Q.d_cnt = reinterpret_cast<uint32_t*>(c->d_cnt) + 8 * c->parity;
  Q.d_cnt_next = reinterpret_cast<uint32_t*>(c->d_cnt) + 8 * (c->parity ^ 1);
  c->parity ^= 1;
  const tmx_ctx::EdPlan plan = c->plan.valid ? c->plan : ed_plan(c, n_lanes, false);
  const bool tiny = plan.tiny, warm = plan.warm;
  // Compacted launch (kernels.h EdQuad): wherever the dedup opens the chain on s -- the warm schedules above 16384 lanes and every cold
  // launch.  (Up to 16384 lanes a warm launch opens with the hash role instead, the dedup beside it: those launches are latency-bound,
  // fewer lanes do not shorten them -- measured: 0.244 / 0.261 / 0.285 ms compacted against 0.200 / 0.251 / 0.261 at 32 / 64 / 128 proofs.)
  Q.compact = (c->dummy_ready && K.compact && !tiny && n_lanes > 2048 && !plan.hash_first && Q.mode != 0 && Q.kc.cap != 0) ? 1u : 0u;
  Q.d_live = c->d_live; Q.d_dummy_ed = c->d_dummy_ed;
  Q.use_new = tiny ? 0u : 1u;
  Q.warm = warm ? 1u : 0u;
  hipError_t e;
  // (the launch's hash table was cleared, and the cache committed, on side2 by the previous launch.  Skipping this wait when s has
  // already waited for side2's tail was measured: k_proof 0.45 -> 0.52 ms beside it and the step +2 % at 256 proofs -- the packet stays.)
  // Hash first (warm schedule, not tiny): SHA-512 mod l reads the lane records only, so it opens the chain on s while the dedup (cache
  // probe) and the key pipeline run on a side stream -- the walk waits for its ev_part[0] either way.  s then carries no wait for
  // ev_hash_clean; the side stream that runs the dedup does (the previous launch's tail may have run on the other one).
  const bool hash_first = plan.hash_first;
  // Events that mark the end of one kernel ride on its dispatch (completion signal) instead of a record packet behind it: on the
  // chain every packet is latency.  `x` = that is on and the kernel really is launched.
  const bool x = K.ext_events && n_lanes != 0, xt = x && Q.mode != 0 && Q.kc.cap != 0;
  int rc = 0;
  const bool sb_with_hash_hf = plan.sb_with_hash;
  if (plan.split) {
    // ---- warm, split by residency (round 5).  Three chains that meet only where the data says so:
    //   s      [dedup] -> hash -> walk of the RESIDENT lanes -> [s*B] finish of the RESIDENT lanes          (the critical chain: waits for no table)
    //   side3  [dedup] -> k_ed_keys -> anchors -> multiples -> cache epilogue                                (the new-key pipeline: usually four empty launches)
    //   side2  s*B -> [tables, hash] walk of the NEW-KEY lanes -> table-free lanes -> finish of both         (usually: s*B and three empty launches)
    // Round 4 ran the new-key kernels and s*B one after the other on side2 and made the whole walk and the whole finish wait for them:
    // four new keys of 401 -- the daily churn of a validator set -- stalled 32 768 lanes behind a decode + 252-doubling chain + 60 us of
    // s*B (+28 % step).  Now only the lanes that need the new tables wait for them, and the lanes' D.1b rows are written by whichever
    // finish owns the lane.  Everything that reads ALL lanes' verdicts (k_verdict, on side2 itself or behind ev_direct) is ordered after both.
    hipStream_t kq = c->side3;
    if (hash_first) {
      if ((e = hipEventRecord(c->ev_fork2, s)) != hipSuccess) return (int)e;
    } else {
      if ((e = hipStreamWaitEvent(s, c->ev_hash_clean, 0)) != hipSuccess) return (int)e;
      if ((rc = launch_ed_dedup(Q, s, x ? c->ev_fork2 : nullptr))) return rc;
      if (!x && (e = hipEventRecord(c->ev_fork2, s)) != hipSuccess) return (int)e;
    }
    rc = plan.sb_with_hash ? launch_ed_phase1(Q, s, x ? c->ev_hash : nullptr) : launch_ed_hash(Q, s, x ? c->ev_hash : nullptr);
    if (rc) return rc;
    if (!x && (e = hipEventRecord(c->ev_hash, s)) != hipSuccess) return (int)e;
    c->ev_hash_recorded = true;
    // Enqueue order = the order of need: a step that starts on an idle device (a single call behind a synchronize) is launched at the pace of
    // the host, ~4 us per call, and the resident walk used to be the twelfth launch of the stage (it started at ~125 instead of ~70 us):
    // the head of the new-key chain (it is the longer one when there are new keys), s*B, the resident walk and finish, then the rest.
    if (hash_first) {  // the dedup beside the hash role; the walk reads the owners it writes (ev_keys: free in this schedule)
      if ((e = hipStreamWaitEvent(kq, c->ev_fork2, 0)) != hipSuccess) return (int)e;
      // The previous launch's key-pipeline tail (hash table cleared, cache committed) may sit on ANOTHER stream than kq: a tiny call, a cold
      // or a non-split launch leave it on side2.  Without this wait the dedup raced it on d_hash / owner_of / the cache.
      // A no-op packet when the tail was on kq itself (split launch after split launch: the usual case).
      if ((e = hipStreamWaitEvent(kq, c->ev_hash_clean, 0)) != hipSuccess) return (int)e;
      if ((rc = launch_ed_dedup(Q, kq, x ? c->ev_keys : nullptr))) return rc;
      if (!x && (e = hipEventRecord(c->ev_keys, kq)) != hipSuccess) return (int)e;
      if ((e = hipStreamWaitEvent(s, c->ev_keys, 0)) != hipSuccess) return (int)e;
    }
    // side3: the new-key pipeline and the end of the launch's cache bookkeeping
    if (!hash_first && (e = hipStreamWaitEvent(kq, c->ev_fork2, 0)) != hipSuccess) return (int)e;
    if ((rc = launch_ed_keys(Q, kq, xt ? c->ev_part[1] : nullptr))) return rc;
    if (!xt && (e = hipEventRecord(c->ev_part[1], kq)) != hipSuccess) return (int)e;
    if ((rc = launch_ed_tab_anchor(Q, 0, 1, kq))) return rc;
    if ((rc = launch_ed_tab_mult(Q, 0, 1, kq, xt ? c->ev_part[0] : nullptr))) return rc;
    if (!xt && (e = hipEventRecord(c->ev_part[0], kq)) != hipSuccess) return (int)e;
    // (the epilogue -- one 1024-thread workgroup that needs sixteen free wave slots on one CU: 50 - 65 us to get them on the busy chip -- and the
    // clearing of the launch's hash table are needed by the NEXT launch only: run_batch enqueues them behind the input sections of side3, which then
    // start at ~70 instead of ~160 us.  0.401 -> 0.391, 0.382 -> 0.376 ms once the step's end was the row writes: round 5, after the compaction)
    if (K.epi_late) {
      c->epilogue_pending = true; c->pending_q = Q;
    } else {
      if ((size_t)c->hash_mask + 1 > KC_EPILOGUE_CLEARS_UP_TO && (e = hipMemsetAsync(c->d_hash, 0xff, ((size_t)c->hash_mask + 1) * 4, kq)) != hipSuccess) return (int)e;
      if ((rc = launch_kc_epilogue(Q, kq))) return rc;
      if ((e = hipEventRecord(c->ev_hash_clean, kq)) != hipSuccess) return (int)e;
    }
    // side2: s*B (a launch of its own above 16384 lanes), then the lanes that are not resident
    if ((e = hipStreamWaitEvent(c->side2, c->ev_fork2, 0)) != hipSuccess) return (int)e;
    if (!plan.sb_with_hash) {
      if ((rc = launch_ed_base(Q, c->side2, x ? c->ev_base : nullptr))) return rc;
      if (!x && (e = hipEventRecord(c->ev_base, c->side2)) != hipSuccess) return (int)e;
    }
    // s: the resident lanes
    if ((rc = launch_ed_mul_tab(Q, 0, 1, s, 1u))) return rc;
    if (!plan.sb_with_hash && (e = hipStreamWaitEvent(s, c->ev_base, 0)) != hipSuccess) return (int)e;
    rc = launch_ed_fin(Q, s, false, 1u);
    c->fin_done_attached = rc == 0 && Q.fin_done != nullptr;
    if (rc) return rc;
    // (the table-free lanes -- keys the cache has no room for: usually an empty launch -- need the decoded keys, not the tables: in front of
    // the wait for the tables instead of 12 us between the walk and the finish of the new-key lanes)
    if ((e = hipStreamWaitEvent(c->side2, c->ev_part[1], 0)) != hipSuccess) return (int)e;
    if ((e = hipStreamWaitEvent(c->side2, c->ev_hash, 0)) != hipSuccess) return (int)e;
    if ((rc = launch_ed_mul_direct(Q, c->side2, false, nullptr))) return rc;
    if ((e = hipStreamWaitEvent(c->side2, c->ev_part[0], 0)) != hipSuccess) return (int)e;
    if ((rc = launch_ed_mul_tab(Q, 0, 1, c->side2, 2u))) return rc;
    {
      EdQuad Q2 = Q;
      Q2.fin_done = nullptr;
      if ((rc = launch_ed_fin(Q2, c->side2, false, 2u))) return rc;
    }
    if ((e = hipEventRecord(c->ev_direct, c->side2)) != hipSuccess) return (int)e;
    return rc;
  }
  if (hash_first) {
    if ((e = hipEventRecord(c->ev_fork2, s)) != hipSuccess) return (int)e;
    rc = sb_with_hash_hf ? launch_ed_phase1(Q, s, x ? c->ev_hash : nullptr) : launch_ed_hash(Q, s, x ? c->ev_hash : nullptr);
    if (rc) return rc;
    if (!x && (e = hipEventRecord(c->ev_hash, s)) != hipSuccess) return (int)e;
    c->ev_hash_recorded = true;
    if ((e = hipStreamWaitEvent(c->side2, c->ev_fork2, 0)) != hipSuccess) return (int)e;
    // (the previous launch's tail is on side3 when that launch took the split schedule: side2's own order does not cover it)
    if ((e = hipStreamWaitEvent(c->side2, c->ev_hash_clean, 0)) != hipSuccess) return (int)e;
    rc = launch_ed_dedup(Q, c->side2, nullptr);
    if (rc) return rc;
  } else {
    if ((e = hipStreamWaitEvent(s, c->ev_hash_clean, 0)) != hipSuccess) return (int)e;
    rc = launch_ed_dedup(Q, s, x ? c->ev_fork2 : nullptr);
    if (rc) return rc;
    if (!x && (e = hipEventRecord(c->ev_fork2, s)) != hipSuccess) return (int)e;
    if ((e = hipStreamWaitEvent(c->side2, c->ev_fork2, 0)) != hipSuccess) return (int)e;
  }
  rc = launch_ed_keys(Q, c->side2, x ? c->ev_keys : nullptr);
  if (rc) return rc;
  if (!x && (e = hipEventRecord(c->ev_keys, c->side2)) != hipSuccess) return (int)e;
  // the end of side2's part of a launch: the launch's hash table cleared for the next one, the cache committed
  auto side2_tail = [&]() -> int {
    hipError_t e2;
    if ((size_t)c->hash_mask + 1 > KC_EPILOGUE_CLEARS_UP_TO &&  // (small tables: the epilogue clears them itself)
        (e2 = hipMemsetAsync(c->d_hash, 0xff, ((size_t)c->hash_mask + 1) * 4, c->side2)) != hipSuccess) return (int)e2;
    // (with parts > 1 the multiples of the earlier parts run on s: the epilogue only needs the counters k_ed_keys left, not the tables)
    int r2 = launch_kc_epilogue(Q, c->side2);
    if (r2) return r2;
    return (int)hipEventRecord(c->ev_hash_clean, c->side2);
  };
  // the table-free lanes on side2, behind the hash role of s (their h) -- off s, where an empty launch costs ~10 us of the chain
  auto direct_on_side2 = [&](bool fuse) -> int {
    hipError_t e2;
    if ((e2 = hipStreamWaitEvent(c->side2, c->ev_hash, 0)) != hipSuccess) return (int)e2;
    int r2 = launch_ed_mul_direct(Q, c->side2, fuse, x ? c->ev_direct : nullptr);
    if (r2) return r2;
    if (!x && (e2 = hipEventRecord(c->ev_direct, c->side2)) != hipSuccess) return (int)e2;
    return 0;
  };

  if (tiny || warm) {
    // hash role (tiny: all of phase 1 -- a launch of a few waves is pure latency, its roles side by side) on s
    // (a few thousand lanes: s*B as a role of the same launch as the hash, as in a tiny launch -- on side2 it started behind three empty
    // launches, ran into the walk and held the finish back by ~50 us: profiles/r04_p32_timeline.txt)
    const bool sb_with_hash = plan.sb_with_hash;
    if (!hash_first) {
      rc = sb_with_hash ? launch_ed_phase1(Q, s, x ? c->ev_hash : nullptr) : launch_ed_hash(Q, s, x ? c->ev_hash : nullptr);
      if (rc) return rc;
      if (!x && (e = hipEventRecord(c->ev_hash, s)) != hipSuccess) return (int)e;
      c->ev_hash_recorded = true;
    }
    // (up to 256 lanes the finish of the table-free lanes is fused into their kernel: at 512 the 512 waves of the fused kernel slow
    // k_proof, the longer of the two there, by more than they save)
    const bool fuse = tiny && n_lanes <= 256;
    if (tiny && (rc = direct_on_side2(fuse))) return rc;  // first on side2: a cold single proof is this kernel's latency
    rc = launch_ed_tab_anchor(Q, 0, 1, c->side2);
    if (rc) return rc;
    rc = launch_ed_tab_mult(Q, 0, 1, c->side2, xt ? c->ev_part[0] : nullptr);
    if (rc) return rc;
    if (!xt && (e = hipEventRecord(c->ev_part[0], c->side2)) != hipSuccess) return (int)e;
    if (!tiny) {  // s*B (only the finish needs it) beside the hash role and the walk, then the table-free lanes: ev_direct = both done
      rc = sb_with_hash ? 0 : launch_ed_base(Q, c->side2);
      if (rc) return rc;
      if ((rc = direct_on_side2(false))) return rc;
    }
    if ((rc = side2_tail())) return rc;
    if (!tiny && (e = hipStreamWaitEvent(s, c->ev_part[0], 0)) != hipSuccess) return (int)e;  // (an expectation that fails: s waits for the build)
    rc = launch_ed_mul_tab(Q, 0, 1, s);
    if (rc) return rc;
    if ((e = hipStreamWaitEvent(s, c->ev_direct, 0)) != hipSuccess) return (int)e;
    rc = launch_ed_fin(Q, s, fuse);
    c->fin_done_attached = rc == 0 && Q.fin_done != nullptr;
    return rc;
  }

  // The anchor chain is cut into `parts` launches on side2; the cached multiples of part p are built on s (idle once phase 1 is
  // done) while side2 doubles part p+1, so that only the multiples of the last part follow the chain.
  // (small launches: one part -- nothing to overlap, and every part is one more launch on s)
  const uint32_t parts = n_lanes <= 2048 ? 1u : 2u;  // (4 parts: no further gain)
  for (uint32_t p = 0; p < parts; p++) {
    const bool last = p + 1 == parts;
    rc = launch_ed_tab_anchor(Q, p, parts, c->side2, xt && !last ? c->ev_part[p] : nullptr);
    if (rc) return rc;
    if (last) {  // the multiples of the last part stay on the high-priority stream: nothing is left to overlap them with
      rc = launch_ed_tab_mult(Q, p, parts, c->side2, xt ? c->ev_part[p] : nullptr);
      if (rc) return rc;
    }
    if (!xt && (e = hipEventRecord(c->ev_part[p], c->side2)) != hipSuccess) return (int)e;
  }
  if ((rc = side2_tail())) return rc;
  // phase 1 (SHA-512 mod l, s*B: throughput work for every lane) on s
  rc = launch_ed_phase1(Q, s, x ? c->ev_hash : nullptr);
  if (rc) return rc;
  if (!x && (e = hipEventRecord(c->ev_hash, s)) != hipSuccess) return (int)e;
  c->ev_hash_recorded = true;
  if ((e = hipStreamWaitEvent(s, c->ev_keys, 0)) != hipSuccess) return (int)e;
  rc = launch_ed_mul_direct(Q, s);  // exits at once for the lanes that walk a table: enqueued before the wait for the tables
  if (rc) return rc;
  // The table walk can follow the table build part by part (partial sums in mulout).  Measured: +4 % step time at 256-512 proofs (the
  // parts sit on the caller's normal-priority queue beside the chain), -2.7 % at 1024, where the walk dominates -> by batch size.
  const bool walk_parts = parts > 1 && (K.walk_parts >= 0 ? K.walk_parts == 1 : n_lanes >= 65536);
  for (uint32_t p = 0; p < parts; p++) {
    if ((e = hipStreamWaitEvent(s, c->ev_part[p], 0)) != hipSuccess) return (int)e;
    if (p + 1 < parts) {
      rc = launch_ed_tab_mult(Q, p, parts, s);
      if (rc) return rc;
    }
    if (walk_parts) {
      rc = launch_ed_mul_tab(Q, p, parts, s);
      if (rc) return rc;
    }
  }
  if (!walk_parts) {
    rc = launch_ed_mul_tab(Q, 0, 1, s);
    if (rc) return rc;
  }
  rc = launch_ed_fin(Q, s);
  c->fin_done_attached = rc == 0 && Q.fin_done != nullptr && n_lanes != 0;
  return rc;
}

extern "C" {

uint32_t tmx_version(void) { return 0x000100; }

const char* tmx_status_str(int32_t s) {
  switch (s) {
    case TMX_OK: return "ok";
    case TMX_ERR_BAD_ARG: return "bad argument";
    case TMX_ERR_SET_TOO_LARGE: return "validator set larger than VALIDATOR_SET_SIZE_MAX";
    case TMX_ERR_HIP: return "HIP error (no usable MI355X device?)";
    case TMX_ERR_CAPACITY: return "capacity too small";
    case TMX_ERR_PARSE: return "malformed JSON";
    case TMX_ERR_MSG_TOO_LONG: return "sign-bytes longer than 124 bytes";
    case TMX_ERR_RCCL: return "RCCL error";
    default: return "unknown";
  }
}

uint64_t tmx_elem_count(int32_t kind, uint32_t n) {
  if ((kind != TMX_KIND_SKIP && kind != TMX_KIND_STEP) || n == 0 || n > TMX_N_MAX_LIMIT) return 0;
  const uint64_t tn = tree_nodes(n);
  if (kind == TMX_KIND_SKIP) return 1776ull * n + 5320 + 1235ull * n + 630ull * n + 2 * tn * 256 + (4 * 1280 + 88) + 34;
  return 1517ull * n + 6919 + 1235ull * n + tn * 256 + (5 * 1280 + 88) + 25;
}
uint64_t tmx_elem_stride(int32_t kind, uint32_t n) { return (tmx_elem_count(kind, n) + 15) & ~15ull; }  // rows start on a 128-byte line
uint64_t tmx_hint_elem_count(int32_t kind, uint32_t n) {
  if (n == 0 || n > TMX_N_MAX_LIMIT) return 0;
  return kind == TMX_KIND_SKIP ? 1776ull * n + 5320 : 1517ull * n + 6919;
}

const char* tmx_last_error(const tmx_ctx* ctx) { return ctx ? ctx->err.c_str() : (g_tls_err.empty() ? "null context" : g_tls_err.c_str()); }

}  // extern "C"

// One set of streams per device, shared by every context of the process.  The HIP runtime multiplexes streams onto at most four
// hardware queues per priority level, and two streams that land on the same queue lose their concurrency (measured: a step goes
// from 0.86 ms to 1.3-1.7 ms once a fifth stream of a level exists) -- so streams are treated as a scarce per-process resource.
// Sharing is safe: every cross-stream dependency of a batch is an event of its own context; stream order only adds ordering.
namespace {
struct DeviceStreams {
  hipStream_t stream = nullptr, side = nullptr, side2 = nullptr, side3 = nullptr;
  int refs = 0;
};
std::mutex g_streams_mu;
DeviceStreams g_streams[64];

hipError_t acquire_streams(int device, DeviceStreams** out) {
  std::lock_guard<std::mutex> lk(g_streams_mu);
  DeviceStreams& d = g_streams[device];
  if (d.refs == 0) {
    // The key pipeline is on the critical path (high priority); the early serialization only fills otherwise idle HBM cycles and
    // must not starve the caller's stream (low priority).  k_proof's stream is at the caller's priority: the sections that wait for
    // it are the last thing a step writes, and at low priority the kernel stretched from 0.30 ms (alone) to 0.5 ms (measured:
    // normal -2 % step time at 256 proofs, -0.8 % at 1024; high +40 %: two high-priority queues fight).
    int prio_low = 0, prio_high = 0;
    hipError_t e;
    if ((e = hipDeviceGetStreamPriorityRange(&prio_low, &prio_high)) != hipSuccess) return e;
    if ((e = hipStreamCreateWithPriority(&d.side, hipStreamNonBlocking, (prio_low + prio_high) / 2)) != hipSuccess) return e;
    if ((e = hipStreamCreateWithPriority(&d.side2, hipStreamNonBlocking, prio_high)) != hipSuccess) return e;
    if ((e = hipStreamCreateWithPriority(&d.side3, hipStreamNonBlocking, prio_low)) != hipSuccess) return e;
  }
  d.refs++;
  *out = &d;
  return hipSuccess;
}
// The normal-priority stream the host-buffer entry points work on: created on first use, because the device-pointer entry points
// run on the caller's stream and an idle extra stream would still take one of the four normal-priority hardware queues.
hipError_t host_stream(int device, hipStream_t* out) {
  std::lock_guard<std::mutex> lk(g_streams_mu);
  DeviceStreams& d = g_streams[device];
  if (!d.stream) {
    hipError_t e = hipStreamCreateWithFlags(&d.stream, hipStreamNonBlocking);
    if (e != hipSuccess) return e;
  }
  *out = d.stream;
  return hipSuccess;
}
void release_streams(int device) {
  std::lock_guard<std::mutex> lk(g_streams_mu);
  DeviceStreams& d = g_streams[device];
  if (d.refs > 0 && --d.refs == 0) {
    if (d.side3) (void)hipStreamDestroy(d.side3);
    if (d.side2) (void)hipStreamDestroy(d.side2);
    if (d.side) (void)hipStreamDestroy(d.side);
    if (d.stream) (void)hipStreamDestroy(d.stream);
    d = DeviceStreams();
  }
}
}  // namespace

// The EdDSA values of a lane that did not sign -- the dummy public key, signature and message of plonky2x (verify.rs:248-259 evaluates every
// such lane on them) -- are the same for every lane of every proof: computed ONCE per context, by the same kernels, on a one-lane launch over
// an all-zero lane record (flags = 0: did not sign); the key cache is emptied again afterwards, so a context starts as it always did.
static int32_t dummy_record(tmx_ctx* c) {
  // (on the context's own side stream, idle at creation: a stream created and destroyed here would shift which hardware queues the
  // streams of the process's NEXT context land on -- second context of a process: 0.63 instead of 0.38 ms per step, tools/churn_probe.py)
  hipStream_t ts = c->side;
  int32_t st = TMX_OK;
  if (use_tiny(c, 1)) st = run_tiny_lanes(c, 1, c->d_dummy_in, c->d_dummy_ed, ED_STRIDE, ts);
  else if (int rc = run_eddsa(c, 1, c->d_dummy_in, c->d_dummy_ed, ED_STRIDE, ts)) st = fail(c, TMX_ERR_HIP, std::string("dummy record: ") + hipGetErrorString((hipError_t)rc));
  hipError_t e = hipStreamSynchronize(ts);
  if (e == hipSuccess) e = hipStreamSynchronize(c->side2);
  if (e == hipSuccess) e = hipStreamSynchronize(c->side);
  if (e == hipSuccess) e = hipStreamSynchronize(c->side3);
  if (st) return st;
  if (e != hipSuccess) return fail(c, TMX_ERR_HIP, std::string("dummy record: ") + hipGetErrorString(e));
  int rc = launch_kc_reset(c->kc, c->side2);
  if (rc) return fail(c, TMX_ERR_HIP, std::string("k_kc_reset launch: ") + hipGetErrorString((hipError_t)rc));
  HIPCK(c, hipStreamSynchronize(c->side2));
  c->h_hint[0] = 0; c->h_hint[1] = 0;
  c->last_stream_valid = false;
  c->dummy_ready = true;
  return TMX_OK;
}

// (re)allocate the key cache for `keys` slots and empty it.  The caller has made sure nothing of this context is in flight.
// Transactional: the new buffers are allocated and reset first; the old cache is freed only when all of that succeeded, so a failed
// (e.g. oversized) request leaves the context exactly as it was.  A request whose tables cannot fit in the device's free memory is
// refused up front with TMX_ERR_CAPACITY.
static int32_t alloc_key_cache(tmx_ctx* c, uint32_t keys) {
  const size_t lanes = (size_t)c->cfg.max_batch * c->cfg.n_max;
  if (keys > (1u << 20)) keys = 1u << 20;
  if (keys == 0) keys = 1;
  uint64_t hsz = 1;
  while (hsz < 4 * (uint64_t)keys) hsz <<= 1;
  KeyCache kc = c->kc;
  kc.cap = keys;
  kc.hash_mask = (uint32_t)(hsz - 1);
  kc.new_cap = (uint32_t)std::min<size_t>(std::min<size_t>(keys, 4096), lanes);  // tables one launch builds at most: the anchor scratch
  kc.persist = c->knobs.key_cache ? 1u : 0u;
  kc.hint = const_cast<uint32_t*>(c->h_hint);
  kc.d_hash = kc.d_pk = kc.d_used = kc.d_free = kc.d_state = nullptr;
  const size_t sizes[8] = {(size_t)hsz * 4, (size_t)keys * 32, (size_t)keys * 4, (size_t)keys * 4, (size_t)KC_STATE_WORDS * 4,
                           ((size_t)keys + lanes) * key_bytes_per_key(), (size_t)kc.new_cap * anchor_bytes_per_key(), (size_t)keys * keytab_bytes_per_key()};
  {
    size_t want = 0, free_b = 0, total_b = 0;
    for (size_t b : sizes) want += b;
    // (the new cache is allocated before the old one is released, so it has to fit beside it)
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && want > free_b)
      return fail(c, TMX_ERR_CAPACITY, "key cache of " + std::to_string(keys) + " keys needs " + std::to_string(want >> 20) + " MiB, " +
                                           std::to_string(free_b >> 20) + " MiB free on the device");
  }
  void* nb[8] = {};
  hipError_t e = hipSuccess;
  for (int i = 0; i < 8 && e == hipSuccess; i++) e = hipMalloc(&nb[i], sizes[i]);
  if (e == hipSuccess) {
    kc.d_hash = (uint32_t*)nb[0]; kc.d_pk = (uint32_t*)nb[1]; kc.d_used = (uint32_t*)nb[2]; kc.d_free = (uint32_t*)nb[3]; kc.d_state = (uint32_t*)nb[4];
    e = (hipError_t)launch_kc_reset(kc, c->side2);
    if (e == hipSuccess) e = hipStreamSynchronize(c->side2);
  }
  if (e != hipSuccess) {  // the old cache stays in place and usable
    (void)hipGetLastError();
    for (void* b : nb)
      if (b) (void)hipFree(b);
    return fail(c, e == hipErrorOutOfMemory ? TMX_ERR_CAPACITY : TMX_ERR_HIP, std::string("key cache allocation: ") + hipGetErrorString(e));
  }
  void* old[] = {c->kc.d_hash, c->kc.d_pk, c->kc.d_used, c->kc.d_free, c->kc.d_state, c->d_keyrec, c->d_anchors, c->d_keytab};
  for (void* b : old)
    if (b) (void)hipFree(b);
  c->kc = kc;
  c->d_keyrec = nb[5]; c->d_anchors = nb[6]; c->d_keytab = nb[7];
  c->h_hint[0] = 0; c->h_hint[1] = 0;
  return TMX_OK;
}

extern "C" {

void tmx_ctx_destroy(tmx_ctx* c) {
  if (!c) return;
  (void)tmx_comm_destroy(c);
  if (c->have_streams) {  // side streams may still hold work enqueued by the last call (hash-table reset for the next launch)
    (void)hipSetDevice(c->cfg.device);
    (void)hipDeviceSynchronize();
  }
  void* bufs[] = {c->d_lut[0], c->d_lut[1], c->d_wave_sec[0], c->d_wave_sec[1], c->d_seams[0], c->d_seams[1], c->d_table, c->d_qtable, c->d_pre, c->d_mulout, c->d_hash, c->d_cnt, c->d_live, c->d_dummy_ed, c->d_dummy_in, c->setc.table, c->setc.state, c->setc.slots, c->d_span_ctr, c->d_owner_of, c->d_slot_of_owner, c->d_slot_of_uid, c->d_owners, c->d_keyrec,
                  c->d_anchors, c->d_keytab, c->kc.d_hash, c->kc.d_pk, c->kc.d_used, c->kc.d_free, c->kc.d_state, c->d_ed, c->d_tl, c->d_lr, c->d_pf, c->d_nodes_t, c->d_nodes_r,
                  c->d_reports, c->d_in_proofs, c->d_in_targets, c->d_in_trusteds, c->d_out, c->d_pack, c->d_trace_tmp, c->d_tiny, c->d_shadow, c->d_commit,
                  c->d_val_lut[0], c->d_val_lut[1], c->d_value};
  for (void* b : bufs)
    if (b) (void)hipFree(b);
  for (void* w : c->d_ntt_w)
    if (w) (void)hipFree(w);
  for (auto& dir : c->d_ntt_m)
    for (void* m : dir)
      if (m) (void)hipFree(m);
  for (void* v : c->d_lde_s)
    if (v) (void)hipFree(v);
  if (c->d_ntt_tmp) (void)hipFree(c->d_ntt_tmp);
  if (c->d_pos_consts) (void)hipFree(c->d_pos_consts);
  for (auto& set : c->ev)
    for (auto& e : set)
      if (e) (void)hipEventDestroy(e);
  for (auto& set : c->ev_side)
    for (auto& e : set)
      if (e) (void)hipEventDestroy(e);
  if (c->ev_join) (void)hipEventDestroy(c->ev_join);
  if (c->ev_fork2) (void)hipEventDestroy(c->ev_fork2);
  for (hipEvent_t ev : c->ev_part)
    if (ev) (void)hipEventDestroy(ev);
  if (c->ev_keys) (void)hipEventDestroy(c->ev_keys);
  if (c->ev_direct) (void)hipEventDestroy(c->ev_direct);
  if (c->h_hint) (void)hipHostFree((void*)c->h_hint);
  if (c->ev_hash) (void)hipEventDestroy(c->ev_hash);
  if (c->ev_leaves) (void)hipEventDestroy(c->ev_leaves);
  if (c->ev_hash_clean) (void)hipEventDestroy(c->ev_hash_clean);
  if (c->ev_join3) (void)hipEventDestroy(c->ev_join3);
  if (c->ev_value) (void)hipEventDestroy(c->ev_value);
  if (c->ev_base) (void)hipEventDestroy(c->ev_base);
  for (hipEvent_t e : c->ev_trace_rest)
    if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : c->ev_trace)
    if (e) (void)hipEventDestroy(e);
  if (c->ev_tail) (void)hipEventDestroy(c->ev_tail);
  for (hipEvent_t e : c->ev_commit)
    if (e) (void)hipEventDestroy(e);
  if (c->have_streams) release_streams(c->cfg.device);
  delete c;
}

int32_t tmx_ctx_create(const tmx_config* cfg, tmx_ctx** out) {
  if (!cfg || !out) return TMX_ERR_BAD_ARG;
  *out = nullptr;
  if (cfg->n_max == 0 || cfg->n_max > TMX_N_MAX_LIMIT || cfg->chain_id_len > 50 || cfg->max_batch == 0) return TMX_ERR_BAD_ARG;
  tmx_ctx* c = new tmx_ctx();
  c->cfg = *cfg;
  *out = c;  // returned even on HIP failure so that tmx_last_error is readable; caller destroys it
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0) return fail(c, TMX_ERR_HIP, std::string("no HIP device: ") + hipGetErrorString(e));
  if (cfg->device < 0 || cfg->device >= ndev) return fail(c, TMX_ERR_BAD_ARG, "device ordinal out of range");
  HIPCK(c, hipSetDevice(cfg->device));
  if (cfg->device >= 64) return fail(c, TMX_ERR_BAD_ARG, "device ordinal out of range");
  DeviceStreams* ds = nullptr;
  HIPCK(c, acquire_streams(cfg->device, &ds));
  c->side = ds->side; c->side2 = ds->side2; c->side3 = ds->side3;
  c->have_streams = true;
  HIPCK(c, hipEventCreateWithFlags(&c->ev_join3, hipEventDisableTiming));
  HIPCK(c, hipEventCreateWithFlags(&c->ev_value, hipEventDisableTiming));
  HIPCK(c, hipEventCreateWithFlags(&c->ev_base, hipEventDisableTiming));
  HIPCK(c, hipEventCreateWithFlags(&c->ev_tail, hipEventDisableTiming));
  HIPCK(c, hipEventCreateWithFlags(&c->ev_fork2, hipEventDisableTiming));
  for (auto& ev : c->ev_part) HIPCK(c, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  c->knobs = read_knobs();
  HIPCK(c, hipEventCreateWithFlags(&c->ev_keys, hipEventDisableTiming));
  HIPCK(c, hipEventCreateWithFlags(&c->ev_direct, hipEventDisableTiming));
  {
    void* hp = nullptr;
    HIPCK(c, hipHostMalloc(&hp, 64, hipHostMallocDefault));
    std::memset(hp, 0, 64);
    c->h_hint = reinterpret_cast<volatile uint32_t*>(hp);
  }
  HIPCK(c, hipEventCreateWithFlags(&c->ev_hash, hipEventDisableTiming));
  HIPCK(c, hipEventCreateWithFlags(&c->ev_leaves, hipEventDisableTiming));
  HIPCK(c, hipEventCreateWithFlags(&c->ev_hash_clean, hipEventDisableTiming));
  HIPCK(c, hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
  for (auto& set : c->ev_side)
    for (auto& ev : set) HIPCK(c, hipEventCreate(&ev));
  for (auto& set : c->ev)
    for (auto& ev : set) HIPCK(c, hipEventCreate(&ev));
  const uint32_t n = cfg->n_max;
  const size_t B = cfg->max_batch, lanes = B * n;
  if (lanes > ((size_t)1 << 30)) return fail(c, TMX_ERR_CAPACITY, "max_batch * n_max exceeds 2^30 lanes");
  for (int k = 0; k < 2; k++) {
    c->prog[k] = build_program(k, n);
    c->prog[k].sp.lane_fast = c->knobs.ser_lanes ? 1u : 0u;
    if (c->prog[k].sp.elem_count != tmx_elem_count(k, n) || c->prog[k].d1b_start == 0xffffffffu) return fail(c, TMX_ERR_BAD_ARG, "internal: layout size mismatch");
    HIPCK(c, hipMalloc(&c->d_lut[k], c->prog[k].lut.size() * 4));
    HIPCK(c, hipMemcpyAsync(c->d_lut[k], c->prog[k].lut.data(), c->prog[k].lut.size() * 4, hipMemcpyHostToDevice, c->side2));
    HIPCK(c, hipMalloc(&c->d_wave_sec[k], c->prog[k].wave_sec.size()));
    HIPCK(c, hipMemcpyAsync(c->d_wave_sec[k], c->prog[k].wave_sec.data(), c->prog[k].wave_sec.size(), hipMemcpyHostToDevice, c->side2));
    HIPCK(c, hipMalloc(&c->d_seams[k], c->prog[k].seam_waves.size() * 4 + 4));
    HIPCK(c, hipMemcpyAsync(c->d_seams[k], c->prog[k].seam_waves.data(), c->prog[k].seam_waves.size() * 4, hipMemcpyHostToDevice, c->side2));
  }
  HIPCK(c, hipMalloc(&c->d_ed, lanes * ED_STRIDE));
  HIPCK(c, hipMalloc(&c->d_tl, lanes * TL_STRIDE));
  HIPCK(c, hipMalloc(&c->d_lr, lanes * LANE_STRIDE));
  HIPCK(c, hipMalloc(&c->d_pf, B * PF_STRIDE));
  const size_t tn = tree_nodes(n);
  HIPCK(c, hipMalloc(&c->d_nodes_t, B * (tn + 1) * 32));
  HIPCK(c, hipMalloc(&c->d_nodes_r, B * (tn + 1) * 32));
  HIPCK(c, hipMalloc(&c->d_reports, B * sizeof(tmx_report)));
  HIPCK(c, hipMalloc(&c->d_tiny, tiny_counter_words(cfg->max_batch) * 4));
  HIPCK(c, hipMemsetAsync(c->d_tiny, 0, tiny_counter_words(cfg->max_batch) * 4, c->side2));
  HIPCK(c, hipMalloc(&c->d_shadow, (size_t)TINY_MAX_LANES * VR_STRIDE));
  HIPCK(c, hipMemsetAsync(c->d_shadow, 0, (size_t)TINY_MAX_LANES * VR_STRIDE, c->side2));
  // fixed-base table of B (13-bit signed windows: kernels.hip BASE_W), affine form, then the quad layout the kernels read
  HIPCK(c, hipMalloc(&c->d_table, base_table_bytes()));
  int rc = launch_init_base(c->d_table, c->side2);
  if (rc) return fail(c, TMX_ERR_HIP, std::string("k_init_base launch: ") + hipGetErrorString((hipError_t)rc));
  HIPCK(c, hipMalloc(&c->d_qtable, quad_table_bytes()));
  HIPCK(c, hipMalloc(&c->d_pre, lanes * pre_bytes_per_lane()));
  HIPCK(c, hipMalloc(&c->d_mulout, lanes * mulout_bytes_per_lane()));
  {
    uint64_t cap = 1;
    while (cap < 2 * (uint64_t)lanes) cap <<= 1;  // <= 2^31: the open-addressing table of the launch's own key deduplication
    c->hash_mask = (uint32_t)(cap - 1);
    HIPCK(c, hipMalloc(&c->d_hash, (size_t)cap * 4));
    HIPCK(c, hipMalloc(&c->d_cnt, 64));
    HIPCK(c, hipMalloc(reinterpret_cast<void**>(&c->d_span_ctr), 256));
    HIPCK(c, hipMemsetAsync(c->d_span_ctr, 0, 256, c->side2));
    HIPCK(c, hipMemsetAsync(c->d_hash, 0xff, (size_t)cap * 4, c->side2));
    HIPCK(c, hipMemsetAsync(c->d_cnt, 0, 64, c->side2));
    HIPCK(c, hipMalloc(&c->d_live, lanes * 4));
    if (c->knobs.set_cache) {  // 256 validator sets (20 KB each at N = 128, 82 KB at N = 512), a table of 1024 entries
      SetCache& sc = c->setc;
      // (TMX_SET_CACHE_SETS: a smaller cache for the eviction tests; the table keeps four entries per slot)
      const char* v = std::getenv("TMX_SET_CACHE_SETS");
      const long want = v ? std::atol(v) : 0;
      sc.cap = want >= 4 && want <= (long)SETC_MAX_SLOTS ? (uint32_t)want : SETC_MAX_SLOTS;
      sc.tab_mask = 1023; sc.slot_bytes = (setcache_slot_bytes(n, tree_nodes(n)) + 63u) & ~63u;
      HIPCK(c, hipMalloc(reinterpret_cast<void**>(&sc.table), (sc.tab_mask + 1) * 4));
      HIPCK(c, hipMalloc(reinterpret_cast<void**>(&sc.state), (SETC_STATE_WORDS + SETC_MAX_SLOTS) * 4));
      HIPCK(c, hipMalloc(reinterpret_cast<void**>(&sc.slots), (size_t)sc.cap * sc.slot_bytes));
      HIPCK(c, hipMemsetAsync(sc.table, 0, (sc.tab_mask + 1) * 4, c->side2));
      HIPCK(c, hipMemsetAsync(sc.state, 0, (SETC_STATE_WORDS + SETC_MAX_SLOTS) * 4, c->side2));
    }
    HIPCK(c, hipMalloc(&c->d_dummy_ed, 512));
    HIPCK(c, hipMalloc(&c->d_dummy_in, VR_STRIDE));
    HIPCK(c, hipMemsetAsync(c->d_dummy_ed, 0, 512, c->side2));
    HIPCK(c, hipMemsetAsync(c->d_dummy_in, 0, VR_STRIDE, c->side2));
    HIPCK(c, hipMalloc(&c->d_owner_of, lanes * 4));
    HIPCK(c, hipMalloc(&c->d_slot_of_owner, lanes * 4));
    HIPCK(c, hipMalloc(&c->d_slot_of_uid, lanes * 4));
    HIPCK(c, hipMalloc(&c->d_owners, lanes * 4));
  }
  // key cache: by default room for the keys of 4 x max_batch x n_max lanes at 8 lanes per key, between 1024 and 8192 keys (655 KB each:
  // 0.7 - 5.4 GB of the 288 GB); TMX_KEY_CACHE_KEYS / tmx_key_cache_resize choose another capacity
  {
    size_t keys = c->knobs.key_cache_keys ? c->knobs.key_cache_keys : std::min<size_t>(8192, std::max<size_t>(1024, lanes / 2));
    // TMX_KEY_CACHE=0: a table lives for one launch only, so the slots one launch can build (<= 4096, <= lanes) are all it ever uses
    if (!c->knobs.key_cache && !c->knobs.key_cache_keys) keys = std::max<size_t>(1, std::min<size_t>(std::min<size_t>(4096, lanes), keys));
    int32_t st = alloc_key_cache(c, (uint32_t)keys);
    if (st) return st;
  }
  rc = launch_init_base_quad(c->d_table, c->d_qtable, c->side2);
  if (rc) return fail(c, TMX_ERR_HIP, std::string("k_init_base_quad launch: ") + hipGetErrorString((hipError_t)rc));
  HIPCK(c, hipEventRecord(c->ev_hash_clean, c->side2));
  HIPCK(c, hipStreamSynchronize(c->side2));
  return dummy_record(c);
}

static int32_t ensure_host_stream(tmx_ctx* c) {
  if (c->stream) return TMX_OK;
  HIPCK(c, hipSetDevice(c->cfg.device));
  HIPCK(c, host_stream(c->cfg.device, &c->stream));
  return TMX_OK;
}
void* tmx_ctx_stream(tmx_ctx* c) { return c && ensure_host_stream(c) == TMX_OK ? reinterpret_cast<void*>(c->stream) : nullptr; }

// Nothing of this context may be in flight when the cache is reshaped: wait for the last batch and the side streams' tails.
static int32_t quiesce(tmx_ctx* c) {
  HIPCK(c, hipSetDevice(c->cfg.device));
  if (c->last_stream_valid) HIPCK(c, hipEventSynchronize(c->ev_done));
  HIPCK(c, hipStreamSynchronize(c->side2));
  HIPCK(c, hipStreamSynchronize(c->side));
  HIPCK(c, hipStreamSynchronize(c->side3));
  if (c->stream) HIPCK(c, hipStreamSynchronize(c->stream));
  return TMX_OK;
}
static int32_t read_kc_state(tmx_ctx* c, uint32_t st[KC_STATE_WORDS]) {
  int32_t q = quiesce(c);
  if (q) return q;
  HIPCK(c, hipMemcpy(st, c->kc.d_state, KC_STATE_WORDS * 4, hipMemcpyDeviceToHost));
  return TMX_OK;
}

// distinct effective public keys among the lanes of the last EdDSA launch (resident ones that were hit + new ones) and whether any lane
// walked a per-key table (blocks until that launch is done)
int32_t tmx_last_dedup(tmx_ctx* c, uint32_t* n_unique, uint32_t* used_tables) {
  if (!c || !n_unique || !used_tables) return TMX_ERR_BAD_ARG;
  *n_unique = 0; *used_tables = 0;
  if (c->last_lanes == 0) return TMX_OK;
  uint32_t st[KC_STATE_WORDS];
  int32_t r = read_kc_state(c, st);
  if (r) return r;
  *n_unique = st[KC_LAST_NEW] + st[KC_LAST_HIT_KEYS];
  *used_tables = (c->knobs.dedup_mode != 0 && (st[KC_LAST_HIT_LANES] != 0 || (st[KC_LAST_BUILT] != 0 && st[KC_LAST_USE_NEW] != 0))) ? 1u : 0u;
  return TMX_OK;
}

int32_t tmx_key_cache_stats(tmx_ctx* c, tmx_key_cache_info* out) {
  if (!c || !out) return TMX_ERR_BAD_ARG;
  uint32_t st[KC_STATE_WORDS];
  int32_t r = read_kc_state(c, st);
  if (r) return r;
  uint64_t tot[8];
  std::memcpy(tot, st + KC_TOTALS, sizeof(uint64_t) * 6);
  std::memset(out, 0, sizeof *out);
  out->capacity_keys = c->kc.cap; out->resident_keys = st[KC_RESIDENT]; out->enabled = c->kc.persist; out->epoch = st[KC_EPOCH];
  out->bytes_per_key = keytab_bytes_per_key() + key_bytes_per_key() + 32;
  out->last_new_keys = st[KC_LAST_NEW]; out->last_hit_keys = st[KC_LAST_HIT_KEYS]; out->last_hit_lanes = st[KC_LAST_HIT_LANES];
  out->last_built_keys = st[KC_LAST_BUILT];
  out->hit_lanes = tot[KC_TOT_HIT_LANES]; out->miss_lanes = tot[KC_TOT_MISS_LANES]; out->built_keys = tot[KC_TOT_BUILT];
  out->evicted_keys = tot[KC_TOT_EVICTED]; out->evictions = tot[KC_TOT_GC_RUNS]; out->launches = tot[KC_TOT_LAUNCHES];
  return TMX_OK;
}

int32_t tmx_key_cache_flush(tmx_ctx* c) {
  if (!c) return TMX_ERR_BAD_ARG;
  int32_t q = quiesce(c);
  if (q) return q;
  int rc = launch_kc_reset(c->kc, c->side2);
  if (rc) return fail(c, TMX_ERR_HIP, std::string("k_kc_reset launch: ") + hipGetErrorString((hipError_t)rc));
  if (c->setc.table) {  // (the validator-set cache goes with it: "flush" = a context as new)
    HIPCK(c, hipMemsetAsync(c->setc.table, 0, (c->setc.tab_mask + 1) * 4, c->side2));
    HIPCK(c, hipMemsetAsync(c->setc.state, 0, (SETC_STATE_WORDS + SETC_MAX_SLOTS) * 4, c->side2));
  }
  HIPCK(c, hipStreamSynchronize(c->side2));
  c->h_hint[0] = 0; c->h_hint[1] = 0;  // an empty cache: the next enqueue takes the cold schedule, as the first call of a context does
  return TMX_OK;
}

int32_t tmx_set_cache_stats(tmx_ctx* c, uint32_t out[8]) {
  if (!c || !out) return TMX_ERR_BAD_ARG;
  std::memset(out, 0, 8 * sizeof(uint32_t));
  if (!c->setc.table) return TMX_OK;
  int32_t q = quiesce(c);
  if (q) return q;
  uint32_t st[SETC_STATE_WORDS];
  HIPCK(c, hipMemcpy(st, c->setc.state, sizeof st, hipMemcpyDeviceToHost));
  const uint32_t hi = st[0] > c->setc.cap ? c->setc.cap : st[0], fr = (int32_t)st[4] > 0 ? st[4] : 0u;
  out[0] = hi - (fr > hi ? hi : fr);  // resident = slots ever used - slots on the free list
  out[1] = st[1]; out[2] = st[2]; out[3] = st[3]; out[4] = st[6]; out[5] = c->setc.cap;
  return TMX_OK;
}

int32_t tmx_key_cache_config(tmx_ctx* c, uint32_t enabled, uint32_t max_keys) {
  if (!c) return TMX_ERR_BAD_ARG;
  int32_t q = quiesce(c);
  if (q) return q;
  const bool was = c->knobs.key_cache;
  c->knobs.key_cache = enabled != 0;
  if (max_keys != 0 && max_keys != c->kc.cap) {
    int32_t st = alloc_key_cache(c, max_keys);  // (on failure the old cache, and the old setting, stay)
    if (st) { c->knobs.key_cache = was; return st; }
  } else if (was && !enabled) {
    // persistence switched off: every launch starts from an empty cache, so the resident keys go now (a full cache would otherwise
    // leave the launches without free slots, i.e. without per-launch tables)
    int rc = launch_kc_reset(c->kc, c->side2);
    if (rc) return fail(c, TMX_ERR_HIP, std::string("k_kc_reset launch: ") + hipGetErrorString((hipError_t)rc));
    HIPCK(c, hipStreamSynchronize(c->side2));
    c->h_hint[0] = 0; c->h_hint[1] = 0;
  }
  c->kc.persist = enabled ? 1u : 0u;
  return TMX_OK;
}

int32_t tmx_sync(tmx_ctx* c) {
  if (!c) return TMX_ERR_BAD_ARG;
  if (c->stream) HIPCK(c, hipStreamSynchronize(c->stream));  // (the host-buffer entry points already synchronise before returning)
  return TMX_OK;
}

int32_t tmx_witness_batch_device(tmx_ctx* c, int32_t kind, uint32_t n_proofs, const void* d_proofs, const void* d_targets,
                                 const void* d_trusteds, void* d_out_elems, void* d_reports, void* hip_stream) {
  int32_t st = check_batch_args(c, kind, n_proofs, d_proofs, d_targets, d_trusteds);
  if (st) return st;
  if (n_proofs == 0) return TMX_OK;
  hipStream_t s = reinterpret_cast<hipStream_t>(hip_stream);  // NULL = the HIP default stream, exactly as passed
  if (use_tiny(c, (uint64_t)n_proofs * c->cfg.n_max)) return run_tiny(c, kind, n_proofs, d_proofs, d_targets, d_trusteds, d_out_elems, d_reports, s);
  return run_batch(c, kind, n_proofs, d_proofs, d_targets, d_trusteds, d_out_elems, d_reports, s, true, [&](hipStream_t ss) -> int32_t {
    int rc = run_eddsa(c, n_proofs * c->cfg.n_max, d_targets, reinterpret_cast<uint8_t*>(c->d_tl) + TL_OFF_ED, TL_STRIDE, ss);
    if (rc) return fail(c, TMX_ERR_HIP, std::string("EdDSA kernel launch: ") + hipGetErrorString((hipError_t)rc));
    return TMX_OK;
  });
}

int32_t tmx_eddsa_lanes_device(tmx_ctx* c, uint32_t n_lanes, const void* d_lanes, void* d_ed_out, void* hip_stream) {
  if (!c || !d_lanes || !d_ed_out) return TMX_ERR_BAD_ARG;
  if (n_lanes == 0) return TMX_OK;
  hipStream_t s = reinterpret_cast<hipStream_t>(hip_stream);
  if ((uint64_t)n_lanes > (uint64_t)c->cfg.max_batch * c->cfg.n_max) return fail(c, TMX_ERR_CAPACITY, "n_lanes exceeds max_batch * n_max");
  if (use_tiny(c, n_lanes)) return run_tiny_lanes(c, n_lanes, d_lanes, d_ed_out, ED_STRIDE, s);
  int rc = run_eddsa(c, n_lanes, d_lanes, d_ed_out, ED_STRIDE, s);
  if (rc) return fail(c, TMX_ERR_HIP, std::string("EdDSA kernel launch: ") + hipGetErrorString((hipError_t)rc));
  return TMX_OK;
}

int32_t tmx_finish_batch_device(tmx_ctx* c, int32_t kind, uint32_t n_proofs, const void* d_proofs, const void* d_targets,
                                const void* d_trusteds, const void* d_ed, void* d_out_elems, void* d_reports, void* hip_stream) {
  int32_t st = check_batch_args(c, kind, n_proofs, d_proofs, d_targets, d_trusteds);
  if (st) return st;
  if (!d_ed) return TMX_ERR_BAD_ARG;
  if (n_proofs == 0) return TMX_OK;
  hipStream_t s = reinterpret_cast<hipStream_t>(hip_stream);
  return run_batch(c, kind, n_proofs, d_proofs, d_targets, d_trusteds, d_out_elems, d_reports, s, false, [&](hipStream_t ss) -> int32_t {
    // caller's records are 448 B apart; place them into the ED part of the unified per-lane records
    HIPCK(c, hipMemcpy2DAsync(reinterpret_cast<uint8_t*>(c->d_tl) + TL_OFF_ED, TL_STRIDE, d_ed, ED_STRIDE, ED_STRIDE,
                              (size_t)n_proofs * c->cfg.n_max, hipMemcpyDeviceToDevice, ss));
    return TMX_OK;
  });
}

uint64_t tmx_trace_elem_count(int32_t kind, uint32_t n) {
  if ((kind != TMX_KIND_SKIP && kind != TMX_KIND_STEP) || n == 0 || n > TMX_N_MAX_LIMIT) return 0;
  return trace_elems((uint32_t)kind, n);
}

}  // extern "C"
// lane0 / lane_count: the ladder and SHA-512 rows (the per-lane sections) of lanes [lane0, lane0 + lane_count) of the batch only -- the
// lane-sharded form; the other sections are written for every proof either way
static int32_t trace_rows_impl(tmx_ctx* c, int32_t kind, uint32_t n_proofs, const void* d_targets, const void* d_trusteds, void* d_trace_out,
                               uint32_t sections, void* hip_stream, uint32_t lane0, uint32_t lane_count) {
  if (!c || (kind != TMX_KIND_SKIP && kind != TMX_KIND_STEP) || !d_targets || !d_trace_out || (sections & ~(uint32_t)TMX_TRACE_ALL) || sections == 0)
    return TMX_ERR_BAD_ARG;
  if (kind == TMX_KIND_SKIP && !d_trusteds) return fail(c, TMX_ERR_BAD_ARG, "skip needs the trusted hash fields");
  if (n_proofs > c->cfg.max_batch) return fail(c, TMX_ERR_CAPACITY, "n_proofs exceeds the context's max_batch");
  if (n_proofs == 0) return TMX_OK;
  // the trace kernels read the Level-1 lane records of the context: they must be those of a batch of this kind and at least this size
  if (c->last_kind != kind || c->last_n_proofs < n_proofs)
    return fail(c, TMX_ERR_BAD_ARG, "tmx_trace_rows_device: call tmx_witness_batch_device for the same batch (kind, >= n_proofs) first");
  if ((sections & TMX_TRACE_LADDERS) && !c->d_trace_tmp) {  // 82 KB per ladder between the two ladder passes: allocated (blocking) on the first call
    HIPCK(c, hipSetDevice(c->cfg.device));
    HIPCK(c, hipMalloc(&c->d_trace_tmp, trace_tmp_bytes(c->cfg.n_max, c->cfg.max_batch)));
  }
  hipStream_t s = reinterpret_cast<hipStream_t>(hip_stream);
  const uint8_t* edr = reinterpret_cast<const uint8_t*>(c->d_tl) + TL_OFF_ED;
  const uint32_t n = c->cfg.n_max;
  int rc = 0;
  // The other sections (0.6 ms of short, memory-bound kernels per 256-proof batch) beside the ladders, whose first segment is one
  // latency-bound wave per SIMD with nothing else on the chip: on the low-priority side stream, joined at the end (all sections 5.6 -> 5.3 ms)
  const bool rest_aside = (sections & TMX_TRACE_LADDERS) && (sections & ~(uint32_t)TMX_TRACE_LADDERS);
  if (rest_aside) {
    for (auto& e : c->ev_trace_rest)
      if (!e) HIPCK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    HIPCK(c, hipEventRecord(c->ev_trace_rest[0], s));
    HIPCK(c, hipStreamWaitEvent(c->side3, c->ev_trace_rest[0], 0));
    const TraceLevel1 L1 = {reinterpret_cast<const uint8_t*>(c->d_tl) + TL_OFF_LT, c->d_lr, c->d_nodes_t, c->d_nodes_r, c->d_pf, TL_STRIDE};
    rc = launch_trace_rest((uint32_t)kind, n, n_proofs, d_targets, d_trusteds, L1, d_trace_out, sections, c->side3, lane0, lane_count);
    if (rc) return fail(c, TMX_ERR_HIP, std::string("k_trace launch: ") + hipGetErrorString((hipError_t)rc));
    HIPCK(c, hipEventRecord(c->ev_trace_rest[1], c->side3));
  }
  if (sections & TMX_TRACE_LADDERS) {
    // The chain (pass 1: 1024 latency-bound waves per 256 proofs) in segments on the caller's stream; the affine rows of a segment (pass 2)
    // follow on the side stream while the next segment is being doubled.  Measured per 256-proof batch at N = 128 (round 4 kernels): two
    // segments of 128 rows 4.95 ms, four 5.4, eight 6.7; both passes one after the other on one stream 6.4.  TMX_TRACE_SEGS overrides.
    static const uint32_t segs_env = std::getenv("TMX_TRACE_SEGS") ? (uint32_t)std::atoi(std::getenv("TMX_TRACE_SEGS")) : 2u;
    const uint32_t segs = (segs_env == 1 || segs_env == 4 || segs_env == 8 || segs_env == 16) ? segs_env : 2u;
    for (auto& e : c->ev_trace)
      if (!e) HIPCK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (uint32_t g = 0; g < segs && !rc; g++) {
      const uint32_t r0 = TR_LADDER_ROWS * g / segs, r1 = TR_LADDER_ROWS * (g + 1) / segs;
      rc = launch_trace_ladder_pass1(n, n_proofs, d_targets, edr, TL_STRIDE, c->d_trace_tmp, r0, r1, s, lane0, lane_count);
      if (rc) break;
      hipStream_t p2 = c->side;
      {
        HIPCK(c, hipEventRecord(c->ev_trace[g], s));
        HIPCK(c, hipStreamWaitEvent(p2, c->ev_trace[g], 0));
      }
      rc = launch_trace_ladder_pass2((uint32_t)kind, n, n_proofs, d_targets, edr, TL_STRIDE, c->d_trace_tmp, d_trace_out, r0, r1, p2, lane0, lane_count);
    }
    if (!rc) {
      HIPCK(c, hipEventRecord(c->ev_trace[16], c->side));
    }
  }
  if (!rc && rest_aside) HIPCK(c, hipStreamWaitEvent(s, c->ev_trace_rest[1], 0));
  if (!rc && (sections & ~(uint32_t)TMX_TRACE_LADDERS) && !rest_aside) {
    const TraceLevel1 L1 = {reinterpret_cast<const uint8_t*>(c->d_tl) + TL_OFF_LT, c->d_lr, c->d_nodes_t, c->d_nodes_r, c->d_pf, TL_STRIDE};
    rc = launch_trace_rest((uint32_t)kind, n, n_proofs, d_targets, d_trusteds, L1, d_trace_out, sections, s, lane0, lane_count);
  }
  if (rc) return fail(c, TMX_ERR_HIP, std::string("k_trace launch: ") + hipGetErrorString((hipError_t)rc));
  if ((sections & TMX_TRACE_LADDERS) && c->ev_trace[16]) HIPCK(c, hipStreamWaitEvent(s, c->ev_trace[16], 0));  // the call ends on the caller's stream
  return TMX_OK;
}
extern "C" {
int32_t tmx_trace_rows_device(tmx_ctx* c, int32_t kind, uint32_t n_proofs, const void* d_targets, const void* d_trusteds, void* d_trace_out,
                              uint32_t sections, void* hip_stream) {
  return trace_rows_impl(c, kind, n_proofs, d_targets, d_trusteds, d_trace_out, sections, hip_stream, 0, 0xffffffffu);
}

int32_t tmx_kernel_ms_mean(tmx_ctx* c, uint32_t last_k, float ms[TMX_N_KERNELS]) {
  if (!c || !ms || last_k == 0) return TMX_ERR_BAD_ARG;
  if (c->n_calls == 0) return fail(c, TMX_ERR_BAD_ARG, "no batch has been enqueued yet");
  if (last_k > tmx_ctx::EV_RING) last_k = tmx_ctx::EV_RING;
  if (last_k > c->n_calls) last_k = (uint32_t)c->n_calls;
  double acc[TMX_N_KERNELS] = {0};
  for (uint32_t j = 0; j < last_k; j++) {
    hipEvent_t* ev = c->ev[(c->n_calls - 1 - j) % tmx_ctx::EV_RING];
    HIPCK(c, hipEventSynchronize(ev[3]));
    if (c->slot_tiny[(c->n_calls - 1 - j) % tmx_ctx::EV_RING]) {
      // a small launch: ev[0] .. ev[1] = k_tiny (the EdDSA lanes and the proof roles side by side: attributed ONCE, to the EdDSA slot),
      // ev[1] .. ev[2] = the gap between the two launches (the serializer slot), ev[2] .. ev[3] = k_tiny_tail (the verdict slot) -- the four
      // figures are disjoint intervals, so their sum is the small launch's length as it is for a classic one
      float t = 0;
      HIPCK(c, hipEventElapsedTime(&t, ev[0], ev[1])); acc[TMX_K_EDDSA] += t;
      HIPCK(c, hipEventElapsedTime(&t, ev[2], ev[3])); acc[TMX_K_VERDICT] += t;
      HIPCK(c, hipEventElapsedTime(&t, ev[1], ev[2])); acc[TMX_K_SERIALIZE] += t;
      continue;
    }
    hipEvent_t* evs = c->ev_side[(c->n_calls - 1 - j) % tmx_ctx::EV_RING];
    HIPCK(c, hipEventSynchronize(evs[1]));
    float t = 0;
    HIPCK(c, hipEventElapsedTime(&t, ev[0], ev[1])); acc[TMX_K_EDDSA] += t;
    HIPCK(c, hipEventElapsedTime(&t, evs[0], evs[1])); acc[TMX_K_PROOF] += t;
    HIPCK(c, hipEventElapsedTime(&t, evs[2], evs[3])); acc[TMX_K_VERDICT] += t;
    HIPCK(c, hipEventElapsedTime(&t, c->knobs.ser_split ? ev[1] : ev[2], ev[3])); acc[TMX_K_SERIALIZE] += t;
  }
  for (int k = 0; k < TMX_N_KERNELS; k++) ms[k] = (float)(acc[k] / last_k);
  return TMX_OK;
}
int32_t tmx_last_kernel_ms(tmx_ctx* c, float ms[TMX_N_KERNELS]) { return tmx_kernel_ms_mean(c, 1, ms); }

static int32_t ensure_staging_inputs(tmx_ctx* c) {
  int32_t hs = ensure_host_stream(c);
  if (hs) return hs;
  if (c->d_in_proofs) return TMX_OK;
  HIPCK(c, hipSetDevice(c->cfg.device));
  const size_t B = c->cfg.max_batch, lanes = B * c->cfg.n_max;
  HIPCK(c, hipMalloc(&c->d_in_proofs, B * sizeof(tmx_proof_rec)));
  HIPCK(c, hipMalloc(&c->d_in_targets, lanes * sizeof(tmx_validator_rec)));
  HIPCK(c, hipMalloc(&c->d_in_trusteds, lanes * sizeof(tmx_hashfield_rec)));
  return TMX_OK;
}
static int32_t ensure_staging(tmx_ctx* c) {  // + the row buffer of the element-producing host entry points (the typed-value calls never need it)
  int32_t st = ensure_staging_inputs(c);
  if (st) return st;
  if (c->d_out) return TMX_OK;
  const size_t B = c->cfg.max_batch;
  uint64_t stride = tmx_elem_stride(TMX_KIND_SKIP, c->cfg.n_max), st2 = tmx_elem_stride(TMX_KIND_STEP, c->cfg.n_max);
  c->d_out_elems = (stride > st2 ? stride : st2) * B;
  HIPCK(c, hipMalloc(&c->d_out, c->d_out_elems * 8));
  return TMX_OK;
}

// Host-buffer path shared by tmx_witness_batch (full rows of tmx_elem_stride u64) and tmx_witness_batch_opts (dense rows of the selected
// sections, u64 or u32): H2D of the records, the batch on the context's stream, D2H of what the caller asked for.
static int32_t witness_batch_host(tmx_ctx* c, int32_t kind, uint32_t n_proofs, const tmx_proof_rec* proofs, const tmx_validator_rec* targets,
                                  const tmx_hashfield_rec* trusteds, bool opts, uint32_t sections, uint32_t format, void* out, uint64_t cap_bytes,
                                  tmx_report* reports) {
  if (!c || !proofs || !targets || (kind != TMX_KIND_SKIP && kind != TMX_KIND_STEP)) return TMX_ERR_BAD_ARG;
  if (kind == TMX_KIND_SKIP && !trusteds) return fail(c, TMX_ERR_BAD_ARG, "skip needs the trusted hash fields");
  if (opts && ((sections & ~(uint32_t)TMX_SEC_ALL) || sections == 0 || (format != TMX_OUT_U64 && format != TMX_OUT_U32)))
    return fail(c, TMX_ERR_BAD_ARG, "sections must be a non-empty subset of TMX_SEC_ALL and format TMX_OUT_U64 or TMX_OUT_U32");
  if (n_proofs == 0) return TMX_OK;
  if (n_proofs > c->cfg.max_batch) return fail(c, TMX_ERR_CAPACITY, "n_proofs exceeds the context's max_batch");
  const uint32_t n = c->cfg.n_max;
  const uint64_t stride = tmx_elem_stride(kind, n), count = tmx_elem_count(kind, n), hint = tmx_hint_elem_count(kind, n);
  const uint64_t first = opts && sections == TMX_SEC_DERIVED ? hint : 0;
  const uint64_t row_elems = opts ? tmx_out_row_elems(kind, n, sections) : count;
  const uint64_t esz = opts && format == TMX_OUT_U32 ? 4 : 8;
  if (out && opts && cap_bytes < (uint64_t)n_proofs * row_elems * esz) return fail(c, TMX_ERR_CAPACITY, "out buffer too small");
  if (out && !opts && cap_bytes < ((uint64_t)(n_proofs - 1) * stride + count) * 8) return fail(c, TMX_ERR_CAPACITY, "out_elems too small");
  for (uint32_t p = 0; p < n_proofs; p++)  // reference input/mod.rs:439-444, 338-342
    if (proofs[p].nb_a > n || (kind == TMX_KIND_SKIP && proofs[p].nb_b > n)) return fail(c, TMX_ERR_SET_TOO_LARGE, "validator set larger than VALIDATOR_SET_SIZE_MAX");
  int32_t st = ensure_staging(c);
  if (st) return st;
  const size_t lanes = (size_t)n_proofs * n;
  HIPCK(c, hipMemcpyAsync(c->d_in_proofs, proofs, (size_t)n_proofs * sizeof(tmx_proof_rec), hipMemcpyHostToDevice, c->stream));
  HIPCK(c, hipMemcpyAsync(c->d_in_targets, targets, lanes * sizeof(tmx_validator_rec), hipMemcpyHostToDevice, c->stream));
  if (kind == TMX_KIND_SKIP)
    HIPCK(c, hipMemcpyAsync(c->d_in_trusteds, trusteds, lanes * sizeof(tmx_hashfield_rec), hipMemcpyHostToDevice, c->stream));
  c->sections = opts ? sections : (uint32_t)TMX_SEC_ALL;
  st = tmx_witness_batch_device(c, kind, n_proofs, c->d_in_proofs, c->d_in_targets, kind == TMX_KIND_SKIP ? c->d_in_trusteds : nullptr,
                                out ? c->d_out : nullptr, c->d_reports, c->stream);
  c->sections = TMX_SEC_ALL;
  if (st) return st;
  if (out && !opts) {
    // rows are stride apart on the device; the last row is copied without its pad element
    const size_t bytes = ((size_t)(n_proofs - 1) * stride + count) * 8;
    HIPCK(c, hipMemcpyAsync(out, c->d_out, bytes, hipMemcpyDeviceToHost, c->stream));
  } else if (out) {
    const size_t bytes = (size_t)n_proofs * row_elems * esz;
    const void* src = c->d_out;
    if (!(first == 0 && row_elems == stride && esz == 8)) {  // dense rows of the selection, u64 or narrowed to u32, then ONE contiguous copy
      if (c->d_pack_bytes < bytes) {
        if (c->d_pack) { HIPCK(c, hipStreamSynchronize(c->stream)); HIPCK(c, hipFree(c->d_pack)); c->d_pack = nullptr; c->d_pack_bytes = 0; }
        const size_t want = (size_t)c->cfg.max_batch * (size_t)(tmx_elem_count(TMX_KIND_SKIP, n) > tmx_elem_count(TMX_KIND_STEP, n) ? tmx_elem_count(TMX_KIND_SKIP, n) : tmx_elem_count(TMX_KIND_STEP, n)) * 8;
        HIPCK(c, hipMalloc(&c->d_pack, want));
        c->d_pack_bytes = want;
      }
      int rc = launch_pack_rows(c->d_out, c->d_pack, (uint32_t)stride, (uint32_t)first, (uint32_t)row_elems, n_proofs, esz == 4, c->stream);
      if (rc) return fail(c, TMX_ERR_HIP, std::string("k_pack_rows launch: ") + hipGetErrorString((hipError_t)rc));
      src = c->d_pack;
    }
    HIPCK(c, hipMemcpyAsync(out, src, bytes, hipMemcpyDeviceToHost, c->stream));
  }
  if (reports) HIPCK(c, hipMemcpyAsync(reports, c->d_reports, (size_t)n_proofs * sizeof(tmx_report), hipMemcpyDeviceToHost, c->stream));
  HIPCK(c, hipStreamSynchronize(c->stream));
  return TMX_OK;
}

int32_t tmx_witness_batch(tmx_ctx* c, int32_t kind, uint32_t n_proofs, const tmx_proof_rec* proofs, const tmx_validator_rec* targets,
                          const tmx_hashfield_rec* trusteds, uint64_t* out_elems, uint64_t cap_elems, tmx_report* reports) {
  return witness_batch_host(c, kind, n_proofs, proofs, targets, trusteds, false, TMX_SEC_ALL, TMX_OUT_U64, out_elems, cap_elems * 8, reports);
}

int32_t tmx_witness_batch_opts(tmx_ctx* c, int32_t kind, uint32_t n_proofs, const tmx_proof_rec* proofs, const tmx_validator_rec* targets,
                               const tmx_hashfield_rec* trusteds, uint32_t sections, uint32_t format, void* out, uint64_t cap_bytes,
                               tmx_report* reports) {
  return witness_batch_host(c, kind, n_proofs, proofs, targets, trusteds, true, sections, format, out, cap_bytes, reports);
}

uint64_t tmx_out_row_elems(int32_t kind, uint32_t n, uint32_t sections) {
  const uint64_t count = tmx_elem_count(kind, n), hint = tmx_hint_elem_count(kind, n);
  if (count == 0) return 0;
  return (sections & TMX_SEC_ALL) == TMX_SEC_ALL ? count : (sections & TMX_SEC_HINT) ? hint : (sections & TMX_SEC_DERIVED) ? count - hint : 0;
}

int32_t tmx_witness_batch_device_sections(tmx_ctx* c, int32_t kind, uint32_t n_proofs, const void* d_proofs, const void* d_targets,
                                          const void* d_trusteds, void* d_out_elems, void* d_reports, void* hip_stream, uint32_t sections) {
  if (!c || (sections & ~(uint32_t)TMX_SEC_ALL) || sections == 0) return TMX_ERR_BAD_ARG;
  c->sections = sections;
  const int32_t st = tmx_witness_batch_device(c, kind, n_proofs, d_proofs, d_targets, d_trusteds, d_out_elems, d_reports, hip_stream);
  c->sections = TMX_SEC_ALL;
  return st;
}

int32_t tmx_skip_witness(tmx_ctx* c, const tmx_proof_rec* proof, const tmx_validator_rec* target, const tmx_hashfield_rec* trusted,
                         uint64_t* out_elems, uint64_t cap_elems, tmx_report* report) {
  return tmx_witness_batch(c, TMX_KIND_SKIP, 1, proof, target, trusted, out_elems, cap_elems, report);
}
int32_t tmx_step_witness(tmx_ctx* c, const tmx_proof_rec* proof, const tmx_validator_rec* target, uint64_t* out_elems, uint64_t cap_elems,
                         tmx_report* report) {
  return tmx_witness_batch(c, TMX_KIND_STEP, 1, proof, target, nullptr, out_elems, cap_elems, report);
}

}  // extern "C"

// ---- the typed value of the hint (include/tmx.h "TYPED VALUE"; value.h / value.hip) ----------------------------------------------------
static ValueLayout value_layout_for(int32_t kind, uint32_t n, uint32_t sections) { return value_layout((uint32_t)kind, n, tree_nodes(n), sections); }

// fixed-part gather table of one kind, straight from the public structs' offsetof: whatever include/tmx.h declares is what the device writes
static std::vector<uint16_t> value_fixed_lut(int32_t kind) {
  std::vector<uint16_t> lut(VAL_FIXED_MAX, VAL_LUT_ZERO);
  auto put = [&](size_t dst, size_t len, uint32_t src, uint32_t off) {
    for (size_t i = 0; i < len; i++) lut[dst + i] = (uint16_t)((src << 12) | (off + (uint32_t)i));
  };
  auto chain_id_proof = [&](size_t at) {
    put(at + offsetof(tmx_chain_id_proof_value, proof), 128, VSRC_PF, PF_OFF_AUNTS + 128 * 0);
    put(at + offsetof(tmx_chain_id_proof_value, enc_chain_id_byte_length), 4, VSRC_PF, PF_OFF_CIDLEN);
    put(at + offsetof(tmx_chain_id_proof_value, chain_id), 52, VSRC_PF, PF_OFF_CID52);
  };
  auto height_proof = [&](size_t at) {
    put(at + offsetof(tmx_height_proof_value, proof), 128, VSRC_PF, PF_OFF_AUNTS + 128 * 1);
    put(at + offsetof(tmx_height_proof_value, enc_height_byte_length), 4, VSRC_PF, PF_OFF_HLEN);
    put(at + offsetof(tmx_height_proof_value, height), 8, VSRC_PF, PF_OFF_HEIGHT);
  };
  auto hash_proof = [&](size_t at, int q, uint32_t leaf_off) {
    put(at + offsetof(tmx_hash_inclusion_proof_value, proof), 128, VSRC_PF, PF_OFF_AUNTS + 128 * (uint32_t)q);
    put(at + offsetof(tmx_hash_inclusion_proof_value, leaf), 34, VSRC_PF, leaf_off);
  };
  if (kind == TMX_KIND_SKIP) {
    put(offsetof(tmx_skip_inputs_fixed, target_header), 32, VSRC_PF, PF_OFF_HEADER);
    put(offsetof(tmx_skip_inputs_fixed, trusted_header), 32, VSRC_PROOF, PR_OFF_HASH);
    put(offsetof(tmx_skip_inputs_fixed, round), 8, VSRC_PROOF, PR_OFF_ROUND);
    put(offsetof(tmx_skip_inputs_fixed, nb_target_validators), 4, VSRC_PROOF, PR_OFF_NB_A);
    put(offsetof(tmx_skip_inputs_fixed, nb_trusted_validators), 4, VSRC_PROOF, PR_OFF_NB_B);
    chain_id_proof(offsetof(tmx_skip_inputs_fixed, target_block_chain_id_proof));
    height_proof(offsetof(tmx_skip_inputs_fixed, target_block_height_proof));
    hash_proof(offsetof(tmx_skip_inputs_fixed, target_block_validators_hash_proof), 2, PF_OFF_LEAFV);
    hash_proof(offsetof(tmx_skip_inputs_fixed, trusted_block_validators_hash_proof), 3, PF_OFF_LEAFX);
    put(offsetof(tmx_skip_inputs_fixed, report), sizeof(tmx_report), VSRC_REPORT, 0);
  } else {
    put(offsetof(tmx_step_inputs_fixed, next_header), 32, VSRC_PF, PF_OFF_HEADER);
    put(offsetof(tmx_step_inputs_fixed, round), 8, VSRC_PROOF, PR_OFF_ROUND);
    put(offsetof(tmx_step_inputs_fixed, nb_validators), 4, VSRC_PROOF, PR_OFF_NB_A);
    chain_id_proof(offsetof(tmx_step_inputs_fixed, next_block_chain_id_proof));
    height_proof(offsetof(tmx_step_inputs_fixed, next_block_height_proof));
    hash_proof(offsetof(tmx_step_inputs_fixed, next_block_validators_hash_proof), 2, PF_OFF_LEAFV);
    const size_t lb = offsetof(tmx_step_inputs_fixed, next_block_last_block_id_proof);
    put(lb + offsetof(tmx_block_id_inclusion_proof_value, proof), 128, VSRC_PF, PF_OFF_AUNTS + 128 * 3);
    put(lb + offsetof(tmx_block_id_inclusion_proof_value, leaf), 72, VSRC_PF, PF_OFF_LEAFX);
    hash_proof(offsetof(tmx_step_inputs_fixed, prev_block_next_validators_hash_proof), 4, PF_OFF_LEAFY);
    put(offsetof(tmx_step_inputs_fixed, report), sizeof(tmx_report), VSRC_REPORT, 0);
  }
  return lut;
}

extern "C" {

int32_t tmx_value_layout_of(int32_t kind, uint32_t n, uint32_t sections, tmx_value_layout* out) {
  if (!out || (kind != TMX_KIND_SKIP && kind != TMX_KIND_STEP) || n == 0 || n > TMX_N_MAX_LIMIT || (sections != TMX_SEC_HINT && sections != TMX_SEC_ALL))
    return TMX_ERR_BAD_ARG;
  const ValueLayout L = value_layout_for(kind, n, sections);
  std::memset(out, 0, sizeof *out);
  auto at = [&](uint32_t p) { return L.off[p + 1] > L.off[p] ? L.off[p] : 0u; };
  out->bytes = L.off[VP_COUNT];
  out->fixed_bytes = L.off[VP_FIXED + 1] - L.off[VP_FIXED];
  out->off_validators = at(VP_VALIDATORS); out->off_hashfields = at(VP_HASHFIELDS);
  out->off_target_lanes = at(VP_LANE_T); out->off_trusted_lanes = at(VP_LANE_R);
  out->off_nodes_target = at(VP_NODES_T); out->off_nodes_trusted = at(VP_NODES_R); out->off_proof_derived = at(VP_PROOF_D);
  out->tree_nodes = L.tree_nodes;
  return TMX_OK;
}

int32_t tmx_inputs_value_batch_device(tmx_ctx* c, int32_t kind, uint32_t n_proofs, const void* d_proofs, const void* d_targets, const void* d_trusteds,
                                      uint32_t sections, void* d_out, void* hip_stream) {
  int32_t st = check_batch_args(c, kind, n_proofs, d_proofs, d_targets, d_trusteds);
  if (st) return st;
  if (!d_out || (sections != TMX_SEC_HINT && sections != TMX_SEC_ALL)) return fail(c, TMX_ERR_BAD_ARG, "d_out must be set and sections TMX_SEC_HINT or TMX_SEC_ALL");
  if (n_proofs == 0) return TMX_OK;
  hipStream_t s = reinterpret_cast<hipStream_t>(hip_stream);
  if (!c->d_val_lut[kind]) {  // once per kind (blocking: the table is a local)
    const std::vector<uint16_t> lut = value_fixed_lut(kind);
    HIPCK(c, hipSetDevice(c->cfg.device));
    HIPCK(c, hipMalloc(&c->d_val_lut[kind], lut.size() * 2));
    HIPCK(c, hipMemcpy(c->d_val_lut[kind], lut.data(), lut.size() * 2, hipMemcpyHostToDevice));
  }
  // the Level-1 kernels without a single element row: no serializer launch, only the lane / proof records and the reports
  st = tmx_witness_batch_device(c, kind, n_proofs, d_proofs, d_targets, d_trusteds, nullptr, c->d_reports, hip_stream);
  if (st) return st;
  const ValueLayout L = value_layout_for(kind, c->cfg.n_max, sections);
  ValueSources V;
  V.proofs = (const uint8_t*)d_proofs; V.targets = (const uint8_t*)d_targets; V.trusteds = (const uint8_t*)d_trusteds;
  V.tl = (const uint8_t*)c->d_tl; V.lr = (const uint8_t*)c->d_lr; V.pf = (const uint8_t*)c->d_pf;
  V.nodes_t = (const uint8_t*)c->d_nodes_t; V.nodes_r = (const uint8_t*)c->d_nodes_r; V.reports = (const uint8_t*)c->d_reports;
  int rc = launch_pack_value(L, V, c->d_val_lut[kind], n_proofs, d_out, s);
  if (rc) return fail(c, TMX_ERR_HIP, std::string("k_pack_value launch: ") + hipGetErrorString((hipError_t)rc));
  // the batch now ends with this launch: a later call on another stream must start behind it, not behind the Level-1 kernels' end
  HIPCK(c, hipEventRecord(c->ev_value, s));
  c->ev_done = c->ev_value;
  return TMX_OK;
}

// the device address of `p` if it is mapped page-locked host memory (hipHostMalloc / hipHostRegister with the mapped flag), else null
static void* mapped_device_ptr(const void* p) {
  hipPointerAttribute_t a;
  std::memset(&a, 0, sizeof a);
  if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  if (a.type != hipMemoryTypeHost || !a.devicePointer) return nullptr;
  return a.devicePointer;
}

int32_t tmx_inputs_value_batch(tmx_ctx* c, int32_t kind, uint32_t n_proofs, const tmx_proof_rec* proofs, const tmx_validator_rec* targets,
                               const tmx_hashfield_rec* trusteds, uint32_t sections, void* out, uint64_t cap_bytes) {
  if (!c || !proofs || !targets || !out || (kind != TMX_KIND_SKIP && kind != TMX_KIND_STEP)) return TMX_ERR_BAD_ARG;
  if (kind == TMX_KIND_SKIP && !trusteds) return fail(c, TMX_ERR_BAD_ARG, "skip needs the trusted hash fields");
  if (sections != TMX_SEC_HINT && sections != TMX_SEC_ALL) return fail(c, TMX_ERR_BAD_ARG, "sections must be TMX_SEC_HINT or TMX_SEC_ALL");
  if (n_proofs == 0) return TMX_OK;
  if (n_proofs > c->cfg.max_batch) return fail(c, TMX_ERR_CAPACITY, "n_proofs exceeds the context's max_batch");
  const uint32_t n = c->cfg.n_max;
  const ValueLayout L = value_layout_for(kind, n, sections);
  const size_t bytes = (size_t)n_proofs * L.off[VP_COUNT];
  if (cap_bytes < bytes) return fail(c, TMX_ERR_CAPACITY, "out buffer too small");
  for (uint32_t p = 0; p < n_proofs; p++)  // reference input/mod.rs:439-444, 338-342
    if (proofs[p].nb_a > n || (kind == TMX_KIND_SKIP && proofs[p].nb_b > n)) return fail(c, TMX_ERR_SET_TOO_LARGE, "validator set larger than VALIDATOR_SET_SIZE_MAX");
  int32_t st = ensure_staging_inputs(c);
  if (st) return st;
  const size_t lanes = (size_t)n_proofs * n;
  HIPCK(c, hipMemcpyAsync(c->d_in_proofs, proofs, (size_t)n_proofs * sizeof(tmx_proof_rec), hipMemcpyHostToDevice, c->stream));
  HIPCK(c, hipMemcpyAsync(c->d_in_targets, targets, lanes * sizeof(tmx_validator_rec), hipMemcpyHostToDevice, c->stream));
  if (kind == TMX_KIND_SKIP)
    HIPCK(c, hipMemcpyAsync(c->d_in_trusteds, trusteds, lanes * sizeof(tmx_hashfield_rec), hipMemcpyHostToDevice, c->stream));
  // A page-locked `out` is written by k_pack_value itself (posted writes over PCIe: no device-to-host copy to enqueue and wait for) up to
  // TMX_VALUE_DIRECT_MAX bytes; larger transfers, and pageable buffers, go through device staging and one copy.
  static const size_t direct_max = std::getenv("TMX_VALUE_DIRECT_MAX") ? (size_t)std::atoll(std::getenv("TMX_VALUE_DIRECT_MAX")) : ((size_t)1 << 20);
  void* direct = bytes <= direct_max ? mapped_device_ptr(out) : nullptr;
  void* dst = direct;
  if (!dst) {
    if (c->d_value_bytes < bytes) {
      if (c->d_value) { HIPCK(c, hipStreamSynchronize(c->stream)); HIPCK(c, hipFree(c->d_value)); c->d_value = nullptr; c->d_value_bytes = 0; }
      const size_t per = std::max<size_t>(value_layout_for(TMX_KIND_STEP, n, TMX_SEC_ALL).off[VP_COUNT], value_layout_for(TMX_KIND_SKIP, n, TMX_SEC_ALL).off[VP_COUNT]);
      const size_t want = (size_t)c->cfg.max_batch * per;
      HIPCK(c, hipMalloc(&c->d_value, want));
      c->d_value_bytes = want;
    }
    dst = c->d_value;
  }
  st = tmx_inputs_value_batch_device(c, kind, n_proofs, c->d_in_proofs, c->d_in_targets, kind == TMX_KIND_SKIP ? c->d_in_trusteds : nullptr, sections, dst,
                                     c->stream);
  if (st) return st;
  if (!direct) HIPCK(c, hipMemcpyAsync(out, c->d_value, bytes, hipMemcpyDeviceToHost, c->stream));
  HIPCK(c, hipStreamSynchronize(c->stream));
  return TMX_OK;
}

int32_t tmx_skip_inputs_value(tmx_ctx* c, const tmx_proof_rec* proof, const tmx_validator_rec* target, const tmx_hashfield_rec* trusted, uint32_t sections,
                              void* out, uint64_t cap_bytes) {
  return tmx_inputs_value_batch(c, TMX_KIND_SKIP, 1, proof, target, trusted, sections, out, cap_bytes);
}
int32_t tmx_step_inputs_value(tmx_ctx* c, const tmx_proof_rec* proof, const tmx_validator_rec* target, uint32_t sections, void* out, uint64_t cap_bytes) {
  return tmx_inputs_value_batch(c, TMX_KIND_STEP, 1, proof, target, nullptr, sections, out, cap_bytes);
}

void* tmx_host_alloc(tmx_ctx* c, uint64_t bytes) {
  if (!c || bytes == 0) return nullptr;
  if (hipSetDevice(c->cfg.device) != hipSuccess) return nullptr;
  void* p = nullptr;
  if (hipHostMalloc(&p, bytes, hipHostMallocMapped | hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  return p;
}
void tmx_host_free(tmx_ctx* c, void* p) {
  (void)c;
  if (p) (void)hipHostFree(p);
}

}  // extern "C"

extern "C" {

int32_t tmx_eddsa_lanes(tmx_ctx* c, uint32_t n_lanes, const tmx_validator_rec* lanes, uint8_t* out) {
  if (!c || !lanes || !out) return TMX_ERR_BAD_ARG;
  if (n_lanes == 0) return TMX_OK;
  if ((uint64_t)n_lanes > (uint64_t)c->cfg.max_batch * c->cfg.n_max) return fail(c, TMX_ERR_CAPACITY, "n_lanes exceeds max_batch * n_max");
  int32_t st = ensure_staging_inputs(c);
  if (st) return st;
  HIPCK(c, hipMemcpyAsync(c->d_in_targets, lanes, (size_t)n_lanes * sizeof(tmx_validator_rec), hipMemcpyHostToDevice, c->stream));
  if (use_tiny(c, n_lanes)) {
    st = run_tiny_lanes(c, n_lanes, c->d_in_targets, c->d_ed, ED_STRIDE, c->stream);
    if (st) return st;
  } else {
    int rc = run_eddsa(c, n_lanes, c->d_in_targets, c->d_ed, ED_STRIDE, c->stream);
    if (rc) return fail(c, TMX_ERR_HIP, std::string("k_eddsa launch: ") + hipGetErrorString((hipError_t)rc));
  }
  HIPCK(c, hipMemcpyAsync(out, c->d_ed, (size_t)n_lanes * ED_STRIDE, hipMemcpyDeviceToHost, c->stream));
  HIPCK(c, hipStreamSynchronize(c->stream));
  return TMX_OK;
}

int32_t tmx_valid_skip_batch(tmx_ctx* c, uint32_t n_cand, const tmx_addr_rec* start, uint32_t n_start, const tmx_addr_rec* targets,
                             const uint32_t* n_targets, const tmx_addr_rec* sigs, const uint32_t* n_sigs, uint8_t* valid, uint64_t* shared_power,
                             uint64_t* total_power) {
  if (!c || !start || !targets || !n_targets || !sigs || !n_sigs || !valid || !shared_power || !total_power) return TMX_ERR_BAD_ARG;
  if (n_cand == 0) return TMX_OK;
  const uint32_t n = c->cfg.n_max;
  if (n_start > n) return fail(c, TMX_ERR_SET_TOO_LARGE, "start validator set larger than VALIDATOR_SET_SIZE_MAX");
  static_assert(sizeof(tmx_addr_rec) == 32, "address record layout");
  int32_t hs = ensure_host_stream(c);
  if (hs) return hs;
  const size_t per = (size_t)n * 32;
  // small, latency-only call: temporary device buffers on the context's stream
  void *d_start = nullptr, *d_t = nullptr, *d_s = nullptr, *d_nt = nullptr, *d_ns = nullptr, *d_v = nullptr, *d_sh = nullptr, *d_to = nullptr;
  auto cleanup = [&]() { for (void* p : {d_start, d_t, d_s, d_nt, d_ns, d_v, d_sh, d_to}) if (p) (void)hipFree(p); };
  hipError_t e = hipSuccess;
  auto ck = [&](hipError_t r) { if (e == hipSuccess && r != hipSuccess) e = r; };
  ck(hipMalloc(&d_start, per)); ck(hipMalloc(&d_t, per * n_cand)); ck(hipMalloc(&d_s, per * n_cand)); ck(hipMalloc(&d_nt, 4 * (size_t)n_cand));
  ck(hipMalloc(&d_ns, 4 * (size_t)n_cand)); ck(hipMalloc(&d_v, n_cand)); ck(hipMalloc(&d_sh, 8 * (size_t)n_cand)); ck(hipMalloc(&d_to, 8 * (size_t)n_cand));
  if (e == hipSuccess) {
    ck(hipMemcpyAsync(d_start, start, (size_t)n_start * 32, hipMemcpyHostToDevice, c->stream));
    ck(hipMemcpyAsync(d_t, targets, per * n_cand, hipMemcpyHostToDevice, c->stream));
    ck(hipMemcpyAsync(d_s, sigs, per * n_cand, hipMemcpyHostToDevice, c->stream));
    ck(hipMemcpyAsync(d_nt, n_targets, 4 * (size_t)n_cand, hipMemcpyHostToDevice, c->stream));
    ck(hipMemcpyAsync(d_ns, n_sigs, 4 * (size_t)n_cand, hipMemcpyHostToDevice, c->stream));
  }
  if (e == hipSuccess) ck((hipError_t)launch_valid_skip(n_cand, n, d_start, n_start, d_t, d_nt, d_s, d_ns, d_v, d_sh, d_to, c->stream));
  if (e == hipSuccess) {
    ck(hipMemcpyAsync(valid, d_v, n_cand, hipMemcpyDeviceToHost, c->stream));
    ck(hipMemcpyAsync(shared_power, d_sh, 8 * (size_t)n_cand, hipMemcpyDeviceToHost, c->stream));
    ck(hipMemcpyAsync(total_power, d_to, 8 * (size_t)n_cand, hipMemcpyDeviceToHost, c->stream));
    ck(hipStreamSynchronize(c->stream));
  }
  cleanup();
  if (e != hipSuccess) return fail(c, TMX_ERR_HIP, std::string("tmx_valid_skip_batch: ") + hipGetErrorString(e));
  return TMX_OK;
}

// ---- public I/O packing (big-endian, abi.encodePacked): TendermintX.sol:104-108, skip.rs:120-122, step.rs:107-108
static void be64(uint64_t v, uint8_t* o) { for (int i = 0; i < 8; i++) o[i] = (uint8_t)(v >> (56 - 8 * i)); }
static uint64_t rd_be64(const uint8_t* p) { uint64_t v = 0; for (int i = 0; i < 8; i++) v = (v << 8) | p[i]; return v; }
void tmx_pack_skip_input(uint64_t trusted_block, const uint8_t h[32], uint64_t target_block, uint8_t out[48]) {
  be64(trusted_block, out); std::memcpy(out + 8, h, 32); be64(target_block, out + 40);
}
void tmx_unpack_skip_input(const uint8_t in[48], uint64_t* trusted_block, uint8_t h[32], uint64_t* target_block) {
  *trusted_block = rd_be64(in); std::memcpy(h, in + 8, 32); *target_block = rd_be64(in + 40);
}
void tmx_pack_step_input(uint64_t prev_block, const uint8_t h[32], uint8_t out[40]) { be64(prev_block, out); std::memcpy(out + 8, h, 32); }
void tmx_unpack_step_input(const uint8_t in[40], uint64_t* prev_block, uint8_t h[32]) { *prev_block = rd_be64(in); std::memcpy(h, in + 8, 32); }


// ---- Goldilocks NTT / coset LDE ------------------------------------------------------------------------------------------------
static int32_t ntt_table(tmx_ctx* c, uint32_t log_n, hipStream_t s, void** w) {
  if (!c->d_ntt_w[log_n]) {
    const size_t half = log_n ? ((size_t)1 << (log_n - 1)) : 1;
    HIPCK(c, hipSetDevice(c->cfg.device));
    HIPCK(c, hipMalloc(&c->d_ntt_w[log_n], half * 8));
    int rc = launch_ntt_table(c->d_ntt_w[log_n], log_n, c->ntt_root, s);
    if (rc) return fail(c, TMX_ERR_HIP, std::string("k_ntt_table launch: ") + hipGetErrorString((hipError_t)rc));
    HIPCK(c, hipStreamSynchronize(s));  // once per size: later calls may come on another stream
  }
  *w = c->d_ntt_w[log_n];
  return TMX_OK;
}
// the twiddles between the two passes of a four-step transform (log_n > 11), as a matrix in the layout pass A writes: N x 8 bytes per
// size and direction, built once (the passes then read them with the coalescing of their own stores instead of gathering 8-byte words)
static int32_t ntt_matrix(tmx_ctx* c, uint32_t log_n, uint32_t log_n2, bool inverse, const void* w, hipStream_t s, void** m) {
  void*& slot = c->d_ntt_m[inverse ? 1 : 0][log_n];
  if (!slot) {
    HIPCK(c, hipSetDevice(c->cfg.device));
    HIPCK(c, hipMalloc(&slot, (size_t)8 << log_n));
    int rc = launch_ntt_matrix(slot, w, log_n, log_n2, inverse, s);
    if (rc) return fail(c, TMX_ERR_HIP, std::string("k_ntt_matrix launch: ") + hipGetErrorString((hipError_t)rc));
    HIPCK(c, hipStreamSynchronize(s));
  }
  *m = slot;
  return TMX_OK;
}
static int32_t ntt_scratch(tmx_ctx* c, size_t bytes, hipStream_t s) {
  if (c->ntt_tmp_bytes >= bytes) return TMX_OK;
  HIPCK(c, hipSetDevice(c->cfg.device));
  if (c->d_ntt_tmp) {
    HIPCK(c, hipStreamSynchronize(s));  // earlier transforms on this stream may still read it
    HIPCK(c, hipFree(c->d_ntt_tmp));
    c->d_ntt_tmp = nullptr; c->ntt_tmp_bytes = 0;
  }
  HIPCK(c, hipMalloc(&c->d_ntt_tmp, bytes));
  c->ntt_tmp_bytes = bytes;
  return TMX_OK;
}
static uint64_t gl_pow_host(uint64_t b, uint64_t e) {
  const unsigned __int128 p = 0xffffffff00000001ull;
  unsigned __int128 r = 1, x = b % p;
  while (e) {
    if (e & 1) r = r * x % p;
    x = x * x % p;
    e >>= 1;
  }
  return (uint64_t)r;
}
// columns of 2^log_n elements, column c at element c * col_stride; `tmp` (n_cols << log_n elements) only for log_n > 11
// post_vec (or null): the outputs are multiplied by post_vec[k] INSTEAD of the inverse transform's 1 / N (a vector that carries it);
// nonzero (or 0): only the first `nonzero` inputs of every column are non-zero and read (a multiple of N2 for a two-pass transform).
static int32_t ntt_run(tmx_ctx* c, uint32_t log_n, uint32_t n_cols, const void* d_in, uint64_t in_stride, void* d_out, uint64_t out_stride,
                       void* tmp, bool inverse, hipStream_t s, const void* post_vec = nullptr, uint64_t nonzero = 0) {
  void* w = nullptr;
  int32_t st = ntt_table(c, log_n, s, &w);
  if (st) return st;
  const uint64_t n_inv = inverse ? gl_pow_host((uint64_t)1 << log_n, 0xffffffff00000001ull - 2) : 1;
  NttPass P;
  std::memset(&P, 0, sizeof P);
  P.log_n = log_n; P.inverse = inverse ? 1 : 0;
  {  // omega_16 of the domain (its inverse for an inverse transform) is an odd power k of 2^12: the kernel loads in the order kinv i
    const uint64_t w16 = gl_pow_host(c->ntt_root, (1ull << 28) * (inverse ? 15u : 1u));
    uint32_t k = 0;
    for (uint32_t t = 1; t < 16; t += 2)
      if (gl_pow_host(1ull << 12, t) == w16) k = t;
    if (!k) return fail(c, TMX_ERR_BAD_ARG, "omega_16 of the NTT domain is not a power of 2^12");
    for (uint32_t t = 1; t < 16; t += 2)
      if (((t * k) & 15u) == 1u) P.kinv = t;
  }
  int rc;
  if (log_n <= 11) {  // one pass: whole columns in LDS, T columns per tile
    P.log_l = log_n; P.log_t = 12 - log_n;
    if (((uint64_t)1 << P.log_t) > n_cols) { P.log_t = 0; while (((uint64_t)2 << P.log_t) <= n_cols) P.log_t++; }
    const uint32_t T = 1u << P.log_t;
    // the "columns" of the kernel are groups of T real columns: sub-transform t of a group = column, stride = column stride
    P.tiles_per_col = (n_cols + T - 1) / T; P.n_sub = n_cols;
    P.col_stride_in = 0; P.t_stride_in = in_stride; P.j_stride_in = 1;
    P.col_stride_out = 0; P.t_stride_out = out_stride; P.j_stride_out = 1;
    P.scale = post_vec ? 1 : n_inv; P.post_twiddle = post_vec ? 1 : 0; P.tm_t_stride = 0; P.j_nonzero = (uint32_t)nonzero;
    rc = launch_ntt_pass(P, 1, d_in, d_out, w, post_vec, s);
    if (rc) return fail(c, TMX_ERR_HIP, std::string("k_ntt_tile launch: ") + hipGetErrorString((hipError_t)rc));
    return TMX_OK;
  }
  // N = N1 N2, N1 = 2^a (pass A, strided), N2 = 2^b (pass B, contiguous).  (Round 3 re-measured the split and the tile sizes at 2^16 / 2^20 /
  // 2^22: a = ceil(log_n / 2) -1 / -2 / +1 and tiles of 2^12 / 2^13 / 2^14 elements everywhere -- the balanced split with the tile rule below wins each; round 4 re-measured pass A's tile alone at
  // 2^16 / 2^20 / 2^22: 0.161 / 4.53 / 2.11 ms with 2^12, 0.169 / 3.29 / 1.71 with 2^13, 0.191 / 3.32 / 1.24 with 2^14 -- the rule's choices.)
  const uint32_t a = (log_n + 1) / 2, b = log_n - a;
  const uint64_t N = (uint64_t)1 << log_n, N1 = (uint64_t)1 << a, N2 = (uint64_t)1 << b;
  // tiles of the two strided passes: T >= 8 sub-transforms side by side (runs of >= 64 B along the unit-stride dimension; with T = 4 at
  // N1 = 2^10, FETCH_SIZE was 4x the data), in the smallest tile that allows it (2^12 .. 2^14 elements: three, two or one workgroup per
  // CU's LDS).  Measured at 2^16 / 2^20 / 2^22: tiles of 2^12 / 2^13 / 2^14 elements are each the fastest there.
  auto tile_log_of = [&](uint32_t log_l) { return std::min(14u, std::max(12u, log_l + 3u)); };
  P.log_l = a; P.log_t = std::min(tile_log_of(a) - a, b); P.n_sub = N2; P.tiles_per_col = (uint32_t)(N2 >> P.log_t);
  P.col_stride_in = in_stride; P.t_stride_in = 1; P.j_stride_in = N2;
  P.col_stride_out = N; P.t_stride_out = 1; P.j_stride_out = N2;
  P.post_twiddle = 1; P.scale = 1; P.tm_t_stride = P.t_stride_out;
  P.j_nonzero = nonzero ? (uint32_t)(nonzero >> b) : 0u;  // input n1 N2 + n2 is zero from n1 = nonzero / N2 on
  void* m = nullptr;
  st = ntt_matrix(c, log_n, b, inverse, w, s, &m);
  if (st) return st;
  rc = launch_ntt_pass(P, n_cols, d_in, tmp, w, m, s);
  if (rc) return fail(c, TMX_ERR_HIP, std::string("k_ntt_tile launch: ") + hipGetErrorString((hipError_t)rc));
  P.log_l = b; P.log_t = std::min(tile_log_of(b) - b, a); P.n_sub = N1; P.tiles_per_col = (uint32_t)(N1 >> P.log_t);
  P.col_stride_in = N; P.t_stride_in = N2; P.j_stride_in = 1;
  P.col_stride_out = out_stride; P.t_stride_out = 1; P.j_stride_out = N1;
  P.post_twiddle = post_vec ? 1 : 0; P.scale = post_vec ? 1 : n_inv; P.tm_t_stride = P.t_stride_out; P.j_nonzero = 0;
  rc = launch_ntt_pass(P, n_cols, tmp, d_out, w, post_vec, s);
  if (rc) return fail(c, TMX_ERR_HIP, std::string("k_ntt_tile launch: ") + hipGetErrorString((hipError_t)rc));
  return TMX_OK;
}

int32_t tmx_ntt_set_domain(tmx_ctx* c, uint64_t root_2_32, uint64_t coset_shift) {
  if (!c) return TMX_ERR_BAD_ARG;
  const uint64_t P = 0xffffffff00000001ull;
  if (root_2_32 >= P || coset_shift == 0 || coset_shift >= P || gl_pow_host(root_2_32, 1ull << 31) != P - 1)
    return fail(c, TMX_ERR_BAD_ARG, "root_2_32 must be a primitive 2^32-th root of unity of the Goldilocks field, coset_shift a non-zero element");
  if (root_2_32 != c->ntt_root || coset_shift != c->ntt_shift) {  // the tables belong to the old domain
    HIPCK(c, hipSetDevice(c->cfg.device));
    HIPCK(c, hipDeviceSynchronize());
    for (auto& v : c->d_lde_s)
      if (v) { (void)hipFree(v); v = nullptr; }
  }
  if (root_2_32 != c->ntt_root) {
    for (auto& w : c->d_ntt_w)
      if (w) { (void)hipFree(w); w = nullptr; }
    for (auto& dir : c->d_ntt_m)
      for (auto& m : dir)
        if (m) { (void)hipFree(m); m = nullptr; }
  }
  c->ntt_root = root_2_32; c->ntt_shift = coset_shift;
  return TMX_OK;
}

int32_t tmx_ntt_goldilocks_device(tmx_ctx* c, uint32_t log_n, uint32_t n_cols, const uint64_t* d_in, uint64_t* d_out, int32_t inverse,
                                  void* hip_stream) {
  if (!c || !d_in || !d_out) return TMX_ERR_BAD_ARG;
  if (log_n > TMX_NTT_MAX_LOG) return fail(c, TMX_ERR_CAPACITY, "log_n exceeds TMX_NTT_MAX_LOG");
  if (n_cols == 0) return TMX_OK;
  hipStream_t s = reinterpret_cast<hipStream_t>(hip_stream);
  const uint64_t N = (uint64_t)1 << log_n;
  if (log_n > 11) {
    int32_t st = ntt_scratch(c, (size_t)n_cols * N * 8, s);
    if (st) return st;
  }
  return ntt_run(c, log_n, n_cols, d_in, N, d_out, N, c->d_ntt_tmp, inverse != 0, s);
}

int32_t tmx_lde_goldilocks_device(tmx_ctx* c, uint32_t log_n, uint32_t log_blowup, uint32_t n_cols, const uint64_t* d_in, uint64_t* d_out,
                                  void* hip_stream) {
  if (!c || !d_in || !d_out) return TMX_ERR_BAD_ARG;
  const uint32_t log_m = log_n + log_blowup;
  if (log_m > TMX_NTT_MAX_LOG) return fail(c, TMX_ERR_CAPACITY, "log_n + log_blowup exceeds TMX_NTT_MAX_LOG");
  if (n_cols == 0) return TMX_OK;
  hipStream_t s = reinterpret_cast<hipStream_t>(hip_stream);
  const uint64_t N = (uint64_t)1 << log_n, M = (uint64_t)1 << log_m;
  // scratch: [coefficients, n_cols x M] [four-step intermediate, n_cols x M]
  int32_t st = ntt_scratch(c, (size_t)n_cols * M * 8 * 2, s);
  if (st) return st;
  uint64_t* coef = reinterpret_cast<uint64_t*>(c->d_ntt_tmp);
  uint64_t* tmp = coef + (size_t)n_cols * M;
  // Fused form: the inverse transform's last pass multiplies coefficient i by S[i] = shift^i / N (one vector per column length, built
  // once), and the forward transform does not read the zero padding -- its pass A loads the first N / N2 inputs of every strided
  // sub-transform only.  (Before: k_lde_expand rewrote all M slots, a gl_pow per coefficient, and pass A read them back: 17 of the ~55 GB
  // an 8x LDE of 2^15-row columns moved.)  A split whose N2 exceeds N keeps the padded form.
  const uint32_t b_fwd = log_m > 11 ? log_m - (log_m + 1) / 2 : 0;
  const bool fused = !(std::getenv("TMX_LDE_FUSED") && std::getenv("TMX_LDE_FUSED")[0] == '0') && log_n >= b_fwd;
  if (fused) {
    if (!c->d_lde_s[log_n]) {
      HIPCK(c, hipSetDevice(c->cfg.device));
      HIPCK(c, hipMalloc(&c->d_lde_s[log_n], (size_t)8 << log_n));
      int rc = launch_lde_scale_table(c->d_lde_s[log_n], log_n, c->ntt_shift, gl_pow_host(N, 0xffffffff00000001ull - 2), s);
      if (rc) return fail(c, TMX_ERR_HIP, std::string("k_lde_scale_table launch: ") + hipGetErrorString((hipError_t)rc));
      HIPCK(c, hipStreamSynchronize(s));  // once per size: later calls may come on another stream
    }
    st = ntt_run(c, log_n, n_cols, d_in, N, coef, M, tmp, true, s, c->d_lde_s[log_n]);  // scaled coefficients into the first N slots of every M-slot column
    if (st) return st;
    return ntt_run(c, log_m, n_cols, coef, M, d_out, M, tmp, false, s, nullptr, N);
  }
  st = ntt_run(c, log_n, n_cols, d_in, N, coef, M, tmp, true, s);  // coefficients into the first N slots of every M-slot column
  if (st) return st;
  int rc = launch_lde_expand(coef, log_n, log_m, n_cols, c->ntt_shift, s);  // c_i shift^i, zero padding
  if (rc) return fail(c, TMX_ERR_HIP, std::string("k_lde_expand launch: ") + hipGetErrorString((hipError_t)rc));
  return ntt_run(c, log_m, n_cols, coef, M, d_out, M, tmp, false, s);
}

// self-test hook: both inversions mod 2^255 - 19 (the Fermat chain and the safegcd one k_ed_fin uses) on n values of eight LE words
int32_t tmx_selftest_f16(tmx_ctx* c, uint32_t n, uint32_t doublings, const uint32_t* in_words, uint32_t* out_words) {
  if (!c || !in_words || !out_words) return TMX_ERR_BAD_ARG;
  if (n == 0) return TMX_OK;
  HIPCK(c, hipSetDevice(c->cfg.device));
  void *d_in = nullptr, *d_out = nullptr;
  hipError_t e = hipMalloc(&d_in, (size_t)n * 512);
  if (e == hipSuccess) e = hipMalloc(&d_out, (size_t)n * 1024);
  if (e == hipSuccess) e = hipMemcpy(d_in, in_words, (size_t)n * 512, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemsetAsync(d_out, 0, (size_t)n * 1024, c->side2);
  if (e == hipSuccess) e = (hipError_t)launch_selftest_f16(n, doublings, d_in, d_out, c->side2);
  if (e == hipSuccess) e = hipStreamSynchronize(c->side2);
  if (e == hipSuccess) e = hipMemcpy(out_words, d_out, (size_t)n * 1024, hipMemcpyDeviceToHost);
  if (d_in) (void)hipFree(d_in);
  if (d_out) (void)hipFree(d_out);
  if (e != hipSuccess) return fail(c, TMX_ERR_HIP, std::string("tmx_selftest_f16: ") + hipGetErrorString(e));
  return TMX_OK;
}

int32_t tmx_selftest_fe_invert(tmx_ctx* c, uint32_t n, const uint32_t* in_words, uint32_t* out_words) {
  if (!c || !in_words || !out_words) return TMX_ERR_BAD_ARG;
  if (n == 0) return TMX_OK;
  HIPCK(c, hipSetDevice(c->cfg.device));
  void *d_in = nullptr, *d_out = nullptr;
  hipError_t e = hipMalloc(&d_in, (size_t)n * 32);
  if (e == hipSuccess) e = hipMalloc(&d_out, (size_t)n * 64);
  if (e == hipSuccess) e = hipMemcpy(d_in, in_words, (size_t)n * 32, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = (hipError_t)launch_selftest_invert(n, d_in, d_out, c->side2);
  if (e == hipSuccess) e = hipStreamSynchronize(c->side2);
  if (e == hipSuccess) e = hipMemcpy(out_words, d_out, (size_t)n * 64, hipMemcpyDeviceToHost);
  if (d_in) (void)hipFree(d_in);
  if (d_out) (void)hipFree(d_out);
  if (e != hipSuccess) return fail(c, TMX_ERR_HIP, std::string("tmx_selftest_fe_invert: ") + hipGetErrorString(e));
  return TMX_OK;
}

}  // extern "C"

// ---- Poseidon over Goldilocks + Merkle caps --------------------------------------------------------------------------------------
// default round constants: the Poseidon paper's Grain LFSR (80 bits: field 1 | sbox 0 | n 64 | t 12 | R_F 8 | R_P 22 | thirty ones; 160 bits
// discarded; bits taken in pairs, the second kept when the first is 1; 64-bit big-endian integers, rejected when >= p) -- a third
// implementation beside the two of the test oracle (tests compare the streams)
static void poseidon_default_constants(std::vector<uint64_t>& k) {
  k.assign(POS_CONST_WORDS, 0);
  uint8_t st[80];
  int n = 0;
  const uint32_t fields[6][2] = {{1, 2}, {0, 4}, {64, 12}, {POS_T, 12}, {POS_RF, 10}, {POS_RP, 10}};
  for (auto& f : fields)
    for (int i = (int)f[1] - 1; i >= 0; i--) st[n++] = (uint8_t)((f[0] >> i) & 1u);
  while (n < 80) st[n++] = 1;
  int head = 0;
  auto next = [&]() {
    const uint8_t b = st[(head + 62) % 80] ^ st[(head + 51) % 80] ^ st[(head + 38) % 80] ^ st[(head + 23) % 80] ^ st[(head + 13) % 80] ^ st[head];
    st[head] = b;
    head = (head + 1) % 80;
    return b;
  };
  for (int i = 0; i < 160; i++) (void)next();
  uint32_t got = 0;
  while (got < POS_ROUNDS * POS_T) {
    uint64_t v = 0;
    for (int bits = 0; bits < 64;) {
      const uint8_t a = next(), b = next();
      if (a) { v = (v << 1) | b; bits++; }
    }
    if (v < 0xffffffff00000001ull) k[got++] = v;
  }
  static const uint64_t circ[POS_T] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
  for (uint32_t i = 0; i < POS_T; i++) { k[POS_ROUNDS * POS_T + i] = circ[i]; k[POS_ROUNDS * POS_T + POS_T + i] = i == 0 ? 8 : 0; }
}
// The device buffer behind the caller-visible constants: the tables of the merged partial rounds (poseidon.h), from the MDS matrix over
// the integers.  Returns the mode the kernels can run in.
static int poseidon_tables(const std::vector<uint64_t>& k, std::vector<uint64_t>& dev) {
  const unsigned __int128 P = 0xffffffff00000001ull;
  dev.assign(POS_CONST_WORDS_EXT, 0);
  std::copy(k.begin(), k.begin() + POS_CONST_WORDS, dev.begin());
  const uint64_t* circ = &k[POS_ROUNDS * POS_T];
  const uint64_t* diag = circ + POS_T;
  for (uint32_t i = 0; i < 2 * POS_T; i++)
    if (circ[i] >> 16) return POS_MODE_GENERAL;
  // M[r][j] = circ[(j - r) mod 12] + (j == r) diag[r]   (out[r] = sum_i circ[i] s[(i + r) mod 12] + diag[r] s[r], poseidon.hip: pos_mds)
  unsigned __int128 M[3][POS_T][POS_T];
  const unsigned __int128 LIM = (unsigned __int128)1 << 26;
  for (uint32_t r = 0; r < POS_T; r++)
    for (uint32_t j = 0; j < POS_T; j++) M[0][r][j] = circ[(j + POS_T - r) % POS_T] + (j == r ? diag[r] : 0);
  for (int e = 1; e < 3; e++)
    for (uint32_t r = 0; r < POS_T; r++)
      for (uint32_t j = 0; j < POS_T; j++) {
        unsigned __int128 a = 0;
        for (uint32_t t = 0; t < POS_T; t++) a += M[e - 1][r][t] * M[0][t][j];
        if (a >= LIM) return POS_MODE_SMALL;  // (entries only grow: M^3 decides)
        M[e][r][j] = a;
      }
  for (uint32_t r = 0; r < POS_T; r++)
    for (uint32_t j = 0; j < POS_T; j++)
      if (M[0][r][j] >= LIM) return POS_MODE_SMALL;
  uint32_t* U = reinterpret_cast<uint32_t*>(&dev[POS_X_U32]);
  for (uint32_t i = 0; i < POS_T; i++) {
    U[POS_U_R1 + i] = (uint32_t)M[0][0][i]; U[POS_U_R2 + i] = (uint32_t)M[1][0][i];
    U[POS_U_C2 + i] = (uint32_t)M[1][i][0]; U[POS_U_C1 + i] = (uint32_t)M[0][i][0];
    for (uint32_t j = 0; j < POS_T; j++) U[POS_U_M3 + POS_T * i + j] = (uint32_t)M[2][i][j];
  }
  auto mat_vec = [&](int e, const uint64_t* v, uint64_t* out) {  // out = M^(e+1) v mod p
    for (uint32_t r = 0; r < POS_T; r++) {
      unsigned __int128 a = 0;
      for (uint32_t j = 0; j < POS_T; j++) a = (a + M[e][r][j] * v[j]) % P;
      out[r] = (uint64_t)a;
    }
  };
  for (uint32_t g = 0; g < POS_MERGE_GROUPS; g++) {
    const uint32_t r = POS_MERGE_FIRST + 3 * g;
    const uint64_t *rc1 = &k[(r + 1) * POS_T], *rc2 = &k[(r + 2) * POS_T], *rc3 = &k[(r + 3) * POS_T];
    uint64_t m_rc1[POS_T], m2_rc1[POS_T], m_rc2[POS_T];
    mat_vec(0, rc1, m_rc1); mat_vec(1, rc1, m2_rc1); mat_vec(0, rc2, m_rc2);
    uint64_t* G = &dev[POS_X_GROUPS + g * POS_X_GROUP_WORDS];
    G[0] = rc1[0];
    G[1] = (uint64_t)(((unsigned __int128)m_rc1[0] + rc2[0]) % P);
    for (uint32_t i = 0; i < POS_T; i++) G[2 + i] = (uint64_t)(((unsigned __int128)m2_rc1[i] + m_rc2[i] + rc3[i]) % P);
  }
  return POS_MODE_MERGE3;
}
static int32_t poseidon_ready(tmx_ctx* c, hipStream_t s) {
  if (c->pos_consts.empty()) poseidon_default_constants(c->pos_consts);
  if (!c->d_pos_consts) {
    HIPCK(c, hipSetDevice(c->cfg.device));
    HIPCK(c, hipMalloc(&c->d_pos_consts, POS_CONST_WORDS_EXT * 8));
    c->pos_dirty = true;
  }
  if (c->pos_dirty) {
    std::vector<uint64_t> dev;
    int mode = poseidon_tables(c->pos_consts, dev);
    if (const char* v = std::getenv("TMX_POSEIDON_MERGE"))  // TMX_POSEIDON_MERGE=0: one dense layer per partial round (times the merged form against it)
      if (v[0] == '0' && mode == POS_MODE_MERGE3) mode = POS_MODE_SMALL;
    HIPCK(c, hipMemcpyAsync(c->d_pos_consts, dev.data(), POS_CONST_WORDS_EXT * 8, hipMemcpyHostToDevice, s));
    HIPCK(c, hipStreamSynchronize(s));  // once per change of the tables: later calls may come on another stream (and `dev` is a local)
    c->pos_dirty = false;
    c->pos_mode = mode;
  }
  return TMX_OK;
}

extern "C" {

int32_t tmx_poseidon_set_constants(tmx_ctx* c, const uint64_t* rc, const uint64_t* circ, const uint64_t* diag) {
  if (!c) return TMX_ERR_BAD_ARG;
  if (c->pos_consts.empty()) poseidon_default_constants(c->pos_consts);
  const uint64_t P = 0xffffffff00000001ull;
  if (rc) {
    for (uint32_t i = 0; i < POS_ROUNDS * POS_T; i++) c->pos_consts[i] = rc[i] % P;
    c->pos_rc_injected = true;
  }
  if (circ) for (uint32_t i = 0; i < POS_T; i++) c->pos_consts[POS_ROUNDS * POS_T + i] = circ[i] % P;
  if (diag) for (uint32_t i = 0; i < POS_T; i++) c->pos_consts[POS_ROUNDS * POS_T + POS_T + i] = diag[i] % P;
  if (c->d_pos_consts) {  // kernels in flight may still read the old tables
    HIPCK(c, hipSetDevice(c->cfg.device));
    HIPCK(c, hipDeviceSynchronize());
  }
  c->pos_dirty = true;
  return TMX_OK;
}

int32_t tmx_poseidon_constants_injected(const tmx_ctx* c) { return c ? (c->pos_rc_injected ? 1 : 0) : TMX_ERR_BAD_ARG; }

uint64_t tmx_poseidon_merkle_digests(uint32_t log_n, uint32_t cap_height) {
  if (log_n > 30 || cap_height > log_n) return 0;
  uint64_t total = 0;
  for (uint32_t k = 0; k + cap_height <= log_n; k++) total += 1ull << (log_n - k);
  return total;
}

int32_t tmx_poseidon_merkle_device(tmx_ctx* c, uint32_t log_n, uint32_t n_cols, const uint64_t* d_cols, uint32_t cap_height, uint64_t* d_levels,
                                   void* hip_stream) {
  if (!c || !d_cols || !d_levels || n_cols == 0) return TMX_ERR_BAD_ARG;
  if (log_n > 30 || cap_height > log_n) return fail(c, TMX_ERR_BAD_ARG, "cap_height must not exceed log_n (<= 30)");
  hipStream_t s = reinterpret_cast<hipStream_t>(hip_stream);
  int32_t st = poseidon_ready(c, s);
  if (st) return st;
  int rc = launch_poseidon_leaves(c->d_pos_consts, c->pos_mode, log_n, n_cols, d_cols, d_levels, s);
  uint64_t* cur = d_levels;
  for (uint32_t k = 0; !rc && k + cap_height < log_n; k++) {
    const uint64_t cnt = 1ull << (log_n - k);
    rc = launch_poseidon_level(c->d_pos_consts, c->pos_mode, cnt / 2, cur, cur + 4 * cnt, s);
    cur += 4 * cnt;
  }
  if (rc) return fail(c, TMX_ERR_HIP, std::string("k_poseidon launch: ") + hipGetErrorString((hipError_t)rc));
  return TMX_OK;
}

int32_t tmx_poseidon_permute(tmx_ctx* c, uint32_t n, const uint64_t* in, uint64_t* out) {
  if (!c || !in || !out) return TMX_ERR_BAD_ARG;
  if (n == 0) return TMX_OK;
  HIPCK(c, hipSetDevice(c->cfg.device));
  int32_t st = poseidon_ready(c, c->side2);
  if (st) return st;
  void *d_in = nullptr, *d_out = nullptr;
  hipError_t e = hipMalloc(&d_in, (size_t)n * 96);
  if (e == hipSuccess) e = hipMalloc(&d_out, (size_t)n * 96);
  if (e == hipSuccess) e = hipMemcpy(d_in, in, (size_t)n * 96, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = (hipError_t)launch_poseidon_permute(c->d_pos_consts, c->pos_mode, n, d_in, d_out, c->side2);
  if (e == hipSuccess) e = hipStreamSynchronize(c->side2);
  if (e == hipSuccess) e = hipMemcpy(out, d_out, (size_t)n * 96, hipMemcpyDeviceToHost);
  if (d_in) (void)hipFree(d_in);
  if (d_out) (void)hipFree(d_out);
  if (e != hipSuccess) return fail(c, TMX_ERR_HIP, std::string("tmx_poseidon_permute: ") + hipGetErrorString(e));
  return TMX_OK;
}

}  // extern "C"

// ---- multi-GPU: RCCL behind the C ABI (include/tmx.h "multi-GPU") -------------------------------------------------------
// RCCL is bound at run time: dlopen / dlsym, never a DT_NEEDED -- the process that loads libtmx decides which librccl (and which HIP
// runtime under it) it runs on.  PyTorch bundles its own; a C / Rust host gets /opt/rocm's.
#include <dlfcn.h>
#include <link.h>

namespace {
struct RcclId { char internal[TMX_UNIQUE_ID_BYTES]; };
static_assert(sizeof(RcclId) == 128, "ncclUniqueId is 128 bytes");
struct Rccl {
  void* lib = nullptr;
  std::string err;
  int (*GetUniqueId)(RcclId*) = nullptr;
  int (*CommInitRank)(void**, int, RcclId, int) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  int (*CommAbort)(void*) = nullptr;               // optional (every NCCL >= 2.4 / RCCL has both): without them an abort is a destroy
  int (*CommGetAsyncError)(void*, int*) = nullptr;
  int (*Broadcast)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
std::mutex g_rccl_mu;
Rccl g_rccl;
int find_loaded_rccl(struct dl_phdr_info* info, size_t, void* out) {
  if (info->dlpi_name && std::strstr(info->dlpi_name, "librccl")) {
    *reinterpret_cast<std::string*>(out) = info->dlpi_name;
    return 1;
  }
  return 0;
}
// empty on success, else why not (returned by value: the caller's copy cannot change under it)
std::string rccl_load() {
  std::lock_guard<std::mutex> lk(g_rccl_mu);
  Rccl& R = g_rccl;
  if (R.lib) return std::string();
  // an explicit TMX_RCCL_LIB is the ONLY candidate (a host that names a library means it, and a wrong name must be an error, not a silent
  // fall-through to some other copy -- tests/fake_rccl runs two ranks on one GPU this way); otherwise the copy the process already runs on
  // (PyTorch's), then the loader's search path
  std::vector<std::string> names;
  const char* forced = std::getenv("TMX_RCCL_LIB");
  if (forced && forced[0]) {
    names.push_back(forced);
  } else {
    std::string loaded;
    (void)dl_iterate_phdr(find_loaded_rccl, &loaded);
    if (!loaded.empty()) names.push_back(loaded);
    names.push_back("librccl.so.1");
    names.push_back("librccl.so");
    names.push_back("/opt/rocm/lib/librccl.so.1");
  }
  void* h = nullptr;
  std::string why;
  for (const std::string& nm : names) {
    if ((h = dlopen(nm.c_str(), RTLD_NOW | RTLD_GLOBAL))) break;
    const char* e = dlerror();  // (one call: dlerror clears the message it returns)
    if (why.empty()) why = e ? e : "?";
  }
  if (!h) return R.err = "librccl not loadable: " + why;
  Rccl T;
  auto sym = [&](const char* n) { return dlsym(h, n); };
  T.GetUniqueId = reinterpret_cast<decltype(T.GetUniqueId)>(sym("ncclGetUniqueId"));
  T.CommInitRank = reinterpret_cast<decltype(T.CommInitRank)>(sym("ncclCommInitRank"));
  T.CommDestroy = reinterpret_cast<decltype(T.CommDestroy)>(sym("ncclCommDestroy"));
  T.CommAbort = reinterpret_cast<decltype(T.CommAbort)>(sym("ncclCommAbort"));
  T.CommGetAsyncError = reinterpret_cast<decltype(T.CommGetAsyncError)>(sym("ncclCommGetAsyncError"));
  T.Broadcast = reinterpret_cast<decltype(T.Broadcast)>(sym("ncclBroadcast"));
  T.AllGather = reinterpret_cast<decltype(T.AllGather)>(sym("ncclAllGather"));
  T.GroupStart = reinterpret_cast<decltype(T.GroupStart)>(sym("ncclGroupStart"));
  T.GroupEnd = reinterpret_cast<decltype(T.GroupEnd)>(sym("ncclGroupEnd"));
  T.GetErrorString = reinterpret_cast<decltype(T.GetErrorString)>(sym("ncclGetErrorString"));
  if (!T.GetUniqueId || !T.CommInitRank || !T.CommDestroy || !T.Broadcast || !T.AllGather || !T.GroupStart || !T.GroupEnd || !T.GetErrorString) {
    (void)dlclose(h);  // nothing of a half-bound library stays: the next attempt starts from scratch
    return R.err = "librccl lacks one of ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclBroadcast / ncclAllGather / ncclGroupStart / "
                   "ncclGroupEnd / ncclGetErrorString";
  }
  T.lib = h;
  R = T;
  return std::string();
}
int32_t rccl_fail(tmx_ctx* c, const char* what, int rc) {
  return fail(c, TMX_ERR_RCCL, std::string(what) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?"));
}
// A collective is entered by every rank or by none.  A rank that fails LOCALLY between the start of a sharded call and its exchange (its shard
// exceeds its context's max_batch, a launch fails) would leave the others waiting in the collective for ever: it aborts the communicator
// instead (ncclCommAbort: its peers' pending operations fail), and so does a rank whose collective reports an error.  Either way the call
// returns TMX_ERR_RCCL, the context refuses further sharded calls, and every rank makes a new communicator (tmx_comm_create).
void comm_abort(tmx_ctx* c) {
  if (!c->comm) return;
  if (c->have_streams) (void)hipSetDevice(c->cfg.device);
  if (g_rccl.CommAbort) (void)g_rccl.CommAbort(c->comm);
  else if (g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
  c->comm = nullptr;
  c->comm_aborted = true;
}
int32_t shard_fail(tmx_ctx* c, int32_t st) {
  if (st == TMX_OK || !c->comm) return st;
  const std::string local = c->err;
  comm_abort(c);
  return fail(c, TMX_ERR_RCCL, "local failure in front of a collective (" + std::string(tmx_status_str(st)) + (local.empty() ? "" : ": " + local) +
                                   "): communicator aborted so that no peer waits for this rank; tmx_comm_create again on every rank");
}
int32_t comm_usable(tmx_ctx* c) {
  return c->comm_aborted ? fail(c, TMX_ERR_RCCL, "the communicator of this context was aborted: tmx_comm_create again on every rank") : TMX_OK;
}
int32_t rccl_fail_abort(tmx_ctx* c, const char* what, int rc) {
  const int32_t st = rccl_fail(c, what, rc);
  comm_abort(c);
  return st;
}
constexpr int RCCL_UINT8 = 1;  // ncclUint8
// every rank's slice [lo_r, hi_r) of `n_items` records of `rec_bytes` becomes resident on every rank: one exchange, in place.
// Equal slices (n_items a multiple of the world: 256 proofs or 512 lanes over 2 / 4 / 8 ranks, BASELINE configs[3] / [4]) are ONE
// ncclAllGather -- the collective north_star names and the one RCCL schedules over all seven xGMI links at once -- in its in-place form
// (sendbuff = recvbuff + rank * count); ragged slices fall back to a group of broadcasts, each rank the root of its own slice.
int32_t exchange_slices(tmx_ctx* c, void* d_buf, uint64_t n_items, size_t rec_bytes, hipStream_t s) {
  if (!c->comm || !d_buf || n_items == 0) return TMX_OK;
  int rc;
  if (n_items % c->comm_world == 0) {
    const size_t per = (size_t)(n_items / c->comm_world) * rec_bytes;
    uint8_t* base = reinterpret_cast<uint8_t*>(d_buf);
    rc = g_rccl.AllGather(base + (size_t)c->comm_rank * per, base, per, RCCL_UINT8, c->comm, s);
    if (rc) return rccl_fail_abort(c, "ncclAllGather", rc);
    return TMX_OK;
  }
  rc = g_rccl.GroupStart();
  if (rc) return rccl_fail_abort(c, "ncclGroupStart", rc);
  for (uint32_t r = 0; r < c->comm_world; r++) {
    uint64_t lo, hi;
    tmx_shard_range(n_items, r, c->comm_world, &lo, &hi);
    if (hi == lo) continue;
    uint8_t* p = reinterpret_cast<uint8_t*>(d_buf) + lo * rec_bytes;
    rc = g_rccl.Broadcast(p, p, (size_t)(hi - lo) * rec_bytes, RCCL_UINT8, (int)r, c->comm, s);
    if (rc) { (void)g_rccl.GroupEnd(); return rccl_fail_abort(c, "ncclBroadcast", rc); }
  }
  rc = g_rccl.GroupEnd();
  if (rc) return rccl_fail_abort(c, "ncclGroupEnd", rc);
  return TMX_OK;
}
}  // namespace

extern "C" {

void tmx_shard_range(uint64_t n_items, uint32_t rank, uint32_t world, uint64_t* lo, uint64_t* hi) {
  if (world == 0) world = 1;
  if (rank >= world) rank = world - 1;
  const uint64_t base = n_items / world, rem = n_items % world;
  const uint64_t a = (uint64_t)rank * base + (rank < rem ? rank : rem);
  if (lo) *lo = a;
  if (hi) *hi = a + base + (rank < rem ? 1 : 0);
}

int32_t tmx_comm_unique_id(uint8_t out[TMX_UNIQUE_ID_BYTES]) {
  if (!out) return TMX_ERR_BAD_ARG;
  if (!(g_tls_err = rccl_load()).empty()) return TMX_ERR_RCCL;
  RcclId id;
  std::memset(&id, 0, sizeof id);
  if (const int rc = g_rccl.GetUniqueId(&id)) {
    g_tls_err = std::string("ncclGetUniqueId: ") + g_rccl.GetErrorString(rc);
    return TMX_ERR_RCCL;
  }
  std::memcpy(out, id.internal, TMX_UNIQUE_ID_BYTES);
  return TMX_OK;
}

int32_t tmx_comm_create(tmx_ctx* c, const uint8_t* unique_id, uint32_t rank, uint32_t world) {
  if (!c || world == 0 || rank >= world || (world > 1 && !unique_id)) return TMX_ERR_BAD_ARG;
  if (unique_id && c->cfg.device < 0) return TMX_ERR_BAD_ARG;
  int32_t st = tmx_comm_destroy(c);
  if (st) return st;
  c->comm_aborted = false;
  if (world == 1 && !unique_id) { c->comm_rank = 0; c->comm_world = 1; return TMX_OK; }  // nothing to exchange, nothing to load
  // (world == 1 WITH an id makes a real one-rank communicator: the exchange then runs through RCCL -- a broadcast to itself -- which is
  // how the 1-GPU boxes exercise this path)
  {
    const std::string why = rccl_load();
    if (!why.empty()) return fail(c, TMX_ERR_RCCL, why);
  }
  HIPCK(c, hipSetDevice(c->cfg.device));
  RcclId id;
  std::memcpy(id.internal, unique_id, TMX_UNIQUE_ID_BYTES);
  void* comm = nullptr;
  const int rc = g_rccl.CommInitRank(&comm, (int)world, id, (int)rank);
  if (rc) return rccl_fail(c, "ncclCommInitRank", rc);
  c->comm = comm; c->comm_rank = rank; c->comm_world = world;
  c->comm_aborted = false;
  return TMX_OK;
}

int32_t tmx_comm_destroy(tmx_ctx* c) {
  if (!c) return TMX_ERR_BAD_ARG;
  if (c->comm) {
    if (c->have_streams) { (void)hipSetDevice(c->cfg.device); (void)hipDeviceSynchronize(); }
    const int rc = g_rccl.CommDestroy ? g_rccl.CommDestroy(c->comm) : 0;
    c->comm = nullptr;
    if (rc) return rccl_fail(c, "ncclCommDestroy", rc);
  }
  c->comm_rank = 0; c->comm_world = 1;
  return TMX_OK;
}

int32_t tmx_comm_info(const tmx_ctx* c, uint32_t* rank, uint32_t* world) {
  if (!c) return TMX_ERR_BAD_ARG;
  if (rank) *rank = c->comm_rank;
  if (world) *world = c->comm_world;
  return TMX_OK;
}

int32_t tmx_comm_abort(tmx_ctx* c) {
  if (!c) return TMX_ERR_BAD_ARG;
  comm_abort(c);
  return TMX_OK;
}

int32_t tmx_comm_sync(tmx_ctx* c, void* hip_stream, uint32_t timeout_ms) {
  if (!c) return TMX_ERR_BAD_ARG;
  if (c->comm_aborted) return comm_usable(c);
  HIPCK(c, hipSetDevice(c->cfg.device));
  hipStream_t s = reinterpret_cast<hipStream_t>(hip_stream);
  const auto t0 = std::chrono::steady_clock::now();
  for (uint32_t spin = 0;; spin++) {
    const hipError_t q = hipStreamQuery(s);
    if (q == hipSuccess) return TMX_OK;
    if (q != hipErrorNotReady) { (void)hipGetLastError(); comm_abort(c); return fail(c, TMX_ERR_HIP, std::string("hipStreamQuery: ") + hipGetErrorString(q)); }
    (void)hipGetLastError();
    if (c->comm && g_rccl.CommGetAsyncError && (spin & 15) == 0) {
      int async_rc = 0;
      const int rc = g_rccl.CommGetAsyncError(c->comm, &async_rc);
      if (rc || async_rc) return rccl_fail_abort(c, "asynchronous error of the communicator (a peer aborted or died)", rc ? rc : async_rc);
    }
    if (timeout_ms) {
      const auto ms = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
      if (ms >= (long long)timeout_ms) {
        comm_abort(c);
        return fail(c, TMX_ERR_RCCL, "tmx_comm_sync: the stream did not drain within " + std::to_string(timeout_ms) + " ms: communicator aborted");
      }
    }
    if (spin > 200) std::this_thread::sleep_for(std::chrono::microseconds(50));
  }
}

int32_t tmx_witness_batch_sharded_device(tmx_ctx* c, int32_t kind, uint32_t n_total, const void* d_proofs, const void* d_targets,
                                         const void* d_trusteds, void* d_out_elems, void* d_reports, uint32_t gather, void* hip_stream) {
  int32_t st = check_batch_args(c, kind, 0, d_proofs, d_targets, d_trusteds);
  if (st) return st;
  if (n_total == 0) return TMX_OK;
  // (a one-rank communicator made WITH an id exchanges through RCCL too: the guard is on the communicator, not on the world size)
  if (gather && (c->comm || c->comm_world > 1) && (!d_out_elems || !d_reports)) return fail(c, TMX_ERR_BAD_ARG, "gather needs the row and report buffers");
  uint64_t lo, hi;
  tmx_shard_range(n_total, c->comm_rank, c->comm_world, &lo, &hi);
  if (gather && (st = comm_usable(c))) return st;
  // (from here on a failure is LOCAL -- this rank's shard, this rank's context -- while the peers go on to the exchange: shard_fail)
  if (hi - lo > c->cfg.max_batch) { st = fail(c, TMX_ERR_CAPACITY, "this rank's shard exceeds the context's max_batch"); return gather ? shard_fail(c, st) : st; }
  const uint32_t n = c->cfg.n_max;
  const size_t row_bytes = (size_t)tmx_elem_stride(kind, n) * 8;
  auto at = [](const void* p, size_t off) { return p ? reinterpret_cast<const uint8_t*>(p) + off : nullptr; };
  if (hi > lo) {
    st = tmx_witness_batch_device(c, kind, (uint32_t)(hi - lo), at(d_proofs, lo * PR_STRIDE), at(d_targets, lo * n * VR_STRIDE),
                                  at(d_trusteds, lo * n * HR_STRIDE), const_cast<uint8_t*>(at(d_out_elems, lo * row_bytes)),
                                  const_cast<uint8_t*>(at(d_reports, lo * sizeof(tmx_report))), hip_stream);
    if (st) return gather ? shard_fail(c, st) : st;
  }
  if (!gather) return TMX_OK;
  hipStream_t s = reinterpret_cast<hipStream_t>(hip_stream);
  if ((st = exchange_slices(c, d_out_elems, n_total, row_bytes, s))) return st;
  return exchange_slices(c, d_reports, n_total, sizeof(tmx_report), s);
}

int32_t tmx_witness_validator_sharded_device(tmx_ctx* c, int32_t kind, uint32_t n_proofs, const void* d_proofs, const void* d_targets,
                                             const void* d_trusteds, void* d_out_elems, void* d_reports, void* hip_stream) {
  int32_t st = check_batch_args(c, kind, n_proofs, d_proofs, d_targets, d_trusteds);
  if (st) return st;
  if (n_proofs == 0) return TMX_OK;
  if ((st = comm_usable(c))) return st;
  const uint64_t lanes = (uint64_t)n_proofs * c->cfg.n_max;
  uint64_t lo, hi;
  tmx_shard_range(lanes, c->comm_rank, c->comm_world, &lo, &hi);
  uint8_t* ed = reinterpret_cast<uint8_t*>(c->d_ed);
  if (hi > lo) {
    st = tmx_eddsa_lanes_device(c, (uint32_t)(hi - lo), reinterpret_cast<const uint8_t*>(d_targets) + lo * VR_STRIDE, ed + lo * ED_STRIDE, hip_stream);
    if (st) return shard_fail(c, st);
  }
  if ((st = exchange_slices(c, ed, lanes, ED_STRIDE, reinterpret_cast<hipStream_t>(hip_stream)))) return st;
  return tmx_finish_batch_device(c, kind, n_proofs, d_proofs, d_targets, d_trusteds, ed, d_out_elems, d_reports, hip_stream);
}

// ---- Level-2 trace rows across the ranks (north_star: "validators AND TRACE ROWS shard across the 8 GPUs ... with an RCCL all-gather").
// The rows are the one payload of this path that is big enough for xGMI to matter: 41 MB per proof at N = 128, 136 MB at N = 512.
int32_t tmx_trace_rows_sharded_device(tmx_ctx* c, int32_t kind, uint32_t n_total, const void* d_targets, const void* d_trusteds, void* d_trace_out,
                                      uint32_t sections, uint32_t gather, void* hip_stream) {
  if (!c || (kind != TMX_KIND_SKIP && kind != TMX_KIND_STEP) || !d_targets || !d_trace_out) return TMX_ERR_BAD_ARG;
  if (n_total == 0) return TMX_OK;
  uint64_t lo, hi;
  tmx_shard_range(n_total, c->comm_rank, c->comm_world, &lo, &hi);
  const uint32_t n = c->cfg.n_max;
  const size_t row_bytes = (size_t)trace_elems((uint32_t)kind, n) * 8;
  auto at = [](const void* p, size_t off) { return p ? reinterpret_cast<const uint8_t*>(p) + off : nullptr; };
  if (gather) { const int32_t su = comm_usable(c); if (su) return su; }
  if (hi > lo) {  // the Level-1 records this reads are those of the rank's own shard: tmx_witness_batch_sharded_device of the same n_total came first
    int32_t st = tmx_trace_rows_device(c, kind, (uint32_t)(hi - lo), at(d_targets, lo * n * VR_STRIDE), at(d_trusteds, lo * n * HR_STRIDE),
                                       const_cast<uint8_t*>(at(d_trace_out, lo * row_bytes)), sections, hip_stream);
    if (st) return gather ? shard_fail(c, st) : st;
  }
  if (!gather) return TMX_OK;
  return exchange_slices(c, d_trace_out, n_total, row_bytes, reinterpret_cast<hipStream_t>(hip_stream));
}

int32_t tmx_trace_rows_validator_sharded_device(tmx_ctx* c, int32_t kind, uint32_t n_proofs, const void* d_targets, const void* d_trusteds, void* d_trace_out,
                                                uint32_t sections, void* hip_stream) {
  if (!c || (kind != TMX_KIND_SKIP && kind != TMX_KIND_STEP) || !d_targets || !d_trace_out) return TMX_ERR_BAD_ARG;
  if (n_proofs == 0) return TMX_OK;
  const uint32_t n = c->cfg.n_max;
  const uint64_t lanes = (uint64_t)n_proofs * n;
  uint64_t lo, hi;
  tmx_shard_range(lanes, c->comm_rank, c->comm_world, &lo, &hi);
  // the per-lane sections (ladders: 266 KB per lane, SHA-512: 23 KB) for this rank's lanes; the small per-proof sections on every rank
  int32_t st = comm_usable(c);
  if (st) return st;
  st = trace_rows_impl(c, kind, n_proofs, d_targets, d_trusteds, d_trace_out, sections, hip_stream, (uint32_t)lo, (uint32_t)(hi - lo));
  if (st) return shard_fail(c, st);
  if (!c->comm) return st;
  hipStream_t s = reinterpret_cast<hipStream_t>(hip_stream);
  uint8_t* out = reinterpret_cast<uint8_t*>(d_trace_out);
  const size_t row_bytes = (size_t)trace_elems((uint32_t)kind, n) * 8;
  struct Sec { uint32_t bit; size_t off, per_lane; };
  const Sec secs[2] = {{TMX_TRACE_LADDERS, 0, (size_t)2 * TR_LADDER_ROWS * TR_LADDER_ROW * 8},
                       {TMX_TRACE_SHA512, (size_t)n * 2 * TR_LADDER_ROWS * TR_LADDER_ROW * 8, (size_t)2 * 80 * TR_SHA512_ROW * 8}};
  for (const Sec& sc : secs) {
    if (!(sections & sc.bit)) continue;
    if (n_proofs == 1) {  // one proof: its lanes' slabs are contiguous -- the all-gather of SURVEY 8(e) when the lanes divide evenly
      if ((st = exchange_slices(c, out + sc.off, lanes, sc.per_lane, s))) return st;
      continue;
    }
    // several proofs: a rank's lanes are one run per proof it touches; every run is broadcast by its owner, grouped
    int rc = g_rccl.GroupStart(), in_group = 0;
    if (rc) return rccl_fail_abort(c, "ncclGroupStart", rc);
    for (uint32_t r = 0; r < c->comm_world; r++) {
      uint64_t rlo, rhi;
      tmx_shard_range(lanes, r, c->comm_world, &rlo, &rhi);
      for (uint64_t p = rlo / n; p < n_proofs && p * n < rhi; p++) {
        const uint64_t a = std::max<uint64_t>(rlo, p * n), b = std::min<uint64_t>(rhi, (p + 1) * n);
        if (b <= a) continue;
        uint8_t* ptr = out + p * row_bytes + sc.off + (a - p * n) * sc.per_lane;
        rc = g_rccl.Broadcast(ptr, ptr, (size_t)(b - a) * sc.per_lane, RCCL_UINT8, (int)r, c->comm, s);
        if (rc) { (void)g_rccl.GroupEnd(); return rccl_fail_abort(c, "ncclBroadcast", rc); }
        if (++in_group == 32) {  // (bounded groups: a group is one fused launch, and a thousand-entry group helps nobody)
          if ((rc = g_rccl.GroupEnd())) return rccl_fail_abort(c, "ncclGroupEnd", rc);
          if ((rc = g_rccl.GroupStart())) return rccl_fail_abort(c, "ncclGroupStart", rc);
          in_group = 0;
        }
      }
    }
    if ((rc = g_rccl.GroupEnd())) return rccl_fail_abort(c, "ncclGroupEnd", rc);
  }
  return TMX_OK;
}

}  // extern "C"

// ---- the commit pipeline on the device: section rows -> columns -> LDE -> Poseidon Merkle cap (include/tmx.h) ---------------------------
extern "C" {

int32_t tmx_trace_commit_shape(int32_t kind, uint32_t n, uint32_t section, uint32_t* log_rows, uint32_t* width) {
  if ((kind != TMX_KIND_SKIP && kind != TMX_KIND_STEP) || n == 0 || n > TMX_N_MAX_LIMIT) return TMX_ERR_BAD_ARG;
  uint64_t off, rows;
  uint32_t w;
  if (!trace_section_geom((uint32_t)kind, n, section, &off, &rows, &w)) return TMX_ERR_BAD_ARG;
  uint32_t lg = 6;  // (the transpose moves 64 rows per workgroup)
  while (((uint64_t)1 << lg) < rows) lg++;
  if (log_rows) *log_rows = lg;
  if (width) *width = w;
  return TMX_OK;
}

int32_t tmx_trace_commit_device(tmx_ctx* c, int32_t kind, uint32_t n_proofs, uint32_t section, uint32_t log_blowup, uint32_t cap_height,
                                const void* d_trace_rows, uint64_t* d_cap, void* hip_stream) {
  if (!c || !d_trace_rows || !d_cap || n_proofs == 0) return TMX_ERR_BAD_ARG;
  uint32_t log_n = 0, width = 0;
  if (tmx_trace_commit_shape(kind, c->cfg.n_max, section, &log_n, &width)) return fail(c, TMX_ERR_BAD_ARG, "section must be one row table of the trace block");
  const uint32_t log_m = log_n + log_blowup;
  if (log_m > TMX_NTT_MAX_LOG) return fail(c, TMX_ERR_CAPACITY, "log_rows + log_blowup exceeds TMX_NTT_MAX_LOG");
  if (cap_height > log_m) return fail(c, TMX_ERR_BAD_ARG, "cap_height exceeds the height of the tree");
  const uint64_t n_cols64 = (uint64_t)n_proofs * width;
  if (n_cols64 > 0xffffffffull) return fail(c, TMX_ERR_CAPACITY, "too many columns");
  const uint32_t n_cols = (uint32_t)n_cols64;
  hipStream_t s = reinterpret_cast<hipStream_t>(hip_stream);
  uint64_t off, rows;
  uint32_t w;
  (void)trace_section_geom((uint32_t)kind, c->cfg.n_max, section, &off, &rows, &w);
  const size_t cols_b = ((size_t)n_cols << log_n) * 8, lde_b = ((size_t)n_cols << log_m) * 8;
  const size_t lev_b = (size_t)tmx_poseidon_merkle_digests(log_m, cap_height) * 32, want = cols_b + lde_b + lev_b;
  HIPCK(c, hipSetDevice(c->cfg.device));
  if (c->commit_bytes < want) {
    if (c->d_commit) { HIPCK(c, hipStreamSynchronize(s)); HIPCK(c, hipFree(c->d_commit)); c->d_commit = nullptr; c->commit_bytes = 0; }
    size_t free_b = 0, total_b = 0;
    // (the LDE's own scratch is twice the extended columns again: refuse what cannot fit instead of driving the device out of memory)
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && want + 2 * lde_b > free_b + c->ntt_tmp_bytes)
      return fail(c, TMX_ERR_CAPACITY, "commit pipeline needs " + std::to_string((want + 2 * lde_b) >> 20) + " MiB of scratch, " + std::to_string(free_b >> 20) + " MiB free");
    HIPCK(c, hipMalloc(&c->d_commit, want));
    c->commit_bytes = want;
  }
  for (auto& e : c->ev_commit)
    if (!e) HIPCK(c, hipEventCreate(&e));
  uint8_t* base = reinterpret_cast<uint8_t*>(c->d_commit);
  uint64_t* cols = reinterpret_cast<uint64_t*>(base);
  uint64_t* lde = reinterpret_cast<uint64_t*>(base + cols_b);
  uint64_t* levels = reinterpret_cast<uint64_t*>(base + cols_b + lde_b);
  HIPCK(c, hipEventRecord(c->ev_commit[0], s));
  int rc = launch_trace_to_columns(d_trace_rows, trace_elems((uint32_t)kind, c->cfg.n_max), off, rows, width, log_n, n_proofs, cols, s);
  if (rc) return fail(c, TMX_ERR_HIP, std::string("k_trace_to_columns launch: ") + hipGetErrorString((hipError_t)rc));
  HIPCK(c, hipEventRecord(c->ev_commit[1], s));
  int32_t st = tmx_lde_goldilocks_device(c, log_n, log_blowup, n_cols, cols, lde, hip_stream);
  if (st) return st;
  HIPCK(c, hipEventRecord(c->ev_commit[2], s));
  st = tmx_poseidon_merkle_device(c, log_m, n_cols, lde, cap_height, levels, hip_stream);
  if (st) return st;
  const uint64_t n_dig = tmx_poseidon_merkle_digests(log_m, cap_height), n_cap = (uint64_t)1 << cap_height;
  HIPCK(c, hipMemcpyAsync(d_cap, levels + 4 * (n_dig - n_cap), n_cap * 32, hipMemcpyDeviceToDevice, s));
  HIPCK(c, hipEventRecord(c->ev_commit[3], s));
  return TMX_OK;
}

// the commit of a proof-sharded batch: every rank commits the columns of ITS proofs (their rows are already in its HBM: no row crosses a
// link), its cap lands in slot `rank` of d_caps, and ONE all-gather of the caps (16 digests each at cap_height 4: 512 B) leaves all of
// them on every rank -- the batch's commitment is the `world` caps, each over n_shard * width columns
int32_t tmx_trace_commit_sharded_device(tmx_ctx* c, int32_t kind, uint32_t n_total, uint32_t section, uint32_t log_blowup, uint32_t cap_height,
                                        const void* d_trace_rows, uint64_t* d_caps, void* hip_stream) {
  if (!c || !d_trace_rows || !d_caps || n_total == 0) return TMX_ERR_BAD_ARG;
  uint64_t lo, hi;
  tmx_shard_range(n_total, c->comm_rank, c->comm_world, &lo, &hi);
  const size_t row_bytes = (size_t)trace_elems((uint32_t)kind, c->cfg.n_max) * 8, cap_bytes = ((size_t)4 << cap_height) * 8;
  hipStream_t s = reinterpret_cast<hipStream_t>(hip_stream);
  uint64_t* mine = d_caps + (size_t)c->comm_rank * (cap_bytes / 8);
  if (hi > lo) {
    int32_t st = comm_usable(c);
    if (st) return st;
    st = tmx_trace_commit_device(c, kind, (uint32_t)(hi - lo), section, log_blowup, cap_height, reinterpret_cast<const uint8_t*>(d_trace_rows) + lo * row_bytes,
                                 mine, hip_stream);
    if (st) return shard_fail(c, st);
  } else {
    HIPCK(c, hipMemsetAsync(mine, 0, cap_bytes, s));  // an empty shard commits to nothing: a zero cap
  }
  return exchange_slices(c, d_caps, c->comm_world, cap_bytes, s);
}

int32_t tmx_trace_commit_last_ms(tmx_ctx* c, float ms[3]) {
  if (!c || !ms) return TMX_ERR_BAD_ARG;
  if (!c->ev_commit[3]) return fail(c, TMX_ERR_BAD_ARG, "no tmx_trace_commit_device call yet");
  HIPCK(c, hipEventSynchronize(c->ev_commit[3]));
  for (int k = 0; k < 3; k++) HIPCK(c, hipEventElapsedTime(&ms[k], c->ev_commit[k], c->ev_commit[k + 1]));
  return TMX_OK;
}

}  // extern "C"
