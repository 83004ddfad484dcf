// The small-launch path (<= TINY_MAX_LANES validator lanes: one proof at any N, a handful at N <= 128) -- included by kernels.hip, inside
// namespace tmx, behind the helpers it reuses.
//
// A single skip proof is what the reference really asks for (one `verify_skip` per request: reference circuits/builder/verify.rs:528-563,
// bin/tendermintx.rs:171), and for it the classic launch graph is a relay: dedup -> phase 1 -> walk -> finish -> tail on the caller's stream
// with k_proof, the key pipeline and the serializer sections on three side streams -- 9+ launches, ~50 us of cross-stream hand-off and a
// 120-us one-workgroup k_proof for 128 signatures (profiles/r03_single_proof_timeline.txt).  Here the whole proof is TWO launches on the
// caller's stream and nothing else:
//
//   k_tiny       workgroups take ROLES by block index; none of them waits for another (no flags, no spinning):
//                  lane role      one workgroup (two waves) per validator lane: wave 1 probes the key cache and forms s*B with its sixteen
//                                 quads while thread 0 of wave 0 runs SHA-512 mod l; then wave 0 walks the key's table with sixteen quads
//                                 (two windows each + a four-level merge) and finishes in the limb-parallel form (one Fermat inversion for
//                                 the three Z's, ~35 us instead of the 56-us one-thread safegcd finish); a lane whose key is NOT resident
//                                 decodes it (wave 1, beside the hash) and takes the table-free h*A in the same workgroup; the classic
//                                 key pipeline (k_ed_dedup -> k_ed_keys -> tables -> k_kc_epilogue) runs BEHIND this kernel on the side
//                                 stream over a shadow copy of the lanes' keys: it inserts what was missed, for the next call
//                  tree role      one workgroup per (proof, validator set): marshal + leaf hashes, then the fixed-shape tree
//                  header role    both header trees, the inclusion proofs, the sign-bytes checks of every lane
//                  tally role     N x N match and the four prefix sums
//                  input role     the serializer spans that only expand the input records
//   k_tiny_tail  one workgroup per proof for thresholds / checks / verdict / the sections that carry them / the seam spans, and the
//                serializer spans of every section that needed k_tiny's results.
//
// Same values, bit for bit, as the classic path (tests: the whole -m gpu suite runs both, TMX_TINY=0 selects the classic one).
#pragma once

constexpr uint32_t TINY_THREADS = 128;
// Profiling build only (-DTMX_TINY_PROF, tools/build_variant.sh): TMX_TINY_DBG is a bit mask that switches roles / stages off so that the
// others can be timed by the kernel's events -- the outputs of such a run are wrong by construction.
#ifdef TMX_TINY_PROF
#define TINY_DBG(bit) (g_tiny_dbg & (bit))
__device__ uint32_t g_tiny_dbg;
#else
#define TINY_DBG(bit) false
#endif
// context-owned counters (zero between launches: the final role of k_tiny_tail resets what it reads)
enum : uint32_t { TN_WORDS = 8, TN_PER_PROOF = 16 };
// per-proof words behind TN_WORDS: [0] sign-bytes failures [1] varint sign-bit failures [2] u64 overflow in a tally

struct TinyEd {
  uint32_t n_lanes;
  const uint8_t* in_target;
  uint8_t* ed;
  uint32_t ed_stride;
  const int32_t* qtable;
  const int32_t* keytab;
  const uint32_t* keyrec;
  KeyCache kc;
  RowOut row;
  uint32_t* tiny;   // per-proof counters of the proof roles (zero between launches: k_tiny_tail resets what it reads)
  uint8_t* shadow;  // per lane a VR_STRIDE record of the context: key bytes + flags, read by the key pipeline behind this launch
};
struct TinyProof {
  ProofParams P;
  uint32_t n_proofs;
  const uint8_t *in_proof, *in_target, *in_trusted;
  uint8_t* lt;
  uint32_t lt_stride;
  uint8_t *lr, *pf, *nodes_t, *nodes_r, *reports;
  RowOut row;  // the lane's ten k_proof-derived D.1b elements go straight into the rows from the roles that compute them (rows = null: none)
};
struct TinySer {
  SerializeProgram S;  // resolved for this batch
  const uint32_t* lut;
  const uint8_t* wave_sec;
  const uint32_t* seam_waves;
  uint32_t n_seams;
  uint64_t* out;         // null: no rows
  uint32_t mask;         // sections this launch writes through its span roles
  uint32_t first_block;  // span block (4 spans) range [first_block, first_block + n_blocks) per proof
  uint32_t n_blocks;
  uint32_t tail_mask;    // k_tiny_tail: the sections that carry the verdict (their spans below tail_first_span go through the span roles)
  uint32_t tail_first_span;  // the first span of a row with an element that depends on the final checks: from here to the row end the
                             // per-proof final role expands element by element, behind its checks
};

// the sixteen quads of a wave hold partial sums of one point: after four rounds quad 0 holds the total
__device__ __forceinline__ fe quad_merge16(fe mine, int q) {
#pragma unroll 1
  for (int step = 1; step < 16; step *= 2) {
    const fe c = quad_to_cached(mine, q);
    fe other;
#pragma unroll
    for (int l = 0; l < 10; l++) other.v[l] = __shfl_down(c.v[l], 4 * step, 64);
    mine = quad_add(mine, other, q);
  }
  return mine;
}
// s*B of one lane by the sixteen quads of a wave (the table of B has NW = 20 windows at 13 bits: one or two per quad)
template <int W>
__device__ __forceinline__ fe tiny_base(const uint8_t* __restrict__ rec, const int32_t* __restrict__ qtable, int q, int part) {
  using C = BaseWin<W>;
  const bool is_signed = rec[VR_OFF_FLAGS] & 1;
  uint32_t s[8], e[C::NWORDS];
#pragma unroll
  for (int w = 0; w < 8; w++) s[w] = is_signed ? ld32(rec + VR_OFF_SIG + 32 + 4 * w) : K_DUMMY_SIG[8 + w];
  const uint32_t carry = recode_signed_w<W>(s, e);
  fe mine = fe_zero();
  mine.v[0] = (q == 1 || q == 2) ? 1 : 0;
  constexpr int PERP = (C::NW + 15) / 16;
  const int i0 = (C::NW * part + 15) / 16, i1 = (C::NW * (part + 1) + 15) / 16;
  fe ent[PERP];
  int dg[PERP];
#pragma unroll
  for (int k = 0; k < PERP; k++) {
    const int i = i0 + k;
    const int d = i < i1 ? key_digit_v<W>(e, i) : 0;
    dg[k] = d;
    const int m = d == 0 ? 0 : (d < 0 ? -d : d) - 1;
    const int32_t* src = qtable + ((size_t)((i < C::NW ? i : C::NW - 1) * C::HALF + m) * 4 + cached_component(q, d < 0)) * 10;
#pragma unroll
    for (int l = 0; l < 10; l++) ent[k].v[l] = src[l];
  }
#pragma unroll
  for (int k = 0; k < PERP; k++)
    if (dg[k] != 0) mine = quad_add(mine, ent[k], q, dg[k] < 0);
  if (carry && part == 0) {  // (no 256-bit scalar carries out of twenty 13-bit windows; kept for other widths)
    const int32_t* src = qtable + ((size_t)(C::ENTRIES - 1) * 4 + q) * 10;
    fe c;
#pragma unroll
    for (int l = 0; l < 10; l++) c.v[l] = src[l];
    mine = quad_add(mine, c, q);
  }
  return quad_merge16(mine, q);
}
// h*A of one lane from its key's table by the sixteen quads of a wave
template <int KW>
__device__ __forceinline__ fe tiny_walk(const uint32_t* h_lds, const int32_t* __restrict__ tab, int q, int part) {
  using C = KeyWin<KW>;
  uint32_t h[8], e[C::NWORDS];
#pragma unroll
  for (int w = 0; w < 8; w++) h[w] = h_lds[w];
  recode_signed_w<KW>(h, e);  // (h < 2^253: no carry out of the top window)
  fe mine = fe_zero();
  mine.v[0] = (q == 1 || q == 2) ? 1 : 0;
  constexpr int PERP = (C::NW + 15) / 16;
  const int i0 = (C::NW * part + 15) / 16, i1 = (C::NW * (part + 1) + 15) / 16;
  fe ent[PERP];
  int dg[PERP];
#pragma unroll
  for (int k = 0; k < PERP; k++) {
    const int i = i0 + k;
    const int d = i < i1 ? key_digit_v<KW>(e, i) : 0;
    dg[k] = d;
    const int m = d == 0 ? 0 : (d < 0 ? -d : d) - 1;
    const int32_t* src = tab + (size_t)((i < C::NW ? i : C::NW - 1) * C::HALF + m) * 40 + 10 * cached_component(q, d < 0);
#pragma unroll
    for (int l = 0; l < 10; l++) ent[k].v[l] = src[l];
  }
#pragma unroll
  for (int k = 0; k < PERP; k++)
    if (dg[k] != 0) mine = quad_add(mine, ent[k], q, dg[k] < 0);
  return quad_merge16(mine, q);
}

struct TinyLaneLds {
  int32_t pt[2][40];   // h*A, s*B: extended coordinates, ten limbs each (what the quad kernels keep in mulout)
  uint32_t h[8];
  uint32_t key[64];    // key record of a lane whose key is not resident (KEY_STRIDE / 4 words)
  uint32_t slot;
  uint32_t sh[6][64];
  uint64_t w[2][80];   // W_t + K_t of the two SHA-512 blocks
  uint32_t h16[6][8];  // canonical words of sB.x sB.y hA.x hA.y D.x D.y
};

// SHA-512(R | A | M) mod l of one lane by lanes 0 and 1 of a wave (blk = the calling lane): each assembles the sixteen message words of
// ITS block from aligned dword loads (the pad byte and the length by masks, no byte loop) and runs that block's message schedule -- the two
// schedules in lockstep, i.e. for the instructions of one -- leaving W_t + K_t in LDS; lane 0 then runs the 2 x 80 compression rounds
// alone.  Same digest / h as ed_role_hram (8.8 k instead of 11.7 k instructions on the lane's critical path).
__device__ __forceinline__ void tiny_hram(const uint8_t* __restrict__ rec, uint8_t* __restrict__ o, TinyLaneLds& L, uint32_t blk) {
  const bool is_signed = rec[VR_OFF_FLAGS] & 1;
  uint32_t mlen = is_signed ? (uint32_t)rec[VR_OFF_MLEN] | ((uint32_t)rec[VR_OFF_MLEN + 1] << 8) : 32u;
  if (mlen > 124u) mlen = 124u;
  const uint32_t total = 64 + mlen;
  const bool two_blocks = total + 17 > 128;
  uint64_t w[16];
#pragma unroll
  for (int k = 0; k < 16; k++) {
    uint32_t half[2];
#pragma unroll
    for (int hh = 0; hh < 2; hh++) {
      const int c = 2 * k + hh;                 // dword c of this lane's block = dword 32 blk + c of the message
      const uint32_t j = 32u * blk + (uint32_t)c;
      uint32_t data;
      if (c < 16) {                             // block 0: R | A from the signature / key (or the dummy pair); block 1: message dwords 16 ..
        const uint32_t mw = 16u + (uint32_t)c;  // (block 1 only: bytes 64 .. of M)
        const uint8_t* src = blk ? rec + VR_OFF_MSG + 4 * (mw < 31u ? mw : 0u) : (c < 8 ? rec + VR_OFF_SIG + 4 * c : rec + VR_OFF_PK + 4 * (c - 8));
        data = ld32(src);
        if (!is_signed) data = blk ? 0u : (c < 8 ? K_DUMMY_SIG[c] : K_DUMMY_PK[c < 8 ? 0 : c - 8]);
        if (blk && mw >= 31u) data = 0u;
      } else {                                  // message dword c - 16 (block 0) or c + 16 (block 1: beyond the 124 bytes)
        const uint32_t mw = (uint32_t)c - 16u;
        data = blk ? 0u : ld32(rec + VR_OFF_MSG + 4 * (mw < 31u ? mw : 0u));
        if (!is_signed || (!blk && mw >= 31u)) data = 0u;
      }
      const int32_t rem = (int32_t)total - (int32_t)(4u * j);  // message bytes left from this dword on
      const uint32_t keep = rem >= 4 ? 0xffffffffu : (rem <= 0 ? 0u : (1u << (8 * rem)) - 1u);
      const uint32_t mark = (rem >= 0 && rem < 4) ? 0x80u << (8 * rem) : 0u;
      half[hh] = __builtin_bswap32((data & keep) | mark);
    }
    w[k] = ((uint64_t)half[0] << 32) | half[1];
  }
  if (blk == (two_blocks ? 1u : 0u)) w[15] = (uint64_t)total * 8;
#pragma unroll
  for (int i = 0; i < 80; i++) {
    if (i >= 16) {
      const uint64_t w15 = w[(i + 1) & 15], w2 = w[(i + 14) & 15];
      const uint64_t s0 = rotr64(w15, 1) ^ rotr64(w15, 8) ^ (w15 >> 7);
      const uint64_t s1 = rotr64(w2, 19) ^ rotr64(w2, 61) ^ (w2 >> 6);
      w[i & 15] = w[i & 15] + s0 + w[(i + 9) & 15] + s1;
    }
    L.w[blk][i] = w[i & 15] + K_SHA512[i];
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  if (blk != 0) return;
  uint64_t dg[8];
  sha512_init(dg);
#pragma unroll 1
  for (uint32_t b = 0; b < (two_blocks ? 2u : 1u); b++) {
    uint64_t a = dg[0], bb = dg[1], c = dg[2], d = dg[3], e = dg[4], f = dg[5], g = dg[6], h = dg[7];
#pragma unroll
    for (int i = 0; i < 80; i++) {
      const uint64_t t1 = h + (rotr64(e, 14) ^ rotr64(e, 18) ^ rotr64(e, 41)) + ((e & f) ^ (~e & g)) + L.w[b][i];
      const uint64_t t2 = (rotr64(a, 28) ^ rotr64(a, 34) ^ rotr64(a, 39)) + ((a & bb) ^ (a & c) ^ (bb & c));
      h = g; g = f; f = e; e = d + t1; d = c; c = bb; bb = a; a = t1 + t2;
    }
    dg[0] += a; dg[1] += bb; dg[2] += c; dg[3] += d; dg[4] += e; dg[5] += f; dg[6] += g; dg[7] += h;
  }
  uint32_t dw[16], hs[8];
#pragma unroll
  for (int i = 0; i < 8; i++) { dw[2 * i] = bswap32((uint32_t)(dg[i] >> 32)); dw[2 * i + 1] = bswap32((uint32_t)dg[i]); }
  sc_reduce512(dw, hs);
#pragma unroll
  for (int w2 = 0; w2 < 16; w2++) st32(o + ED_OFF_DIGEST + 4 * w2, dw[w2]);
#pragma unroll
  for (int w2 = 0; w2 < 8; w2++) { st32(o + ED_OFF_H + 4 * w2, hs[w2]); L.h[w2] = hs[w2]; }
}

// fin_finish with the canonical words of the six coordinates already in LDS (L.h16) and h*A's projective rows in L.sh[3] (only the long way --
// a lane whose D does not encode to the signature's R bytes -- converts them): same record, same row elements as fin_finish.
__device__ __forceinline__ void tiny_finish_words(const uint32_t* __restrict__ kr, const uint8_t* __restrict__ rec, bool is_signed, TinyLaneLds& L,
                                                  uint8_t* __restrict__ o, uint64_t* __restrict__ row) {
  const bool okA = kr[KEY_OFF_OK / 4] != 0;
  uint32_t rb[8], s[8];
#pragma unroll
  for (int w = 0; w < 8; w++) {
    rb[w] = is_signed ? ld32(rec + VR_OFF_SIG + 4 * w) : K_DUMMY_SIG[w];
    s[w] = is_signed ? ld32(rec + VR_OFF_SIG + 32 + 4 * w) : K_DUMMY_SIG[8 + w];
  }
  uint32_t pts[6][8], rxy[2][8];
#pragma unroll
  for (int p = 0; p < 4; p++)
#pragma unroll
    for (int w = 0; w < 8; w++) pts[p][w] = L.h16[p][w];
#pragma unroll
  for (int w = 0; w < 8; w++) { rxy[0][w] = L.h16[4][w]; rxy[1][w] = L.h16[5][w]; }
  uint32_t mism = 0;
#pragma unroll
  for (int w = 0; w < 8; w++) mism |= rb[w] ^ (rxy[1][w] | (w == 7 ? (rxy[0][0] & 1u) << 31 : 0u));
  bool okR = true;
  uint32_t diff = 0;
  if (okA && mism != 0) {
    ge_ext hA, R;
    hA.X = f16_row_to_fe(&L.sh[3][0]); hA.Y = f16_row_to_fe(&L.sh[3][16]); hA.Z = f16_row_to_fe(&L.sh[3][32]); hA.T = f16_row_to_fe(&L.sh[3][48]);
    okR = ge_decode(rb, R);
    fe_to_words(R.X, rxy[0]);
    fe_to_words(R.Y, rxy[1]);
    ge_affc Rq;
    Rq.ypx = fe_add(R.Y, R.X); Rq.ymx = fe_sub(R.Y, R.X); Rq.xy2d = fe_mul(R.T, K_2D);
    const ge_proj sum = comp_to_proj(ge_add_affc(hA, Rq));
    const fe zi_sum = fe_invert_safegcd(sum.Z);
    fe_to_words(fe_mul(sum.X, zi_sum), pts[4]); fe_to_words(fe_mul(sum.Y, zi_sum), pts[5]);
#pragma unroll
    for (int w = 0; w < 8; w++) diff |= (pts[0][w] ^ pts[4][w]) | (pts[1][w] ^ pts[5][w]);
  } else {
#pragma unroll
    for (int w = 0; w < 8; w++) { pts[4][w] = pts[0][w]; pts[5][w] = pts[1][w]; }
  }
  const bool decode_ok = okA && okR;
  const bool ok = decode_ok && diff == 0 && sc_is_canonical(s);
#pragma unroll
  for (int w = 0; w < 16; w++) st32(o + ED_OFF_PTS + 4 * w, decode_ok ? kr[KEY_OFF_XY / 4 + w] : 0u);
#pragma unroll
  for (int w = 0; w < 8; w++) {
    st32(o + ED_OFF_PTS + 64 + 4 * w, decode_ok ? rxy[0][w] : 0u);
    st32(o + ED_OFF_PTS + 96 + 4 * w, decode_ok ? rxy[1][w] : 0u);
  }
#pragma unroll
  for (int p = 0; p < 6; p++)
#pragma unroll
    for (int w = 0; w < 8; w++) st32(o + ED_OFF_PTS + 128 + 32 * p + 4 * w, decode_ok ? pts[p][w] : 0u);
  st32(o + ED_OFF_OK, ok ? 1u : 0u);
  st32(o + ED_OFF_DECODE_OK, decode_ok ? 1u : 0u);
#pragma unroll
  for (int w = 0; w < 6; w++) st32(o + 424 + 4 * w, 0u);
  if (row) {
#pragma unroll
    for (int w = 0; w < 8; w++) row[w] = L.h[w];
#pragma unroll
    for (int w = 0; w < 16; w++) row[8 + w] = decode_ok ? kr[KEY_OFF_XY / 4 + w] : 0u;
#pragma unroll
    for (int w = 0; w < 8; w++) { row[24 + w] = decode_ok ? rxy[0][w] : 0u; row[32 + w] = decode_ok ? rxy[1][w] : 0u; }
#pragma unroll
    for (int p = 0; p < 6; p++)
#pragma unroll
      for (int w = 0; w < 8; w++) row[40 + 8 * p + w] = decode_ok ? pts[p][w] : 0u;
    row[88] = ok ? 1u : 0u;
  }
}

// One validator lane, one workgroup of two waves:
//   wave 1   probe the key cache | key not resident: decode it (k_ed_keys' limb-parallel form) | s*B with sixteen quads
//   wave 0   SHA-512 mod l (thread 0) | h*A: walk of the key's table with sixteen quads, or the table-free chain (k_ed_mul16's) | finish
// The cache is only read here (and the lane's key bytes copied to the context's shadow records): the classic key pipeline runs behind
// this kernel on the side stream over those shadows -- it inserts the keys this launch missed and builds their tables for the next call.
template <int KW, int BW>
__device__ __forceinline__ void tiny_lane(const TinyEd& A, uint32_t lane, TinyLaneLds& L) {
  const uint32_t tid = threadIdx.x;
  const uint8_t* rec = A.in_target + (size_t)lane * VR_STRIDE;
  uint8_t* o = A.ed + (size_t)lane * A.ed_stride;
  const KeyCache& kc = A.kc;
  const f16::Ctx c = f16::make_ctx((int)(tid & 63u));
  if (tid >= 64) {
    const int t1 = (int)tid - 64;
    uint32_t slot = DEDUP_EMPTY;
    uint32_t pk[8];
    effective_pk(rec, pk);
    if (t1 == 0) {
      uint32_t ps = pk_hash(pk) & kc.hash_mask;
      for (;;) {
        const uint32_t e = kc.d_hash[ps];
        if (e == DEDUP_EMPTY) break;
        const uint32_t* kp = kc.d_pk + (size_t)e * 8;
        uint32_t d = 0;
#pragma unroll
        for (int w = 0; w < 8; w++) d |= kp[w] ^ pk[w];
        if (d == 0) { slot = e; break; }
        ps = (ps + 1) & kc.hash_mask;
      }
      L.slot = slot;
      uint8_t* sh = A.shadow + (size_t)lane * VR_STRIDE;  // what effective_pk() reads: the key bytes and the word with the flags byte
#pragma unroll
      for (int w = 0; w < 8; w++) st32(sh + VR_OFF_PK + 4 * w, ld32(rec + VR_OFF_PK + 4 * w));
      st32(sh + VR_OFF_MLEN, ld32(rec + VR_OFF_MLEN));
    }
    slot = (uint32_t)__shfl((int)slot, 0);
    if (slot == DEDUP_EMPTY) {  // RFC 8032 5.1.3 as in ge_decode: x = u v^3 (u v^7)^((p-5)/8), u = y^2 - 1, v = d y^2 + 1 (all four rows on the one key)
      uint32_t word = pk[0];
#pragma unroll
      for (int w = 1; w < 8; w++) word = (c.k >> 1) == (uint32_t)w ? pk[w] : word;
      uint32_t y = (c.k & 1) ? word >> 16 : word & 0xffffu;
      if (c.k == 15) y &= 0x7fffu;
      const uint32_t one = c.k == 0 ? 1u : 0u;
      const uint32_t yy = f16::mul(y, y, c);
      const uint32_t uu = f16::carry(yy + c.b4 - one, c);
      const uint32_t v = f16::mul(yy, f16::const_d(c.k), c) + one;
      const uint32_t v3 = f16::mul(f16::mul(v, v, c), v, c);
      const uint32_t uv7 = f16::mul(f16::mul(f16::mul(v3, v3, c), v, c), uu, c);
      const uint32_t x = f16::mul(f16::mul(f16::pow_p58<true>(uv7, c), v3, c), uu, c);  // (one key in all four rows: they share the chain)
      const uint32_t xi = f16::mul(x, f16::const_sqrtm1(c.k), c);
      L.sh[0][t1] = x; L.sh[1][t1] = xi; L.sh[2][t1] = f16::mul(f16::mul(x, x, c), v, c); L.sh[3][t1] = uu;
      L.sh[4][t1] = f16::mul(x, y, c); L.sh[5][t1] = f16::mul(xi, y, c);
    }
  } else if (tid < 2) {
    tiny_hram(rec, o, L, tid);
  }
  __syncthreads();  // B1: h, the slot, (miss) the decode's products
  if (TINY_DBG(0x100)) return;
  const uint32_t slot = L.slot;
  const bool hit = slot != DEDUP_EMPTY;
  const uint32_t* kr = hit ? A.keyrec + (size_t)slot * (KEY_STRIDE / 4) : L.key;
  uint32_t hA16 = 0;
  if (!hit) {
    if (tid == 64) {  // the tail of ge_decode in the ten-limb form: no multiplications left
      uint32_t pk[8];
      effective_pk(rec, pk);
      const fe vxx = f16_row_to_fe(&L.sh[2][0]), U = f16_row_to_fe(&L.sh[3][0]);
      const bool ok_direct = fe_is_zero(fe_sub(vxx, U)), ok_flipped = fe_is_zero(fe_add(vxx, U));
      fe X = fe_select(f16_row_to_fe(&L.sh[0][0]), f16_row_to_fe(&L.sh[1][0]), !ok_direct);
      fe T = fe_select(f16_row_to_fe(&L.sh[4][0]), f16_row_to_fe(&L.sh[5][0]), !ok_direct);
      const bool sign = (pk[7] >> 31) & 1;
      uint32_t xw[8];
      fe_to_words(X, xw);
      const bool x_zero = words_is_zero(xw), flip = (bool)(xw[0] & 1) != sign;
      X = fe_select(X, fe_neg(X), flip);
      T = fe_select(T, fe_neg(T), flip);
      fe_to_words(X, xw);
      key_store(L.key, X, fe_carry32(fe_from_words(pk)), T, xw, (ok_direct || ok_flipped) && !(x_zero && sign));
    }
    __syncthreads();  // B2 (miss): the key record
  }
  if (tid >= 64) {
    const int t1 = (int)tid - 64, q = t1 & 3, part = t1 >> 2;
    const fe sB = tiny_base<BW>(rec, A.qtable, q, part);
    if (part == 0) {
#pragma unroll
      for (int i = 0; i < 10; i++) L.pt[1][10 * q + i] = sB.v[i];
    }
  } else if (hit) {
    const int q = (int)tid & 3, part = (int)tid >> 2;
    const fe hA = tiny_walk<KW>(L.h, A.keytab + (size_t)slot * KeyWin<KW>::NW * KeyWin<KW>::HALF * 40, q, part);
    if (part == 0) {
#pragma unroll
      for (int i = 0; i < 10; i++) L.pt[0][10 * q + i] = hA.v[i];
    }
  } else {  // the table-free h*A (k_ed_mul16's chain: 252 doublings + 64 additions in the limb-parallel form)
    uint32_t h[8], e[8];
#pragma unroll
    for (int w = 0; w < 8; w++) h[w] = L.h[w];
    recode_signed16(h, e);
    const u32x8 ev = vec8_of(e);
    const uint32_t Ak = f16::from_limbs10(reinterpret_cast<const int32_t*>(L.key + KEY_OFF_A / 4 + 10 * c.row), c);
    uint32_t tab[8], Q;
    tab[0] = f16::to_cached(Ak, c);
    f16::dbl(f16::with_xy(Ak, c), Q, c);
    tab[1] = f16::to_cached(Q, c);
#pragma unroll
    for (int j = 2; j < 8; j++) {
      Q = f16::add_cached(Q, tab[0], c);
      tab[j] = f16::to_cached(Q, c);
    }
    const u32x8 tabv = vec8_of(tab);
    uint32_t acc = (c.k == 0 && (c.row == 1 || c.row == 2)) ? 1u : 0u;
#pragma unroll 1
    for (int i = 63; i >= 0; i--) {
      if (i != 63) {
        uint32_t V = f16::with_xy(acc, c);
        V = f16::dbl(V, acc, c); V = f16::dbl(V, acc, c); V = f16::dbl(V, acc, c); f16::dbl(V, acc, c);
      }
      const int d = __builtin_amdgcn_readfirstlane(digit_at(ev, i));
      if (d != 0) {
        const int m = (d < 0 ? -d : d) - 1;
        uint32_t qq = tabv[m & 7];
        if (d < 0) qq = f16::neg_cached(qq, c);
        acc = f16::add_cached(acc, qq, c);
      }
    }
    hA16 = acc;
  }
  __syncthreads();  // B3: s*B (and the walk's h*A) in LDS
  if (TINY_DBG(0x200)) return;
  // ---- finish in the limb-parallel form: D = s*B - h*A, ONE inversion for the three Z's, the six affine coordinates; thread 0 then runs
  // the scalar second half (comparison with the signature's R bytes, the lane's record and its D.1b elements)
  if (tid < 64) {
    if (hit) hA16 = f16::from_limbs10(&L.pt[0][10 * c.row], c);
    const uint32_t sB = f16::from_limbs10(&L.pt[1][10 * c.row], c);
    const uint32_t D = f16::add_cached(sB, f16::neg_cached(f16::to_cached(hA16, c), c), c);
    const uint32_t zS = f16::rows(sB).r2, zH = f16::rows(hA16).r2, zD = f16::rows(D).r2;
    const uint32_t p12 = f16::mul(zS, zH, c);
    const uint32_t inv = f16::invert<true>(f16::mul1(p12, zD, c), c);  // (one element in all four rows: the rows share its products)
    const uint32_t inv12 = f16::mul(inv, zD, c);
    L.sh[0][tid] = f16::mul(sB, f16::mul(inv12, zH, c), c);
    L.sh[1][tid] = f16::mul(hA16, f16::mul(inv12, zS, c), c);
    L.sh[2][tid] = f16::mul(D, f16::mul(inv, p12, c), c);
    L.sh[3][tid] = hA16;
  }
  __syncthreads();
  if (TINY_DBG(0x400)) return;
  // the six affine coordinates (s*B, h*A, D) made canonical by six threads at once -- the same instructions on six rows -- instead of one
  // after the other by thread 0 (8 -> ~4 us of a lane's chain); thread 0 then compares D with the signature's R bytes and writes the lane
  if (tid < 6) {
    uint32_t w[8];
    fe_to_words(f16_row_to_fe(&L.sh[tid >> 1][16 * (tid & 1)]), w);
#pragma unroll
    for (int k = 0; k < 8; k++) L.h16[tid][k] = w[k];
  }
  __syncthreads();
  if (tid == 0) tiny_finish_words(kr, rec, rec[VR_OFF_FLAGS] & 1, L, o, row_d1b(A.row, lane));
}

// ------------------------------------------------------------------------------------------------ proof roles
// (each is a part of proof_body that depends on the input records only; the values and the buffers they go to are proof_body's)

// marshal + leaf hashes of one validator set, then its fixed-shape tree (proof_body phases 1b, 3d)
__device__ __forceinline__ void tiny_tree(const TinyProof& A, uint32_t p, uint32_t set, uint32_t* tp, uint32_t (*s_leaf)[8]) {
  const uint32_t t = threadIdx.x, n = A.P.n;
  const uint8_t* pr = A.in_proof + (size_t)p * PR_STRIDE;
  const uint32_t nbs = ld32(pr + (set ? PR_OFF_NB_B : PR_OFF_NB_A));
  uint8_t* nodes = (set ? A.nodes_r : A.nodes_t) + (size_t)p * A.P.tree_nodes * 32;
  uint32_t bad = 0;
  for (uint32_t i = t; i < n; i += blockDim.x) {
    const uint8_t* rec = set ? A.in_trusted + ((size_t)p * n + i) * HR_STRIDE : A.in_target + ((size_t)p * n + i) * VR_STRIDE;
    uint8_t* o = set ? A.lr + ((size_t)p * n + i) * LANE_STRIDE : A.lt + ((size_t)p * n + i) * A.lt_stride;
    if (ld64(rec + (set ? HR_OFF_POWER : VR_OFF_POWER)) >> 63) bad++;  // marshal_int64_varint asserts bit 63 == 0 (shared.rs:80)
    uint32_t dig[8], pkw[8];
    if (A.P.leaves_done) {  // (k_leaves has run: very large batches serialize the leaf sections early)
#pragma unroll
      for (int k = 0; k < 8; k++) dig[k] = bswap32(ld32(o + LN_OFF_LEAF + 4 * k));
    } else {
      marshal_and_leaf(rec, set != 0, o, dig, pkw);
    }
#pragma unroll
    for (int k = 0; k < 8; k++) s_leaf[i][k] = dig[k];
  }
  if (bad) atomicAdd(&tp[1], bad);
  __syncthreads();
  uint32_t sz = n, lvl = 0, out_off = 0;
  while (sz > 1) {
    const uint32_t nx = (sz + 1) / 2;
    uint32_t dig[2][8];
    int cnt = 0;
    for (uint32_t i = t; i < nx; i += blockDim.x, cnt++) {
      const bool both = (2 * i + 1 < sz) && ((((uint64_t)(2 * i + 1)) << lvl) < nbs);
      uint32_t lw[8], rw[8];
#pragma unroll
      for (int k = 0; k < 8; k++) { lw[k] = s_leaf[2 * i][k]; rw[k] = (2 * i + 1 < sz) ? s_leaf[2 * i + 1][k] : 0u; }
      if (both) sha256_inner(lw, rw, dig[cnt]);
      else {
#pragma unroll
        for (int k = 0; k < 8; k++) dig[cnt][k] = lw[k];
      }
      store_digest(nodes + (size_t)(out_off + i) * 32, dig[cnt]);
    }
    __syncthreads();
    cnt = 0;
    for (uint32_t i = t; i < nx; i += blockDim.x, cnt++) {
#pragma unroll
      for (int k = 0; k < 8; k++) s_leaf[i][k] = dig[cnt][k];
    }
    __syncthreads();
    out_off += nx; sz = nx; lvl++;
  }
}

// both header trees, the inclusion proofs, the header-derived fields of the proof record and the sign-bytes checks of every target lane
// (proof_body phases 1a, 2, 4, 3a and the header part of 5).  Owns bytes [0, PF_OFF_TALLY_T) and [PF_OFF_HEIGHT, PF_STRIDE) of the record.
__device__ __forceinline__ void tiny_header(const TinyProof& A, uint32_t p, uint32_t* tp, uint32_t (*s_hdr)[28][8]) {
  const ProofParams& P = A.P;
  const int t = threadIdx.x;
  const uint32_t n = P.n;
  const bool skip = P.kind == 0;
  const uint8_t* pr = A.in_proof + (size_t)p * PR_STRIDE;
  const uint8_t* tg = A.in_target + (size_t)p * n * VR_STRIDE;
  uint8_t* ltp = A.lt + (size_t)p * n * A.lt_stride;
  uint8_t* pf = A.pf + (size_t)p * PF_STRIDE;
  const uint64_t block_b = ld64(pr + PR_OFF_BLOCK_B), round_ = ld64(pr + PR_OFF_ROUND);
  const uint32_t nb = ld32(pr + PR_OFF_NB_A), nbt = ld32(pr + PR_OFF_NB_B);
  for (uint32_t i = t; i < PF_STRIDE / 4; i += blockDim.x)
    if (4 * i < PF_OFF_TALLY_T || 4 * i >= PF_OFF_HEIGHT) st32(pf + 4 * i, 0u);
  if (t < 28) {
    const int hsel = t / 14, li = t % 14;
    const uint8_t* hrec = pr + (hsel ? PR_OFF_HDR_B : PR_OFF_HDR_A);
    uint32_t len = hrec[li];
    if (len > 79) len = 79;
    const uint8_t* leaf = hrec + 16 + 80 * li;
    uint32_t fw0[20], dig[8];  // (twenty word loads masked to the field's length: proof_body phase 1a)
#pragma unroll
    for (int w = 0; w < 20; w++) fw0[w] = ld32(leaf + 4 * w) & field_keep(len, w);
    sha256_leaf80(fw0, len, dig);
#pragma unroll
    for (int k = 0; k < 8; k++) s_hdr[hsel][li][k] = dig[k];
  }
  __syncthreads();
  for (int lvl = 0; lvl < 4; lvl++) {  // pairwise with promotion of an odd last node == RFC 6962 split (tendermint_utils.rs:316-349)
    const int sz = HDR_LEVEL_SZ[lvl], nx = HDR_LEVEL_SZ[lvl + 1];
    if (t < 2 * nx) {
      const int hsel = t / nx, i = t % nx;
      const uint32_t* l = s_hdr[hsel][HDR_LEVEL_OFF[lvl] + 2 * i];
      uint32_t lw[8], rw[8], dig[8];
      const bool has_r = 2 * i + 1 < sz;
#pragma unroll
      for (int k = 0; k < 8; k++) { lw[k] = l[k]; rw[k] = has_r ? s_hdr[hsel][HDR_LEVEL_OFF[lvl] + 2 * i + 1][k] : 0u; }
      if (has_r) sha256_inner(lw, rw, dig);
      else {
#pragma unroll
        for (int k = 0; k < 8; k++) dig[k] = lw[k];
      }
#pragma unroll
      for (int k = 0; k < 8; k++) s_hdr[hsel][HDR_LEVEL_OFF[lvl + 1] + i][k] = dig[k];
    }
    __syncthreads();
  }
  uint32_t hdr_hash[8];
#pragma unroll
  for (int k = 0; k < 8; k++) hdr_hash[k] = s_hdr[0][27][k];
  const uint8_t* hdr_a = pr + PR_OFF_HDR_A;
  const uint8_t* hdr_b = pr + PR_OFF_HDR_B;
  // ---- the inclusion proofs (proof_body phase 4, same code)
  const uint32_t n_incl = skip ? 4 : 5;
  uint64_t height_a = 0;
  {
    uint32_t hl = hdr_a[2];
    if (hl > 79) hl = 79;
    const uint32_t hw0 = ld32(hdr_a + 16 + 80 * 2), hw1 = ld32(hdr_a + 16 + 80 * 2 + 4), hw2 = ld32(hdr_a + 16 + 80 * 2 + 8);
#pragma unroll
    for (uint32_t k = 1; k <= 10; k++) {  // (bytes 1 .. 10 of the height field from three word loads: proof_body phase 4)
      const uint32_t word = k < 4 ? hw0 : (k < 8 ? hw1 : hw2), byte = (word >> (8 * (k & 3))) & 0x7fu;
      if (k < hl) height_a |= (uint64_t)byte << (7 * (k - 1));
    }
  }
  if (t < (int)n_incl) {
    const int q = t;
    const int hsel = (q == 3 && skip) || q == 4 ? 1 : 0;
    const int index = q == 0 ? 1 : q == 1 ? 2 : q == 2 ? 7 : q == 3 ? (skip ? 7 : 4) : 8;
    const uint8_t* hrec = hsel ? hdr_b : hdr_a;
    const bool is_height = q == 1;
    const uint32_t fixed = q == 0 ? 52u : (q == 3 && !skip) ? 72u : 34u;
    uint32_t flen = hrec[index];
    if (flen > 79) flen = 79;
    const uint32_t take = is_height ? 0u : (flen > fixed ? fixed : flen);
    const uint32_t blen = (q == 0 || is_height) ? flen : fixed;
    const uint8_t* fsrc = hrec + 16 + 80 * index;
    uint32_t fw[20];
#pragma unroll
    for (int w = 0; w < 20; w++) {
      const uint32_t v = ld32(fsrc + 4 * w);
      const uint32_t keep = take >= 4u * w + 4 ? 0xffffffffu : (take > 4u * w ? (1u << (8 * (take - 4u * w))) - 1u : 0u);
      fw[w] = v & keep;
    }
    if (is_height) {
      int last = 0;
#pragma unroll
      for (int s9 = 0; s9 < 9; s9++) if ((height_a >> (7 * s9)) & 0x7f) last = s9;
      uint32_t hb[12];
      hb[0] = 0x08; hb[10] = 0; hb[11] = 0;
#pragma unroll
      for (int s9 = 0; s9 < 9; s9++) hb[1 + s9] = (uint32_t)(((height_a >> (7 * s9)) & 0x7f) | (s9 < last ? 0x80 : 0));
#pragma unroll
      for (int w = 0; w < 3; w++) {
        const uint32_t keep = flen >= 4u * w + 4 ? 0xffffffffu : (flen > 4u * w ? (1u << (8 * (flen - 4u * w))) - 1u : 0u);
        fw[w] = (hb[4 * w] | (hb[4 * w + 1] << 8) | (hb[4 * w + 2] << 16) | (hb[4 * w + 3] << 24)) & keep;
      }
      pf[PF_OFF_HLEAF] = 0;
#pragma unroll
      for (int k = 0; k < 10; k++) pf[PF_OFF_HLEAF + 1 + k] = (uint8_t)hb[k];
      if (height_a >> 63) atomicAdd(&tp[1], 1u);
    } else {
      const uint32_t dst = q == 0 ? PF_OFF_CID52 : q == 2 ? PF_OFF_LEAFV : q == 3 ? PF_OFF_LEAFX : PF_OFF_LEAFY;
#pragma unroll
      for (int w = 0; w < 18; w++) {
        if (4u * w + 4 <= fixed) st32(pf + dst + 4 * w, fw[w]);
        else if (4u * w < fixed) { pf[dst + 4 * w] = (uint8_t)fw[w]; pf[dst + 4 * w + 1] = (uint8_t)(fw[w] >> 8); }
      }
    }
    bool same_leaf = blen == flen && (is_height || take == flen);
#pragma unroll
    for (int w = 0; w < 20; w++) {
      const uint32_t keep = flen >= 4u * w + 4 ? 0xffffffffu : (flen > 4u * w ? (1u << (8 * (flen - 4u * w))) - 1u : 0u);
      same_leaf = same_leaf && (ld32(fsrc + 4 * w) & keep) == (fw[w] & keep);
    }
    uint32_t cur[8];
    if (same_leaf) {
#pragma unroll
      for (int k = 0; k < 8; k++) cur[k] = s_hdr[hsel][HDR_LEVEL_OFF[0] + index][k];
    } else {
      sha256_leaf80(fw, blen, cur);
    }
    store_digest(pf + PF_OFF_PROOFD + 160 * q, cur);
    bool on_tree = true;
#pragma unroll
    for (int k = 0; k < 8; k++) on_tree = on_tree && cur[k] == s_hdr[hsel][HDR_LEVEL_OFF[0] + index][k];
    for (int lvl = 0; lvl < 4; lvl++) {
      const int node = index >> lvl;
      uint32_t aunt[8], nxt[8];
#pragma unroll
      for (int k = 0; k < 8; k++) aunt[k] = s_hdr[hsel][HDR_LEVEL_OFF[lvl] + (node ^ 1)][k];
      store_digest(pf + PF_OFF_AUNTS + 128 * q + 32 * lvl, aunt);
      if (on_tree) {
#pragma unroll
        for (int k = 0; k < 8; k++) nxt[k] = s_hdr[hsel][HDR_LEVEL_OFF[lvl + 1] + (node >> 1)][k];
      } else {
        const bool right = node & 1;
        uint32_t lw[8], rw[8];
#pragma unroll
        for (int k = 0; k < 8; k++) { lw[k] = right ? aunt[k] : cur[k]; rw[k] = right ? cur[k] : aunt[k]; }
        sha256_inner(lw, rw, nxt);
      }
#pragma unroll
      for (int k = 0; k < 8; k++) cur[k] = nxt[k];
      store_digest(pf + PF_OFF_PROOFD + 160 * q + 32 + 32 * lvl, cur);
    }
  }
  if (t == 64) {  // (a lane of the other wave: the five proof lanes above are busy)
    store_digest(pf + PF_OFF_HEADER, hdr_hash);
    st64(pf + PF_OFF_HEIGHT, height_a);
    st32(pf + PF_OFF_CIDLEN, hdr_a[1]);
    st32(pf + PF_OFF_HLEN, hdr_a[2]);
    st32(pf + PF_OFF_NB_A, nb); st32(pf + PF_OFF_NB_B, nbt); st64(pf + PF_OFF_ROUND, round_);
  }
  // ---- sign-bytes checks per target lane (proof_body phase 3a; validator.rs:80-183)
  uint32_t fails = 0;
  for (uint32_t i = t; i < n; i += blockDim.x) {
    const uint8_t* rec = tg + (size_t)i * VR_STRIDE;
    const bool sgn = rec[VR_OFF_FLAGS] & 1, enabled = i < nb;
    // (the first 60 sign-bytes as fifteen word loads, the 32-byte comparison word-wise at both possible offsets: proof_body phase 3a)
    uint32_t mw[15];
#pragma unroll
    for (int w = 0; w < 15; w++) mw[w] = ld32(rec + VR_OFF_MSG + 4 * w);
    auto word_at = [&](int q) -> uint32_t { return (q & 3) ? __builtin_amdgcn_alignbyte(mw[(q >> 2) + 1], mw[q >> 2], q & 3) : mw[q >> 2]; };
    uint32_t d16 = 0, d25 = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const uint32_t want = bswap32(hdr_hash[k]);
      d16 |= word_at(16 + 4 * k) ^ want; d25 |= word_at(25 + 4 * k) ^ want;
    }
    const bool hash_in_msg = (round_ == 0 ? d16 : d25) == 0;
    const bool is_precommit = (mw[0] & 0x00ffff00u) == 0x00020800u;
    const uint64_t mh = (uint64_t)mw[1] | ((uint64_t)mw[2] << 32);
    const uint64_t mr = (uint64_t)word_at(13) | ((uint64_t)word_at(17) << 32);
    const bool height_ok = mh == block_b;
    const bool round_ok = round_ == 0 ? true : (mr == round_);
    const bool valid = sgn && enabled && hash_in_msg && is_precommit && height_ok && round_ok;
    const bool sigdata_ok = sgn == valid;
    uint8_t* o = ltp + (size_t)i * A.lt_stride + LN_OFF_FLAGS;
    st32(o, (enabled ? 1u : 0u) | (hash_in_msg ? 1u << 8 : 0u) | (is_precommit ? 1u << 16 : 0u) | (height_ok ? 1u << 24 : 0u));
    st32(o + 4, (round_ok ? 1u : 0u) | (sigdata_ok ? 1u << 8 : 0u));
    if (uint64_t* row = row_d1b(A.row, p * n + i)) {  // D.1b elements 89 .. 94: the six flags (k_verdict writes them on the classic path)
      row[D1B_ED_ELEMS + 0] = enabled; row[D1B_ED_ELEMS + 1] = hash_in_msg; row[D1B_ED_ELEMS + 2] = is_precommit; row[D1B_ED_ELEMS + 3] = height_ok;
      row[D1B_ED_ELEMS + 4] = round_ok; row[D1B_ED_ELEMS + 5] = sigdata_ok;
    }
    if (!sigdata_ok) fails++;
  }
  if (fails) atomicAdd(&tp[0], fails);
}

// N x N pubkey match and the prefix-sum tallies (proof_body phases 3b, 3c and the tally part of 5).  Owns [PF_OFF_TALLY_T, PF_OFF_VERDICTS).
struct TinyTallyLds {
  uint32_t pk[TMX_N_LIMIT][8];
  uint8_t sgn[TMX_N_LIMIT];
  uint8_t matched[TMX_N_LIMIT];
  u96 wave_tot[PROOF_THREADS_WIDE / 64];
  uint32_t htab[2 * TMX_N_LIMIT];  // open-addressing table of the target lanes that signed (lane index, or 0xffffffff)
};
__device__ __forceinline__ void tiny_tally(const TinyProof& A, uint32_t p, uint32_t* tp, TinyTallyLds& L) {
  const ProofParams& P = A.P;
  const uint32_t t = threadIdx.x, n = P.n;
  const bool skip = P.kind == 0;
  const uint8_t* pr = A.in_proof + (size_t)p * PR_STRIDE;
  const uint8_t* tg = A.in_target + (size_t)p * n * VR_STRIDE;
  const uint8_t* tr = skip ? A.in_trusted + (size_t)p * n * HR_STRIDE : nullptr;
  uint8_t* ltp = A.lt + (size_t)p * n * A.lt_stride;
  uint8_t* lrp = A.lr + (size_t)p * n * LANE_STRIDE;
  uint8_t* pf = A.pf + (size_t)p * PF_STRIDE;
  const uint32_t nb = ld32(pr + PR_OFF_NB_A), nbt = ld32(pr + PR_OFF_NB_B);
  for (uint32_t i = t; i < n; i += blockDim.x) {
#pragma unroll
    for (int w = 0; w < 8; w++) L.pk[i][w] = ld32(tg + (size_t)i * VR_STRIDE + 4 * w);
    L.sgn[i] = tg[(size_t)i * VR_STRIDE + VR_OFF_FLAGS] & 1;
  }
  __syncthreads();
  if (skip) {
    // flag[j] = OR_i signed[i] && pk_target[i] == pk_trusted[j] (verify.rs:398-418).  Not as N x N comparisons (164 us of this role at
    // N = 512): the target lanes that signed go into a hash table in LDS -- every one of them, duplicates of a key in slots of their own --
    // and a trusted lane probes its key's run, comparing all 32 bytes with every entry it meets: the same OR, in O(N).
    for (uint32_t h = t; h < 2 * TMX_N_LIMIT; h += blockDim.x) L.htab[h] = DEDUP_EMPTY;
    __syncthreads();
    for (uint32_t i = t; i < n; i += blockDim.x) {
      if (!L.sgn[i]) continue;
      uint32_t h = pk_hash(L.pk[i]) & (2 * TMX_N_LIMIT - 1);
      while (atomicCAS(&L.htab[h], DEDUP_EMPTY, i) != DEDUP_EMPTY) h = (h + 1) & (2 * TMX_N_LIMIT - 1);
    }
    __syncthreads();
    for (uint32_t j = t; j < n; j += blockDim.x) {
      uint32_t pkr[8];
#pragma unroll
      for (int w = 0; w < 8; w++) pkr[w] = ld32(tr + (size_t)j * HR_STRIDE + 4 * w);
      bool m = false;
      for (uint32_t h = pk_hash(pkr) & (2 * TMX_N_LIMIT - 1);; h = (h + 1) & (2 * TMX_N_LIMIT - 1)) {
        const uint32_t i = L.htab[h];
        if (i == DEDUP_EMPTY) break;  // (at most N of the 2 N slots are taken: every run ends)
        uint32_t d = 0;
#pragma unroll
        for (int w = 0; w < 8; w++) d |= L.pk[i][w] ^ pkr[w];
        m = m || d == 0;
      }
      L.matched[j] = m;
      st32(lrp + (size_t)j * LANE_STRIDE + LN_OFF_FLAGS, (j < nbt ? 1u : 0u) | (m ? 1u << 8 : 0u));
      st32(lrp + (size_t)j * LANE_STRIDE + LN_OFF_FLAGS + 4, 0u);
    }
  }
  __syncthreads();
  const u96 t_total = block_prefix_sums(n, [&](uint32_t i) { return i < nb ? ld64(tg + (size_t)i * VR_STRIDE + VR_OFF_POWER) : 0ull; },
                                        [&](uint32_t i, uint64_t v) {
                                          st64(ltp + (size_t)i * A.lt_stride + LN_OFF_TOT, v);
                                          if (uint64_t* row = row_d1b(A.row, p * n + i)) { row[D1B_ED_ELEMS + 6] = (uint32_t)v; row[D1B_ED_ELEMS + 7] = v >> 32; }
                                        }, L.wave_tot);
  const u96 t_acc = block_prefix_sums(n, [&](uint32_t i) { return L.sgn[i] ? ld64(tg + (size_t)i * VR_STRIDE + VR_OFF_POWER) : 0ull; },
                                      [&](uint32_t i, uint64_t v) {
                                        st64(ltp + (size_t)i * A.lt_stride + LN_OFF_ACC, v);
                                        if (uint64_t* row = row_d1b(A.row, p * n + i)) { row[D1B_ED_ELEMS + 8] = (uint32_t)v; row[D1B_ED_ELEMS + 9] = v >> 32; }
                                      }, L.wave_tot);
  u96 r_total = {0, 0}, r_acc = {0, 0};
  if (skip) {
    r_total = block_prefix_sums(n, [&](uint32_t j) { return j < nbt ? ld64(tr + (size_t)j * HR_STRIDE + HR_OFF_POWER) : 0ull; },
                                [&](uint32_t j, uint64_t v) { st64(lrp + (size_t)j * LANE_STRIDE + LN_OFF_TOT, v); }, L.wave_tot);
    r_acc = block_prefix_sums(n, [&](uint32_t j) { return L.matched[j] ? ld64(tr + (size_t)j * HR_STRIDE + HR_OFF_POWER) : 0ull; },
                              [&](uint32_t j, uint64_t v) { st64(lrp + (size_t)j * LANE_STRIDE + LN_OFF_ACC, v); }, L.wave_tot);
  }
  if (t == 0) {
    bool no_overflow = t_total.hi == 0 && t_acc.hi == 0 && r_total.hi == 0 && r_acc.hi == 0;
    const uint64_t t_sa = t_acc.lo * 3ull, t_st = t_total.lo * 2ull;  // voting.rs:91-108: wrapping multiply, checked by division
    if (t_sa / 3ull != t_acc.lo || t_st / 2ull != t_total.lo) no_overflow = false;
    st64(pf + PF_OFF_TALLY_T, t_total.lo); st64(pf + PF_OFF_TALLY_T + 8, t_acc.lo); st64(pf + PF_OFF_TALLY_T + 16, t_sa); st64(pf + PF_OFF_TALLY_T + 24, t_st);
    const uint64_t r_sa = r_acc.lo * 3ull, r_st = r_total.lo;
    if (skip && r_sa / 3ull != r_acc.lo) no_overflow = false;
    st64(pf + PF_OFF_TALLY_R, skip ? r_total.lo : 0ull); st64(pf + PF_OFF_TALLY_R + 8, skip ? r_acc.lo : 0ull);
    st64(pf + PF_OFF_TALLY_R + 16, skip ? r_sa : 0ull); st64(pf + PF_OFF_TALLY_R + 24, skip ? r_st : 0ull);
    if (!no_overflow) atomicAdd(&tp[2], 1u);
  }
}

// thresholds, checks, verdict of one proof from what the roles left in the proof record, the tree nodes and the counters (proof_body
// phase 5).  The whole workgroup (any size): the 32-byte comparisons and the chain-id bytes are one byte per thread; thread 0 assembles.  Owns [PF_OFF_VERDICTS, PF_OFF_HEIGHT) of the record and the report.
__device__ __forceinline__ void tiny_final_checks(const TinyProof& A, uint32_t p, uint32_t* tp, uint32_t* s_eq /*[10]*/) {
  const ProofParams& P = A.P;
  const bool skip = P.kind == 0;
  const uint32_t n = P.n, t = threadIdx.x;
  const uint8_t* pr = A.in_proof + (size_t)p * PR_STRIDE;
  uint8_t* pf = A.pf + (size_t)p * PF_STRIDE;
  // the computed validators hash of a set: the last node of its tree (a one-lane tree: the leaf hash itself)
  const uint8_t* root_t = P.tree_nodes ? A.nodes_t + ((size_t)p * P.tree_nodes + P.tree_nodes - 1) * 32 : A.lt + (size_t)p * n * A.lt_stride + LN_OFF_LEAF;
  const uint8_t* root_r = P.tree_nodes ? A.nodes_r + ((size_t)p * P.tree_nodes + P.tree_nodes - 1) * 32 : A.lr + (size_t)p * n * LANE_STRIDE + LN_OFF_LEAF;
  const uint8_t* hdr_hash = pf + PF_OFF_HEADER;
  auto incl_root = [&](int q) { return (const uint8_t*)(pf + PF_OFF_PROOFD + 160 * q + 128); };
  if (t < 10) s_eq[t] = t == 8 ? (P.chain_id_len <= 50 ? 1u : 0u) : 1u;
  __syncthreads();
  for (uint32_t idx = t; idx < 320; idx += blockDim.x) {
    if (idx < 256) {  // comparison c = idx / 32, byte idx % 32
      const uint32_t c = idx >> 5, k = idx & 31u;
      const uint8_t *x, *y;
      if (skip) {
        x = c == 0 ? incl_root(3) : c == 1 ? root_r : c == 2 ? root_t : c == 3 ? incl_root(2) : c == 4 ? incl_root(0) : incl_root(1);
        y = c == 0 ? pr + PR_OFF_HASH : c == 1 ? pf + PF_OFF_LEAFX + 2 : c == 2 ? pf + PF_OFF_LEAFV + 2 : hdr_hash;
      } else {
        x = c == 0 ? root_t : c == 1 ? incl_root(2) : c == 2 ? incl_root(0) : c == 3 ? incl_root(1) : c == 4 ? incl_root(3)
            : c == 5 ? pf + PF_OFF_LEAFX + 2 : c == 6 ? incl_root(4) : pf + PF_OFF_LEAFV + 2;
        y = c == 0 ? pf + PF_OFF_LEAFV + 2 : c <= 4 ? hdr_hash : c == 5 ? pr + PR_OFF_HASH : c == 6 ? pr + PR_OFF_HASH : pf + PF_OFF_LEAFY + 2;
      }
      if (c < (skip ? 6u : 8u) && x[k] != y[k]) s_eq[c] = 0;  // (every writer stores the same value)
    } else {
      const uint32_t k = idx - 256;
      if (k < P.chain_id_len && k < 50 && pf[PF_OFF_CID52 + 2 + k] != P.chain_id[k]) s_eq[8] = 0;
    }
  }
  __syncthreads();
  if (t != 0) return;
  const uint64_t block_a = ld64(pr + PR_OFF_BLOCK_A), block_b = ld64(pr + PR_OFF_BLOCK_B), round_ = ld64(pr + PR_OFF_ROUND);
  const uint32_t nb = ld32(pr + PR_OFF_NB_A), nbt = ld32(pr + PR_OFF_NB_B);
  const uint64_t t_sa = ld64(pf + PF_OFF_TALLY_T + 16), t_st = ld64(pf + PF_OFF_TALLY_T + 24);
  const uint64_t r_sa = ld64(pf + PF_OFF_TALLY_R + 16), r_st = ld64(pf + PF_OFF_TALLY_R + 24);
  const bool gt_t = t_sa > t_st;
  bool gt_r = false, dist_gt = false, dist_le = false;
  if (skip) {
    gt_r = r_sa > r_st;
    dist_gt = block_b > block_a + 1;            // verify.rs:515-518
    dist_le = block_b <= block_a + P.skip_max;  // verify.rs:520-525
  }
  for (uint32_t i = PF_OFF_VERDICTS; i < PF_OFF_HEIGHT; i += 4) st32(pf + i, 0u);
  st32(pf + PF_OFF_VERDICTS, gt_t); st32(pf + PF_OFF_VERDICTS + 4, gt_r); st32(pf + PF_OFF_VERDICTS + 8, dist_gt); st32(pf + PF_OFF_VERDICTS + 12, dist_le);
  const bool chain_ok = s_eq[8] != 0;
  const uint64_t height_a = ld64(pf + PF_OFF_HEIGHT);
  const bool all_sigdata = tp[0] == 0, all_eddsa = true /* patched by the verdict merge */, varint_ok = tp[1] == 0, no_overflow = tp[2] == 0;
  bool checks[16];
  int nc = 0;
  if (skip) {
    checks[nc++] = s_eq[0];                                       // verify.rs:374-379
    checks[nc++] = s_eq[1];                                       // verify.rs:382-389
    checks[nc++] = s_eq[2];                                       // verify.rs:279-280
    checks[nc++] = s_eq[3];                                       // verify.rs:283-286
    checks[nc++] = s_eq[4];                                       // verify.rs:205-209
    checks[nc++] = chain_ok;                                      // verify.rs:211-221
    checks[nc++] = s_eq[5];                                       // shared.rs:197-203
    checks[nc++] = height_a == block_b;                           // shared.rs:206
    checks[nc++] = all_sigdata; checks[nc++] = all_eddsa; checks[nc++] = no_overflow; checks[nc++] = varint_ok;
    checks[nc++] = (round_ >> 63) == 0;                           // validator.rs:73-78 (asserted at :141)
  } else {
    checks[nc++] = s_eq[0];
    checks[nc++] = s_eq[1];
    checks[nc++] = s_eq[2];
    checks[nc++] = chain_ok;
    checks[nc++] = s_eq[3];
    checks[nc++] = height_a == block_b;
    checks[nc++] = all_sigdata; checks[nc++] = all_eddsa; checks[nc++] = no_overflow; checks[nc++] = varint_ok;
    checks[nc++] = s_eq[4];                                       // verify.rs:144-147
    checks[nc++] = s_eq[5];                                       // verify.rs:150-153
    checks[nc++] = s_eq[6];                                       // verify.rs:166-170
    checks[nc++] = s_eq[7];                                       // verify.rs:173-177
    checks[nc++] = (round_ >> 63) == 0;
  }
  bool all_ok = true;
  uint32_t fail_mask = 0;
  for (int k = 0; k < nc; k++) {
    st32(pf + PF_OFF_CHECKS + 4 * k, checks[k] ? 1u : 0u);
    all_ok = all_ok && checks[k];
    if (!checks[k]) fail_mask |= 1u << k;
  }
  all_ok = all_ok && gt_t && (skip ? (gt_r && dist_gt && dist_le) : true);
  st32(pf + PF_OFF_ALLOK, all_ok ? 1u : 0u);
  uint8_t* rp = A.reports + (size_t)p * 64;
  for (int k = 0; k < 32; k += 4) st32(rp + k, ld32(hdr_hash + k));
  st32(rp + 32, all_ok ? 1u : 0u);
  st32(rp + 36, fail_mask);
  st32(rp + 40, 0xffffffffu);
  st32(rp + 44, gt_t ? 1u : 0u);
  st32(rp + 48, gt_r ? 1u : 0u);
  st32(rp + 52, (skip && dist_gt && dist_le) ? 1u : 0u);
  st32(rp + 56, (nb > n ? 1u : 0u) | (skip && nbt > n ? 2u : 0u));
  st32(rp + 60, 0u);
  tp[0] = 0; tp[1] = 0; tp[2] = 0;  // (the counters of the next launch)
}

// ------------------------------------------------------------------------------------------------ the two kernels
union TinyLds {
  TinyLaneLds lane;
  uint32_t leaf[TMX_N_LIMIT][8];
  uint32_t hdr[2][28][8];
  TinyTallyLds tally;
};

template <int KW, int BW>
__global__ __launch_bounds__(TINY_THREADS) void k_tiny(TinyEd E, TinyProof PA, TinySer SA, uint32_t roles_per_proof) {
  __shared__ TinyLds lds;
  const uint32_t b = blockIdx.x;
  const uint32_t proof_blocks = PA.n_proofs * roles_per_proof;
  if (b < E.n_lanes) {
    if (TINY_DBG(0x1)) return;
    tiny_lane<KW, BW>(E, b, lds.lane);
  } else if (b < E.n_lanes + proof_blocks) {
    const uint32_t r = b - E.n_lanes, p = r / roles_per_proof, role = r % roles_per_proof;
    uint32_t* tp = E.tiny + TN_WORDS + TN_PER_PROOF * p;
    // roles of a proof: 0 target tree, 1 header, 2 tallies, 3 trusted tree (skip only)
    if (TINY_DBG(role == 1 ? 0x4 : role == 2 ? 0x8 : 0x2)) return;
    if (role == 0) tiny_tree(PA, p, 0, tp, lds.leaf);
    else if (role == 1) tiny_header(PA, p, tp, lds.hdr);
    else if (role == 2) tiny_tally(PA, p, tp, lds.tally);
    else tiny_tree(PA, p, 1, tp, lds.leaf);
  } else if (SA.out) {  // input role: two serializer spans per workgroup
    if (TINY_DBG(0x10)) return;
    const uint32_t r = b - E.n_lanes - proof_blocks;
    const uint32_t per = 2 * SA.n_blocks, proof = r / per, k = r - proof * per;
    const uint32_t span = __builtin_amdgcn_readfirstlane(SA.first_block * 4 + 2 * k + (threadIdx.x >> 6));
    serialize_span<SER_SPAN_ELEMS>(SA.S, SA.lut, SA.wave_sec, SA.out, SA.mask, proof, span);
  }
}

// behind k_tiny on the same stream.  Workgroup p < n_proofs finishes proof p: checks, verdict merge, then the few elements from the first
// one that depends on them to the row end; workgroup n_proofs + p expands the seam spans of row p in front of that point; the others
// expand whole spans, sixteen each, of every section that needed k_tiny's results.
constexpr uint32_t TINY_TAIL_THREADS = 1024;
__global__ __launch_bounds__(TINY_TAIL_THREADS) void k_tiny_tail(TinyProof PA, TinySer SA, const uint8_t* __restrict__ ed, uint32_t ed_stride, uint32_t* tiny) {
  const uint32_t b = blockIdx.x, T = blockDim.x;
  const SerializeProgram& S = SA.S;
  if (b >= 2 * PA.n_proofs) {
    if (!SA.out || TINY_DBG(0x2000)) return;
    const uint32_t per = (SA.n_blocks + 3) / 4;  // workgroups per proof: sixteen spans = four span blocks each
    const uint32_t r = b - 2 * PA.n_proofs, proof = r / per, k = r - proof * per;
    const uint32_t span = __builtin_amdgcn_readfirstlane(SA.first_block * 4 + k * 16 + (threadIdx.x >> 6));
    if (span < (SA.first_block + SA.n_blocks) * 4 && span < SA.tail_first_span) serialize_span<SER_SPAN_ELEMS>(S, SA.lut, SA.wave_sec, SA.out, SA.mask, proof, span);
    return;
  }
  if (b >= PA.n_proofs) {  // seam spans in front of the dependent tail
    if (!SA.out || TINY_DBG(0x8000)) return;
    const uint32_t proof = b - PA.n_proofs;
    uint64_t* row = SA.out + (size_t)proof * S.elem_stride;
    for (uint32_t i = threadIdx.x; i < SA.n_seams * S.span; i += T) {
      const uint32_t sp = SA.seam_waves[i / S.span], e = sp * S.span + i % S.span;
      if (sp < SA.tail_first_span && e < S.elem_stride) row[e] = e < S.elem_count ? serialize_one(S.sec[section_of(S, e)], SA.lut, proof, e) : 0;
    }
    return;
  }
  const uint32_t proof = b;
  if (TINY_DBG(0x1000)) return;
  __shared__ uint32_t s_eq[10];
  if (!TINY_DBG(0x4000)) tiny_final_checks(PA, proof, tiny + TN_WORDS + TN_PER_PROOF * proof, s_eq);
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");  // (the record is patched and read back inside this workgroup only)
  __syncthreads();
  if (threadIdx.x < 64) {  // verdict merge of this proof (verdict_body indexes by blockIdx.x == proof); the lanes' D.1b elements are in the rows already
    verdict_body(PA.P.kind, PA.P.n, ed, ed_stride, PA.pf, PA.reports, RowOut{});
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
  __syncthreads();
  if (!SA.out || TINY_DBG(0x8000)) return;
  uint64_t* row = SA.out + (size_t)proof * S.elem_stride;
  for (uint32_t e = SA.tail_first_span * S.span + threadIdx.x; e < S.elem_stride; e += T)
    row[e] = e < S.elem_count ? serialize_one(S.sec[section_of(S, e)], SA.lut, proof, e) : 0;
}

// ------------------------------------------------------------------------------------------------ k_proof as roles (any batch size)
// The classic launch graph's k_proof was one workgroup per proof running its phases one after the other (120 us for one proof, 260 us
// inside a 256-proof step).  The same values from FOUR workgroups per proof (three for step) that run side by side -- target tree, header
// (+ inclusion proofs + sign-bytes checks), tallies (+ N x N match), trusted tree -- and the workgroup that finishes LAST for a proof
// (an agent-scope counter; nobody waits for anybody) runs the thresholds / checks / verdict on what all of them left.
__global__ __launch_bounds__(TINY_THREADS) void k_proof_roles(TinyProof PA, uint32_t* __restrict__ tiny, uint32_t roles) {
  __shared__ union { uint32_t leaf[TMX_N_LIMIT][8]; uint32_t hdr[2][28][8]; TinyTallyLds tally; } lds;
  __shared__ uint32_t s_last, s_eq[10];
  if (gridDim.x >= 2048) __builtin_amdgcn_s_setprio(2);  // (large batches: ahead of the EdDSA throughput waves, as k_proof did)
  const uint32_t b = blockIdx.x, p = b / roles, role = b - p * roles;
  uint32_t* tp = tiny + TN_WORDS + TN_PER_PROOF * p;
  if (role == 0) tiny_tree(PA, p, 0, tp, lds.leaf);
  else if (role == 1) tiny_header(PA, p, tp, lds.hdr);
  else if (role == 2) tiny_tally(PA, p, tp, lds.tally);
  else tiny_tree(PA, p, 1, tp, lds.leaf);
  __threadfence();  // release: this workgroup's records, for whichever workgroup turns out to be the last
  __syncthreads();
  if (threadIdx.x == 0) s_last = atomicAdd(&tp[3], 1u) == roles - 1 ? 1u : 0u;
  __syncthreads();
  if (!s_last) return;
  __threadfence();  // acquire: the other roles' records
  if (threadIdx.x == 0) tp[3] = 0;
  tiny_final_checks(PA, p, tp, s_eq);
}
