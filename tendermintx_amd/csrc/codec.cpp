// Input codec of libtmx: CometBFT RPC / reference-fixture JSON  ->  packed input records (include/tmx.h).
//
// Host-side mirror of the reference's fixture-mode data path, up to (not including) any hashing or signature check:
//   InputDataFetcher::{get_signed_header_from_number, get_validator_set_from_number}   reference circuits/input/mod.rs:188-282
//   get_validator_data_from_block / get_signed_message_data                             reference circuits/input/conversion.rs:22-137
//   validator_hash_field_from_block                                                     reference circuits/input/conversion.rs:139-178
//   get_vote_from_commit_sig                                                            reference circuits/input/tendermint_utils.rs:404-441
//   header -> 14 protobuf field encodings                                               reference circuits/input/tendermint_utils.rs:374-393
// The protobuf / amino-JSON shapes come from tendermint-rs 0.33.2 + tendermint-proto (un-vendored); they are
// restated here from the wire format and pinned by the reference fixtures (tests/test_abi_codec.py).
// The host Ed25519 check of conversion.rs:48-49 is NOT done here: it runs on the GPU (k_eddsa) and is reported as
// tmx_report.first_bad_sig.
#include <algorithm>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "tmx.h"

namespace {

// ------------------------------------------------------------------------------------------------ tiny JSON
struct JVal;
using JPtr = std::shared_ptr<JVal>;
struct JVal {
  enum Kind { Null, Bool, Num, Str, Arr, Obj } kind = Null;
  bool b = false;
  std::string s;  // Str, and the raw text of Num
  std::vector<JPtr> a;
  std::vector<std::pair<std::string, JPtr>> o;
  const JVal* get(const char* key) const {
    if (kind != Obj) return nullptr;
    for (auto& kv : o)
      if (kv.first == key) return kv.second.get();
    return nullptr;
  }
};
struct JParser {
  const char* p;
  const char* end;
  bool ok = true;
  void ws() { while (p < end && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) p++; }
  bool lit(const char* t) {
    size_t n = std::strlen(t);
    if ((size_t)(end - p) >= n && std::memcmp(p, t, n) == 0) { p += n; return true; }
    return false;
  }
  JPtr parse() {
    ws();
    auto v = std::make_shared<JVal>();
    if (p >= end) { ok = false; return v; }
    if (*p == '{') {
      v->kind = JVal::Obj; p++; ws();
      if (p < end && *p == '}') { p++; return v; }
      while (ok) {
        ws();
        JPtr k = parse();
        if (!ok || k->kind != JVal::Str) { ok = false; break; }
        ws();
        if (p >= end || *p != ':') { ok = false; break; }
        p++;
        JPtr val = parse();
        v->o.emplace_back(k->s, val);
        ws();
        if (p < end && *p == ',') { p++; continue; }
        if (p < end && *p == '}') { p++; break; }
        ok = false;
      }
    } else if (*p == '[') {
      v->kind = JVal::Arr; p++; ws();
      if (p < end && *p == ']') { p++; return v; }
      while (ok) {
        v->a.push_back(parse());
        ws();
        if (p < end && *p == ',') { p++; continue; }
        if (p < end && *p == ']') { p++; break; }
        ok = false;
      }
    } else if (*p == '"') {
      v->kind = JVal::Str; p++;
      while (p < end && *p != '"') {
        if (*p == '\\' && p + 1 < end) {
          p++;
          switch (*p) {
            case 'n': v->s += '\n'; break;
            case 't': v->s += '\t'; break;
            case 'r': v->s += '\r'; break;
            case 'b': v->s += '\b'; break;
            case 'f': v->s += '\f'; break;
            case 'u': v->s += '?'; p += (end - p > 4) ? 4 : 0; break;  // no non-ASCII in this data
            default: v->s += *p;
          }
          p++;
        } else {
          v->s += *p++;
        }
      }
      if (p >= end) ok = false; else p++;
    } else if (lit("true")) { v->kind = JVal::Bool; v->b = true; }
    else if (lit("false")) { v->kind = JVal::Bool; }
    else if (lit("null")) { v->kind = JVal::Null; }
    else {
      v->kind = JVal::Num;
      const char* s = p;
      while (p < end && (std::strchr("+-0123456789.eE", *p) != nullptr)) p++;
      if (p == s) ok = false;
      v->s.assign(s, p);
    }
    return v;
  }
};

bool to_u64(const JVal* v, uint64_t& out) {
  if (!v || (v->kind != JVal::Str && v->kind != JVal::Num) || v->s.empty()) return false;
  uint64_t x = 0;
  size_t i = 0;
  bool neg = false;
  if (v->s[0] == '-') { neg = true; i = 1; }
  for (; i < v->s.size(); i++) {
    if (v->s[i] < '0' || v->s[i] > '9') return false;
    x = x * 10 + (uint64_t)(v->s[i] - '0');
  }
  out = neg ? (uint64_t)(-(int64_t)x) : x;
  return true;
}
bool from_hex(const std::string& s, std::vector<uint8_t>& out) {
  out.clear();
  if (s.size() % 2) return false;
  auto nib = [](char c) -> int { return c >= '0' && c <= '9' ? c - '0' : c >= 'a' && c <= 'f' ? c - 'a' + 10 : c >= 'A' && c <= 'F' ? c - 'A' + 10 : -1; };
  for (size_t i = 0; i < s.size(); i += 2) {
    int a = nib(s[i]), b = nib(s[i + 1]);
    if (a < 0 || b < 0) return false;
    out.push_back((uint8_t)(a * 16 + b));
  }
  return true;
}
bool from_b64(const std::string& s, std::vector<uint8_t>& out) {
  out.clear();
  uint32_t acc = 0;
  int bits = 0;
  for (char c : s) {
    int v;
    if (c >= 'A' && c <= 'Z') v = c - 'A';
    else if (c >= 'a' && c <= 'z') v = c - 'a' + 26;
    else if (c >= '0' && c <= '9') v = c - '0' + 52;
    else if (c == '+') v = 62;
    else if (c == '/') v = 63;
    else if (c == '=') break;
    else return false;
    acc = (acc << 6) | (uint32_t)v;
    bits += 6;
    if (bits >= 8) { bits -= 8; out.push_back((uint8_t)(acc >> bits)); }
  }
  return true;
}

// RFC 3339 UTC "YYYY-MM-DDTHH:MM:SS[.fraction]Z" -> (seconds since epoch, nanos)
bool parse_time(const std::string& ts, int64_t& secs, uint32_t& nanos) {
  int y, mo, d, hh, mm, ss;
  if (ts.size() < 20 || ts.back() != 'Z') return false;
  if (std::sscanf(ts.c_str(), "%4d-%2d-%2dT%2d:%2d:%2d", &y, &mo, &d, &hh, &mm, &ss) != 6) return false;
  nanos = 0;
  size_t dot = ts.find('.');
  if (dot != std::string::npos) {
    uint32_t scale = 100000000;
    for (size_t i = dot + 1; i + 1 < ts.size() && scale; i++, scale /= 10) {
      if (ts[i] < '0' || ts[i] > '9') return false;
      nanos += (uint32_t)(ts[i] - '0') * scale;
    }
  }
  // days from civil (proleptic Gregorian)
  int yy = y - (mo <= 2);
  int era = (yy >= 0 ? yy : yy - 399) / 400;
  unsigned yoe = (unsigned)(yy - era * 400);
  unsigned doy = (153u * (unsigned)(mo + (mo > 2 ? -3 : 9)) + 2) / 5 + (unsigned)d - 1;
  unsigned doe = yoe * 365 + yoe / 4 - yoe / 100 + doy;
  int64_t days = (int64_t)era * 146097 + (int64_t)doe - 719468;
  secs = days * 86400 + hh * 3600 + mm * 60 + ss;
  return true;
}

// ------------------------------------------------------------------------------------------------ protobuf
using Bytes = std::vector<uint8_t>;
void put_varint(Bytes& o, uint64_t v) {
  while (v >= 0x80) { o.push_back((uint8_t)(v | 0x80)); v >>= 7; }
  o.push_back((uint8_t)v);
}
void put_bytes_field(Bytes& o, int field, const Bytes& b) {
  o.push_back((uint8_t)((field << 3) | 2));
  put_varint(o, b.size());
  o.insert(o.end(), b.begin(), b.end());
}
void put_varint_field(Bytes& o, int field, uint64_t v) {
  o.push_back((uint8_t)((field << 3) | 0));
  put_varint(o, v);
}
Bytes enc_timestamp(int64_t secs, uint32_t nanos) {  // google.protobuf.Timestamp, zero fields omitted
  Bytes o;
  if (secs) put_varint_field(o, 1, (uint64_t)secs);
  if (nanos) put_varint_field(o, 2, nanos);
  return o;
}
Bytes enc_block_id(const Bytes& hash, uint64_t total, const Bytes& psh_hash) {  // BlockID / CanonicalBlockID
  Bytes psh, o;
  if (total) put_varint_field(psh, 1, total);
  if (!psh_hash.empty()) put_bytes_field(psh, 2, psh_hash);
  if (!hash.empty()) put_bytes_field(o, 1, hash);
  put_bytes_field(o, 2, psh);
  return o;
}
Bytes wrap_bytes(const Bytes& b) {  // google.protobuf.BytesValue; empty value -> empty encoding
  Bytes o;
  if (!b.empty()) put_bytes_field(o, 1, b);
  return o;
}

struct BlockId {
  Bytes hash, psh_hash;
  uint64_t total = 0;
  bool present = false;
};
bool parse_block_id(const JVal* v, BlockId& out) {
  out = BlockId();
  if (!v || v->kind != JVal::Obj) return true;  // absent -> default
  const JVal* h = v->get("hash");
  if (!h || h->kind != JVal::Str || h->s.empty()) return true;
  if (!from_hex(h->s, out.hash)) return false;
  const JVal* parts = v->get("parts");
  if (!parts) parts = v->get("part_set_header");
  if (parts) {
    const JVal* ph = parts->get("hash");
    if (ph && ph->kind == JVal::Str && !from_hex(ph->s, out.psh_hash)) return false;
    if (!to_u64(parts->get("total"), out.total)) return false;
  }
  out.present = true;
  return true;
}

// the 14 header leaves (tendermint_utils.rs:374-393)
int32_t encode_header(const JVal* h, tmx_header_rec* out) {
  std::memset(out, 0, sizeof *out);
  if (!h || h->kind != JVal::Obj) return TMX_ERR_PARSE;
  std::vector<Bytes> leaves(14);
  auto hexfield = [&](const char* name, Bytes& b) -> bool {
    const JVal* v = h->get(name);
    b.clear();
    if (!v || v->kind == JVal::Null) return true;
    return v->kind == JVal::Str && from_hex(v->s, b);
  };
  const JVal* ver = h->get("version");
  uint64_t vb = 0, va = 0;
  if (ver) { (void)to_u64(ver->get("block"), vb); (void)to_u64(ver->get("app"), va); }
  if (vb) put_varint_field(leaves[0], 1, vb);
  if (va) put_varint_field(leaves[0], 2, va);
  const JVal* cid = h->get("chain_id");
  if (!cid || cid->kind != JVal::Str) return TMX_ERR_PARSE;
  if (!cid->s.empty()) put_bytes_field(leaves[1], 1, Bytes(cid->s.begin(), cid->s.end()));
  uint64_t height = 0;
  if (!to_u64(h->get("height"), height)) return TMX_ERR_PARSE;
  if (height) put_varint_field(leaves[2], 1, height);
  const JVal* tm = h->get("time");
  int64_t secs; uint32_t nanos;
  if (!tm || tm->kind != JVal::Str || !parse_time(tm->s, secs, nanos)) return TMX_ERR_PARSE;
  leaves[3] = enc_timestamp(secs, nanos);
  BlockId lb;
  if (!parse_block_id(h->get("last_block_id"), lb)) return TMX_ERR_PARSE;
  if (lb.present) leaves[4] = enc_block_id(lb.hash, lb.total, lb.psh_hash);
  const char* names[9] = {"last_commit_hash", "data_hash", "validators_hash", "next_validators_hash", "consensus_hash",
                          "app_hash", "last_results_hash", "evidence_hash", "proposer_address"};
  for (int k = 0; k < 9; k++) {
    Bytes b;
    if (!hexfield(names[k], b)) return TMX_ERR_PARSE;
    leaves[5 + k] = wrap_bytes(b);
  }
  for (int i = 0; i < 14; i++) {
    if (leaves[i].size() > 79) return TMX_ERR_PARSE;
    out->leaf_len[i] = (uint8_t)leaves[i].size();
    if (!leaves[i].empty()) std::memcpy(out->leaf[i], leaves[i].data(), leaves[i].size());  // (an empty vector's data() may be null)
  }
  return TMX_OK;
}

struct Validator {
  Bytes pubkey, address;
  uint64_t power = 0;
  uint32_t byte_len = 0;  // Info::hash_bytes().len()  (conversion.rs:75, 154)
};
struct CommitSig {
  int flag = 1;
  std::string timestamp;
  Bytes signature;
};
struct SignedHeader {
  const JVal* header = nullptr;
  std::string chain_id;
  uint64_t height = 0, round = 0;
  BlockId block_id;
  std::vector<CommitSig> sigs;
};

// one or more validators?page=k responses back to back (reference pages 100 per request, mod.rs:219-241)
int32_t parse_validators(const char* json, std::vector<Validator>& out, std::vector<JPtr>& keep) {
  out.clear();
  JParser P{json, json + std::strlen(json)};
  while (true) {
    P.ws();
    if (P.p >= P.end) break;
    JPtr root = P.parse();
    if (!P.ok) return TMX_ERR_PARSE;
    keep.push_back(root);
    const JVal* res = root->get("result");
    const JVal* vals = res ? res->get("validators") : nullptr;
    if (!vals && res) {  // SignedBlockResponse (tendermint_utils.rs:52-55, 97-112): result.validator_set.validators
      const JVal* vs = res->get("validator_set");
      vals = vs ? vs->get("validators") : nullptr;
    }
    if (!vals || vals->kind != JVal::Arr) return TMX_ERR_PARSE;
    for (auto& v : vals->a) {
      Validator x;
      const JVal* pk = v->get("pub_key");
      const JVal* pkv = pk ? pk->get("value") : nullptr;
      const JVal* addr = v->get("address");
      if (!pkv || pkv->kind != JVal::Str || !from_b64(pkv->s, x.pubkey) || x.pubkey.size() != 32) return TMX_ERR_PARSE;
      if (!addr || addr->kind != JVal::Str || !from_hex(addr->s, x.address)) return TMX_ERR_PARSE;
      if (!to_u64(v->get("voting_power"), x.power)) return TMX_ERR_PARSE;
      // SimpleValidator{pub_key: PublicKey{ed25519}, voting_power}: 0a 22 0a 20 pk [10 varint(power)]
      Bytes vp;
      if (x.power) put_varint_field(vp, 2, x.power);
      x.byte_len = 36 + (uint32_t)vp.size();
      out.push_back(std::move(x));
    }
  }
  return out.empty() ? TMX_ERR_PARSE : TMX_OK;
}

int32_t parse_commit(const char* json, SignedHeader& sh, std::vector<JPtr>& keep) {
  JParser P{json, json + std::strlen(json)};
  JPtr root = P.parse();
  if (!P.ok) return TMX_ERR_PARSE;
  keep.push_back(root);
  const JVal* res = root->get("result");
  const JVal* s = res ? res->get("signed_header") : nullptr;
  // CommitResponse {result: {signed_header: {header, commit}}} (tendermint_utils.rs:58-66) or SignedBlockResponse {result: {header,
  // data, commit, validator_set}} (:52-55, 97-112; the signed_block.json fixtures): the same header + commit either way
  if (!s && res && res->get("header") && res->get("commit")) s = res;
  if (!s) return TMX_ERR_PARSE;
  sh.header = s->get("header");
  const JVal* c = s->get("commit");
  if (!sh.header || !c) return TMX_ERR_PARSE;
  const JVal* cid = sh.header->get("chain_id");
  if (!cid || cid->kind != JVal::Str) return TMX_ERR_PARSE;
  sh.chain_id = cid->s;
  if (!to_u64(c->get("height"), sh.height) || !to_u64(c->get("round"), sh.round)) return TMX_ERR_PARSE;
  if (!parse_block_id(c->get("block_id"), sh.block_id)) return TMX_ERR_PARSE;
  const JVal* sigs = c->get("signatures");
  if (!sigs || sigs->kind != JVal::Arr) return TMX_ERR_PARSE;
  for (auto& e : sigs->a) {
    CommitSig cs;
    uint64_t flag = 0;
    if (!to_u64(e->get("block_id_flag"), flag)) return TMX_ERR_PARSE;
    cs.flag = (int)flag;
    const JVal* ts = e->get("timestamp");
    if (ts && ts->kind == JVal::Str) cs.timestamp = ts->s;
    const JVal* sg = e->get("signature");
    if (sg && sg->kind == JVal::Str && !from_b64(sg->s, cs.signature)) return TMX_ERR_PARSE;
    sh.sigs.push_back(std::move(cs));
  }
  return TMX_OK;
}

// DUMMY_PUBLIC_KEY / DUMMY_SIGNATURE (plonky2x constants imported at conversion.rs:3-5; RFC 8032 pair of seed 01x32
// over the 32-byte zero message -- same literals as kernels.hip, pinned by tests/test_oracle_kat.py)
const uint8_t DUMMY_PK[32] = {0x8a, 0x88, 0xe3, 0xdd, 0x74, 0x09, 0xf1, 0x95, 0xfd, 0x52, 0xdb, 0x2d, 0x3c, 0xba, 0x5d, 0x72,
                              0xca, 0x67, 0x09, 0xbf, 0x1d, 0x94, 0x12, 0x1b, 0xf3, 0x74, 0x88, 0x01, 0xb4, 0x0f, 0x6f, 0x5c};
const uint8_t DUMMY_SIG[64] = {0x37, 0x14, 0x68, 0x9e, 0x54, 0x78, 0xc2, 0x11, 0x06, 0xed, 0x9d, 0xa4, 0x55, 0x58, 0x9e, 0x89,
                               0xbb, 0x77, 0xbb, 0xf0, 0x9f, 0x49, 0x50, 0x3f, 0x85, 0xa2, 0x4a, 0x5b, 0x30, 0x35, 0x06, 0x8a,
                               0x01, 0x29, 0x16, 0x79, 0xf9, 0x2e, 0xc6, 0x91, 0x9b, 0x66, 0x03, 0xd2, 0xa8, 0x87, 0xad, 0x37,
                               0xfc, 0x48, 0x2d, 0x7e, 0xa9, 0xb2, 0xbf, 0x07, 0x99, 0x43, 0x70, 0x5a, 0x96, 0x21, 0x8c, 0x07};

// CanonicalVote sign-bytes, length-delimited (SignedVote::sign_bytes, conversion.rs:33-36)
int32_t sign_bytes(const SignedHeader& sh, const CommitSig& cs, Bytes& out) {
  Bytes body;
  put_varint_field(body, 1, 2);  // SIGNED_MSG_TYPE_PRECOMMIT (tendermint_utils.rs:430)
  auto sfixed64 = [&](int field, uint64_t v) {
    body.push_back((uint8_t)((field << 3) | 1));
    for (int i = 0; i < 8; i++) body.push_back((uint8_t)(v >> (8 * i)));
  };
  if (sh.height) sfixed64(2, sh.height);
  if (sh.round) sfixed64(3, sh.round);
  if (cs.flag == 2) put_bytes_field(body, 4, enc_block_id(sh.block_id.hash, sh.block_id.total, sh.block_id.psh_hash));
  int64_t secs; uint32_t nanos;
  if (!parse_time(cs.timestamp, secs, nanos)) return TMX_ERR_PARSE;
  put_bytes_field(body, 5, enc_timestamp(secs, nanos));
  if (!sh.chain_id.empty()) put_bytes_field(body, 6, Bytes(sh.chain_id.begin(), sh.chain_id.end()));
  out.clear();
  put_varint(out, body.size());
  out.insert(out.end(), body.begin(), body.end());
  return TMX_OK;
}

// get_validator_data_from_block (conversion.rs:59-137)
int32_t build_target_lanes(const std::vector<Validator>& vals, const SignedHeader& sh, uint32_t n_max, tmx_validator_rec* out) {
  if (sh.sigs.size() > n_max || vals.size() > n_max) return TMX_ERR_SET_TOO_LARGE;
  if (sh.sigs.size() > vals.size()) return TMX_ERR_PARSE;
  std::memset(out, 0, sizeof(tmx_validator_rec) * n_max);
  for (uint32_t i = 0; i < n_max; i++) {
    tmx_validator_rec& r = out[i];
    if (i < sh.sigs.size()) {
      const Validator& v = vals[i];
      std::memcpy(r.pubkey, v.pubkey.data(), 32);
      r.voting_power = v.power;
      r.validator_byte_length = (uint8_t)v.byte_len;
      if (sh.sigs[i].flag == 2) {  // is_commit()
        Bytes msg;
        int32_t st = sign_bytes(sh, sh.sigs[i], msg);
        if (st) return st;
        if (msg.size() > TMX_VALIDATOR_MESSAGE_BYTES_LENGTH_MAX) return TMX_ERR_MSG_TOO_LONG;
        if (sh.sigs[i].signature.size() != 64) return TMX_ERR_PARSE;
        std::memcpy(r.signature, sh.sigs[i].signature.data(), 64);
        std::memcpy(r.message, msg.data(), msg.size());
        r.message_byte_length = (uint16_t)msg.size();
        r.flags = TMX_FLAG_SIGNED | TMX_FLAG_PRESENT;
      } else {  // in the validator set, did not commit: dummy signature lane (conversion.rs:98-114)
        std::memcpy(r.signature, DUMMY_SIG, 64);
        r.message_byte_length = 32;
        r.flags = TMX_FLAG_PRESENT;
      }
    } else {  // padding lane (conversion.rs:118-134)
      std::memcpy(r.pubkey, DUMMY_PK, 32);
      std::memcpy(r.signature, DUMMY_SIG, 64);
      r.message_byte_length = 32;
      r.voting_power = 0;
      r.validator_byte_length = TMX_VALIDATOR_BYTE_LENGTH_MAX;
      r.flags = 0;
    }
  }
  return TMX_OK;
}

// validator_hash_field_from_block (conversion.rs:139-178); Set::new orders by (power desc, address asc)
int32_t build_trusted_lanes(std::vector<Validator> vals, size_t n_sigs, uint32_t n_max, tmx_hashfield_rec* out) {
  if (n_sigs > n_max || vals.size() > n_max) return TMX_ERR_SET_TOO_LARGE;
  if (n_sigs > vals.size()) return TMX_ERR_PARSE;
  std::stable_sort(vals.begin(), vals.end(), [](const Validator& a, const Validator& b) {
    if (a.power != b.power) return a.power > b.power;
    return a.address < b.address;
  });
  std::memset(out, 0, sizeof(tmx_hashfield_rec) * n_max);
  for (uint32_t i = 0; i < n_max; i++) {
    tmx_hashfield_rec& r = out[i];
    if (i < n_sigs) {
      std::memcpy(r.pubkey, vals[i].pubkey.data(), 32);
      r.voting_power = vals[i].power;
      r.validator_byte_length = (uint8_t)vals[i].byte_len;
      r.flags = TMX_FLAG_PRESENT;
    } else {
      std::memcpy(r.pubkey, DUMMY_PK, 32);
      r.validator_byte_length = TMX_VALIDATOR_BYTE_LENGTH_MAX;
    }
  }
  return TMX_OK;
}

}  // namespace

extern "C" {

int32_t tmx_skip_inputs_from_json(const char* trusted_commit_json, const char* trusted_validators_json, const char* target_commit_json,
                                  const char* target_validators_json, uint32_t n_max, uint64_t trusted_block,
                                  const uint8_t trusted_header_hash[32], uint64_t target_block, tmx_proof_rec* proof,
                                  tmx_validator_rec* target, tmx_hashfield_rec* trusted) {
  if (!trusted_commit_json || !trusted_validators_json || !target_commit_json || !target_validators_json || !trusted_header_hash || !proof ||
      !target || !trusted || n_max == 0 || n_max > TMX_N_MAX_LIMIT)
    return TMX_ERR_BAD_ARG;
  std::vector<JPtr> keep;
  std::vector<Validator> tv, gv;
  SignedHeader tsh, gsh;
  int32_t st;
  if ((st = parse_validators(trusted_validators_json, tv, keep))) return st;
  if ((st = parse_validators(target_validators_json, gv, keep))) return st;
  if (tv.size() > n_max || gv.size() > n_max) return TMX_ERR_SET_TOO_LARGE;  // input/mod.rs:439-444
  if ((st = parse_commit(trusted_commit_json, tsh, keep))) return st;
  if ((st = parse_commit(target_commit_json, gsh, keep))) return st;
  std::memset(proof, 0, sizeof *proof);
  proof->block_a = trusted_block;
  proof->block_b = target_block;
  std::memcpy(proof->hash, trusted_header_hash, 32);
  proof->round = gsh.round;
  proof->nb_a = (uint32_t)gv.size();
  proof->nb_b = (uint32_t)tv.size();
  if ((st = encode_header(gsh.header, &proof->header_a))) return st;
  if ((st = encode_header(tsh.header, &proof->header_b))) return st;
  if ((st = build_target_lanes(gv, gsh, n_max, target))) return st;
  return build_trusted_lanes(tv, tsh.sigs.size(), n_max, trusted);
}

int32_t tmx_step_inputs_from_json(const char* prev_commit_json, const char* next_commit_json, const char* next_validators_json, uint32_t n_max,
                                  uint64_t prev_block, const uint8_t prev_header_hash[32], tmx_proof_rec* proof, tmx_validator_rec* target) {
  if (!prev_commit_json || !next_commit_json || !next_validators_json || !prev_header_hash || !proof || !target || n_max == 0 ||
      n_max > TMX_N_MAX_LIMIT)
    return TMX_ERR_BAD_ARG;
  std::vector<JPtr> keep;
  std::vector<Validator> nv;
  SignedHeader psh, nsh;
  int32_t st;
  if ((st = parse_commit(prev_commit_json, psh, keep))) return st;
  if ((st = parse_commit(next_commit_json, nsh, keep))) return st;
  if ((st = parse_validators(next_validators_json, nv, keep))) return st;
  if (nv.size() > n_max) return TMX_ERR_SET_TOO_LARGE;  // input/mod.rs:338-342
  std::memset(proof, 0, sizeof *proof);
  proof->block_a = prev_block;
  proof->block_b = prev_block + 1;
  std::memcpy(proof->hash, prev_header_hash, 32);
  proof->round = nsh.round;
  proof->nb_a = (uint32_t)nv.size();
  proof->nb_b = 0;
  if ((st = encode_header(nsh.header, &proof->header_a))) return st;
  if ((st = encode_header(psh.header, &proof->header_b))) return st;
  return build_target_lanes(nv, nsh, n_max, target);
}

// records for tmx_valid_skip_batch: validator sets by address, commit signatures by address (tendermint_utils.rs:444-482)
int32_t tmx_skipcheck_inputs_from_json(const char* start_validators_json, const char* target_validators_json, const char* target_commit_json,
                                       uint32_t n_max, tmx_addr_rec* start, uint32_t* n_start, tmx_addr_rec* target, uint32_t* n_target,
                                       tmx_addr_rec* sigs, uint32_t* n_sigs) {
  if (!start_validators_json || !target_validators_json || !target_commit_json || !start || !n_start || !target || !n_target || !sigs || !n_sigs ||
      n_max == 0 || n_max > TMX_N_MAX_LIMIT)
    return TMX_ERR_BAD_ARG;
  std::vector<JPtr> keep;
  std::vector<Validator> sv, tv;
  int32_t st;
  if ((st = parse_validators(start_validators_json, sv, keep))) return st;
  if ((st = parse_validators(target_validators_json, tv, keep))) return st;
  if (sv.size() > n_max || tv.size() > n_max) return TMX_ERR_SET_TOO_LARGE;
  auto by_power = [](const Validator& a, const Validator& b) {  // Set::new order (power desc, address asc); the result does not depend on it
    if (a.power != b.power) return a.power > b.power;
    return a.address < b.address;
  };
  std::stable_sort(sv.begin(), sv.end(), by_power);
  std::stable_sort(tv.begin(), tv.end(), by_power);
  auto fill = [&](const std::vector<Validator>& v, tmx_addr_rec* out) {
    std::memset(out, 0, sizeof(tmx_addr_rec) * n_max);
    for (size_t i = 0; i < v.size(); i++) {
      if (v[i].address.size() != 20) return false;
      std::memcpy(out[i].address, v[i].address.data(), 20);
      out[i].has_address = 1;
      out[i].voting_power = v[i].power;
    }
    return true;
  };
  if (!fill(sv, start) || !fill(tv, target)) return TMX_ERR_PARSE;
  *n_start = (uint32_t)sv.size();
  *n_target = (uint32_t)tv.size();
  // signatures: raw JSON again for the addresses (SignedHeader above keeps only what the witness path needs)
  JParser P{target_commit_json, target_commit_json + std::strlen(target_commit_json)};
  JPtr root = P.parse();
  if (!P.ok) return TMX_ERR_PARSE;
  const JVal* res = root->get("result");
  const JVal* sh = res ? res->get("signed_header") : nullptr;
  if (!sh && res && res->get("commit")) sh = res;  // SignedBlockResponse
  const JVal* cm = sh ? sh->get("commit") : nullptr;
  const JVal* sg = cm ? cm->get("signatures") : nullptr;
  if (!sg || sg->kind != JVal::Arr) return TMX_ERR_PARSE;
  if (sg->a.size() > n_max) return TMX_ERR_SET_TOO_LARGE;
  std::memset(sigs, 0, sizeof(tmx_addr_rec) * n_max);
  for (size_t i = 0; i < sg->a.size(); i++) {
    uint64_t flag = 0;
    if (!to_u64(sg->a[i]->get("block_id_flag"), flag)) return TMX_ERR_PARSE;
    const JVal* ad = sg->a[i]->get("validator_address");
    Bytes addr;
    if (flag != 1 && ad && ad->kind == JVal::Str && from_hex(ad->s, addr) && addr.size() == 20) {  // absent votes carry no address
      std::memcpy(sigs[i].address, addr.data(), 20);
      sigs[i].has_address = 1;
    }
  }
  *n_sigs = (uint32_t)sg->a.size();
  return TMX_OK;
}

}  // extern "C"
