/* A compiled (non-Python) host driving libtmx through the C ABI only -- the closest stand-in available here for the reference's Rust hint
 * bodies (SkipOffchainInputs::hint, reference circuits/skip.rs:64-102; StepOffchainInputs::hint, circuits/step.rs:56-89).
 * Reads the reference-format fixture JSON, converts it with the library's codec, runs the witness on the GPU and writes the FULL row to a
 * file (tests/test_c_host.py compares it element by element with the oracle's) and the report to stdout.  Modes:
 *   skip    <fixtures> <n_max> <chain_id> <out.bin> <trusted_block> <trusted_hash_hex> <target_block>      tmx_skip_witness, u64 row
 *   step    <fixtures> <n_max> <chain_id> <out.bin> <prev_block> <prev_hash_hex>                           tmx_step_witness, u64 row
 *   hint32  <fixtures> <n_max> <chain_id> <out.bin> <trusted_block> <trusted_hash_hex> <target_block>      tmx_witness_batch_opts(TMX_SEC_HINT,
 *                                                                                                           TMX_OUT_U32): the hint section as u32
 *   value   <fixtures> <n_max> <chain_id> <out.bin> <trusted_block> <trusted_hash_hex> <target_block>      tmx_skip_inputs_value(TMX_SEC_ALL) into
 *           page-locked memory from tmx_host_alloc: the typed SkipInputs value + derived values; prints fields READ BY NAME from the structs
 *           (what a hint body assigns into VerifySkipStruct, reference circuits/skip.rs:85-98) and the host-to-host ms per call of 50 warm calls
 *   stepvalue <fixtures> <n_max> <chain_id> <out.bin> <prev_block> <prev_hash_hex>                         tmx_step_inputs_value(TMX_SEC_ALL)
 *   threads <fixtures> <n_max> <chain_id> <out.bin> <trusted_block> <trusted_hash_hex> <target_block> <iters>
 *           two host threads, each with a context of its own, run the same skip witness `iters` times concurrently (the calling pattern of
 *           a tokio host with several hint workers, reference circuits/skip.rs:37-44: contexts of one device share the library's internal
 *           streams); every row of every iteration must equal the first one; prints ms per call alone and with both threads running.
 *   sharded <fixtures> <n_max> <chain_id> <out.bin> <trusted_block> <trusted_hash_hex> <target_block>
 *           the multi-GPU entry points from a compiled host, on the ranks this box has (one): tmx_comm_unique_id -> tmx_comm_create with
 *           the id (a real RCCL communicator; librccl is dlopen'ed by libtmx, this program does not link it) ->
 *           tmx_witness_validator_sharded_device (lanes split, the lane records exchanged through RCCL, finish) on device buffers this
 *           program owns, then tmx_witness_batch_sharded_device with the row exchange: both rows must be the same; writes the row.     */
#include <hip/hip_runtime_api.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "tmx.h"

static char* slurp(const char* dir, unsigned long height, const char* name) {
  char path[1024];
  snprintf(path, sizeof path, "%s/%lu/%s", dir, height, name);
  FILE* f = fopen(path, "rb");
  if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
  fseek(f, 0, SEEK_END);
  long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  char* buf = (char*)malloc((size_t)n + 1);
  if (fread(buf, 1, (size_t)n, f) != (size_t)n) exit(2);
  buf[n] = 0;
  fclose(f);
  return buf;
}
static void unhex32(const char* s, uint8_t out[32]) {
  for (int i = 0; i < 32; i++) { unsigned v; sscanf(s + 2 * i, "%2x", &v); out[i] = (uint8_t)v; }
}
static double now_ms(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return 1e3 * ts.tv_sec + 1e-6 * ts.tv_nsec;
}
static tmx_ctx* make_ctx(uint32_t n, const char* chain) {
  tmx_config cfg;
  memset(&cfg, 0, sizeof cfg);
  cfg.n_max = n; cfg.chain_id_len = (uint32_t)strlen(chain); memcpy(cfg.chain_id, chain, cfg.chain_id_len);
  cfg.skip_max = 100800; cfg.device = 0; cfg.max_batch = 1;
  tmx_ctx* ctx = 0;
  int32_t st = tmx_ctx_create(&cfg, &ctx);
  if (st) { fprintf(stderr, "tmx_ctx_create: %s -- %s\n", tmx_status_str(st), tmx_last_error(ctx)); exit(1); }
  return ctx;
}
static void print_report(const tmx_report* rep, unsigned long long count) {
  printf("header ");
  for (int i = 0; i < 32; i++) printf("%02x", rep->header[i]);
  printf("\nall_ok %u fail_mask %u first_bad_sig %d gt_target %u gt_trusted %u dist_ok %u precond %u elems %llu\n", rep->all_ok, rep->fail_mask,
         rep->first_bad_sig, rep->gt_target, rep->gt_trusted, rep->dist_ok, rep->precond, count);
}
static void dump(const char* path, const void* p, size_t bytes) {
  FILE* f = fopen(path, "wb");
  if (!f || fwrite(p, 1, bytes, f) != bytes) { fprintf(stderr, "cannot write %s\n", path); exit(2); }
  fclose(f);
}

struct skip_in {
  uint32_t n;
  tmx_proof_rec proof;
  tmx_validator_rec* tg;
  tmx_hashfield_rec* tr;
};
static void load_skip(struct skip_in* in, const char* dir, uint32_t n, unsigned long trusted, const uint8_t hash[32], unsigned long target) {
  in->n = n;
  in->tg = (tmx_validator_rec*)calloc(n, sizeof *in->tg);
  in->tr = (tmx_hashfield_rec*)calloc(n, sizeof *in->tr);
  char *tc = slurp(dir, trusted, "commit.json"), *tv = slurp(dir, trusted, "validators_1.json");
  char *gc = slurp(dir, target, "commit.json"), *gv = slurp(dir, target, "validators_1.json");
  int32_t st = tmx_skip_inputs_from_json(tc, tv, gc, gv, n, trusted, hash, target, &in->proof, in->tg, in->tr);
  if (st) { fprintf(stderr, "codec: %s\n", tmx_status_str(st)); exit(1); }
  free(tc); free(tv); free(gc); free(gv);
}

struct worker {
  const struct skip_in* in;
  const char* chain;
  int iters;
  uint64_t count;
  uint64_t* first;  /* row of the first call */
  int mismatches;
  double ms_per_call;
  pthread_barrier_t* start;
};
static void* work(void* arg) {
  struct worker* w = (struct worker*)arg;
  tmx_ctx* ctx = make_ctx(w->in->n, w->chain);
  uint64_t* row = (uint64_t*)malloc(w->count * 8);
  tmx_report rep;
  w->first = (uint64_t*)malloc(w->count * 8);
  int32_t st = tmx_skip_witness(ctx, &w->in->proof, w->in->tg, w->in->tr, w->first, w->count, &rep);  /* cold call (also warms the key cache) */
  if (st || !rep.all_ok) { fprintf(stderr, "worker: %s -- %s\n", tmx_status_str(st), tmx_last_error(ctx)); exit(1); }
  if (w->start) pthread_barrier_wait(w->start);
  const double t0 = now_ms();
  for (int i = 0; i < w->iters; i++) {
    st = tmx_skip_witness(ctx, &w->in->proof, w->in->tg, w->in->tr, row, w->count, &rep);
    if (st || !rep.all_ok || memcmp(row, w->first, w->count * 8) != 0) w->mismatches++;
  }
  w->ms_per_call = (now_ms() - t0) / w->iters;
  free(row);
  tmx_ctx_destroy(ctx);
  return 0;
}

int main(int argc, char** argv) {
  if (argc < 7) { fprintf(stderr, "usage: see the header of this file\n"); return 2; }
  const char *mode = argv[1], *dir = argv[2], *chain = argv[4], *out = argv[5];
  const uint32_t n = (uint32_t)strtoul(argv[3], 0, 10);
  const unsigned long block = strtoul(argv[6], 0, 10);
  uint8_t hash[32];
  unhex32(argv[7], hash);
  tmx_report rep;
  if (!strcmp(mode, "stepvalue")) {
    tmx_ctx* ctx = make_ctx(n, chain);
    tmx_proof_rec proof;
    tmx_validator_rec* tg = (tmx_validator_rec*)calloc(n, sizeof *tg);
    char *pc = slurp(dir, block, "commit.json"), *nc = slurp(dir, block + 1, "commit.json"), *nv = slurp(dir, block + 1, "validators_1.json");
    int32_t st = tmx_step_inputs_from_json(pc, nc, nv, n, block, hash, &proof, tg);
    if (st) { fprintf(stderr, "codec: %s\n", tmx_status_str(st)); return 1; }
    tmx_value_layout lay;
    if (tmx_value_layout_of(TMX_KIND_STEP, n, TMX_SEC_ALL, &lay)) return 1;
    uint8_t* val = (uint8_t*)tmx_host_alloc(ctx, lay.bytes);
    if (!val) { fprintf(stderr, "tmx_host_alloc failed\n"); return 1; }
    st = tmx_step_inputs_value(ctx, &proof, tg, TMX_SEC_ALL, val, lay.bytes);
    if (st) { fprintf(stderr, "tmx_step_inputs_value: %s -- %s\n", tmx_status_str(st), tmx_last_error(ctx)); return 1; }
    const tmx_step_inputs_fixed* f = (const tmx_step_inputs_fixed*)val;
    const tmx_validator_value* vals = (const tmx_validator_value*)(val + lay.off_validators);
    print_report(&f->report, lay.bytes);
    printf("nb_validators %u round %llu height %llu enc_height_len %u last_block_id_leaf0 %02x%02x signed0 %u power0 %llu\n", f->nb_validators,
           (unsigned long long)f->round, (unsigned long long)f->next_block_height_proof.height, f->next_block_height_proof.enc_height_byte_length,
           f->next_block_last_block_id_proof.leaf[0], f->next_block_last_block_id_proof.leaf[1], vals[0].signed_, (unsigned long long)vals[0].voting_power);
    dump(out, val, lay.bytes);
    const int same = memcmp(f->next_header, f->report.header, 32) == 0;
    tmx_host_free(ctx, val);
    tmx_ctx_destroy(ctx);
    return same ? 0 : 1;
  }
  if (!strcmp(mode, "step")) {
    tmx_ctx* ctx = make_ctx(n, chain);
    tmx_proof_rec proof;
    tmx_validator_rec* tg = (tmx_validator_rec*)calloc(n, sizeof *tg);
    char *pc = slurp(dir, block, "commit.json"), *nc = slurp(dir, block + 1, "commit.json"), *nv = slurp(dir, block + 1, "validators_1.json");
    int32_t st = tmx_step_inputs_from_json(pc, nc, nv, n, block, hash, &proof, tg);
    if (st) { fprintf(stderr, "codec: %s\n", tmx_status_str(st)); return 1; }
    const uint64_t count = tmx_elem_count(TMX_KIND_STEP, n);
    uint64_t* elems = (uint64_t*)malloc(count * 8);
    st = tmx_step_witness(ctx, &proof, tg, elems, count, &rep);
    if (st) { fprintf(stderr, "tmx_step_witness: %s -- %s\n", tmx_status_str(st), tmx_last_error(ctx)); return 1; }
    print_report(&rep, count);
    dump(out, elems, count * 8);
    tmx_ctx_destroy(ctx);
    return 0;
  }
  if (argc < 9) { fprintf(stderr, "usage: see the header of this file\n"); return 2; }
  const unsigned long target = strtoul(argv[8], 0, 10);
  struct skip_in in;
  load_skip(&in, dir, n, block, hash, target);
  const uint64_t count = tmx_elem_count(TMX_KIND_SKIP, n);
  if (!strcmp(mode, "skip")) {
    tmx_ctx* ctx = make_ctx(n, chain);
    uint64_t* elems = (uint64_t*)malloc(count * 8);
    int32_t st = tmx_skip_witness(ctx, &in.proof, in.tg, in.tr, elems, count, &rep);
    if (st) { fprintf(stderr, "tmx_skip_witness: %s -- %s\n", tmx_status_str(st), tmx_last_error(ctx)); return 1; }
    print_report(&rep, count);
    dump(out, elems, count * 8);
    tmx_key_cache_info ki;
    st = tmx_key_cache_stats(ctx, &ki);
    if (st) return 1;
    printf("key_cache enabled %u resident %u capacity %u last_new %u last_hit_lanes %u\n", ki.enabled, ki.resident_keys, ki.capacity_keys, ki.last_new_keys,
           ki.last_hit_lanes);
    tmx_ctx_destroy(ctx);
    return 0;
  }
  if (!strcmp(mode, "value")) {
    tmx_ctx* ctx = make_ctx(n, chain);
    tmx_value_layout lay;
    if (tmx_value_layout_of(TMX_KIND_SKIP, n, TMX_SEC_ALL, &lay)) return 1;
    uint8_t* val = (uint8_t*)tmx_host_alloc(ctx, lay.bytes);
    if (!val) { fprintf(stderr, "tmx_host_alloc failed\n"); return 1; }
    int32_t st = tmx_skip_inputs_value(ctx, &in.proof, in.tg, in.tr, TMX_SEC_ALL, val, lay.bytes);   /* cold: also makes the keys resident */
    if (st) { fprintf(stderr, "tmx_skip_inputs_value: %s -- %s\n", tmx_status_str(st), tmx_last_error(ctx)); return 1; }
    const double t0 = now_ms();
    for (int i = 0; i < 50 && !st; i++) st = tmx_skip_inputs_value(ctx, &in.proof, in.tg, in.tr, TMX_SEC_ALL, val, lay.bytes);
    const double ms = (now_ms() - t0) / 50;
    if (st) { fprintf(stderr, "tmx_skip_inputs_value: %s -- %s\n", tmx_status_str(st), tmx_last_error(ctx)); return 1; }
    /* the fields a hint body assigns into VerifySkipStruct (circuits/skip.rs:85-98), read by name */
    const tmx_skip_inputs_fixed* f = (const tmx_skip_inputs_fixed*)val;
    const tmx_validator_value* vals = (const tmx_validator_value*)(val + lay.off_validators);
    const tmx_hashfield_value* hfs = (const tmx_hashfield_value*)(val + lay.off_hashfields);
    print_report(&f->report, lay.bytes);
    printf("nb_target %u nb_trusted %u round %llu height %llu enc_chain_id_len %u chain_id %.*s signed0 %u power0 %llu trusted_power0 %llu ms_per_call %.4f\n",
           f->nb_target_validators, f->nb_trusted_validators, (unsigned long long)f->round, (unsigned long long)f->target_block_height_proof.height,
           f->target_block_chain_id_proof.enc_chain_id_byte_length, (int)f->target_block_chain_id_proof.enc_chain_id_byte_length - 2,
           (const char*)f->target_block_chain_id_proof.chain_id + 2, vals[0].signed_, (unsigned long long)vals[0].voting_power,
           (unsigned long long)hfs[0].voting_power, ms);
    dump(out, val, lay.bytes);
    const int same = memcmp(f->target_header, f->report.header, 32) == 0 && memcmp(f->trusted_header, hash, 32) == 0;
    tmx_host_free(ctx, val);
    tmx_ctx_destroy(ctx);
    return same ? 0 : 1;
  }
  if (!strcmp(mode, "hint32")) {
    tmx_ctx* ctx = make_ctx(n, chain);
    const uint64_t row = tmx_out_row_elems(TMX_KIND_SKIP, n, TMX_SEC_HINT);
    uint32_t* elems = (uint32_t*)malloc(row * 4);
    int32_t st = tmx_witness_batch_opts(ctx, TMX_KIND_SKIP, 1, &in.proof, in.tg, in.tr, TMX_SEC_HINT, TMX_OUT_U32, elems, row * 4, &rep);
    if (st) { fprintf(stderr, "tmx_witness_batch_opts: %s -- %s\n", tmx_status_str(st), tmx_last_error(ctx)); return 1; }
    print_report(&rep, row);
    dump(out, elems, row * 4);
    tmx_ctx_destroy(ctx);
    return 0;
  }
  if (!strcmp(mode, "threads")) {
    const int iters = argc > 9 ? atoi(argv[9]) : 50;
    struct worker solo = {&in, chain, iters, count, 0, 0, 0, 0};
    work(&solo);  /* one thread alone */
    pthread_barrier_t bar;
    pthread_barrier_init(&bar, 0, 2);
    struct worker w[2] = {{&in, chain, iters, count, 0, 0, 0, &bar}, {&in, chain, iters, count, 0, 0, 0, &bar}};
    pthread_t th[2];
    for (int k = 0; k < 2; k++) pthread_create(&th[k], 0, work, &w[k]);
    for (int k = 0; k < 2; k++) pthread_join(th[k], 0);
    const int same = memcmp(w[0].first, w[1].first, count * 8) == 0 && memcmp(w[0].first, solo.first, count * 8) == 0;
    printf("threads mismatches %d %d %d rows_equal %d ms_alone %.4f ms_two_threads %.4f %.4f\n", solo.mismatches, w[0].mismatches, w[1].mismatches, same,
           solo.ms_per_call, w[0].ms_per_call, w[1].ms_per_call);
    dump(out, w[1].first, count * 8);
    return (solo.mismatches || w[0].mismatches || w[1].mismatches || !same) ? 1 : 0;
  }
  if (!strcmp(mode, "sharded")) {
    tmx_ctx* ctx = make_ctx(n, chain);
    uint8_t id[TMX_UNIQUE_ID_BYTES];
    int32_t st = tmx_comm_unique_id(id);
    if (st) { fprintf(stderr, "tmx_comm_unique_id: %s\n", tmx_status_str(st)); return 1; }
    st = tmx_comm_create(ctx, id, 0, 1);
    if (st) { fprintf(stderr, "tmx_comm_create: %s -- %s\n", tmx_status_str(st), tmx_last_error(ctx)); return 1; }
    uint32_t rank = 9, world = 9;
    tmx_comm_info(ctx, &rank, &world);
    uint64_t lo = 9, hi = 9;
    tmx_shard_range(n, rank, world, &lo, &hi);
    const uint64_t stride = tmx_elem_stride(TMX_KIND_SKIP, n);
    void *d_p = 0, *d_t = 0, *d_r = 0, *d_o = 0, *d_o2 = 0, *d_rep = 0;
    hipStream_t s = 0;
    if (hipStreamCreate(&s) || hipMalloc(&d_p, sizeof in.proof) || hipMalloc(&d_t, n * sizeof *in.tg) || hipMalloc(&d_r, n * sizeof *in.tr) ||
        hipMalloc(&d_o, stride * 8) || hipMalloc(&d_o2, stride * 8) || hipMalloc(&d_rep, sizeof rep)) { fprintf(stderr, "hipMalloc failed\n"); return 1; }
    hipMemcpy(d_p, &in.proof, sizeof in.proof, hipMemcpyHostToDevice);
    hipMemcpy(d_t, in.tg, n * sizeof *in.tg, hipMemcpyHostToDevice);
    hipMemcpy(d_r, in.tr, n * sizeof *in.tr, hipMemcpyHostToDevice);
    st = tmx_witness_validator_sharded_device(ctx, TMX_KIND_SKIP, 1, d_p, d_t, d_r, d_o, d_rep, s);
    if (st) { fprintf(stderr, "tmx_witness_validator_sharded_device: %s -- %s\n", tmx_status_str(st), tmx_last_error(ctx)); return 1; }
    st = tmx_witness_batch_sharded_device(ctx, TMX_KIND_SKIP, 1, d_p, d_t, d_r, d_o2, d_rep, 1, s);
    if (st) { fprintf(stderr, "tmx_witness_batch_sharded_device: %s -- %s\n", tmx_status_str(st), tmx_last_error(ctx)); return 1; }
    if (hipStreamSynchronize(s)) { fprintf(stderr, "stream failed\n"); return 1; }
    uint64_t *a = (uint64_t*)malloc(count * 8), *b = (uint64_t*)malloc(count * 8);
    hipMemcpy(a, d_o, count * 8, hipMemcpyDeviceToHost);
    hipMemcpy(b, d_o2, count * 8, hipMemcpyDeviceToHost);
    hipMemcpy(&rep, d_rep, sizeof rep, hipMemcpyDeviceToHost);
    print_report(&rep, count);
    printf("rank %u world %u shard_lo %llu shard_hi %llu rows_equal %d\n", rank, world, (unsigned long long)lo, (unsigned long long)hi, memcmp(a, b, count * 8) == 0);
    dump(out, a, count * 8);
    st = tmx_comm_destroy(ctx);
    tmx_ctx_destroy(ctx);
    return st ? 1 : (memcmp(a, b, count * 8) == 0 ? 0 : 1);
  }
  fprintf(stderr, "unknown mode %s\n", mode);
  return 2;
}
