/* A compiled (non-Python) host driving libtmx through the C ABI only -- the closest stand-in available here for the reference's Rust
 * hint body: reads the reference-format fixture JSON, converts it with the library's codec, runs the skip witness on
 * the GPU and prints the Level-0 output and verdict.  Built and run by tests/test_c_host.py (-m gpu).
 *   usage: skip_host <fixture_dir> <trusted_block> <trusted_hash_hex> <target_block> <n_max> <chain_id>                      */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

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

int main(int argc, char** argv) {
  if (argc != 7) { fprintf(stderr, "usage\n"); return 2; }
  const char* dir = argv[1];
  unsigned long trusted = strtoul(argv[2], 0, 10), target = strtoul(argv[4], 0, 10);
  uint32_t n = (uint32_t)strtoul(argv[5], 0, 10);
  uint8_t hash[32];
  for (int i = 0; i < 32; i++) { unsigned v; sscanf(argv[3] + 2 * i, "%2x", &v); hash[i] = (uint8_t)v; }

  tmx_config cfg;
  memset(&cfg, 0, sizeof cfg);
  cfg.n_max = n; cfg.chain_id_len = (uint32_t)strlen(argv[6]); memcpy(cfg.chain_id, argv[6], cfg.chain_id_len);
  cfg.skip_max = 100800; cfg.device = 0; cfg.max_batch = 1;
  tmx_ctx* ctx = 0;
  int32_t st = tmx_ctx_create(&cfg, &ctx);
  if (st) { fprintf(stderr, "tmx_ctx_create: %s -- %s\n", tmx_status_str(st), tmx_last_error(ctx)); return 1; }

  tmx_proof_rec proof;
  tmx_validator_rec* tg = (tmx_validator_rec*)calloc(n, sizeof *tg);
  tmx_hashfield_rec* tr = (tmx_hashfield_rec*)calloc(n, sizeof *tr);
  char *tc = slurp(dir, trusted, "commit.json"), *tv = slurp(dir, trusted, "validators_1.json");
  char *gc = slurp(dir, target, "commit.json"), *gv = slurp(dir, target, "validators_1.json");
  st = tmx_skip_inputs_from_json(tc, tv, gc, gv, n, trusted, hash, target, &proof, tg, tr);
  if (st) { fprintf(stderr, "codec: %s\n", tmx_status_str(st)); return 1; }

  uint64_t count = tmx_elem_count(TMX_KIND_SKIP, n);
  uint64_t* elems = (uint64_t*)malloc(count * 8);
  tmx_report rep;
  st = tmx_skip_witness(ctx, &proof, tg, tr, elems, count, &rep);
  if (st) { fprintf(stderr, "tmx_skip_witness: %s -- %s\n", tmx_status_str(st), tmx_last_error(ctx)); return 1; }
  uint64_t sum = 0;
  for (uint64_t i = 0; i < count; i++) sum = sum * 1099511628211ull + elems[i];  /* order-sensitive checksum of the row */
  printf("header ");
  for (int i = 0; i < 32; i++) printf("%02x", rep.header[i]);
  printf("\nall_ok %u fail_mask %u first_bad_sig %d elems %llu checksum %llu\n", rep.all_ok, rep.fail_mask, rep.first_bad_sig,
         (unsigned long long)count, (unsigned long long)sum);
  tmx_ctx_destroy(ctx);
  return 0;
}
