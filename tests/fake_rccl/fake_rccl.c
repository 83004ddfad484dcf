/* TEST INFRASTRUCTURE ONLY -- a stand-in for librccl that lets TWO (or more) processes sharing ONE GPU run libtmx's multi-GPU entry
 * points (tmx_comm_create, tmx_witness_batch_sharded_device, tmx_witness_validator_sharded_device, tmx_trace_rows_sharded_device) at
 * world > 1 on a single-GPU box.  Real RCCL refuses two ranks on one device ("duplicate GPU"), and the boxes this repo is built and
 * judged on have one GPU, so without this the world > 1 code paths of api.cpp (exchange_slices: ncclAllGather for equal shards, grouped
 * ncclBroadcast for ragged ones; the shard arithmetic; the in-place slices) would first execute on somebody's 8-GPU node.
 *
 * Selected with TMX_RCCL_LIB=<this .so> (libtmx binds RCCL with dlopen; an explicit TMX_RCCL_LIB wins over a librccl the process has
 * already loaded).  It implements exactly the ten symbols libtmx binds (ncclCommAbort / ncclCommGetAsyncError since round 6), with NCCL's signatures and semantics as far as libtmx uses
 * them: byte counts, in-place operation, group calls executed at ncclGroupEnd in issue order.  Data moves through a file-backed shared
 * mapping (the "wire"): the owner of a slice copies it device -> wire, everybody meets at a barrier in the mapping, the others copy
 * wire -> device.  Every call is synchronous with respect to the host (it synchronises the stream it was given first), which is a legal
 * implementation of a stream-ordered collective.  No HIP runtime is linked: like libtmx it runs on the one the process already has.
 * Nothing of the product links, loads or names this file; only tests/ do. */
#define _GNU_SOURCE
#include <errno.h>
#include <fcntl.h>
#include <sched.h>
#include <stdatomic.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

/* the HIP runtime of the process (resolved at load time from the global scope, as in libtmx) */
typedef int hipError_t;
typedef void* hipStream_t;
extern hipError_t hipMemcpy(void* dst, const void* src, size_t n, int kind);
extern hipError_t hipStreamSynchronize(hipStream_t s);
enum { H2D = 1, D2H = 2, D2D = 3 };

#define WIRE_BYTES ((size_t)32 << 20)
#define BARRIER_TIMEOUT_S 120.0
typedef struct {
  _Atomic uint32_t arrived;     /* ranks that reached the current barrier */
  _Atomic uint32_t generation;  /* bumped by the last arriver */
  _Atomic uint32_t joined;      /* ranks that mapped the file (ncclCommInitRank) */
  _Atomic uint32_t failed;      /* a rank hit an error: everybody bails out of its barriers */
  uint8_t pad[4096 - 16];
  uint8_t wire[];
} shared_t;

typedef struct { char internal[128]; } ncclUniqueId;
typedef struct comm {
  shared_t* sh;
  int rank, world;
  char path[128];
} comm_t;
typedef comm_t* ncclComm_t;

enum { OP_BCAST, OP_ALLGATHER };
typedef struct { int kind; const void* send; void* recv; size_t bytes; int root; comm_t* c; hipStream_t s; } op_t;
static __thread op_t g_ops[64];
static __thread int g_n_ops, g_depth;

static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

static int barrier(comm_t* c) {
  shared_t* sh = c->sh;
  const uint32_t gen = atomic_load(&sh->generation);
  if (atomic_fetch_add(&sh->arrived, 1) + 1 == (uint32_t)c->world) {
    atomic_store(&sh->arrived, 0);
    atomic_fetch_add(&sh->generation, 1);
    return 0;
  }
  const double t0 = now_s();
  while (atomic_load(&sh->generation) == gen) {
    if (atomic_load(&sh->failed)) return 6;
    if (now_s() - t0 > BARRIER_TIMEOUT_S) { atomic_store(&sh->failed, 1); return 6; }   /* never hang a GPU box: fail the collective */
    sched_yield();
  }
  return 0;
}

static const size_t type_bytes[] = {1, 1, 4, 4, 8, 8, 2, 4, 8, 2};   /* ncclInt8 .. ncclBfloat16 */

/* the slice `bytes` long that `owner` holds at `src` becomes `dst` on every rank */
static int move_slice(comm_t* c, int owner, const void* src, void* dst, size_t bytes) {
  for (size_t off = 0; off < bytes; off += WIRE_BYTES) {
    const size_t n = bytes - off < WIRE_BYTES ? bytes - off : WIRE_BYTES;
    int rc = 0;
    if (c->rank == owner) {
      if (hipMemcpy(c->sh->wire, (const uint8_t*)src + off, n, D2H)) rc = 1;
      if (!rc && src != dst && hipMemcpy((uint8_t*)dst + off, (const uint8_t*)src + off, n, D2D)) rc = 1;
    }
    if (rc) atomic_store(&c->sh->failed, 1);
    if ((rc = barrier(c))) return rc;
    if (c->rank != owner && hipMemcpy((uint8_t*)dst + off, c->sh->wire, n, H2D)) { atomic_store(&c->sh->failed, 1); rc = 1; }
    int rb = barrier(c);   /* the wire is free again */
    if (rc || rb) return rc ? rc : rb;
  }
  return 0;
}

static int run_op(const op_t* o) {
  comm_t* c = o->c;
  if (hipStreamSynchronize(o->s)) return 1;   /* everything enqueued in front of the collective is done */
  if (o->kind == OP_BCAST) return move_slice(c, o->root, o->send, o->recv, o->bytes);
  for (int r = 0; r < c->world; r++) {
    uint8_t* dst = (uint8_t*)o->recv + (size_t)r * o->bytes;
    int rc = move_slice(c, r, o->send, dst, o->bytes);   /* (only rank r reads o->send) */
    if (rc) return rc;
  }
  return 0;
}

static int submit(op_t o) {
  if (g_depth == 0) return run_op(&o);
  if (g_n_ops == 64) return 5;
  g_ops[g_n_ops++] = o;
  return 0;
}

int ncclGetUniqueId(ncclUniqueId* id) {
  static _Atomic int counter;
  const char* dir = getenv("FAKE_RCCL_DIR");
  if (!dir || !dir[0]) dir = getenv("TMPDIR");
  if (!dir || !dir[0]) dir = "/tmp";
  memset(id, 0, sizeof *id);
  snprintf(id->internal, sizeof id->internal, "%s/fake_rccl_%ld_%d_%ld", dir, (long)getpid(), atomic_fetch_add(&counter, 1), (long)time(0));
  const int fd = open(id->internal, O_CREAT | O_EXCL | O_RDWR, 0600);
  if (fd < 0) return 2;
  const int rc = ftruncate(fd, (off_t)(sizeof(shared_t) + WIRE_BYTES));   /* zero-filled: counters start at 0 */
  close(fd);
  return rc ? 2 : 0;
}

int ncclCommInitRank(ncclComm_t* out, int nranks, ncclUniqueId id, int rank) {
  if (!out || nranks < 1 || rank < 0 || rank >= nranks) return 4;
  id.internal[sizeof id.internal - 1] = 0;
  const int fd = open(id.internal, O_RDWR);
  if (fd < 0) return 2;
  void* p = mmap(0, sizeof(shared_t) + WIRE_BYTES, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) return 2;
  comm_t* c = (comm_t*)calloc(1, sizeof *c);
  c->sh = (shared_t*)p; c->rank = rank; c->world = nranks;
  snprintf(c->path, sizeof c->path, "%s", id.internal);
  atomic_fetch_add(&c->sh->joined, 1);
  const double t0 = now_s();
  while ((int)atomic_load(&c->sh->joined) < nranks) {   /* like ncclCommInitRank: returns when every rank has joined */
    if (now_s() - t0 > BARRIER_TIMEOUT_S) { munmap(p, sizeof(shared_t) + WIRE_BYTES); free(c); return 6; }
    sched_yield();
  }
  *out = c;
  return 0;
}

int ncclCommDestroy(ncclComm_t c) {
  if (!c) return 4;
  if (c->rank == 0) unlink(c->path);   /* the mapping stays valid for the ranks that still hold it */
  munmap(c->sh, sizeof(shared_t) + WIRE_BYTES);
  free(c);
  return 0;
}

/* ncclCommAbort: this rank leaves for good; every barrier of the communicator fails from now on (what a peer of an aborted NCCL
 * communicator sees as ncclRemoteError / ncclSystemError).  ncclCommGetAsyncError: that same flag, without entering a collective. */
int ncclCommAbort(ncclComm_t c) {
  if (!c) return 4;
  atomic_store(&c->sh->failed, 1);
  return ncclCommDestroy(c);
}
int ncclCommGetAsyncError(ncclComm_t c, int* async_error) {
  if (!c || !async_error) return 4;
  *async_error = atomic_load(&c->sh->failed) ? 6 : 0;
  return 0;
}

int ncclBroadcast(const void* send, void* recv, size_t count, int datatype, int root, ncclComm_t c, hipStream_t s) {
  if (!c || datatype < 0 || datatype > 9 || root < 0 || root >= c->world) return 4;
  op_t o = {OP_BCAST, send, recv, count * type_bytes[datatype], root, c, s};
  return submit(o);
}

int ncclAllGather(const void* send, void* recv, size_t sendcount, int datatype, ncclComm_t c, hipStream_t s) {
  if (!c || datatype < 0 || datatype > 9) return 4;
  op_t o = {OP_ALLGATHER, send, recv, sendcount * type_bytes[datatype], 0, c, s};
  return submit(o);
}

int ncclGroupStart(void) { g_depth++; return 0; }
int ncclGroupEnd(void) {
  if (g_depth == 0) return 5;
  if (--g_depth) return 0;
  int rc = 0;
  for (int i = 0; i < g_n_ops && !rc; i++) rc = run_op(&g_ops[i]);
  g_n_ops = 0;
  return rc;
}

const char* ncclGetErrorString(int rc) {
  switch (rc) {
    case 0: return "fake_rccl: success";
    case 1: return "fake_rccl: HIP call failed";
    case 2: return "fake_rccl: cannot create / map the wire file";
    case 4: return "fake_rccl: invalid argument";
    case 5: return "fake_rccl: invalid usage (group nesting / too many grouped calls)";
    case 6: return "fake_rccl: a rank failed or a barrier timed out";
    default: return "fake_rccl: error";
  }
}
