// Where does the dispatcher put the waves of a small grid?  A latency-bound kernel of G single-wave workgroups (k_ed_hash, k_ed_fin: 512
// waves on a chip of 1024 SIMDs) runs at the lone-wave issue rate only if every wave has a SIMD to itself.
//   hipcc --offload-arch=gfx950 -O2 -o placement placement.hip && ./placement
// Every wave records HW_ID (wave slot, SIMD, CU, SH, SE) and XCC_ID and then runs a dependent VALU chain long enough (~50 us) for the whole
// grid to be resident at once; the host prints how many waves shared a SIMD / a CU, and the kernel time against one wave alone.
// Grids: G workgroups x T threads for the shapes the chain kernels could take (512 x 64, 256 x 128, 128 x 256, 1024 x 64 ...).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

#define CK(x)                                                                              \
  do {                                                                                     \
    hipError_t e_ = (x);                                                                   \
    if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } \
  } while (0)

__global__ void k_place(unsigned* out, int iters, unsigned seed) {
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  unsigned a = seed + threadIdx.x;
  const unsigned b = seed | 1u;
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();  // 100 MHz, chip-wide
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) asm volatile(".rept 64\nv_mad_u32_u24 %0, %0, %1, %0\n.endr" : "+v"(a) : "v"(b));
  const unsigned long long t1 = __builtin_readcyclecounter();
  const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if ((threadIdx.x & 63) == 0) {
    out[4 * wave] = hw; out[4 * wave + 1] = xcc | ((unsigned)r0 << 4); out[4 * wave + 2] = (unsigned)(t1 - t0); out[4 * wave + 3] = a;
  }
}

// Straight-line code against a loop: the same dependent chain as ONE run of N instructions (N x 8 bytes of code: 12800 = 100 KB, more than
// the 64-KB instruction cache two CUs share) and as a 64-instruction loop body.  k_ed_hash / k_ed_fin / k_proof are 94 / 132 / 190 KB of
// straight-line code that every wave runs through once.
#define STRAIGHT(NAME, N)                                                                                     \
  __global__ void NAME(unsigned* out, int reps, unsigned seed) {                                              \
    unsigned a = seed + threadIdx.x;                                                                          \
    const unsigned b = seed | 1u;                                                                             \
    const unsigned long long t0 = __builtin_readcyclecounter();                                               \
    for (int it = 0; it < reps; it++) asm volatile(".rept " #N "\nv_mad_u32_u24 %0, %0, %1, %0\n.endr" : "+v"(a) : "v"(b)); \
    const unsigned long long t1 = __builtin_readcyclecounter();                                               \
    const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;                                       \
    if ((threadIdx.x & 63) == 0) { out[4 * wave] = 0; out[4 * wave + 1] = 0; out[4 * wave + 2] = (unsigned)(t1 - t0); out[4 * wave + 3] = a; } \
  }
STRAIGHT(k_s64, 64)
STRAIGHT(k_s1600, 1600)
STRAIGHT(k_s6400, 6400)
// (a loop around more than 128 KB of code does not assemble -- the back edge exceeds simm16 -- and the chain kernels run their code once anyway)
#define ONCE(NAME, N, COPIES)                                                                                 \
  __global__ void NAME(unsigned* out, int, unsigned seed) {                                                   \
    unsigned a = seed + threadIdx.x;                                                                          \
    const unsigned b = seed | 1u;                                                                             \
    const unsigned long long t0 = __builtin_readcyclecounter();                                               \
    asm volatile(".rept " #N "\nv_mad_u32_u24 %0, %0, %1, %0\n.endr" : "+v"(a) : "v"(b));                       \
    if (COPIES > 1) asm volatile(".rept " #N "\nv_mad_u32_u24 %0, %0, %1, %0\n.endr" : "+v"(a) : "v"(b));        \
    const unsigned long long t1 = __builtin_readcyclecounter();                                               \
    const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;                                       \
    if ((threadIdx.x & 63) == 0) { out[4 * wave] = 0; out[4 * wave + 1] = 0; out[4 * wave + 2] = (unsigned)(t1 - t0); out[4 * wave + 3] = a; } \
  }
ONCE(k_s12800, 12800, 2)
ONCE(k_s25600, 25600, 1)
template <typename K>
static void run_straight(const char* name, K kern, int n, unsigned* d, int G) {
  const int reps = 25600 / n;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kern, dim3(G), dim3(64), 0, 0, d, reps, 12345u);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL(kern, dim3(G), dim3(64), 0, 0, d, reps, 12345u);
  CK(hipEventRecord(e1, 0));
  CK(hipDeviceSynchronize());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned> h(4 * G);
  CK(hipMemcpy(h.data(), d, 4 * 4 * G, hipMemcpyDeviceToHost));
  unsigned long long cyc = 0;
  for (int w = 0; w < G; w++) cyc += h[4 * w + 2];
  std::printf("%-10s %5d instructions (%4d KB) x %3d, grid %4d x 64: %.1f us, %.2f cycles per instruction\n", name, n, n * 8 / 1024, reps, G, ms * 1e3,
              (double)cyc / G / 25600.0);
}


// (round 5) Is workgroup b of a dispatch on XCD b % 8 also when other queues are dispatching, and what do k of the 8 XCDs write?  A kernel whose
// workgroups leave at once unless (mask >> (blockIdx.x & 7)) & 1 is then confined to those XCDs without a CU-masked stream (which would have
// no priority and could not be the caller's stream).
__global__ void k_xcd_check(unsigned* bad, unsigned* per_xcd) {
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  xcc &= 0xf;
  if (threadIdx.x == 0) {
    if (xcc != (blockIdx.x & 7u)) atomicAdd(bad, 1u);
    atomicAdd(&per_xcd[(blockIdx.x & 7u) * 8 + (xcc & 7u)], 1u);  // [b % 8][XCC_ID]
  }
}
__global__ __launch_bounds__(256) void k_store_xcd(uint4* out, size_t n16, unsigned mask) {
  if (!((mask >> (blockIdx.x & 7u)) & 1u)) return;
  const unsigned pc = __builtin_popcount(mask), rank = __builtin_popcount(mask & ((1u << (blockIdx.x & 7u)) - 1u));
  const size_t wg = (size_t)(blockIdx.x >> 3) * pc + rank, nwg = (size_t)(gridDim.x >> 3) * pc;
  typedef unsigned v4 __attribute__((ext_vector_type(4)));
  v4 v = {blockIdx.x, threadIdx.x, 1u, 2u};
  for (size_t i = wg * 256 + threadIdx.x; i < n16; i += nwg * 256) __builtin_nontemporal_store(v, reinterpret_cast<v4*>(out) + i);
}
static void xcd_section() {
  unsigned *bad, *per;
  CK(hipMalloc(&bad, 4)); CK(hipMalloc(&per, 256));
  hipStream_t st[3];
  for (auto& s_ : st) CK(hipStreamCreateWithFlags(&s_, hipStreamNonBlocking));
  CK(hipMemset(bad, 0, 4)); CK(hipMemset(per, 0, 256));
  CK(hipDeviceSynchronize());
  const int grids[3] = {512, 2048, 1000};
  for (int rep = 0; rep < 50; rep++)
    for (int k = 0; k < 3; k++) hipLaunchKernelGGL(k_xcd_check, dim3(grids[k]), dim3(64), 0, st[k], bad, per);
  CK(hipDeviceSynchronize());
  unsigned hb = 0, hp[64];
  CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hp, per, 256, hipMemcpyDeviceToHost));
  std::printf("XCC_ID != b %% 8 for %u of %d workgroups (three queues dispatching at once); XCC_ID values seen for b %% 8 = 0..7:", hb, 50 * (512 + 2048 + 1000));
  for (int b = 0; b < 8; b++) {
    std::printf(" [");
    for (int x = 0; x < 8; x++) if (hp[b * 8 + x]) std::printf("%d:%u ", x, hp[b * 8 + x]);
    std::printf("]");
  }
  std::printf("\n");
  const size_t bytes = (size_t)1 << 30;
  uint4* buf;
  CK(hipMalloc(&buf, bytes));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (unsigned mask : {0xffu, 0xf0u, 0xe0u, 0xc0u, 0x80u})
    for (int G : {2048, 8192}) {
      hipLaunchKernelGGL(k_store_xcd, dim3(G), dim3(256), 0, 0, buf, bytes / 16, mask);
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL(k_store_xcd, dim3(G), dim3(256), 0, 0, buf, bytes / 16, mask);
      CK(hipEventRecord(e1, 0));
      CK(hipDeviceSynchronize());
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      std::printf("1 GiB of non-temporal 16-byte stores from XCD mask 0x%02x, %4d workgroups launched: %.1f us, %.0f GB/s\n", mask, G, ms * 1e3, bytes / ms / 1e6);
    }
}

int main() {
  xcd_section();
  const int iters = 400;  // 25.6 k dependent instructions ~ 50 us for a lone wave
  const int shapes[][2] = {{1, 64}, {256, 64}, {512, 64}, {1024, 64}, {2048, 64}, {256, 128}, {128, 256}, {64, 512}, {512, 128}, {256, 256}, {1024, 128}, {4096, 64}, {8192, 64}, {2048, 128}};
  unsigned* d;
  CK(hipMalloc(&d, 4 * 4 * 16384));
  for (auto& sh : shapes) {
    const int G = sh[0], T = sh[1], waves = G * T / 64;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_place, dim3(G), dim3(T), 0, 0, d, iters, 12345u);  // warm
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k_place, dim3(G), dim3(T), 0, 0, d, iters, 12345u);
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned> h(4 * waves);
    CK(hipMemcpy(h.data(), d, 4 * 4 * waves, hipMemcpyDeviceToHost));
    std::map<unsigned, int> per_simd, per_cu;
    unsigned long long cyc = 0;
    unsigned rmin = 0xffffffffu, rmax = 0;
    std::vector<unsigned> starts;
    for (int w = 0; w < waves; w++) {
      const unsigned hw = h[4 * w], xcc = h[4 * w + 1] & 0xf, r0 = h[4 * w + 1] >> 4;
      starts.push_back(r0); if (r0 < rmin) rmin = r0; if (r0 > rmax) rmax = r0;
      const unsigned simd = (hw >> 4) & 3, cu = (hw >> 8) & 0xf, shid = (hw >> 12) & 1, se = (hw >> 13) & 7;
      const unsigned cu_key = xcc << 16 | se << 8 | shid << 4 | cu;
      per_cu[cu_key]++; per_simd[cu_key << 2 | simd]++;
      cyc += h[4 * w + 2];
    }
    std::map<int, int> hist_simd, hist_cu;
    for (auto& kv : per_simd) hist_simd[kv.second]++;
    for (auto& kv : per_cu) hist_cu[kv.second]++;
    std::printf("grid %4d x %3d (%4d waves): %.1f us, mean wave cycles %.0f | CUs used %zu, SIMDs used %zu | waves per SIMD:", G, T, waves, ms * 1e3,
                (double)cyc / waves, per_cu.size(), per_simd.size());
    for (auto& kv : hist_simd) std::printf(" %dx%d", kv.second, kv.first);
    std::printf(" | waves per CU:");
    for (auto& kv : hist_cu) std::printf(" %dx%d", kv.second, kv.first);
    int late = 0;
    for (unsigned r0 : starts) if (((r0 - rmin) & 0x0fffffffu) > 500) late++;  // started more than 5 us behind the first wave
    std::printf(" | start spread %.1f us, %d waves started > 5 us late\n", ((rmax - rmin) & 0x0fffffffu) / 100.0, late);
  }
  for (int G : {1, 512, 1024}) {
    run_straight("loop64", k_s64, 64, d, G);
    run_straight("run1600", k_s1600, 1600, d, G);
    run_straight("run6400", k_s6400, 6400, d, G);
    run_straight("run12800", k_s12800, 12800, d, G);
    run_straight("run25600", k_s25600, 25600, d, G);
  }
  return 0;
}
