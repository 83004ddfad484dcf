// Can a VALU-bound kernel and an HBM-store-bound kernel share the chip?  A: a dependent v_mad_u64_u32 chain, 8 / WAVES_DIV waves per SIMD on every SIMD (issue-saturated from one wave on; with all 8 wave slots taken the store kernel cannot even start);
// B: 16-byte stores over 1.1 GB (one store per thread, the fastest pattern of store_bw.hip).  Each alone, then both at once on two streams.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ __launch_bounds__(256) void k_valu(uint64_t* out, int iters) {
#ifdef VALU_PRIO
  __builtin_amdgcn_s_setprio(VALU_PRIO);
#endif
  uint64_t a = threadIdx.x + 1, b = blockIdx.x + 3, c = 7, d = 11;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int j = 0; j < 16; j++) { a = (uint64_t)(uint32_t)a * (uint32_t)b + c; c = (uint64_t)(uint32_t)c * (uint32_t)d + a; b = (uint64_t)(uint32_t)b * (uint32_t)a + d; d = (uint64_t)(uint32_t)d * (uint32_t)c + b; }
  }
  if (a + b + c + d == 0x1234567) out[0] = a;
}
__global__ __launch_bounds__(256) void k_store(uint64_t* out, size_t n_pairs) {
#ifdef STORE_PRIO
  __builtin_amdgcn_s_setprio(STORE_PRIO);
#endif
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_pairs) return;
  ulonglong2 v = {i, i + 1};
  *reinterpret_cast<ulonglong2*>(out + 2 * i) = v;
}
#ifndef WAVES_DIV
#define WAVES_DIV 4
#endif
int main() {
  size_t bytes = 1109ull << 20, n_pairs = bytes / 16;
  uint64_t *out, *dummy;
  CK(hipMalloc(&out, bytes)); CK(hipMalloc(&dummy, 64));
  hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  hipEvent_t a0, a1, b0, b1; CK(hipEventCreate(&a0)); CK(hipEventCreate(&a1)); CK(hipEventCreate(&b0)); CK(hipEventCreate(&b1));
  const int iters = 700 * WAVES_DIV, reps_store = 6;   // ~1 ms of VALU, ~1 ms of stores
  const dim3 gv(256 * 4 * 2 / WAVES_DIV), gs((unsigned)((n_pairs + 255) / 256));   // 2048 blocks x 4 waves = 8 waves per SIMD
  auto valu = [&](hipStream_t s) { hipLaunchKernelGGL(k_valu, gv, dim3(256), 0, s, dummy, iters); };
  auto store = [&](hipStream_t s) { for (int r = 0; r < reps_store; r++) hipLaunchKernelGGL(k_store, gs, dim3(256), 0, s, out, n_pairs); };
  for (int w = 0; w < 2; w++) { valu(s1); store(s2); }
  CK(hipDeviceSynchronize());
  float ta, tb, ta2, tb2;
  CK(hipEventRecord(a0, s1)); valu(s1); CK(hipEventRecord(a1, s1)); CK(hipDeviceSynchronize()); CK(hipEventElapsedTime(&ta, a0, a1));
  CK(hipEventRecord(b0, s2)); store(s2); CK(hipEventRecord(b1, s2)); CK(hipDeviceSynchronize()); CK(hipEventElapsedTime(&tb, b0, b1));
  CK(hipEventRecord(a0, s1)); CK(hipEventRecord(b0, s2)); valu(s1); store(s2); CK(hipEventRecord(a1, s1)); CK(hipEventRecord(b1, s2));
  CK(hipDeviceSynchronize()); CK(hipEventElapsedTime(&ta2, a0, a1)); CK(hipEventElapsedTime(&tb2, b0, b1));
  printf("VALU chain alone %.3f ms | stores alone %.3f ms (%.0f GB/s) | together: VALU %.3f ms, stores %.3f ms (%.0f GB/s) | sum alone %.3f, max alone %.3f\n", ta, tb,
         reps_store * bytes / tb * 1e-6, ta2, tb2, reps_store * bytes / tb2 * 1e-6, ta + tb, ta > tb ? ta : tb);
  return 0;
}
