// Row-writer structure microbenchmark (round 6).  gfx950 counts vector loads and stores in ONE vmcnt and they complete out of order with each
// other, so a persistent wave that loads, stores and loads again waits for its STORES before it can use its next loads (the compiler must emit
// vmcnt(0)): a k_serialize_few wave pays load latency + store latency per 2 KB.  Here:
//   mixed   persistent waves, each iteration: LUT word -> source word -> two 16-byte stores (2 KB per wave), like k_serialize_few
//   ws      the same work by wave PAIRS: a loader wave (only loads: vmcnt stays in order, DEPTH iterations in flight) hands the values to a
//           storer wave through LDS (lgkmcnt), which only stores and never waits for them
// usage: ws_store [workgroups]     (1.1 GB per launch; reports GB/s for both forms at several grid sizes)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void span_values(const uint32_t* __restrict__ lut, const uint32_t* __restrict__ src, size_t span, uint32_t lane, u64x2 v[2]) {
#pragma unroll
  for (int h = 0; h < 2; h++) {
    const uint32_t p = (uint32_t)((span * 256 + 128 * h + 2 * lane) % 15000u);
    const uint32_t e0 = lut[p], e1 = lut[p + 1];
    const uint32_t w0 = src[((e0 & 0xffffu) + span * 64) & 0x3ffffu], w1 = src[((e1 & 0xffffu) + span * 64) & 0x3ffffu];
    v[h].x = (w0 >> (e0 >> 27)) & 1u; v[h].y = (w1 >> (e1 >> 27)) & 1u;
  }
}
__global__ __launch_bounds__(256) void k_mixed(uint64_t* __restrict__ out, size_t n_spans, const uint32_t* __restrict__ lut, const uint32_t* __restrict__ src) {
  const uint32_t lane = threadIdx.x & 63u;
  for (size_t span = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6); span < n_spans; span += (size_t)gridDim.x * 4) {
    u64x2 v[2];
    span_values(lut, src, span, lane, v);
    u64x2* d = reinterpret_cast<u64x2*>(out + span * 256) + lane;
    __builtin_nontemporal_store(v[0], d); __builtin_nontemporal_store(v[1], d + 64);
  }
}
template <int DEPTH>
__global__ __launch_bounds__(128) void k_ws(uint64_t* __restrict__ out, size_t n_spans, const uint32_t* __restrict__ lut, const uint32_t* __restrict__ src) {
  __shared__ u64x2 buf[DEPTH][2][64];
  const uint32_t lane = threadIdx.x & 63u, role = threadIdx.x >> 6;
  const size_t first = blockIdx.x, step = gridDim.x;
  const size_t n_mine = first < n_spans ? (n_spans - first + step - 1) / step : 0;
  // rounds of DEPTH spans: the loader fills the DEPTH buffers of round r while the storer drains those of round r - 1 (two buffer sets would
  // allow full overlap; here one barrier per round separates them: the loader's DEPTH loads are all in flight together)
  __shared__ u64x2 buf2[DEPTH][2][64];
  for (size_t r = 0; r * DEPTH < n_mine + DEPTH; r++) {
    if (role == 0) {
      if (r * DEPTH < n_mine) {
        u64x2 v[DEPTH][2];
#pragma unroll
        for (int k = 0; k < DEPTH; k++) {
          const size_t i = r * DEPTH + k;
          if (i < n_mine) span_values(lut, src, first + i * step, lane, v[k]);
        }
        auto& B = (r & 1) ? buf2 : buf;
#pragma unroll
        for (int k = 0; k < DEPTH; k++) { B[k][0][lane] = v[k][0]; B[k][1][lane] = v[k][1]; }
      }
    } else if (r > 0) {
      auto& B = ((r - 1) & 1) ? buf2 : buf;
#pragma unroll
      for (int k = 0; k < DEPTH; k++) {
        const size_t i = (r - 1) * DEPTH + k;
        if (i < n_mine) {
          u64x2* d = reinterpret_cast<u64x2*>(out + (first + i * step) * 256) + lane;
          __builtin_nontemporal_store(B[k][0][lane], d); __builtin_nontemporal_store(B[k][1][lane], d + 64);
        }
      }
    }
    __syncthreads();
  }
}
template <typename F>
int timeit(const char* name, unsigned wgs, size_t bytes, F launch) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  launch(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a)); for (int r = 0; r < 8; r++) launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 8;
  printf("%-10s %5u workgroups  %7.3f ms  %7.1f GB/s\n", name, wgs, ms, bytes / ms * 1e-6);
  return 0;
}
int main() {
  const size_t bytes = 1109ull << 20, n_spans = bytes / 2048;
  uint64_t* out; uint32_t *lut, *src;
  CK(hipMalloc(&out, bytes)); CK(hipMalloc(&lut, 15008 * 4)); CK(hipMalloc(&src, 1 << 20));
  CK(hipMemset(lut, 0x11, 15008 * 4)); CK(hipMemset(src, 0x5a, 1 << 20));
  for (unsigned wgs : {256u, 512u, 1024u, 2048u, 4096u}) {
    timeit("mixed", wgs, bytes, [&] { hipLaunchKernelGGL(k_mixed, dim3(wgs), dim3(256), 0, 0, out, n_spans, lut, src); });
    timeit("ws d=2", wgs * 2, bytes, [&] { hipLaunchKernelGGL(k_ws<2>, dim3(wgs * 2), dim3(128), 0, 0, out, n_spans, lut, src); });
    timeit("ws d=4", wgs * 2, bytes, [&] { hipLaunchKernelGGL(k_ws<4>, dim3(wgs * 2), dim3(128), 0, 0, out, n_spans, lut, src); });
    timeit("ws d=8", wgs * 2, bytes, [&] { hipLaunchKernelGGL(k_ws<8>, dim3(wgs * 2), dim3(128), 0, 0, out, n_spans, lut, src); });
  }
  return 0;
}
