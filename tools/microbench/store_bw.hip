// HBM write-bandwidth microbenchmark for the serializer design (k_serialize writes ~1.1 GB per launch):
// plain vs non-temporal 16-B stores, 8-B stores, and a dependent (LUT -> byte -> store) chain like the real kernel.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k_fill(uint64_t* out, size_t n_pairs, const uint32_t* lut, const uint8_t* srcb) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_pairs) return;
  if (MODE == 0) { ulonglong2 v = {i, i + 1}; *reinterpret_cast<ulonglong2*>(out + 2 * i) = v; }
  else if (MODE == 1) { u64x2 v = {i, i + 1}; __builtin_nontemporal_store(v, reinterpret_cast<u64x2*>(out + 2 * i)); }
  else if (MODE == 2) { out[2 * i] = i; out[2 * i + 1] = i + 1; }
  else if (MODE == 3) {  // dependent loads like the serializer: LUT (60 KB, cached) -> byte (small buffer) -> bit
    uint32_t e0 = (uint32_t)((2 * i) % 15000), e1 = (uint32_t)((2 * i + 1) % 15000);
    uint32_t l0 = lut[e0], l1 = lut[e1];
    uint32_t b0 = srcb[(l0 & 0xffff) + (i >> 9) % 4096 * 16], b1 = srcb[(l1 & 0xffff) + (i >> 9) % 4096 * 16];
    u64x2 v = {(b0 >> (l0 >> 29)) & 1, (b1 >> (l1 >> 29)) & 1};
    __builtin_nontemporal_store(v, reinterpret_cast<u64x2*>(out + 2 * i));
  } else if (MODE == 5) {  // the second-generation serializer's pattern: a wave writes 4 KB as four coalesced 1-KB stores
    u64x2 v = {i, i + 1};
    size_t wave = i / 64, lane = i % 64;
    if (wave * 256 + 255 < n_pairs) {
      u64x2* d = reinterpret_cast<u64x2*>(out) + wave * 256 + lane;
#pragma unroll
      for (int j = 0; j < 4; j++) __builtin_nontemporal_store(v, d + 64 * j);
    }
  } else if (MODE == 6) {  // the same, plain stores
    u64x2 v = {i, i + 1};
    size_t wave = i / 64, lane = i % 64;
    if (wave * 256 + 255 < n_pairs) {
      u64x2* d = reinterpret_cast<u64x2*>(out) + wave * 256 + lane;
#pragma unroll
      for (int j = 0; j < 4; j++) d[64 * j] = v;
    }
  } else if (MODE == 4) {  // 4 elements (32 B) per thread, plain
    ulonglong2 v = {i, i + 1};
    size_t j = (i / 64) * 128 + (i % 64);
    if (j + 64 < n_pairs) { *reinterpret_cast<ulonglong2*>(out + 2 * j) = v; *reinterpret_cast<ulonglong2*>(out + 2 * (j + 64)) = v; }
  }
}
template <int MODE>
int run(const char* name, uint64_t* out, size_t n_pairs, const uint32_t* lut, const uint8_t* srcb) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const size_t threads = (MODE == 5 || MODE == 6) ? n_pairs / 4 : n_pairs;
  dim3 grid((unsigned)((threads + 255) / 256));
  k_fill<MODE><<<grid, 256>>>(out, n_pairs, lut, srcb);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int r = 0; r < 10; r++) k_fill<MODE><<<grid, 256>>>(out, n_pairs, lut, srcb);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 10;
  // one isolated launch (what a kernel of the step sees: ramp-up and tail included)
  float one = 1e9f;
  for (int r = 0; r < 5; r++) {
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    k_fill<MODE><<<grid, 256>>>(out, n_pairs, lut, srcb);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float t; CK(hipEventElapsedTime(&t, a, b));
    one = t < one ? t : one;
  }
  const double per = 1.0;
  printf("%-34s %8.3f ms  %8.1f GB/s   isolated launch %8.3f ms %8.1f GB/s\n", name, ms, per * n_pairs * 16.0 / ms * 1e-6, one, per * n_pairs * 16.0 / one * 1e-6);
  return 0;
}
int main() {
  size_t bytes = 1109ull << 20, n_pairs = bytes / 16;
  uint64_t* out; uint32_t* lut; uint8_t* srcb;
  CK(hipMalloc(&out, bytes)); CK(hipMalloc(&lut, 15000 * 4)); CK(hipMalloc(&srcb, 1 << 20));
  CK(hipMemset(lut, 0x11, 15000 * 4)); CK(hipMemset(srcb, 0x5a, 1 << 20));
  run<0>("16B plain store", out, n_pairs, lut, srcb);
  run<1>("16B non-temporal store", out, n_pairs, lut, srcb);
  run<2>("2 x 8B plain store", out, n_pairs, lut, srcb);
  run<3>("LUT->byte->16B nt store", out, n_pairs, lut, srcb);
  run<4>("2 x 16B per thread plain", out, n_pairs, lut, srcb);
  run<5>("4 x 16B per thread nt (4 KB/wave)", out, n_pairs, lut, srcb);
  run<6>("4 x 16B per thread plain", out, n_pairs, lut, srcb);
  return 0;
}
