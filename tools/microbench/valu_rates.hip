// VALU integer-rate microbenchmark for gfx950 (MI355X).
// Decides the limb schedule for the curve25519 field arithmetic:
// measures wave64 issue cost of v_mad_u64_u32, v_mul_lo_u32, v_mul_hi_u32, 32/64-bit adds and DFMA.
// Build: hipcc --offload-arch=gfx950 -O3 valu_rates.hip -o valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 2048;
constexpr int UNROLL = 8;

template <int MODE>
__global__ __launch_bounds__(256) void k(uint64_t* out, uint32_t seed) {
  uint32_t t = threadIdx.x + blockIdx.x * blockDim.x;
  uint32_t a = seed * 2654435761u + t, b = a ^ 0x9e3779b9u;
  uint64_t acc[UNROLL];
  uint32_t acc32[UNROLL];
  double accd[UNROLL];
#pragma unroll
  for (int i = 0; i < UNROLL; i++) { acc[i] = t + i; acc32[i] = t * 3 + i; accd[i] = (double)(t + i); }
  double da = (double)a * 1e-9, db = (double)b * 1e-9;
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int i = 0; i < UNROLL; i++) {
      if (MODE == 0) {  // v_mad_u64_u32, 8 independent accumulators
        acc[i] = (uint64_t)(a + i) * (uint32_t)(b ^ (uint32_t)acc[i]) + acc[i];
      } else if (MODE == 1) {  // v_mad_u64_u32 dependent chain (latency)
        acc[0] = (uint64_t)(a + i) * (uint32_t)acc[0] + acc[0];
      } else if (MODE == 2) {  // v_mul_lo_u32
        acc32[i] = acc32[i] * (a + i);
      } else if (MODE == 3) {  // v_mul_hi_u32
        acc32[i] = __umulhi(acc32[i], a + i) + 1u;
      } else if (MODE == 4) {  // 32-bit add (full-rate reference)
        acc32[i] = acc32[i] + (a ^ acc32[(i + 1) % UNROLL]);
      } else if (MODE == 5) {  // 64-bit add
        acc[i] = acc[i] + (acc[(i + 1) % UNROLL] ^ b);
      } else if (MODE == 6) {  // DFMA
        accd[i] = __builtin_fma(accd[i], da, db);
      } else if (MODE == 7) {  // v_mad_u32_u24
        acc32[i] = __umul24(acc32[i], a + i) + acc32[i];
      } else if (MODE == 8) {  // 64-bit shift + mask (carry step)
        acc[i] = (acc[i] >> 26) + (acc[(i + 1) % UNROLL] & 0x3ffffff);
      }
    }
  }
  uint64_t r = 0;
#pragma unroll
  for (int i = 0; i < UNROLL; i++) r += acc[i] + acc32[i] + (uint64_t)accd[i];
  out[t] = r;
}

template <int MODE>
int run(const char* name, int blocks, uint64_t* d_out) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  k<MODE><<<blocks, 256>>>(d_out, 1);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  const int reps = 5;
  for (int r = 0; r < reps; r++) k<MODE><<<blocks, 256>>>(d_out, r + 2);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  double ops = (double)blocks * 256 * ITERS * UNROLL;
  double waves = (double)blocks * 4;
  // cycles per wave-instruction per SIMD assuming 1024 SIMDs @ 2.4 GHz and perfect balance
  double wave_insts = waves * ITERS * UNROLL;
  double cyc = ms * 1e-3 * 2.4e9 * 1024.0 / wave_insts;
  printf("%-28s blocks=%5d  %8.3f ms  %8.2f Gop/s(lane)  ~%.2f cyc/wave-inst/SIMD\n", name, blocks, ms, ops / ms * 1e-6, cyc);
  return 0;
}

int main() {
  uint64_t* d_out;
  CK(hipMalloc(&d_out, sizeof(uint64_t) * 256 * 8192));
  int props_cu = 0;
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("device %s CUs=%d clock=%d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
  for (int blocks : {256, 1024, 2048, 8192}) {
    run<0>("mad_u64_u32 x8 indep", blocks, d_out);
    run<1>("mad_u64_u32 dependent", blocks, d_out);
    run<2>("mul_lo_u32", blocks, d_out);
    run<3>("mul_hi_u32", blocks, d_out);
    run<4>("add_u32", blocks, d_out);
    run<5>("add_u64", blocks, d_out);
    run<6>("dfma", blocks, d_out);
    run<7>("mad_u32_u24", blocks, d_out);
    run<8>("shr64+and+add64", blocks, d_out);
  }
  (void)props_cu;
  return 0;
}
