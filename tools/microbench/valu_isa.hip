// ISA-level VALU issue-rate microbenchmark for gfx950 (MI355X): one opcode per kernel, written as inline asm so that what is timed is
// exactly the instruction named (the round-2 version timed C++ source operations, which the compiler folded or split).
//   hipcc --offload-arch=gfx950 -O2 -o valu_isa valu_isa.hip && ./valu_isa
// For every opcode: cycles (s_memtime, shader clock) per wave64 instruction
//   indep  8 independent accumulators, W = 1, 2, 4, 8 waves per SIMD  -> issue cost per instruction of one wave / throughput per SIMD
//   dep    one dependent chain, one wave per SIMD                      -> latency
//   half   independent, one wave per SIMD, EXEC = low 32 lanes         -> does a half-empty wave64 issue faster?
// Output: one line per (opcode, mode).  cyc/inst/wave = wave cycles / instructions of that wave; cyc/inst/SIMD = / W.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <algorithm>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                              \
  do {                                                                                     \
    hipError_t e_ = (x);                                                                   \
    if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } \
  } while (0)

constexpr int ITERS = 2000;   // loop trips
constexpr int PER_TRIP = 64;  // instructions per trip (8 accumulators x 8, or a chain of 64)

// BODY_I: text of one instruction on accumulator %k (independent form), BODY_D: on the chain register %0
#define DEF_KERNEL(NAME, INDEP8, DEP1)                                                                                              \
  __global__ __launch_bounds__(64) void k_##NAME(unsigned long long* out, int mode, unsigned seed) {                                \
    unsigned a0 = seed + threadIdx.x, a1 = a0 * 3u + 1u, a2 = a0 * 5u + 2u, a3 = a0 * 7u + 3u, a4 = a0 ^ 0x55u, a5 = a0 ^ 0x3333u,     \
             a6 = a0 + 77u, a7 = a0 + 99u;                                                                                          \
    unsigned long long w0 = a0, w1 = a1, w2 = a2, w3 = a3, w4 = a4, w5 = a5, w6 = a6, w7 = a7;                                      \
    const unsigned b = seed | 1u, c = seed * 9u + 5u;                                                                               \
    if (mode == 2 && threadIdx.x >= 32) return; /* the wave goes on with EXEC = its low 32 lanes */                                  \
    const unsigned long long t0 = __builtin_readcyclecounter();                                                                     \
    if (mode == 1) {                                                                                                                \
      for (int it = 0; it < ITERS; it++) {                                                                                          \
        asm volatile(".rept 64\n" DEP1 "\n.endr" : "+v"(a0), "+v"(w0) : "v"(b), "v"(c));                                             \
      }                                                                                                                             \
    } else {                                                                                                                        \
      for (int it = 0; it < ITERS; it++) {                                                                                          \
        asm volatile(".rept 8\n" INDEP8 "\n.endr"                                                                                  \
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(w0), "+v"(w1), "+v"(w2),   \
                       "+v"(w3), "+v"(w4), "+v"(w5), "+v"(w6), "+v"(w7)                                                             \
                     : "v"(b), "v"(c));                                                                                            \
      }                                                                                                                             \
    }                                                                                                                               \
    const unsigned long long t1 = __builtin_readcyclecounter();                                                                     \
    if (threadIdx.x == 0) out[blockIdx.x * 2] = t1 - t0;                                                                            \
    if (threadIdx.x == 0)                                                                                                           \
      out[blockIdx.x * 2 + 1] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + w0 + w1 + w2 + w3 + w4 + w5 + w6 + w7;                        \
  }

// operands: %0..%7 = a0..a7 (32-bit), %8..%15 = w0..w7 (64-bit pairs), %16 = b, %17 = c
#define I8(op_fmt) op_fmt(0) "\n" op_fmt(1) "\n" op_fmt(2) "\n" op_fmt(3) "\n" op_fmt(4) "\n" op_fmt(5) "\n" op_fmt(6) "\n" op_fmt(7)
// dependent forms use %0 (a0), %1 (w0), %2 (b), %3 (c)

#define ADD_I(k) "v_add_u32 %" #k ", %" #k ", %16"
DEF_KERNEL(v_add_u32, I8(ADD_I), "v_add_u32 %0, %0, %2")
#define XAD_I(k) "v_xad_u32 %" #k ", %" #k ", %16, %17"
DEF_KERNEL(v_xad_u32, I8(XAD_I), "v_xad_u32 %0, %0, %2, %3")
#define ADD3_I(k) "v_add3_u32 %" #k ", %" #k ", %16, %17"
DEF_KERNEL(v_add3_u32, I8(ADD3_I), "v_add3_u32 %0, %0, %2, %3")
#define LSHLADD_I(k) "v_lshl_add_u32 %" #k ", %" #k ", 3, %16"
DEF_KERNEL(v_lshl_add_u32, I8(LSHLADD_I), "v_lshl_add_u32 %0, %0, 3, %2")
#define ANDOR_I(k) "v_and_or_b32 %" #k ", %" #k ", %16, %17"
DEF_KERNEL(v_and_or_b32, I8(ANDOR_I), "v_and_or_b32 %0, %0, %2, %3")
#define BFE_I(k) "v_bfe_u32 %" #k ", %" #k ", 3, 29"
DEF_KERNEL(v_bfe_u32, I8(BFE_I), "v_bfe_u32 %0, %0, 3, 29")
#define ALIGN_I(k) "v_alignbit_b32 %" #k ", %" #k ", %16, 7"
DEF_KERNEL(v_alignbit_b32, I8(ALIGN_I), "v_alignbit_b32 %0, %0, %2, 7")
#define CNDMASK_I(k) "v_cndmask_b32 %" #k ", %" #k ", %16, vcc"
DEF_KERNEL(v_cndmask_b32, I8(CNDMASK_I), "v_cndmask_b32 %0, %0, %2, vcc")
#define MUL24_I(k) "v_mul_u32_u24 %" #k ", %" #k ", %16"
DEF_KERNEL(v_mul_u32_u24, I8(MUL24_I), "v_mul_u32_u24 %0, %0, %2")
#define MUL24DPP_I(k) "v_mul_u32_u24_dpp %" #k ", %" #k ", %16 row_ror:3 row_mask:0xf bank_mask:0xf"
DEF_KERNEL(v_mul_u32_u24_dpp_row_ror, I8(MUL24DPP_I), "v_mul_u32_u24_dpp %0, %0, %2 row_ror:3 row_mask:0xf bank_mask:0xf")
#define MAD24_I(k) "v_mad_u32_u24 %" #k ", %" #k ", %16, %17"
DEF_KERNEL(v_mad_u32_u24, I8(MAD24_I), "v_mad_u32_u24 %0, %0, %2, %3")
#define MULLO_I(k) "v_mul_lo_u32 %" #k ", %" #k ", %16"
DEF_KERNEL(v_mul_lo_u32, I8(MULLO_I), "v_mul_lo_u32 %0, %0, %2")
#define MULHI_I(k) "v_mul_hi_u32 %" #k ", %" #k ", %16"
DEF_KERNEL(v_mul_hi_u32, I8(MULHI_I), "v_mul_hi_u32 %0, %0, %2")
// 64-bit accumulators: w_k += b * c   (sdst = vcc)
#define MAD64U_I(k) "v_mad_u64_u32 %" W(k) ", vcc, %16, %17, %" W(k)
#define W(k) W_##k
#define W_0 "8"
#define W_1 "9"
#define W_2 "10"
#define W_3 "11"
#define W_4 "12"
#define W_5 "13"
#define W_6 "14"
#define W_7 "15"
DEF_KERNEL(v_mad_u64_u32, I8(MAD64U_I), "v_mad_u64_u32 %1, vcc, %2, %3, %1")
#define MAD64I_I(k) "v_mad_i64_i32 %" W(k) ", vcc, %16, %17, %" W(k)
DEF_KERNEL(v_mad_i64_i32, I8(MAD64I_I), "v_mad_i64_i32 %1, vcc, %2, %3, %1")
#define LSHL64_I(k) "v_lshlrev_b64 %" W(k) ", 1, %" W(k)
DEF_KERNEL(v_lshlrev_b64, I8(LSHL64_I), "v_lshlrev_b64 %1, 1, %1")
#define ASHR64_I(k) "v_ashrrev_i64 %" W(k) ", 1, %" W(k)
DEF_KERNEL(v_ashrrev_i64, I8(ASHR64_I), "v_ashrrev_i64 %1, 1, %1")
#define MOVDPPQ_I(k) "v_mov_b32_dpp %" #k ", %" #k " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
DEF_KERNEL(v_mov_b32_dpp_quad_perm, I8(MOVDPPQ_I), "v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
#define MOVDPPB_I(k) "v_mov_b32_dpp %" #k ", %" #k " row_newbcast:5 row_mask:0xf bank_mask:0xf"
DEF_KERNEL(v_mov_b32_dpp_row_newbcast, I8(MOVDPPB_I), "v_mov_b32_dpp %0, %0 row_newbcast:5 row_mask:0xf bank_mask:0xf")
#define ADDDPP_I(k) "v_add_u32_dpp %" #k ", %" #k ", %16 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
DEF_KERNEL(v_add_u32_dpp_quad_perm, I8(ADDDPP_I), "v_add_u32_dpp %0, %0, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
#define PERM16_I(k) "v_permlane16_swap_b32 %" #k ", %" W(k)
DEF_KERNEL(v_permlane16_swap_b32, "v_permlane16_swap_b32 %0, %1\nv_permlane16_swap_b32 %2, %3\nv_permlane16_swap_b32 %4, %5\nv_permlane16_swap_b32 %6, %7\nv_permlane16_swap_b32 %0, %2\nv_permlane16_swap_b32 %1, %3\nv_permlane16_swap_b32 %4, %6\nv_permlane16_swap_b32 %5, %7",
           "v_permlane16_swap_b32 %0, %2")
DEF_KERNEL(v_permlane32_swap_b32, "v_permlane32_swap_b32 %0, %1\nv_permlane32_swap_b32 %2, %3\nv_permlane32_swap_b32 %4, %5\nv_permlane32_swap_b32 %6, %7\nv_permlane32_swap_b32 %0, %2\nv_permlane32_swap_b32 %1, %3\nv_permlane32_swap_b32 %4, %6\nv_permlane32_swap_b32 %5, %7",
           "v_permlane32_swap_b32 %0, %2")
#define SUB_I(k) "v_sub_u32 %" #k ", %" #k ", %16"
DEF_KERNEL(v_sub_u32, I8(SUB_I), "v_sub_u32 %0, %0, %2")
#define AND_I(k) "v_and_b32 %" #k ", %" #k ", %16"
DEF_KERNEL(v_and_b32, I8(AND_I), "v_and_b32 %0, %0, %2")
#define OR_I(k) "v_or_b32 %" #k ", %" #k ", %16"
DEF_KERNEL(v_or_b32, I8(OR_I), "v_or_b32 %0, %0, %2")
#define LSHL_I(k) "v_lshlrev_b32 %" #k ", 3, %" #k
DEF_KERNEL(v_lshlrev_b32, I8(LSHL_I), "v_lshlrev_b32 %0, 3, %0")
#define LSHR_I(k) "v_lshrrev_b32 %" #k ", 3, %" #k
DEF_KERNEL(v_lshrrev_b32, I8(LSHR_I), "v_lshrrev_b32 %0, 3, %0")
#define ASHR_I(k) "v_ashrrev_i32 %" #k ", 3, %" #k
DEF_KERNEL(v_ashrrev_i32, I8(ASHR_I), "v_ashrrev_i32 %0, 3, %0")
#define MOV_I(k) "v_mov_b32 %" #k ", %16"
DEF_KERNEL(v_mov_b32, I8(MOV_I), "v_mov_b32 %0, %2")
#define NOT_I(k) "v_not_b32 %" #k ", %" #k
DEF_KERNEL(v_not_b32, I8(NOT_I), "v_not_b32 %0, %0")
#define MINU_I(k) "v_min_u32 %" #k ", %" #k ", %16"
DEF_KERNEL(v_min_u32, I8(MINU_I), "v_min_u32 %0, %0, %2")
#define CND64_I(k) "v_cndmask_b32_e64 %" #k ", %" #k ", %16, s[10:11]"
DEF_KERNEL(v_cndmask_b32_e64_sgpr, I8(CND64_I), "v_cndmask_b32_e64 %0, %0, %2, s[10:11]")
#define CMP_I(k) "v_cmp_lt_u32 vcc, %" #k ", %16"
DEF_KERNEL(v_cmp_lt_u32_vcc, I8(CMP_I), "v_cmp_lt_u32 vcc, %0, %2")
#define ROR_I(k) "v_alignbit_b32 %" #k ", %" #k ", %" #k ", 7"
DEF_KERNEL(v_alignbit_rotate, I8(ROR_I), "v_alignbit_b32 %0, %0, %0, 7")
// the pair the compiler emits for a select: compare into VCC, v_cndmask reading VCC (two instructions per pair: reported per instruction)
#define CMPSEL_I(k) "v_cmp_lt_u32 vcc, %" #k ", %16\nv_cndmask_b32 %" #k ", %" #k ", %17, vcc"
DEF_KERNEL(v_cmp_vcc_cndmask_pair, CMPSEL_I(0) "\n" CMPSEL_I(1) "\n" CMPSEL_I(2) "\n" CMPSEL_I(3), "v_cmp_lt_u32 vcc, %0, %2\nv_cndmask_b32 %0, %0, %3, vcc")
#define CMPSEL64_I(k) "v_cmp_lt_u32_e64 s[10:11], %" #k ", %16\nv_cndmask_b32_e64 %" #k ", %" #k ", %17, s[10:11]"
DEF_KERNEL(v_cmp_sgpr_cndmask_pair, CMPSEL64_I(0) "\n" CMPSEL64_I(1) "\n" CMPSEL64_I(2) "\n" CMPSEL64_I(3), "v_cmp_lt_u32_e64 s[10:11], %0, %2\nv_cndmask_b32_e64 %0, %0, %3, s[10:11]")
#define FMA_I(k) "v_fma_f32 %" #k ", %" #k ", %16, %17"
DEF_KERNEL(v_fma_f32, I8(FMA_I), "v_fma_f32 %0, %0, %2, %3")
// SHA-2 style ops
#define XOR_I(k) "v_xor_b32 %" #k ", %" #k ", %16"
DEF_KERNEL(v_xor_b32, I8(XOR_I), "v_xor_b32 %0, %0, %2")
#define BFI_I(k) "v_bfi_b32 %" #k ", %" #k ", %16, %17"
DEF_KERNEL(v_bfi_b32, I8(BFI_I), "v_bfi_b32 %0, %0, %2, %3")
// carry chain: v_add_co_u32 + v_addc_co_u32 counted as TWO instructions per pair (reported per instruction)
#define ADDC_I(k) "v_add_co_u32 %" #k ", vcc, %" #k ", %16\nv_addc_co_u32 %" #k ", vcc, %" #k ", %17, vcc"
DEF_KERNEL(v_add_co_addc_pair, ADDC_I(0) "\n" ADDC_I(1) "\n" ADDC_I(2) "\n" ADDC_I(3), "v_add_co_u32 %0, vcc, %0, %2\nv_addc_co_u32 %0, vcc, %0, %3, vcc")

struct Entry {
  const char* name;
  void (*fn)(unsigned long long*, int, unsigned);
  int per_trip_indep, per_trip_dep;
};

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int simds = prop.multiProcessorCount * 4;
  std::printf("# device %s, %d CUs, %d SIMDs, clock %d MHz; ITERS %d\n", prop.gcnArchName, prop.multiProcessorCount, simds, prop.clockRate / 1000, ITERS);
  unsigned long long* d_out;
  const int max_blocks = simds * 8;
  CK(hipMalloc(&d_out, sizeof(unsigned long long) * 2 * max_blocks));
  std::vector<unsigned long long> h(2 * max_blocks);
#define E(NAME, PI, PD) {#NAME, k_##NAME, PI, PD}
  const Entry entries[] = {
      E(v_add_u32, 64, 64), E(v_xad_u32, 64, 64), E(v_add3_u32, 64, 64), E(v_lshl_add_u32, 64, 64), E(v_and_or_b32, 64, 64), E(v_bfe_u32, 64, 64),
      E(v_alignbit_b32, 64, 64), E(v_cndmask_b32, 64, 64), E(v_xor_b32, 64, 64), E(v_bfi_b32, 64, 64), E(v_mul_u32_u24, 64, 64),
      E(v_mul_u32_u24_dpp_row_ror, 64, 64), E(v_mad_u32_u24, 64, 64), E(v_mul_lo_u32, 64, 64), E(v_mul_hi_u32, 64, 64), E(v_mad_u64_u32, 64, 64),
      E(v_mad_i64_i32, 64, 64), E(v_lshlrev_b64, 64, 64), E(v_ashrrev_i64, 64, 64), E(v_mov_b32_dpp_quad_perm, 64, 64),
      E(v_mov_b32_dpp_row_newbcast, 64, 64), E(v_add_u32_dpp_quad_perm, 64, 64), E(v_permlane16_swap_b32, 64, 64),
      E(v_permlane32_swap_b32, 64, 64), E(v_sub_u32, 64, 64), E(v_and_b32, 64, 64), E(v_or_b32, 64, 64), E(v_lshlrev_b32, 64, 64),
      E(v_lshrrev_b32, 64, 64), E(v_ashrrev_i32, 64, 64), E(v_mov_b32, 64, 64), E(v_not_b32, 64, 64), E(v_min_u32, 64, 64), E(v_cndmask_b32_e64_sgpr, 64, 64),
      E(v_cmp_lt_u32_vcc, 64, 64), E(v_alignbit_rotate, 64, 64), E(v_cmp_vcc_cndmask_pair, 64, 128), E(v_cmp_sgpr_cndmask_pair, 64, 128), E(v_fma_f32, 64, 64), E(v_add_co_addc_pair, 64, 128)};
  std::printf("%-28s %9s %9s %9s | %s\n", "opcode", "lat W=1", "dep W=1", "half W=1",
              "throughput: cycles per wave64 instruction per SIMD at W = 1, 2, 4, 8, 16 waves per SIMD offered (kernel wall time x 2.4 GHz x SIMDs / wave-instructions)");
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (const Entry& e : entries) {
    double lat[3] = {0}, thr[5] = {0};
    for (int pass = 0; pass < 8; pass++) {
      const int mode = pass < 5 ? 0 : (pass == 5 ? 0 : (pass == 6 ? 1 : 2));
      const int W = pass < 5 ? (1 << pass) : 1;
      const int blocks = simds * W;
      float ms = 0;
      for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(64), 0, 0, d_out, mode, 12345u + rep);
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        float t = 0;
        CK(hipEventElapsedTime(&t, e0, e1));
        ms = rep == 0 ? t : (t < ms ? t : ms);
      }
      const double insts = (double)ITERS * (mode == 1 ? e.per_trip_dep : e.per_trip_indep);
      if (pass < 5) {
        thr[pass] = (double)ms * 1e-3 * 2.4e9 * simds / ((double)blocks * insts);
      } else {
        CK(hipMemcpy(h.data(), d_out, sizeof(unsigned long long) * 2 * blocks, hipMemcpyDeviceToHost));
        std::vector<unsigned long long> t(blocks);
        for (int b = 0; b < blocks; b++) t[b] = h[2 * b];
        std::sort(t.begin(), t.end());
        lat[pass - 5] = (double)t[blocks / 2] / insts;   // median wave, s_memtime cycles per instruction of that wave
      }
    }
    std::printf("%-28s %9.2f %9.2f %9.2f | %6.2f %6.2f %6.2f %6.2f %6.2f\n", e.name, lat[0], lat[1], lat[2], thr[0], thr[1], thr[2], thr[3], thr[4]);
  }
  CK(hipFree(d_out));
  return 0;
}
