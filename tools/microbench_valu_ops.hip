// microbench_valu_ops.hip -- issue cost of single vector instructions on gfx950, one kind per launch: which of the candidate
// formulations of the relaxation's pixel update (u32 / f32 / packed 16-bit min, max, med3; compare + select; DPP moves,
// with the ALU op or as a move) are full rate and which are half rate.  Companion of microbench_valu.hip (round 3).
// 16 independent chains per lane, 6 waves per SIMD (the occupancy class of k_relax), 4096 iterations.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/_build/microbench_valu_ops tools/microbench_valu_ops.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
constexpr int NCH = 16;

// id, printable name, asm text (%0 = chain register, read and written; %1 / %2 = other sources)
#define OPS(X) \
  X(0, "v_add_u32", "v_add_u32 %0, %0, %1") \
  X(1, "v_sub_u32", "v_sub_u32 %0, %0, %1") \
  X(2, "v_min_u32", "v_min_u32 %0, %0, %1") \
  X(3, "v_max_u32", "v_max_u32 %0, %0, %1") \
  X(4, "v_min_i32", "v_min_i32 %0, %0, %1") \
  X(5, "v_and_b32", "v_and_b32 %0, %0, %1") \
  X(6, "v_or_b32", "v_or_b32 %0, %0, %1") \
  X(7, "v_lshlrev_b32", "v_lshlrev_b32 %0, 1, %0") \
  X(8, "v_min_f32", "v_min_f32 %0, %0, %1") \
  X(9, "v_max_f32", "v_max_f32 %0, %0, %1") \
  X(10, "v_add_f32", "v_add_f32 %0, %0, %1") \
  X(11, "v_min_u16", "v_min_u16 %0, %0, %1") \
  X(12, "v_min_f16", "v_min_f16 %0, %0, %1") \
  X(13, "v_pk_min_u16", "v_pk_min_u16 %0, %0, %1") \
  X(14, "v_pk_max_u16", "v_pk_max_u16 %0, %0, %1") \
  X(15, "v_pk_add_u16", "v_pk_add_u16 %0, %0, %1") \
  X(16, "v_pk_min_i16", "v_pk_min_i16 %0, %0, %1") \
  X(17, "v_pk_min_f16", "v_pk_min_f16 %0, %0, %1") \
  X(18, "v_pk_max_f16", "v_pk_max_f16 %0, %0, %1") \
  X(19, "v_pk_add_f16", "v_pk_add_f16 %0, %0, %1") \
  X(20, "v_mul_u32_u24", "v_mul_u32_u24 %0, %0, %1") \
  X(21, "v_mov_b32", "v_mov_b32 %0, %1") \
  X(22, "v_min3_u32", "v_min3_u32 %0, %0, %1, %2") \
  X(23, "v_med3_u32", "v_med3_u32 %0, %0, %1, %2") \
  X(24, "v_max3_u32", "v_max3_u32 %0, %0, %1, %2") \
  X(25, "v_min3_i32", "v_min3_i32 %0, %0, %1, %2") \
  X(26, "v_min3_f32", "v_min3_f32 %0, %0, %1, %2") \
  X(27, "v_med3_f32", "v_med3_f32 %0, %0, %1, %2") \
  X(28, "v_max3_f32", "v_max3_f32 %0, %0, %1, %2") \
  X(29, "v_min3_u16", "v_min3_u16 %0, %0, %1, %2") \
  X(30, "v_med3_u16", "v_med3_u16 %0, %0, %1, %2") \
  X(31, "v_min3_f16", "v_min3_f16 %0, %0, %1, %2") \
  X(32, "v_med3_f16", "v_med3_f16 %0, %0, %1, %2") \
  X(33, "v_add3_u32", "v_add3_u32 %0, %0, %1, %2") \
  X(34, "v_lshl_add_u32", "v_lshl_add_u32 %0, %0, 1, %1") \
  X(35, "v_lshl_or_b32", "v_lshl_or_b32 %0, %0, 1, %1") \
  X(36, "v_and_or_b32", "v_and_or_b32 %0, %0, %1, %2") \
  X(37, "v_bfe_u32", "v_bfe_u32 %0, %0, 3, 8") \
  X(38, "v_bfi_b32", "v_bfi_b32 %0, %0, %1, %2") \
  X(39, "v_perm_b32", "v_perm_b32 %0, %0, %1, %2") \
  X(40, "v_alignbit_b32", "v_alignbit_b32 %0, %0, %1, 16") \
  X(41, "v_fma_f32", "v_fma_f32 %0, %0, %1, %2") \
  X(42, "v_cmp_lt_u32 + v_cndmask_b32 (2 instr)", "v_cmp_lt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc") \
  X(43, "v_cmp_lt_u32 vcc (alone)", "v_cmp_lt_u32 vcc, %0, %1") \
  X(44, "v_cndmask_b32 (alone)", "v_cndmask_b32 %0, %0, %2, vcc") \
  X(45, "v_mov_b32_dpp row_shr:1", "s_nop 1\n\tv_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf") \
  X(46, "v_mov_b32_dpp wave_shr:1", "s_nop 1\n\tv_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf") \
  X(47, "v_add_u32_dpp row_shr:1", "s_nop 1\n\tv_add_u32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf") \
  X(48, "v_min_u32_dpp row_shr:1", "s_nop 1\n\tv_min_u32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf") \
  X(49, "v_min_u32_dpp wave_shr:1", "s_nop 1\n\tv_min_u32_dpp %0, %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf") \
  X(50, "v_min_f32_dpp row_shr:1", "s_nop 1\n\tv_min_f32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf") \
  X(51, "v_add_u32_sdwa (src1 WORD_1)", "v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1") \
  X(52, "v_pk_mul_lo_u16", "v_pk_mul_lo_u16 %0, %0, %1") \
  X(53, "v_pk_sub_u16", "v_pk_sub_u16 %0, %0, %1") \
  X(54, "v_pk_lshlrev_b16", "v_pk_lshlrev_b16 %0, 1, %0") \
  X(55, "v_max_i32", "v_max_i32 %0, %0, %1") \
  X(56, "v_cmp_lt_u32 s[10:11] + v_cndmask e64 (2 instr)", "v_cmp_lt_u32 s[10:11], %0, %1\n\tv_cndmask_b32 %0, %0, %2, s[10:11]") \
  X(57, "v_xor_b32", "v_xor_b32 %0, %0, %1") \
  X(58, "v_med3_f32 (const operand)", "v_med3_f32 %0, %0, %1, 1.0") \
  X(59, "v_sad_u32", "v_sad_u32 %0, %0, %1, %2") \
  X(60, "v_pk_fma_f16", "v_pk_fma_f16 %0, %0, %1, %2") \
  X(61, "v_pk_add_u16 op_sel (hi,lo swap)", "v_pk_add_u16 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0]")
constexpr int NOPS = 62;

template <int OP>
__global__ __launch_bounds__(256) void k_op(uint32_t *out, int iters, uint32_t x) {
  uint32_t a[NCH], b = (x ^ threadIdx.x) | 0x00010001u, c = (x + 7u * threadIdx.x) | 0x3c003c00u;
#pragma unroll
  for (int i = 0; i < NCH; ++i) a[i] = (x * (i + 1) + threadIdx.x) & 0x3fff3fffu;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
#define X(id, name, text) if (OP == id) asm volatile(text : "+v"(a[i]) : "v"(b), "v"(c) : "vcc", "s10", "s11");
      OPS(X)
#undef X
    }
  }
  uint32_t s = 0;
#pragma unroll
  for (int i = 0; i < NCH; ++i) s += a[i];
  if (s == 0x12345678u) out[threadIdx.x] = s;
}

template <int OP>
float run(uint32_t *d_out, int grid, int iters) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  k_op<OP><<<grid, 256>>>(d_out, iters, 3u);
  k_op<OP><<<grid, 256>>>(d_out, iters, 3u);
  CHECK(hipEventRecord(e0));
  for (int i = 0; i < 5; ++i) k_op<OP><<<grid, 256>>>(d_out, iters, 3u);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
  return ms / 5;
}

template <int OP>
void all(uint32_t *d_out, int cus, const char *const *names) {
  const int iters = 4096, w = 6;
  const float ms = run<OP>(d_out, cus * w, iters);
  const double instr_per_simd = (double)iters * NCH * w;
  printf("%-52s %10.4f ms %8.2f cyc/wave-instr/SIMD (at 2.4 GHz) %8.2f Tlane-ops/s\n", names[OP], ms, ms * 1e-3 * 2.4e9 / instr_per_simd,
         (double)iters * NCH * 256.0 * cus * w / (ms * 1e-3) / 1e12);
  if constexpr (OP + 1 < NOPS) all<OP + 1>(d_out, cus, names);
}

int main() {
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  uint32_t *d_out; CHECK(hipMalloc(&d_out, 4096));
  static const char *names[NOPS] = {
#define X(id, name, text) name,
      OPS(X)
#undef X
  };
  for (int i = 0; i < 50; ++i) k_op<0><<<cus * 8, 256>>>(d_out, 4096, 3u);      // clocks up
  CHECK(hipDeviceSynchronize());
  printf("one instruction kind per launch, 16 independent chains per lane, 6 waves per SIMD (2-instruction rows: per PAIR)\n");
  all<0>(d_out, cus, names);
  return 0;
}
