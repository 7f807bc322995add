// VALU issue-rate microbenchmark on gfx950: wave-instructions per cycle per SIMD for the ops the resample kernels use.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

#define DEFK(NAME, ASMSTR)                                                                          \
  __global__ void __launch_bounds__(256) NAME(unsigned *out, int iters, unsigned a, unsigned b) {   \
    unsigned r[16];                                                                                 \
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i;                                       \
    unsigned x = a + threadIdx.x, y = b;                                                            \
    for (int it = 0; it < iters; it++) {                                                            \
      _Pragma("unroll") for (int i = 0; i < 16; i++) {                                              \
        asm volatile(ASMSTR : "+v"(r[i]) : "v"(x), "v"(y));                                         \
      }                                                                                             \
    }                                                                                               \
    unsigned s = 0;                                                                                 \
    for (int i = 0; i < 16; i++) s += r[i];                                                         \
    if (s == 0x12345) out[0] = s;                                                                   \
  }

DEFK(k_fma_f32, "v_fma_f32 %0, %1, %2, %0")
DEFK(k_mad_i24, "v_mad_i32_i24 %0, %1, %2, %0")
DEFK(k_mul_i24_sdwa, "v_mul_i32_i24_sdwa %0, %2, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1")
DEFK(k_add3, "v_add3_u32 %0, %1, %2, %0")
DEFK(k_alignbyte, "v_alignbyte_b32 %0, %1, %0, %2")
DEFK(k_perm, "v_perm_b32 %0, %1, %0, %2")
DEFK(k_cvt_ubyte, "v_cvt_f32_ubyte1_e32 %0, %1")
DEFK(k_ashr_pk, "v_ashr_pk_u8_i32 %0, %1, %2, 22")
DEFK(k_dot2_i32_i16, "v_dot2_i32_i16 %0, %1, %2, %0")
DEFK(k_dot4_i32_i8, "v_dot4_i32_i8 %0, %1, %2, %0")
DEFK(k_mul_lo_u32, "v_mul_lo_u32 %0, %1, %2")
DEFK(k_bfe, "v_bfe_u32 %0, %1, 8, 8")
DEFK(k_mad_u32_u24, "v_mad_u32_u24 %0, %1, %2, %0")
DEFK(k_lshl_or, "v_lshl_or_b32 %0, %1, 8, %0")
DEFK(k_fma_mix_lo, "v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]")
DEFK(k_fma_mix_hi, "v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]")
DEFK(k_cvt_f32_f16, "v_cvt_f32_f16_e32 %0, %1")
DEFK(k_cvt_f32_f16_sdwa, "v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1")
DEFK(k_mul_f32, "v_mul_f32_e32 %0, %1, %2")
DEFK(k_add_f32, "v_add_f32_e32 %0, %1, %0")
DEFK(k_cndmask, "v_cndmask_b32_e64 %0, %0, %1, vcc")
DEFK(k_lshlrev16, "v_lshlrev_b32_e32 %0, 16, %1")
DEFK(k_and_hi, "v_and_b32_e32 %0, 0xffff0000, %1")
DEFK(k_cndmask_e32, "v_cndmask_b32_e32 %0, %0, %1, vcc")
DEFK(k_or_b32, "v_or_b32_e32 %0, %1, %0")
DEFK(k_xor_b32, "v_xor_b32_e32 %0, %1, %0")
DEFK(k_max_f32, "v_max_f32_e32 %0, %1, %0")
DEFK(k_and_or, "v_and_or_b32 %0, %0, %1, %2")
DEFK(k_bfi, "v_bfi_b32 %0, %1, %0, %2")
DEFK(k_mov, "v_mov_b32_e32 %0, %1")
DEFK(k_sub_f32, "v_sub_f32_e32 %0, %1, %0")
DEFK(k_add_u32, "v_add_u32_e32 %0, %1, %0")
DEFK(k_lshrrev, "v_lshrrev_b32_e32 %0, 22, %0")
DEFK(k_mul_f32_sdwa, "v_mul_f32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD")

// 64-bit register pair version for pk_fma
__global__ void __launch_bounds__(256) k_pk_fma_f32_64(unsigned *out, int iters, float a, float b) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 r[8];
  for (int i = 0; i < 8; i++) r[i] = (f2){(float)threadIdx.x, (float)i};
  f2 x = {a, a + 1}, y = {b, b};
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(r[i]) : "v"(x), "v"(y));
#pragma unroll
    for (int i = 0; i < 8; i++) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(r[i]) : "v"(x), "v"(y));
  }
  float s = 0;
  for (int i = 0; i < 8; i++) s += r[i].x + r[i].y;
  if (s == 12345.f) out[0] = 1;
}

template <typename K, typename... A>
void run(const char *name, K kern, int waves_per_simd, A... args) {
  int iters = 4000;
  unsigned *d;
  hipMalloc(&d, 64);
  // 256 CUs x (waves_per_simd) blocks of 256 threads (= 4 waves = 1 per SIMD)
  dim3 grid(256 * waves_per_simd), block(256);
  hipLaunchKernelGGL(kern, grid, block, 0, 0, d, 10, args...);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, grid, block, 0, 0, d, iters, args...);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double instr_per_simd = (double)iters * 16 * waves_per_simd;  // wave-instructions each SIMD issued
  double ns_per = ms * 1e6 / instr_per_simd;
  printf("%-18s waves/SIMD=%d  %.3f ms  %.3f ns per wave-instr per SIMD  (= %.2f cyc @2.4GHz)\n", name, waves_per_simd, ms,
         ns_per, ns_per * 2.4);
  hipFree(d);
}

int main() {
  for (int w : {2, 4}) {
    run("v_fma_f32", k_fma_f32, w, 3u, 5u);
    run("v_mad_i32_i24", k_mad_i24, w, 3u, 5u);
    run("v_mul_i32_i24_sdwa", k_mul_i24_sdwa, w, 3u, 5u);
    run("v_add3_u32", k_add3, w, 3u, 5u);
    run("v_alignbyte_b32", k_alignbyte, w, 3u, 1u);
    run("v_perm_b32", k_perm, w, 3u, 0x03020100u);
    run("v_cvt_f32_ubyte1", k_cvt_ubyte, w, 3u, 5u);
    run("v_ashr_pk_u8_i32", k_ashr_pk, w, 3u, 5u);
    run("v_dot2_i32_i16", k_dot2_i32_i16, w, 3u, 5u);
    run("v_dot4_i32_i8", k_dot4_i32_i8, w, 3u, 5u);
    run("v_mul_lo_u32", k_mul_lo_u32, w, 3u, 5u);
    run("v_bfe_u32", k_bfe, w, 3u, 5u);
    run("v_mad_u32_u24", k_mad_u32_u24, w, 3u, 5u);
    run("v_lshl_or_b32", k_lshl_or, w, 3u, 5u);
    run("v_mul_f32_sdwa", k_mul_f32_sdwa, w, 3u, 5u);
    run("v_pk_fma_f32(64b)", k_pk_fma_f32_64, w, 1.0f, 0.5f);
    run("v_fma_mix_f32 lo", k_fma_mix_lo, w, 3u, 5u);
    run("v_fma_mix_f32 hi", k_fma_mix_hi, w, 3u, 5u);
    run("v_cvt_f32_f16", k_cvt_f32_f16, w, 3u, 5u);
    run("v_cvt_f32_f16_sdwa", k_cvt_f32_f16_sdwa, w, 3u, 5u);
    run("v_mul_f32", k_mul_f32, w, 3u, 5u);
    run("v_add_f32", k_add_f32, w, 3u, 5u);
    run("v_cndmask_b32", k_cndmask, w, 3u, 5u);
    run("v_lshlrev_b32 16", k_lshlrev16, w, 3u, 5u);
    run("v_and_b32 hi", k_and_hi, w, 3u, 5u);
    run("v_cndmask_b32_e32", k_cndmask_e32, w, 3u, 5u);
    run("v_or_b32", k_or_b32, w, 3u, 5u);
    run("v_xor_b32", k_xor_b32, w, 3u, 5u);
    run("v_max_f32", k_max_f32, w, 3u, 5u);
    run("v_and_or_b32", k_and_or, w, 3u, 5u);
    run("v_bfi_b32", k_bfi, w, 3u, 5u);
    run("v_mov_b32", k_mov, w, 3u, 5u);
    run("v_sub_f32", k_sub_f32, w, 3u, 5u);
    run("v_add_u32", k_add_u32, w, 3u, 5u);
    run("v_lshrrev_b32", k_lshrrev, w, 3u, 5u);
    printf("\n");
  }
  return 0;
}
