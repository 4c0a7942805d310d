// VALU issue-rate micro-benchmark (gfx950): cycles per wave64 instruction for the ops of the attention softmax.
// Build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP 64
#define ITERS 200
template <int OP>
__global__ void k(float* out, unsigned long long* cyc, float seed) {
  float a[8];
  typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
  typedef __attribute__((ext_vector_type(16))) float f32x16;
  f32x16 acc[4];
  for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  bf16x8_t fa, fb;
  for (int r = 0; r < 8; ++r) { fa[r] = (__bf16)(seed + r); fb[r] = (__bf16)(seed - r); }
  for (int i = 0; i < 8; ++i) a[i] = seed + threadIdx.x * 0.001f + i;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int r = 0; r < REP / 8; ++r) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (OP == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
        if (OP == 1) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a[i]));
        if (OP == 2) asm volatile("v_max3_f32 %0, %0, %0, %0" : "+v"(a[i]));
        if (OP == 3) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %0" : "+v"(a[i]));
        if (OP == 4) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[i]), "+v"(a[(i + 1) & 7]));
        if (OP == 5) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(*(double*)&a[i & 6]));
        if (OP == 6) asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(*(double*)&a[i & 6]));
        if (OP == 7) asm volatile("v_mul_f32 %0, %0, %0" : "+v"(a[i]));
        if (OP == 8) asm volatile("v_exp_f16 %0, %0" : "+v"(a[i]));
        if (OP == 9) asm volatile("v_sub_f32 %0, %0, %0" : "+v"(a[i]));
        if (OP == 10) acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[i & 3], 0, 0, 0);
        if (OP == 11 || OP == 12 || OP == 13) {       // co-issue test: every 4th slot an MFMA, the others VALU
          if ((i & 3) == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[(i >> 2) & 1]) : "v"(fa), "v"(fb));
          else if (OP == 11) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
          else if (OP == 12) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a[i]));
          else asm volatile("v_cvt_pk_bf16_f32 %0, %0, %0" : "+v"(a[i]));
        }
      }
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < 8; ++i) s += a[i];
  if (OP == 10) for (int j = 0; j < 4; ++j) s += acc[j][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int OP> void run(const char* name) {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 16 * 1024 * 1024); hipMalloc(&cyc, 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int waves_per_simd : {1, 2, 4, 8}) {
    // one workgroup of 4*waves_per_simd waves per CU (two workgroups of 16 waves for 8), 256 CUs
    const int blocks = waves_per_simd == 8 ? 512 : 256, threads = waves_per_simd == 8 ? 1024 : 256 * waves_per_simd;
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, out, cyc, 1.0f);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, out, cyc, 1.0f);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("  [wall %.1f us -> %.2f ticks/ns] ", ms * 1e3, (double)h / (ms * 1e6));
    // s_memtime counts at a constant 100 MHz reference on gfx9: report raw ticks per instruction too
    printf("%-22s waves/SIMD=%d  ticks=%llu  ticks/inst/wave=%.4f  ticks/inst (SIMD)=%.4f\n", name, waves_per_simd, h,
           (double)h / (ITERS * REP), (double)h / (ITERS * REP) / waves_per_simd);
  }
}
int main() {
  run<10>("mfma_32x32x16_bf16"); run<11>("1 mfma + 3 v_exp"); run<12>("1 mfma + 3 v_fma"); run<13>("1 mfma + 3 v_cvt_pk"); run<1>("v_fma_f32"); run<0>("v_exp_f32"); run<2>("v_max3_f32"); run<3>("v_cvt_pk_bf16_f32"); run<4>("v_permlane32_swap");
  run<5>("v_pk_fma_f32"); run<6>("v_pk_mul_f32"); run<7>("v_mul_f32"); run<8>("v_exp_f16"); run<9>("v_sub_f32");
  return 0;
}
