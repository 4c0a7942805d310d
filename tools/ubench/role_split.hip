// Two waves per SIMD, ONE 512-thread workgroup per CU: does a matrix-only wave overlap with a VALU-only wave on the same SIMD?
// (the premise of the 8-wave ping-pong attention, attention5.hip).  Phases of 28 MFMAs (32x32x16 bf16, 4 accumulators) vs
// {64 v_exp_f32, 32 v_cvt_pk_bf16_f32, 16 v_or3_b32} -- the per-tile work of the d = 40 attention -- in these arrangements:
//   0: every wave runs M then S back to back, no barriers                       (two independent streams per SIMD)
//   1: waves 0-3 run only M phases, waves 4-7 only S phases, no barriers         (pure role split: max(M, S) if they overlap)
//   2: as 1 with the roles exchanged (waves 0-3 = S)
//   3: ping-pong: waves 0-3 M while waves 4-7 S, s_barrier, swap, s_barrier ...  (attention5's structure)
//   4: M only on all 8 waves;  5: S only on all 8 waves;  6: M only on waves 0-3 (others exit);  7: S only on waves 0-3
//   8: as 3 with s_setprio 1 in the M phases;  9: as 3 with s_setprio 1 in the S phases
// Prints shader cycles per phase-pair (one M + one S of every wave), from wall time x the sclk estimate and from s_memtime.
// Build: hipcc --offload-arch=gfx950 -O3 role_split.hip -o role_split
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define ITERS 400

__device__ __forceinline__ void m_phase(f32x16 (&acc)[4], bf16x8_t fa, bf16x8_t fb) {
#pragma unroll
  for (int i = 0; i < 28; ++i) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[i & 3]) : "v"(fa), "v"(fb));
}
__device__ __forceinline__ void s_phase(float (&a)[16], unsigned& orv) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
#pragma unroll
    for (int i = 0; i < 16; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
#pragma unroll
    for (int i = 0; i < 16; i += 2) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(a[i]) : "v"(a[i + 1]));
#pragma unroll
    for (int i = 0; i < 16; i += 4) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(orv) : "v"(a[i]), "v"(a[i + 2]));
  }
}

template <int MODE>
__global__ __launch_bounds__(512, 2) void k(float* out, unsigned long long* cyc, float seed) {
  f32x16 acc[4];
  for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  bf16x8_t fa, fb;
  for (int r = 0; r < 8; ++r) { fa[r] = (__bf16)(seed + r); fb[r] = (__bf16)(seed - r); }
  float a[16];
  for (int i = 0; i < 16; ++i) a[i] = seed * 0.01f + threadIdx.x * 1e-6f + i * 1e-3f;
  unsigned orv = 0;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int grp = wave >> 2;
  if ((MODE == 6 || MODE == 7) && grp == 1) return;
  unsigned long long t0 = __builtin_readcyclecounter();
  if (MODE == 3 || MODE == 8 || MODE == 9) {
    if (grp == 1) __syncthreads();
    for (int it = 0; it < ITERS; ++it) {
      if (MODE == 8) __builtin_amdgcn_s_setprio(1);
      m_phase(acc, fa, fb);
      if (MODE == 8) __builtin_amdgcn_s_setprio(0);
      __syncthreads();
      if (MODE == 9) __builtin_amdgcn_s_setprio(1);
      s_phase(a, orv);
      if (MODE == 9) __builtin_amdgcn_s_setprio(0);
      __syncthreads();
    }
    if (grp == 0) __syncthreads();
  } else {
    for (int it = 0; it < ITERS; ++it) {
      const bool do_m = MODE == 0 || MODE == 4 || MODE == 6 || (MODE == 1 && grp == 0) || (MODE == 2 && grp == 1);
      const bool do_s = MODE == 0 || MODE == 5 || MODE == 7 || (MODE == 1 && grp == 1) || (MODE == 2 && grp == 0);
      if (do_m) m_phase(acc, fa, fb);
      if (do_s) s_phase(a, orv);
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < 16; ++i) s += a[i];
  for (int j = 0; j < 4; ++j) s += acc[j][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + orv;
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
}

template <int MODE> void run(const char* name) {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 16 * 1024 * 1024); hipMalloc(&cyc, 64); hipMemset(cyc, 0, 64);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, out, cyc, 1.0f);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, out, cyc, 1.0f);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[8]; hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
  printf("mode %d %-58s wall %8.1f us = %7.1f ns per iteration | s_memtime ticks per iteration: wave0 %7.1f wave4 %7.1f\n", MODE, name,
         ms * 1e3, ms * 1e6 / ITERS, (double)h[0] / ITERS, (double)h[4] / ITERS);
  hipFree(out); hipFree(cyc);
}
int main() {
  run<6>("M only, waves 0-3 (1 wave/SIMD)");
  run<7>("S only, waves 0-3 (1 wave/SIMD)");
  run<4>("M only, 8 waves");
  run<5>("S only, 8 waves");
  run<0>("M then S on every wave, no barriers");
  run<1>("waves 0-3 M only, waves 4-7 S only, no barriers");
  run<2>("waves 0-3 S only, waves 4-7 M only, no barriers");
  run<3>("ping-pong with barriers");
  run<8>("ping-pong, s_setprio 1 in M phases");
  run<9>("ping-pong, s_setprio 1 in S phases");
  return 0;
}
