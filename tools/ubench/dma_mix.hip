// Does an LDS-DMA instruction hold its wave?  (dma_rate.hip's "5.6 B/clk per wave" came out of a loop whose address update
// was a 64-bit modulo -- ~180 cycles of VALU per piece -- so it may have measured the divide, not the memory pipe.)  Here the
// address update is one add + and, and the same loop is run as
//   mode 0: LDS-DMA only (W pieces in flight per wave),   mode 1: NM MFMAs per iteration only,
//   mode 2: one LDS-DMA piece + NM MFMAs per iteration  (NM = 4: the persistent GEMM's ratio, 9 pieces per 40 MFMAs)
//   mode 3: burst -- BURST pieces issued back to back with no wait, timed with s_memtime (issue cost per instruction)
// on `nwaves` waves of one 512-thread workgroup per CU, 256 workgroups.  If mode 2 ~ max(mode 0, mode 1) the pieces ride under
// the MFMAs; if ~ sum, the wave is held.  Prints wall ns per iteration and s_memtime cycles per iteration of wave 0.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/dma_mix.hip -o tools/ubench/dma_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int MODE, int NM, int W>
__global__ __launch_bounds__(512, 1) void mix_kernel(const char* __restrict__ src, int iters, int nwaves, float* sink,
                                                     unsigned long long* cyc, float seed) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (wave >= nwaves) return;
  // 64 regions of 64 KB, each shared by 4 CUs: L2-resident after the first touch
  const char* base = src + (size_t)(blockIdx.x & 63) * 65536 + lane * 16;
  unsigned pc = wave;                                                    // piece index (1 KB pieces, 64 per region)
  f32x16 acc[4];
  for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  bf16x8_t fa, fb;
  for (int r = 0; r < 8; ++r) { fa[r] = (__bf16)(seed + r); fb[r] = (__bf16)(seed - r); }
  const unsigned lds0 = (unsigned)(size_t)smem + wave * 16384;           // a 16-KB slice per wave, 16 slots of 1 KB
  const unsigned long long t0 = __builtin_readcyclecounter();
  if (MODE == 3) {
    constexpr int BURST = 9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < BURST; ++j) {
        const char* a = base + (size_t)(pc & 63u) * 1024;
        const unsigned lds = __builtin_amdgcn_readfirstlane(lds0 + (j & 15) * 1024);
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds), "v"(a) : "memory");
        pc += nwaves;
      }
#pragma unroll
      for (int i = 0; i < NM * BURST; ++i) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[i & 3]) : "v"(fa), "v"(fb));
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  } else {
    for (int it = 0; it < iters; ++it) {
      if (MODE == 0 || MODE == 2) {
        const char* a = base + (size_t)(pc & 63u) * 1024;
        const unsigned lds = __builtin_amdgcn_readfirstlane(lds0 + (it & 15) * 1024);
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off" ::"n"(W - 1), "s"(lds), "v"(a) : "memory");
        pc += nwaves;
      }
      if (MODE == 1 || MODE == 2) {
#pragma unroll
        for (int i = 0; i < NM; ++i) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[i & 3]) : "v"(fa), "v"(fb));
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (blockIdx.x == 0 && lane == 0 && wave == 0) cyc[0] = t1 - t0;
  float s = 0.f;
  for (int j = 0; j < 4; ++j) s += acc[j][lane & 15];
  if (s == 12345.678f) sink[0] = s + *reinterpret_cast<float*>(smem + lane * 4);
}

template <int MODE, int NM, int W>
void run(const char* src, int nwaves, float* sink, unsigned long long* cyc, const char* what) {
  const int iters = MODE == 3 ? 512 : 4096, grid = 256;
  auto k = mix_kernel<MODE, NM, W>;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k, dim3(grid), dim3(512), 128 * 1024, 0, src, iters, nwaves, sink, cyc, 1.0f);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k, dim3(grid), dim3(512), 128 * 1024, 0, src, iters, nwaves, sink, cyc, 1.0f);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long c = 0; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const int pieces = MODE == 3 ? 9 : ((MODE == 0 || MODE == 2) ? 1 : 0);
  const int mfmas = MODE == 3 ? 9 * NM : ((MODE == 1 || MODE == 2) ? NM : 0);
  const double cyc_it = (double)c / iters;
  printf("%-46s waves %d W %2d: wall %7.1f ns/iter, wave0 %7.1f cyc/iter", what, nwaves, W, ms * 1e6 / iters, cyc_it);
  if (pieces) printf(" | DMA %5.1f B/clk per CU, %5.1f cyc per piece and wave", pieces * 1024.0 * nwaves / cyc_it, cyc_it / pieces);
  if (mfmas) printf(" | MFMA pipe %4.2f", mfmas * 32.0 * ((nwaves + 3) / 4) / cyc_it);
  printf("\n");
}

int main() {
  char* src; float* sink; unsigned long long* cyc;
  hipMalloc(&src, 64 * 65536); hipMalloc(&sink, 64); hipMalloc(&cyc, 64);
  hipMemset(src, 1, 64 * 65536);
  for (int nw : {1, 4, 8}) {
    if (nw == 1) { run<0, 4, 4>(src, 1, sink, cyc, "mode 0 LDS-DMA only"); run<0, 4, 16>(src, 1, sink, cyc, "mode 0 LDS-DMA only"); }
    if (nw == 4) { run<0, 4, 4>(src, 4, sink, cyc, "mode 0 LDS-DMA only"); run<0, 4, 16>(src, 4, sink, cyc, "mode 0 LDS-DMA only"); }
    if (nw == 8) { run<0, 4, 4>(src, 8, sink, cyc, "mode 0 LDS-DMA only"); run<0, 4, 16>(src, 8, sink, cyc, "mode 0 LDS-DMA only"); }
  }
  run<1, 4, 4>(src, 4, sink, cyc, "mode 1 4 MFMAs only");
  run<1, 4, 4>(src, 8, sink, cyc, "mode 1 4 MFMAs only");
  run<2, 4, 8>(src, 4, sink, cyc, "mode 2 1 piece + 4 MFMAs");
  run<2, 4, 8>(src, 8, sink, cyc, "mode 2 1 piece + 4 MFMAs");
  run<2, 4, 16>(src, 8, sink, cyc, "mode 2 1 piece + 4 MFMAs");
  run<2, 8, 16>(src, 8, sink, cyc, "mode 2 1 piece + 8 MFMAs");
  run<2, 2, 16>(src, 8, sink, cyc, "mode 2 1 piece + 2 MFMAs");
  run<3, 4, 16>(src, 8, sink, cyc, "mode 3 burst of 9 pieces, then 36 MFMAs, wait");
  run<3, 4, 16>(src, 4, sink, cyc, "mode 3 burst of 9 pieces, then 36 MFMAs, wait");
  run<3, 0, 16>(src, 8, sink, cyc, "mode 3 burst of 9 pieces, wait (no MFMA)");
  run<3, 0, 16>(src, 1, sink, cyc, "mode 3 burst of 9 pieces, wait (no MFMA)");
  return 0;
}
