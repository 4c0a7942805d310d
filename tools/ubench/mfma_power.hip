// What does the data movement of a GEMM K loop cost in CLOCK?  mfma_sustain.hip showed that the chip sustains pure
// v_mfma_f32_32x32x16_bf16 at 1.79 GHz (1.88 PFLOP/s = 0.75 of the 2.5 PFLOP/s spec, which assumes 2.4 GHz); the persistent GEMM
// kernel runs at 1.4-1.9 GHz with clock x pipe-busy ~ constant.  This program adds the K loop's traffic to the pure-MFMA loop,
// one ingredient at a time, at the kernel's own ratios per 40 MFMAs of a wave (one K-tile):
//   40 MFMAs; + 28 ds_read_b128 (7 fragment reads per 10 MFMAs); + 9 LDS-DMA pieces of 1 KB (L2-resident); both
// and two what-if mixes: the A rows loaded straight into registers (20 reads + 5 LDS-DMA pieces + 8 global_load_dwordx4), and
// the 128 x 160 wave tile of a 4-wave workgroup (80 MFMAs, 36 reads, 18 pieces per wave and K-tile).
// 256 workgroups x 8 waves, tens of milliseconds each; prints wall, TFLOP/s and the clock (s_memtime cycles of the younger
// wave 4 -- MFMA issue is oldest-first, wave 4 finishes last -- over wall time).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/mfma_power.hip -o tools/ubench/mfma_power
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int NRD, int NMF, int NDMA, int NGL>
__global__ __launch_bounds__(512, 1) void power_kernel(const char* __restrict__ src, int iters, float* sink, unsigned long long* cyc, unsigned seed) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  f32x16 acc[4];
  for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  bf16x8_t fa, fb;
  unsigned x = seed + threadIdx.x * 2654435761u + blockIdx.x * 40503u;
  for (int r = 0; r < 8; ++r) {
    x = x * 1664525u + 1013904223u; fa[r] = (__bf16)(((x >> 9) & 0xffff) / 65536.0f - 0.5f);
    x = x * 1664525u + 1013904223u; fb[r] = (__bf16)(((x >> 9) & 0xffff) / 65536.0f - 0.5f);
  }
  // pseudo-random LDS contents (so that the read data toggles), 128 KB
  for (int i = threadIdx.x; i < 32768; i += 512) { x = x * 1664525u + 1013904223u; reinterpret_cast<unsigned*>(smem)[i] = x; }
  __syncthreads();
  const unsigned rd0 = (unsigned)(size_t)smem + wave * 8192 + lane * 16;     // conflict-free 1-KB reads from the wave's 8-KB slice
  const unsigned wr0 = (unsigned)(size_t)smem + 65536 + wave * 8192;         // DMA target slices in the upper half
  const char* base = src + (size_t)(blockIdx.x & 63) * 65536 + lane * 16;
  unsigned pc = wave;
  u32x4 d0, d1, d2, d3, d4, d5, d6, d7, d8, g0, g1;
  unsigned keep = 0;
  constexpr bool LDSR = NRD > 0;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (NRD == 5) {
        asm volatile("ds_read_b128 %0, %5\n\tds_read_b128 %1, %5 offset:1024\n\tds_read_b128 %2, %5 offset:2048\n\tds_read_b128 %3, %5 offset:3072\n\t"
                     "ds_read_b128 %4, %5 offset:4096"
                     : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3), "=&v"(d4) : "v"(rd0) : "memory");
      } else if (NRD >= 7) {
        asm volatile("ds_read_b128 %0, %7\n\tds_read_b128 %1, %7 offset:1024\n\tds_read_b128 %2, %7 offset:2048\n\tds_read_b128 %3, %7 offset:3072\n\t"
                     "ds_read_b128 %4, %7 offset:4096\n\tds_read_b128 %5, %7 offset:5120\n\tds_read_b128 %6, %7 offset:6144"
                     : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3), "=&v"(d4), "=&v"(d5), "=&v"(d6) : "v"(rd0) : "memory");
        if (NRD == 9)
          asm volatile("ds_read_b128 %0, %2 offset:7168\n\tds_read_b128 %1, %2 offset:512" : "=&v"(d7), "=&v"(d8) : "v"(rd0) : "memory");
      }
      if (NDMA > 0) {
#pragma unroll
        for (int j = 0; j < (NDMA + 3 - ks) / 4; ++j) {                      // NDMA pieces spread over the four k-steps
          const char* a = base + (size_t)(pc & 63u) * 1024;
          const unsigned lds = __builtin_amdgcn_readfirstlane(wr0 + ((ks * 2 + j) & 7) * 1024);
          asm volatile("s_waitcnt vmcnt(8)\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds), "v"(a) : "memory");
          pc += 8;
        }
      }
      if (NGL > 0) {                                                         // operand rows straight into registers (2 per k-step)
        if (it | ks) {                                                       // ... consumed one k-step (10 MFMAs) after their issue
          asm volatile("s_waitcnt vmcnt(0)" : "+v"(g0), "+v"(g1)::"memory");
          keep ^= g0[0] ^ g1[1];
        }
        const char* a0 = base + (size_t)(pc & 63u) * 1024;
        const char* a1 = base + (size_t)((pc + 8) & 63u) * 1024;
        asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %3, off" : "=&v"(g0), "=&v"(g1) : "v"(a0), "v"(a1) : "memory");
        pc += 16;
      }
#pragma unroll
      for (int i = 0; i < NMF; ++i) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[i & 3]) : "v"(fa), "v"(fb));
      if (NRD == 5) {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4)::"memory");
        keep ^= d0[0] ^ d1[1] ^ d2[2] ^ d3[3] ^ d4[0];
      } else if (NRD >= 7) {
        // the destination registers stay allocated until the data has landed ("+v": the compiler must not reuse them earlier)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6)::"memory");
        keep ^= d0[0] ^ d1[1] ^ d2[2] ^ d3[3] ^ d4[0] ^ d5[1] ^ d6[2];
        if (NRD == 9) {
          asm volatile("" : "+v"(d7), "+v"(d8)::"memory");
          keep ^= d7[0] ^ d8[1];
        }
      }
    }
  }
  if (NGL > 0) asm volatile("s_waitcnt vmcnt(0)" : "+v"(g0), "+v"(g1)::"memory");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (blockIdx.x == 0 && lane == 0 && wave == (int)(blockDim.x >> 6) - 1) cyc[0] = t1 - t0;   // the youngest wave finishes last
  float s = 0.f;
  for (int j = 0; j < 4; ++j) s += acc[j][lane & 15];
  if (s == 12345.678f || keep == 0x12345u) sink[0] = s;
}

template <int NRD, int NMF, int NDMA, int NGL>
void run(const char* src, int waves, int iters, float* sink, unsigned long long* cyc, const char* what) {
  auto k = power_kernel<NRD, NMF, NDMA, NGL>;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k, dim3(256), dim3(64 * waves), 128 * 1024, 0, src, 64, sink, cyc, 1u);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k, dim3(256), dim3(64 * waves), 128 * 1024, 0, src, iters, sink, cyc, 7u);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long c = 0; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double mfmas = (double)iters * 4 * NMF * waves * 256;
  const double tf = mfmas * 32768.0 / (ms * 1e-3) / 1e12, ghz = (double)c / (ms * 1e-3) / 1e9;
  const double pipe = (double)iters * 4 * NMF * 32.0 * (waves / 4) / (double)c;
  printf("%-58s %8.2f ms  %7.1f TFLOP/s (%.2f of 2.5 PF)  clock %5.3f GHz  matrix pipe busy %4.2f\n", what, ms, tf, tf / 2500.0, ghz, pipe);
}

int main(int argc, char** argv) {
  char* src; float* sink; unsigned long long* cyc;
  hipMalloc(&src, 64 * 65536); hipMalloc(&sink, 64); hipMalloc(&cyc, 64);
  hipMemset(src, 0x5a, 64 * 65536);
  const int it = argc > 1 ? atoi(argv[1]) : 30000;      // 30000 K-tiles x 2560 pipe cycles ~ 40 ms
  for (int rep = 0; rep < 2; ++rep) {
    run<0, 10, 0, 0>(src, 8, it, sink, cyc, "8 waves: 40 MFMAs per wave and K-tile");
    run<7, 10, 0, 0>(src, 8, it, sink, cyc, "8 waves: + 28 ds_read_b128");
    run<0, 10, 9, 0>(src, 8, it, sink, cyc, "8 waves: + 9 LDS-DMA pieces");
    run<7, 10, 9, 0>(src, 8, it, sink, cyc, "8 waves: + 28 ds_read_b128 + 9 LDS-DMA   (the kernel's mix)");
    // what-if mixes for a next kernel generation
    run<5, 10, 5, 8>(src, 8, it, sink, cyc, "8 waves: A rows direct: 20 reads + 5 LDS-DMA + 8 global loads");
    run<9, 20, 18, 0>(src, 4, it / 2, sink, cyc, "4 waves (128x160 wave tile): 80 MFMAs, 36 reads, 18 LDS-DMA");
    run<9, 20, 0, 0>(src, 4, it / 2, sink, cyc, "4 waves: 80 MFMAs, 36 reads");
  }
  return 0;
}
