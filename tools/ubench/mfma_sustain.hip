// What matrix rate does the chip SUSTAIN?  256 workgroups x 8 waves issue nothing but v_mfma_f32_32x32x16_bf16 (4 independent
// accumulators per wave, pseudo-random operands so the datapath toggles) for tens of milliseconds; the shader clock the run
// averaged is s_memtime cycles / wall time, the rate 2*32*32*16 flops per MFMA / wall.  DUTY < 100: after every 4 MFMAs the wave
// idles with s_nop so that the matrix pipe is busy that fraction of the time -- does the clock come back up?
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/mfma_sustain.hip -o tools/ubench/mfma_sustain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int IDLE>   // IDLE = s_nop groups (16 cycles each, per wave) after every 4 MFMAs
__global__ __launch_bounds__(512, 1) void sustain(int iters, float* sink, unsigned long long* cyc, unsigned seed) {
  const int lane = threadIdx.x & 63;
  f32x16 acc[4];
  for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  bf16x8_t fa, fb;
  unsigned x = seed + threadIdx.x * 2654435761u + blockIdx.x * 40503u;
  for (int r = 0; r < 8; ++r) {
    x = x * 1664525u + 1013904223u; fa[r] = (__bf16)(((x >> 9) & 0xffff) / 65536.0f - 0.5f);
    x = x * 1664525u + 1013904223u; fb[r] = (__bf16)(((x >> 9) & 0xffff) / 65536.0f - 0.5f);
  }
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
#pragma unroll
      for (int i = 0; i < 4; ++i) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(fa), "v"(fb));
#pragma unroll
      for (int i = 0; i < IDLE; ++i) asm volatile("s_nop 15");
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
  float s = 0.f;
  for (int j = 0; j < 4; ++j) s += acc[j][lane & 15];
  if (s == 12345.678f) sink[0] = s;
}

template <int IDLE>
void run(int waves, int iters, float* sink, unsigned long long* cyc) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(sustain<IDLE>, dim3(256), dim3(64 * waves), 0, 0, 64, sink, cyc, 1u);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(sustain<IDLE>, dim3(256), dim3(64 * waves), 0, 0, iters, sink, cyc, 7u);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long c = 0; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double mfmas = (double)iters * 32 * waves * 256;
  const double tf = mfmas * 32768.0 / (ms * 1e-3) / 1e12;
  const double ghz = (double)c / (ms * 1e-3) / 1e9;
  const double pipe = (double)iters * 32 * 32.0 * ((waves + 3) / 4) / (double)c;
  printf("waves %d idle %2d: %8.2f ms  %7.1f TFLOP/s  clock %5.3f GHz  matrix pipe busy %4.2f of the cycles  (%.2f of 2.5 PF)\n", waves, IDLE,
         ms, tf, ghz, pipe, tf / 2500.0);
}

int main(int argc, char** argv) {
  float* sink; unsigned long long* cyc;
  hipMalloc(&sink, 64); hipMalloc(&cyc, 64);
  const int it = argc > 1 ? atoi(argv[1]) : 40000;      // 40000 x 32 MFMAs x 32 cycles x 2 waves/SIMD ~ 40 ms at 2 GHz
  run<0>(4, it, sink, cyc);
  run<0>(8, it, sink, cyc);
  run<0>(8, it * 4, sink, cyc);
  run<2>(8, it, sink, cyc);       // 4 MFMAs (128 pipe cycles) + 32 idle cycles per wave, two waves interleave: pipe ~ 100 %
  run<8>(8, it, sink, cyc);       // 128 busy + 128 idle per wave: two waves -> pipe ~ 100 % if they interleave, else 50 %
  run<8>(4, it, sink, cyc);       // one wave per SIMD: pipe 50 %
  run<24>(4, it, sink, cyc);      // pipe 25 %
  run<0>(8, it, sink, cyc);
  return 0;
}
