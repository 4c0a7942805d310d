// SUPERSEDED by dma_mix.hip: the loop below updates its address with `pc % pieces` on 64-bit values (~180 cycles of VALU per
// piece), which is what its per-wave numbers measured (5.6 B/clk; the real figure is 14 B/clk per wave, 64 B/clk per CU).  Kept
// because profiles/r02_ubench_dma_rate.log and the round-2 analysis in DESIGN.md cite it.
// How fast can one CU fill LDS from L2?  (the staging rate that bounds the GEMM / conv K-loop: a 256 x 320 tile needs
// 28.8 B/clk/CU at 100 % MFMA).  One 512-thread workgroup per CU, every wave streams 1-KiB pieces of an L2-resident region:
//   mode 0: global_load_lds_dwordx4 (LDS-DMA), window of W pieces in flight per wave
//   mode 1: global_load_dwordx4 into VGPRs only (batches of W)
//   mode 2: global_load_dwordx4 + ds_write_b128 (register-staged fill, batches of W)
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/dma_rate.hip -o tools/ubench/dma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int W>
__global__ __launch_bounds__(512, 1) void fill_kernel(const char* __restrict__ src, size_t region, int iters, int nwaves, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (wave >= nwaves) return;
  const char* base = src + (size_t)(blockIdx.x % 64) * region;          // 64 regions shared by 4 CUs each: L2 / MALL resident
  const size_t pieces = region / 1024;
  unsigned acc = 0;
  size_t pc = wave;
  if (MODE == 0) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < W; ++j) {
        const char* a = base + (pc % pieces) * 1024 + lane * 16;
        const unsigned lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(smem + ((j * 8 + wave) % 128) * 1024));
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off" ::"n"(W - 1), "s"(lds), "v"(a) : "memory");
        pc += nwaves;
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    for (int it = 0; it < iters; ++it) {
      u32x4 v[W];
#pragma unroll
      for (int j = 0; j < W; ++j) {
        const char* a = base + (pc % pieces) * 1024 + lane * 16;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v[j]) : "v"(a) : "memory");
        pc += nwaves;
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int j = 0; j < W; ++j) {
        if (MODE == 2) {
          *reinterpret_cast<u32x4*>(smem + ((j * 8 + wave) % 128) * 1024 + lane * 16) = v[j];
        } else {
          asm volatile("" ::"v"(v[j]));
        }
      }
      if (MODE == 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
  if (acc == 12345u) sink[0] = acc + *reinterpret_cast<unsigned*>(smem + lane * 4);
}

template <int MODE, int W>
void run(const char* src, size_t region, int nwaves, unsigned* sink) {
  const int iters = 4096 / W, grid = 256;
  auto k = fill_kernel<MODE, W>;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k, dim3(grid), dim3(512), 128 * 1024, 0, src, region, iters, nwaves, sink);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k, dim3(grid), dim3(512), 128 * 1024, 0, src, region, iters, nwaves, sink);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)grid * nwaves * iters * W * 1024.0;
  const double per_cu = bytes / grid / (ms * 1e-3);             // B/s per CU
  printf("mode %d W %2d waves %d region %4zu KB: %7.1f us  %6.2f TB/s chip  %5.1f GB/s per CU = %5.1f B/clk at 2.1 GHz\n", MODE, W, nwaves,
         region / 1024, ms * 1e3, bytes / (ms * 1e-3) / 1e12, per_cu / 1e9, per_cu / 2.1e9);
}

int main() {
  const size_t total = 256u << 20;
  char* src; unsigned* sink;
  hipMalloc(&src, total); hipMalloc(&sink, 64);
  hipMemset(src, 1, total);
  for (size_t region : {(size_t)64 << 10, (size_t)1 << 20}) {
    for (int nw : {1, 2, 4, 8}) {
      run<0, 4>(src, region, nw, sink);
      run<0, 8>(src, region, nw, sink);
      run<0, 16>(src, region, nw, sink);
    }
    for (int nw : {4, 8}) {
      run<1, 4>(src, region, nw, sink);
      run<1, 8>(src, region, nw, sink);
      run<2, 4>(src, region, nw, sink);
      run<2, 8>(src, region, nw, sink);
    }
  }
  return 0;
}
