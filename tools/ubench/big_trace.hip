// Where does the persistent GEMM / conv kernel (gemm_big.hip) spend its cycles?  Torch-free: builds the kernel file into
// this program with -DIDF_BIG_TRACE, launches the shapes of the 64-row UNet forward through idf_launch_big and prints, next to
// the HIP-event time of the launch, the s_memtime cycles per segment of waves 0 (enqueues the next K-tile right behind the
// barrier) and 4 (enqueues it from the middle of its MFMAs) of the first and of a middle workgroup:
//   per K-tile: vmwait (s_waitcnt vmcnt before the barrier), barrier, fill (early K-tile enqueue), compute (fragment reads +
//   MFMA issue + late enqueue);  per tile: head (accumulator clear), epilogue.
// 40 MFMAs of 32 cycles per wave and K-tile, two waves per SIMD: a K-tile period of 2560 cycles = 100 % matrix pipe.
// Build (from the repo root; ~1 min):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -DIDF_BIG_TRACE -Iinclude -Iinstancediffusion_amd/csrc \
//         tools/ubench/big_trace.hip -o tools/ubench/big_trace
// Without -DIDF_BIG_TRACE the program only times the launches (and prints the output checksums): the A/B harness for K-loop
// schedule variants (-DIDF_LATE_OLD=0|1 -DIDF_LATE_NUM=.. -DIDF_LATE_DEN=..: which waves enqueue late, and where), whose outputs must be bit-identical.
// Run: tools/ubench/big_trace [reps]     (about 10 s on the GPU; no Python)
#include "../../instancediffusion_amd/csrc/gemm_big.hip"
#include <cstdio>
#include <cstring>
#include <vector>

__global__ void checksum_kernel(const unsigned* x, size_t n, unsigned long long* out) {
  unsigned long long acc = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    acc += (unsigned long long)x[i] * (unsigned long long)((i & 0xffff) + 1);
  atomicAdd(out, acc);
}

struct Shape { const char* name; int M, N, K; int epi; bool conv; int B, H, Cin; int stride = 1, up = 0; };

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 10;
  const size_t max_elems = (size_t)262144 * 2560;               // largest A / out / residual of the list
  unsigned short *a, *w, *o, *r;
  float *bias, *ws;
  unsigned long long* csum; hipMalloc(&csum, 8);
  hipMalloc(&a, max_elems * 2); hipMalloc(&o, max_elems * 2); hipMalloc(&r, max_elems * 2);
  hipMalloc(&w, (size_t)64 << 20 << 1); hipMalloc(&bias, 16384 * 4); hipMalloc(&ws, (size_t)256 << 20);
  {
    std::vector<unsigned short> h((size_t)32 << 20);
    unsigned x = 12345u;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; const float f = (((x >> 8) & 0xffff) / 65536.0f - 0.5f) * 0.25f;
                        union { float f; unsigned u; } cv; cv.f = f; v = (unsigned short)(cv.u >> 16); }
    for (size_t off = 0; off < max_elems; off += h.size()) {
      const size_t n = std::min(h.size(), max_elems - off);
      hipMemcpy(a + off, h.data(), n * 2, hipMemcpyHostToDevice);
      hipMemcpy(r + off, h.data(), n * 2, hipMemcpyHostToDevice);
    }
    for (size_t off = 0; off < ((size_t)64 << 20); off += h.size()) hipMemcpy(w + off, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipMemset(bias, 0, 16384 * 4);
  }
  const int BR = IDF_EPI_BIAS | IDF_EPI_RES;
  const Shape shapes[] = {
      {"proj 64^2 plain", 262144, 320, 320, 0, false, 0, 0, 0},
      {"proj 64^2 bias+res", 262144, 320, 320, BR, false, 0, 0, 0},
      {"proj 32^2 plain", 65536, 640, 640, 0, false, 0, 0, 0},
      {"proj 32^2 bias+res", 65536, 640, 640, BR, false, 0, 0, 0},
      {"proj 16^2 plain", 16384, 1280, 1280, 0, false, 0, 0, 0},
      {"proj 16^2 bias+res", 16384, 1280, 1280, BR, false, 0, 0, 0},
      {"ff-out 64^2 bias+res", 262144, 320, 1280, BR, false, 0, 0, 0},
      {"ff-out 32^2 bias+res", 65536, 640, 2560, BR, false, 0, 0, 0},
      {"ff-out 16^2 bias+res", 16384, 1280, 5120, BR, false, 0, 0, 0},
      {"qk 64^2 bias", 262144, 640, 320, IDF_EPI_BIAS, false, 0, 0, 0},
      {"geglu-in(as plain) 64^2", 262144, 2560, 320, IDF_EPI_BIAS, false, 0, 0, 0},
      {"square 8k", 8192, 8192, 8192, 0, false, 0, 0, 0},
      {"conv 64^2 320 bias", 0, 320, 0, IDF_EPI_BIAS, true, 64, 64, 320},
      {"conv 32^2 640 bias", 0, 640, 0, IDF_EPI_BIAS, true, 64, 32, 640},
      {"conv 16^2 1280 bias", 0, 1280, 0, IDF_EPI_BIAS, true, 64, 16, 1280},
      {"conv 32^2 640 stride 2", 0, 640, 0, IDF_EPI_BIAS, true, 64, 32, 640, 2, 0},
      {"conv 16^2->32^2 1280 up", 0, 1280, 0, IDF_EPI_BIAS, true, 64, 16, 1280, 1, 1},
      {"conv 64^2 640->320 +res", 0, 320, 0, IDF_EPI_BIAS | IDF_EPI_RES, true, 64, 64, 640},
  };
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (const Shape& sh : shapes) {
    CoreParams p{};
    if (sh.conv) {
      const int hup = sh.H << sh.up;
      p.Ho = (hup - 1) / sh.stride + 1; p.Wo = p.Ho; p.Hin = sh.H; p.Win = sh.H; p.Cin = sh.Cin; p.stride = sh.stride; p.up = sh.up;
      p.M = sh.B * p.Ho * p.Wo; p.K = 9 * sh.Cin; p.lda = sh.Cin; p.ldw = 9 * sh.Cin; p.rows_per_batch = p.Ho * p.Wo;
    } else {
      p.M = sh.M; p.K = sh.K; p.lda = sh.K; p.ldw = sh.K; p.rows_per_batch = sh.M;
    }
    p.N = sh.N; p.n_valid = sh.N; p.W = w; p.A = a; p.out = o; p.ldo = sh.N; p.res = r; p.ldr = sh.N; p.bias = bias; p.epi = sh.epi;
    p.ws = ws; p.ws_bytes = (size_t)256 << 20;
    int rc = idf_launch_big(p, IDF_BF16, sh.conv, true, 0, nullptr);
    if (rc) { printf("%-26s launch rc %d\n", sh.name, rc); continue; }
    idf_launch_big(p, IDF_BF16, sh.conv, true, 0, nullptr);
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) idf_launch_big(p, IDF_BF16, sh.conv, true, 0, nullptr);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    // checksum of the 16-bit output (every schedule / prefetch variant must reproduce it bit for bit)
    hipMemsetAsync(csum, 0, 8, 0);
    hipLaunchKernelGGL(checksum_kernel, dim3(1024), dim3(256), 0, 0, reinterpret_cast<const unsigned*>(o), (size_t)p.M * p.N / 2, csum);
    unsigned long long cs = 0; hipMemcpy(&cs, csum, 8, hipMemcpyDeviceToHost);
    const double us = ms * 1e3 / reps, tf = 2.0 * p.M * (double)p.N * p.K / (us * 1e-6) / 1e12;
    const int bn = (sh.N % 320 == 0) ? 320 : 256;
    const long tiles = (long)(sh.N / bn) * ((p.M + 255) / 256);
    printf("%-26s M%-7d N%-5d K%-5d tiles %5ld (%.2f rounds)  %8.1f us  %7.1f TF\n", sh.name, p.M, p.N, p.K, tiles, tiles / 256.0, us, tf);
    printf("    checksum %016llx\n", cs);
#ifdef IDF_BIG_TRACE
    unsigned long long tr[4][8];
    hipMemcpyFromSymbol(tr, HIP_SYMBOL(idf_big_trace_buf), sizeof(tr));
    const char* who[4] = {"wg0 wave0", "wg0 wave4", "wgM wave0", "wgM wave4"};
    for (int g = 0; g < 4; ++g) {
      const double nk = (double)tr[g][6], nt = (double)tr[g][7];
      if (nk == 0 || nt == 0) continue;
      const double per_kt = (tr[g][0] + tr[g][1] + tr[g][2] + tr[g][3]) / nk;
      const double total = (double)(tr[g][0] + tr[g][1] + tr[g][2] + tr[g][3] + tr[g][4] + tr[g][5]);
      printf("    %s  per K-tile: vmwait %5.0f barrier %5.0f fill %5.0f compute %5.0f = %5.0f (2560 = pipe full) | per tile (%2.0f K-tiles): head %5.0f "
             "epilogue %6.0f | tiles %3.0f, total %8.0f cyc, K-loop share %.2f\n",
             who[g], tr[g][0] / nk, tr[g][1] / nk, tr[g][2] / nk, tr[g][3] / nk, per_kt, nk / nt, tr[g][5] / nt, tr[g][4] / nt, nt, total,
             (total - tr[g][4] - tr[g][5]) / total);
    }
#endif
  }
  return 0;
}
