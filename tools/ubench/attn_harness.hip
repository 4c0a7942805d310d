// Torch-free A/B harness for idf_attention (the C ABI of libidf_gfx950.so): the attention shapes of the 64-row UNet forward, laid
// out as the engine passes them (q | k as column slices of the fused projection buffer, V^T batch-interleaved), timed with HIP
// events per IDF_TUNE_ATTN2 mode, with a checksum of the output per (shape, mode) -- a GPU call with it costs seconds, not the
// minutes of tools/attn_ab.py (Python + torch import).  The library is dlopen'ed by path, so two builds can be compared in one run.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude tools/ubench/attn_harness.hip -o tools/ubench/attn_harness -ldl
// Run:   tools/ubench/attn_harness instancediffusion_amd/libidf_gfx950.so [batch=64] [modes=1,2,0] [other.so ...]
// A library built with -DIDF_ATTN_TRACE on attention4.hip (instancediffusion_amd/csrc/build.sh has the flags; link the other
// objects unchanged) additionally exports idf_attn_trace_read: the harness then prints the s_memtime cycles per segment of the
// d = 40 kernel's common-path tile.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <string>
#include "idf.h"

__global__ void fill_kernel(unsigned short* x, size_t n, unsigned seed, float scale) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u + seed;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    const float f = (((h >> 8) & 0xffff) / 65536.0f - 0.5f) * scale;
    x[i] = (unsigned short)(__float_as_uint(f) >> 16);                      // bf16 by truncation
  }
}
__global__ void checksum_kernel(const unsigned* x, size_t n, unsigned long long* out) {
  unsigned long long acc = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    acc += (unsigned long long)x[i] * (unsigned long long)((i & 0xffff) + 1);
  atomicAdd(out, acc);
}

struct Lib {
  void* h; std::string path;
  int (*attention)(const idf_attn_args*, void*);
  int (*set_tuning)(int, int);
  int (*trace_read)(unsigned long long*);          // only in a -DIDF_ATTN_TRACE build of the library
};

struct Shape { const char* name; int N, d, n0, n1; };

int main(int argc, char** argv) {
  if (argc < 2) { fprintf(stderr, "usage: attn_harness lib.so [batch] [modes] [more libs]\n"); return 2; }
  const int B = argc > 2 ? atoi(argv[2]) : 64;
  std::vector<int> modes;
  { std::string m = argc > 3 ? argv[3] : "1,2,0"; size_t p = 0; while (p < m.size()) { modes.push_back(atoi(m.c_str() + p)); p = m.find(',', p); if (p == std::string::npos) break; ++p; } }
  std::vector<Lib> libs;
  for (int i = 1; i < argc; ++i) {
    if (i == 2 || i == 3) continue;
    Lib l; l.path = argv[i]; l.h = dlopen(argv[i], RTLD_NOW | RTLD_LOCAL);
    if (!l.h) { fprintf(stderr, "dlopen %s: %s\n", argv[i], dlerror()); return 2; }
    l.attention = (int (*)(const idf_attn_args*, void*))dlsym(l.h, "idf_attention");
    l.set_tuning = (int (*)(int, int))dlsym(l.h, "idf_set_tuning");
    l.trace_read = (int (*)(unsigned long long*))dlsym(l.h, "idf_attn_trace_read");
    if (!l.attention || !l.set_tuning) { fprintf(stderr, "%s: missing symbols\n", argv[i]); return 2; }
    libs.push_back(l);
  }
  const int H = 8;
  const Shape shapes[] = {{"self 64^2", 4096, 40, 4096, 0},   {"gated 64^2", 4096, 40, 4096, 184}, {"gated 32^2", 1024, 80, 1024, 184},
                          {"gated 16^2", 256, 160, 256, 184}, {"cross 64^2", 4096, 40, 77, 0},      {"cross 32^2", 1024, 80, 77, 0}};
  const size_t maxC = 1280, maxTok = (size_t)B * 4096;
  unsigned short *qk, *vt, *k1, *vt1, *kc, *vtc, *o;
  unsigned long long* csum;
  hipMalloc(&qk, maxTok * 640 * 2); hipMalloc(&vt, maxTok * 320 * 2); hipMalloc(&o, maxTok * 320 * 2);
  hipMalloc(&k1, (size_t)B * 184 * maxC * 2); hipMalloc(&vt1, (size_t)B * maxC * 192 * 2);
  hipMalloc(&kc, (size_t)B * 77 * maxC * 2); hipMalloc(&vtc, (size_t)B * maxC * 128 * 2); hipMalloc(&csum, 8);
  hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, 0, qk, maxTok * 640, 1u, 1.0f);
  hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, 0, vt, maxTok * 320, 2u, 1.0f);
  hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, 0, k1, (size_t)B * 184 * maxC, 3u, 1.0f);
  hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, 0, vt1, (size_t)B * maxC * 192, 4u, 1.0f);
  hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, 0, kc, (size_t)B * 77 * maxC, 5u, 1.0f);
  hipMemset(vtc, 0, (size_t)B * maxC * 128 * 2);
  hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, 0, vtc, (size_t)B * maxC * 128, 6u, 1.0f);   // (pad columns beyond 77 are never read)
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (const Shape& sh : shapes) {
    const int C = H * sh.d, N = sh.N;
    idf_attn_args a; memset(&a, 0, sizeof(a));
    a.q = qk; a.ldq = 2 * C; a.strideQ = (long long)N * 2 * C; a.nq = N;
    if (sh.n0 == 77) {
      a.k0 = kc; a.ldk0 = C; a.strideK0 = 77LL * C; a.vt0 = vtc; a.ldv0 = 128; a.strideV0 = (long long)C * 128; a.n0 = 77;
    } else {
      a.k0 = qk + C; a.ldk0 = 2 * C; a.strideK0 = (long long)N * 2 * C;
      a.vt0 = vt; a.ldv0 = B * N; a.strideV0 = N; a.n0 = N;                    // V^T[c][b][token]: sample b through base + b N, ld = B N
    }
    if (sh.n1) { a.k1 = k1; a.ldk1 = C; a.strideK1 = 184LL * C; a.vt1 = vt1; a.ldv1 = 192; a.strideV1 = (long long)C * 192; a.n1 = 184; }
    a.out = o; a.ldo = C; a.strideO = (long long)N * C;
    a.B = B; a.H = H; a.d = sh.d; a.scale = 1.0f / sqrtf((float)sh.d); a.dtype = IDF_BF16;
    const double flops = 4.0 * B * (double)N * (sh.n0 + sh.n1) * C;
    for (int w = 0; w < 3; ++w) libs[0].attention(&a, nullptr);               // clock / cache warm-up
    for (size_t li = 0; li < libs.size(); ++li)
      for (int m : modes) {
        if (libs[li].set_tuning(IDF_TUNE_ATTN2, m) < 0) continue;
        double best = 1e30;
        int rc = 0;
        for (int rep = 0; rep < 3; ++rep) {
          rc |= libs[li].attention(&a, nullptr);
          hipEventRecord(e0, 0);
          for (int i = 0; i < 10; ++i) rc |= libs[li].attention(&a, nullptr);
          hipEventRecord(e1, 0);
          hipDeviceSynchronize();
          float ms = 0; hipEventElapsedTime(&ms, e0, e1);
          if (ms / 10 < best) best = ms / 10;
        }
        hipMemsetAsync(csum, 0, 8, 0);
        hipLaunchKernelGGL(checksum_kernel, dim3(1024), dim3(256), 0, 0, reinterpret_cast<const unsigned*>(o), (size_t)B * N * C / 2, csum);
        unsigned long long cs = 0; hipMemcpy(&cs, csum, 8, hipMemcpyDeviceToHost);
        printf("%-12s d=%-3d keys %4d+%-3d  lib %zu mode %d: rc %d  %8.1f us  %7.1f TF  checksum %016llx\n", sh.name, sh.d, sh.n0, sh.n1, li, m, rc,
               best * 1e3, flops / (best * 1e-3) / 1e12, cs);
        if (libs[li].trace_read && sh.d == 40 && sh.n0 > 77 && m >= 1) {
          unsigned long long tr[2][10];
          if (libs[li].trace_read(&tr[0][0]) == 0) {
            const char* seg[9] = {"dma", "qk", "exp0", "pv0", "exp1", "kfrag", "pv1", "wait+bar", "check"};
            for (int g = 0; g < 2; ++g) {
              const double nt = (double)tr[g][9];
              if (nt == 0) continue;
              double tot = 0;
              printf("    %s wave 0, per common-path tile (%.0f tiles; 896 cycles of MFMA each):", g ? "wgM" : "wg0", nt);
              for (int i = 0; i < 9; ++i) { printf(" %s %.0f", seg[i], tr[g][i] / nt); tot += tr[g][i] / nt; }
              printf(" = %.0f\n", tot);
            }
          }
        }
        libs[li].set_tuning(IDF_TUNE_ATTN2, 1);
      }
  }
  return 0;
}
