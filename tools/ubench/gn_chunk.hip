// GroupNorm in sample chunks: does running stats + apply over <= ~100 MB at a time let the apply pass read its input from the
// 256 MB Infinity Cache instead of HBM?  Torch-free, through the C ABI (links libidf_gfx950.so).
// Build (from the repo root):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude tools/ubench/gn_chunk.hip -Linstancediffusion_amd -l:libidf_gfx950.so \
//         -Wl,-rpath,'$ORIGIN/../../instancediffusion_amd' -o tools/ubench/gn_chunk
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#include "idf.h"

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 5;
  struct S { int B, HW, C; } shapes[] = {{128, 4096, 320}, {128, 4096, 640}, {128, 4096, 960}, {128, 1024, 640}, {128, 1024, 1280}, {128, 256, 1280}, {128, 256, 2560}};
  const size_t maxe = (size_t)128 * 4096 * 960;
  unsigned short *x, *y, *src; float *g, *b, *ws;
  hipMalloc(&x, maxe * 2); hipMalloc(&y, maxe * 2); hipMalloc(&src, maxe * 2);
  hipMalloc(&g, 4096 * 4); hipMalloc(&b, 4096 * 4);
  {
    std::vector<unsigned short> h((size_t)16 << 20);
    unsigned s = 777u;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; const float f = (((s >> 8) & 0xffff) / 65536.0f - 0.4f) * 2.0f;
                        union { float f; unsigned u; } cv; cv.f = f; v = (unsigned short)(cv.u >> 16); }
    for (size_t off = 0; off < maxe; off += h.size()) hipMemcpy(src + off, h.data(), std::min(h.size(), maxe - off) * 2, hipMemcpyHostToDevice);
    std::vector<float> o(4096, 1.0f), z(4096, 0.1f);
    hipMemcpy(g, o.data(), 4096 * 4, hipMemcpyHostToDevice); hipMemcpy(b, z.data(), 4096 * 4, hipMemcpyHostToDevice);
  }
  hipMalloc(&ws, (size_t)idf_groupnorm_ws_floats(128, 4096) * 4 + (1 << 20));
  hipMemset(ws, 0, (size_t)idf_groupnorm_ws_floats(128, 4096) * 4 + (1 << 20));       // idf_groupnorm's contract (include/idf.h)
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (const S& sh : shapes) {
    const size_t n = (size_t)sh.B * sh.HW * sh.C;
    printf("groupnorm (%d, %d, %d): %.0f MB in, %.0f MB out\n", sh.B, sh.HW, sh.C, n * 2e-6, n * 2e-6);
    for (int chunks : {1, 2, 4, 8, 16, 32}) {
      const int bs = sh.B / chunks;
      std::vector<double> t;
      for (int rep = 0; rep < reps; ++rep) {
        hipMemcpyAsync(x, src, n * 2, hipMemcpyDeviceToDevice, 0);       // the "producer": x freshly written, like a conv's output
        hipEventRecord(e0, 0);
        for (int c = 0; c < chunks; ++c) {
          const size_t off = (size_t)c * bs * sh.HW * sh.C;
          int rc = idf_groupnorm(x + off, y + off, g, b, ws, bs, sh.HW, sh.C, 1e-5f, 1, IDF_BF16, nullptr);
          if (rc) { printf("rc %d\n", rc); return 1; }
        }
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        t.push_back(ms * 1e3);
      }
      std::sort(t.begin(), t.end());
      const double us = t[t.size() / 2];
      printf("    %2d chunk(s) of %3d samples (%6.1f MB): %8.1f us   %.2f TB/s on algorithmic bytes (2 B in + 2 B out)\n", chunks, bs,
             (double)bs * sh.HW * sh.C * 2e-6, us, n * 4.0 / (us * 1e-6) / 1e12);
    }
  }
  return 0;
}
