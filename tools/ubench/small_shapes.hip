// A/B harness for the small-batch GEMM / conv launches through the C ABI, torch-free: every dense GEMM and 3x3 conv shape of
// an R-row UNet forward (R = 2: BASELINE config 2; R = 16: the second MIS phase of the reference's own 8-image batch) is
// launched with the latency kernel off (IDF_TUNE_GEMM_RING = 0: K-loop variants 1 / 2) and on (threshold = every tile grid),
// interleaved per shape in one process.  Prints the median time of each, the launch weight of the shape in one forward and
// how far the two outputs are apart (same tiles and K order: bit-identical unless the split-K choice differs), then the
// forward-weighted totals per threshold candidate.
// Build (from the repo root):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude tools/ubench/small_shapes.hip -Linstancediffusion_amd -l:libidf_gfx950.so \
//         -Wl,-rpath,'$ORIGIN/../../instancediffusion_amd' -o tools/ubench/small_shapes
// Run: tools/ubench/small_shapes [reps] [rounds]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
#include "idf.h"

__global__ void diff_kernel(const unsigned short* a, const unsigned short* b, size_t n, unsigned long long* mism, float* maxabs, float* maxref) {
  unsigned long long c = 0; float mx = 0.f, mr = 0.f;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float x = __uint_as_float((unsigned)a[i] << 16), y = __uint_as_float((unsigned)b[i] << 16);
    if (a[i] != b[i]) ++c;
    if (!(fabsf(x - y) <= mx)) mx = fabsf(x - y);              // NaN propagates
    mr = fmaxf(mr, fabsf(y));
  }
  atomicAdd(mism, c);
  atomicMax(reinterpret_cast<unsigned*>(maxabs), __float_as_uint(mx));      // non-negative floats order like their bits
  atomicMax(reinterpret_cast<unsigned*>(maxref), __float_as_uint(mr));
}

struct Shape { const char* name; int count; bool conv; int M, N, K, epi; int H, Cin, stride, up; };

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 20;
  const int rounds = argc > 2 ? atoi(argv[2]) : 3;
  const size_t max_elems = (size_t)16 * 4096 * 2560;            // largest operand / output (16 rows, GEGLU output at 64^2)
  unsigned short *a, *w, *o0, *o1, *r;
  float *bias, *ws, *stats, *lnc;
  unsigned long long* mism; float *maxabs, *maxref;
  hipMalloc(&mism, 8); hipMalloc(&maxabs, 4); hipMalloc(&maxref, 4);
  hipMalloc(&a, max_elems * 2); hipMalloc(&o0, max_elems * 2); hipMalloc(&o1, max_elems * 2); hipMalloc(&r, max_elems * 2);
  hipMalloc(&w, (size_t)64 << 20 << 1); hipMalloc(&bias, 32768 * 4); hipMalloc(&lnc, 32768 * 4); hipMalloc(&ws, (size_t)256 << 20);
  hipMalloc(&stats, (size_t)16 * 4096 * 2 * 4);
  {
    std::vector<unsigned short> h((size_t)16 << 20);
    unsigned x = 12345u;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; const float f = (((x >> 8) & 0xffff) / 65536.0f - 0.5f) * 0.25f;
                        unsigned u; memcpy(&u, &f, 4); v = (unsigned short)(u >> 16); }
    for (size_t off = 0; off < max_elems; off += h.size()) {
      const size_t n = std::min(h.size(), max_elems - off);
      hipMemcpy(a + off, h.data(), n * 2, hipMemcpyHostToDevice);
      hipMemcpy(r + off, h.data() + 7, (n - 7) * 2, hipMemcpyHostToDevice);
    }
    for (size_t off = 0; off < ((size_t)64 << 20); off += h.size()) hipMemcpy(w + off, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    std::vector<float> hb(32768);
    for (auto& v : hb) { x = x * 1664525u + 1013904223u; v = (((x >> 8) & 0xffff) / 65536.0f - 0.5f) * 0.5f; }
    hipMemcpy(bias, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
    for (auto& v : hb) { x = x * 1664525u + 1013904223u; v = (((x >> 8) & 0xffff) / 65536.0f - 0.5f) * 0.1f; }
    hipMemcpy(lnc, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
    std::vector<float> hs((size_t)16 * 4096 * 2);
    for (size_t i = 0; i < hs.size(); i += 2) { x = x * 1664525u + 1013904223u; hs[i] = (((x >> 8) & 0xffff) / 65536.0f - 0.5f) * 0.02f; hs[i + 1] = 0.9f + (x & 255) / 1024.0f; }
    hipMemcpy(stats, hs.data(), hs.size() * 4, hipMemcpyHostToDevice);
  }
  const int B_ = IDF_EPI_BIAS, BR = IDF_EPI_BIAS | IDF_EPI_RES, GLU = IDF_EPI_BIAS | IDF_EPI_GEGLU | IDF_EPI_GEGLU_P32;
  const int LNB = IDF_EPI_BIAS | IDF_EPI_LN_ROW;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int thresholds[] = {128, 256, 512, 1024, 1 << 30};
  for (int R : {2, 16}) {
    std::vector<Shape> shapes;
    const int HW[4] = {4096, 1024, 256, 64}, CH[4] = {320, 640, 1280, 1280}, HH[4] = {64, 32, 16, 8};
    // transformer layers per level (down + up): 64^2: 2 + 3, 32^2: 2 + 3, 16^2: 2 + 3, 8^2: the middle block's one
    const int layers[4] = {5, 5, 5, 1};
    static char names[64][48]; int ni = 0;
    auto nm = [&](const char* f, int l) { snprintf(names[ni], 48, f, HH[l]); return names[ni++]; };
    for (int l = 0; l < 4; ++l) {
      const int M = R * HW[l], C = CH[l], L = layers[l];
      shapes.push_back({nm("proj-in/out %d^2 bias+res", l), 2 * L, false, M, C, C, BR, 0, 0, 1, 0});
      shapes.push_back({nm("q|k LN %d^2", l), 2 * L, false, M, 2 * C, C, LNB, 0, 0, 1, 0});
      shapes.push_back({nm("V^T %d^2 (M = C)", l), 2 * L, false, C, M, C, 0, 0, 0, 1, 0});
      shapes.push_back({nm("attn out / q %d^2 bias+res", l), 4 * L, false, M, C, C, BR, 0, 0, 1, 0});
      shapes.push_back({nm("geglu %d^2", l), L, false, M, 8 * C, C, GLU, 0, 0, 1, 0});
      shapes.push_back({nm("ff-out %d^2 bias+res", l), L, false, M, C, 4 * C, BR, 0, 0, 1, 0});
    }
    // ResBlock convs (counts of the SD-1.5 UNet: 22 ResBlocks, 3 down, 3 up)
    shapes.push_back({"conv 64^2 320->320", 7, true, 0, 320, 0, B_, 64, 320, 1, 0});
    shapes.push_back({"conv 64^2 640->320 +res", 2, true, 0, 320, 0, BR, 64, 640, 1, 0});
    shapes.push_back({"conv 64^2 960->320", 1, true, 0, 320, 0, B_, 64, 960, 1, 0});
    shapes.push_back({"conv 32^2 640->640", 6, true, 0, 640, 0, B_, 32, 640, 1, 0});
    shapes.push_back({"conv 32^2 1280->640", 2, true, 0, 640, 0, B_, 32, 1280, 1, 0});
    shapes.push_back({"conv 32^2 1920->640", 1, true, 0, 640, 0, B_, 32, 1920, 1, 0});
    shapes.push_back({"conv 16^2 1280->1280", 6, true, 0, 1280, 0, B_, 16, 1280, 1, 0});
    shapes.push_back({"conv 16^2 2560->1280", 2, true, 0, 1280, 0, B_, 16, 2560, 1, 0});
    shapes.push_back({"conv 8^2 1280->1280 +res", 11, true, 0, 1280, 0, BR, 8, 1280, 1, 0});
    shapes.push_back({"conv 8^2 2560->1280", 3, true, 0, 1280, 0, B_, 8, 2560, 1, 0});
    shapes.push_back({"conv 64^2 320 stride 2", 1, true, 0, 320, 0, B_, 64, 320, 2, 0});
    shapes.push_back({"conv 16^2 1280 stride 2", 1, true, 0, 1280, 0, B_, 16, 1280, 2, 0});
    shapes.push_back({"conv 16^2->32^2 1280 up", 1, true, 0, 1280, 0, B_, 16, 1280, 1, 1});
    shapes.push_back({"conv 8^2->16^2 1280 up", 1, true, 0, 1280, 0, B_, 8, 1280, 1, 1});
    shapes.push_back({"time-emb 22 x emb_layers", 1, false, R, 20160, 1280, B_, 0, 0, 1, 0});

    printf("==== %d-row forward\n", R);
    double tot0 = 0, tot1 = 0, tot_thr[5] = {0, 0, 0, 0, 0};
    for (const Shape& sh : shapes) {
      idf_gemm_args g{}; idf_conv3x3_args c{};
      int M = sh.M, K = sh.K, n_out = (sh.epi & IDF_EPI_GEGLU) ? sh.N / 2 : sh.N;
      if (sh.conv) {
        const int hup = sh.H << sh.up, ho = (hup - 1) / sh.stride + 1;
        M = R * ho * ho; K = 9 * sh.Cin;
        c.x = a; c.W = w; c.bias = bias; c.res = r; c.B = R; c.Hin = sh.H; c.Win = sh.H; c.Cin = sh.Cin; c.Cout = sh.N;
        c.stride = sh.stride; c.upsample = sh.up; c.ldx = sh.Cin; c.ldo = sh.N; c.ldr = sh.N; c.epi = sh.epi; c.dtype = IDF_BF16;
        c.ws = ws; c.ws_bytes = (long long)256 << 20;
      } else {
        g.A = a; g.W = w; g.bias = bias; g.res = r; g.M = M; g.N = sh.N; g.K = K; g.lda = K; g.ldw = K; g.ldo = n_out; g.ldr = n_out;
        g.batch = 1; g.epi = sh.epi; g.dtype = IDF_BF16; g.ws = ws; g.ws_bytes = (long long)256 << 20;
        if (sh.epi & IDF_EPI_LN_ROW) { g.ln_stats = stats; g.ln_c = lnc; }
      }
      auto run = [&](unsigned short* out) { if (sh.conv) { c.out = out; return idf_conv3x3(&c, nullptr); } g.out = out; return idf_gemm(&g, nullptr); };
      std::vector<double> t[2];
      long long ring_launches = 0;
      int rc0 = 0;
      for (int rd = 0; rd < rounds && !rc0; ++rd)
        for (int v = 0; v < 2; ++v) {
          idf_set_tuning(IDF_TUNE_GEMM_RING, v ? (1 << 30) : 0);
          unsigned short* out = v ? o1 : o0;
          if (rd == 0) hipMemsetAsync(out, 0xff, (size_t)M * n_out * 2, 0);
          const long long before = idf_get_stat(IDF_STAT_GEMM_RING_LAUNCHES);
          int rc = run(out);                                 // warm
          if (rc) { printf("%-30s rc %d (ring %d)\n", sh.name, rc, v); rc0 = rc; break; }
          if (v) ring_launches = idf_get_stat(IDF_STAT_GEMM_RING_LAUNCHES) - before;
          hipEventRecord(e0, 0);
          for (int i = 0; i < reps; ++i) run(out);
          hipEventRecord(e1, 0);
          if (hipDeviceSynchronize() != hipSuccess) { printf("%-30s device error: %s\n", sh.name, hipGetErrorString(hipGetLastError())); return 1; }
          float ms = 0; hipEventElapsedTime(&ms, e0, e1);
          t[v].push_back(ms * 1e3 / reps);
        }
      if (rc0) continue;
      hipMemset(mism, 0, 8); hipMemset(maxabs, 0, 4); hipMemset(maxref, 0, 4);
      hipLaunchKernelGGL(diff_kernel, dim3(256), dim3(256), 0, 0, o1, o0, (size_t)M * n_out, mism, maxabs, maxref);
      unsigned long long nm_ = 0; float ma = 0, mr = 0;
      hipMemcpy(&nm_, mism, 8, hipMemcpyDeviceToHost); hipMemcpy(&ma, maxabs, 4, hipMemcpyDeviceToHost); hipMemcpy(&mr, maxref, 4, hipMemcpyDeviceToHost);
      std::sort(t[0].begin(), t[0].end()); std::sort(t[1].begin(), t[1].end());
      const double u0 = t[0][t[0].size() / 2], u1 = t[1][t[1].size() / 2];
      const bool t128 = (sh.epi & IDF_EPI_GEGLU) || sh.N % 128 == 0 || sh.N > 1024 || (sh.conv && sh.N > 128);
      const long tiles = (long)((sh.N + (t128 ? 127 : 63)) / (t128 ? 128 : 64)) * ((M + 127) / 128);
      printf("%-30s x%-3d M%-6d N%-5d K%-5d tiles %5ld  base %7.1f us  ring %7.1f us  %+6.1f %%  ring launches %lld  differing %llu of %zu, max |d| %.3g (max |ref| %.3g)\n",
             sh.name, sh.count, M, sh.N, K, tiles, u0, u1, (u0 / u1 - 1.0) * 100.0, ring_launches, nm_, (size_t)M * n_out, ma, mr);
      fflush(stdout);
      tot0 += sh.count * u0; tot1 += sh.count * u1;
      for (int i = 0; i < 5; ++i) tot_thr[i] += sh.count * (tiles <= thresholds[i] ? u1 : u0);
    }
    {   // GroupNorm of the same forward (two launches each; round-3 profile at 2 rows: 19-22 us per call)
      const int gn[][3] = {{4096, 320, 13}, {4096, 640, 2}, {4096, 960, 1}, {1024, 640, 11}, {1024, 1280, 1}, {1024, 1920, 1}, {1024, 320, 1},
                           {256, 1280, 11}, {256, 2560, 2}, {256, 640, 1}, {256, 1920, 1}, {64, 1280, 12}, {64, 2560, 3}};
      const size_t gws_bytes = (size_t)idf_groupnorm_ws_floats(R, 4096) * 4 + (1 << 20);
      float* gws; hipMalloc(&gws, gws_bytes); hipMemset(gws, 0, gws_bytes);      // zero-filled: idf_groupnorm's contract
      double tot = 0;
      for (auto& s3 : gn) {
        std::vector<double> tt;
        for (int rd = 0; rd < rounds; ++rd) {
          int rc = idf_groupnorm(a, o0, bias, lnc, gws, R, s3[0], s3[1], 1e-5f, 1, IDF_BF16, nullptr);
          if (rc) { printf("groupnorm rc %d\n", rc); break; }
          hipEventRecord(e0, 0);
          for (int i = 0; i < reps; ++i) idf_groupnorm(a, o0, bias, lnc, gws, R, s3[0], s3[1], 1e-5f, 1, IDF_BF16, nullptr);
          hipEventRecord(e1, 0);
          hipDeviceSynchronize();
          float ms = 0; hipEventElapsedTime(&ms, e0, e1);
          tt.push_back(ms * 1e3 / reps);
        }
        if (tt.empty()) continue;
        std::sort(tt.begin(), tt.end());
        printf("groupnorm (%d, %d, %d) x%-2d %7.1f us\n", R, s3[0], s3[1], s3[2], tt[tt.size() / 2]);
        tot += s3[2] * tt[tt.size() / 2];
      }
      printf("  forward-weighted GroupNorm time at %d rows: %.2f ms\n", R, tot / 1e3);
      hipFree(gws);
    }
    printf("  forward-weighted GEMM + conv time at %d rows: base %.2f ms, ring everywhere %.2f ms;", R, tot0 / 1e3, tot1 / 1e3);
    for (int i = 0; i < 5; ++i) printf("  thr %d: %.2f", thresholds[i], tot_thr[i] / 1e3);
    printf(" ms\n");
  }
  idf_set_tuning(IDF_TUNE_GEMM_RING, 0);
  return 0;
}
