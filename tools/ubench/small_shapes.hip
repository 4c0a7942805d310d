// A/B harness for the dispatch of the small and mid-size GEMM / conv launches, through the C ABI, torch-free: every dense GEMM
// and 3x3 conv shape of an R-row UNet forward (R = 2: BASELINE config 2; 16: the second MIS phase of the reference's own
// 8-image batch; 64 / 128: the bench's forwards) under four dispatch settings, interleaved per shape in one process:
// round-3 dispatch / + latency kernel for grids of <= 256 tiles / persistent kernel forced / latency kernel + 50 % occupancy
// bar (the round-4 default).  Every launch of a shape reads ANOTHER weight matrix (cold weights, as in a forward) and the time
// is the GPU time of a captured graph of `reps` launches.  Prints the median per setting, then the forward-weighted totals;
// the GroupNorms of the same forward are timed at the end of each block.  (Bitwise agreement of the kernels is the GPU suite's
// business: tests/test_kernels_gpu.py::test_gemm_ring_*.)
// Build (from the repo root):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude tools/ubench/small_shapes.hip -Linstancediffusion_amd -l:libidf_gfx950.so \
//         -Wl,-rpath,'$ORIGIN/../../instancediffusion_amd' -o tools/ubench/small_shapes
// Run: tools/ubench/small_shapes [reps] [rounds] [largest R: 16 | 64 | 128]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "idf.h"

struct Shape { const char* name; int count; bool conv; int M, N, K, epi; int H, Cin, stride, up; };

// GPU time per call of `fn(stream)`: `reps` calls captured into ONE hipGraph (what the engine replays), launched three times,
// the third timed -- eager back-to-back launches of 10-us kernels measure the host's launch rate instead (first version of
// this harness: profiles/r04_small_family_eager_host_bound.log).
template <class F> static double graph_us(F&& fn, int reps, hipStream_t st, hipEvent_t e0, hipEvent_t e1) {
  hipGraph_t g; hipGraphExec_t ge;
  if (hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) != hipSuccess) return -1;
  for (int i = 0; i < reps; ++i) fn(st);
  if (hipStreamEndCapture(st, &g) != hipSuccess) return -1;
  if (hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) != hipSuccess) return -1;
  hipGraphLaunch(ge, st); hipGraphLaunch(ge, st);
  hipEventRecord(e0, st);
  hipGraphLaunch(ge, st);
  hipEventRecord(e1, st);
  if (hipStreamSynchronize(st) != hipSuccess) return -2;
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  hipGraphExecDestroy(ge); hipGraphDestroy(g);
  return ms * 1e3 / reps;
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 20;
  const int rounds = argc > 2 ? atoi(argv[2]) : 3;
  const int RMAX = argc > 3 ? atoi(argv[3]) : 16;
  const size_t max_elems = (size_t)RMAX * 4096 * 2560;          // largest operand / output (GEGLU output at 64^2)
  // 1 GiB of weights: every captured launch of a shape reads ANOTHER weight matrix (as the layers of a forward do -- 2.5 GB of
  // weights stream from HBM once per forward; re-launching one layer would serve them from L2 / the 256 MB Infinity Cache)
  const size_t W_ELEMS = (size_t)512 << 20;
  unsigned short *a, *w, *o0, *o1, *r;
  float *bias, *ws, *stats, *lnc;
  hipMalloc(&a, max_elems * 2); hipMalloc(&o0, max_elems * 2); hipMalloc(&o1, max_elems * 2); hipMalloc(&r, max_elems * 2);
  hipMalloc(&w, W_ELEMS * 2); hipMalloc(&bias, 32768 * 4); hipMalloc(&lnc, 32768 * 4); hipMalloc(&ws, (size_t)256 << 20);
  hipMalloc(&stats, (size_t)RMAX * 4096 * 2 * 4);
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
    for (size_t off = 0; off < W_ELEMS; off += h.size()) hipMemcpy(w + off, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    std::vector<float> hb(32768);
    for (auto& v : hb) { x = x * 1664525u + 1013904223u; v = (((x >> 8) & 0xffff) / 65536.0f - 0.5f) * 0.5f; }
    hipMemcpy(bias, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
    for (auto& v : hb) { x = x * 1664525u + 1013904223u; v = (((x >> 8) & 0xffff) / 65536.0f - 0.5f) * 0.1f; }
    hipMemcpy(lnc, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
    std::vector<float> hs((size_t)RMAX * 4096 * 2);
    for (size_t i = 0; i < hs.size(); i += 2) { x = x * 1664525u + 1013904223u; hs[i] = (((x >> 8) & 0xffff) / 65536.0f - 0.5f) * 0.02f; hs[i + 1] = 0.9f + (x & 255) / 1024.0f; }
    hipMemcpy(stats, hs.data(), hs.size() * 4, hipMemcpyHostToDevice);
  }
  const int B_ = IDF_EPI_BIAS, BR = IDF_EPI_BIAS | IDF_EPI_RES, GLU = IDF_EPI_BIAS | IDF_EPI_GEGLU | IDF_EPI_GEGLU_P32;
  const int LNB = IDF_EPI_BIAS | IDF_EPI_LN_ROW;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipStream_t st; hipStreamCreate(&st);
  const int thresholds[] = {128, 256, 512, 1024, 1 << 30};
  for (int R : {2, 16, 64, 128}) {
    if (R > RMAX) break;
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
    double tot0 = 0, tot1 = 0, tot_thr[5] = {0, 0, 0, 0, 0}, totv[4] = {0, 0, 0, 0}, totbest = 0;
    const bool gn_only = getenv("SMALL_SHAPES_GN_ONLY") != nullptr;
    for (const Shape& sh : shapes) {
      if (gn_only) break;
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
      const size_t w_elems = (((size_t)sh.N * K + 4095) / 4096) * 4096, w_slots = W_ELEMS / w_elems;
      size_t w_i = 0;
      auto run = [&](unsigned short* out, hipStream_t s_ = nullptr) {
        const unsigned short* wp = w + (w_i++ % w_slots) * w_elems;
        if (sh.conv) { c.W = wp; c.out = out; return idf_conv3x3(&c, s_); }
        g.W = wp; g.out = out; return idf_gemm(&g, s_);
      };
      // variants: 0 = round-3 dispatch (K-loop variants 1 / 2 below the persistent kernel's 80 % occupancy bar), 1 = + latency
      // kernel for grids of <= 256 tiles, 2 = persistent kernel FORCED, 3 = latency kernel + occupancy bar at 50 %
      std::vector<double> t[4];
      long long ring_launches = 0;
      int rc0 = 0;
      for (int rd = 0; rd < rounds && !rc0; ++rd)
        for (int v = 0; v < 4; ++v) {
          idf_set_tuning(IDF_TUNE_GEMM_RING, (v == 1 || v == 3) ? 256 : 0);
          idf_set_tuning(IDF_TUNE_GEMM_BIG, v == 2 ? 2 : 1);
          idf_set_tuning(IDF_TUNE_BIG_MIN_EFF, v == 3 ? 50 : 80);
          unsigned short* out = v == 1 ? o1 : o0;
          if (rd == 0) hipMemsetAsync(out, 0xff, (size_t)M * n_out * 2, 0);
          const long long before = idf_get_stat(IDF_STAT_GEMM_RING_LAUNCHES);
          w_i = 0;                                           // every variant leaves the product with the SAME weights in `out` ...
          int rc = run(out);                                 // warm
          if (rc) { printf("%-30s rc %d (variant %d)\n", sh.name, rc, v); rc0 = rc; break; }
          if (v == 1) ring_launches = idf_get_stat(IDF_STAT_GEMM_RING_LAUNCHES) - before;
          if (hipDeviceSynchronize() != hipSuccess) { printf("%-30s device error: %s\n", sh.name, hipGetErrorString(hipGetLastError())); return 1; }
          const double us = graph_us([&](hipStream_t s_) { run(out, s_); }, reps, st, e0, e1);
          if (us < 0) { printf("%-30s graph capture / replay failed (%g)\n", sh.name, us); return 1; }
          t[v].push_back(us);
        }
      idf_set_tuning(IDF_TUNE_GEMM_BIG, 1); idf_set_tuning(IDF_TUNE_BIG_MIN_EFF, 80);
      if (rc0) continue;
      double u[4];
      for (int v = 0; v < 4; ++v) { std::sort(t[v].begin(), t[v].end()); u[v] = t[v][t[v].size() / 2]; }
      const double u0 = u[0], u1 = u[1];
      const bool t128 = (sh.epi & IDF_EPI_GEGLU) || sh.N % 128 == 0 || sh.N > 1024 || (sh.conv && sh.N > 128);
      const long tiles = (long)((sh.N + (t128 ? 127 : 63)) / (t128 ? 128 : 64)) * ((M + 127) / 128);
      printf("%-30s x%-3d M%-6d N%-5d K%-5d tiles %5ld  r3 %7.1f  +ring256 %7.1f (%lld)  big-forced %7.1f  ring256+eff50 %7.1f us  %+6.1f %%\n",
             sh.name, sh.count, M, sh.N, K, tiles, u[0], u[1], ring_launches, u[2], u[3], (u[0] / u[3] - 1.0) * 100.0);
      fflush(stdout);
      for (int v = 0; v < 4; ++v) totv[v] += sh.count * u[v];
      totbest += sh.count * std::min(std::min(u[0], u[1]), std::min(u[2], u[3]));
      tot0 += sh.count * u0; tot1 += sh.count * u1;
      for (int i = 0; i < 5; ++i) tot_thr[i] += sh.count * (tiles <= thresholds[i] ? u1 : u0);
    }
    {   // GroupNorm of the same forward (two launches each: statistics, normalisation).  A one-pass form -- row chunk held in
        // registers, device-wide rendezvous per sample, one launch -- was built and measured here in round 4: correct, and 1.3x
        // (2 rows) to 2.4-3x (128 rows) SLOWER (profiles/r04_gn_onepass_*.log, profiles/DESIGN_r01_r05_full.md "Measured and NOT shipped").
      const int gn[][3] = {{4096, 320, 13}, {4096, 640, 2}, {4096, 960, 1}, {1024, 640, 11}, {1024, 1280, 1}, {1024, 1920, 1}, {1024, 320, 1},
                           {256, 1280, 11}, {256, 2560, 2}, {256, 640, 1}, {256, 1920, 1}, {64, 1280, 12}, {64, 2560, 3}};
      const size_t gws_bytes = (size_t)idf_groupnorm_ws_floats(R, 4096) * 4 + (1 << 20);
      float* gws; hipMalloc(&gws, gws_bytes);
      double tot = 0;
      for (auto& s3 : gn) {
        std::vector<double> tt;
        for (int rd = 0; rd < rounds; ++rd) {
          int rc = idf_groupnorm(a, o0, bias, lnc, gws, R, s3[0], s3[1], 1e-5f, 1, IDF_BF16, nullptr);
          if (rc) { printf("groupnorm rc %d\n", rc); break; }
          if (hipDeviceSynchronize() != hipSuccess) { printf("groupnorm device error\n"); return 1; }
          const double us = graph_us([&](hipStream_t s_) { idf_groupnorm(a, o0, bias, lnc, gws, R, s3[0], s3[1], 1e-5f, 1, IDF_BF16, s_); }, reps, st, e0, e1);
          if (us < 0) { printf("groupnorm graph capture / replay failed\n"); return 1; }
          tt.push_back(us);
        }
        if (tt.empty()) continue;
        std::sort(tt.begin(), tt.end());
        printf("groupnorm (%d, %d, %d) x%-2d %7.1f us\n", R, s3[0], s3[1], s3[2], tt[tt.size() / 2]);
        tot += s3[2] * tt[tt.size() / 2];
      }
      printf("  forward-weighted GroupNorm time at %d rows: %.2f ms\n", R, tot / 1e3);
      hipFree(gws);
    }
    printf("  forward-weighted GEMM + conv time at %d rows: round-3 dispatch %.2f ms, + latency kernel (<= 256 tiles) %.2f, persistent kernel forced %.2f, "
           "latency kernel + 50 %% bar %.2f, best of the four per shape %.2f ms\n", R, totv[0] / 1e3, totv[1] / 1e3, totv[2] / 1e3, totv[3] / 1e3, totbest / 1e3);
    (void)tot0; (void)tot1; (void)tot_thr; (void)thresholds;
  }
  idf_set_tuning(IDF_TUNE_GEMM_RING, 0);
  return 0;
}
