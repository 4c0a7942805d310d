// Fused q | k | v projection of the C = 320 level through the C ABI (idf_gemm with vt_out), torch-free: qkv320w_kernel
// (IDF_TUNE_QKV_ROW = 1, qkv_fused.hip) against the persistent GEMM kernel (0) on the same operands, statistics handed in.
//   * correctness of both against an fp64 host restatement on sampled rows (LayerNorm fold: rstd * (x . w - mu c) + d), bf16 / fp16;
//   * whole-output comparison of the two kernels (max |diff|, differing elements) and checksums;
//   * HIP-event timing at M = 262144 / 524288.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -Iinclude tools/ubench/qkv_harness.hip -o tools/ubench/qkv_harness -ldl
// Run:   tools/ubench/qkv_harness [lib = instancediffusion_amd/libidf_gfx950.so] [reps = 10] [Mcheck = 262144]
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
#include "idf.h"

static unsigned short f2h(float f, int dt) {
  if (dt == IDF_BF16) { unsigned u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (unsigned short)(u >> 16); }
  _Float16 h = (_Float16)f; unsigned short s; memcpy(&s, &h, 2); return s;
}
static float h2f(unsigned short s, int dt) {
  if (dt == IDF_BF16) { unsigned u = (unsigned)s << 16; float f; memcpy(&f, &u, 4); return f; }
  _Float16 h; memcpy(&h, &s, 2); return (float)h;
}

int main(int argc, char** argv) {
  setvbuf(stdout, nullptr, _IONBF, 0);
  const char* path = argc > 1 ? argv[1] : "instancediffusion_amd/libidf_gfx950.so";
  const int reps = argc > 2 ? atoi(argv[2]) : 10;
  const int Mc = argc > 3 ? atoi(argv[3]) : 262144;
  void* h = dlopen(path, RTLD_NOW | RTLD_GLOBAL);
  if (!h) { fprintf(stderr, "dlopen %s: %s\n", path, dlerror()); return 2; }
  auto gemm = (int (*)(const idf_gemm_args*, void*))dlsym(h, "idf_gemm");
  auto tune = (int (*)(int, int))dlsym(h, "idf_set_tuning");
  auto stat = (long long (*)(int))dlsym(h, "idf_get_stat");
  const int C = 320, N = 960, Mmax = 524288;
  unsigned rng = 777u;
  auto uni = [&]() { rng = rng * 1664525u + 1013904223u; return ((rng >> 8) & 0xffff) / 65536.0f - 0.5f; };
  for (int dt : {IDF_BF16, IDF_F16}) {
    std::vector<unsigned short> hx((size_t)Mmax * C), hw((size_t)N * C);
    std::vector<float> hst((size_t)Mmax * 2), hc(N), hd(N);
    {
      std::vector<unsigned short> blk((size_t)1 << 20);
      for (auto& v : blk) v = f2h(uni() * 3.0f + 0.4f, dt);
      for (size_t i = 0; i < hx.size(); ++i) hx[i] = blk[i & ((1u << 20) - 1)];
      for (int m = 0; m < Mmax; ++m) {
        double s = 0, q = 0;
        for (int k = 0; k < C; ++k) { const double v = h2f(hx[(size_t)m * C + k], dt); s += v; q += v * v; }
        const double mu = s / C, var = q / C - mu * mu;
        hst[2 * (size_t)m] = (float)mu; hst[2 * (size_t)m + 1] = (float)(1.0 / std::sqrt(var + 1e-5));
      }
      for (auto& v : hw) v = f2h(uni() * 0.12f, dt);
      for (int n = 0; n < N; ++n) {
        double s = 0;
        for (int k = 0; k < C; ++k) s += h2f(hw[(size_t)n * C + k], dt);
        hc[n] = (float)s; hd[n] = uni() * 0.5f;
      }
    }
    unsigned short *dx, *dw, *dq[2], *dv[2];
    float *dst, *dc, *dd;
    hipMalloc(&dx, hx.size() * 2); hipMalloc(&dw, hw.size() * 2); hipMalloc(&dst, hst.size() * 4); hipMalloc(&dc, N * 4); hipMalloc(&dd, N * 4);
    for (int k = 0; k < 2; ++k) { hipMalloc(&dq[k], (size_t)Mmax * 640 * 2); hipMalloc(&dv[k], (size_t)C * Mmax * 2); }
    hipMemcpy(dx, hx.data(), hx.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dst, hst.data(), hst.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dc, hc.data(), N * 4, hipMemcpyHostToDevice);
    hipMemcpy(dd, hd.data(), N * 4, hipMemcpyHostToDevice);
    auto run = [&](int M, int k) {
      idf_gemm_args a{};
      a.A = dx; a.W = dw; a.out = dq[k]; a.bias = dd; a.M = M; a.N = N; a.K = C; a.lda = C; a.ldw = C; a.ldo = 640; a.batch = 1;
      a.rows_per_batch = M; a.epi = IDF_EPI_BIAS | IDF_EPI_LN_ROW; a.dtype = dt; a.ln_stats = dst; a.ln_c = dc; a.ln_eps = 1e-5f;
      a.vt_out = dv[k]; a.ld_vt = M; a.vt_col0 = 640;
      return gemm(&a, nullptr);
    };
    printf("[%s]\n", dt == IDF_BF16 ? "bf16" : "fp16");
    std::vector<unsigned short> oq[2], ov[2];
    for (int mode = 0; mode < 2; ++mode) {
      tune(IDF_TUNE_QKV_ROW, mode);
      hipMemset(dq[mode], 0xff, (size_t)Mc * 640 * 2); hipMemset(dv[mode], 0xff, (size_t)C * Mc * 2);
      const long long s0 = stat(IDF_STAT_QKV_ROW_LAUNCHES);
      const int rc = run(Mc, mode);
      const hipError_t e = hipDeviceSynchronize();
      printf("  mode %d: rc %d, sync %s, served by qkv320w: %lld\n", mode, rc, hipGetErrorString(e), stat(IDF_STAT_QKV_ROW_LAUNCHES) - s0);
      if (rc || e != hipSuccess) return 1;
      oq[mode].resize((size_t)Mc * 640); ov[mode].resize((size_t)C * Mc);
      hipMemcpy(oq[mode].data(), dq[mode], oq[mode].size() * 2, hipMemcpyDeviceToHost);
      hipMemcpy(ov[mode].data(), dv[mode], ov[mode].size() * 2, hipMemcpyDeviceToHost);
      double se = 0, sr = 0, mx = 0; size_t bad = 0;
      for (int t : {0, 1, 777, Mc / 128 - 1})
        for (int r = 0; r < 128; r += 3) {
          const int m = t * 128 + r;
          const double mu = hst[2 * (size_t)m], rstd = hst[2 * (size_t)m + 1];
          for (int n = 0; n < N; ++n) {
            double s = 0;
            for (int k = 0; k < C; ++k) s += (double)h2f(hx[(size_t)m * C + k], dt) * h2f(hw[(size_t)n * C + k], dt);
            const double want = rstd * (s - mu * hc[n]) + hd[n];
            const double got = n < 640 ? h2f(oq[mode][(size_t)m * 640 + n], dt) : h2f(ov[mode][(size_t)(n - 640) * Mc + m], dt);
            if (!(std::fabs(got) < 1e30)) { ++bad; continue; }
            se += (got - want) * (got - want); sr += want * want; mx = std::max(mx, std::fabs(got - want));
          }
        }
      unsigned long long fnv = 1469598103934665603ull;
      for (auto v : oq[mode]) { fnv ^= v; fnv *= 1099511628211ull; }
      for (auto v : ov[mode]) { fnv ^= v; fnv *= 1099511628211ull; }
      printf("  mode %d vs fp64 (172 rows x 960): rel-rms %.3e max-abs %.3e non-finite %zu; checksum %016llx\n", mode, std::sqrt(se / sr), mx, bad, fnv);
    }
    {
      size_t nd = 0; double md = 0; size_t first = (size_t)-1;
      for (size_t i = 0; i < oq[0].size(); ++i) if (oq[0][i] != oq[1][i]) { ++nd; if (first == (size_t)-1) first = i; md = std::max(md, (double)std::fabs(h2f(oq[0][i], dt) - h2f(oq[1][i], dt))); }
      size_t nv = 0; double mv = 0; size_t firstv = (size_t)-1;
      for (size_t i = 0; i < ov[0].size(); ++i) if (ov[0][i] != ov[1][i]) { ++nv; if (firstv == (size_t)-1) firstv = i; mv = std::max(mv, (double)std::fabs(h2f(ov[0][i], dt) - h2f(ov[1][i], dt))); }
      printf("  kernel 1 vs kernel 0: q|k differing %zu of %zu (max %.3e, first at row %zu col %zu); V^T differing %zu of %zu (max %.3e, first at ch %zu tok %zu)\n",
             nd, oq[0].size(), md, first == (size_t)-1 ? 0 : first / 640, first == (size_t)-1 ? 0 : first % 640, nv, ov[0].size(), mv,
             firstv == (size_t)-1 ? 0 : firstv / Mc, firstv == (size_t)-1 ? 0 : firstv % Mc);
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int Mt : {262144, 524288}) {
      std::vector<double> t[2];
      for (int rd = 0; rd < 3; ++rd)
        for (int mode = 0; mode < 2; ++mode) {
          tune(IDF_TUNE_QKV_ROW, mode);
          run(Mt, mode);
          hipEventRecord(e0, 0);
          for (int i = 0; i < reps; ++i) run(Mt, mode);
          hipEventRecord(e1, 0);
          hipDeviceSynchronize();
          float ms = 0; hipEventElapsedTime(&ms, e0, e1);
          t[mode].push_back(ms * 1e3 / reps);
        }
      std::sort(t[0].begin(), t[0].end()); std::sort(t[1].begin(), t[1].end());
      const double flop = 2.0 * Mt * N * C, bytes = (double)Mt * (C + N) * 2;
      if (auto rd = (int (*)(unsigned long long*))dlsym(h, "idf_qkvw_trace_read")) {
        unsigned long long tr[4][16];
        tune(IDF_TUNE_QKV_ROW, 1); run(Mt, 1); hipDeviceSynchronize();
        if (rd(&tr[0][0]) == 0)
          for (int w = 0; w < 4; w += 3) {
            const double nq = (double)tr[w][4], nv = (double)tr[w][10], nt = nq / 9.0;
            if (nq == 0) continue;
            printf("    wgM wave%d  qq step: top %5.0f pre+dma-gaps %5.0f gaps..23 %5.0f gaps 24..+trail %5.0f = %5.0f | vv step: top %5.0f dma %5.0f ..23 %5.0f rest %5.0f = %5.0f (1280 = pipe full)"
                   " | per tile: head %6.0f pro %6.0f qq-steps %6.0f qv+vv+v_ %6.0f\n", w, tr[w][0] / nq, tr[w][1] / nq, tr[w][2] / nq, tr[w][3] / nq,
                   (tr[w][0] + tr[w][1] + tr[w][2] + tr[w][3]) / nq, tr[w][6] / nv, tr[w][7] / nv, tr[w][8] / nv, tr[w][9] / nv,
                   (tr[w][6] + tr[w][7] + tr[w][8] + tr[w][9]) / nv, tr[w][12] / nt, tr[w][13] / nt, tr[w][14] / nt, 0.0);
          }
      }
      printf("  M %6d: gemm_big %7.1f us (%6.1f TF, %4.2f TB/s)   qkv320w %7.1f us (%6.1f TF, %4.2f TB/s)   %+5.1f %%\n", Mt, t[0][1], flop / t[0][1] * 1e-6,
             bytes / t[0][1] * 1e-6, t[1][1], flop / t[1][1] * 1e-6, bytes / t[1][1] * 1e-6, (t[0][1] / t[1][1] - 1.0) * 100.0);
    }
    hipFree(dx); hipFree(dw); hipFree(dst); hipFree(dc); hipFree(dd);
    for (int k = 0; k < 2; ++k) { hipFree(dq[k]); hipFree(dv[k]); }
  }
  return 0;
}
