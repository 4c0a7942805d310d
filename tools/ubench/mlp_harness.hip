// Fused GEGLU feed-forward (idf_mlp_geglu, mlp_fused.hip) through the C ABI, torch-free:
//   * correctness: rows of three 128-row tiles against an fp64 host restatement of LN-fold -> GEGLU (exact erf) -> Linear ->
//     gated residual on the same 16-bit operands (rel-RMS and max-abs / max printed), for bf16 and fp16;
//   * the same rows of the two-idf_gemm path it replaces (so the two are held to the same reference);
//   * timing of both at the 64-row and 128-row forward widths (M = 262144 / 524288), interleaved.
// Build (from the repo root):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude tools/ubench/mlp_harness.hip -Linstancediffusion_amd -l:libidf_gfx950.so \
//         -Wl,-rpath,'$ORIGIN/../../instancediffusion_amd' -o tools/ubench/mlp_harness
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
#include "idf.h"

static unsigned short f2h(float f, int dt) {            // round-to-nearest-even to bf16 / fp16
  if (dt == IDF_BF16) { unsigned u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (unsigned short)(u >> 16); }
  _Float16 h = (_Float16)f; unsigned short s; memcpy(&s, &h, 2); return s;
}
static float h2f(unsigned short s, int dt) {
  if (dt == IDF_BF16) { unsigned u = (unsigned)s << 16; float f; memcpy(&f, &u, 4); return f; }
  _Float16 h; memcpy(&h, &s, 2); return (float)h;
}
static const int PERM[16] = {0, 1, 2, 3, 8, 9, 10, 11, 4, 5, 6, 7, 12, 13, 14, 15};

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 5;
  const int C = 320, H = 1280, N1 = 2560;
  const int Mmax = 524288;
  unsigned rng = 2024u;
  auto uni = [&]() { rng = rng * 1664525u + 1013904223u; return ((rng >> 8) & 0xffff) / 65536.0f - 0.5f; };
  for (int dt : {IDF_BF16, IDF_F16}) {
    // ---- operands
    std::vector<unsigned short> hx((size_t)Mmax * C), hw1((size_t)N1 * C), hw2((size_t)C * H), hw2p((size_t)C * H);
    std::vector<float> hst((size_t)Mmax * 2), hc(N1), hd(N1), hcd(N1 * 2), hb2(C);
    {
      std::vector<float> row(C);
      // a 1M-element random block tiled over x (rows differ inside a tile: 3276 rows per period)
      std::vector<unsigned short> blk((size_t)1 << 20);
      for (auto& v : blk) v = f2h(uni() * 3.0f + 0.3f, dt);
      for (size_t i = 0; i < hx.size(); ++i) hx[i] = blk[i & ((1u << 20) - 1)];
      for (int m = 0; m < Mmax; ++m) {
        double s = 0, q = 0;
        for (int k = 0; k < C; ++k) { const double v = h2f(hx[(size_t)m * C + k], dt); s += v; q += v * v; }
        const double mu = s / C, var = q / C - mu * mu;
        hst[2 * (size_t)m] = (float)mu; hst[2 * (size_t)m + 1] = (float)(1.0 / std::sqrt(var + 1e-5));
      }
      for (auto& v : hw1) v = f2h(uni() * 0.12f, dt);
      for (auto& v : hw2) v = f2h(uni() * 0.08f, dt);
      for (int n = 0; n < N1; ++n) {
        double s = 0;
        for (int k = 0; k < C; ++k) s += h2f(hw1[(size_t)n * C + k], dt);
        hc[n] = (float)s; hd[n] = uni() * 0.5f;
      }
      for (int j = 0; j < N1 / 64; ++j)
        for (int i = 0; i < 64; ++i) { hcd[j * 128 + i] = hc[j * 64 + i]; hcd[j * 128 + 64 + i] = hd[j * 64 + i]; }
      for (int n = 0; n < C; ++n) {
        hb2[n] = uni() * 0.5f;
        for (int g = 0; g < H / 16; ++g)
          for (int pp = 0; pp < 16; ++pp) hw2p[(size_t)n * H + 16 * g + pp] = hw2[(size_t)n * H + 16 * g + PERM[pp]];
      }
    }
    unsigned short *dx, *dw1, *dw2, *dw2p, *dout, *dmid, *dout2;
    float *dst, *dc, *dd, *dcd, *db2, *dgate;
    hipMalloc(&dx, hx.size() * 2); hipMalloc(&dout, hx.size() * 2); hipMalloc(&dout2, hx.size() * 2); hipMalloc(&dmid, (size_t)Mmax * H * 2);
    hipMalloc(&dw1, hw1.size() * 2); hipMalloc(&dw2, hw2.size() * 2); hipMalloc(&dw2p, hw2p.size() * 2);
    hipMalloc(&dst, hst.size() * 4); hipMalloc(&dc, N1 * 4); hipMalloc(&dd, N1 * 4); hipMalloc(&dcd, N1 * 8); hipMalloc(&db2, C * 4); hipMalloc(&dgate, 4);
    hipMemcpy(dx, hx.data(), hx.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dw1, hw1.data(), hw1.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dw2, hw2.data(), hw2.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dw2p, hw2p.data(), hw2p.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dst, hst.data(), hst.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dc, hc.data(), N1 * 4, hipMemcpyHostToDevice); hipMemcpy(dd, hd.data(), N1 * 4, hipMemcpyHostToDevice);
    hipMemcpy(dcd, hcd.data(), N1 * 8, hipMemcpyHostToDevice); hipMemcpy(db2, hb2.data(), C * 4, hipMemcpyHostToDevice);
    const float gate = 0.7f;
    hipMemcpy(dgate, &gate, 4, hipMemcpyHostToDevice);

    auto fused = [&](int M, unsigned short* out) {
      idf_mlp_args a{};
      a.x = dx; a.ldx = C; a.ln_stats = dst; a.w1 = dw1; a.ldw1 = C; a.cd = dcd; a.w2p = dw2p; a.ldw2 = H; a.b2 = db2; a.gate = dgate;
      a.out = out; a.ldo = C; a.M = M; a.C = C; a.dtype = dt;
      return idf_mlp_geglu(&a, nullptr);
    };
    auto two_gemms = [&](int M, unsigned short* out) {
      idf_gemm_args g{};
      g.A = dx; g.W = dw1; g.out = dmid; g.bias = dd; g.M = M; g.N = N1; g.K = C; g.lda = C; g.ldw = C; g.ldo = H; g.batch = 1;
      g.rows_per_batch = M; g.epi = IDF_EPI_BIAS | IDF_EPI_GEGLU | IDF_EPI_GEGLU_P32 | IDF_EPI_LN_ROW; g.dtype = dt; g.ln_stats = dst; g.ln_c = dc;
      int rc = idf_gemm(&g, nullptr);
      if (rc) return rc;
      idf_gemm_args h{};
      h.A = dmid; h.W = dw2; h.out = out; h.bias = db2; h.res = dx; h.gate = dgate; h.M = M; h.N = C; h.K = H; h.lda = H; h.ldw = H; h.ldo = C;
      h.ldr = C; h.batch = 1; h.rows_per_batch = M; h.epi = IDF_EPI_BIAS | IDF_EPI_RES | IDF_EPI_GATE; h.dtype = dt;
      return idf_gemm(&h, nullptr);
    };

    // ---- correctness at M = 262144: tiles 0, 1000 (strided walk: second round of some workgroup) and the last one
    const int M = 262144;
    hipMemset(dout, 0xff, (size_t)M * C * 2); hipMemset(dout2, 0xff, (size_t)M * C * 2);
    int rc1 = fused(M, dout), rc2 = two_gemms(M, dout2);
    hipError_t e = hipDeviceSynchronize();
    printf("[%s] idf_mlp_geglu rc %d, two idf_gemm rc %d, sync %s\n", dt == IDF_BF16 ? "bf16" : "fp16", rc1, rc2, hipGetErrorString(e));
    if (rc1 || rc2 || e != hipSuccess) return 1;
    std::vector<unsigned short> ho((size_t)M * C), ho2((size_t)M * C);
    hipMemcpy(ho.data(), dout, ho.size() * 2, hipMemcpyDeviceToHost); hipMemcpy(ho2.data(), dout2, ho2.size() * 2, hipMemcpyDeviceToHost);
    const int tiles_chk[3] = {0, 1000, M / 128 - 1};
    double se1 = 0, se2 = 0, sr = 0, mx1 = 0, mx2 = 0, mxr = 0, se12 = 0;
    std::vector<double> pre(N1), hh(H);
    for (int t : tiles_chk)
      for (int r = 0; r < 128; r += 1) {
        const int m = t * 128 + r;
        const double mu = hst[2 * (size_t)m], rstd = hst[2 * (size_t)m + 1];
        for (int n = 0; n < N1; ++n) {
          double s = 0;
          for (int k = 0; k < C; ++k) s += (double)h2f(hx[(size_t)m * C + k], dt) * h2f(hw1[(size_t)n * C + k], dt);
          pre[n] = rstd * (s - mu * hc[n]) + hd[n];
        }
        for (int h = 0; h < H; ++h) {
          const int g = h / 16, i = h % 16;
          const double val = pre[32 * g + i], gat = pre[32 * g + 16 + i];
          hh[h] = h2f(f2h((float)(val * 0.5 * gat * (1.0 + std::erf(gat / std::sqrt(2.0)))), dt), dt);
        }
        for (int n = 0; n < C; ++n) {
          double s = hb2[n];
          for (int h = 0; h < H; ++h) s += hh[h] * h2f(hw2[(size_t)n * H + h], dt);
          const double want = h2f(hx[(size_t)m * C + n], dt) + gate * s;
          const double g1 = h2f(ho[(size_t)m * C + n], dt), g2 = h2f(ho2[(size_t)m * C + n], dt);
          se1 += (g1 - want) * (g1 - want); se2 += (g2 - want) * (g2 - want); sr += want * want; se12 += (g1 - g2) * (g1 - g2);
          mx1 = std::max(mx1, std::fabs(g1 - want)); mx2 = std::max(mx2, std::fabs(g2 - want)); mxr = std::max(mxr, std::fabs(want));
        }
      }
    printf("    fused    vs fp64 reference (384 rows): rel-rms %.3e  max-abs/max %.3e\n", std::sqrt(se1 / sr), mx1 / mxr);
    printf("    two-gemm vs fp64 reference (384 rows): rel-rms %.3e  max-abs/max %.3e\n", std::sqrt(se2 / sr), mx2 / mxr);
    printf("    fused vs two-gemm: rel-rms %.3e\n", std::sqrt(se12 / sr));
    {   // every row written, and rows outside the checked tiles agree with the two-GEMM path to rounding
      double s12 = 0, s2 = 0; size_t bad = 0;
      for (size_t i = 0; i < ho.size(); i += 7) {
        const double a = h2f(ho[i], dt), b = h2f(ho2[i], dt);
        if (!(std::fabs(a) < 1e30)) { ++bad; continue; }
        s12 += (a - b) * (a - b); s2 += b * b;
      }
      printf("    whole output (every 7th element): fused vs two-gemm rel-rms %.3e, non-finite %zu\n", std::sqrt(s12 / s2), bad);
      unsigned long long fnv = 1469598103934665603ull;            // checksum of the WHOLE fused output: schedule variants must agree bit for bit
      for (size_t i = 0; i < ho.size(); ++i) { fnv ^= ho[i]; fnv *= 1099511628211ull; }
      printf("    fused output checksum %016llx\n", fnv);
    }
    // ---- timing
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    // (8192 ... 65536 = 2 ... 16-row forwards: where the fused kernel's 128-row tiles stop filling the chip -- the engine's
    // row threshold comes from these lines; the two-GEMM side follows the library's IDF_GEMM_RING default)
    for (int Mt : {8192, 16384, 32768, 49152, 65536, 262144, 524288}) {
      std::vector<double> tf, tg;
      for (int rd = 0; rd < 3; ++rd)
        for (int which = 0; which < 2; ++which) {
          if (which == 0) fused(Mt, dout); else two_gemms(Mt, dout2);
          hipEventRecord(e0, 0);
          for (int i = 0; i < reps; ++i) { if (which == 0) fused(Mt, dout); else two_gemms(Mt, dout2); }
          hipEventRecord(e1, 0);
          hipDeviceSynchronize();
          float ms = 0; hipEventElapsedTime(&ms, e0, e1);
          (which == 0 ? tf : tg).push_back(ms * 1e3 / reps);
        }
      std::sort(tf.begin(), tf.end()); std::sort(tg.begin(), tg.end());
      const double flop = 2.0 * Mt * ((double)N1 * C + (double)C * H);
      printf("    M %6d: fused %8.1f us (%6.1f TF)   two gemms %8.1f us (%6.1f TF)   %+5.1f %%\n", Mt, tf[1], flop / tf[1] * 1e-6, tg[1],
             flop / tg[1] * 1e-6, (tg[1] / tf[1] - 1.0) * 100.0);
      // a -DIDF_MLPW_TRACE build (tools/build_mlpw_variant.sh <name> MW_TRACE=1 -- -DIDF_MLPW_TRACE): the stream kernel's trace
      if (auto rdw = (int (*)(unsigned long long*))dlsym(RTLD_DEFAULT, "idf_mlpw_trace_read")) {
        fused(Mt, dout); hipDeviceSynchronize();
        unsigned long long tr[4][12];
        if (rdw(&tr[0][0]) == 0 && Mt >= 262144) {
          for (int w = 0; w < 4; ++w) {
            const double nb = (double)tr[w][6];
            if (nb == 0) continue;
            const double nt = nb / 38.0;
            double per = 0; for (int i = 0; i < 6; ++i) per += tr[w][i] / nb;
            printf("      wgM wave%d per steady iteration: top %5.0f pre %5.0f dma-gaps %5.0f g2-rest %5.0f g1a %5.0f g1b+trail %5.0f = %5.0f cycles (1920 = pipe full)"
                   " | per tile: steady %7.0f load %6.0f pro/01/10/drain %6.0f epilogue %6.0f\n", w, tr[w][0] / nb, tr[w][1] / nb, tr[w][2] / nb,
                   tr[w][3] / nb, tr[w][4] / nb, tr[w][5] / nb, per, tr[w][10] / nt, tr[w][7] / nt, tr[w][9] / nt, tr[w][8] / nt);
          }
        }
      }
      // a -DIDF_MLP_TRACE build of the library exports the cycle trace of the last fused launch
      if (auto rd = (int (*)(unsigned long long*))dlsym(RTLD_DEFAULT, "idf_mlp_trace_read")) {
        fused(Mt, dout); hipDeviceSynchronize();
        unsigned long long tr[4][10];
        if (rd(&tr[0][0]) == 0) {
          const char* who[4] = {"wg0 wave0", "wg0 wave4", "wgM wave0", "wgM wave4"};
          for (int w = 0; w < 4; ++w) {
            const double nc = (double)tr[w][9];
            if (nc == 0) continue;
            double per = 0; for (int i = 0; i < 7; ++i) per += tr[w][i] / nc;
            printf("      %s per chunk: vmwait %5.0f barrier %5.0f fill %5.0f gemm1 %5.0f geglu %5.0f xbarrier %5.0f gemm2 %5.0f = %5.0f cycles "
                   "(1920 = pipe full) | per tile: epilogue %6.0f load %6.0f\n", who[w], tr[w][0] / nc, tr[w][1] / nc, tr[w][2] / nc, tr[w][3] / nc,
                   tr[w][4] / nc, tr[w][5] / nc, tr[w][6] / nc, per, tr[w][7] / (nc / 40), tr[w][8] / (nc / 40));
          }
        }
      }
    }
    hipFree(dx); hipFree(dout); hipFree(dout2); hipFree(dmid); hipFree(dw1); hipFree(dw2); hipFree(dw2p);
    hipFree(dst); hipFree(dc); hipFree(dd); hipFree(dcd); hipFree(db2); hipFree(dgate);
  }
  return 0;
}
