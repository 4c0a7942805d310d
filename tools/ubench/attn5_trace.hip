// Per-segment cycle trace of the ping-pong attention kernel (attention5.hip built with -DIDF_ATTN5_TRACE) on the headline
// gated self-attention shape.  Build (from the repo root):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -DIDF_ATTN5_TRACE -Iinstancediffusion_amd/csrc \
//         tools/ubench/attn5_trace.hip -o tools/ubench/attn5_trace
// Segments (s_memtime cycles summed over the tiles of one block, waves 0 and 4): 1 DMA issue, 2 P.V (16 MFMA), 3 K.Q^T
// (12 MFMA), 4 counted vmcnt wait, 5 barrier after M, 6 S phase (exp / pack / guard / fragment reads), 7 barrier after S.
#include "archive/attention5.hip"
#include <cstdio>
#include <cstring>
#include <vector>
int g_mode = 9;
int idf_attn2_mode() { return g_mode; }
int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 16, N = 4096, C = 320, H = 8, n1 = 184;
  const int vt_batched = argc > 2 ? atoi(argv[2]) : 0;     // 0: batch-interleaved [C][B][N] image (engine default at N >= 1024), 1: [B][C][N]
  unsigned short *qk, *vt, *k1, *vt1, *out;
  const size_t nqk = (size_t)B * N * 2 * C, nvt = (size_t)C * B * N, nk1 = (size_t)B * n1 * C, nv1 = (size_t)B * C * 192;
  hipMalloc(&qk, nqk * 2); hipMalloc(&vt, nvt * 2); hipMalloc(&k1, nk1 * 2); hipMalloc(&vt1, nv1 * 2); hipMalloc(&out, (size_t)B * N * C * 2);
  std::vector<unsigned short> h(nqk);
  unsigned x = 12345u;
  auto rnd = [&]() { x = x * 1664525u + 1013904223u; const float f = ((x >> 8) & 0xffff) / 65536.0f - 0.5f;
                     union { float f; unsigned u; } cv; cv.f = f; return (unsigned short)(cv.u >> 16); };
  for (auto& v : h) v = rnd();
  hipMemcpy(qk, h.data(), nqk * 2, hipMemcpyHostToDevice);
  hipMemcpy(vt, h.data(), nvt * 2, hipMemcpyHostToDevice);
  hipMemcpy(k1, h.data(), nk1 * 2, hipMemcpyHostToDevice);
  hipMemcpy(vt1, h.data(), nv1 * 2, hipMemcpyHostToDevice);
  AttnParams p{};
  p.q = qk; p.ldq = 2 * C; p.sQ = (long long)N * 2 * C; p.nq = N;
  p.k[0] = qk + C; p.ldk[0] = 2 * C; p.sK[0] = (long long)N * 2 * C;
  p.vt[0] = vt; p.ldv[0] = vt_batched ? N : B * N; p.sV[0] = vt_batched ? (long long)C * N : N; p.n[0] = N;
  printf("V^T layout: %s\n", vt_batched ? "[B][C][N]" : "[C][B][N] (batch-interleaved)");
  p.k[1] = k1; p.ldk[1] = C; p.sK[1] = (long long)n1 * C; p.vt[1] = vt1; p.ldv[1] = 192; p.sV[1] = (long long)C * 192; p.n[1] = n1;
  p.out = out; p.ldo = C; p.sO = (long long)N * C; p.H = H; p.d = 40; p.scale_log2 = 0.15811388f * 1.44269504f;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode : {9, 10, 11, 12, 13, 14}) {
    g_mode = mode;
    idf_launch_attn5(p, B, IDF_BF16, 0);
    hipEventRecord(e0, 0);
    for (int i = 0; i < 5; ++i) idf_launch_attn5(p, B, IDF_BF16, 0);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double flops = 4.0 * B * N * (N + n1) * C;
    printf("mode %d: %.1f us per launch, %.1f TFLOP/s\n", mode, ms * 200.0, flops / (ms / 5 * 1e-3) / 1e12);
#ifdef IDF_ATTN5_TRACE
    unsigned long long tr[2][16];
    hipMemcpyFromSymbol(tr, HIP_SYMBOL(idf_attn5_trace_buf), sizeof(tr));
    const int T = 64 + 3;
    for (int g = 0; g < 2; ++g) {
      printf("  wave %d cycles per tile:", g * 4);
      const char* names[8] = {"gap", "dma", "kf+pv", "qk", "vmwait", "barM", "S", "barS"};
      double tot = 0;
      for (int i = 0; i < 8; ++i) { printf(" %s %.0f", names[i], (double)tr[g][i] / T); tot += (double)tr[g][i] / T; }
      printf(" | total %.0f\n", tot);
    }
#endif
  }
  return 0;
}
