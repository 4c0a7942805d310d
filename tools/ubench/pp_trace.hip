// Per-segment cycle trace of the ping-pong GEMM (gemm_big.hip built with -DIDF_PP_TRACE).  Build (from the repo root):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -DIDF_PP_TRACE -Iinclude -Iinstancediffusion_amd/csrc \
//         tools/ubench/pp_trace.hip -o tools/ubench/pp_trace
// Segments (s_memtime cycles per half-tile, waves 0 (group X) and 4 (group Y) of workgroup 0):
//   0 counted vmcnt wait (X)  1 barrier before L  2 L: fragment reads (+ DL pieces) and their lgkmcnt wait
//   3 counted vmcnt wait (Y)  4 barrier before C  5 C: MFMAs + LDS-DMA pieces  6 loader advance
//   per tile: 7 tail barrier (X)  8 epilogue  9 head barrier (Y) + accumulator clear
#include "archive/gemm_big_r02.hip"
#include <cstdio>
#include <vector>
int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 8192, N = argc > 2 ? atoi(argv[2]) : 8192, K = argc > 3 ? atoi(argv[3]) : 8192;
  unsigned short *a, *w, *o;
  hipMalloc(&a, (size_t)M * K * 2); hipMalloc(&w, (size_t)N * K * 2); hipMalloc(&o, (size_t)M * N * 2);
  std::vector<unsigned short> h((size_t)(M > N ? M : N) * K);
  unsigned x = 12345u;
  for (auto& v : h) { x = x * 1664525u + 1013904223u; const float f = ((x >> 8) & 0xffff) / 65536.0f - 0.5f;
                      union { float f; unsigned u; } cv; cv.f = f; v = (unsigned short)(cv.u >> 16); }
  hipMemcpy(a, h.data(), (size_t)M * K * 2, hipMemcpyHostToDevice);
  hipMemcpy(w, h.data(), (size_t)N * K * 2, hipMemcpyHostToDevice);
  CoreParams p{};
  p.W = w; p.ldw = K; p.N = N; p.A = a; p.lda = K; p.M = M; p.K = K; p.out = o; p.ldo = N; p.epi = 0; p.n_valid = N;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int geom : {0, 3, 5, 6, 2}) {
    if (geom == 2) printf("IDF_GEMM_PP_DL=%s\n", getenv("IDF_GEMM_PP_DL") ? getenv("IDF_GEMM_PP_DL") : "0");
    idf_big_set_geom(geom);
    int rc = idf_launch_big(p, IDF_BF16, false, true, 0, nullptr);
    if (rc) { printf("launch rc %d\n", rc); return 1; }
    hipEventRecord(e0, 0);
    for (int i = 0; i < 5; ++i) idf_launch_big(p, IDF_BF16, false, true, 0, nullptr);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    printf("geom %d: %.1f us per launch, %.1f TFLOP/s\n", geom, ms * 200.0, 2.0 * M * N * K / (ms / 5 * 1e-3) / 1e12);
#ifdef IDF_PP_TRACE
    if (geom == 2) {
      unsigned long long tr[2][16];
      hipMemcpyFromSymbol(tr, HIP_SYMBOL(idf_pp_trace_buf), sizeof(tr));
      const int bn = (N % 320 == 0) ? 320 : 256;
      const long tiles = (long)(N / bn) * ((M + 255) / 256);
      const long my_tiles = (tiles + 255) / 256;
      const double halves = (double)my_tiles * (K / 32);
      const char* names[10] = {"vmwaitX", "barL", "L", "vmwaitY", "barC", "C", "advance", "tailbar", "epilogue", "head"};
      for (int g = 0; g < 2; ++g) {
        printf("  wave %d per half-tile:", g * 4);
        double tot = 0;
        for (int i = 0; i < 7; ++i) { printf(" %s %.0f", names[i], tr[g][i] / halves); tot += tr[g][i] / halves; }
        printf(" | sum %.0f ;  per tile:", tot);
        for (int i = 7; i < 10; ++i) printf(" %s %.0f", names[i], (double)tr[g][i] / my_tiles);
        printf("\n");
      }
    }
#endif
  }
  return 0;
}
