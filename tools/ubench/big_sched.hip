// A/B harness for the round-4 schedule settings of the persistent GEMM / conv kernel (gemm_big.hip): tile walk (strided /
// chunked), de-phased workgroup start, counted vmcnt behind the epilogue.  Torch-free: builds the kernel file into this
// program, launches the shapes of the 128-row UNet forward through idf_launch_big with every setting INTERLEAVED per shape
// (one process, one box: same clock history), prints the median time per setting and whether its output checksum equals the
// baseline's (every setting must be bit-identical).
// Build (from the repo root):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -Iinclude -Iinstancediffusion_amd/csrc \
//         tools/ubench/big_sched.hip -o tools/ubench/big_sched
// Run: tools/ubench/big_sched [reps] [rounds]
#include "../../instancediffusion_amd/csrc/gemm_big.hip"
#include <algorithm>
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
struct Cfg { const char* name; int walk, dephase, epivm; };

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 5;
  const int rounds = argc > 2 ? atoi(argv[2]) : 3;
  const size_t max_elems = (size_t)524288 * 1280;
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
    std::vector<float> hb(16384);
    for (auto& v : hb) { x = x * 1664525u + 1013904223u; v = (((x >> 8) & 0xffff) / 65536.0f - 0.5f) * 0.5f; }
    hipMemcpy(bias, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
  }
  const int BR = IDF_EPI_BIAS | IDF_EPI_RES;
  const int GLU = IDF_EPI_BIAS | IDF_EPI_GEGLU | IDF_EPI_GEGLU_P32;
  const Shape shapes[] = {
      {"geglu 64^2", 524288, 2560, 320, GLU, false, 0, 0, 0},
      {"geglu 32^2", 131072, 5120, 640, GLU, false, 0, 0, 0},
      {"geglu 16^2", 32768, 10240, 1280, GLU, false, 0, 0, 0},
      {"proj 64^2 bias+res", 524288, 320, 320, BR, false, 0, 0, 0},
      {"q|k|v-sized 64^2 bias", 524288, 960, 320, IDF_EPI_BIAS, false, 0, 0, 0},
      {"proj 32^2 bias+res", 131072, 640, 640, BR, false, 0, 0, 0},
      {"q|k|v-sized 32^2 bias", 131072, 1920, 640, IDF_EPI_BIAS, false, 0, 0, 0},
      {"proj 16^2 bias+res", 32768, 1280, 1280, BR, false, 0, 0, 0},
      {"ff-out 64^2 bias+res", 524288, 320, 1280, BR, false, 0, 0, 0},
      {"ff-out 32^2 bias+res", 131072, 640, 2560, BR, false, 0, 0, 0},
      {"ff-out 16^2 bias+res", 32768, 1280, 5120, BR, false, 0, 0, 0},
      {"square 8k", 8192, 8192, 8192, 0, false, 0, 0, 0},
      {"conv 64^2 320 bias", 0, 320, 0, IDF_EPI_BIAS, true, 128, 64, 320},
      {"conv 32^2 640 bias", 0, 640, 0, IDF_EPI_BIAS, true, 128, 32, 640},
      {"conv 16^2 1280 bias", 0, 1280, 0, IDF_EPI_BIAS, true, 128, 16, 1280},
      {"conv 64^2 640->320 +res", 0, 320, 0, IDF_EPI_BIAS | IDF_EPI_RES, true, 128, 64, 640},
      // correctness corners of the staged epilogue: a last m-tile with 156 valid rows, the 256-wide tiles (plain, period-64
      // and period-32 GEGLU), a ragged conv
      {"proj ragged M bias+res", 262044, 320, 320, BR, false, 0, 0, 0},
      {"geglu ragged M", 131000, 2560, 320, GLU, false, 0, 0, 0},
      {"256-wide bias+res", 65536, 1024, 1024, BR, false, 0, 0, 0},
      {"geglu period-64 256-wide", 131072, 5120, 640, IDF_EPI_BIAS | IDF_EPI_GEGLU, false, 0, 0, 0},
      {"geglu period-32 256-wide", 65536, 2048, 640, GLU, false, 0, 0, 0},
      {"conv 24^2 640 ragged +res", 0, 640, 0, IDF_EPI_BIAS | IDF_EPI_RES, true, 33, 24, 640},
  };
#ifdef BIG_SCHED_SHORT
  const Cfg cfgs[] = {{"base", 0, 0, 0}, {"epivm", 0, 0, 1}};
#else
  const Cfg cfgs[] = {
      {"base", 0, 0, 0},      {"walk", 1, 0, 0},          {"epivm", 0, 0, 1},          {"walk+epivm", 1, 0, 1},
      {"deph2", 0, 2, 0},     {"deph4", 0, 4, 0},         {"deph8", 0, 8, 0},          {"walk+epivm+deph2", 1, 2, 1},
      {"walk+epivm+deph4", 1, 4, 1}, {"walk+epivm+deph8", 1, 8, 1},
  };
#endif
  constexpr int NC = sizeof(cfgs) / sizeof(cfgs[0]);
  BigSched& sc = big_sched();
  sc.dephase_min_rounds = 1;                 // (sc.walk: 0 / 1 forced per setting below; the library default is 2 = automatic)
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
    const int n_out = (sh.epi & IDF_EPI_GEGLU) ? sh.N / 2 : sh.N;
    p.N = sh.N; p.n_valid = sh.N; p.W = w; p.A = a; p.out = o; p.ldo = n_out; p.res = r; p.ldr = n_out; p.bias = bias; p.epi = sh.epi;
    p.ws = ws; p.ws_bytes = (size_t)256 << 20;
    std::vector<double> t[NC];
    unsigned long long cs[NC];
    bool same[NC];
    int rc0 = 0;
    for (int rd = 0; rd < rounds && !rc0; ++rd)
      for (int c = 0; c < NC; ++c) {
        sc.walk = cfgs[c].walk; sc.dephase = cfgs[c].dephase; sc.epi_vmcnt = cfgs[c].epivm;
        hipMemsetAsync(o, 0xff, (size_t)p.M * n_out * 2, 0);
        int rc = idf_launch_big(p, IDF_BF16, sh.conv, true, 0, nullptr);
        if (rc) { printf("%-26s launch rc %d\n", sh.name, rc); rc0 = rc; break; }
        hipMemsetAsync(csum, 0, 8, 0);
        hipLaunchKernelGGL(checksum_kernel, dim3(1024), dim3(256), 0, 0, reinterpret_cast<const unsigned*>(o), (size_t)p.M * n_out / 2, csum);
        unsigned long long v = 0; hipMemcpy(&v, csum, 8, hipMemcpyDeviceToHost);
        if (rd == 0) { cs[c] = v; same[c] = true; } else if (v != cs[c]) same[c] = false;
        hipEventRecord(e0, 0);
        for (int i = 0; i < reps; ++i) idf_launch_big(p, IDF_BF16, sh.conv, true, 0, nullptr);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        t[c].push_back(ms * 1e3 / reps);
      }
    if (rc0) continue;
    const int bn = (sh.N % 320 == 0) ? 320 : 256;
    const long tiles = (long)(sh.N / bn) * ((p.M + 255) / 256);
    printf("%-26s M%-7d N%-5d K%-5d tiles %5ld (%.1f rounds)\n", sh.name, p.M, p.N, p.K, tiles, tiles / 256.0);
    double base = 0;
    for (int c = 0; c < NC; ++c) {
      std::sort(t[c].begin(), t[c].end());
      const double us = t[c][t[c].size() / 2];
      if (c == 0) base = us;
      const double tf = 2.0 * p.M * (double)p.N * p.K / (us * 1e-6) / 1e12;
      printf("    %-20s %8.1f us  %7.1f TF  %+6.1f %%   checksum %016llx %s%s\n", cfgs[c].name, us, tf, (base / us - 1.0) * 100.0, cs[c],
             cs[c] == cs[0] ? "== base" : "!= BASE", same[c] ? "" : "  UNSTABLE");
    }
    fflush(stdout);
  }
  return 0;
}
