// First conv (idf_conv_in) through the C ABI, torch-free: time at the 64- and 128-row forward widths and compare a sample of
// output pixels with an fp64 host evaluation of the 3x3 conv on the same fp32 operands.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude tools/ubench/conv_in_bench.hip -Linstancediffusion_amd -l:libidf_gfx950.so \
//        -Wl,-rpath,'$ORIGIN/../../instancediffusion_amd' -o tools/ubench/conv_in_bench
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
#include "idf.h"

int main() {
  const int Cin = 4, H = 64, W = 64, Cout = 320, Bmax = 128;
  std::vector<float> hx((size_t)Bmax * Cin * H * W), hw((size_t)Cout * Cin * 9), hb(Cout);
  unsigned s = 99u;
  auto uni = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
  for (auto& v : hx) v = uni() * 4.0f;
  for (auto& v : hw) v = uni() * 0.4f;
  for (auto& v : hb) v = uni();
  float *dx, *dw, *db; unsigned short* dout;
  hipMalloc(&dx, hx.size() * 4); hipMalloc(&dw, hw.size() * 4); hipMalloc(&db, hb.size() * 4); hipMalloc(&dout, (size_t)Bmax * H * W * Cout * 2);
  hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(db, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int dt : {IDF_BF16, IDF_F16})
    for (int B : {64, 128}) {
      int rc = idf_conv_in(dx, dw, db, dout, B, Cin, H, W, Cout, dt, nullptr);
      hipDeviceSynchronize();
      hipEventRecord(e0, 0);
      for (int i = 0; i < 10; ++i) idf_conv_in(dx, dw, db, dout, B, Cin, H, W, Cout, dt, nullptr);
      hipEventRecord(e1, 0); hipDeviceSynchronize();
      float ms = 0; hipEventElapsedTime(&ms, e0, e1);
      std::vector<unsigned short> ho((size_t)B * H * W * Cout);
      hipMemcpy(ho.data(), dout, ho.size() * 2, hipMemcpyDeviceToHost);
      double se = 0, sr = 0, mx = 0;
      for (int smp = 0; smp < 4000; ++smp) {
        const int b = (smp * 7) % B, y = (smp * 13) % H, x = (smp * 29) % W, co = (smp * 31) % Cout;
        double acc = hb[co];
        for (int ci = 0; ci < Cin; ++ci)
          for (int ky = 0; ky < 3; ++ky)
            for (int kx = 0; kx < 3; ++kx) {
              const int yy = y + ky - 1, xx = x + kx - 1;
              if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
              acc += (double)hx[(((size_t)b * Cin + ci) * H + yy) * W + xx] * hw[((size_t)co * Cin + ci) * 9 + ky * 3 + kx];
            }
        const unsigned short u = ho[(((size_t)b * H + y) * W + x) * Cout + co];
        float got;
        if (dt == IDF_BF16) { unsigned v = (unsigned)u << 16; memcpy(&got, &v, 4); } else { _Float16 h; memcpy(&h, &u, 2); got = (float)h; }
        se += (got - acc) * (got - acc); sr += acc * acc; mx = std::max(mx, std::fabs(got - acc));
      }
      const double us = ms * 100.0, bytes = (double)B * H * W * Cout * 2;
      printf("conv_in %s B %3d: rc %d  %7.1f us  %.2f TB/s written   sample rel-rms %.3e max-abs %.3e\n", dt == IDF_BF16 ? "bf16" : "fp16", B, rc, us,
             bytes / (us * 1e-6) / 1e12, std::sqrt(se / sr), mx);
    }
  return 0;
}
