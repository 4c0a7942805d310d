// convnext.hip -- the non-GEMM pieces of UniFusion's instance-mask tokenizer (reference text_grounding_net.py:226-231,
// convnext.py:15-110) for gfx950.  Runs ONCE per conditioning (step-invariant), so these are simple coalesced kernels:
//   idf_seg_in_conv : Conv2d(30 -> 3, 3x3, pad 1) on the fp32 [B,30,S,S] mask stack, written directly as the stem's
//                     4x4/stride-4 patch matrix  P[b*(S/4)^2 + py*(S/4)+px][c*16 + ky*4 + kx]  (16-bit, K padded to 64)
//                     so the stem conv (convnext.py:73) is a plain idf_gemm.
//   idf_dwconv7x7   : depthwise 7x7, pad 3, NHWC 16-bit, fp32 accumulate (convnext.py:28,39).
// The 1x1 "pwconv" linears, the stem and the 2x2/stride-2 downsample convs are idf_gemm calls; the channel LayerNorms are
// idf_layernorm (with the 2x2 patch-gather output mapping for the downsample layers).
#include "common.h"

namespace {

template <int DT>
__global__ __launch_bounds__(256) void seg_in_conv_kernel(const float* __restrict__ segs, const float* __restrict__ w,
                                                         const float* __restrict__ bias, unsigned short* __restrict__ out,
                                                         int B, int Cin, int S, int ldo) {
  __shared__ float wl[3 * 32 * 9];                              // [co][ci][tap], Cin <= 32
  for (int i = threadIdx.x; i < 3 * Cin * 9; i += 256) wl[i] = w[i];
  __syncthreads();
  const size_t npix = (size_t)B * S * S;
  const size_t pix = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (pix >= npix) return;
  const int b = (int)(pix / ((size_t)S * S));
  const int r = (int)(pix - (size_t)b * S * S);
  const int y = r / S, x = r - y * S;
  float acc[3] = {bias[0], bias[1], bias[2]};
  for (int ci = 0; ci < Cin; ++ci) {
    const float* sp = segs + ((size_t)b * Cin + ci) * S * S;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int yy = y + ky - 1;
      if (yy < 0 || yy >= S) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int xx = x + kx - 1;
        if (xx < 0 || xx >= S) continue;
        const float v = sp[(size_t)yy * S + xx];
#pragma unroll
        for (int co = 0; co < 3; ++co) acc[co] = fmaf(v, wl[(co * Cin + ci) * 9 + ky * 3 + kx], acc[co]);
      }
    }
  }
  const int P = S >> 2;
  const size_t row = ((size_t)b * P + (y >> 2)) * P + (x >> 2);
  const int k0 = (y & 3) * 4 + (x & 3);
#pragma unroll
  for (int co = 0; co < 3; ++co) out[row * ldo + co * 16 + k0] = Elem<DT>::from_f32(acc[co]);
}

// w: [49][C] fp32 (tap-major so 8 consecutive channels are one 32-B run), one thread per (pixel, 8 channels)
template <int DT>
__global__ __launch_bounds__(256) void dwconv7_kernel(const unsigned short* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ bias, unsigned short* __restrict__ out,
                                                     int B, int H, int W, int C) {
  const int cg = C >> 3;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)B * H * W * cg) return;
  const int g = (int)(i % cg);
  const size_t pix = i / cg;
  const int b = (int)(pix / ((size_t)H * W));
  const int r = (int)(pix - (size_t)b * H * W);
  const int y = r / W, xx = r - y * W;
  float acc[8];
  {
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(bias + g * 8), b1 = *reinterpret_cast<const f32x4*>(bias + g * 8 + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) { acc[j] = b0[j]; acc[j + 4] = b1[j]; }
  }
  const unsigned short* xb = x + (size_t)b * H * W * C + g * 8;
  for (int ky = 0; ky < 7; ++ky) {
    const int yy = y + ky - 3;
    if (yy < 0 || yy >= H) continue;
#pragma unroll
    for (int kx = 0; kx < 7; ++kx) {
      const int xc = xx + kx - 3;
      if (xc < 0 || xc >= W) continue;
      const u32x4 v = *reinterpret_cast<const u32x4*>(xb + ((size_t)yy * W + xc) * C);
      float f[8];
      unpack8<DT>(v, f);
      const float* wp = w + (size_t)(ky * 7 + kx) * C + g * 8;
      const f32x4 w0 = *reinterpret_cast<const f32x4*>(wp), w1 = *reinterpret_cast<const f32x4*>(wp + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) { acc[j] = fmaf(f[j], w0[j], acc[j]); acc[j + 4] = fmaf(f[j + 4], w1[j], acc[j + 4]); }
    }
  }
  *reinterpret_cast<u32x4*>(out + pix * C + g * 8) = pack8<DT>(acc);
}

}  // namespace

extern "C" int idf_seg_in_conv(const float* segs, const float* w, const float* bias, void* out, int B, int Cin, int S,
                               int ldo, int dtype, void* stream) {
  if (!segs || !w || !bias || !out || B <= 0 || Cin <= 0 || Cin > 32 || S <= 0 || (S & 3) || ldo < 48) return IDF_E_ARG;
  hipStream_t s = (hipStream_t)stream;
  const size_t npix = (size_t)B * S * S;
  dim3 grid((unsigned)((npix + 255) / 256));
  if (dtype == IDF_BF16) hipLaunchKernelGGL(seg_in_conv_kernel<IDF_BF16>, grid, dim3(256), 0, s, segs, w, bias, (unsigned short*)out, B, Cin, S, ldo);
  else if (dtype == IDF_F16) hipLaunchKernelGGL(seg_in_conv_kernel<IDF_F16>, grid, dim3(256), 0, s, segs, w, bias, (unsigned short*)out, B, Cin, S, ldo);
  else return IDF_E_UNSUPPORTED;
  return idf_launch_status();
}

extern "C" int idf_dwconv7x7(const void* x, const float* w_tap_major, const float* bias, void* out, int B, int H, int W, int C,
                             int dtype, void* stream) {
  if (!x || !w_tap_major || !bias || !out || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C % 8)) return IDF_E_ARG;
  if (!aligned16(x) || !aligned16(out) || !aligned16(w_tap_major) || !aligned16(bias)) return IDF_E_ALIGN;
  hipStream_t s = (hipStream_t)stream;
  const size_t n = (size_t)B * H * W * (C / 8);
  dim3 grid((unsigned)((n + 255) / 256));
  if (dtype == IDF_BF16) hipLaunchKernelGGL(dwconv7_kernel<IDF_BF16>, grid, dim3(256), 0, s, (const unsigned short*)x, w_tap_major, bias, (unsigned short*)out, B, H, W, C);
  else if (dtype == IDF_F16) hipLaunchKernelGGL(dwconv7_kernel<IDF_F16>, grid, dim3(256), 0, s, (const unsigned short*)x, w_tap_major, bias, (unsigned short*)out, B, H, W, C);
  else return IDF_E_UNSUPPORTED;
  return idf_launch_status();
}
