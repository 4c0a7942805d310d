// scaleu.hip -- ScaleU skip re-scaling (reference openaimodel.py:519-539, Fourier_filter :25-48) for gfx950.
//
// The reference runs FFT2 -> scale the 2x2 centred low-frequency window by s -> IFFT2 -> real part, then
// concatenates with the channel-scaled backbone features.  Because only the four bins (u,v) in {-1,0}^2 are
// touched, the filter is EXACTLY
//     y = x + (s-1) * (1/HW) * [ S0 + sum_{k=1..3} (C_k cos(phi_k) + S_k sin(phi_k)) ]
// with phi_1 = 2*pi*h/H, phi_2 = 2*pi*w/W, phi_3 = phi_1 + phi_2, S0 = sum x, C_k = sum x cos(phi_k),
// S_k = sum x sin(phi_k) per (b, c) plane (verified against the FFT in tests/test_oracle_golden.py).
// Two launches: (1) the 7 plane sums (one workgroup owns 64 channels x all HW rows: full 128-B lines per row);
// (2) one fused pass that writes BOTH halves of the concat buffer.  HBM-bound; algorithmic bytes =
// 2 B * (|h| + |skip| read + |h|+|skip| written).
#include "common.h"

namespace {


// coef[b][c][8]: S0, C1, S1, C2, S2, C3, S3, (pad)
template <int DT>
__global__ __launch_bounds__(256) void scaleu_stats_kernel(const unsigned short* __restrict__ skip, float* __restrict__ coef,
                                                          int H, int W, int Cs) {
  __shared__ float tw[4 * 128];                                  // cosH[H], sinH[H], cosW[W], sinW[W] (H,W <= 128)
  __shared__ float red[32][8][8][7];                             // 57 KB                         // [ty][tx][j][7]
  const int b = blockIdx.y;
  const int tid = threadIdx.x, tx = tid & 7, ty = tid >> 3;      // 8 chunk-columns x 32 row-lanes
  const int cc = blockIdx.x * 8 + tx;                            // 16-B chunk column
  const int cpr = Cs >> 3;
  for (int i = tid; i < H; i += 256) { float s, c; sincospif(2.0f * (float)i / (float)H, &s, &c); tw[i] = c; tw[128 + i] = s; }
  for (int i = tid; i < W; i += 256) { float s, c; sincospif(2.0f * (float)i / (float)W, &s, &c); tw[256 + i] = c; tw[384 + i] = s; }
  __syncthreads();
  float acc[8][7];
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int k = 0; k < 7; ++k) acc[j][k] = 0.f;
  const int HW = H * W;
  if (cc < cpr) {
    const unsigned short* sb = skip + (size_t)b * HW * Cs + cc * 8;
    for (int r = ty; r < HW; r += 32) {
      const int h = r / W, w = r - h * W;
      const float ch = tw[h], sh = tw[128 + h], cw = tw[256 + w], sw = tw[384 + w];
      const float c3 = ch * cw - sh * sw, s3 = sh * cw + ch * sw;
      u32x4 v = *reinterpret_cast<const u32x4*>(sb + (size_t)r * Cs);
      float f[8];
      unpack8<DT>(v, f);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        acc[j][0] += f[j];
        acc[j][1] = fmaf(f[j], ch, acc[j][1]); acc[j][2] = fmaf(f[j], sh, acc[j][2]);
        acc[j][3] = fmaf(f[j], cw, acc[j][3]); acc[j][4] = fmaf(f[j], sw, acc[j][4]);
        acc[j][5] = fmaf(f[j], c3, acc[j][5]); acc[j][6] = fmaf(f[j], s3, acc[j][6]);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int k = 0; k < 7; ++k) red[ty][tx][j][k] = acc[j][k];
  __syncthreads();
  // 8 columns x 8 channels x 7 sums = 448 outputs, fixed-order reduction over ty
  for (int o = tid; o < 8 * 8 * 7; o += 256) {
    const int k = o % 7, j = (o / 7) % 8, x = o / 56;
    float s = 0.f;
    for (int y = 0; y < 32; ++y) s += red[y][x][j][k];
    const int c2 = blockIdx.x * 8 + x;
    if (c2 < cpr) coef[((size_t)b * Cs + c2 * 8 + j) * 8 + k] = s;
  }
}

template <int DT>
__global__ __launch_bounds__(256) void scaleu_apply_kernel(const unsigned short* __restrict__ hin, const unsigned short* __restrict__ skip,
                                                          unsigned short* __restrict__ out, const float* __restrict__ hscale,
                                                          const float* __restrict__ sm1p, const float* __restrict__ coef,
                                                          int B, int H, int W, int Ch, int Cs) {
  const int Ct = Ch + Cs;
  const int cprT = Ct >> 3, cprH = Ch >> 3;
  const int HW = H * W;
  const size_t total = (size_t)B * HW * cprT;
  const float sm1 = sm1p[0];                                     // tanh(scaleu_s) = s - 1
  const float inv_hw = 1.0f / (float)HW;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int cc = (int)(i % cprT);
    const size_t pix = i / cprT;                                 // b*HW + r
    float f[8];
    if (cc < cprH) {
      u32x4 v = *reinterpret_cast<const u32x4*>(hin + pix * Ch + cc * 8);
      unpack8<DT>(v, f);
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] *= hscale[cc * 8 + j];
    } else {
      const int cs = cc - cprH;
      const int b = (int)(pix / HW), r = (int)(pix - (size_t)b * HW);
      const int h = r / W, w = r - h * W;
      float sh, ch, sw, cw;
      sincospif(2.0f * (float)h / (float)H, &sh, &ch);
      sincospif(2.0f * (float)w / (float)W, &sw, &cw);
      const float c3 = ch * cw - sh * sw, s3 = sh * cw + ch * sw;
      u32x4 v = *reinterpret_cast<const u32x4*>(skip + pix * Cs + cs * 8);
      unpack8<DT>(v, f);
      const float* cf = coef + ((size_t)b * Cs + cs * 8) * 8;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(cf + j * 8), bq = *reinterpret_cast<const f32x4*>(cf + j * 8 + 4);
        const float low = (a[0] + a[1] * ch + a[2] * sh + a[3] * cw + bq[0] * sw + bq[1] * c3 + bq[2] * s3) * inv_hw;
        f[j] = fmaf(sm1, low, f[j]);
      }
    }
    *reinterpret_cast<u32x4*>(out + pix * Ct + cc * 8) = pack8<DT>(f);
  }
}

}  // namespace

extern "C" int idf_scaleu_concat(const void* h, const void* skip, void* out, const float* hscale, const float* sm1,
                                 float* ws, int B, int H, int W, int Ch, int Cs, int dtype, void* stream) {
  if (!h || !skip || !out || !hscale || !sm1 || !ws) return IDF_E_ARG;
  if (B <= 0 || H <= 0 || W <= 0 || H > 128 || W > 128 || (Ch % 8) || (Cs % 8) || Ch <= 0 || Cs <= 0) return IDF_E_ARG;
  if (!aligned16(h) || !aligned16(skip) || !aligned16(out) || !aligned16(ws)) return IDF_E_ALIGN;
  hipStream_t s = (hipStream_t)stream;
  dim3 g1((Cs / 8 + 7) / 8, B);
  const size_t total = (size_t)B * H * W * ((Ch + Cs) / 8);
  int nblk = (int)((total + 256 * 4 - 1) / (256 * 4));
  if (nblk < 1) nblk = 1;
  if (nblk > 4096) nblk = 4096;
  if (dtype == IDF_BF16) {
    hipLaunchKernelGGL(scaleu_stats_kernel<IDF_BF16>, g1, dim3(256), 0, s, (const unsigned short*)skip, ws, H, W, Cs);
    hipLaunchKernelGGL(scaleu_apply_kernel<IDF_BF16>, dim3(nblk), dim3(256), 0, s, (const unsigned short*)h,
                       (const unsigned short*)skip, (unsigned short*)out, hscale, sm1, ws, B, H, W, Ch, Cs);
  } else if (dtype == IDF_F16) {
    hipLaunchKernelGGL(scaleu_stats_kernel<IDF_F16>, g1, dim3(256), 0, s, (const unsigned short*)skip, ws, H, W, Cs);
    hipLaunchKernelGGL(scaleu_apply_kernel<IDF_F16>, dim3(nblk), dim3(256), 0, s, (const unsigned short*)h,
                       (const unsigned short*)skip, (unsigned short*)out, hscale, sm1, ws, B, H, W, Ch, Cs);
  } else {
    return IDF_E_UNSUPPORTED;
  }
  return idf_launch_status();
}
