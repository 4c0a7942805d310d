// scaleu.hip -- ScaleU skip re-scaling (reference openaimodel.py:519-539, Fourier_filter :25-48) for gfx950.
//
// The reference runs FFT2 -> scale the 2x2 centred low-frequency window by s -> IFFT2 -> real part, then
// concatenates with the channel-scaled backbone features.  Because only the four bins (u,v) in {-1,0}^2 are
// touched, the filter is EXACTLY
//     y = x + (s-1) * (1/HW) * [ S0 + sum_{k=1..3} (C_k cos(phi_k) + S_k sin(phi_k)) ]
// with phi_1 = 2*pi*h/H, phi_2 = 2*pi*w/W, phi_3 = phi_1 + phi_2, S0 = sum x, C_k = sum x cos(phi_k),
// S_k = sum x sin(phi_k) per (b, c) plane (verified against the FFT in tests/test_oracle_golden.py).
// Two launches: (1) partial plane sums per row slice (a workgroup owns 64 channels x HW/S rows: full 128-B lines per row,
// 4 rows in flight per thread); (2) one fused pass that sums the slices in fixed order, keeps a thread's 8 x 7 coefficients
// in registers and writes BOTH halves of the concat buffer.  HBM-bound; algorithmic bytes =
// 2 B * (|h| + |skip| read + |h|+|skip| written).
#include "common.h"

namespace {


constexpr int MAX_SLICES = 8;

// Partial plane sums.  part[b][z][c][8]: S0, C1, S1, C2, S2, C3, S3, (pad) over the rows of slice z.
// grid (ceil(Cs/64), S, B): a workgroup owns 64 channels (8 x 16-B chunks: full 128-B lines per row) x HW/S rows.
template <int DT>
__global__ __launch_bounds__(256) void scaleu_stats_kernel(const unsigned short* __restrict__ skip, float* __restrict__ part,
                                                          int H, int W, int Cs, int rows_per_slice) {
  __shared__ float tw[4 * 128];                                  // cosH[H], sinH[H], cosW[W], sinW[W] (H,W <= 128)
  __shared__ float red[4][8][8][7];                              // per-wave partials: 7 KB
  const int b = blockIdx.z, z = blockIdx.y, S = gridDim.y;
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
  const int r_begin = z * rows_per_slice, r_end = min(HW, r_begin + rows_per_slice);
  if (cc < cpr) {
    const unsigned short* sb = skip + (size_t)b * HW * Cs + cc * 8;
    for (int r0 = r_begin + ty; r0 < r_end; r0 += 128) {          // 4 rows (r0, +32, +64, +96) in flight per thread
      u32x4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int r = min(r0 + 32 * u, r_end - 1);
        v[u] = *reinterpret_cast<const u32x4*>(sb + (size_t)r * Cs);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int r = r0 + 32 * u;
        if (r < r_end) {
          const int h = r / W, w = r - h * W;
          const float ch = tw[h], sh = tw[128 + h], cw = tw[256 + w], sw = tw[384 + w];
          const float c3 = ch * cw - sh * sw, s3 = sh * cw + ch * sw;
          float f[8];
          unpack8<DT>(v[u], f);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            acc[j][0] += f[j];
            acc[j][1] = fmaf(f[j], ch, acc[j][1]); acc[j][2] = fmaf(f[j], sh, acc[j][2]);
            acc[j][3] = fmaf(f[j], cw, acc[j][3]); acc[j][4] = fmaf(f[j], sw, acc[j][4]);
            acc[j][5] = fmaf(f[j], c3, acc[j][5]); acc[j][6] = fmaf(f[j], s3, acc[j][6]);
          }
        }
      }
    }
  }
  // fixed-order reduction: the 8 row-lanes of a wave by lane shuffles (lane = ty_in_wave*8 + tx), then the 4 waves via LDS
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int k = 0; k < 7; ++k) {
      float a = acc[j][k];
      a += __shfl_xor(a, 8, 64);
      a += __shfl_xor(a, 16, 64);
      a += __shfl_xor(a, 32, 64);
      acc[j][k] = a;
    }
  const int wave = tid >> 6, lane = tid & 63;
  if (lane < 8) {
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int k = 0; k < 7; ++k) red[wave][lane][j][k] = acc[j][k];
  }
  __syncthreads();
  for (int o = tid; o < 8 * 8 * 7; o += 256) {
    const int k = o % 7, j = (o / 7) % 8, x = o / 56;
    const float s = (red[0][x][j][k] + red[1][x][j][k]) + (red[2][x][j][k] + red[3][x][j][k]);
    const int c2 = blockIdx.x * 8 + x;
    if (c2 < cpr) part[(((size_t)b * S + z) * Cs + c2 * 8 + j) * 8 + k] = s;
  }
}

// One fused pass writing BOTH halves of the concat buffer.  grid (ceil((Ch+Cs)/64), R, B); a thread keeps ONE 16-B channel
// chunk (its 8 x 7 low-frequency coefficients, summed over the S slices and pre-multiplied by sm1 / HW, live in registers)
// and walks the rows of its slice; the per-row twiddles come from LDS tables.
template <int DT>
__global__ __launch_bounds__(256) void scaleu_apply_kernel(const unsigned short* __restrict__ hin, const unsigned short* __restrict__ skip,
                                                          unsigned short* __restrict__ out, const float* __restrict__ hscale,
                                                          const float* __restrict__ sm1p, const float* __restrict__ part,
                                                          int S, int H, int W, int Ch, int Cs, int rows_per_slice) {
  __shared__ float tw[4 * 128];
  const int b = blockIdx.z, z = blockIdx.y;
  const int tid = threadIdx.x, tx = tid & 7, ty = tid >> 3;
  const int Ct = Ch + Cs;
  const int cprT = Ct >> 3, cprH = Ch >> 3;
  const int cc = blockIdx.x * 8 + tx;
  const int HW = H * W;
  for (int i = tid; i < H; i += 256) { float s, c; sincospif(2.0f * (float)i / (float)H, &s, &c); tw[i] = c; tw[128 + i] = s; }
  for (int i = tid; i < W; i += 256) { float s, c; sincospif(2.0f * (float)i / (float)W, &s, &c); tw[256 + i] = c; tw[384 + i] = s; }
  __syncthreads();
  if (cc >= cprT) return;
  const int r_begin = z * rows_per_slice, r_end = min(HW, r_begin + rows_per_slice);
  unsigned short* ob = out + (size_t)b * HW * Ct + cc * 8;
  if (cc < cprH) {
    float hs[8];
    *reinterpret_cast<f32x4*>(hs) = *reinterpret_cast<const f32x4*>(hscale + cc * 8);
    *reinterpret_cast<f32x4*>(hs + 4) = *reinterpret_cast<const f32x4*>(hscale + cc * 8 + 4);
    const unsigned short* hb = hin + (size_t)b * HW * Ch + cc * 8;
    for (int r0 = r_begin + ty; r0 < r_end; r0 += 128) {
      u32x4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const u32x4*>(hb + (size_t)min(r0 + 32 * u, r_end - 1) * Ch);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int r = r0 + 32 * u;
        if (r < r_end) {
          float f[8];
          unpack8<DT>(v[u], f);
#pragma unroll
          for (int j = 0; j < 8; ++j) f[j] *= hs[j];
          *reinterpret_cast<u32x4*>(ob + (size_t)r * Ct) = pack8<DT>(f);
        }
      }
    }
  } else {
    const int cs = cc - cprH;
    const float k0 = sm1p[0] / (float)HW;                        // tanh(scaleu_s) = s - 1, and the 1/HW of the inverse DFT
    float cf[8][7];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int k = 0; k < 7; ++k) cf[j][k] = 0.f;
    for (int zz = 0; zz < S; ++zz) {                             // fixed order over the row slices
      const float* pp = part + (((size_t)b * S + zz) * Cs + cs * 8) * 8;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(pp + j * 8), bq = *reinterpret_cast<const f32x4*>(pp + j * 8 + 4);
        cf[j][0] += a[0]; cf[j][1] += a[1]; cf[j][2] += a[2]; cf[j][3] += a[3];
        cf[j][4] += bq[0]; cf[j][5] += bq[1]; cf[j][6] += bq[2];
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int k = 0; k < 7; ++k) cf[j][k] *= k0;
    const unsigned short* sb = skip + (size_t)b * HW * Cs + cs * 8;
    for (int r0 = r_begin + ty; r0 < r_end; r0 += 128) {
      u32x4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const u32x4*>(sb + (size_t)min(r0 + 32 * u, r_end - 1) * Cs);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int r = r0 + 32 * u;
        if (r < r_end) {
          const int h = r / W, w = r - h * W;
          const float ch = tw[h], sh = tw[128 + h], cw = tw[256 + w], sw = tw[384 + w];
          const float c3 = ch * cw - sh * sw, s3 = sh * cw + ch * sw;
          float f[8];
          unpack8<DT>(v[u], f);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            float low = cf[j][0];
            low = fmaf(cf[j][1], ch, low); low = fmaf(cf[j][2], sh, low);
            low = fmaf(cf[j][3], cw, low); low = fmaf(cf[j][4], sw, low);
            low = fmaf(cf[j][5], c3, low); low = fmaf(cf[j][6], s3, low);
            f[j] += low;
          }
          *reinterpret_cast<u32x4*>(ob + (size_t)r * Ct) = pack8<DT>(f);
        }
      }
    }
  }
}

}  // namespace

extern "C" int idf_scaleu_concat(const void* h, const void* skip, void* out, const float* hscale, const float* sm1,
                                 float* ws, int B, int H, int W, int Ch, int Cs, int dtype, void* stream) {
  if (!h || !skip || !out || !hscale || !sm1 || !ws) return IDF_E_ARG;
  if (B <= 0 || H <= 0 || W <= 0 || H > 128 || W > 128 || (Ch % 8) || (Cs % 8) || Ch <= 0 || Cs <= 0) return IDF_E_ARG;
  if (!aligned16(h) || !aligned16(skip) || !aligned16(out) || !aligned16(ws) || !aligned16(hscale)) return IDF_E_ALIGN;
  hipStream_t s = (hipStream_t)stream;
  const int HW = H * W;
  // row slices: enough workgroups to fill 256 CUs several times over; at least 128 rows (4 per thread) per slice
  int S = HW / 512;
  if (S < 1) S = 1;
  if (S > MAX_SLICES) S = MAX_SLICES;
  const int rps = (HW + S - 1) / S;
  int R = HW / 1024;
  if (R < 1) R = 1;
  if (R > 16) R = 16;
  const int rpa = (HW + R - 1) / R;
  dim3 g1((Cs / 8 + 7) / 8, S, B), g2(((Ch + Cs) / 8 + 7) / 8, R, B);
  if (dtype == IDF_BF16) {
    hipLaunchKernelGGL(scaleu_stats_kernel<IDF_BF16>, g1, dim3(256), 0, s, (const unsigned short*)skip, ws, H, W, Cs, rps);
    hipLaunchKernelGGL(scaleu_apply_kernel<IDF_BF16>, g2, dim3(256), 0, s, (const unsigned short*)h,
                       (const unsigned short*)skip, (unsigned short*)out, hscale, sm1, ws, S, H, W, Ch, Cs, rpa);
  } else if (dtype == IDF_F16) {
    hipLaunchKernelGGL(scaleu_stats_kernel<IDF_F16>, g1, dim3(256), 0, s, (const unsigned short*)skip, ws, H, W, Cs, rps);
    hipLaunchKernelGGL(scaleu_apply_kernel<IDF_F16>, g2, dim3(256), 0, s, (const unsigned short*)h,
                       (const unsigned short*)skip, (unsigned short*)out, hscale, sm1, ws, S, H, W, Ch, Cs, rpa);
  } else {
    return IDF_E_UNSUPPORTED;
  }
  return idf_launch_status();
}
