// ARCHIVED (round 5, VERDICT r4 'What's weak' 6): the one-pass GroupNorm of round 4 -- row chunk in registers, device-wide rendezvous per
// sample -- as it stood in commit ab130a9^ (instancediffusion_amd/csrc/norms.hip) before it was removed: correct on 14 shapes,
// 10x slower with agent-scope fences, 1.3-3x slower with device-scope atomics only (profiles/r04_gn_onepass_*.log).
// Not built, not linked; kept next to its logs like attention2 / attention5 / gemm_big_r02.
// norms.hip -- GroupNorm(32)+SiLU and LayerNorm for gfx950.  HBM-bandwidth-bound kernels:
//   every global access is a 16-byte (8 x 16-bit) per-lane vector, fully coalesced on the NHWC / token-major layout;
//   statistics are fp32 (the reference forces fp32 GroupNorm, util.py:223-226).
// GroupNorm is two launches: (1) per-(batch, row-chunk) partial (mean, M2) per group -- every thread accumulates its rows
// SHIFTED by its first row's value (no E[x^2] - mu^2 cancellation however large |mean| / std is), partials are merged with
// Chan's parallel-variance formula in a fixed order (no float atomics -> bitwise run-to-run reproducible);
// (2) merge of the chunk partials in fp64 + normalise + affine (+SiLU).
// Algorithmic bytes: 2 B read + 2 B written per element (the 2nd read of x is served by L2 / Infinity Cache for the
// <= 100 MB activations of this UNet).
#include "common.h"
#include <cstdlib>

namespace {

constexpr int GN_GROUPS = 32;
constexpr int GN_MAX_CHUNKS = 64;     // row-chunks per batch element (large batches)
constexpr int GN_MAX_CHUNKS_SMALL = 256;

// Row chunks per batch element: HW / 32, at most 64 -- and for SMALL batches (B x chunks < 512 workgroups: the 2-row forwards
// of BASELINE config 2) finer, down to 8 rows per chunk and at most 256 chunks: there the statistics pass is a chain of
// dependent trips to memory per thread (rows / (TY x 4) of them, ~10 us for a 64 x 64 x 320 sample on 128 workgroups), not
// bandwidth.  A function of (B, HW) only, so a given launch shape always reduces in the same order (bitwise reproducible).
inline int gn_nchunks(int B, int HW) {
  int n = HW / 32;
  if (n < 1) n = 1;
  if (n > GN_MAX_CHUNKS) n = GN_MAX_CHUNKS;
  while ((long long)B * n < 512 && HW / (2 * n) >= 8 && 2 * n <= GN_MAX_CHUNKS_SMALL) n *= 2;
  return n;
}

// Chan et al. merge of two (count, mean, M2) summaries
__device__ __forceinline__ void chan_merge(float& n, float& mean, float& m2, float nb, float mb, float m2b) {
  if (nb == 0.f) return;
  const float nt = n + nb;
  const float d = mb - mean;
  const float w = nb / nt;
  mean = fmaf(d, w, mean);
  m2 = m2 + m2b + d * d * n * w;
  n = nt;
}

// partial[b][chunk][g][2] = (mean, M2) over the rows of the chunk (count = rows in the chunk x channels per group)
template <int DT, int UNR>
__device__ __forceinline__ void gn_stats_body(float* sm /* [2][TY][C]: mean, M2 per (row lane, channel) */,
                                              const unsigned short* __restrict__ x, float* __restrict__ partial,
                                              int HW, int C, int nchunks) {
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int cpr = C >> 3;                                        // 16-B chunks per row
  const int TX = cpr < 256 ? cpr : 256;
  const int TY = 256 / TX;
  const int tid = threadIdx.x;
  const int tx = tid % TX, ty = tid / TX;
  const int rows_per = (HW + nchunks - 1) / nchunks;
  const int r_begin = chunk * rows_per;
  const int r_end = min(HW, r_begin + rows_per);
  const unsigned short* xb = x + (size_t)b * HW * C;
  float* s_mean = sm;
  float* s_m2 = sm + (size_t)TY * C;
  if (ty < TY) {
    for (int cc = tx; cc < cpr; cc += TX) {
      float s[8], q[8], piv[8];
      int cnt = 0;
#pragma unroll
      for (int j = 0; j < 8; ++j) { s[j] = 0.f; q[j] = 0.f; piv[j] = 0.f; }
      // four rows in flight per thread (latency-bound otherwise)
      for (int r0 = r_begin + ty; r0 < r_end; r0 += UNR * TY) {
        u32x4 v[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
          const int r = min(r0 + u * TY, r_end - 1);
          v[u] = *reinterpret_cast<const u32x4*>(xb + (size_t)r * C + cc * 8);
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
          if (r0 + u * TY >= r_end) break;
          float f[8];
          unpack8<DT>(v[u], f);
          if (cnt == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) piv[j] = f[j];          // shift = this thread's first value of the channel
          }
          ++cnt;
#pragma unroll
          for (int j = 0; j < 8; ++j) { const float d = f[j] - piv[j]; s[j] += d; q[j] = fmaf(d, d, q[j]); }
        }
      }
      const float n = (float)cnt, inv = cnt > 0 ? 1.0f / n : 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float ms = s[j] * inv;                             // mean of the shifted values (small)
        s_mean[ty * C + cc * 8 + j] = piv[j] + ms;
        s_m2[ty * C + cc * 8 + j] = fmaxf(q[j] - s[j] * ms, 0.f);
      }
    }
  }
  __syncthreads();
  if (tid < GN_GROUPS) {
    const int cpg = C / GN_GROUPS;
    float n = 0.f, mean = 0.f, m2 = 0.f;
    const float inv_cpg = 1.0f / (float)cpg;
    for (int y = 0; y < TY; ++y) {
      // the cpg channel summaries of one row lane have EQUAL counts: merged without divisions (mean of means,
      // M2 = sum M2_j + n_y * sum (mean_j - mean)^2), then one general Chan merge per row lane
      const int i0 = y * C + tid * cpg;
      const int rows_y = (r_end - r_begin - y + TY - 1) / TY;        // rows r_begin + y, + TY, ... of this chunk
      if (rows_y <= 0) continue;
      const float ny = (float)rows_y;
      float ms = 0.f, q = 0.f;
      for (int j = 0; j < cpg; ++j) { ms += s_mean[i0 + j]; q += s_m2[i0 + j]; }
      const float my = ms * inv_cpg;
      float dev = 0.f;
      for (int j = 0; j < cpg; ++j) { const float d = s_mean[i0 + j] - my; dev = fmaf(d, d, dev); }
      chan_merge(n, mean, m2, ny * (float)cpg, my, fmaf(ny, dev, q));
    }
    float* o = partial + (((size_t)b * nchunks + chunk) * GN_GROUPS + tid) * 2;
    o[0] = mean; o[1] = m2;
  }
}

template <int DT, int UNR>
__global__ __launch_bounds__(256) void gn_stats_kernel(const unsigned short* __restrict__ x, float* __restrict__ partial,
                                                      int HW, int C, int nchunks) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  gn_stats_body<DT, UNR>(sm, x, partial, HW, C, nchunks);
}

// `partial` is read through PLOAD: a plain load in the two-launch form, a device-scope atomic load in the single-launch form
// (the partials were written by other workgroups of the SAME launch, possibly on another XCD with its own L2)
template <int DT, int UNR, bool COHERENT>
__device__ __forceinline__ void gn_apply_body(float* sm /* scale[C], shift[C], mean[32], rstd[32] */,
                                              const unsigned short* __restrict__ x, unsigned short* __restrict__ out,
                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                              const float* partial, int HW, int C, int nchunks,
                                              float eps, int silu, int nblk_x) {
  auto PLOAD = [](const float* q) -> float {
    if constexpr (COHERENT) return __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return *q;
  };
  float* sc = sm;
  float* sh = sm + C;
  float* mean = sm + (2 * C > 512 ? 2 * C : 512);                // the first 2 KB double as the fp64 reduction scratch
  float* rstd = mean + GN_GROUPS;
  const int b = blockIdx.y, tid = threadIdx.x;
  const int cpg = C / GN_GROUPS;
  // Merge the chunk partials (mean_k, M2_k; count n_k = rows of chunk k x cpg): mean = sum n_k mean_k / N,
  // M2 = sum M2_k + sum n_k (mean_k - mean)^2 -- the pairwise (Chan) update written as two weighted sums, so that the 256
  // threads share the work (8 chunk slices x 32 groups, fp64, fixed order -> bitwise reproducible) and no thread runs a
  // chain of 64 fp64 divisions in front of every row block.
  {
    double* red = reinterpret_cast<double*>(sm);                  // [8][32], reused for scale/shift afterwards
    const int g = tid & (GN_GROUPS - 1), part = tid / GN_GROUPS;
    const int rows_per_c = (HW + nchunks - 1) / nchunks;
    const double N = (double)HW * (double)cpg;
    double acc = 0.0;
    for (int k = part; k < nchunks; k += 256 / GN_GROUPS) {
      const int rows_k = min(HW, (k + 1) * rows_per_c) - min(HW, k * rows_per_c);
      acc += (double)rows_k * (double)cpg * (double)PLOAD(partial + (((size_t)b * nchunks + k) * GN_GROUPS + g) * 2);
    }
    red[part * GN_GROUPS + g] = acc;
    __syncthreads();
    double mu = 0.0;
#pragma unroll
    for (int q = 0; q < 256 / GN_GROUPS; ++q) mu += red[q * GN_GROUPS + g];
    mu /= N;
    __syncthreads();
    acc = 0.0;
    for (int k = part; k < nchunks; k += 256 / GN_GROUPS) {
      const int rows_k = min(HW, (k + 1) * rows_per_c) - min(HW, k * rows_per_c);
      const float* pp = partial + (((size_t)b * nchunks + k) * GN_GROUPS + g) * 2;
      const double d = (double)PLOAD(pp) - mu;
      acc += (double)PLOAD(pp + 1) + (double)rows_k * (double)cpg * d * d;
    }
    red[part * GN_GROUPS + g] = acc;
    __syncthreads();
    if (tid < GN_GROUPS) {
      double m2 = 0.0;
#pragma unroll
      for (int q = 0; q < 256 / GN_GROUPS; ++q) m2 += red[q * GN_GROUPS + tid];
      mean[tid] = (float)mu;
      rstd[tid] = (float)(1.0 / sqrt(m2 / N + (double)eps));
    }
    __syncthreads();
  }
  for (int ch = tid; ch < C; ch += 256) {
    const int g = ch / cpg;
    const float w = gamma[ch] * rstd[g];
    sc[ch] = w;
    sh[ch] = beta[ch] - mean[g] * w;
  }
  __syncthreads();
  // Stream this block's row chunk with a FIXED channel chunk per thread (tx = 16-B column, ty = row lane): the 8
  // scale/shift pairs live in registers for the whole row loop (the previous version re-read them from LDS per
  // element with an 8-float lane stride: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.86).
  const int cpr = C >> 3;
  const int TX = cpr < 256 ? cpr : 256;
  const int TY = 256 / TX;
  const int tx = tid % TX, ty = tid / TX;
  const int rows_per = (HW + nblk_x - 1) / nblk_x;
  const int r_begin = blockIdx.x * rows_per;
  const int r_end = min(HW, r_begin + rows_per);
  const unsigned short* xb = x + (size_t)b * HW * C;
  unsigned short* ob = out + (size_t)b * HW * C;
  if (ty >= TY) return;
  for (int cc = tx; cc < cpr; cc += TX) {
    float scr[8], shr[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { scr[j] = sc[cc * 8 + j]; shr[j] = sh[cc * 8 + j]; }
    // four rows in flight per thread (one 16-B load each before the first use): the kernel is latency-bound otherwise
    for (int r0 = r_begin + ty; r0 < r_end; r0 += UNR * TY) {
      u32x4 v[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int r = min(r0 + u * TY, r_end - 1);
        v[u] = *reinterpret_cast<const u32x4*>(xb + (size_t)r * C + cc * 8);
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int r = r0 + u * TY;
        if (r >= r_end) break;
        float f[8];
        unpack8<DT>(v[u], f);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float y = fmaf(f[j], scr[j], shr[j]);
          f[j] = silu ? silu_f(y) : y;
        }
        *reinterpret_cast<u32x4*>(ob + (size_t)r * C + cc * 8) = pack8<DT>(f);
      }
    }
  }
}

template <int DT, int UNR>
__global__ __launch_bounds__(256) void gn_apply_kernel(const unsigned short* __restrict__ x, unsigned short* __restrict__ out,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      const float* __restrict__ partial, int HW, int C, int nchunks,
                                                      float eps, int silu, int nblk_x) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  gn_apply_body<DT, UNR, false>(sm, x, out, gamma, beta, partial, HW, C, nchunks, eps, silu, nblk_x);
}

// Single-launch form for small batches (B x chunks <= 256 workgroups, all resident at once): statistics of the workgroup's
// row chunk -> partial; arrive on the sample's counter; wait until all `nchunks` workgroups of the sample have arrived; merge
// the partials and normalise the SAME chunk (its second read is an L2 / Infinity-Cache hit a few microseconds after the
// first).  One launch instead of two for the ~60 GroupNorms of a forward whose cost at 2 rows is launches, not bytes.
// sync[2 b] = arrivals, sync[2 b + 1] = departures; the last workgroup to depart zeroes both, so a workspace that was zero
// before its first use stays valid for every later launch (idf_groupnorm's contract).  The wait is bounded: a workspace that
// was NOT zeroed makes the launch give up after ~0.5 s with wrong output instead of hanging the device.
template <int DT, int UNR>
__global__ __launch_bounds__(256) void gn_fused_kernel(const unsigned short* __restrict__ x, unsigned short* __restrict__ out,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      float* partial, unsigned* sync, int HW, int C, int nchunks,
                                                      float eps, int silu) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int b = blockIdx.y, tid = threadIdx.x;
  gn_stats_body<DT, UNR>(sm, x, partial, HW, C, nchunks);
  __threadfence();                                               // this workgroup's partial is visible device-wide ...
  __syncthreads();
  if (tid == 0) {
    __hip_atomic_fetch_add(sync + 2 * b, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);      // ... before it arrives
    int spins = 0;
    while (__hip_atomic_load(sync + 2 * b, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)nchunks && ++spins < (1 << 22))
      __builtin_amdgcn_s_sleep(4);
  }
  __syncthreads();
  gn_apply_body<DT, UNR, true>(sm, x, out, gamma, beta, partial, HW, C, nchunks, eps, silu, nchunks);
  __syncthreads();
  if (tid == 0) {
    const unsigned gone = __hip_atomic_fetch_add(sync + 2 * b + 1, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (gone == (unsigned)nchunks - 1u) {                        // every workgroup of the sample has read the partials
      __hip_atomic_store(sync + 2 * b, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(sync + 2 * b + 1, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// LayerNorm: one wave64 per row, row held in registers (<= 3 x 16-B chunks per lane -> C <= 1536), exact two-pass.
template <int DT>
__global__ __launch_bounds__(256) void ln_kernel(const unsigned short* __restrict__ x, int ldx, unsigned short* __restrict__ out,
                                                int ldo, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                int M, int C, float eps, int pH, int pW) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int cpr = C >> 3;
  const unsigned short* xr = x + (size_t)row * ldx;
  float f[3][8];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int cc = lane + 64 * i;
    if (cc < cpr) {
      u32x4 v = *reinterpret_cast<const u32x4*>(xr + cc * 8);
      unpack8<DT>(v, f[i]);
#pragma unroll
      for (int j = 0; j < 8; ++j) s += f[i][j];
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) f[i][j] = 0.f;
    }
  }
  const float mu = wave_sum(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int cc = lane + 64 * i;
    if (cc < cpr) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float dlt = f[i][j] - mu; q = fmaf(dlt, dlt, q); }
    }
  }
  const float rs = rsqrtf(wave_sum(q) / (float)C + eps);
  // output row: identity, or (pW > 0) the 2x2/stride-2 patch gather of an [B, pH, pW, C] image:
  // pixel (b, y, x) -> row (b, y/2, x/2), column block ((y&1)*2 + (x&1)) * C   (ConvNeXt downsample, convnext.py:78-82)
  size_t orow_idx = (size_t)row;
  int ocol = 0;
  if (pW > 0) {
    const int hw = pH * pW;
    const int b = row / hw, r = row - b * hw;
    const int y = r / pW, x2 = r - y * pW;
    orow_idx = ((size_t)b * (pH >> 1) + (y >> 1)) * (pW >> 1) + (x2 >> 1);
    ocol = ((y & 1) * 2 + (x2 & 1)) * C;
  }
  unsigned short* orow = out + orow_idx * ldo + ocol;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int cc = lane + 64 * i;
    if (cc < cpr) {
      const f32x4 g0 = *reinterpret_cast<const f32x4*>(gamma + cc * 8), g1 = *reinterpret_cast<const f32x4*>(gamma + cc * 8 + 4);
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(beta + cc * 8), b1 = *reinterpret_cast<const f32x4*>(beta + cc * 8 + 4);
      float y[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        y[j] = fmaf((f[i][j] - mu) * rs, g0[j], b0[j]);
        y[j + 4] = fmaf((f[i][j + 4] - mu) * rs, g1[j], b1[j]);
      }
      *reinterpret_cast<u32x4*>(orow + cc * 8) = pack8<DT>(y);
    }
  }
}

// row chunks of the single-launch form: every workgroup of the launch must be resident at once (<= 256 of them), small batches only
inline int gn_nchunks_fused(int B, int HW) {
  if (B > 8) return 0;
  int n = HW / 32;
  if (n < 1) n = 1;
  if (n > GN_MAX_CHUNKS) n = GN_MAX_CHUNKS;
  if ((long long)B * n > 256) return 0;
  while ((long long)B * n * 2 <= 256 && HW / (2 * n) >= 8) n *= 2;
  return n;
}
#ifndef IDF_GN_FUSED_DEFAULT
#define IDF_GN_FUSED_DEFAULT 0
#endif
int g_gn_fused = -1;
inline int gn_fused_mode() {
  if (g_gn_fused < 0) { const char* e = getenv("IDF_GN_FUSED"); g_gn_fused = e ? (atoi(e) != 0) : IDF_GN_FUSED_DEFAULT; }
  return g_gn_fused;
}

}  // namespace

int idf_gn_fused_set(int v) {                                   // idf_set_tuning(IDF_TUNE_GN_FUSED): returns the previous value
  const int prev = gn_fused_mode();
  g_gn_fused = v != 0;
  return prev;
}

// [B][chunks][32][2] partial (mean, M2) + 2 B counters of the single-launch form (zero before the first use, see idf.h)
extern "C" long long idf_groupnorm_ws_floats(int B, int HW) {
  return (long long)B * gn_nchunks(B, HW) * GN_GROUPS * 2 + 2LL * B;
}

extern "C" int idf_groupnorm(const void* x, void* out, const float* gamma, const float* beta, float* ws,
                             int B, int HW, int C, float eps, int silu, int dtype, void* stream) {
  if (!x || !out || !gamma || !beta || !ws) return IDF_E_ARG;
  if (B <= 0 || HW <= 0 || C <= 0 || (C % 32) || (C % 8)) return IDF_E_ARG;
  if (!aligned16(x) || !aligned16(out)) return IDF_E_ALIGN;
  hipStream_t s = (hipStream_t)stream;
  const int nchunks = gn_nchunks(B, HW);
  const int cpr = C / 8, TX = cpr < 256 ? cpr : 256, TY = 256 / TX;
  const size_t sm1 = (size_t)2 * TY * C * sizeof(float);
  const size_t sm2 = (size_t)((2 * C > 512 ? 2 * C : 512) + 2 * GN_GROUPS) * sizeof(float);
  if (sm1 > 64 * 1024 || sm2 > 64 * 1024) return IDF_E_UNSUPPORTED;
  int nblk = (HW + TY * 8 - 1) / (TY * 8);                       // >= 8 rows per thread-row, <= 256 blocks per batch
  if ((long long)B * nblk < 512) nblk = (HW + TY * 4 - 1) / (TY * 4);   // small batch: one trip of 4 rows in flight per thread
  if (nblk < 1) nblk = 1;
  if (nblk > 256) nblk = 256;
  dim3 g1(nchunks, B), g2(nblk, B);
  static int unr = -1;                                          // rows in flight per thread (IDF_GN_UNROLL=1|4 for A/B runs)
  if (unr < 0) { const char* e = getenv("IDF_GN_UNROLL"); unr = (e && atoi(e) == 1) ? 1 : 4; }
  const int nf = gn_fused_mode() ? gn_nchunks_fused(B, HW) : 0;
  if (nf > 0 && unr == 4) {                                      // small batch: statistics + normalisation in ONE launch
    unsigned* sync = reinterpret_cast<unsigned*>(ws + (size_t)B * nchunks * GN_GROUPS * 2);
    const size_t smf = sm1 > sm2 ? sm1 : sm2;
    dim3 gf(nf, B);
    if (dtype == IDF_BF16)
      hipLaunchKernelGGL((gn_fused_kernel<IDF_BF16, 4>), gf, dim3(256), smf, s, (const unsigned short*)x, (unsigned short*)out, gamma, beta,
                         ws, sync, HW, C, nf, eps, silu);
    else if (dtype == IDF_F16)
      hipLaunchKernelGGL((gn_fused_kernel<IDF_F16, 4>), gf, dim3(256), smf, s, (const unsigned short*)x, (unsigned short*)out, gamma, beta,
                         ws, sync, HW, C, nf, eps, silu);
    else
      return IDF_E_UNSUPPORTED;
    return idf_launch_status();
  }
#define IDF_GN_LAUNCH(DT, U)                                                                                              \
  hipLaunchKernelGGL((gn_stats_kernel<DT, U>), g1, dim3(256), sm1, s, (const unsigned short*)x, ws, HW, C, nchunks);      \
  hipLaunchKernelGGL((gn_apply_kernel<DT, U>), g2, dim3(256), sm2, s, (const unsigned short*)x, (unsigned short*)out,     \
                     gamma, beta, ws, HW, C, nchunks, eps, silu, nblk);
  if (dtype == IDF_BF16) {
    if (unr == 1) { IDF_GN_LAUNCH(IDF_BF16, 1) } else { IDF_GN_LAUNCH(IDF_BF16, 4) }
  } else if (dtype == IDF_F16) {
    if (unr == 1) { IDF_GN_LAUNCH(IDF_F16, 1) } else { IDF_GN_LAUNCH(IDF_F16, 4) }
  } else {
    return IDF_E_UNSUPPORTED;
  }
#undef IDF_GN_LAUNCH
  return idf_launch_status();
}

extern "C" int idf_layernorm(const void* x, int ldx, void* out, int ldo, const float* gamma, const float* beta,
                             int M, int C, float eps, int dtype, void* stream) {
  if (!x || !out || !gamma || !beta) return IDF_E_ARG;
  if (M <= 0 || C <= 0 || (C % 8) || C > 1536) return IDF_E_ARG;
  if ((ldx % 8) || (ldo % 8) || !aligned16(x) || !aligned16(out) || !aligned16(gamma) || !aligned16(beta)) return IDF_E_ALIGN;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((M + 3) / 4);
  if (dtype == IDF_BF16)
    hipLaunchKernelGGL(ln_kernel<IDF_BF16>, grid, dim3(256), 0, s, (const unsigned short*)x, ldx, (unsigned short*)out, ldo, gamma, beta, M, C, eps, 0, 0);
  else if (dtype == IDF_F16)
    hipLaunchKernelGGL(ln_kernel<IDF_F16>, grid, dim3(256), 0, s, (const unsigned short*)x, ldx, (unsigned short*)out, ldo, gamma, beta, M, C, eps, 0, 0);
  else
    return IDF_E_UNSUPPORTED;
  return idf_launch_status();
}

// Row statistics for a LayerNorm folded into the consumer GEMM (IDF_EPI_LN_ROW / IDF_EPI_LN_COL): one wave64 per row, the
// row in registers, exact two-pass like ln_kernel -- but nothing is written except (mu, rstd): half of LayerNorm's HBM
// traffic, and the normalised matrix is never re-read.
template <int DT>
__global__ __launch_bounds__(256) void row_stats_kernel(const unsigned short* __restrict__ x, int ldx, float* __restrict__ stats,
                                                       int M, int C, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int cpr = C >> 3;
  const unsigned short* xr = x + (size_t)row * ldx;
  float f[3][8];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int cc = lane + 64 * i;
    if (cc < cpr) {
      u32x4 v = *reinterpret_cast<const u32x4*>(xr + cc * 8);
      unpack8<DT>(v, f[i]);
#pragma unroll
      for (int j = 0; j < 8; ++j) s += f[i][j];
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) f[i][j] = 0.f;
    }
  }
  const float mu = wave_sum(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int cc = lane + 64 * i;
    if (cc < cpr) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float dlt = f[i][j] - mu; q = fmaf(dlt, dlt, q); }
    }
  }
  const float rs = rsqrtf(wave_sum(q) / (float)C + eps);
  if (lane == 0) *reinterpret_cast<f32x2*>(stats + 2 * (size_t)row) = f32x2{mu, rs};
}

extern "C" int idf_row_stats(const void* x, int ldx, float* stats, int M, int C, float eps, int dtype, void* stream) {
  if (!x || !stats) return IDF_E_ARG;
  if (M <= 0 || C <= 0 || (C % 8) || C > 1536) return IDF_E_ARG;
  if ((ldx % 8) || !aligned16(x) || (((uintptr_t)stats) & 7u)) return IDF_E_ALIGN;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((M + 3) / 4);
  if (dtype == IDF_BF16) hipLaunchKernelGGL(row_stats_kernel<IDF_BF16>, grid, dim3(256), 0, s, (const unsigned short*)x, ldx, stats, M, C, eps);
  else if (dtype == IDF_F16) hipLaunchKernelGGL(row_stats_kernel<IDF_F16>, grid, dim3(256), 0, s, (const unsigned short*)x, ldx, stats, M, C, eps);
  else return IDF_E_UNSUPPORTED;
  return idf_launch_status();
}

extern "C" int idf_layernorm_patch2(const void* x, void* out, int ldo, const float* gamma, const float* beta,
                                    int B, int H, int W, int C, float eps, int dtype, void* stream) {
  if (!x || !out || !gamma || !beta) return IDF_E_ARG;
  if (B <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1) || C <= 0 || (C % 8) || C > 1536 || ldo < 4 * C) return IDF_E_ARG;
  if ((ldo % 8) || !aligned16(x) || !aligned16(out) || !aligned16(gamma) || !aligned16(beta)) return IDF_E_ALIGN;
  hipStream_t s = (hipStream_t)stream;
  const int M = B * H * W;
  dim3 grid((M + 3) / 4);
  if (dtype == IDF_BF16)
    hipLaunchKernelGGL(ln_kernel<IDF_BF16>, grid, dim3(256), 0, s, (const unsigned short*)x, C, (unsigned short*)out, ldo, gamma, beta, M, C, eps, H, W);
  else if (dtype == IDF_F16)
    hipLaunchKernelGGL(ln_kernel<IDF_F16>, grid, dim3(256), 0, s, (const unsigned short*)x, C, (unsigned short*)out, ldo, gamma, beta, M, C, eps, H, W);
  else
    return IDF_E_UNSUPPORTED;
  return idf_launch_status();
}
