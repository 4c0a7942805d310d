// misc.hip -- small latency-bound kernels of the sampling path (gfx950): timestep embedding, UniFusion token-MLP
// input builder, first conv from the fp32 NCHW latent, fused CFG / PLMS update, Multi-instance-Sampler merge.
#include "common.h"

namespace {

// util.py:160-180: out[b] = [cos(t f_k) | sin(t f_k)], f_k = exp(-ln(10000) k / half)
template <int DT>
__global__ void temb_kernel(const float* __restrict__ t, unsigned short* __restrict__ out, int B, int dim) {
  const int half = dim >> 1;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * half) return;
  const int b = i / half, k = i - b * half;
  const float f = expf(-9.210340371976184f * (float)k / (float)half);
  const float a = t[b] * f;
  out[(size_t)b * dim + k] = Elem<DT>::from_f32(cosf(a));
  out[(size_t)b * dim + half + k] = Elem<DT>::from_f32(sinf(a));
}

// text_grounding_net.py:216-287 + util.py:12-26.  One thread per output element of [rows, text_dim + 32*D].
template <int DT>
__global__ void unifusion_embed_kernel(const float* __restrict__ text, const float* __restrict__ loc,
                                       const float* __restrict__ tmask, const float* __restrict__ lmask,
                                       const float* __restrict__ null_text, const float* __restrict__ null_loc,
                                       const float* __restrict__ freqs,
                                       unsigned short* __restrict__ out, int rows, int text_dim, int D) {
  const int width = text_dim + 32 * D;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)rows * width) return;
  const int r = (int)(i / width), col = (int)(i - (size_t)r * width);
  float v;
  if (col < text_dim) {
    const float m = tmask[r];
    v = text[(size_t)r * text_dim + col] * m + (1.0f - m) * null_text[col];
  } else {
    const int e = col - text_dim;                 // e = (2*j + is_cos) * D + i
    const int blk = e / D, idx = e - blk * D;
    const int j = blk >> 1;
    const float freq = freqs[j];                  // freq_bands = 100 ** (j / 16), host-computed (util.py:17)
    const float a = freq * loc[(size_t)r * D + idx];
    const float fe = (blk & 1) ? cosf(a) : sinf(a);
    const float m = lmask[r];
    v = fe * m + (1.0f - m) * null_loc[e];
  }
  out[i] = Elem<DT>::from_f32(v);
}

// first conv (Cin = 4): NCHW fp32 latent -> NHWC 16-bit.  HBM-bound on the output (B*H*W*Cout*2 B; 168 MB at 64 rows of
// 64x64x320).  Round 3: PERSISTENT workgroups (two per CU) stage the fp32 weights (Cout*Cin*9 floats, 46 KB for 320x4) in
// LDS ONCE as [tap][ci][co] and then walk the (pixel quad, 8-channel group) work items grid-stride -- round 2 re-staged and
// transposed the whole weight tensor for every 64 pixels (361 us per launch = 0.46 TB/s).  A thread owns 8 output channels
// of FOUR horizontally adjacent pixels: the two 16-B weight vectors of a (tap, ci) are read from LDS once per quad
// (LDS traffic / 4), the 6 input values of a (ci, ky) row are shared by the quad's 3 x 4 (kx, pixel) products; threads of
// one quad differ in the channel group only, so their input reads broadcast and their 16-B stores tile a pixel's row.
template <int DT>
__global__ __launch_bounds__(256) void conv_in_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ bias, unsigned short* __restrict__ out,
                                                     int B, int Cin, int H, int W, int Cout) {
  extern __shared__ __attribute__((aligned(16))) float wl[];   // [9*Cin][Cout]
  const int K = 9 * Cin;
  for (int i = threadIdx.x; i < K * Cout; i += 256) {
    const int co = i / K, k = i - co * K;                        // w is [co][ci][ky][kx]; k = ci*9 + tap
    const int ci = k / 9, tap = k - ci * 9;
    wl[(tap * Cin + ci) * Cout + co] = w[i];
  }
  __syncthreads();
  const int cg = Cout >> 3;
  const int Wq = (W + 3) >> 2;
  const long long items = (long long)B * H * Wq * cg;
  for (long long it = (long long)blockIdx.x * 256 + threadIdx.x; it < items; it += (long long)gridDim.x * 256) {
    const int g = (int)(it % cg);
    const long long q = it / cg;
    const int xq = (int)(q % Wq);
    const long long r = q / Wq;
    const int y = (int)(r % H), b = (int)(r / H);
    const int x0 = xq * 4;
    float acc[4][8];
    {
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(bias + g * 8), b1 = *reinterpret_cast<const f32x4*>(bias + g * 8 + 4);
#pragma unroll
      for (int pq = 0; pq < 4; ++pq)
#pragma unroll
        for (int j = 0; j < 4; ++j) { acc[pq][j] = b0[j]; acc[pq][j + 4] = b1[j]; }
    }
    for (int ci = 0; ci < Cin; ++ci) {
      const float* xp = x + ((size_t)b * Cin + ci) * H * W;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const int yy = y + ky - 1;
        if (yy < 0 || yy >= H) continue;
        float xv[6];
#pragma unroll
        for (int t = 0; t < 6; ++t) {
          const int xc = x0 - 1 + t;
          xv[t] = (xc >= 0 && xc < W) ? xp[yy * W + xc] : 0.0f;
        }
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const float* wp = wl + ((ky * 3 + kx) * Cin + ci) * Cout + g * 8;
          const f32x4 w0 = *reinterpret_cast<const f32x4*>(wp), w1 = *reinterpret_cast<const f32x4*>(wp + 4);
#pragma unroll
          for (int pq = 0; pq < 4; ++pq) {
            const float v = xv[pq + kx];
#pragma unroll
            for (int j = 0; j < 4; ++j) { acc[pq][j] = fmaf(v, w0[j], acc[pq][j]); acc[pq][j + 4] = fmaf(v, w1[j], acc[pq][j + 4]); }
          }
        }
      }
    }
    unsigned short* op = out + (((size_t)b * H + y) * W + x0) * Cout + g * 8;
#pragma unroll
    for (int pq = 0; pq < 4; ++pq)
      if (x0 + pq < W) *reinterpret_cast<u32x4*>(op + (size_t)pq * Cout) = pack8<DT>(acc[pq]);
  }
}

// First conv on the matrix cores (round 4; Cin = 4, Cout = 320: the UNet's).  The quad kernel above is VALU-bound: 12 GFLOP of
// fp32 FMAs per 128-row launch = 96 us of pure issue, 309 us measured, against a 67-us HBM bound on the 335 MB it writes.
// As an implicit GEMM the work is M = B H W pixels x N = 320 x K = 36 -- nothing for the MFMA pipe -- but the operands are fp32
// and the layer feeds everything else, so the product is kept at fp32-class accuracy by splitting BOTH operands into a bf16
// head and a bf16 tail (x = xh + xl, w = wh + wl, each exact to 2^-17 relative) and summing xh wh + xh wl + xl wh on the
// MFMAs: K = 3 x 36 = 108 (+ 4 zero columns = 7 k-steps of 16), error ~2^-16 of a product, below the rounding of the 16-bit
// output.  One persistent 512-thread workgroup per CU: the split weight image [320][wh | wl | wh] lives in LDS for the whole
// launch (75 KB, 240-B rows: an odd number of 16-B slots, conflict-free fragment reads); per 256-pixel tile the workgroup
// gathers the 36 taps of its pixels from the NCHW latent (consecutive threads = consecutive pixels: coalesced), splits them and
// writes the A tile [256][xh | xh | xl] (60 KB); 8 waves as 4 (64 pixels) x 2 (160 channels), 70 MFMAs per wave and tile; then
// + bias and the 16-B stores.  Store-bound like the GEMM epilogues (~12 B/clk per CU): ~80 us per 128-row launch.
constexpr int CIM_KP = 120;                       // padded K (elements) of an LDS row: 240 B

template <int DT>
__global__ __launch_bounds__(512, 2) void conv_in_mfma_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ bias, unsigned short* __restrict__ out,
                                                              int B, int H, int W, int tiles) {
  constexpr int Cout = 320, Cin = 4, BM = 256;
  extern __shared__ __attribute__((aligned(16))) unsigned short cim_smem[];
  unsigned short* const Wl = cim_smem;                         // [320][120]
  unsigned short* const Al = cim_smem + Cout * CIM_KP;         // [256][120]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  const int wn = wave & 1, wm = wave >> 1;
  auto head = [](float v) { return Elem<IDF_BF16>::from_f32(v); };
  auto tail = [](float v, unsigned short h) { return Elem<IDF_BF16>::from_f32(v - Elem<IDF_BF16>::to_f32(h)); };
  // ---- weight image: w is [co][ci][ky][kx]; column t = tap * 4 + ci
  for (int i = tid; i < Cout * CIM_KP; i += 512) Wl[i] = 0;
  for (int i = tid; i < BM * CIM_KP; i += 512) Al[i] = 0;
  __syncthreads();
  for (int i = tid; i < Cout * 36; i += 512) {
    const int co = i / 36, k = i - co * 36;                    // k = ci * 9 + tap in the source
    const int ci = k / 9, tap = k - ci * 9;
    const float v = w[i];
    const unsigned short h = head(v), l = tail(v, h);
    unsigned short* row = Wl + co * CIM_KP + tap * 4 + ci;
    row[0] = h; row[36] = l; row[72] = h;
  }
  const long long HW = (long long)H * W, P = (long long)B * HW;
  for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    __syncthreads();                                           // the previous tile's fragment reads are done (and the images exist)
    // ---- A tile: thread -> pixel tid & 255 and the 18 columns t = (tid >> 8) + 2 r (t = tap * 4 + ci): the pixel's coordinates
    // are worked out once, the 18 gathers are independent loads in flight together
    {
      const int px = tid & 255, t0 = tid >> 8;
      const long long m = (long long)tile * BM + px;
      const bool live = m < P;
      const int b = live ? (int)(m / HW) : 0;
      const int rem = live ? (int)(m - (long long)b * HW) : 0;
      const int y = rem / W, xx0 = rem - y * W;
      const float* xb = x + (size_t)b * Cin * HW;
      float v[18];
#pragma unroll
      for (int r = 0; r < 18; ++r) {
        const int t = t0 + 2 * r;                               // t0 is 0 or 1: tap and channel follow from r and t0
        const int tap = t >> 2, ci = t & 3;
        const int yy = y + tap / 3 - 1, xx = xx0 + (tap - (tap / 3) * 3) - 1;
        const bool ok = live && yy >= 0 && yy < H && xx >= 0 && xx < W;
        v[r] = ok ? xb[(size_t)ci * HW + (size_t)yy * W + xx] : 0.0f;
      }
      unsigned short* row = Al + px * CIM_KP + t0;
#pragma unroll
      for (int r = 0; r < 18; ++r) {
        const unsigned short h = head(v[r]), l = tail(v[r], h);
        row[2 * r] = h; row[2 * r + 36] = h; row[2 * r + 72] = l;
      }
    }
    __syncthreads();
    // ---- 7 k-steps: acc[a][b][4 q + e] = D[channel 160 wn + 32 a + 8 q + 4 hi + e][pixel 64 wm + 32 b + l31]
    f32x16 acc[5][2];
#pragma unroll
    for (int a = 0; a < 5; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
    const unsigned short* wrow = Wl + (160 * wn + l31) * CIM_KP + 8 * hi;
    const unsigned short* arow = Al + (64 * wm + l31) * CIM_KP + 8 * hi;
#pragma unroll
    for (int ks = 0; ks < 7; ++ks) {
      u32x4 af[2], wf[5];
#pragma unroll
      for (int b = 0; b < 2; ++b) af[b] = *reinterpret_cast<const u32x4*>(arow + b * 32 * CIM_KP + 16 * ks);
#pragma unroll
      for (int a = 0; a < 5; ++a) wf[a] = *reinterpret_cast<const u32x4*>(wrow + a * 32 * CIM_KP + 16 * ks);
#pragma unroll
      for (int a = 0; a < 5; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = Elem<IDF_BF16>::mfma32(wf[a], af[b], acc[a][b]);
    }
    // ---- + bias, 16-bit stores: two v_permlane32_swap per register pair leave a lane with 16 consecutive channels of its pixel
#pragma unroll
    for (int a = 0; a < 5; ++a) {
      f32x4 bv[4];                                             // the lane's 16 consecutive channels of this fragment
#pragma unroll
      for (int j = 0; j < 4; ++j) bv[j] = *reinterpret_cast<const f32x4*>(bias + 160 * wn + 32 * a + 16 * hi + 4 * j);
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        float v[16];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const auto s02 = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[a][b][e]), __float_as_uint(acc[a][b][8 + e]), false, false);
          const auto s13 = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[a][b][4 + e]), __float_as_uint(acc[a][b][12 + e]), false, false);
          v[e] = __uint_as_float(s02[0]); v[4 + e] = __uint_as_float(s02[1]);
          v[8 + e] = __uint_as_float(s13[0]); v[12 + e] = __uint_as_float(s13[1]);
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] += bv[j >> 2][j & 3];
        const long long m = (long long)tile * BM + 64 * wm + 32 * b + l31;
        if (m < P) {
          unsigned short* o = out + (size_t)m * Cout + 160 * wn + 32 * a + 16 * hi;
          *reinterpret_cast<u32x4*>(o) = pack8<DT>(v);
          *reinterpret_cast<u32x4*>(o + 8) = pack8<DT>(v + 8);
        }
      }
    }
  }
}

__global__ void cfg_kernel(const float* __restrict__ ec, const float* __restrict__ eu, float g, float* __restrict__ et, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { const float u = eu[i]; et[i] = u + g * (ec[i] - u); }
}

// plms.py:130-165.  mode 0: e' = e_t; 1: (e_t + e_next)/2; 2: (3 e_t - e1)/2; 3: (23 e_t - 16 e1 + 5 e2)/12;
// 4: (55 e_t - 59 e1 + 37 e2 - 9 e3)/24.   x_prev = sqrt(a_prev) (x - sqrt(1-a_t) e')/sqrt(a_t) + sqrt(1-a_prev) e'
__global__ void plms_kernel(const float* __restrict__ x, const float* __restrict__ et, const float* __restrict__ e1,
                            const float* __restrict__ e2, const float* __restrict__ e3, const float* __restrict__ en,
                            int mode, float sqrt_at, float sqrt_aprev, float sqrt_1m_at, float sqrt_1m_aprev,
                            float* __restrict__ xo, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float e = et[i];
  float ep;
  switch (mode) {
    case 0: ep = e; break;
    case 1: ep = (e + en[i]) / 2.0f; break;
    case 2: ep = (3.0f * e - e1[i]) / 2.0f; break;
    case 3: ep = (23.0f * e - 16.0f * e1[i] + 5.0f * e2[i]) / 12.0f; break;
    default: ep = (55.0f * e - 59.0f * e1[i] + 37.0f * e2[i] - 9.0f * e3[i]) / 24.0f; break;
  }
  const float pred_x0 = (x[i] - sqrt_1m_at * ep) / sqrt_at;
  xo[i] = sqrt_aprev * pred_x0 + sqrt_1m_aprev * ep;
}

// plms_instance.py:112-135
__global__ void mis_merge_kernel(const float* __restrict__ lat, const int* __restrict__ boxes, float* __restrict__ out,
                                 int n_inst, int B, int C, int H, int W, int mode) {
  const long long per = (long long)B * C * H * W;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= per) return;
  if (mode == 0) {
    float s = 0.f;
    for (int k = 0; k <= n_inst; ++k) s += lat[(long long)k * per + i];
    out[i] = s / (float)(n_inst + 1);                // torch.mean(torch.stack(...), 0)
  } else {
    const int w = (int)(i % W), h = (int)((i / W) % H);
    float v = lat[i];
    for (int k = 0; k < n_inst; ++k) {               // later instances overwrite earlier ones, as in the loop :131-132
      const int* bb = boxes + 4 * k;
      // reference slices dim-2 (h) with bbox[0]:bbox[2] and dim-3 (w) with bbox[1]:bbox[3]
      if (h >= bb[0] && h < bb[2] && w >= bb[1] && w < bb[3]) v = lat[(long long)(k + 1) * per + i];
    }
    out[i] = v;
  }
}

template <int DT>
__global__ void cast_kernel(const float* __restrict__ x, unsigned short* __restrict__ out, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = Elem<DT>::from_f32(x[i]);
}

// Row softmax of an fp32 score matrix into 16-bit probabilities: p[r][j] = exp(scale*(s[r][j] - max_j s[r][:])) / sum.
// Used by the single-head, head-dim-512 attention of the VAE decoder mid block (model.py:178-196), whose head dim is
// beyond the register budget of the flash kernels: there the scores come from one batched MFMA GEMM (fp32 out), this
// kernel normalises them, and a second GEMM applies V.  One 256-thread workgroup per row; the row (n*4 bytes, 16 KB at
// n = 4096) is read three times (max, sum, write), passes 2 and 3 from L2.  HBM-bound: 4 B read + 2 B written / score.
template <int DT>
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ s, unsigned short* __restrict__ p,
                                                          int n, long long lds, long long ldp, float scale_log2e) {
  __shared__ float red[8];
  const float* sr = s + (size_t)blockIdx.x * (size_t)lds;
  unsigned short* pr = p + (size_t)blockIdx.x * (size_t)ldp;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n4 = n >> 2;
  float m = -3.0e38f;
  for (int i = tid; i < n4; i += 256) {
    const f32x4 v = reinterpret_cast<const f32x4*>(sr)[i];
    m = fmaxf(m, fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])));
  }
  m = wave_max(m);
  if (lane == 0) red[wave] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  const float mb = m * scale_log2e;                              // scale > 0: max of the scaled row
  float sum = 0.f;
  for (int i = tid; i < n4; i += 256) {
    const f32x4 v = reinterpret_cast<const f32x4*>(sr)[i];
#pragma unroll
    for (int j = 0; j < 4; ++j) sum += __builtin_amdgcn_exp2f(fmaf(v[j], scale_log2e, -mb));
  }
  sum = wave_sum(sum);
  if (lane == 0) red[4 + wave] = sum;
  __syncthreads();
  sum = (red[4] + red[5]) + (red[6] + red[7]);
  const float inv = 1.0f / sum;
  for (int i = tid; i < n4; i += 256) {
    const f32x4 v = reinterpret_cast<const f32x4*>(sr)[i];
    float e[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) e[j] = __builtin_amdgcn_exp2f(fmaf(v[j], scale_log2e, -mb)) * inv;
    *reinterpret_cast<u32x2*>(pr + 4 * i) = u32x2{pack2<DT>(e[0], e[1]), pack2<DT>(e[2], e[3])};
  }
}

// 1x1 convolution between small channel counts on fp32 NCHW (AutoencoderKL.post_quant_conv, autoencoder.py:34-35,
// with the 1/scale_factor of :33 folded in as in_scale): out[b][co][p] = bias[co] + sum_ci w[co][ci] * in_scale * x[b][ci][p].
__global__ void pointwise_nchw_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                      float* __restrict__ out, int B, int Cin, int Cout, long long HW, float in_scale) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)B * HW) return;
  const long long b = i / HW, pix = i - b * HW;
  float xv[16];
  for (int ci = 0; ci < Cin; ++ci) xv[ci] = x[(b * Cin + ci) * HW + pix] * in_scale;
  for (int co = 0; co < Cout; ++co) {
    float a = bias ? bias[co] : 0.f;
    for (int ci = 0; ci < Cin; ++ci) a = fmaf(w[co * Cin + ci], xv[ci], a);
    out[(b * Cout + co) * HW + pix] = a;
  }
}

inline dim3 grid1d(long long n, int bs = 256) { return dim3((unsigned)((n + bs - 1) / bs)); }

}  // namespace

extern "C" int idf_abi_version(void) { return IDF_ABI_VERSION; }
extern "C" const char* idf_build_info(void) { return "libidf_gfx950 (hand-written HIP, gfx950 / CDNA4, wave64, MFMA 32x32x16) " __DATE__; }

extern "C" int idf_timestep_embedding(const float* t, void* out, int B, int dim, int dtype, void* stream) {
  if (!t || !out || B <= 0 || dim <= 0 || (dim & 1)) return IDF_E_ARG;
  hipStream_t s = (hipStream_t)stream;
  const long long n = (long long)B * (dim / 2);
  if (dtype == IDF_BF16) hipLaunchKernelGGL(temb_kernel<IDF_BF16>, grid1d(n), dim3(256), 0, s, t, (unsigned short*)out, B, dim);
  else if (dtype == IDF_F16) hipLaunchKernelGGL(temb_kernel<IDF_F16>, grid1d(n), dim3(256), 0, s, t, (unsigned short*)out, B, dim);
  else return IDF_E_UNSUPPORTED;
  return idf_launch_status();
}

extern "C" int idf_unifusion_embed(const float* text, const float* loc, const float* tmask, const float* lmask,
                                   const float* null_text, const float* null_loc, const float* freqs, void* out,
                                   int rows, int text_dim, int D, int dtype, void* stream) {
  if (!text || !loc || !tmask || !lmask || !null_text || !null_loc || !freqs || !out || rows <= 0 || text_dim <= 0 || D <= 0) return IDF_E_ARG;
  hipStream_t s = (hipStream_t)stream;
  const long long n = (long long)rows * (text_dim + 32 * D);
  if (dtype == IDF_BF16)
    hipLaunchKernelGGL(unifusion_embed_kernel<IDF_BF16>, grid1d(n), dim3(256), 0, s, text, loc, tmask, lmask, null_text, null_loc, freqs, (unsigned short*)out, rows, text_dim, D);
  else if (dtype == IDF_F16)
    hipLaunchKernelGGL(unifusion_embed_kernel<IDF_F16>, grid1d(n), dim3(256), 0, s, text, loc, tmask, lmask, null_text, null_loc, freqs, (unsigned short*)out, rows, text_dim, D);
  else return IDF_E_UNSUPPORTED;
  return idf_launch_status();
}

extern "C" int idf_conv_in(const float* x_nchw, const float* w, const float* bias, void* out,
                           int B, int Cin, int H, int W, int Cout, int dtype, void* stream) {
  if (!x_nchw || !w || !bias || !out || B <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0 || (Cout % 8)) return IDF_E_ARG;
  if (!aligned16(out)) return IDF_E_ALIGN;
  hipStream_t s = (hipStream_t)stream;
  if (Cin == 4 && Cout == 320 && (dtype == IDF_BF16 || dtype == IDF_F16) && aligned16(bias)) {     // the UNet's first conv: matrix cores
    const void* fn = dtype == IDF_F16 ? (const void*)conv_in_mfma_kernel<IDF_F16> : (const void*)conv_in_mfma_kernel<IDF_BF16>;
    constexpr int smem_m = (320 + 256) * CIM_KP * 2;
    static std::atomic<unsigned long long> attr_m[2];
    if (const int e = idf_lds_optin(fn, smem_m, attr_m[dtype == IDF_F16 ? 1 : 0])) return e;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    const long long tiles = ((long long)B * H * W + 255) / 256;
    if (tiles < (1ll << 31)) {
      const dim3 grid((unsigned)(tiles < cus ? tiles : cus));
      if (dtype == IDF_BF16) hipLaunchKernelGGL(conv_in_mfma_kernel<IDF_BF16>, grid, dim3(512), smem_m, s, x_nchw, w, bias, (unsigned short*)out, B, H, W, (int)tiles);
      else hipLaunchKernelGGL(conv_in_mfma_kernel<IDF_F16>, grid, dim3(512), smem_m, s, x_nchw, w, bias, (unsigned short*)out, B, H, W, (int)tiles);
      return idf_launch_status();
    }
  }
  const size_t smem = (size_t)9 * Cin * Cout * sizeof(float);
  if (smem > 144 * 1024) return IDF_E_UNSUPPORTED;               // gfx950: 160 KB LDS per CU
  if (smem > 64 * 1024) {                                        // e.g. the VAE decoder's 4 -> 512 first conv (72 KB)
    static std::atomic<unsigned long long> attr_set[2];
    const void* fn = dtype == IDF_F16 ? (const void*)conv_in_kernel<IDF_F16> : (const void*)conv_in_kernel<IDF_BF16>;
    if (const int e = idf_lds_optin(fn, 144 * 1024, attr_set[dtype == IDF_F16 ? 1 : 0])) return e;
  }
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    n_cu = n;
  }
  const long long items = (long long)B * H * ((W + 3) / 4) * (Cout / 8);
  const long long need = (items + 255) / 256;
  dim3 grid((unsigned)(need < 2ll * n_cu ? need : 2ll * n_cu));   // persistent: the weights are staged once per workgroup
  if (dtype == IDF_BF16) hipLaunchKernelGGL(conv_in_kernel<IDF_BF16>, grid, dim3(256), smem, s, x_nchw, w, bias, (unsigned short*)out, B, Cin, H, W, Cout);
  else if (dtype == IDF_F16) hipLaunchKernelGGL(conv_in_kernel<IDF_F16>, grid, dim3(256), smem, s, x_nchw, w, bias, (unsigned short*)out, B, Cin, H, W, Cout);
  else return IDF_E_UNSUPPORTED;
  return idf_launch_status();
}

extern "C" int idf_cfg_combine(const float* eps_cond, const float* eps_uncond, float guidance, float* e_t, long long n, void* stream) {
  if (!eps_cond || !eps_uncond || !e_t || n <= 0) return IDF_E_ARG;
  hipLaunchKernelGGL(cfg_kernel, grid1d(n), dim3(256), 0, (hipStream_t)stream, eps_cond, eps_uncond, guidance, e_t, n);
  return idf_launch_status();
}

extern "C" int idf_plms_update(const float* x, const float* e_t, const float* e1, const float* e2, const float* e3,
                               const float* e_next, int mode, float a_t, float a_prev, float sqrt_1m_at,
                               float* x_out, long long n, void* stream) {
  if (!x || !e_t || !x_out || n <= 0 || mode < 0 || mode > 4) return IDF_E_ARG;
  if ((mode == 1 && !e_next) || (mode >= 2 && !e1) || (mode >= 3 && !e2) || (mode >= 4 && !e3)) return IDF_E_ARG;
  // float32 host arithmetic mirrors the reference's torch.full(...).sqrt() on float32 scalars (plms.py:130-144)
  const float sqrt_at = sqrtf(a_t), sqrt_aprev = sqrtf(a_prev), sqrt_1m_aprev = sqrtf(1.0f - a_prev);
  hipLaunchKernelGGL(plms_kernel, grid1d(n), dim3(256), 0, (hipStream_t)stream, x, e_t, e1, e2, e3, e_next, mode,
                     sqrt_at, sqrt_aprev, sqrt_1m_at, sqrt_1m_aprev, x_out, n);
  return idf_launch_status();
}

extern "C" int idf_mis_merge(const float* lat, const int* boxes, float* out, int n_inst, int B, int C, int H, int W,
                             int mode, void* stream) {
  if (!lat || !out || n_inst < 0 || B <= 0 || C <= 0 || H <= 0 || W <= 0 || (mode != 0 && mode != 1)) return IDF_E_ARG;
  if (mode == 1 && !boxes) return IDF_E_ARG;
  const long long per = (long long)B * C * H * W;
  hipLaunchKernelGGL(mis_merge_kernel, grid1d(per), dim3(256), 0, (hipStream_t)stream, lat, boxes, out, n_inst, B, C, H, W, mode);
  return idf_launch_status();
}

extern "C" int idf_cast_f32_to_16(const float* x, void* out, long long n, int dtype, void* stream) {
  if (!x || !out || n <= 0) return IDF_E_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == IDF_BF16) hipLaunchKernelGGL(cast_kernel<IDF_BF16>, grid1d(n), dim3(256), 0, s, x, (unsigned short*)out, n);
  else if (dtype == IDF_F16) hipLaunchKernelGGL(cast_kernel<IDF_F16>, grid1d(n), dim3(256), 0, s, x, (unsigned short*)out, n);
  else return IDF_E_UNSUPPORTED;
  return idf_launch_status();
}

extern "C" int idf_softmax_rows(const float* s, void* p, long long rows, int n, long long lds, long long ldp, float scale,
                                int dtype, void* stream) {
  if (!s || !p || rows <= 0 || n <= 0 || (n & 3) || lds < n || ldp < n || !(scale > 0.0f)) return IDF_E_ARG;
  if (rows > 0x7fffffffLL) return IDF_E_ARG;
  if ((lds & 3) || (ldp & 3) || !aligned16(s) || (((uintptr_t)p) & 7u)) return IDF_E_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  const float sl2 = scale * 1.4426950408889634f;
  if (dtype == IDF_BF16) hipLaunchKernelGGL(softmax_rows_kernel<IDF_BF16>, dim3((unsigned)rows), dim3(256), 0, st, s, (unsigned short*)p, n, lds, ldp, sl2);
  else if (dtype == IDF_F16) hipLaunchKernelGGL(softmax_rows_kernel<IDF_F16>, dim3((unsigned)rows), dim3(256), 0, st, s, (unsigned short*)p, n, lds, ldp, sl2);
  else return IDF_E_UNSUPPORTED;
  return idf_launch_status();
}

extern "C" int idf_pointwise_nchw(const float* x, const float* w, const float* bias, float* out, int B, int Cin, int Cout,
                                  long long HW, float in_scale, void* stream) {
  if (!x || !w || !out || B <= 0 || Cin <= 0 || Cin > 16 || Cout <= 0 || HW <= 0) return IDF_E_ARG;
  hipLaunchKernelGGL(pointwise_nchw_kernel, grid1d((long long)B * HW), dim3(256), 0, (hipStream_t)stream, x, w, bias, out, B, Cin,
                     Cout, HW, in_scale);
  return idf_launch_status();
}
