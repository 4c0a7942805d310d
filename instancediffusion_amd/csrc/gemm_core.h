// gemm_core.h -- pieces shared by the MFMA GEMM / implicit-GEMM conv translation units (gemm_conv.hip, gemm_big.hip):
// the launch parameter block, the K-tile depth and the 8-column epilogue.
#pragma once
#include "common.h"
#include <atomic>

namespace idfcore {

struct CoreParams {
  const unsigned short* W; int ldw; long long strideW; int N;
  const unsigned short* A; int lda; long long strideA; int M; int K;
  int Hin, Win, Cin, Ho, Wo, stride, up;           // conv gather
  void* out; int ldo; long long strideO;
  const float* bias; const unsigned short* rowbias; int ld_rowbias; int rows_per_batch;
  const unsigned short* res; int ldr; long long strideR;
  const float* gate; int epi; int n_valid;
  float* ws; size_t ws_bytes; int splitk; int kt_per_slice;   // split-K: fp32 partial slabs ws[slice][tail_rows][N]
  // hybrid split (persistent kernel): work items [0, full_items) are whole tiles (kt_full K-tiles, normal epilogue); the
  // remaining tiles -- rows [tail_m0, tail_m0 + tail_rows) -- are cut into `splitk` K-slices each.  Uniform split-K:
  // full_items = 0, tail_m0 = 0, tail_rows = M.
  int full_items; int kt_full; int tail_m0; int tail_rows;
  // LayerNorm folded into the GEMM (IDF_EPI_LN_ROW / IDF_EPI_LN_COL, include/idf.h): (mu, rstd) pairs of the normalised
  // operand's rows, the column sums c of the gamma-folded weight and (LN_COL) the beta term d
  const float* ln_stats; long long stride_ln_stats; const float* ln_c; const float* ln_d;
  // LN_ROW with ln_stats == nullptr: the kernel computes (mu, rstd) of A's rows itself (eps = ln_eps) and, when
  // ln_stats_out != nullptr, leaves them there ([M][2]) for an LN_COL consumer of the same matrix
  float ln_eps; float* ln_stats_out;
  // fused q | k | v projection: output columns n >= vt_col0 are stored transposed, vt_out[(n - vt_col0) * ld_vt + m]
  unsigned short* vt_out; int ld_vt; int vt_col0;
  // out_stats by-product of the persistent kernel: every wave leaves (mean, M2) of the BN/2 output columns it owns of a row
  // in stat_parts[(m * parts + tile_n * 2 + wn) * 2 ..] (fp32, workspace); stats_finalize merges the `parts` slots of a row
  float* stat_parts; int parts;
  // GroupNorm partials of the output as a by-product of the persistent kernel's conv epilogue (round 5): (mean, M2) per
  // (sample, 64-row chunk, group) in gn_partial[sample][gn_hw / 64][32][2]; gn_hw = rows per sample
  float* gn_partial; int gn_hw;
  // persistent-kernel schedule (round 4; every setting computes the same bits): tile_walk 0 = strided, 1 = chunked;
  // dephase = P start groups per XCD (0 / 1: off), dephase_units = one tile's estimated duration in units of 1024 cycles;
  // epi_vmcnt = 1: the first K-tile behind an epilogue waits with a counted vmcnt (the epilogue's stores drain under it)
  int tile_walk; int dephase; int dephase_units; int epi_vmcnt;
};

constexpr int BK = 64;

// Epilogue for 8 consecutive output columns n..n+7 of row m (n % 8 == 0).  Shared by the main kernel (after the
// accumulators were transposed through LDS so that a lane owns a contiguous 8-column run -> 16-B coalesced residual /
// rowbias loads and FULL-LINE 16-B stores) and by the split-K reducer.
template <int DT>
__device__ __forceinline__ void epilogue8(const CoreParams& p, int bz, int m, int n, float* v, float gate) {
  const int epi = p.epi;
  const bool full = (n + 7 < p.N);
  if (epi & IDF_EPI_LN_ROW) {                       // v = rstd_m * (acc - mu_m * c[n]); the beta term arrives as bias
    const f32x2 st = *reinterpret_cast<const f32x2*>(p.ln_stats + (size_t)bz * p.stride_ln_stats + 2 * (size_t)m);
#pragma unroll
    for (int e = 0; e < 8; ++e) if (n + e < p.N) v[e] = st[1] * fmaf(-st[0], p.ln_c[n + e], v[e]);
  }
  if (epi & IDF_EPI_LN_COL) {                       // v = rstd_n * (acc - c[m] * mu_n) + d[m]
    const float cm = p.ln_c[m], dm = p.ln_d[m];
    const float* st = p.ln_stats + (size_t)bz * p.stride_ln_stats + 2 * (size_t)n;
#pragma unroll
    for (int e = 0; e < 8; ++e) if (n + e < p.N) v[e] = fmaf(st[2 * e + 1], fmaf(-cm, st[2 * e], v[e]), dm);
  }
  if (epi & IDF_EPI_BIAS) {
    if (full) {
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.bias + n), b1 = *reinterpret_cast<const f32x4*>(p.bias + n + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[e] += b0[e]; v[e + 4] += b1[e]; }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) if (n + e < p.N) v[e] += p.bias[n + e];
    }
  }
  if (epi & IDF_EPI_ROWBIAS) {
    const unsigned short* rb = p.rowbias + (size_t)(m / p.rows_per_batch) * p.ld_rowbias + n;
    if (full && ((p.ld_rowbias & 7) == 0)) {
      float r[8];
      unpack8<DT>(*reinterpret_cast<const u32x4*>(rb), r);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += r[e];
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) if (n + e < p.N) v[e] += Elem<DT>::to_f32(rb[e]);
    }
  }
  if (epi & IDF_EPI_SILU) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = silu_f(v[e]);
  }
  if (epi & IDF_EPI_GELU) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = gelu_erf_f(v[e]);
  }
  if (epi & IDF_EPI_RES) {
    const unsigned short* rr = p.res + (size_t)bz * p.strideR + (size_t)m * p.ldr + n;
    const float gm = (epi & IDF_EPI_GATE) ? gate : 1.0f;
    if (full && ((p.ldr & 7) == 0)) {
      float r[8];
      unpack8<DT>(*reinterpret_cast<const u32x4*>(rr), r);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = fmaf(gm, v[e], r[e]);
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) if (n + e < p.N) v[e] = fmaf(gm, v[e], Elem<DT>::to_f32(rr[e]));
    }
  }
  if (epi & IDF_EPI_OUT_NCHW) {
    const int hw = p.Ho * p.Wo;
    const int bb = m / hw, rem = m - bb * hw;
    float* o = reinterpret_cast<float*>(p.out);
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (n + e < p.n_valid) o[((size_t)bb * p.n_valid + (n + e)) * hw + rem] = v[e];
  } else if (epi & IDF_EPI_OUT_F32) {
    float* o = reinterpret_cast<float*>(p.out) + (size_t)bz * p.strideO + (size_t)m * p.ldo + n;
    if (full && ((p.ldo & 3) == 0)) {
      *reinterpret_cast<f32x4*>(o) = f32x4{v[0], v[1], v[2], v[3]};
      *reinterpret_cast<f32x4*>(o + 4) = f32x4{v[4], v[5], v[6], v[7]};
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) if (n + e < p.N) o[e] = v[e];
    }
  } else {
    unsigned short* o = reinterpret_cast<unsigned short*>(p.out) + (size_t)bz * p.strideO + (size_t)m * p.ldo + n;
    if (full && ((p.ldo & 7) == 0)) {
      *reinterpret_cast<u32x4*>(o) = pack8<DT>(v);
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) if (n + e < p.N) o[e] = Elem<DT>::from_f32(v[e]);
    }
  }
}


}  // namespace idfcore

// Big-tile persistent kernel (gemm_big.hip).  Returns IDF_BIG_UNSUPPORTED when the shape does not qualify.
#define IDF_BIG_UNSUPPORTED (-100)
extern std::atomic<long long> idf_stat_big_launches;     // process-global launch counter (idf_get_stat)
// *splitk_out > 1 on return: the kernel left fp32 partials of that many K-slices in p.ws; the caller runs the reducer
// *parts_out (optional): > 0 when the kernel left per-wave partial output-row statistics in p.stat_parts (p.stat_parts
// offered and the launch was an unsplit dense GEMM), the number of slots per row.
// *gst_out (optional): 1 when the kernel left the GroupNorm partials of its output in p.gn_partial (see CoreParams).
int idf_launch_big(const idfcore::CoreParams& p, int dtype, bool conv, bool force, hipStream_t s, int* splitk_out,
                   int* parts_out = nullptr, int* tail_m0_out = nullptr, int* gst_out = nullptr);
int idf_launch_qkv320w(const idfcore::CoreParams& p, int dtype, hipStream_t s);   // qkv_fused.hip; IDF_BIG_UNSUPPORTED = not its shape
int idf_launch_geglu640w(const idfcore::CoreParams& p, int dtype, hipStream_t s);   // geglu_fused.hip; IDF_BIG_UNSUPPORTED = not its shape
int idf_gegluw_set_mode(int v);                         // 0 = never, 1 = when the shape qualifies; returns the previous mode
int idf_launch_qkv640w(const idfcore::CoreParams& p, int dtype, hipStream_t s);   // qkv640_fused.hip
int idf_qkv640w_set_mode(int v);
int idf_qkvw_set_mode(int v);                           // 0 = never, 1 = when the shape qualifies; returns the previous mode
int idf_mlp_set_mode(int v);                            // mlp_fused.hip: 0 = mlp320_kernel, 1 = mlp320w_kernel; returns the previous mode
int idf_big_min_eff_pct(int set);                        // automatic rule's occupancy bar in per cent (set < 0: query)
int idf_num_cu();                                        // CUs of the current device (cached)
// (mu, rstd) per row from `parts` equal-count (mean, M2) slots per row (fixed merge order): out_stats[m] = f32x2
int idf_stats_finalize(const float* stat_parts, int parts, int cols_per_part, float* out_stats, int M, float eps, hipStream_t s);
