// gemm_conv.hip -- MFMA GEMM core for gfx950 shared by idf_gemm (linear / 1x1 conv) and idf_conv3x3 (implicit
// GEMM with an on-the-fly NHWC im2col gather, optional stride 2 and folded nearest-x2 upsample).
//
// Design (MI355X): 256-thread workgroup = 4 wave64; v_mfma_f32_32x32x16_{bf16,f16}; block tile BM x BN x 64,
// double-buffered LDS (row stride 144 B = odd number of 16-B slots -> conflict-free ds_read_b128 fragment reads),
// register-staged global->LDS with the next tile's loads in flight during the MFMAs, ONE barrier per K-tile.
// The MFMA "A" operand is the WEIGHT tile and "B" the activation tile, so every lane ends up owning 4 consecutive
// output columns of one output row: epilogue = 8-byte packed stores, float4 bias loads, in-register GEGLU.
//
// Three K-loop variants share the tile epilogue: 1 = register-staged (conv gather), 2 = LDS-DMA double buffer, 3 = LDS-DMA ring with
// its K-tiles in flight from the first instruction (the latency kernel of the small-batch launches, round 4); the persistent
// 256 x {320,256,128} kernel that takes every launch big enough to fill the chip is gemm_big.hip.
//
// Roofline: MFMA-bound (2.5 PFLOP/s dense bf16); algorithmic flops = 2*M*N*K.
#include "gemm_core.h"
#include "attn_core.h"
#include <cstdlib>

using namespace idfcore;

namespace {

#ifndef IDF_GEMM_BIG_DEFAULT
#define IDF_GEMM_BIG_DEFAULT 1
#endif
constexpr int LSTR = 72;   // LDS row stride in elements (144 B)

// Tile epilogue shared by both K-loop variants (see the comment inside).
template <int DT, int BM, int BN, int WM, int WN, int KLOOP_LDS_BYTES>
__device__ __forceinline__ void tile_epilogue(const CoreParams& p, f32x16 (&acc)[WN / 32][WM / 32], unsigned short* smem,
                                              int m0, int n0, int bz, int wm, int wn, int lane, int wave) {
  constexpr int TM = WM / 32, TN = WN / 32;
  const int l31 = lane & 31, hi = lane >> 5;
  // ---- epilogue.  acc[a][b][r] holds D[n = a*32 + (r&3) + 8*(r>>2) + 4*hi][m = b*32 + l31] of the wave's WM x WN tile.
  // Transpose it through LDS (the K-loop tiles are dead now) so that each lane owns 8 CONSECUTIVE columns of one row:
  // residual / rowbias loads become coalesced 16-B vectors and the output leaves as full 128-B lines (8 lanes x 16 B)
  // instead of 16-B fragments scattered over 32 rows per store instruction.
  constexpr int CSTR = WN + 4;                                   // fp32 row stride: rows 16-B aligned, bank-spread
  static_assert(4 * WM * CSTR * 4 <= KLOOP_LDS_BYTES, "C staging must fit in the allocated LDS");
  __syncthreads();
  float* Cl = reinterpret_cast<float*>(smem) + wave * (WM * CSTR);
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<f32x4*>(Cl + (b * 32 + l31) * CSTR + a * 32 + 8 * q + 4 * hi) =
            f32x4{acc[a][b][4 * q], acc[a][b][4 * q + 1], acc[a][b][4 * q + 2], acc[a][b][4 * q + 3]};
  // No block barrier here: each wave reads back only ITS OWN staging region and a wave's LDS instructions execute in
  // program order; the wave-level barrier only stops the COMPILER from hoisting the loads above the stores.
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

  const int epi = p.epi;
  const float gate = (epi & IDF_EPI_GATE) ? p.gate[0] : 0.0f;
  const int mw = m0 + wm * WM, nw = n0 + wn * WN;
  if (epi & IDF_EPI_GEGLU) {
    if constexpr (WN >= 64) {
      // wave columns are [P/2 value | P/2 gate] per P (P = 64, or 32 with IDF_EPI_GEGLU_P32); a lane pairs value chunk c with
      // the gate chunk of the same outputs of one row
      constexpr int CHV = WN / 16;                               // value chunks (8 columns each) per row
      constexpr int RPP = 64 / CHV;                              // rows per pass
      const int P = (epi & IDF_EPI_GEGLU_P32) ? 32 : 64, half = P >> 1, cpg = P >> 4;   // value chunks per group
#pragma unroll
      for (int pass = 0; pass < WM / RPP; ++pass) {
        const int row = pass * RPP + lane / CHV, c = lane % CHV;
        const int grp = c / cpg, cc = c - grp * cpg;
        const int m = mw + row;
        const int npk = nw + grp * P + cc * 8;                   // packed weight row of the value columns
        if (m < p.M && npk + half < p.N) {
          const float* src = Cl + row * CSTR + grp * P + cc * 8;
          float v[8], g[8];
          *reinterpret_cast<f32x4*>(v) = *reinterpret_cast<const f32x4*>(src);
          *reinterpret_cast<f32x4*>(v + 4) = *reinterpret_cast<const f32x4*>(src + 4);
          *reinterpret_cast<f32x4*>(g) = *reinterpret_cast<const f32x4*>(src + half);
          *reinterpret_cast<f32x4*>(g + 4) = *reinterpret_cast<const f32x4*>(src + half + 4);
          const f32x4 bv0 = *reinterpret_cast<const f32x4*>(p.bias + npk), bv1 = *reinterpret_cast<const f32x4*>(p.bias + npk + 4);
          const f32x4 bg0 = *reinterpret_cast<const f32x4*>(p.bias + npk + half), bg1 = *reinterpret_cast<const f32x4*>(p.bias + npk + half + 4);
          if (epi & IDF_EPI_LN_ROW) {                            // LayerNorm folded in (gemm_core.h epilogue8 has the plain form)
            const f32x2 st = *reinterpret_cast<const f32x2*>(p.ln_stats + (size_t)bz * p.stride_ln_stats + 2 * (size_t)m);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              v[e] = st[1] * fmaf(-st[0], p.ln_c[npk + e], v[e]);
              g[e] = st[1] * fmaf(-st[0], p.ln_c[npk + half + e], g[e]);
            }
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[e] = (v[e] + bv0[e]) * gelu_erf_f(g[e] + bg0[e]);
            v[e + 4] = (v[e + 4] + bv1[e]) * gelu_erf_f(g[e + 4] + bg1[e]);
          }
          const int j = (nw + grp * P) / 2 + cc * 8;
          unsigned short* o = reinterpret_cast<unsigned short*>(p.out) + (size_t)bz * p.strideO + (size_t)m * p.ldo + j;
          if ((p.ldo & 7) == 0) {
            *reinterpret_cast<u32x4*>(o) = pack8<DT>(v);
          } else {
            const u32x4 pk = pack8<DT>(v);
            *reinterpret_cast<u32x2*>(o) = u32x2{pk[0], pk[1]};
            *reinterpret_cast<u32x2*>(o + 4) = u32x2{pk[2], pk[3]};
          }
        }
      }
    }
    return;
  }
  constexpr int CH = WN / 8;                                     // 8-column chunks per row
  constexpr int RPP = 64 / CH;                                   // rows per pass (one wave instruction = RPP full rows)
#pragma unroll
  for (int pass = 0; pass < WM / RPP; ++pass) {
    const int row = pass * RPP + lane / CH, c = lane % CH;
    const int m = mw + row, n = nw + c * 8;
    if (m >= p.M || n >= p.N) continue;
    const float* src = Cl + row * CSTR + c * 8;
    float v[8];
    *reinterpret_cast<f32x4*>(v) = *reinterpret_cast<const f32x4*>(src);
    *reinterpret_cast<f32x4*>(v + 4) = *reinterpret_cast<const f32x4*>(src + 4);
    if (p.splitk > 1) {
      float* o = p.ws + ((size_t)blockIdx.z * p.M + m) * p.N + n;
      if (n + 7 < p.N && ((p.N & 3) == 0)) {
        *reinterpret_cast<f32x4*>(o) = f32x4{v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4*>(o + 4) = f32x4{v[4], v[5], v[6], v[7]};
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) if (n + e < p.N) o[e] = v[e];
      }
    } else {
      epilogue8<DT>(p, bz, m, n, v, gate);
    }
  }
}

template <int DT, int BM, int BN, int WM, int WN, bool CONV>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const CoreParams p) {
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int WAVES_N = BN / WN;
  constexpr int AR = BM / 32, WR = BN / 32;          // rows staged per thread
  static_assert((BM / WM) * (BN / WN) == 4, "4 waves per workgroup");
  extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
  constexpr int BUF = (BM + BN) * LSTR;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  const int wn = wave % WAVES_N, wm = wave / WAVES_N;
  // XCD-aware tile order: workgroup L runs on XCD L % 8 (8 private L2s); give each XCD a CONTIGUOUS chunk of the
  // n-fastest tile list so the n-tiles that share one activation m-tile hit the same L2 (bijective for any count).
  int tile;
  {
    const int T = gridDim.x, L = blockIdx.x;
    const int q = T >> 3, r = T & 7, xcd = L & 7, i = L >> 3;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
  }
  const int tiles_n = (p.N + BN - 1) / BN;
  const int m_tile = tile / tiles_n;
  const int n0 = (tile - m_tile * tiles_n) * BN, m0 = m_tile * BM;
  const int bz = (p.splitk > 1) ? 0 : blockIdx.z;            // split-K launches are never batched

  const unsigned short* Wb = p.W + (size_t)bz * p.strideW;
  const unsigned short* Ab = p.A + (size_t)bz * p.strideA;

  // ---- staging roles: thread -> (16-B chunk c along k, rows r0 + 32 i)
  const int c = tid & 7, r0 = tid >> 3;
  const unsigned short* wrow[WR];
#pragma unroll
  for (int i = 0; i < WR; ++i) {
    int n = min(n0 + r0 + 32 * i, p.N - 1);
    wrow[i] = Wb + (size_t)n * p.ldw + c * 8;
  }
  const unsigned short* arow[AR];
  int ay[AR], ax[AR];                                  // conv: output pixel coords (pre-multiplied by stride, -1)
#pragma unroll
  for (int i = 0; i < AR; ++i) {
    int m = min(m0 + r0 + 32 * i, p.M - 1);
    if (CONV) {
      int hw = p.Ho * p.Wo;
      int b = m / hw, rem = m - b * hw;
      int yo = rem / p.Wo, xo = rem - yo * p.Wo;
      ay[i] = yo * p.stride - 1;
      ax[i] = xo * p.stride - 1;
      arow[i] = Ab + (size_t)b * p.Hin * p.Win * p.lda + c * 8;
    } else {
      ay[i] = ax[i] = 0;
      arow[i] = Ab + (size_t)m * p.lda + c * 8;
    }
  }
  const int nk_all = p.K / BK;
  const int kt_begin = (p.splitk > 1) ? blockIdx.z * p.kt_per_slice : 0;
  const int nk = (p.splitk > 1) ? min(p.kt_per_slice, nk_all - kt_begin) : nk_all;   // K-tiles of this block
  int tap = 0, ci0 = 0;                                // conv: K-tile -> (3x3 tap, channel offset)
  if (CONV) { tap = (kt_begin * BK) / p.Cin; ci0 = kt_begin * BK - tap * p.Cin; }

  // two register sets: tile kt+1 waits in one while tile kt+2 is being fetched into the other (2-deep prefetch,
  // so a global load has two K-tile iterations of MFMA time to land before its ds_write needs it)
  u32x4 raA[AR], rwA[WR], raB[AR], rwB[WR];
  unsigned okA = 0, okB = 0;                           // conv: per-row 'tap inside the image' bits of the staged tile
  auto load_tile = [&](int kt, u32x4* ra, u32x4* rw, unsigned& okbits) {
#pragma unroll
    for (int i = 0; i < WR; ++i) rw[i] = *reinterpret_cast<const u32x4*>(wrow[i] + (size_t)(kt_begin + kt) * BK);
    if (CONV) {
      const int ky = tap / 3, kx = tap - ky * 3;
      const int Hup = p.Hin << p.up, Wup = p.Win << p.up;
      unsigned bits = 0;
#pragma unroll
      for (int i = 0; i < AR; ++i) {
        const int yi = ay[i] + ky, xi = ax[i] + kx;
        const bool ok = (yi >= 0) & (yi < Hup) & (xi >= 0) & (xi < Wup);
        // UNCONDITIONAL load from a clamped (always valid) pixel; the padding taps are zeroed at ds_write time
        // (store_tile) so the number of outstanding loads is static and the counted vmcnt waits stay exact.
        const int ys = min(max(yi, 0), Hup - 1) >> p.up, xs = min(max(xi, 0), Wup - 1) >> p.up;
        ra[i] = *reinterpret_cast<const u32x4*>(arow[i] + ((size_t)ys * p.Win + xs) * p.lda + ci0);
        bits |= (ok ? 1u : 0u) << i;
      }
      okbits = bits;
      ci0 += BK;
      if (ci0 >= p.Cin) { ci0 = 0; ++tap; }
    } else {
#pragma unroll
      for (int i = 0; i < AR; ++i) ra[i] = *reinterpret_cast<const u32x4*>(arow[i] + (size_t)(kt_begin + kt) * BK);
    }
  };
  auto store_tile = [&](int buf, const u32x4* ra, const u32x4* rw, unsigned okbits) {
    unsigned short* Al = smem + buf * BUF;
    unsigned short* Wl = Al + BM * LSTR;
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      u32x4 v = ra[i];
      if (CONV) {
        const unsigned keep = ((okbits >> i) & 1u) ? 0xffffffffu : 0u;
        v[0] &= keep; v[1] &= keep; v[2] &= keep; v[3] &= keep;
      }
      *reinterpret_cast<u32x4*>(Al + (r0 + 32 * i) * LSTR + c * 8) = v;
    }
#pragma unroll
    for (int i = 0; i < WR; ++i) *reinterpret_cast<u32x4*>(Wl + (r0 + 32 * i) * LSTR + c * 8) = rw[i];
  };

  f32x16 acc[TN][TM];
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

  auto compute = [&](int buf) {
    const unsigned short* Al = smem + buf * BUF;
    const unsigned short* Wl = Al + BM * LSTR;
    const unsigned short* af_base = Al + (wm * WM + l31) * LSTR + hi * 8;
    const unsigned short* wf_base = Wl + (wn * WN + l31) * LSTR + hi * 8;
    // fragment reads of K-step ks+1 are issued BEFORE the MFMAs of K-step ks (register double-buffer), so the LDS
    // latency (~100+ cycles) hides under 4-8 MFMAs instead of stalling every K-step (the compiler does not do this).
    u32x4 wf[2][TN], af[2][TM];
#pragma unroll
    for (int a = 0; a < TN; ++a) wf[0][a] = *reinterpret_cast<const u32x4*>(wf_base + a * 32 * LSTR);
#pragma unroll
    for (int b = 0; b < TM; ++b) af[0][b] = *reinterpret_cast<const u32x4*>(af_base + b * 32 * LSTR);
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      const int cur = ks & 1, nxt = cur ^ 1;
      if (ks + 1 < BK / 16) {
#pragma unroll
        for (int a = 0; a < TN; ++a) wf[nxt][a] = *reinterpret_cast<const u32x4*>(wf_base + a * 32 * LSTR + (ks + 1) * 16);
#pragma unroll
        for (int b = 0; b < TM; ++b) af[nxt][b] = *reinterpret_cast<const u32x4*>(af_base + b * 32 * LSTR + (ks + 1) * 16);
      }
#pragma unroll
      for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b) acc[a][b] = Elem<DT>::mfma32(wf[cur][a], af[cur][b], acc[a][b]);
    }
  };

  load_tile(0, raA, rwA, okA);
  store_tile(0, raA, rwA, okA);
  if (nk > 1) load_tile(1, raA, rwA, okA);
  __syncthreads();
  // Invariant at the top of each pair of steps: LDS buffer 0 holds tile kt, register set A holds tile kt+1 (in flight).
  // The steady-state loop issues its loads UNCONDITIONALLY so the compiler's counted vmcnt waits leave the newest
  // tile's loads in flight across the ds_write + barrier; the <= 3 remaining tiles are peeled below.
  int kt = 0;
  for (; kt + 3 < nk; kt += 2) {
    load_tile(kt + 2, raB, rwB, okB);
    __builtin_amdgcn_sched_barrier(0);         // keep the loads issued BEFORE the MFMAs (hipcc otherwise sinks them)
    compute(0);
    store_tile(1, raA, rwA, okA);
    __syncthreads();
    load_tile(kt + 3, raA, rwA, okA);
    __builtin_amdgcn_sched_barrier(0);
    compute(1);
    store_tile(0, raB, rwB, okB);
    __syncthreads();
  }
  const int rem = nk - kt;                     // 1, 2 or 3 tiles left
  if (rem == 3) {
    load_tile(kt + 2, raB, rwB, okB);
    compute(0);
    store_tile(1, raA, rwA, okA);
    __syncthreads();
    compute(1);
    store_tile(0, raB, rwB, okB);
    __syncthreads();
    compute(0);
  } else if (rem == 2) {
    compute(0);
    store_tile(1, raA, rwA, okA);
    __syncthreads();
    compute(1);
  } else {
    compute(0);
  }

  tile_epilogue<DT, BM, BN, WM, WN, 2 * (BM + BN) * LSTR * 2>(p, acc, smem, m0, n0, bz, wm, wn, lane, wave);
}

// ------------------------------------------------------------------------------------------------------------------
// K-loop variant 2: LDS-DMA staging (global_load_lds_dwordx4) for every DENSE operand tile (the weight tile always, the
// activation tile unless CONV).  The VGPR -> LDS store path (ds_write_b128 ~79 B/clk/CU) was the busiest pipe of
// variant 1 -- 64 KB of tile stores per CU per K-tile pair against 1024 MFMA cycles; the DMA path bypasses it and frees
// the staging registers.  An LDS-DMA instruction writes wave-uniform base + lane*16 B, so the tile image is LINEAR
// 128-B rows (64 elements); bank conflicts of the ds_read_b128 fragment reads are removed by an XOR swizzle applied on
// the per-lane GLOBAL source address and, identically, on the reads:  LDS slot = k-chunk ^ ((row >> 1) & 7).
// The conv activation gather keeps the register path (padding taps must be zeroed) and stores into the same image.
// ------------------------------------------------------------------------------------------------------------------
template <int DT, int BM, int BN, int WM, int WN, bool CONV>
__global__ __launch_bounds__(256, 2) void gemm_kernel_dma(const CoreParams p) {
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int WAVES_N = BN / WN;
  constexpr int RS = 64;                                // LDS row stride (elements) = BK, no padding
  constexpr int BUF = (BM + BN) * RS;
  constexpr int W_INST = BN / 8 / 4, A_INST = BM / 8 / 4;     // DMA instructions (8 rows each) per wave per tile
  constexpr int AR = BM / 32;
  constexpr int KLOOP_LDS = 2 * BUF * 2;
  constexpr int CSTAGE = 4 * WM * (WN + 4) * 4;
  constexpr int LDS_BYTES = KLOOP_LDS > CSTAGE ? KLOOP_LDS : CSTAGE;
  static_assert((BM / WM) * (BN / WN) == 4, "4 waves per workgroup");
  extern __shared__ __attribute__((aligned(16))) unsigned short smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int wn = wave % WAVES_N, wm = wave / WAVES_N;
  int tile;
  {
    const int T = gridDim.x, L = blockIdx.x;
    const int q = T >> 3, r = T & 7, xcd = L & 7, i = L >> 3;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
  }
  const int tiles_n = (p.N + BN - 1) / BN;
  const int m_tile = tile / tiles_n;
  const int n0 = (tile - m_tile * tiles_n) * BN, m0 = m_tile * BM;
  const int bz = (p.splitk > 1) ? 0 : blockIdx.z;
  const unsigned short* Wb = p.W + (size_t)bz * p.strideW;
  const unsigned short* Ab = p.A + (size_t)bz * p.strideA;

  const int nk_all = p.K / BK;
  const int kt_begin = (p.splitk > 1) ? blockIdx.z * p.kt_per_slice : 0;
  const int nk = (p.splitk > 1) ? min(p.kt_per_slice, nk_all - kt_begin) : nk_all;

  // ---- DMA roles: instruction j of this wave covers tile rows 8*(wave + 4 j) .. +7; lane -> (row r = lane>>3, slot c)
  const int dr = lane >> 3, dc = lane & 7;
  const unsigned short* wsrc[W_INST];
#pragma unroll
  for (int j = 0; j < W_INST; ++j) {
    const int row = 8 * (wave + 4 * j) + dr;                     // tile-local row
    const int n = min(n0 + row, p.N - 1);
    wsrc[j] = Wb + (size_t)n * p.ldw + (size_t)kt_begin * BK + ((dc ^ ((row >> 1) & 7)) * 8);
  }
  const unsigned short* asrc[A_INST];
  if (!CONV) {
#pragma unroll
    for (int j = 0; j < A_INST; ++j) {
      const int row = 8 * (wave + 4 * j) + dr;
      const int m = min(m0 + row, p.M - 1);
      asrc[j] = Ab + (size_t)m * p.lda + (size_t)kt_begin * BK + ((dc ^ ((row >> 1) & 7)) * 8);
    }
  }
  // ---- conv activation gather (register path): thread -> (16-B chunk c, rows r0 + 32 i)
  const int c = tid & 7, r0 = tid >> 3;
  const unsigned short* arow[AR];
  int ay[AR], ax[AR];
  int tap = 0, ci0 = 0;
  if (CONV) {
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      const int m = min(m0 + r0 + 32 * i, p.M - 1);
      const int hw = p.Ho * p.Wo;
      const int b = m / hw, rem = m - b * hw;
      const int yo = rem / p.Wo, xo = rem - yo * p.Wo;
      ay[i] = yo * p.stride - 1;
      ax[i] = xo * p.stride - 1;
      arow[i] = Ab + (size_t)b * p.Hin * p.Win * p.lda + c * 8;
    }
    tap = (kt_begin * BK) / p.Cin;
    ci0 = kt_begin * BK - tap * p.Cin;
  }
  const int a_sw = (c ^ ((r0 >> 1) & 7)) * 8;            // (row>>1)&7 == (r0>>1)&7 for rows r0 + 32 i
  u32x4 ra[AR];
  unsigned okbits = 0;

  auto dma_tile = [&](int kt, int buf) {                // enqueue the LDS-DMA of K-tile kt into LDS buffer buf
    unsigned short* Al = smem + buf * BUF;
    unsigned short* Wl = Al + BM * RS;
#pragma unroll
    for (int j = 0; j < W_INST; ++j)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc[j] + (size_t)kt * BK),
                                       (__attribute__((address_space(3))) void*)(Wl + 8 * (wave + 4 * j) * RS), 16, 0, 0);
    if (!CONV) {
#pragma unroll
      for (int j = 0; j < A_INST; ++j)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(asrc[j] + (size_t)kt * BK),
                                         (__attribute__((address_space(3))) void*)(Al + 8 * (wave + 4 * j) * RS), 16, 0, 0);
    }
  };
  auto gather_tile = [&]() {                            // conv: issue the activation gather of the next K-tile
    const int ky = tap / 3, kx = tap - ky * 3;
    const int Hup = p.Hin << p.up, Wup = p.Win << p.up;
    unsigned bits = 0;
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      const int yi = ay[i] + ky, xi = ax[i] + kx;
      const bool ok = (yi >= 0) & (yi < Hup) & (xi >= 0) & (xi < Wup);
      const int ys = min(max(yi, 0), Hup - 1) >> p.up, xs = min(max(xi, 0), Wup - 1) >> p.up;
      ra[i] = *reinterpret_cast<const u32x4*>(arow[i] + ((size_t)ys * p.Win + xs) * p.lda + ci0);
      bits |= (ok ? 1u : 0u) << i;
    }
    okbits = bits;
    ci0 += BK;
    if (ci0 >= p.Cin) { ci0 = 0; ++tap; }
  };
  auto scatter_tile = [&](int buf) {                    // conv: zero the padding taps and store into the swizzled image
    unsigned short* Al = smem + buf * BUF;
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      u32x4 v = ra[i];
      const unsigned keep = ((okbits >> i) & 1u) ? 0xffffffffu : 0u;
      v[0] &= keep; v[1] &= keep; v[2] &= keep; v[3] &= keep;
      *reinterpret_cast<u32x4*>(Al + (r0 + 32 * i) * RS + a_sw) = v;
    }
  };

  f32x16 acc[TN][TM];
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

  const int f_sw = (l31 >> 1) & 7;                      // read-side swizzle: fragment rows are (multiple of 32) + l31
  auto compute = [&](int buf) {
    const unsigned short* Al = smem + buf * BUF;
    const unsigned short* Wl = Al + BM * RS;
    const unsigned short* af_base = Al + (wm * WM + l31) * RS;
    const unsigned short* wf_base = Wl + (wn * WN + l31) * RS;
    u32x4 wf[2][TN], af[2][TM];                         // register double-buffered fragments (see variant 1)
    {
      const int slot = (hi ^ f_sw) * 8;
#pragma unroll
      for (int a = 0; a < TN; ++a) wf[0][a] = *reinterpret_cast<const u32x4*>(wf_base + a * 32 * RS + slot);
#pragma unroll
      for (int b = 0; b < TM; ++b) af[0][b] = *reinterpret_cast<const u32x4*>(af_base + b * 32 * RS + slot);
    }
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      const int cur = ks & 1, nxt = cur ^ 1;
      if (ks + 1 < BK / 16) {
        const int slot = (((ks + 1) * 2 + hi) ^ f_sw) * 8;
#pragma unroll
        for (int a = 0; a < TN; ++a) wf[nxt][a] = *reinterpret_cast<const u32x4*>(wf_base + a * 32 * RS + slot);
#pragma unroll
        for (int b = 0; b < TM; ++b) af[nxt][b] = *reinterpret_cast<const u32x4*>(af_base + b * 32 * RS + slot);
      }
#pragma unroll
      for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b) acc[a][b] = Elem<DT>::mfma32(wf[cur][a], af[cur][b], acc[a][b]);
    }
  };

  dma_tile(0, 0);
  if (CONV) { gather_tile(); scatter_tile(0); }
  __syncthreads();                                      // drains vmcnt: tile 0 landed
  for (int kt = 0; kt < nk; ++kt) {
    const bool more = (kt + 1 < nk);
    if (more) {
      dma_tile(kt + 1, (kt + 1) & 1);                   // lands during the MFMAs of tile kt
      if (CONV) gather_tile();
    }
    compute(kt & 1);
    if (CONV && more) scatter_tile((kt + 1) & 1);
    __syncthreads();                                    // (kt+1) complete and visible; buffer kt&1 free for tile kt+2
  }
  tile_epilogue<DT, BM, BN, WM, WN, LDS_BYTES>(p, acc, smem, m0, n0, bz, wm, wn, lane, wave);
}

// ---- K-loop variant 3 (round 4): the LATENCY kernel.  The launches the persistent kernel declines (a 2-row forward is ~250
// GEMMs of 1-2 GFLOP and ~50 convs) spent their time in a serial chain: variant 2 keeps ONE K-tile in flight, so a K = 640
// launch is ten back-to-back trips to L2 / HBM (~2 us each, cold weights) around 0.2 us of MFMA work -- 17-22 us per launch,
// whatever the shape (profiles/r03_shape_profile_B2.log).  Same tile, LDS image, swizzle, fragment reads and epilogue as
// variant 2, but an NS-stage ring with NS - 1 K-tiles in flight from the first instruction (all of K = 320), counted
// s_waitcnt vmcnt instead of a drain per tile, and the conv activation tile arrives by LDS-DMA too (a lane whose tap is
// padding reads a zero page, like the persistent kernel's loader).  LDS-DMA is inline assembly: with the builtin the
// compiler orders every later ds_read behind ALL outstanding DMA.  One workgroup per CU (96-144 KB of LDS): meant for launches
// of at most a few hundred tiles -- the dispatcher's rule is in launch().
__device__ __attribute__((aligned(128))) unsigned short ring_zero_page[64];   // zero-initialised device memory

__device__ __forceinline__ void ring_dma16(const void* addr /* per lane */, const void* lds_lane0) {
  const unsigned l = __builtin_amdgcn_readfirstlane((unsigned)(size_t)lds_lane0);
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(l), "v"(addr) : "memory");
}
// at most `ahead` K-tiles (PT LDS-DMA instructions of this wave each) may still be in flight behind the tile waited for
template <int PT, int MAXA> __device__ __forceinline__ void ring_wait(int ahead) {
  if constexpr (MAXA <= 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    static_assert(MAXA * PT < 64, "vmcnt is a 6-bit field");
    if (ahead >= MAXA) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MAXA * PT) : "memory");
    else ring_wait<PT, MAXA - 1>(ahead);
  }
}

template <int DT, int BM, int BN, int WM, int WN, bool CONV, int NS>
__global__ __launch_bounds__(256, 1) void gemm_kernel_ring(const CoreParams p) {
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int WAVES_N = BN / WN;
  constexpr int RS = 64;                                // LDS row stride (elements) = BK, no padding
  constexpr int BUF = (BM + BN) * RS;                   // elements per ring stage
  constexpr int W_INST = BN / 8 / 4, A_INST = BM / 8 / 4;     // DMA instructions (8 rows each) per wave per tile
  constexpr int PER_TILE = W_INST + A_INST;
  constexpr int KLOOP_LDS = NS * BUF * 2;
  constexpr int CSTAGE = 4 * WM * (WN + 4) * 4;
  constexpr int LDS_BYTES = KLOOP_LDS > CSTAGE ? KLOOP_LDS : CSTAGE;
  static_assert((BM / WM) * (BN / WN) == 4, "4 waves per workgroup");
  static_assert(NS >= 3 && LDS_BYTES <= 160 * 1024, "ring must fit the CU's LDS");
  extern __shared__ __attribute__((aligned(16))) unsigned short smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int wn = wave % WAVES_N, wm = wave / WAVES_N;
  int tile;
  {
    const int T = gridDim.x, L = blockIdx.x;
    const int q = T >> 3, r = T & 7, xcd = L & 7, i = L >> 3;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
  }
  const int tiles_n = (p.N + BN - 1) / BN;
  const int m_tile = tile / tiles_n;
  const int n0 = (tile - m_tile * tiles_n) * BN, m0 = m_tile * BM;
  const int bz = (p.splitk > 1) ? 0 : blockIdx.z;
  const unsigned short* Wb = p.W + (size_t)bz * p.strideW;
  const unsigned short* Ab = p.A + (size_t)bz * p.strideA;

  const int nk_all = p.K / BK;
  const int kt_begin = (p.splitk > 1) ? blockIdx.z * p.kt_per_slice : 0;
  const int nk = (p.splitk > 1) ? min(p.kt_per_slice, nk_all - kt_begin) : nk_all;

  // ---- DMA roles: instruction j of this wave covers tile rows 8*(wave + 4 j) .. +7; lane -> (row r = lane>>3, slot c);
  // the 16-B slot of a row is XOR-swizzled on the SOURCE side (slot c of LDS row `row` holds global chunk c ^ ((row>>1)&7))
  const int dr = lane >> 3, dc = lane & 7;
  const unsigned short* wsrc[W_INST];
#pragma unroll
  for (int j = 0; j < W_INST; ++j) {
    const int row = 8 * (wave + 4 * j) + dr;                     // tile-local row
    const int n = min(n0 + row, p.N - 1);
    wsrc[j] = Wb + (size_t)n * p.ldw + (size_t)kt_begin * BK + ((dc ^ ((row >> 1) & 7)) * 8);
  }
  const unsigned short* asrc[A_INST];                            // dense: row pointer at kt_begin; conv: image base + chunk
  int ay[A_INST], ax[A_INST];
#pragma unroll
  for (int j = 0; j < A_INST; ++j) {
    const int row = 8 * (wave + 4 * j) + dr;
    const int m = min(m0 + row, p.M - 1);
    const int ch = (dc ^ ((row >> 1) & 7)) * 8;
    if (!CONV) {
      asrc[j] = Ab + (size_t)m * p.lda + (size_t)kt_begin * BK + ch;
      ay[j] = ax[j] = 0;
    } else {
      const int hw = p.Ho * p.Wo;
      const int b = m / hw, rem = m - b * hw;
      const int yo = rem / p.Wo, xo = rem - yo * p.Wo;
      ay[j] = yo * p.stride - 1;
      ax[j] = xo * p.stride - 1;
      asrc[j] = Ab + (size_t)b * p.Hin * p.Win * p.lda + ch;
    }
  }
  int tap = 0, ci0 = 0;                                          // conv: tap / first input channel of the next K-tile to issue
  if (CONV) { tap = (kt_begin * BK) / p.Cin; ci0 = kt_begin * BK - tap * p.Cin; }
  int kt_issue = 0, st_issue = 0;                                // next K-tile to enqueue and the ring stage it goes to

  auto issue_tile = [&]() {
    unsigned short* Al = smem + st_issue * BUF;
    unsigned short* Wl = Al + BM * RS;
#pragma unroll
    for (int j = 0; j < W_INST; ++j) ring_dma16(wsrc[j] + (size_t)kt_issue * BK, Wl + 8 * (wave + 4 * j) * RS);
    if (!CONV) {
#pragma unroll
      for (int j = 0; j < A_INST; ++j) ring_dma16(asrc[j] + (size_t)kt_issue * BK, Al + 8 * (wave + 4 * j) * RS);
    } else {
      const int ky = tap / 3, kx = tap - ky * 3;
      const int Hup = p.Hin << p.up, Wup = p.Win << p.up;
#pragma unroll
      for (int j = 0; j < A_INST; ++j) {
        const int yi = ay[j] + ky, xi = ax[j] + kx;
        const bool ok = (yi >= 0) & (yi < Hup) & (xi >= 0) & (xi < Wup);
        const int ys = yi >> p.up, xs = xi >> p.up;
        const unsigned short* src = ok ? asrc[j] + ((size_t)ys * p.Win + xs) * p.lda + ci0 : ring_zero_page + dc * 8;
        ring_dma16(src, Al + 8 * (wave + 4 * j) * RS);
      }
      ci0 += BK;
      if (ci0 >= p.Cin) { ci0 = 0; ++tap; }
    }
    ++kt_issue;
    st_issue = (st_issue + 1 == NS) ? 0 : st_issue + 1;
  };

  f32x16 acc[TN][TM];
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

  const int f_sw = (l31 >> 1) & 7;                      // read-side swizzle: fragment rows are (multiple of 32) + l31
  auto compute = [&](int st) {
    const unsigned short* Al = smem + st * BUF;
    const unsigned short* Wl = Al + BM * RS;
    const unsigned short* af_base = Al + (wm * WM + l31) * RS;
    const unsigned short* wf_base = Wl + (wn * WN + l31) * RS;
    u32x4 wf[2][TN], af[2][TM];                         // register double-buffered fragments
    {
      const int slot = (hi ^ f_sw) * 8;
#pragma unroll
      for (int a = 0; a < TN; ++a) wf[0][a] = *reinterpret_cast<const u32x4*>(wf_base + a * 32 * RS + slot);
#pragma unroll
      for (int b = 0; b < TM; ++b) af[0][b] = *reinterpret_cast<const u32x4*>(af_base + b * 32 * RS + slot);
    }
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      const int cur = ks & 1, nxt = cur ^ 1;
      if (ks + 1 < BK / 16) {
        const int slot = (((ks + 1) * 2 + hi) ^ f_sw) * 8;
#pragma unroll
        for (int a = 0; a < TN; ++a) wf[nxt][a] = *reinterpret_cast<const u32x4*>(wf_base + a * 32 * RS + slot);
#pragma unroll
        for (int b = 0; b < TM; ++b) af[nxt][b] = *reinterpret_cast<const u32x4*>(af_base + b * 32 * RS + slot);
      }
#pragma unroll
      for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b) acc[a][b] = Elem<DT>::mfma32(wf[cur][a], af[cur][b], acc[a][b]);
    }
  };

  const int pre = min(NS - 1, nk);
  for (int i = 0; i < pre; ++i) issue_tile();           // NS - 1 K-tiles in flight before the first wait
  int st = 0;
  for (int kt = 0; kt < nk; ++kt) {
    // K-tile kt has landed when at most the tiles enqueued behind it are outstanding (a wave's LDS-DMA retire in order) ...
    ring_wait<PER_TILE, NS - 2>(kt_issue - kt - 1);
    __builtin_amdgcn_s_barrier();                       // ... for every wave; and every wave is done with stage (kt - 1) % NS,
    if (kt_issue < nk) issue_tile();                    // which K-tile kt + NS - 1 now refills
    compute(st);
    st = (st + 1 == NS) ? 0 : st + 1;
  }
  tile_epilogue<DT, BM, BN, WM, WN, LDS_BYTES>(p, acc, smem, m0, n0, bz, wm, wn, lane, wave);
}

// out = epi(sum over K-slices) for split-K launches: one thread per 8 consecutive columns of one row
template <int DT>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const CoreParams p) {
  const int n8 = (p.N + 7) / 8;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)p.tail_rows * n8) return;                // the slabs cover rows [tail_m0, tail_m0 + tail_rows)
  const int ml = (int)(i / n8), n = (int)(i - (size_t)ml * n8) * 8;
  const int m = p.tail_m0 + ml;
  float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const bool vec = (n + 7 < p.N) && ((p.N & 3) == 0);
  const size_t slab = (size_t)p.tail_rows * p.N;
  const float* w = p.ws + (size_t)ml * p.N + n;
  if (vec) {
    // four slabs' loads in flight per thread (a 2-row forward runs ~170 of these launches, each a chain of `splitk` dependent
    // trips to L2 before: 8.3 us on average), summed in slice order as before -- same bits
    int s = 0;
    for (; s + 4 <= p.splitk; s += 4) {
      f32x4 a[4][2];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        a[u][0] = *reinterpret_cast<const f32x4*>(w + (size_t)(s + u) * slab);
        a[u][1] = *reinterpret_cast<const f32x4*>(w + (size_t)(s + u) * slab + 4);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] += a[u][0][e]; v[e + 4] += a[u][1][e]; }
    }
    for (; s < p.splitk; ++s) {
      const f32x4 w0 = *reinterpret_cast<const f32x4*>(w + (size_t)s * slab), w1 = *reinterpret_cast<const f32x4*>(w + (size_t)s * slab + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[e] += w0[e]; v[e + 4] += w1[e]; }
    }
  } else {
    for (int s = 0; s < p.splitk; ++s) {
#pragma unroll
      for (int e = 0; e < 8; ++e) if (n + e < p.N) v[e] += w[(size_t)s * slab + e];
    }
  }
  const float gate = (p.epi & IDF_EPI_GATE) ? p.gate[0] : 0.0f;
  epilogue8<DT>(p, 0, m, n, v, gate);
}

// K-loop variant choice (measured on MI355X, profiles/r01_diag_B18_*): LDS-DMA staging wins for dense GEMMs
// (+5..18 % on K >= 1280, ~0 on K = 320); the conv gather is faster with the 2-deep register prefetch of variant 1
// (its activation tile cannot use DMA).  IDF_GEMM_DMA=0/1 forces one variant for A/B runs.
// IDF_GEMM_VARIANT=1|2 forces one K-loop variant for A/B runs.  Default: LDS-DMA (2) for dense GEMMs, register
// prefetch (1) for the conv gather.  A third variant (4-stage ring of 32-deep K-steps, 3 DMA steps in flight, counted
// vmcnt + raw s_barrier) was built and measured SLOWER than variant 2 on every shape but one (744 vs 846 TF at 8192^3,
// 485 vs 644 at K=1280: a barrier per 8 MFMAs costs more than the deeper prefetch buys) and was removed; numbers in
// profiles/r01_diag_B18_gemm_ring_vs_dma.log.  That was THROUGHPUT on grids of thousands of tiles, two workgroups per CU; the
// latency kernel above (round 4: 64-deep K-steps, one barrier per 16 MFMAs, one workgroup per CU) serves the opposite regime --
// grids of <= 256 tiles, where a launch is a chain of dependent trips to memory -- and is chosen by tile count in launch().
std::atomic<long long> idf_stat_gn_epi_launches{0};   // idf_conv3x3 calls whose GroupNorm partials came out of the epilogue (idf_get_stat)
std::atomic<long long> idf_stat_ring_launches{0};
std::atomic<long long> idf_stat_gegluw_launches{0};   // GEGLU projections served by geglu640w_kernel (idf_get_stat)
std::atomic<long long> idf_stat_qkvw_launches{0};     // fused q | k | v projections served by qkv320w_kernel (idf_get_stat)     // launches of the latency kernel (idf_get_stat)
int g_big_mode = -2;
inline int gemm_big_mode() {
  if (g_big_mode == -2) {
    const char* e = getenv("IDF_GEMM_BIG");
    const int v = e ? atoi(e) : IDF_GEMM_BIG_DEFAULT;
    g_big_mode = (v < 0 || v > 3) ? IDF_GEMM_BIG_DEFAULT : v;            // out of range = default (as idf_set_tuning rejects it)
  }
  return g_big_mode;
}

inline int kloop_variant(bool conv) {
  static int v = -2;
  if (v == -2) { const char* e = getenv("IDF_GEMM_VARIANT"); v = e ? atoi(e) : -1; }
  if (v == 1 || v == 2) return v;
  return conv ? 1 : 2;
}

template <int DT, int BM, int BN, int WM, int WN, bool CONV>
int launch_cfg(const CoreParams& p, int batch, hipStream_t s) {
  const int variant = kloop_variant(CONV);
  void (*kern)(const CoreParams) = gemm_kernel<DT, BM, BN, WM, WN, CONV>;
  constexpr int cstage = 4 * WM * (WN + 4) * 4;
  constexpr int kloop2 = 2 * (BM + BN) * 64 * 2;
  int smem = 2 * (BM + BN) * LSTR * 2;
  if (variant == 2) { kern = gemm_kernel_dma<DT, BM, BN, WM, WN, CONV>; smem = kloop2 > cstage ? kloop2 : cstage; }
  static std::atomic<unsigned long long> attr_done[3];
  if (const int e = idf_lds_optin(reinterpret_cast<const void*>(kern), smem, attr_done[variant])) return e;
  CoreParams q = p;
  const int tiles = ((p.N + BN - 1) / BN) * ((p.M + BM - 1) / BM);
  const int nk = p.K / BK;
  q.splitk = 1; q.kt_per_slice = nk;
  q.tail_m0 = 0; q.tail_rows = p.M;                         // uniform split-K: the partial slabs cover every row
  // split-K when the tile grid cannot fill the 256 CUs (small-spatial / small-batch layers with long K)
  if (batch == 1 && p.ws && !(p.epi & IDF_EPI_GEGLU) && tiles < 192 && nk >= 8) {
    int want = (512 + tiles - 1) / tiles;
    if (want > nk / 4) want = nk / 4;
    if (want > 64) want = 64;
    while (want > 1 && (size_t)want * p.M * p.N * sizeof(float) > p.ws_bytes) --want;
    if (want > 1) {
      q.kt_per_slice = (nk + want - 1) / want;
      q.splitk = (nk + q.kt_per_slice - 1) / q.kt_per_slice;
    }
  }
  dim3 grid(tiles, 1, q.splitk > 1 ? q.splitk : batch);
  hipLaunchKernelGGL(kern, grid, dim3(256), smem, s, q);
  if (q.splitk > 1) {
    const size_t n8 = (size_t)q.M * ((q.N + 7) / 8);
    hipLaunchKernelGGL(splitk_reduce_kernel<DT>, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, s, q);
  }
  return idf_launch_status();
}

// Latency kernel (variant 3) dispatch: taken when the tile grid has at most g_ring_tiles tiles (0 = never).
// IDF_GEMM_RING / idf_set_tuning(IDF_TUNE_GEMM_RING) set the threshold; default IDF_GEMM_RING_DEFAULT.
#ifndef IDF_GEMM_RING_DEFAULT
#define IDF_GEMM_RING_DEFAULT 256
#endif
int g_ring_tiles = -1;
inline int gemm_ring_tiles() {
  if (g_ring_tiles < 0) {
    const char* e = getenv("IDF_GEMM_RING");
    const int v = e ? atoi(e) : IDF_GEMM_RING_DEFAULT;
    g_ring_tiles = v < 0 ? IDF_GEMM_RING_DEFAULT : v;
  }
  return g_ring_tiles;
}

// returns IDF_BIG_UNSUPPORTED when the launch is left to variants 1 / 2
template <int DT, int BM, int BN, int WM, int WN, bool CONV, int NS>
int launch_ring_cfg(const CoreParams& p, int batch, hipStream_t s) {
  const int tiles = ((p.N + BN - 1) / BN) * ((p.M + BM - 1) / BM);
  const int nk = p.K / BK;
  // the threshold counts WORKGROUPS: a batched launch multiplies the tile grid by its batch (the transposed-V GEMM of a 128-row
  // forward: 20 tiles x 128 = 2560 workgroups of a kernel that runs one per CU on a 120-128 KB ring -- ADVICE r4); IDF_RING_BATCHED=1
  // restores the round-4 rule (per-batch tile count) for A/B runs
  static int batched_old = -1;
  if (batched_old < 0) { const char* e = getenv("IDF_RING_BATCHED"); batched_old = (e && e[0] == '1') ? 1 : 0; }
  if ((long long)tiles * (batched_old ? 1 : batch) > gemm_ring_tiles()) return IDF_BIG_UNSUPPORTED;
  const int slots = idf_num_cu();
  // a grid of 129 ... 191 tiles with a long K: variants 1 / 2 cut it into K-slices for their two workgroups per CU, this
  // kernel (one per CU) cannot -- throughput, not latency, decides there (3x3 conv 16^2 -> 32^2 at 2 rows: 105 vs 141 us)
  if (batch == 1 && p.ws && !(p.epi & IDF_EPI_GEGLU) && tiles * 2 > slots && tiles < 192 && nk >= 8) return IDF_BIG_UNSUPPORTED;
  void (*kern)(const CoreParams) = gemm_kernel_ring<DT, BM, BN, WM, WN, CONV, NS>;
  constexpr int cstage = 4 * WM * (WN + 4) * 4;
  constexpr int kloop = NS * (BM + BN) * 64 * 2;
  constexpr int smem = kloop > cstage ? kloop : cstage;
  static std::atomic<unsigned long long> attr_done{0};
  if (const int e = idf_lds_optin(reinterpret_cast<const void*>(kern), smem, attr_done)) return e;
  CoreParams q = p;
  q.splitk = 1; q.kt_per_slice = nk;
  q.tail_m0 = 0; q.tail_rows = p.M;
  // split-K towards one workgroup per CU (this kernel's occupancy), slices of >= 8 K-tiles (two to three trips to memory with
  // the ring's depth) plus the reducer
  // least K-tiles per slice: 8 (IDF_RING_SLICE_KT for A/B runs; 4 / 8 / 16 / 32 give 5.49 / 5.29 / 5.73 / 6.07 ms of GEMM + conv
  // per 2-row forward, profiles/r04_ring_slice_length.log: shorter slices pay more reducer traffic than their parallelism buys)
  static int slice_kt = 0;
  if (slice_kt == 0) { const char* e = getenv("IDF_RING_SLICE_KT"); slice_kt = e ? atoi(e) : 8; if (slice_kt < 1) slice_kt = 8; }
  if (batch == 1 && p.ws && !(p.epi & IDF_EPI_GEGLU) && tiles * 2 <= slots && nk >= 2 * slice_kt) {
    int want = slots / tiles;
    if (want > nk / slice_kt) want = nk / slice_kt;
    if (want > 64) want = 64;
    while (want > 1 && (size_t)want * p.M * p.N * sizeof(float) > p.ws_bytes) --want;
    if (want > 1) {
      q.kt_per_slice = (nk + want - 1) / want;
      q.splitk = (nk + q.kt_per_slice - 1) / q.kt_per_slice;
    }
  }
  dim3 grid(tiles, 1, q.splitk > 1 ? q.splitk : batch);
  hipLaunchKernelGGL(kern, grid, dim3(256), smem, s, q);
  ++idf_stat_ring_launches;
  if (q.splitk > 1) {
    const size_t n8 = (size_t)q.M * ((q.N + 7) / 8);
    hipLaunchKernelGGL(splitk_reduce_kernel<DT>, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, s, q);
  }
  return idf_launch_status();
}

template <int DT, bool CONV>
int launch(const CoreParams& p, int batch, hipStream_t s, int* parts_out = nullptr, int* gst_out = nullptr) {
  const bool geglu = (p.epi & IDF_EPI_GEGLU) != 0;
  // tile choice: 128x128 when N fills it; 128(M) x 64(N) for N = 64 mod 128 (e.g. 320) or small N.  For the conv
  // (long K = 9*Cin) the denser 64x64 wave tile wins even with a half-empty last column tile at Cout = 320
  // (measured +15..18 %: 512 -> 602 and 596 -> 677 TF); for the short-K dense GEMMs it loses (378 -> 352 TF).
  // IDF_TILE_WIDE=0/1 forces the choice for A/B runs.
  static int wide = -2;
  if (wide == -2) { const char* e = getenv("IDF_TILE_WIDE"); wide = e ? (e[0] == '1' ? 1 : 0) : -1; }
  const bool use_wide = (wide >= 0) ? (wide == 1) : CONV;
  // K-loop variant 4 (gemm_big.hip): persistent 256 x {320,256} tiles.  IDF_GEMM_BIG=0 off, 1 auto (shape + tile
  // quantisation heuristic), 2 forced whenever the shape qualifies.
  const int big = gemm_big_mode();
  // Order (round 4, profiles/r04_dispatch_variants_2_16_64_128_rows.log): a grid of at most 128 small tiles goes to the latency
  // kernel BEFORE the persistent kernel is asked (at 2 rows it beats the persistent kernel's split-K form on every conv and
  // GEMM of the 32^2 ... 8^2 levels); larger grids ask the persistent kernel first (its occupancy bar is 50 %), then the latency
  // kernel up to its threshold, then variants 1 / 2.  Forcing the persistent kernel (mode 2) keeps it first.
  const bool t128p = geglu || (p.N % 128 == 0) || p.N > 1024 || (use_wide && p.N > 128);
  const long long tiles_small = (long long)((p.N + (t128p ? 127 : 63)) / (t128p ? 128 : 64)) * ((p.M + 127) / 128);
  const bool ring_first = big != 2 && gemm_ring_tiles() > 0 && tiles_small <= 128 && tiles_small <= gemm_ring_tiles();
  if (big > 0 && batch == 1 && !ring_first) {
    int splitk = 1, tail_m0 = 0;
    // mode 3 = automatic + hybrid tail split (the dispatcher only cuts a tail when it is given somewhere to report it)
    const int rc = idf_launch_big(p, DT, CONV, big == 2, s, &splitk, parts_out, big == 3 ? &tail_m0 : nullptr, gst_out);
    if (rc != IDF_BIG_UNSUPPORTED) {
      if (rc == 0 && splitk > 1) {
        CoreParams q = p;
        q.splitk = splitk; q.tail_m0 = tail_m0; q.tail_rows = p.M - tail_m0;
        const size_t n8 = (size_t)q.tail_rows * ((q.N + 7) / 8);
        hipLaunchKernelGGL(splitk_reduce_kernel<DT>, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, s, q);
        return idf_launch_status();
      }
      return rc;
    }
  }
  // self-normalising LN_ROW on the small-tile kernels: the statistics pass runs first (into the caller's buffer when it
  // wants them, else into the head of the workspace -- such GEMMs never split K: their K is one activation row)
  CoreParams q = p;
  if ((p.epi & IDF_EPI_LN_ROW) && !p.ln_stats) {
    float* st = p.ln_stats_out ? p.ln_stats_out : p.ws;
    if (!st || (!p.ln_stats_out && p.ws_bytes < (size_t)p.M * 2 * sizeof(float))) return IDF_E_ARG;
    const int rc = idf_row_stats(p.A, p.lda, st, p.M, p.K, p.ln_eps, DT, s);
    if (rc) return rc;
    q.ln_stats = st; q.stride_ln_stats = 0;
    if (!p.ln_stats_out) { q.ws = nullptr; q.ws_bytes = 0; }
  }
  // (64x64 tiles for short-K dense GEMMs were measured 6-20 % slower: profiles/r01_diag_B18_tile64_ab.log)
  const bool t128 = geglu || (q.N % 128 == 0) || q.N > 1024 || (use_wide && q.N > 128);
  if (gemm_ring_tiles() > 0) {                              // small grids: the latency kernel (variant 3), same tiles
    const int rc = t128 ? launch_ring_cfg<DT, 128, 128, 64, 64, CONV, 4>(q, batch, s) : launch_ring_cfg<DT, 128, 64, 64, 32, CONV, 5>(q, batch, s);
    if (rc != IDF_BIG_UNSUPPORTED) return rc;
  }
  if (t128) return launch_cfg<DT, 128, 128, 64, 64, CONV>(q, batch, s);
  return launch_cfg<DT, 128, 64, 64, 32, CONV>(q, batch, s);
}

}  // namespace

// out_stats finalize: (mu, rstd) of a row from the `parts` equal-count (mean, M2) slots the persistent kernel's epilogue left
// (Chan's pairwise update, slots merged in index order: bitwise reproducible)
__global__ __launch_bounds__(256) void stats_finalize_kernel(const float* __restrict__ parts, int P, float cols, float* __restrict__ out,
                                                            int M, float eps) {
  const int m = blockIdx.x * 256 + threadIdx.x;
  if (m >= M) return;
  const f32x2* pp = reinterpret_cast<const f32x2*>(parts) + (size_t)m * P;
  f32x2 t = pp[0];
  float mean = t[0], m2 = t[1], n = cols;
  for (int i = 1; i < P; ++i) {
    t = pp[i];
    const float dlt = t[0] - mean, nn = n + cols;
    mean += dlt * (cols / nn);
    m2 += t[1] + dlt * dlt * (n * cols / nn);
    n = nn;
  }
  reinterpret_cast<f32x2*>(out)[m] = f32x2{mean, rsqrtf(m2 / n + eps)};
}

int idf_stats_finalize(const float* stat_parts, int parts, int cols_per_part, float* out_stats, int M, float eps, hipStream_t s) {
  hipLaunchKernelGGL(stats_finalize_kernel, dim3((M + 255) / 256), dim3(256), 0, s, stat_parts, parts, (float)cols_per_part,
                     out_stats, M, eps);
  return idf_launch_status();
}

extern "C" int idf_set_tuning(int knob, int value) {
  if (knob == IDF_TUNE_GEMM_BIG) {
    if (value < 0 || value > 3) return IDF_E_ARG;
    const int prev = gemm_big_mode();
    g_big_mode = value;
    return prev;
  }
  if (knob == IDF_TUNE_GEMM_RING) {
    if (value < 0) return IDF_E_ARG;
    const int prev = gemm_ring_tiles();
    g_ring_tiles = value;
    return prev;
  }
  if (knob == IDF_TUNE_BIG_MIN_EFF) {
    if (value < 1 || value > 100) return IDF_E_ARG;
    return idf_big_min_eff_pct(value);
  }
  if (knob == IDF_TUNE_ATTN2) {
    if (value < 0 || value > 6) return IDF_E_ARG;
    return idf_attn2_set_mode(value);
  }
  if (knob == IDF_TUNE_ATTN8) {
    if (value < 0 || value > 6) return IDF_E_ARG;
    return idf_attn8_set_mode(value);
  }
  if (knob == IDF_TUNE_GEGLU_ROW) {
    if (value < 0 || value > 1) return IDF_E_ARG;
    return idf_gegluw_set_mode(value);
  }
  if (knob == IDF_TUNE_QKV_ROW) {
    if (value < 0 || value > 1) return IDF_E_ARG;
    return idf_qkvw_set_mode(value);
  }
  if (knob == IDF_TUNE_MLP) {
    if (value < 0 || value > 1) return IDF_E_ARG;
    return idf_mlp_set_mode(value);
  }
  return IDF_E_ARG;
}

extern "C" long long idf_get_stat(int stat) {
  if (stat == IDF_STAT_GEMM_BIG_LAUNCHES) return idf_stat_big_launches.load();
  if (stat == IDF_STAT_ATTN2_LAUNCHES) return idf_stat_attn2_launches.load();
  if (stat == IDF_STAT_GEMM_RING_LAUNCHES) return idf_stat_ring_launches.load();
  if (stat == IDF_STAT_ATTN8_LAUNCHES) return idf_stat_attn8_launches.load();
  if (stat == IDF_STAT_GN_EPI_LAUNCHES) return idf_stat_gn_epi_launches.load();
  if (stat == IDF_STAT_QKV_ROW_LAUNCHES) return idf_stat_qkvw_launches.load();
  if (stat == IDF_STAT_GEGLU_ROW_LAUNCHES) return idf_stat_gegluw_launches.load();
  return -1;
}

extern "C" int idf_gemm(const idf_gemm_args* a, void* stream) {
  if (!a || !a->A || !a->W || !a->out) return IDF_E_ARG;
  if (a->M <= 0 || a->N <= 0 || a->K <= 0 || (a->K % BK) != 0) return IDF_E_ARG;
  if ((a->lda % 8) || (a->ldw % 8) || !aligned16(a->A) || !aligned16(a->W)) return IDF_E_ALIGN;
  if ((a->epi & IDF_EPI_GATE) && !(a->epi & IDF_EPI_RES)) return IDF_E_ARG;
  if ((a->epi & IDF_EPI_RES) && !a->res) return IDF_E_ARG;
  if ((a->epi & (IDF_EPI_BIAS | IDF_EPI_GEGLU)) && !a->bias) return IDF_E_ARG;
  if ((a->epi & IDF_EPI_GEGLU) && ((a->N % 64) || (a->ldo % 4))) return IDF_E_ARG;
  if ((a->epi & IDF_EPI_GEGLU_P32) && !(a->epi & IDF_EPI_GEGLU)) return IDF_E_ARG;
  if ((a->epi & IDF_EPI_ROWBIAS) && (!a->rowbias || a->rows_per_batch <= 0)) return IDF_E_ARG;
  if (a->epi & IDF_EPI_OUT_NCHW) return IDF_E_ARG;
  CoreParams p{};
  p.W = (const unsigned short*)a->W; p.ldw = a->ldw; p.strideW = a->strideW; p.N = a->N;
  p.A = (const unsigned short*)a->A; p.lda = a->lda; p.strideA = a->strideA; p.M = a->M; p.K = a->K;
  p.out = a->out; p.ldo = a->ldo; p.strideO = a->strideO;
  p.bias = a->bias; p.rowbias = (const unsigned short*)a->rowbias; p.ld_rowbias = a->ld_rowbias;
  p.rows_per_batch = a->rows_per_batch;
  p.res = (const unsigned short*)a->res; p.ldr = a->ldr; p.strideR = a->strideR;
  p.gate = a->gate; p.epi = a->epi; p.n_valid = a->N;
  p.ws = (float*)a->ws; p.ws_bytes = a->ws ? (size_t)a->ws_bytes : 0;
  const int batch = a->batch > 0 ? a->batch : 1;
  if (a->epi & (IDF_EPI_LN_ROW | IDF_EPI_LN_COL)) {
    if ((a->epi & IDF_EPI_LN_ROW) && (a->epi & IDF_EPI_LN_COL)) return IDF_E_ARG;
    if (!a->ln_c) return IDF_E_ARG;
    if (!a->ln_stats) {
      // self-normalising LN_ROW: statistics of A's rows computed by the GEMM itself (K = the whole row), unbatched
      if (!(a->epi & IDF_EPI_LN_ROW) || batch != 1 || (a->K % 8) || a->K > 1536 || !(a->ln_eps > 0.0f)) return IDF_E_ARG;
      if (a->ln_stats_out && (((uintptr_t)a->ln_stats_out) & 7u)) return IDF_E_ALIGN;
      p.ln_eps = a->ln_eps; p.ln_stats_out = a->ln_stats_out;
    }
    if ((a->epi & IDF_EPI_LN_ROW) && !(a->epi & IDF_EPI_BIAS)) return IDF_E_ARG;       // the beta term travels as bias
    if ((a->epi & IDF_EPI_LN_COL) && (!a->ln_d || (a->epi & IDF_EPI_GEGLU))) return IDF_E_ARG;
    if ((((uintptr_t)a->ln_stats) & 7u) || (a->stride_ln_stats & 1)) return IDF_E_ALIGN;
    p.ln_stats = a->ln_stats; p.stride_ln_stats = a->stride_ln_stats; p.ln_c = a->ln_c; p.ln_d = a->ln_d;
  }
  if (a->out_stats) {
    // by-product for a LayerNorm that follows: (mu, rstd) of every output row -- 16-bit row-major output only
    if ((a->epi & (IDF_EPI_GEGLU | IDF_EPI_OUT_F32)) || (a->N % 8) || a->N > 1536 || (a->ldo % 8)) return IDF_E_ARG;
    if (batch > 1 && a->strideO != (long long)a->M * a->ldo) return IDF_E_ARG;          // rows of all batches must be ld-regular
  }
  hipStream_t s = (hipStream_t)stream;
  int rc = IDF_E_UNSUPPORTED;
  if (a->vt_out) {
    // Fused q | k | v projection: columns [vt_col0, N) are stored transposed into vt_out.  One launch of the persistent
    // kernel when the shape qualifies; otherwise the two GEMMs it replaces (out = A . W[:vt_col0]^T, then the transposed-V
    // projection vt = W[vt_col0:] . A^T with the LayerNorm statistics on its column side) -- same results up to fp32
    // summation order, so the caller never has to know which form ran.
    const int Nv = a->N - a->vt_col0;
    if (a->vt_col0 <= 0 || Nv <= 0 || batch != 1 || a->ld_vt < a->M || (a->ld_vt % 8) || !aligned16(a->vt_out)) return IDF_E_ARG;
    if (a->epi & ~(IDF_EPI_BIAS | IDF_EPI_LN_ROW)) return IDF_E_ARG;
    // LN_ROW is required: the two-GEMM form carries the transposed columns' bias / beta term as the LN_COL row vector, and a
    // result must not depend on which form the tuning mode and the CU count select (a BIAS-only call used to fail
    // in the fallback AFTER the q | k GEMM had been launched)
    if (!(a->epi & IDF_EPI_LN_ROW)) return IDF_E_ARG;
    if (a->out_stats) return IDF_E_ARG;
    const bool self_ln = (a->epi & IDF_EPI_LN_ROW) && !a->ln_stats;
    if (self_ln && !a->ln_stats_out) return IDF_E_ARG;        // the fallback's second GEMM needs them somewhere
    p.vt_out = (unsigned short*)a->vt_out; p.ld_vt = a->ld_vt; p.vt_col0 = a->vt_col0;
    {   // the row-resident kernel of the C = 320 level (qkv_fused.hip); counted with the persistent-kernel launches
      int r = idf_launch_qkv320w(p, a->dtype, s);
      if (r == IDF_BIG_UNSUPPORTED) r = idf_launch_qkv640w(p, a->dtype, s);       // ... and of the C = 640 level (qkv640_fused.hip)
      if (r != IDF_BIG_UNSUPPORTED) { if (r == 0) { ++idf_stat_big_launches; ++idf_stat_qkvw_launches; } return r; }
    }
    if (gemm_big_mode() > 0) {
      const int r = idf_launch_big(p, a->dtype, false, gemm_big_mode() == 2, s, nullptr);
      if (r != IDF_BIG_UNSUPPORTED) return r;
    }
    CoreParams p1 = p;                                          // q | k: the first vt_col0 columns
    p1.vt_out = nullptr; p1.vt_col0 = 0; p1.N = a->vt_col0; p1.n_valid = a->vt_col0;
    if (a->dtype == IDF_BF16) rc = launch<IDF_BF16, false>(p1, 1, s);
    else if (a->dtype == IDF_F16) rc = launch<IDF_F16, false>(p1, 1, s);
    if (rc) return rc;
    CoreParams p2{};                                            // V^T[Nv, M] = Wv[Nv, K] . A[M, K]^T
    p2.A = p.W + (size_t)a->vt_col0 * p.ldw; p2.lda = p.ldw; p2.M = Nv; p2.K = a->K;
    p2.W = p.A; p2.ldw = p.lda; p2.N = a->M; p2.n_valid = a->M;
    p2.out = a->vt_out; p2.ldo = a->ld_vt;
    p2.ws = p.ws; p2.ws_bytes = p.ws_bytes;
    p2.epi = 0;
    if (a->epi & IDF_EPI_LN_ROW) {
      p2.epi |= IDF_EPI_LN_COL;
      p2.ln_stats = self_ln ? a->ln_stats_out : a->ln_stats; p2.stride_ln_stats = 0;
      p2.ln_c = a->ln_c + a->vt_col0; p2.ln_d = a->bias + a->vt_col0;
    }
    if (a->dtype == IDF_BF16) return launch<IDF_BF16, false>(p2, 1, s);
    if (a->dtype == IDF_F16) return launch<IDF_F16, false>(p2, 1, s);
    return IDF_E_UNSUPPORTED;
  }
  // out_stats: the persistent kernel leaves per-wave (mean, M2) partials of its output rows in the head of the workspace
  // (when it takes the launch, unsplit) and a 16-B-per-row finalize pass merges them; otherwise the statistics pass re-reads
  // the output as before
  int parts = 0;
  // (whether the workspace holds the [M][parts][2] partials is checked where `parts` is chosen: idf_launch_big)
  if ((a->epi & IDF_EPI_GEGLU) && batch == 1 && !a->out_stats) {
    // the row-resident GEGLU kernel of the C = 640 level (geglu_fused.hip); counted with the persistent-kernel launches
    const int r = idf_launch_geglu640w(p, a->dtype, s);
    if (r != IDF_BIG_UNSUPPORTED) { if (r == 0) { ++idf_stat_big_launches; ++idf_stat_gegluw_launches; } return r; }
  }
  if (a->out_stats && batch == 1 && p.ws) p.stat_parts = p.ws;
  if (a->dtype == IDF_BF16) rc = launch<IDF_BF16, false>(p, batch, s, &parts);
  else if (a->dtype == IDF_F16) rc = launch<IDF_F16, false>(p, batch, s, &parts);
  if (rc == 0 && a->out_stats) {
    if (parts > 0) rc = idf_stats_finalize(p.stat_parts, parts, a->N / parts, a->out_stats, a->M, a->out_stats_eps, s);
    else rc = idf_row_stats(a->out, a->ldo, a->out_stats, batch * a->M, a->N, a->out_stats_eps, a->dtype, stream);
  }
  return rc;
}

extern "C" int idf_conv3x3(const idf_conv3x3_args* a, void* stream) {
  if (!a || !a->x || !a->W || !a->out) return IDF_E_ARG;
  if (a->B <= 0 || a->Cin <= 0 || (a->Cin % BK) != 0 || a->Cout <= 0) return IDF_E_ARG;
  if (a->stride != 1 && a->stride != 2) return IDF_E_ARG;
  if (a->upsample != 0 && a->upsample != 1) return IDF_E_ARG;
  if ((a->ldx % 8) || !aligned16(a->x) || !aligned16(a->W)) return IDF_E_ALIGN;
  if (a->epi & (IDF_EPI_GEGLU | IDF_EPI_GATE)) return IDF_E_ARG;
  if ((a->epi & IDF_EPI_RES) && !a->res) return IDF_E_ARG;
  if ((a->epi & IDF_EPI_BIAS) && !a->bias) return IDF_E_ARG;
  CoreParams p{};
  const int Hup = a->Hin << a->upsample, Wup = a->Win << a->upsample;
  p.Ho = (Hup - 1) / a->stride + 1; p.Wo = (Wup - 1) / a->stride + 1;
  p.W = (const unsigned short*)a->W; p.ldw = 9 * a->Cin; p.strideW = 0; p.N = a->Cout;
  p.A = (const unsigned short*)a->x; p.lda = a->ldx; p.strideA = 0; p.M = a->B * p.Ho * p.Wo; p.K = 9 * a->Cin;
  p.Hin = a->Hin; p.Win = a->Win; p.Cin = a->Cin; p.stride = a->stride; p.up = a->upsample;
  p.out = a->out; p.ldo = a->ldo; p.strideO = 0;
  p.bias = a->bias; p.rowbias = (const unsigned short*)a->rowbias; p.ld_rowbias = a->ld_rowbias;
  p.rows_per_batch = p.Ho * p.Wo;
  p.res = (const unsigned short*)a->res; p.ldr = a->ldr; p.strideR = 0;
  p.gate = nullptr; p.epi = a->epi; p.n_valid = a->n_valid > 0 ? a->n_valid : a->Cout;
  p.ws = (float*)a->ws; p.ws_bytes = a->ws ? (size_t)a->ws_bytes : 0;
  if ((a->epi & IDF_EPI_ROWBIAS) && !a->rowbias) return IDF_E_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (a->gn_partial) {
    // GroupNorm partials of the output (see idf.h): whole 64-row chunks per sample, 32 groups, a 16-bit NHWC output matrix
    const int hw = p.Ho * p.Wo;
    if ((hw % 64) || (a->Cout % 32) || a->ldo != a->Cout || (a->epi & (IDF_EPI_OUT_F32 | IDF_EPI_OUT_NCHW))) return IDF_E_ARG;
    if (((uintptr_t)a->gn_partial) & 7u) return IDF_E_ALIGN;
    p.gn_partial = a->gn_partial; p.gn_hw = hw;
  }
  int gst = 0, rc = IDF_E_UNSUPPORTED;
  if (a->dtype == IDF_BF16) rc = launch<IDF_BF16, true>(p, 1, s, nullptr, &gst);
  else if (a->dtype == IDF_F16) rc = launch<IDF_F16, true>(p, 1, s, nullptr, &gst);
  // the contract holds whichever kernel took the launch: where the epilogue did not leave the partials, the statistics pass does
  if (rc == 0 && gst) ++idf_stat_gn_epi_launches;
  if (rc == 0 && a->gn_partial && !gst)
    rc = idf_groupnorm_stats(a->out, a->gn_partial, a->B, p.Ho * p.Wo, a->Cout, p.Ho * p.Wo / 64, a->dtype, stream);
  return rc;
}
