// gemm_big.hip -- K-loop variant 4: persistent big-tile MFMA GEMM / implicit-GEMM conv3x3 for gfx950.
//
// Why (measured on MI355X, profiles/r01_diag_B18_*): the 128x128 / 4-wave kernels of gemm_conv.hip top out at
// ~870 TFLOP/s (35 % of the 2.5 PFLOP/s dense bf16 peak) and at ~400 on the K = 320 layers.  Two causes:
//   * LDS: a 64x64 wave tile reads 4 fragments per 4 MFMAs and the 128x128 block tile stages 32 KB per 512 MFMA
//     cycles -- the LDS pipe is as busy as the MFMA pipe;
//   * a workgroup lives for only K/64 (= 5 for K = 320) K-tiles, so its prologue load latency, epilogue and launch
//     ramp are paid per 1280 MFMA cycles.
// This variant: ONE persistent 512-thread workgroup (8 wave64, 2 per SIMD) per CU walking 256 x BN output tiles
// (BN = 320 or 256 -- every channel count of the SD-1.5 UNet is a multiple of 320, every GEGLU width of 256);
// wave tile 64 x BN/2 (7 or 6 fragment reads per 10 or 8 MFMAs); both operand tiles staged by LDS-DMA
// (global_load_lds_dwordx4) into a 2-stage ring of 72 KB stages; the K-tile stream is FLATTENED across output tiles,
// so the first K-tile of the next tile is in flight during the last MFMAs and the epilogue of the current one.
// The conv3x3 activation gather also uses LDS-DMA: a lane whose tap falls in the zero padding points its source
// address at a 128-B page of zeros instead of masking a register.
// Epilogue: entirely in registers.  GEGLU pairs are combined first (value and gate accumulators of one output sit in the
// same lane); v_permlane32_swap then exchanges register pairs between the two lane halves so that a lane owns 16
// consecutive columns of its row: 16-B stores / residual / rowbias accesses without an LDS transposition or a barrier.
//
// Round 2 (profiles/r02_*): the LDS-DMA pieces are inline assembly and the second half of the workgroup enqueues them from
// the middle of its K-tile (fill schedule, see the K loop); split-K work items for grids that leave most CUs idle
// (SPLIT); LayerNorm folded into the epilogue (IDF_EPI_LN_ROW / LN_COL) with the row statistics optionally summed in the
// K loop itself (LNS); 128-wide tiles with three stages; a role-split ping-pong variant of the same tile
// (gemm_kernel_pp, geometry 2: measured slower, kept for A/B).  What bounds the K loop: tools/ubench/dma_rate.hip --
// a wave moves 5.6 B/clk from L2 into LDS, a CU 34 B/clk, the 256 x 320 tile needs 28.8 B/clk at 100 % MFMA.
//
// Roofline: MFMA-bound (2.5 PFLOP/s dense bf16); algorithmic flops 2*M*N*K.  LDS image and XOR swizzle are the ones
// of gemm_kernel_dma (linear 128-B rows, 16-B slot ^= (row >> 1) & 7 applied on the global source address).
#include "gemm_core.h"
#include <cstdlib>

using namespace idfcore;

namespace {

__device__ __attribute__((aligned(128))) unsigned short idf_zero_page[64];   // zero-initialised device memory

constexpr int WM = 64, TM = 2;                // wave tile: 64 rows x BN/2 columns

template <int I> struct IC { static constexpr int value = I; };
template <int I, int N, int STEP, class F> __device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) { f(IC<I>{}); static_for<I + STEP, N, STEP>(f); }
}

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// LDS-DMA as inline assembly: with the builtin the compiler orders every later ds_read behind ALL outstanding LDS-DMA
// (s_waitcnt vmcnt(0)); both kernels below keep LDS-DMA in flight across their fragment reads and do their own waits.  lds = LDS byte address of lane 0's 16-B slot (lane i lands at lds + 16 i); it goes through M0, which
// nothing else in that kernel uses.
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(size_t)p; }
__device__ __forceinline__ void dma16_sv(const void* sbase /* wave-uniform */, unsigned voff_bytes, unsigned lds) {
  lds = __builtin_amdgcn_readfirstlane(lds);
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds), "v"(voff_bytes), "s"(sbase) : "memory");
}
__device__ __forceinline__ void dma16_v(const void* addr /* per lane */, unsigned lds) {
  lds = __builtin_amdgcn_readfirstlane(lds);
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds), "v"(addr) : "memory");
}

// ---------------- tile epilogue (shared by the lock-step and the ping-pong kernel): no LDS, no barrier.
// acc[a][b][4q+e] = D[n = a*32 + 8q + 4hi + e][m = b*32 + l31].  Two v_permlane32_swap per register pair
// (q0 <-> q2, q1 <-> q3 between the lane halves) leave every lane with 16 CONSECUTIVE columns of its row:
// n = a*32 + 16*hi + [0,16)  ->  two 16-B stores per (a, b), 16-B residual / rowbias loads, float4 bias loads.
// LNS: the (mu, rstd) of the tile's A rows were computed in the K loop (lnm / lnr per 32-row fragment) instead of read.
template <int DT, int BM, int BN, int TN, bool SPLIT, bool LNS = false>
__device__ __forceinline__ void big_epilogue(const CoreParams& p, f32x16 (&acc)[TN][TM], int seq, int slice, int tiles_n, int wm,
                                             int wn, int l31, int hi, float gate, const float* lnm = nullptr,
                                             const float* lnr = nullptr) {
  constexpr int WN = BN / 2;
  const int epi = p.epi;
  const int m_tile = seq / tiles_n;
  const int n0 = (seq - m_tile * tiles_n) * BN, m0 = m_tile * BM;
  const int mw = m0 + wm * WM, nw = n0 + wn * WN;
  auto swap16 = [&](const f32x16& c, float* v) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const auto s02 = __builtin_amdgcn_permlane32_swap(__float_as_uint(c[e]), __float_as_uint(c[8 + e]), false, false);
      const auto s13 = __builtin_amdgcn_permlane32_swap(__float_as_uint(c[4 + e]), __float_as_uint(c[12 + e]), false, false);
      v[e] = __uint_as_float(s02[0]); v[4 + e] = __uint_as_float(s02[1]);
      v[8 + e] = __uint_as_float(s13[0]); v[12 + e] = __uint_as_float(s13[1]);
    }
  };
  if constexpr (LNS) {                                      // leave the row statistics for an LN_COL consumer of A
    if (p.ln_stats_out && n0 == 0 && wn == 0 && hi == 0) {
#pragma unroll
      for (int b = 0; b < TM; ++b) {
        const int m = mw + b * 32 + l31;
        if (m < p.M) *reinterpret_cast<f32x2*>(p.ln_stats_out + 2 * (size_t)m) = f32x2{lnm[b], lnr[b]};
      }
    }
  }
  if constexpr (SPLIT) {                                    // fp32 partials of this K-slice; the reducer applies the epilogue
    static_for<0, TN, 1>([&](auto AI) {
      constexpr int a = decltype(AI)::value;
      const int n = nw + a * 32 + 16 * hi;
#pragma unroll
      for (int b = 0; b < TM; ++b) {
        float v[16];
        swap16(acc[a][b], v);
        const int m = mw + b * 32 + l31;
        if (m >= p.M) continue;
        float* o = p.ws + ((size_t)slice * p.M + m) * p.N + n;
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(o + 4 * j) = f32x4{v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]};
      }
    });
    return;
  }
  if (epi & IDF_EPI_GEGLU) {
    if constexpr ((TN & 1) == 0) {
      static_for<0, TN, 2>([&](auto AI) {
        constexpr int a = decltype(AI)::value;
        const int npk = nw + a * 32;                    // packed weight rows: [32 value | 32 gate]
        f32x4 bv[4], bg[4], cv[4], cg[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          bv[q] = *reinterpret_cast<const f32x4*>(p.bias + npk + 8 * q + 4 * hi);
          bg[q] = *reinterpret_cast<const f32x4*>(p.bias + npk + 32 + 8 * q + 4 * hi);
          if (epi & IDF_EPI_LN_ROW) {
            cv[q] = *reinterpret_cast<const f32x4*>(p.ln_c + npk + 8 * q + 4 * hi);
            cg[q] = *reinterpret_cast<const f32x4*>(p.ln_c + npk + 32 + 8 * q + 4 * hi);
          }
        }
#pragma unroll
        for (int b = 0; b < TM; ++b) {
          f32x16 o;
          if (epi & IDF_EPI_LN_ROW) {               // LayerNorm folded in: rstd * (acc - mu * c) + (beta term + bias)
            const int mr = min(mw + b * 32 + l31, p.M - 1);
            f32x2 st;
            if constexpr (LNS) st = f32x2{lnm[b], lnr[b]};
            else st = *reinterpret_cast<const f32x2*>(p.ln_stats + 2 * (size_t)mr);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const float val = fmaf(st[1], fmaf(-st[0], cv[q][e], acc[a][b][4 * q + e]), bv[q][e]);
                const float gat = fmaf(st[1], fmaf(-st[0], cg[q][e], acc[a + 1][b][4 * q + e]), bg[q][e]);
                o[4 * q + e] = val * gelu_erf_f(gat);
              }
          } else {
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e)
              o[4 * q + e] = (acc[a][b][4 * q + e] + bv[q][e]) * gelu_erf_f(acc[a + 1][b][4 * q + e] + bg[q][e]);
          }
          float v[16];
          swap16(o, v);
          const int m = mw + b * 32 + l31;
          if (m < p.M) {
            unsigned short* op = reinterpret_cast<unsigned short*>(p.out) + (size_t)m * p.ldo + (npk >> 1) + 16 * hi;
            *reinterpret_cast<u32x4*>(op) = pack8<DT>(v);
            *reinterpret_cast<u32x4*>(op + 8) = pack8<DT>(v + 8);
          }
        }
      });
    }
  } else {
    static_for<0, TN, 1>([&](auto AI) {
      constexpr int a = decltype(AI)::value;
      const int n = nw + a * 32 + 16 * hi;
      f32x4 bs[4], cs[4];
      if (epi & IDF_EPI_BIAS) {
#pragma unroll
        for (int j = 0; j < 4; ++j) bs[j] = *reinterpret_cast<const f32x4*>(p.bias + n + 4 * j);
      }
      if (epi & IDF_EPI_LN_ROW) {
#pragma unroll
        for (int j = 0; j < 4; ++j) cs[j] = *reinterpret_cast<const f32x4*>(p.ln_c + n + 4 * j);
      }
#pragma unroll
      for (int b = 0; b < TM; ++b) {
        float v[16];
        swap16(acc[a][b], v);
        const int m = mw + b * 32 + l31;
        if (m >= p.M) continue;
        if (epi & IDF_EPI_LN_ROW) {                 // v = rstd_m * (acc - mu_m * c[n]); the beta term arrives as bias
          f32x2 st;
          if constexpr (LNS) st = f32x2{lnm[b], lnr[b]};
          else st = *reinterpret_cast<const f32x2*>(p.ln_stats + 2 * (size_t)m);
#pragma unroll
          for (int j = 0; j < 16; ++j) v[j] = st[1] * fmaf(-st[0], cs[j >> 2][j & 3], v[j]);
        }
        if (epi & IDF_EPI_LN_COL) {                 // v = rstd_n * (acc - c[m] * mu_n) + d[m]: 16 token columns of row m
          const float cm = p.ln_c[m], dm = p.ln_d[m];
          const f32x4* st4 = reinterpret_cast<const f32x4*>(p.ln_stats + 2 * (size_t)n);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const f32x4 t = st4[j];                   // (mu, rstd) of tokens n + 2j, n + 2j + 1
            v[2 * j] = fmaf(t[1], fmaf(-cm, t[0], v[2 * j]), dm);
            v[2 * j + 1] = fmaf(t[3], fmaf(-cm, t[2], v[2 * j + 1]), dm);
          }
        }
        if (epi & IDF_EPI_BIAS) {
#pragma unroll
          for (int j = 0; j < 16; ++j) v[j] += bs[j >> 2][j & 3];
        }
        if (epi & IDF_EPI_ROWBIAS) {
          const unsigned short* rb = p.rowbias + (size_t)(m / p.rows_per_batch) * p.ld_rowbias + n;
          float r[16];
          unpack8<DT>(*reinterpret_cast<const u32x4*>(rb), r);
          unpack8<DT>(*reinterpret_cast<const u32x4*>(rb + 8), r + 8);
#pragma unroll
          for (int j = 0; j < 16; ++j) v[j] += r[j];
        }
        if (epi & IDF_EPI_SILU) {
#pragma unroll
          for (int j = 0; j < 16; ++j) v[j] = silu_f(v[j]);
        }
        if (epi & IDF_EPI_GELU) {
#pragma unroll
          for (int j = 0; j < 16; ++j) v[j] = gelu_erf_f(v[j]);
        }
        if (epi & IDF_EPI_RES) {
          const unsigned short* rr = p.res + (size_t)m * p.ldr + n;
          const float gm = (epi & IDF_EPI_GATE) ? gate : 1.0f;
          float r[16];
          unpack8<DT>(*reinterpret_cast<const u32x4*>(rr), r);
          unpack8<DT>(*reinterpret_cast<const u32x4*>(rr + 8), r + 8);
#pragma unroll
          for (int j = 0; j < 16; ++j) v[j] = fmaf(gm, v[j], r[j]);
        }
        if (epi & IDF_EPI_OUT_F32) {
          float* o = reinterpret_cast<float*>(p.out) + (size_t)m * p.ldo + n;
#pragma unroll
          for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(o + 4 * j) = f32x4{v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]};
        } else {
          unsigned short* o = reinterpret_cast<unsigned short*>(p.out) + (size_t)m * p.ldo + n;
          *reinterpret_cast<u32x4*>(o) = pack8<DT>(v);
          *reinterpret_cast<u32x4*>(o + 8) = pack8<DT>(v + 8);
        }
      }
    });
  }
}

// Geometry: BM x BN output tile, (BM/64) x 2 waves (wave tile 64 x BN/2), K-tile BKT, NSTG-stage LDS ring.
//   <256, BN, 64, 2>: ONE 8-wave workgroup per CU (2 x 72 KB stages).
//   <128, BN, 32, 3|2>: TWO independent 4-wave workgroups per CU (their barriers, DMA waits and epilogues interleave on
//   the SIMDs instead of coinciding); 64-B LDS rows, 16-B slot ^= (row >> 2) & 3.
template <int DT, int BM, int BN, int BKT, int NSTG, bool CONV, bool SPLIT, bool LNS = false>
__global__ __launch_bounds__(BM * 2, (BM == 128 ? 2 : 1)) void gemm_kernel_big(const CoreParams p, const int tiles_total, const int skew) {
  constexpr int WN = BN / 2, TN = WN / 32;
  constexpr int NW = BM / 32;                              // waves per workgroup (8 or 4)
  constexpr int RS = BKT;                                  // LDS row stride (elements): linear rows, no padding
  constexpr int CPR = BKT / 8;                             // 16-B chunks per row (8 or 4)
  constexpr int RPI = 64 / CPR;                            // rows moved by one LDS-DMA wave instruction (8 or 16)
  constexpr int W_INST = BN / (RPI * NW), A_INST = BM / (RPI * NW);   // LDS-DMA instructions per wave per K-tile
  constexpr int DPW = W_INST + A_INST;
  constexpr int STAGE = (BM + BN) * RS;                    // elements per pipeline stage
  static_assert(BN % (RPI * NW) == 0 && BM % (RPI * NW) == 0, "tile rows must split evenly over the DMA instructions");
  extern __shared__ __attribute__((aligned(16))) unsigned short smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int wn = wave & 1, wm = wave >> 1;
  const int G = gridDim.x;
  // XCD-aware order: workgroup L runs on XCD L % 8; within one round of G tiles XCD x takes the CONTIGUOUS tiles
  // [x*G/8, (x+1)*G/8) of the n-fastest list, so tiles sharing an activation m-tile / weight n-tile share an L2.
  const int slot = ((G & 7) == 0) ? (int)(blockIdx.x & 7) * (G >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  const int tiles_n = p.N / BN;
  // split-K (p.splitk > 1): work item seq = tile * splitk + slice covers K-tiles [slice * nk, (slice + 1) * nk) of its tile
  // and leaves fp32 partials in p.ws[slice][M][N] (reduced + epilogue by splitk_reduce_kernel)
  const int S = SPLIT ? p.splitk : 1;
  const int nk = p.kt_per_slice;                            // K-tiles per work item (= K / BKT without split-K)

  // ---------------- loader state (runs one K-tile ahead of the MFMAs, across output-tile boundaries)
  const int dr = lane / CPR, dc = lane % CPR;
  auto swz = [](int row) { return CPR == 8 ? ((row >> 1) & 7) : ((row >> 2) & 3); };   // conflict-free ds_read_b128 (see header)
  unsigned woff[W_INST], aoff[A_INST];
  int ayx[A_INST];                                        // conv: (yo*stride-1) << 16 | (xo*stride-1) & 0xffff
  int l_seq, l_kt = 0, l_k0 = 0, tap = 0, ci0 = 0;

  auto setup_loader = [&](int item) {
    const int tile = item / S, slice = item - tile * S;
    const int m_tile = tile / tiles_n;
    const int n0 = (tile - m_tile * tiles_n) * BN, m0 = m_tile * BM;
    l_k0 = slice * nk;
#pragma unroll
    for (int j = 0; j < W_INST; ++j) {
      const int row = RPI * (wave + NW * j) + dr;
      woff[j] = (unsigned)(n0 + row) * (unsigned)p.ldw + (unsigned)((dc ^ swz(row)) * 8);
    }
#pragma unroll
    for (int j = 0; j < A_INST; ++j) {
      const int row = RPI * (wave + NW * j) + dr;
      const int m = min(m0 + row, p.M - 1);
      const unsigned sw = (unsigned)((dc ^ swz(row)) * 8);
      if (CONV) {
        const int hw = p.Ho * p.Wo;
        const int b = m / hw, rem = m - b * hw;
        const int yo = rem / p.Wo, xo = rem - yo * p.Wo;
        ayx[j] = ((yo * p.stride - 1) << 16) | ((xo * p.stride - 1) & 0xffff);
        aoff[j] = (unsigned)b * (unsigned)(p.Hin * p.Win) * (unsigned)p.lda + sw;
      } else {
        ayx[j] = 0;
        aoff[j] = (unsigned)m * (unsigned)p.lda + sw;
      }
    }
    if (CONV) { const int k_elem = l_k0 * BKT; tap = k_elem / p.Cin; ci0 = k_elem - tap * p.Cin; }
  };

  // LDS-DMA pieces of the K-tile the loader stands on: piece i < W_INST = weight rows, else activation rows.  Inline
  // assembly (dma16_*): the fill of the NEXT stage is spread between the MFMAs and fragment reads of this one, and with the
  // builtin the compiler would order every later ds_read behind it (s_waitcnt vmcnt(0)) or sink the piece past the MFMAs.
  auto issue_piece = [&](int stage, auto II) {
    constexpr int i = decltype(II)::value;
    unsigned short* Al = smem + stage * STAGE;
    if constexpr (i < W_INST) {
      constexpr int j = i;
      dma16_sv(p.W + (size_t)(l_k0 + l_kt) * BKT, woff[j] * 2u, lds_addr(Al + BM * RS + RPI * (wave + NW * j) * RS));
    } else {
      constexpr int j = i - W_INST;
      const unsigned dst = lds_addr(Al + RPI * (wave + NW * j) * RS);
      if (CONV) {
        const int ky = tap / 3, kx = tap - ky * 3;
        const int Hup = p.Hin << p.up, Wup = p.Win << p.up;
        const int yi = (ayx[j] >> 16) + ky, xi = (int)(short)(ayx[j] & 0xffff) + kx;
        const bool ok = (yi >= 0) & (yi < Hup) & (xi >= 0) & (xi < Wup);
        const int ys = yi >> p.up, xs = xi >> p.up;
        const unsigned short* src = ok ? p.A + ci0 + (aoff[j] + (unsigned)(ys * p.Win + xs) * (unsigned)p.lda) : idf_zero_page + dc * 8;
        dma16_v(src, dst);
      } else {
        dma16_sv(p.A + (size_t)(l_k0 + l_kt) * BKT, aoff[j] * 2u, dst);
      }
    }
  };
  auto advance_loader = [&]() {
    if (CONV) { ci0 += BKT; if (ci0 >= p.Cin) { ci0 = 0; ++tap; } }
    if (++l_kt == nk) {
      l_kt = 0;
      l_seq += G;
      if (l_seq < tiles_total) setup_loader(l_seq);
    }
  };
  auto issue_dma = [&](int stage) {                       // enqueue the whole K-tile into `stage`, then advance
    static_for<0, DPW, 1>([&](auto II) { issue_piece(stage, II); });
    advance_loader();
  };

  // ---------------- MFMA side
  f32x16 acc[TN][TM];
  const int f_sw = swz(l31);                              // fragment rows are (multiple of 32) + l31
  int issued = 0;                                         // K-tiles enqueued so far
  float lsx[TM], lsq[TM];                                 // LNS: per-lane partial sum / sum of squares of its A rows
  constexpr int NPOS = (BKT / 16) * TN;                   // (k-step, weight fragment) positions of a K-tile: TM MFMAs each
  // fill modes (`skew`, see the K loop): 0 burst before the MFMAs; 1 / 2: the second half of the workgroup bursts after the
  // middle / last position; 3: one piece every second position, the two halves on alternating positions
  auto compute = [&](int stage, int mode, int par, int st_fill) {
    const unsigned short* Al = smem + stage * STAGE;
    const unsigned short* Wl = Al + BM * RS;
    const unsigned short* af_base = Al + (wm * WM + l31) * RS;
    const unsigned short* wf_base = Wl + (wn * WN + l31) * RS;
    u32x4 wf[2][TN], af[2][TM];                           // register double-buffered fragments
    {
      const int s8 = (hi ^ f_sw) * 8;
#pragma unroll
      for (int a = 0; a < TN; ++a) wf[0][a] = *reinterpret_cast<const u32x4*>(wf_base + a * 32 * RS + s8);
#pragma unroll
      for (int b = 0; b < TM; ++b) af[0][b] = *reinterpret_cast<const u32x4*>(af_base + b * 32 * RS + s8);
    }
    static_for<0, BKT / 16, 1>([&](auto KS) {
      constexpr int ks = decltype(KS)::value;
      constexpr int cur = ks & 1, nxt = cur ^ 1;
      if constexpr (ks + 1 < BKT / 16) {
        const int s8 = (((ks + 1) * 2 + hi) ^ f_sw) * 8;
#pragma unroll
        for (int a = 0; a < TN; ++a) wf[nxt][a] = *reinterpret_cast<const u32x4*>(wf_base + a * 32 * RS + s8);
#pragma unroll
        for (int b = 0; b < TM; ++b) af[nxt][b] = *reinterpret_cast<const u32x4*>(af_base + b * 32 * RS + s8);
      }
      static_for<0, TN, 1>([&](auto AI) {
        constexpr int a = decltype(AI)::value;
        constexpr int pos = ks * TN + a;
#pragma unroll
        for (int b = 0; b < TM; ++b) acc[a][b] = Elem<DT>::mfma32(wf[cur][a], af[cur][b], acc[a][b]);
        if constexpr (LNS && a == 0) {                      // row sums of the A fragments this k-step multiplies: 16 VALU ops
#pragma unroll
          for (int b = 0; b < TM; ++b)
#pragma unroll
            for (int w = 0; w < 4; ++w) {
              Elem<DT>::dot2c(lsx[b], af[cur][b][w], Elem<DT>::ONES2);
              Elem<DT>::dot2c(lsq[b], af[cur][b][w], af[cur][b][w]);
            }
        }
        if constexpr (pos / 2 < DPW) {
          if (mode == 3 && ((pos & 1) ^ par)) issue_piece(st_fill, IC<pos / 2>{});
        }
        if constexpr (pos == NPOS / 2 - 1) {
          if (mode == 1 && par) static_for<0, DPW, 1>([&](auto II) { issue_piece(st_fill, II); });
        }
      });
    });
    if (mode == 2 && par) static_for<0, DPW, 1>([&](auto II) { issue_piece(st_fill, II); });
    if constexpr (2 * DPW > NPOS) {                       // spread mode, short K-tiles: the pieces that found no position
      if (mode == 3) static_for<NPOS / 2, DPW, 1>([&](auto II) { issue_piece(st_fill, II); });
    }
    if (mode > 0) { advance_loader(); ++issued; }
  };

  int seq = slot;
  if (seq >= tiles_total) return;
  l_seq = seq;
  setup_loader(l_seq);
  // prologue: NSTG-1 K-tiles of the flattened (tile, k) stream in flight
#pragma unroll
  for (int j = 0; j < NSTG - 1; ++j)
    if (l_seq < tiles_total) { issue_dma(j); ++issued; }
  int it = 0, st_it = 0;                                  // K-tile consumed next and its ring stage
  const int epi = p.epi;
  const float gate = (epi & IDF_EPI_GATE) ? p.gate[0] : 0.0f;

  for (; seq < tiles_total; seq += G) {
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
      for (int b = 0; b < TM; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
#pragma unroll
    for (int b = 0; b < TM; ++b) { lsx[b] = 0.0f; lsq[b] = 0.0f; }

    for (int kt = 0; kt < nk; ++kt) {
      // K-tile `it` must have landed: an LDS-DMA is ordered for other waves' ds_reads only by the ISSUING wave's vmcnt
      // wait followed by a barrier.  In steady state the NSTG-2 younger K-tiles stay in flight across the barrier
      // (counted wait; VM ops retire in order); at the tail of the stream fewer are outstanding -> wait for all.
      if (NSTG > 2 && issued - it == NSTG - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSTG - 2) * DPW) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                       // ... and every wave has finished reading the stage refilled below
      asm volatile("" ::: "memory");
      // An LDS-DMA instruction holds the issuing wave for ~180-240 cycles (tools/ubench/dma_rate.hip: 5.6 B/clk per wave,
      // 34 B/clk per CU), during which it issues no MFMA.  `skew` de-phases the two waves of a SIMD so that one feeds the
      // matrix pipe while the other is held in the memory pipe: 1 / 2 = the second half of the workgroup enqueues its pieces
      // from the middle / the end of its K-tile, 3 = every wave spreads its pieces between its MFMAs, the halves alternating.
      const bool fill = l_seq < tiles_total;
      int st_fill = st_it + NSTG - 1;
      if (st_fill >= NSTG) st_fill -= NSTG;
      int mode = 0;
      const int par = wave >= NW / 2 ? 1 : 0;
      if (fill) {
        if (skew == 3) mode = 3;
        else if (skew && par) mode = skew;
        if (mode == 0) {
          issue_dma(st_fill);
          ++issued;
        }
      }
      compute(st_it, mode, par, st_fill);
      ++it;
      if (++st_it == NSTG) st_it = 0;
    }

    // epilogue of tile seq: no LDS, no barrier -- a wave that finishes its MFMAs early runs its epilogue while the other
    // wave of its SIMD is still in the K-loop
    float lnm[TM], lnr[TM];
    if constexpr (LNS) {                                    // a row's 8-element chunks alternate between the lane halves
      const float inv_k = 1.0f / (float)p.K;
#pragma unroll
      for (int b = 0; b < TM; ++b) {
        const float sx = lsx[b] + __shfl_xor(lsx[b], 32, 64), sq = lsq[b] + __shfl_xor(lsq[b], 32, 64);
        const float mu = sx * inv_k;
        lnm[b] = mu;
        lnr[b] = rsqrtf(fmaxf(fmaf(-mu, mu, sq * inv_k), 0.0f) + p.ln_eps);
      }
    }
    big_epilogue<DT, BM, BN, TN, SPLIT, LNS>(p, acc, seq / S, seq - (seq / S) * S, tiles_n, wm, wn, l31, hi, gate, lnm, lnr);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Ping-pong variant (geometry 2).  Same 256 x BN tile, wave tile, LDS image family (64-B rows, BKT = 32) and epilogue,
// but the two waves of a SIMD are in OPPOSITE roles: while waves 0-3 run the 20 (16) MFMAs of half-tile H out of
// registers, waves 4-7 read their fragments of H from LDS (and issue LDS-DMA pieces), and vice versa -- the matrix pipe of
// a SIMD always has one wave feeding it, the LDS/DMA latencies sit in the partner's slot.  Ring: 4 stages of 32-deep
// half K-tiles (3 half-tiles = 108 KB in flight per CU against 72 KB in the lock-step kernel, which is what bounds the
// HBM-streaming K = 320 layers).  Phase p of a tile: X = waves 0-3: L(h) at p = 2h, C(h) at 2h+1;  Y = waves 4-7: L(h) at
// 2h+1, C(h) at 2h+2; one s_barrier between phases.  Both groups enqueue their pieces of half-tile H+3 during L(H)/C(H)
// (its stage was last read in phase 2H-1).  A wave makes its own pieces of H visible with a counted `s_waitcnt vmcnt`
// before the barrier that opens X's L(H): the count is the number of VMEM operations it issued after those pieces
// (younger DMA pieces, plus the epilogue's stores when they fall in between; loads/stores retire in order on gfx9).
// Optional per-segment cycle trace (tools/ubench/pp_trace.hip builds this file with -DIDF_PP_TRACE): s_memtime deltas summed
// per segment over all half-tiles, written by waves 0 and 4 of workgroup 0.
#ifdef IDF_PP_TRACE
__device__ unsigned long long idf_pp_trace_buf[2][16];
#define TR_DECL unsigned long long tr_last = __builtin_readcyclecounter(), tr_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define TR(i) { const unsigned long long tr_now = __builtin_readcyclecounter(); tr_acc[i] += tr_now - tr_last; tr_last = tr_now; }
#define TR_DUMP if (blockIdx.x == 0 && lane == 0 && (wave == 0 || wave == 4)) { for (int i = 0; i < 12; ++i) idf_pp_trace_buf[wave >> 2][i] = tr_acc[i]; }
#else
#define TR_DECL
#define TR(i)
#define TR_DUMP
#endif

template <int DT, int BN, bool CONV, int DL>
__global__ __launch_bounds__(512, 1) void gemm_kernel_pp(const CoreParams p, const int tiles_total) {
  constexpr int BM = 256, BKT = 32, NSTG = 4, NW = 8;
  constexpr int WN = BN / 2, TN = WN / 32;
  constexpr int RS = BKT, CPR = 4, RPI = 16;
  constexpr int A_INST = BM / (RPI * NW);                  // 2 activation pieces per wave per half-tile
  constexpr int W_PIECES = BN / RPI;                       // 20 or 16 weight pieces per half-tile
  constexpr int W_INST = (W_PIECES + NW - 1) / NW;         // 3 (waves 0-3 of BN = 320) or 2
  constexpr int NP = A_INST + W_INST;
  constexpr int STAGE = (BM + BN) * RS;
  constexpr int NMF = 2 * TN * TM;                         // MFMAs per half-tile per wave
  extern __shared__ __attribute__((aligned(16))) unsigned short smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int wn = wave & 1, wmm = wave >> 1;                // wave tile: 64-row slab wmm, column half wn
  const int grp = wave >> 2;                               // role group: waves w and w + 4 share a SIMD
  const int G = gridDim.x;
  const int slot = ((G & 7) == 0) ? (int)(blockIdx.x & 7) * (G >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  const int tiles_n = p.N / BN;
  const int nh = p.K / BKT;
  const int my_w = (wave + NW * (W_INST - 1) < W_PIECES) ? W_INST : W_INST - 1;
  const int P = A_INST + my_w;                             // LDS-DMA pieces this wave issues per half-tile
  const int P_L = (DL < A_INST ? DL : A_INST) + (DL > A_INST ? ((DL - A_INST) < my_w ? (DL - A_INST) : my_w) : 0);   // ... of them in the L phase

  // ---------------- loader state (flattened (tile, half-tile) stream)
  const int dr = lane / CPR, dc = lane % CPR;
  auto swz = [](int row) { return (row >> 2) & 3; };
  unsigned woff[W_INST], aoff[A_INST];
  int ayx[A_INST];
  int l_seq, l_kt = 0, tap = 0, ci0 = 0;

  auto setup_loader = [&](int tile) {
    const int m_tile = tile / tiles_n;
    const int n0 = (tile - m_tile * tiles_n) * BN, m0 = m_tile * BM;
#pragma unroll
    for (int j = 0; j < W_INST; ++j) {
      const int row = min(RPI * (wave + NW * j), BN - RPI) + dr;
      woff[j] = (unsigned)(n0 + row) * (unsigned)p.ldw + (unsigned)((dc ^ swz(row)) * 8);
    }
#pragma unroll
    for (int j = 0; j < A_INST; ++j) {
      const int row = RPI * (wave + NW * j) + dr;
      const int m = min(m0 + row, p.M - 1);
      const unsigned sw = (unsigned)((dc ^ swz(row)) * 8);
      if (CONV) {
        const int hw = p.Ho * p.Wo;
        const int b = m / hw, rem = m - b * hw;
        const int yo = rem / p.Wo, xo = rem - yo * p.Wo;
        ayx[j] = ((yo * p.stride - 1) << 16) | ((xo * p.stride - 1) & 0xffff);
        aoff[j] = (unsigned)b * (unsigned)(p.Hin * p.Win) * (unsigned)p.lda + sw;
      } else {
        ayx[j] = 0;
        aoff[j] = (unsigned)m * (unsigned)p.lda + sw;
      }
    }
    tap = 0; ci0 = 0;
  };

  // piece i of the half-tile the loader stands on: i < A_INST activation rows, else weight rows
  auto issue_piece = [&](int stage, auto II) {
    constexpr int i = decltype(II)::value;
    unsigned short* Al = smem + stage * STAGE;
    if constexpr (i < A_INST) {
      constexpr int j = i;
      const unsigned dst = lds_addr(Al + RPI * (wave + NW * j) * RS);
      if (CONV) {
        const int ky = tap / 3, kx = tap - ky * 3;
        const int Hup = p.Hin << p.up, Wup = p.Win << p.up;
        const int yi = (ayx[j] >> 16) + ky, xi = (int)(short)(ayx[j] & 0xffff) + kx;
        const bool ok = (yi >= 0) & (yi < Hup) & (xi >= 0) & (xi < Wup);
        const int ys = yi >> p.up, xs = xi >> p.up;
        const unsigned short* src = ok ? p.A + ci0 + (aoff[j] + (unsigned)(ys * p.Win + xs) * (unsigned)p.lda) : idf_zero_page + dc * 8;
        dma16_v(src, dst);
      } else {
        dma16_sv(p.A + (size_t)l_kt * BKT, aoff[j] * 2u, dst);
      }
    } else {
      constexpr int j = i - A_INST;
      if (j < W_INST - 1 || wave + NW * j < W_PIECES)
        dma16_sv(p.W + (size_t)l_kt * BKT, woff[j] * 2u, lds_addr(Al + BM * RS + RPI * (wave + NW * j) * RS));
    }
  };
  auto advance_loader = [&]() {
    if (CONV) { ci0 += BKT; if (ci0 >= p.Cin) { ci0 = 0; ++tap; } }
    if (++l_kt == nh) {
      l_kt = 0;
      l_seq += G;
      if (l_seq < tiles_total) setup_loader(l_seq);
    }
  };

  // ---------------- fragments / MFMA
  f32x16 acc[TN][TM];
  u32x4 wf[2][TN], af[2][TM];
  const int f_sw = swz(l31);
  auto load_frags = [&](int stage) {
    const unsigned short* Al = smem + stage * STAGE;
    const unsigned short* af_base = Al + (wmm * WM + l31) * RS;
    const unsigned short* wf_base = Al + BM * RS + (wn * WN + l31) * RS;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int s8 = ((ks * 2 + hi) ^ f_sw) * 8;
#pragma unroll
      for (int a = 0; a < TN; ++a) wf[ks][a] = *reinterpret_cast<const u32x4*>(wf_base + a * 32 * RS + s8);
#pragma unroll
      for (int b = 0; b < TM; ++b) af[ks][b] = *reinterpret_cast<const u32x4*>(af_base + b * 32 * RS + s8);
    }
  };

  int seq = slot;
  if (seq >= tiles_total) return;
  l_seq = seq;
  setup_loader(l_seq);
  int issuedH = 0;                                        // half-tiles enqueued so far (by this wave: its own pieces)
  int fill = 0;                                           // ring stage the next enqueued half-tile goes to
  int e_issued = 0, e_stores = 0;                         // last epilogue: half-tiles enqueued before its stores, store count
  auto enqueue_begin = [&]() { return l_seq < tiles_total; };
  auto enqueue_end = [&]() { advance_loader(); ++issuedH; fill = (fill + 1) & (NSTG - 1); };
#pragma unroll
  for (int j = 0; j < NSTG - 1; ++j)
    if (enqueue_begin()) {
      static_for<0, NP, 1>([&](auto II) { issue_piece(fill, II); });
      enqueue_end();
    }
  // own pieces of half-tile H landed: at most `younger` VMEM operations issued after them may still be outstanding
  auto wait_own = [&](int H, int open_pieces) {
    int younger = (issuedH - 1 - H) * P + (H < e_issued ? e_stores : 0) + open_pieces;
    younger = younger < 0 ? 0 : younger;
    switch (younger) {
#define IDF_W(n) case n: asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory"); break;
      IDF_W(0) IDF_W(1) IDF_W(2) IDF_W(3) IDF_W(4) IDF_W(5) IDF_W(6) IDF_W(7) IDF_W(8) IDF_W(9) IDF_W(10) IDF_W(11) IDF_W(12)
      IDF_W(13) IDF_W(14) IDF_W(15) IDF_W(16) IDF_W(17) IDF_W(18) IDF_W(19) IDF_W(20) IDF_W(21) IDF_W(22) IDF_W(23) IDF_W(24)
      IDF_W(25) IDF_W(26) IDF_W(27) IDF_W(28) IDF_W(29) IDF_W(30) IDF_W(31) IDF_W(32) IDF_W(33) IDF_W(34) IDF_W(35) IDF_W(36)
      IDF_W(37) IDF_W(38) IDF_W(39) IDF_W(40)
#undef IDF_W
      default: asm volatile("s_waitcnt vmcnt(40)" ::: "memory"); break;     // more allowed than encodable here: over-wait
    }
  };
  auto barrier = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
  };
  // L phase: fragments of the half-tile in `stage` into registers, the first DL pieces of the next enqueue
  auto phase_L = [&](int stage, bool enq) {
    load_frags(stage);
    if (enq) static_for<0, (DL < NP ? DL : NP), 1>([&](auto II) { issue_piece(fill, II); });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  };
  // C phase: the MFMAs of the half-tile held in registers, the remaining pieces spread between them
  auto phase_C = [&](bool enq) {
    static_for<0, NMF, 1>([&](auto MI) {
      constexpr int i = decltype(MI)::value;
      constexpr int ks = i / (TN * TM), a = (i % (TN * TM)) / TM, b = i % TM;
      acc[a][b] = Elem<DT>::mfma32(wf[ks][a], af[ks][b], acc[a][b]);
      // one LDS-DMA piece after every third MFMA, starting behind the second
      if constexpr (i >= 1 && (i - 1) % 3 == 0 && DL + (i - 1) / 3 < NP) {
        __builtin_amdgcn_sched_barrier(0);
        if (enq) issue_piece(fill, IC<DL + (i - 1) / 3>{});
        __builtin_amdgcn_sched_barrier(0);
      }
    });
    static_assert(DL + (NMF - 2) / 3 + 1 >= NP, "not every piece gets a slot between the MFMAs");
  };
  auto zero_acc = [&]() {
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
      for (int b = 0; b < TM; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
  };

  const float gate = (p.epi & IDF_EPI_GATE) ? p.gate[0] : 0.0f;
  const int stores_per_tile = (p.epi & IDF_EPI_OUT_F32) ? 0 : ((p.epi & IDF_EPI_GEGLU) ? TN * TM : 2 * TN * TM);
  int H = 0;                                              // global half-tile index of the stream this wave consumes
  int stg = 0;                                            // its ring stage

  // One instruction stream for both groups; Y runs it one phase late (an extra barrier at the head of a tile, where X has
  // one at the tail).  The accumulators are cleared in slack time: X while it would wait for Y's epilogue, Y in its idle
  // phase 0.
  TR_DECL
  for (; seq < tiles_total; seq += G) {
    if (grp == 1) {
      wait_own(H, 0);
      barrier();                                          // phase 0 (Y idle)
    }
    zero_acc();
    TR(9)
    for (int h = 0; h < nh; ++h) {
      if (grp == 0) wait_own(H, 0);
      TR(0)
      barrier();                                          // X: phase 2h, Y: phase 2h + 1
      TR(1)
      const bool enq = enqueue_begin();
      phase_L(stg, enq);
      TR(2)
      if (grp == 1 && h + 1 < nh) wait_own(H + 1, enq ? P_L : 0);
      TR(3)
      barrier();                                          // X: phase 2h + 1, Y: phase 2h + 2
      TR(4)
      phase_C(enq);
      TR(5)
      if (enq) enqueue_end();
      ++H; stg = (stg + 1) & (NSTG - 1);
      TR(6)
    }
    if (grp == 0) barrier();                              // phase 2 nh: the partner's last C
    TR(7)
    big_epilogue<DT, BM, BN, TN, false>(p, acc, seq, 0, tiles_n, wmm, wn, l31, hi, gate);
    // the stores just issued are younger than every piece enqueued so far (full tiles only: a wave whose rows all lie
    // beyond M skips its stores, and an over-estimate here would under-wait)
    const int m_tile = seq / tiles_n;
    e_issued = issuedH;
    e_stores = ((m_tile + 1) * BM <= p.M) ? stores_per_tile : 0;
    TR(8)
  }
  TR_DUMP
}

int g_num_cu = 0;
int g_geom = -2;                                             // 0: one 8-wave 256-row workgroup per CU, 1: two 4-wave 128-row ones, 2: ping-pong

int num_cu() {
  if (g_num_cu == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    g_num_cu = n;
  }
  return g_num_cu;
}

template <int DT, int BM, int BN, int BKT, int NSTG, bool CONV, bool SPLIT = false, bool LNS = false>
int launch_big_cfg(const CoreParams& p, hipStream_t s, int splitk = 1) {
  if constexpr (!SPLIT && !LNS && BM == 256) {
    if (splitk > 1) return launch_big_cfg<DT, BM, BN, BKT, NSTG, CONV, true>(p, s, splitk);
  }
  if constexpr (!SPLIT && !LNS && !CONV && BM == 256 && BKT == 64) {
    if ((p.epi & IDF_EPI_LN_ROW) && !p.ln_stats) return launch_big_cfg<DT, BM, BN, BKT, NSTG, CONV, false, true>(p, s, 1);
  }
  void (*kern)(const CoreParams, const int, const int) = gemm_kernel_big<DT, BM, BN, BKT, NSTG, CONV, SPLIT, LNS>;
  constexpr int smem = NSTG * (BM + BN) * BKT * 2;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  CoreParams q = p;
  q.splitk = splitk; q.kt_per_slice = p.K / BKT / splitk;
  const int tiles = (p.N / BN) * ((p.M + BM - 1) / BM) * splitk;   // work items
  const int slots = num_cu() * (BM == 128 ? 2 : 1);
  const int grid = tiles < slots ? tiles : slots;
  // fill schedule (kernel comment): geometry 3 / 4 / 5 force skew 2 / 1 / 3; geometry 0 = the measured default per K
  const int geom = idf_big_geom();
  int skew = 0;
  if (geom == 3) skew = 2;
  else if (geom == 5) skew = 3;
  else if (geom == 0 || geom == 4 || geom == 6) skew = 1;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(BM * 2), smem, s, q, tiles, skew);
  return idf_launch_status();
}

template <int DT, int BN, bool CONV, int DL>
int launch_pp_cfg(const CoreParams& p, hipStream_t s) {
  void (*kern)(const CoreParams, const int) = gemm_kernel_pp<DT, BN, CONV, DL>;
  constexpr int smem = 4 * (256 + BN) * 32 * 2;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  CoreParams q = p;
  q.splitk = 1; q.kt_per_slice = p.K / 32;
  const int tiles = (p.N / BN) * ((p.M + 255) / 256);
  const int slots = num_cu();
  const int grid = tiles < slots ? tiles : slots;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), smem, s, q, tiles);
  return idf_launch_status();
}

int g_pp_dl = -1;

}  // namespace

std::atomic<long long> idf_stat_big_launches{0};

int idf_big_geom() {
  if (g_geom == -2) { const char* e = getenv("IDF_GEMM_GEOM"); g_geom = e ? atoi(e) : IDF_GEMM_GEOM_DEFAULT; }
  return g_geom;
}
int idf_big_set_geom(int v) { const int prev = idf_big_geom(); g_geom = v; return prev; }

// Shape gate + tile-width choice.  `force` skips the occupancy heuristic, not the shape rules.
int idf_launch_big(const CoreParams& p, int dtype, bool conv, bool force, hipStream_t s, int* splitk_out) {
  if (splitk_out) *splitk_out = 1;
  const bool geglu = (p.epi & IDF_EPI_GEGLU) != 0;
  if (p.K < 2 * BK || (p.K % BK) != 0) return IDF_BIG_UNSUPPORTED;
  if (p.epi & IDF_EPI_OUT_NCHW) return IDF_BIG_UNSUPPORTED;
  if (p.n_valid != p.N) return IDF_BIG_UNSUPPORTED;
  if ((p.lda % 8) || (p.ldw % 8)) return IDF_BIG_UNSUPPORTED;
  // the in-register epilogue uses 16-B vector accesses only
  if ((p.ldo % 8) || ((p.epi & IDF_EPI_RES) && (p.ldr % 8)) || ((p.epi & IDF_EPI_ROWBIAS) && (p.ld_rowbias % 8))) return IDF_BIG_UNSUPPORTED;
  if (!aligned16(p.out) || ((p.epi & IDF_EPI_RES) && !aligned16(p.res)) || ((p.epi & IDF_EPI_ROWBIAS) && !aligned16(p.rowbias)) ||
      ((p.epi & (IDF_EPI_BIAS | IDF_EPI_GEGLU)) && !aligned16(p.bias))) return IDF_BIG_UNSUPPORTED;
  if ((p.epi & (IDF_EPI_LN_ROW | IDF_EPI_LN_COL)) && (!aligned16(p.ln_c) || !aligned16(p.ln_stats))) return IDF_BIG_UNSUPPORTED;
  // self-normalising LN_ROW (no statistics passed): only the lock-step 256-row kernel computes them in its K loop
  const bool self_ln = (p.epi & IDF_EPI_LN_ROW) && !p.ln_stats;
  if (self_ln && (conv || idf_big_geom() == 1 || idf_big_geom() == 2 || idf_big_geom() == 6)) return IDF_BIG_UNSUPPORTED;
  const int geom = idf_big_geom();
  int bn = 0;
  if (geglu) bn = (p.N % 256 == 0) ? 256 : 0;
  else if (p.N % 320 == 0) bn = 320;
  else if (p.N % 256 == 0) bn = 256;
  else if (p.N % 128 == 0 && geom != 1 && geom != 2 && !self_ln) bn = 128;
  if (!bn) return IDF_BIG_UNSUPPORTED;
  const int bm = geom == 1 ? 128 : 256;
  const long long slots = (long long)num_cu() * (geom == 1 ? 2 : 1);
  const long long tiles = (long long)(p.N / bn) * ((p.M + bm - 1) / bm);
  // split-K: when the tile grid leaves most CUs idle and K is long (the 8x8-level convs and ff-out GEMMs: 64 tiles of
  // 180..360 K-tiles), S slices per tile (S | K-tiles, >= 16 K-tiles each) leave fp32 partials in the caller's workspace
  int splitk = 1;
  if (splitk_out && geom != 1 && geom != 2 && bn != 128 && !geglu && !self_ln && p.ws && tiles * 2 <= slots) {
    const int nkt = p.K / BK;
    for (int cand = (int)(slots / tiles); cand >= 2; --cand) {
      if (nkt % cand || nkt / cand < 16) continue;
      if ((size_t)cand * p.M * p.N * sizeof(float) > p.ws_bytes) continue;
      splitk = cand;
      break;
    }
  }
  if (!force) {
    // tile quantisation: a persistent workgroup slot processes ceil(items / slots) work items
    const long long items = tiles * splitk;
    const long long rounds = (items + slots - 1) / slots;
    const double eff = (double)items / (double)(rounds * slots);
    if (eff < 0.80) return IDF_BIG_UNSUPPORTED;
  }
  if (splitk_out) *splitk_out = splitk;
  {                                                       // the loader uses 32-bit element offsets
    const unsigned long long rows = conv ? (unsigned long long)(p.M / (p.Ho * p.Wo)) * p.Hin * p.Win : (unsigned long long)p.M;
    if (rows * (unsigned long long)p.lda >= (1ull << 31) || (unsigned long long)p.N * p.ldw >= (1ull << 31)) return IDF_BIG_UNSUPPORTED;
  }
  ++idf_stat_big_launches;
  if (g_pp_dl < 0) { const char* e = getenv("IDF_GEMM_PP_DL"); g_pp_dl = e ? atoi(e) : 3; }
#define IDF_PP_DISPATCH(DT, DLV)                                                                                          \
  {                                                                                                                       \
    if (conv) return bn == 320 ? launch_pp_cfg<DT, 320, true, DLV>(p, s) : launch_pp_cfg<DT, 256, true, DLV>(p, s);       \
    return bn == 320 ? launch_pp_cfg<DT, 320, false, DLV>(p, s) : launch_pp_cfg<DT, 256, false, DLV>(p, s);               \
  }
#define IDF_BIG_DISPATCH(DT)                                                                                              \
  if (geom == 2) {                                                                                                        \
    if (g_pp_dl == 0) IDF_PP_DISPATCH(DT, 0)                                                                              \
    IDF_PP_DISPATCH(DT, 3)                                                                                                \
  }                                                                                                                       \
  if (geom == 1) {                                                                                                        \
    if (conv) return bn == 320 ? launch_big_cfg<DT, 128, 320, 32, 2, true>(p, s) : launch_big_cfg<DT, 128, 256, 32, 3, true>(p, s);   \
    return bn == 320 ? launch_big_cfg<DT, 128, 320, 32, 2, false>(p, s) : launch_big_cfg<DT, 128, 256, 32, 3, false>(p, s);           \
  }                                                                                                                       \
  if (geom == 6 && bn == 256 && splitk == 1) {                                                                            \
    if (conv) return launch_big_cfg<DT, 256, 256, 32, 4, true>(p, s);                                                     \
    return launch_big_cfg<DT, 256, 256, 32, 4, false>(p, s);                                                              \
  }                                                                                                                       \
  if (bn == 128) {                       /* 128-wide tiles (the VAE's 128-channel convs at 512^2): 3 stages of 48 KB */  \
    if (conv) return launch_big_cfg<DT, 256, 128, 64, 3, true>(p, s);                                                     \
    return launch_big_cfg<DT, 256, 128, 64, 3, false>(p, s);                                                              \
  }                                                                                                                       \
  if (conv) return bn == 320 ? launch_big_cfg<DT, 256, 320, 64, 2, true>(p, s, splitk) : launch_big_cfg<DT, 256, 256, 64, 2, true>(p, s, splitk);     \
  return bn == 320 ? launch_big_cfg<DT, 256, 320, 64, 2, false>(p, s, splitk) : launch_big_cfg<DT, 256, 256, 64, 2, false>(p, s, splitk);
  if (dtype == IDF_BF16) { IDF_BIG_DISPATCH(IDF_BF16) }
  if (dtype == IDF_F16) { IDF_BIG_DISPATCH(IDF_F16) }
#undef IDF_BIG_DISPATCH
#undef IDF_PP_DISPATCH
  return IDF_E_UNSUPPORTED;
}
