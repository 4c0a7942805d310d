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
// K loop itself (LNS); 128-wide tiles with three stages.  What bounds the K loop (re-measured late in round 3, DESIGN.md
// "What bounds the GEMM family"): not the fill rate -- a wave moves 14 B/clk from L2 into LDS, a CU 64 B/clk, the 256 x 320
// tile needs 28.8 B/clk at 100 % MFMA (tools/ubench/dma_mix.hip; round 2's 5.6 / 34 were a ubench artefact) -- but power: the
// kernel's mix of MFMAs, fragment reads and LDS-DMA sustains 1.11 PFLOP/s at 1.35-1.48 GHz even without any synchronisation
// (tools/ubench/mfma_power.hip), pure MFMA 1.75-1.88 at 1.65-1.79 GHz (mfma_sustain.hip).
// Round 3: (a) the geometries that were measured slower (two 4-wave workgroups per CU, the role-split ping-pong kernel, the
// four-stage 32-deep ring, the end / spread fill schedules) left the library -- source snapshot
// profiles/archive_rejected_kernels/gemm_big_r02.hip, numbers profiles/r02_pp_*, r02_shape_profile_B64_fill_*; (b) fused q | k | v
// projection (VT): the output tiles whose columns lie at or beyond `vt_col0` run the SAME K loop with the MFMA operands
// swapped, so a lane ends up with 16 consecutive TOKENS of one channel and stores V transposed (V^T[channel][token], the
// layout the P.V MFMA's A operand wants) -- one launch reads the activation tile once for q, k and V^T instead of a second
// GEMM (M = C, whose 256-row tiles were 62 % padding at C = 320) re-reading it.
//
// Roofline: MFMA-bound (2.5 PFLOP/s dense bf16); algorithmic flops 2*M*N*K.  LDS image and XOR swizzle are the ones
// of gemm_kernel_dma (linear 128-B rows, 16-B slot ^= (row >> 1) & 7 applied on the global source address).
#include "gemm_core.h"
#include <cstdlib>

using namespace idfcore;

namespace {

__device__ __attribute__((aligned(128))) unsigned short idf_zero_page[64];   // zero-initialised device memory

// Wave tile: IDF_WAVE_ROWS rows x BN/2 columns.  64 (library): eight waves per workgroup, two per SIMD, 160 accumulator
// registers per lane.  128 (experiment of tools/ubench/big_trace.hip, see profiles/DESIGN_r01_r05_full.md "What bounds the GEMM family"): four waves,
// one per SIMD with the whole 512-register budget -- 9 fragment reads per 20 MFMAs instead of 7 per 10.
#ifndef IDF_WAVE_ROWS
#define IDF_WAVE_ROWS 64
#endif
constexpr int WM = IDF_WAVE_ROWS, TM = WM / 32;
constexpr int NWAVES = 2 * 256 / WM;          // waves per 256-row workgroup tile (8 or 4)

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
// STATS: by-product for a LayerNorm that follows (idf_gemm's out_stats) -- (mean, M2) of the 16-bit-ROUNDED values this wave
// stores of each of its rows (BN/2 columns), accumulated around a per-lane pivot (no E[x^2] - mu^2 cancellation), the two
// lane halves of a row merged with Chan's formula, one 8-B store per row and wave into p.stat_parts; a 16-B-per-row
// finalize kernel replaces the pass that re-read the whole output (65 of them per forward in round 2).
// GLU: the kernel instantiation that serves ONLY period-32 GEGLU launches on the 5-fragment (320-wide) wave tile: compiled with
// nothing but that branch -- carrying it as a run-time branch in the other 320-wide kernels cost them ~290 spilled VGPRs.
// ---- coalesced epilogue accesses (round 4).  After the v_permlane32_swap exchange a lane owns 16 consecutive columns of ITS
// row, so a 16-B store / residual load instruction touches 32 different rows in 64 separate 16-B pieces: 64 requests per
// instruction, and the CU's vector-memory path retires about one request per cycle -- 8 waves x 20 stores x 64 requests =
// 10 k cycles per tile, which IS the 13-20 k-cycle epilogue of the cycle traces (52 k with the residual loads), whatever the
// HBM does (de-phasing the workgroups changed nothing: profiles/r04_big_sched_walk_dephase_epivm.log).  Now a 32 x 32 fragment
// (2 KB of 16-bit values) goes through a wave-private 2-KB LDS slot -- the 16 KB the two 72-KB ring stages leave of the CU's
// 160 KB -- and is stored (the residual: loaded) with 4 adjacent lanes on the 64 contiguous bytes of one row: 16 requests per
// instruction.  Same values, same arithmetic, same instruction count; wave-private, so still no workgroup barrier.
// Slot image: [32 rows][64 B], 16-B piece ^= f(row) with f = ((b2 ^ b3) << 1) | (b1 ^ b3 ^ b4) of the row's bits: both access
// patterns (lane = row with two pieces: "math layout"; 4 lanes per row: "store layout") are bank-conflict-free for
// ds_read_b128's 16-lane groups and ds_write_b128's 8-lane groups.
// DEFAULT OFF: measured -0.1 ... -8 % (profiles/r04_big_stage_epilogue.log) -- the ~12 B/clk per CU at which an epilogue
// drains is not a request-count limit -- so build.sh does not define IDF_EPI_STAGE and the GPU suite does not cover the staged
// path; it is kept as the A/B build `tools/ubench/big_sched_stage1` measured.  The wave-private 2-KB slots themselves are NOT
// dead weight: the fused q | k | v projection parks its in-loop LayerNorm statistics there (LNS + VT) and, since round 5, the GST
// conv epilogue passes its 2 x 160 column totals through them.
#ifndef IDF_EPI_STAGE
#define IDF_EPI_STAGE 0
#endif
__device__ __forceinline__ int stg_f(int row) {
  return ((((row >> 2) ^ (row >> 3)) & 1) << 1) | (((row >> 1) ^ (row >> 3) ^ (row >> 4)) & 1);
}
// piece `pc` (0..3) of row `row` of the slot
__device__ __forceinline__ u32x4* stg_at(char* stg, int row, int pc) {
  return reinterpret_cast<u32x4*>(stg + row * 64 + ((pc ^ stg_f(row)) << 4));
}

// GST (round 5): GroupNorm statistics of the tile's OUTPUT as a by-product ("statistics out of the producer's
// epilogue") -- per wave, for its 64 rows x BN/2 columns, one (mean, M2) pair per 32-group GroupNorm group into
// p.gn_partial[sample][64-row chunk][group][2]: exactly the partial-summary layout gn_apply merges (norms.hip), so the
// gn_stats pass over the tensor -- one full read of every conv output -- does not run.  From the 16-bit-ROUNDED values the
// wave stores.  The accumulator layout is lane <-> row, register <-> column, so a group sum is a cross-LANE sum: per 32-column
// fragment pair (b = 0, 1 combined in-lane) the lane's 16 shifted values d = y - bias[col] and their squares go through a
// 5-step reduce-scatter over the 32 lanes of a half (31 ds_bpermute + 93 VALU), after which lane l holds the 64-row total of
// d (l < 16) or d^2 (l >= 16) of column 16 hi + (l & 15); the 2 x 160 column totals pass through the wave's private 2-KB LDS
// slot and 160 / cpg lanes merge their group's columns (equal counts: mean of means, M2 = sum M2_c + 64 sum (mean_c - mean)^2).
// ~1100 VALU / LDS operations per wave and tile: 0.5-2 % of a 3x3 conv tile (90-180 K-tiles).
template <int N, int L>
__device__ __forceinline__ void gst_scatter_step(float* t, int l31) {
  // N values per lane -> N/2: the lane keeps the half its bit L/... selects and adds its partner's copy of that half
  constexpr int H = N / 2;
  const bool up = (l31 & H) != 0;
#pragma unroll
  for (int i = 0; i < H; ++i) {
    const float keep = up ? t[i + H] : t[i];
    const float send = up ? t[i] : t[i + H];
    t[i] = keep + __shfl_xor(send, H, 64);
  }
}
template <int DT, int BM, int BN, int TN, bool SPLIT, bool LNS = false, bool STATS = false, bool GLU = false, bool GST = false>
__device__ __forceinline__ void big_epilogue(const CoreParams& p, f32x16 (&acc)[TN][TM], int seq, int slice, int tiles_n, int wm,
                                             int wn, int l31_in, int hi_in, float gate, char* stg_in, const float* lnm = nullptr,
                                             const float* lnr = nullptr) {
  constexpr int WN = BN / 2;
  constexpr bool STG = IDF_EPI_STAGE != 0;
  const int epi = p.epi;
  // opaque copies: everything the epilogue derives from the lane id (row / column offsets, slot addresses) is then computed
  // HERE, per tile, instead of being hoisted out of the tile loop and kept alive -- or spilled -- across the K loop
  int l31 = l31_in, hi = hi_in;
  asm volatile("" : "+v"(l31), "+v"(hi));
  char* const stg = stg_in;
  const int sl_row = (l31 + 32 * hi) >> 2, sl_pc = l31 & 3;       // store layout: lane -> row sl_row (+ 16), piece sl_pc
  // 32 rows x 64 B from the slot to rows m_base.. of a 16-bit matrix (column c0), 16 rows x 64 contiguous bytes per instruction
  auto stg_store64 = [&](unsigned short* base, int ld, int m_base, int c0) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = sl_row + 16 * i;
      const u32x4 t = *stg_at(stg, row, sl_pc);
      if (m_base + row < p.M) *reinterpret_cast<u32x4*>(base + (size_t)(m_base + row) * ld + c0 + sl_pc * 8) = t;
    }
  };
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
        float* o = p.ws + ((size_t)slice * p.tail_rows + (m - p.tail_m0)) * p.N + n;
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(o + 4 * j) = f32x4{v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]};
      }
    });
    return;
  }
  if (GLU || (((TN & 1) == 0) && (epi & IDF_EPI_GEGLU) && (epi & IDF_EPI_GEGLU_P32))) {
   if constexpr (GLU || ((TN & 1) == 0)) {
    // [16 value | 16 gate] per 32 packed rows: fragment a holds value (registers q = 0, 1) and gate (q = 2, 3) of 16 outputs
    // in the same lane.  One v_permlane32_swap per register pair (q = 0 of the upper lanes <-> q = 1 of the lower lanes) leaves
    // a lane with 8 consecutive output columns: one 16-B store.
#pragma unroll
    for (int b = 0; b < TM; ++b) {
     static_for<0, TN, 1>([&](auto AI) {
      constexpr int a = decltype(AI)::value;
      const int npk = nw + a * 32;
      {
        float o[8];
        f32x2 st = {0.0f, 1.0f};
        if (epi & IDF_EPI_LN_ROW) {
          const int mr = min(mw + b * 32 + l31, p.M - 1);
          if constexpr (LNS) st = f32x2{lnm[b], lnr[b]};
          else st = *reinterpret_cast<const f32x2*>(p.ln_stats + 2 * (size_t)mr);
        }
        // bias / c vectors are (re)loaded per register group: keeping all of a fragment's 8 vectors live next to the 160
        // accumulators spilled ~300 VGPRs (scratch is HBM traffic, see attention4.hip)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + npk + 8 * q + 4 * hi);
          const f32x4 bg = *reinterpret_cast<const f32x4*>(p.bias + npk + 16 + 8 * q + 4 * hi);
          f32x4 cv = {0.f, 0.f, 0.f, 0.f}, cg = {0.f, 0.f, 0.f, 0.f};
          if (epi & IDF_EPI_LN_ROW) {
            cv = *reinterpret_cast<const f32x4*>(p.ln_c + npk + 8 * q + 4 * hi);
            cg = *reinterpret_cast<const f32x4*>(p.ln_c + npk + 16 + 8 * q + 4 * hi);
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            // the same operation sequence as the period-64 branch below: the two packings give bit-identical outputs
            float val = acc[a][b][4 * q + e], gat = acc[a][b][4 * (q + 2) + e];
            if (epi & IDF_EPI_LN_ROW) {
              val = fmaf(st[1], fmaf(-st[0], cv[e], val), bv[e]);
              gat = fmaf(st[1], fmaf(-st[0], cg[e], gat), bg[e]);
            } else {
              val += bv[e];
              gat += bg[e];
            }
            o[4 * q + e] = val * gelu_erf_f(gat);
          }
        }
        float v[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(o[e]), __float_as_uint(o[4 + e]), false, false);
          v[e] = __uint_as_float(sw[0]); v[4 + e] = __uint_as_float(sw[1]);
        }
        const int m = mw + b * 32 + l31;
        unsigned short* const obase = reinterpret_cast<unsigned short*>(p.out);
        if constexpr (STG) {
          // a fragment leaves 32 B per row (the lane's 8 columns + its partner half's): two fragments fill the slot's 64-B rows
          *stg_at(stg, l31, 2 * (a & 1) + hi) = pack8<DT>(v);
          if constexpr ((a & 1) == 1) {
            stg_store64(obase, p.ldo, mw + b * 32, (nw >> 1) + (a - 1) * 16);
          } else if constexpr (a == TN - 1) {                 // odd fragment count: the last one goes out alone, 32 B per row
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const int lane = l31 + 32 * hi, row = lane >> 1, pc = lane & 1;
            const u32x4 t = *stg_at(stg, row, pc);
            if (mw + b * 32 + row < p.M)
              *reinterpret_cast<u32x4*>(obase + (size_t)(mw + b * 32 + row) * p.ldo + (nw >> 1) + a * 16 + pc * 8) = t;
          }
        } else if (m < p.M) {
          unsigned short* op = obase + (size_t)m * p.ldo + (npk >> 1) + 8 * hi;
          *reinterpret_cast<u32x4*>(op) = pack8<DT>(v);
        }
      }
     });
    }
   }
  } else if (!GLU && (epi & IDF_EPI_GEGLU)) {
    if constexpr (!GLU && (TN & 1) == 0) {
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
          if constexpr (STG) {
            *stg_at(stg, l31, 2 * hi) = pack8<DT>(v);
            *stg_at(stg, l31, 2 * hi + 1) = pack8<DT>(v + 8);
            stg_store64(reinterpret_cast<unsigned short*>(p.out), p.ldo, mw + b * 32, npk >> 1);
          } else if (m < p.M) {
            unsigned short* op = reinterpret_cast<unsigned short*>(p.out) + (size_t)m * p.ldo + (npk >> 1) + 16 * hi;
            *reinterpret_cast<u32x4*>(op) = pack8<DT>(v);
            *reinterpret_cast<u32x4*>(op + 8) = pack8<DT>(v + 8);
          }
        }
      });
    }
  } else if constexpr (!GLU) {
    float st_p[TM], st_s1[TM], st_s2[TM];                  // STATS: pivot, sum (x - p), sum (x - p)^2 per row fragment
#pragma unroll
    for (int b = 0; b < TM; ++b) { st_p[b] = 0.0f; st_s1[b] = 0.0f; st_s2[b] = 0.0f; }
    float gacc[GST ? TN : 1];                               // GST: per fragment column a, this lane's column total (see above)
    static_for<0, TN, 1>([&](auto AI) {
      constexpr int a = decltype(AI)::value;
      const int n = nw + a * 32 + 16 * hi;
      f32x4 bs[4], cs[4];
      float gt[GST ? 32 : 1];                               // GST: d and d^2 of the lane's 16 columns, summed over b
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
        const int mc = min(m, p.M - 1);             // rows beyond M (last m-tile) compute on row M-1's side inputs, store nothing
        if (epi & IDF_EPI_LN_ROW) {                 // v = rstd_m * (acc - mu_m * c[n]); the beta term arrives as bias
          f32x2 st;
          if constexpr (LNS) st = f32x2{lnm[b], lnr[b]};
          else st = *reinterpret_cast<const f32x2*>(p.ln_stats + 2 * (size_t)mc);
#pragma unroll
          for (int j = 0; j < 16; ++j) v[j] = st[1] * fmaf(-st[0], cs[j >> 2][j & 3], v[j]);
        }
        if (epi & IDF_EPI_LN_COL) {                 // v = rstd_n * (acc - c[m] * mu_n) + d[m]: 16 token columns of row m
          const float cm = p.ln_c[mc], dm = p.ln_d[mc];
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
        [[maybe_unused]] float rbv[16];             // GST: the row bias of the wave's sample is part of the statistics' pivot
        if (epi & IDF_EPI_ROWBIAS) {
          const unsigned short* rb = p.rowbias + (size_t)(mc / p.rows_per_batch) * p.ld_rowbias + n;
          float r[16];
          unpack8<DT>(*reinterpret_cast<const u32x4*>(rb), r);
          unpack8<DT>(*reinterpret_cast<const u32x4*>(rb + 8), r + 8);
#pragma unroll
          for (int j = 0; j < 16; ++j) v[j] += r[j];
          if constexpr (GST) {
#pragma unroll
            for (int j = 0; j < 16; ++j) rbv[j] = r[j];
          }
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
          const float gm = (epi & IDF_EPI_GATE) ? gate : 1.0f;
          float r[16];
          if constexpr (STG) {
            // the fragment's 32 x 64 B of the residual: 16 rows x 64 contiguous bytes per load instruction, then each lane
            // picks its own row's 32 B out of the slot
#pragma unroll
            for (int i = 0; i < 2; ++i) {
              const int row = sl_row + 16 * i;
              const int mr = min(mw + b * 32 + row, p.M - 1);
              *stg_at(stg, row, sl_pc) = *reinterpret_cast<const u32x4*>(p.res + (size_t)mr * p.ldr + nw + a * 32 + sl_pc * 8);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            unpack8<DT>(*stg_at(stg, l31, 2 * hi), r);
            unpack8<DT>(*stg_at(stg, l31, 2 * hi + 1), r + 8);
          } else {
            const unsigned short* rr = p.res + (size_t)mc * p.ldr + n;
            unpack8<DT>(*reinterpret_cast<const u32x4*>(rr), r);
            unpack8<DT>(*reinterpret_cast<const u32x4*>(rr + 8), r + 8);
          }
#pragma unroll
          for (int j = 0; j < 16; ++j) v[j] = fmaf(gm, v[j], r[j]);
        }
        if (epi & IDF_EPI_OUT_F32) {
          if (m < p.M) {
            float* o = reinterpret_cast<float*>(p.out) + (size_t)m * p.ldo + n;
#pragma unroll
            for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(o + 4 * j) = f32x4{v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]};
          }
        } else {
          const u32x4 q0 = pack8<DT>(v), q1 = pack8<DT>(v + 8);
          if constexpr (STG) {
            asm volatile("" ::: "memory");            // the residual reads of the slot above are done before it is rewritten
            *stg_at(stg, l31, 2 * hi) = q0;
            *stg_at(stg, l31, 2 * hi + 1) = q1;
            stg_store64(reinterpret_cast<unsigned short*>(p.out), p.ldo, mw + b * 32, nw + a * 32);
          } else if (m < p.M) {
            unsigned short* o = reinterpret_cast<unsigned short*>(p.out) + (size_t)m * p.ldo + n;
            *reinterpret_cast<u32x4*>(o) = q0;
            *reinterpret_cast<u32x4*>(o + 8) = q1;
          }
          if constexpr (GST) {
            float r[16];
            unpack8<DT>(q0, r);
            unpack8<DT>(q1, r + 8);
#pragma unroll
            for (int j = 0; j < 16; ++j) {
              // pivot of the shifted sums: conv bias + the sample's row bias (the time embedding of a ResBlock's first conv) --
              // the per-column constants that move a channel's mean away from 0 (with the bias alone a 64-row chunk
              // whose mean is ~100 standard deviations lost 2^-22 (mean / std)^2 of its M2 to cancellation)
              float d = (epi & IDF_EPI_BIAS) ? r[j] - bs[j >> 2][j & 3] : r[j];
              if (epi & IDF_EPI_ROWBIAS) d -= rbv[j];
              if (b == 0) { gt[j] = d; gt[16 + j] = d * d; }
              else { gt[j] += d; gt[16 + j] = fmaf(d, d, gt[16 + j]); }
            }
          }
          if constexpr (STATS) {
            float r[16];
            unpack8<DT>(q0, r);
            unpack8<DT>(q1, r + 8);
            if (a == 0) st_p[b] = r[0];
#pragma unroll
            for (int j = 0; j < 16; ++j) { const float dlt = r[j] - st_p[b]; st_s1[b] += dlt; st_s2[b] = fmaf(dlt, dlt, st_s2[b]); }
          }
        }
      }
      if constexpr (GST) {
        gst_scatter_step<32, 0>(gt, l31);
        gst_scatter_step<16, 0>(gt, l31);
        gst_scatter_step<8, 0>(gt, l31);
        gst_scatter_step<4, 0>(gt, l31);
        gst_scatter_step<2, 0>(gt, l31);
        gacc[a] = gt[0];
      }
    });
    if constexpr (GST) {
      // column totals -> the wave's LDS slot: cs[kind][column of the wave's BN/2], kind 0 = sum d, 1 = sum d^2
      float* const csl = reinterpret_cast<float*>(stg);
      constexpr int WNC = BN / 2;
#pragma unroll
      for (int a = 0; a < TN; ++a) csl[(l31 >> 4) * WNC + a * 32 + 16 * hi + (l31 & 15)] = gacc[a];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      const int cpg = p.N >> 5;                             // channels per group (32 groups); WNC % cpg == 0 (dispatcher)
      const int g = l31 + 32 * hi;
      if (g * cpg < WNC && mw < p.M) {
        constexpr float rows = (float)WM, inv_rows = 1.0f / (float)WM;
        const float inv_cpg = 1.0f / (float)cpg;
        float ms = 0.f, q = 0.f;
        for (int j = 0; j < cpg; ++j) {
          const int col = g * cpg + j;
          const float s1 = csl[col], s2 = csl[WNC + col];
          float pc = (epi & IDF_EPI_BIAS) ? p.bias[nw + col] : 0.0f;
          if (epi & IDF_EPI_ROWBIAS)              // (the 64 rows of a wave tile belong to one sample: HW % 64 == 0, dispatcher)
            pc += Elem<DT>::to_f32(p.rowbias[(size_t)(mw / p.rows_per_batch) * p.ld_rowbias + nw + col]);
          const float dm = s1 * inv_rows;
          ms += pc + dm;
          q += fmaxf(s2 - s1 * dm, 0.0f);
        }
        const float mean = ms * inv_cpg;
        float dev = 0.f;
        for (int j = 0; j < cpg; ++j) {
          const int col = g * cpg + j;
          float pc = (epi & IDF_EPI_BIAS) ? p.bias[nw + col] : 0.0f;
          if (epi & IDF_EPI_ROWBIAS)              // (the 64 rows of a wave tile belong to one sample: HW % 64 == 0, dispatcher)
            pc += Elem<DT>::to_f32(p.rowbias[(size_t)(mw / p.rows_per_batch) * p.ld_rowbias + nw + col]);
          const float dd = pc + csl[col] * inv_rows - mean;
          dev = fmaf(dd, dd, dev);
        }
        const int smp = mw / p.gn_hw, chunk = (mw - smp * p.gn_hw) / WM, nch = p.gn_hw / WM;
        *reinterpret_cast<f32x2*>(p.gn_partial + (((size_t)smp * nch + chunk) * 32 + nw / cpg + g) * 2) =
            f32x2{mean, fmaf(rows, dev, q)};
      }
      __builtin_amdgcn_wave_barrier();                      // the slot is reused by this wave's next tile
    }
    if constexpr (STATS) {
      constexpr float cnt = 16.0f * TN;                      // values per lane and row: half of the wave's BN/2 columns
      const int n_tile = (seq - m_tile * tiles_n);
#pragma unroll
      for (int b = 0; b < TM; ++b) {
        const float mean = st_p[b] + st_s1[b] * (1.0f / cnt);
        const float m2 = fmaxf(st_s2[b] - st_s1[b] * st_s1[b] * (1.0f / cnt), 0.0f);
        const float mean_o = __shfl_xor(mean, 32, 64), m2_o = __shfl_xor(m2, 32, 64);
        const float dlt = mean_o - mean;
        const int m = mw + b * 32 + l31;
        if (hi == 0 && m < p.M)
          *reinterpret_cast<f32x2*>(p.stat_parts + ((size_t)m * p.parts + n_tile * 2 + wn) * 2) =
              f32x2{0.5f * (mean + mean_o), m2 + m2_o + dlt * dlt * (0.5f * cnt)};
      }
    }
  }
}

// Epilogue of a TRANSPOSED tile (fused q | k | v projection, columns >= vt_col0): the K loop ran with the MFMA operands
// swapped, acc[a][b][4q+e] = D[m = b*32 + 8q + 4hi + e][n = a*32 + l31]; two v_permlane32_swap per register pair now leave a
// lane with 8 consecutive TOKENS of channel n = a*32 + l31 per half (q in {0,2}: tokens 16*hi + [0,8), q in {1,3}: + [8,16)):
// one 16-B store into V^T[n - vt_col0][m ..] each.  LayerNorm folded in as in LN_ROW, but here the statistics belong to the
// register dimension: 8 (mu, rstd) pairs per half, read as 4 float4 from `stats` -- the caller's array, or (LNS) the
// wave's own LDS copy of what it summed in the K loop -- per (a, b, half) so that no more than 16 of them are live next
// to the 160 accumulators; c[n] and the beta / bias term d[n] are per-lane scalars.  Requires M % 16 == 0.
template <int DT, int BM, int BN, int TN>
__device__ __forceinline__ void big_epilogue_vt(const CoreParams& p, f32x16 (&acc)[TN][TM], int seq, int tiles_n, int wm, int wn,
                                                int l31, int hi, const float* stats /* (mu, rstd) pairs */, int stats_row0) {
  constexpr int WN = BN / 2;
  const int epi = p.epi;
  const int m_tile = seq / tiles_n;
  const int n0 = (seq - m_tile * tiles_n) * BN, m0 = m_tile * BM;
  const int mw = m0 + wm * WM, nw = n0 + wn * WN;
  static_for<0, TN, 1>([&](auto AI) {
    constexpr int a = decltype(AI)::value;
    const int n = nw + a * 32 + l31;
    const float cn = (epi & IDF_EPI_LN_ROW) ? p.ln_c[n] : 0.0f;
    const float bn = (epi & IDF_EPI_BIAS) ? p.bias[n] : 0.0f;
    unsigned short* orow = p.vt_out + (size_t)(n - p.vt_col0) * p.ld_vt;
#pragma unroll
    for (int b = 0; b < TM; ++b) {
#pragma unroll
      for (int half = 0; half < 2; ++half) {                 // registers q = half, half + 2
        float v[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[a][b][4 * half + e]),
                                                           __float_as_uint(acc[a][b][4 * half + 8 + e]), false, false);
          v[e] = __uint_as_float(sw[0]); v[4 + e] = __uint_as_float(sw[1]);
        }
        const int mb = mw + b * 32 + 16 * hi + 8 * half;      // first of these 8 tokens
        if (epi & IDF_EPI_LN_ROW) {
          const f32x4* st4 = reinterpret_cast<const f32x4*>(stats + 2 * (size_t)(min(mb, p.M - 8) - stats_row0));
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const f32x4 t = st4[j];                           // (mu, rstd) of tokens mb + 2j, mb + 2j + 1
            v[2 * j] = t[1] * fmaf(-t[0], cn, v[2 * j]);
            v[2 * j + 1] = t[3] * fmaf(-t[2], cn, v[2 * j + 1]);
          }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += bn;
        if (mb < p.M) *reinterpret_cast<u32x4*>(orow + mb) = pack8<DT>(v);
      }
    }
  });
}

// Optional per-segment cycle trace (tools/ubench/big_trace.hip builds this file with -DIDF_BIG_TRACE; the library build has
// none of it): s_memtime deltas summed per segment by waves 0 (early filler) and 4 (late filler) of the first and of a
// middle workgroup.  Segments: 0 vmcnt wait, 1 barrier, 2 early K-tile enqueue, 3 fragment reads + MFMA issue (+ late
// enqueue), 4 epilogue, 5 tile head (accumulator clear), 6 number of K-tiles, 7 number of tiles.
#ifdef IDF_BIG_TRACE
__device__ unsigned long long idf_big_trace_buf[4][8];
#define TR_DECL unsigned long long tr_last = __builtin_readcyclecounter(), tr_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define TR(i) { const unsigned long long tr_now = __builtin_readcyclecounter(); tr_acc[i] += tr_now - tr_last; tr_last = tr_now; }
#define TR_COUNT(i) { tr_acc[i] += 1; }
#define TR_DUMP { const int trb = blockIdx.x == 0 ? 0 : ((int)blockIdx.x == (int)gridDim.x / 2 ? 1 : -1);                     \
    if (trb >= 0 && lane == 0 && (wave == 0 || wave == NWAVES / 2)) { for (int i = 0; i < 8; ++i) idf_big_trace_buf[trb * 2 + (wave ? 1 : 0)][i] = tr_acc[i]; } }
#else
#define TR_DECL
#define TR(i)
#define TR_COUNT(i)
#define TR_DUMP
#endif

// Geometry: BM x BN output tile, (BM/64) x 2 waves (wave tile 64 x BN/2), K-tile BKT, NSTG-stage LDS ring.
//   <256, {320,256}, 64, 2>: ONE 8-wave workgroup per CU (2 x 72 KB stages); <256, 128, 64, 3>: 3 x 48 KB stages.
template <int DT, int BM, int BN, int BKT, int NSTG, bool CONV, bool SPLIT, bool LNS = false, bool VT = false, bool STATS = false,
          bool GLU = false, bool GST = false>
__global__ __launch_bounds__(64 * NWAVES, 1) void gemm_kernel_big(const CoreParams p, const int tiles_total) {
  constexpr int WN = BN / 2, TN = WN / 32;
  constexpr int NW = NWAVES;                               // waves per workgroup: (BM / WM) x 2
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
  // Tile walk.  Strided (rounds 1-3): workgroup slot s takes tiles s, s + G, ... -- the eight workgroups that share an
  // activation m-tile fetch it at the same moment, one misses to HBM and all eight wait for that line.  Chunked (round 4,
  // tile_walk = 1, unsplit launches): slot s takes the CONTIGUOUS tiles [s q + min(s, r), ...) of the n-fastest list, so a
  // workgroup walks the n-tiles of ONE m-tile back to back: the activation rows come from HBM once and from L2 for the
  // other tiles_n - 1 tiles, and the weight n-tile every workgroup streams at a given moment is the same one (L2-hot).
  // Either walk runs every tile through the same arithmetic: outputs are bit-identical.
  int seq_step = G, seq_end = tiles_total, seq0 = slot;
  if (!SPLIT && p.tile_walk == 1 && tiles_total > G) {
    const int q = tiles_total / G, r = tiles_total - q * G;
    seq0 = slot * q + min(slot, r);
    seq_end = seq0 + q + (slot < r ? 1 : 0);
    seq_step = 1;
  }
  // split-K (p.splitk > 1): work item seq = tile * splitk + slice covers K-tiles [slice * nk, (slice + 1) * nk) of its tile
  // and leaves fp32 partials in p.ws[slice][M][N] (reduced + epilogue by splitk_reduce_kernel)
  // Hybrid (round 3): a tile count that is not close to a whole number of 256-CU rounds used to send the launch to the 128^2
  // kernels (e.g. 288 tiles at 18 rows: 1.125 rounds); now the first `full` items are whole tiles and only the REMAINING
  // tiles are cut into K-slices, one slice per otherwise idle workgroup (uniform split-K is the case full = 0).
  const int S = SPLIT ? p.splitk : 1;
  const int nk = p.kt_per_slice;                            // K-tiles per split item (= K / BKT without split-K)
  const int full = SPLIT ? p.full_items : tiles_total;      // leading whole-tile items
  auto item_map = [&](int item, int& tile, int& slice, int& nki) {
    if (!SPLIT || item < full) { tile = item; slice = 0; nki = SPLIT ? p.kt_full : nk; }
    else { const int r = item - full; const int t = r / S; tile = full + t; slice = r - t * S; nki = nk; }
  };

  // ---------------- loader state (runs one K-tile ahead of the MFMAs, across output-tile boundaries)
  const int dr = lane / CPR, dc = lane % CPR;
  auto swz = [](int row) { return CPR == 8 ? ((row >> 1) & 7) : ((row >> 2) & 3); };   // conflict-free ds_read_b128 (see header)
  unsigned woff[W_INST], aoff[A_INST];
  int ayx[A_INST];                                        // conv: (yo*stride-1) << 16 | (xo*stride-1) & 0xffff
  int l_seq, l_kt = 0, l_k0 = 0, l_nk = 0, tap = 0, ci0 = 0;

  auto setup_loader = [&](int item) {
    int tile, slice;
    item_map(item, tile, slice, l_nk);
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
        const int y0 = yo * p.stride - 1, x0 = xo * p.stride - 1;
        if (p.up == 0) {
          // no upsampling (all but three convs of a forward): the tap only ADDS a wave-uniform (ky * Win + kx) * lda to the
          // element offset of the row's window origin, and whether a tap falls into the zero padding is one of 9 bits worked
          // out here, once per output tile -- 8 VALU instructions per piece in the K loop instead of 25-30
          int mask = 0;
#pragma unroll
          for (int t = 0; t < 9; ++t) {
            const int yi = y0 + t / 3, xi = x0 + t % 3;
            mask |= ((yi >= 0) & (yi < p.Hin) & (xi >= 0) & (xi < p.Win)) << t;
          }
          ayx[j] = mask;
          aoff[j] = (unsigned)b * (unsigned)(p.Hin * p.Win) * (unsigned)p.lda + (unsigned)((y0 * p.Win + x0) * p.lda) + sw;
        } else {
          ayx[j] = (y0 << 16) | (x0 & 0xffff);
          aoff[j] = (unsigned)b * (unsigned)(p.Hin * p.Win) * (unsigned)p.lda + sw;
        }
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
      if (CONV && p.up == 0) {
        const int ky = tap / 3, kx = tap - ky * 3;
        const unsigned tapoff = (unsigned)((ky * p.Win + kx) * p.lda + ci0);          // wave-uniform
        const bool ok = (ayx[j] >> tap) & 1;
        const unsigned short* src = ok ? p.A + (aoff[j] + tapoff) : idf_zero_page + dc * 8;
        dma16_v(src, dst);
      } else if (CONV) {
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
    if (++l_kt == l_nk) {
      l_kt = 0;
      l_seq += seq_step;
      if (l_seq < seq_end) setup_loader(l_seq);
    }
  };
  auto issue_dma = [&](int stage) {                       // enqueue the whole K-tile into `stage`, then advance
    static_for<0, DPW, 1>([&](auto II) { issue_piece(stage, II); });
    advance_loader();
  };

  // ---------------- MFMA side
  const int f_sw = swz(l31);                              // fragment rows are (multiple of 32) + l31
  int issued = 0;                                         // K-tiles enqueued so far
  constexpr int NPOS = (BKT / 16) * TN;                   // (k-step, weight fragment) positions of a K-tile: TM MFMAs each
  // Fill schedule (see the K loop).  OLD_LATE: which half of the workgroup enqueues the next K-tile from inside its MFMAs --
  // the OLDER waves 0..3 (320-wide tiles, at 7/8 of their MFMAs) or the younger waves 4..7 (at 1/2: the round-2 schedule,
  // kept for the 256- and 128-wide tiles, where the new one measured no better).  -DIDF_LATE_OLD / _NUM / _DEN: the
  // experiment knobs of tools/ubench/big_trace.hip.
#ifdef IDF_LATE_NUM
  constexpr bool OLD_LATE = IDF_LATE_OLD != 0;
  constexpr int LATE_POS = NPOS * IDF_LATE_NUM / IDF_LATE_DEN;
#else
  constexpr bool OLD_LATE = BN == 320;
  constexpr int LATE_POS = OLD_LATE ? NPOS * 7 / 8 : NPOS / 2;
#endif
  // one wave per SIMD (NW == 4): there is no partner to hide a burst under, every wave enqueues one piece per position
  constexpr bool SPREAD = NW == 4;
#ifdef IDF_SPREAD_PPP
  constexpr int PPP = IDF_SPREAD_PPP;                      // pieces per position of the spread fill (from position 0 on)
#else
  constexpr int PPP = (DPW + NPOS - 1) / NPOS;
#endif
  // `late_fill`: this wave enqueues the next K-tile's LDS-DMA pieces from the MIDDLE of its MFMAs (see the K loop).
  // SWAPT: MFMA operands swapped (transposed-V tiles of the fused q | k | v projection): acc[a][b] then holds
  // D[m = b*32 + 8q + 4hi + e][n = a*32 + l31] -- a lane owns a channel, its registers run over tokens.
  auto compute = [&](auto SWAPT, f32x16 (&acc)[TN][TM], float (&lsx)[TM], float (&lsq)[TM], int stage, bool late_fill, int st_fill) {
    constexpr bool SWAP = decltype(SWAPT)::value;
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
        if constexpr (NW == 4) {
          // accumulator fragments 0..15 (256 registers) pinned to the AGPR file, 16..19 to the VGPR file
          static_for<0, TM, 1>([&](auto BI) {
            constexpr int b = decltype(BI)::value;
            constexpr bool AG = a * TM + b < 16;
            if constexpr (SWAP) Elem<DT>::template mfma32_pin<AG>(af[cur][b], wf[cur][a], acc[a][b]);
            else Elem<DT>::template mfma32_pin<AG>(wf[cur][a], af[cur][b], acc[a][b]);
          });
        } else {
#pragma unroll
          for (int b = 0; b < TM; ++b) {
            if constexpr (SWAP) acc[a][b] = Elem<DT>::mfma32(af[cur][b], wf[cur][a], acc[a][b]);
            else acc[a][b] = Elem<DT>::mfma32(wf[cur][a], af[cur][b], acc[a][b]);
          }
        }
        if constexpr (LNS && a == 0) {                      // row sums of the A fragments this k-step multiplies: 16 VALU ops
#pragma unroll
          for (int b = 0; b < TM; ++b)
#pragma unroll
            for (int w = 0; w < 4; ++w) {
              Elem<DT>::dot2c(lsx[b], af[cur][b][w], Elem<DT>::ONES2);
              Elem<DT>::dot2c(lsq[b], af[cur][b][w], af[cur][b][w]);
            }
        }
        if constexpr (SPREAD) {
          if constexpr (pos * PPP < DPW) {
            if (late_fill) static_for<pos * PPP, ((pos + 1) * PPP < DPW ? (pos + 1) * PPP : DPW), 1>([&](auto II) { issue_piece(st_fill, II); });
          }
        } else if constexpr (pos == LATE_POS - 1) {
          if (late_fill) static_for<0, DPW, 1>([&](auto II) { issue_piece(st_fill, II); });
        }
      });
    });
    if (late_fill) { advance_loader(); ++issued; }
  };

  int seq = seq0;
  if (seq >= seq_end) return;
  // De-phasing (round 4): a persistent launch whose tiles all cost the same runs its 256 workgroups in lock step, so every
  // CU reaches its epilogue at the same moment: the store burst of a round (160 KB per CU, 41 MB per round) then meets an
  // HBM that idled through the K loops, and the waves sit in store-issue back-pressure (cycle trace of `M262144 N2560 K320`:
  // 19-20 k cycles of epilogue per tile for 20 stores per wave, ~4 TB/s of writes in bursts, 1.9 TB/s on average).  With
  // p.dephase = P > 1 the workgroups of every XCD start in P groups, group g delayed by g / P of one tile's estimated
  // duration, so at any moment only ~1 / P of the CUs are in their store burst.  The delay costs (P - 1) / P of a tile at
  // the start of the launch; the dispatcher asks for it on launches of >= 8 rounds only.
  if (p.dephase > 1) {
    const int ph = (int)(blockIdx.x >> 3) % p.dephase;
    const int units = ph * p.dephase_units / p.dephase;       // units of 1024 cycles
    for (int i = 0; i < units; ++i) __builtin_amdgcn_s_sleep(16);
  }
  l_seq = seq;
  setup_loader(l_seq);
  // prologue: NSTG-1 K-tiles of the flattened (tile, k) stream in flight
#pragma unroll
  for (int j = 0; j < NSTG - 1; ++j)
    if (l_seq < seq_end) { issue_dma(j); ++issued; }
  int it = 0, st_it = 0;                                  // K-tile consumed next and its ring stage
  int young_stores = 0;                                   // lower bound of the stores issued since the last LDS-DMA piece
  TR_DECL
  const int epi = p.epi;
  const float gate = (epi & IDF_EPI_GATE) ? p.gate[0] : 0.0f;
  // wave-private 2-KB slot behind the ring: epilogue staging (big_epilogue) / the LNS statistics of a transposed tile
  char* const stg = reinterpret_cast<char*>(smem + NSTG * STAGE) + wave * 2048;

  // One output tile: accumulators cleared, the K loop, the epilogue.  SWAPT = the transposed-V tiles of the fused q | k | v
  // projection.  The accumulators are LOCAL to a tile kind: with one accumulator block shared by the plain and the swapped
  // K loop the register allocator kept both MFMA forms' tied operands alive and spilled ~400 VGPRs (80 TF instead of 700).
  auto run_tile = [&](auto SWAPT, auto SPLT) {
    constexpr bool SWAP = decltype(SWAPT)::value;
    constexpr bool SPL = decltype(SPLT)::value;               // this item is one K-slice of a tail tile
    int tile, slice, nki;
    item_map(seq, tile, slice, nki);
    f32x16 acc[TN][TM];
    float lsx[TM], lsq[TM];                               // LNS: per-lane partial sum / sum of squares of its A rows
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
      for (int b = 0; b < TM; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
#pragma unroll
    for (int b = 0; b < TM; ++b) { lsx[b] = 0.0f; lsq[b] = 0.0f; }
    TR(5) TR_COUNT(7)

    for (int kt = 0; kt < nki; ++kt) {
      // K-tile `it` must have landed: an LDS-DMA is ordered for other waves' ds_reads only by the ISSUING wave's vmcnt
      // wait followed by a barrier.  In steady state the NSTG-2 younger K-tiles stay in flight across the barrier
      // (counted wait; VM ops retire in order); at the tail of the stream fewer are outstanding -> wait for all.
      if (NSTG > 2 && issued - it == NSTG - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSTG - 2) * DPW) : "memory");
      else if (NSTG == 2 && kt == 0 && young_stores > 0) {
        // First K-tile behind an epilogue (round 4): its LDS-DMA pieces were enqueued during the previous tile's last
        // K-tile, i.e. BEFORE that tile's epilogue stores, and vector memory operations retire in order -- so "at most
        // `young_stores` operations outstanding" already means the pieces have landed, while vmcnt(0) would also wait for
        // the whole store burst to reach L2 (cycle trace at K = 320: 5-7 k cycles per tile).  young_stores is a LOWER bound
        // on the store instructions the previous epilogue issued (0 unless that tile had all 256 rows: a wave without a
        // valid row branches around its stores).
        if (young_stores >= 40) asm volatile("s_waitcnt vmcnt(40)" ::: "memory");
        else if (young_stores >= 20) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
        else if (young_stores >= 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else if (young_stores >= 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        else if (young_stores >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      TR(0)
      __builtin_amdgcn_s_barrier();                       // ... and every wave has finished reading the stage refilled below
      asm volatile("" ::: "memory");
      TR(1)
      // An LDS-DMA piece holds the issuing wave for 55-72 cycles (tools/ubench/dma_mix.hip: 14 B/clk per wave, 64 B/clk per
      // CU -- round 2's 5.6 / 34 came out of a ubench whose address update was a 64-bit modulo), i.e. ~550 cycles per K-tile
      // during which it issues no MFMA.  The fill is therefore SKEWED between the two waves of a SIMD: one enqueues the next
      // K-tile right behind the barrier, its partner from inside its MFMAs.  Round 3 (late): MFMA issue arbitration is
      // OLDEST WAVE FIRST (tools/ubench/mfma_sustain.hip: of two waves streaming MFMAs on one SIMD, wave 0 finishes ALL of
      // its stream before wave 4 gets a slot), so with the round-2 roles the younger, late-filling wave was starved until
      // the older one had finished its K-tile, and only then reached its own enqueue point -- ~550 cycles of its DMA issue with
      // the matrix pipe idle (cycle trace: profiles/r03_big_trace_baseline.log).  Now the OLDER waves 0-3 are the late ones
      // (at 7/8 of their MFMAs: their blocked time is the younger wave's turn) and the younger waves 4-7 enqueue first, under
      // the older wave's MFMAs: K-tile period 3770 -> 3190 cycles at K = 5120; in time -2...-9 % per launch
      // (profiles/r03_big_sched_ab.log), less than in cycles because the chip is power-limited here: the shader clock falls as
      // the pipe fills (clock x pipe-busy ~ constant ~ 1.0-1.2 GHz across all schedules, r03_big_trace_variants*.log).
      const bool fill = l_seq < seq_end;
      int st_fill = st_it + NSTG - 1;
      if (st_fill >= NSTG) st_fill -= NSTG;
      const bool late = fill && (SPREAD || (OLD_LATE ? wave < NW / 2 : wave >= NW / 2));
      if (fill && !late) {
        issue_dma(st_fill);
        ++issued;
      }
      TR(2)
      compute(SWAPT, acc, lsx, lsq, st_it, late, st_fill);
      TR(3) TR_COUNT(6)
      ++it;
      if (++st_it == NSTG) st_it = 0;
    }

    if constexpr (NW == 4) asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");   // inline-asm MFMAs: the compiler does not see their result latency
    // epilogue of tile seq: no workgroup barrier -- a wave that finishes its MFMAs early runs its epilogue while the other
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
    if constexpr (SWAP) {
      if constexpr (LNS) {
        // the statistics this wave summed in the K loop, parked in its own LDS slice (behind the ring) so that the
        // transposed epilogue can read the pair of ANY of the wave's 64 rows: [row][2], row = b*32 + l31
        float* stw = reinterpret_cast<float*>(stg);        // 2 * WM floats of the wave's 2-KB slot
        if (hi == 0) {
#pragma unroll
          for (int b = 0; b < TM; ++b) *reinterpret_cast<f32x2*>(stw + 2 * (b * 32 + l31)) = f32x2{lnm[b], lnr[b]};
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        big_epilogue_vt<DT, BM, BN, TN>(p, acc, seq, tiles_n, wm, wn, l31, hi, stw, (seq / tiles_n) * BM + wm * WM);
      } else {
        big_epilogue_vt<DT, BM, BN, TN>(p, acc, seq, tiles_n, wm, wn, l31, hi, p.ln_stats, 0);
      }
    } else {
      big_epilogue<DT, BM, BN, TN, SPL, LNS, STATS, GLU, GST>(p, acc, tile, slice, tiles_n, wm, wn, l31, hi, gate, stg, lnm, lnr);
    }
    TR(4)
    // store instructions this wave just issued, at least (see the first K-tile's wait above)
    young_stores = 0;
    if (p.epi_vmcnt && (tile / tiles_n) * BM + BM <= p.M) {
      if constexpr (SWAP) young_stores = 2 * TN * TM;
      else if constexpr (SPL) young_stores = 4 * TN * TM;
      else if constexpr (GLU) young_stores = TN * TM;
      else young_stores = (epi & IDF_EPI_GEGLU) ? TN * TM : ((epi & IDF_EPI_OUT_F32) ? 4 * TN * TM : 2 * TN * TM);
    }
  };

  for (; seq < seq_end; seq += seq_step) {
    if constexpr (VT) {
      // fused q | k | v projection: this tile's columns belong to V -> swapped operands, transposed store (workgroup-uniform)
      if ((seq % tiles_n) * BN >= p.vt_col0) run_tile(IC<1>{}, IC<0>{});
      else run_tile(IC<0>{}, IC<0>{});
    } else if constexpr (SPLIT) {
      if (seq < full) run_tile(IC<0>{}, IC<0>{});
      else run_tile(IC<0>{}, IC<1>{});
    } else {
      run_tile(IC<0>{}, IC<0>{});
    }
  }
  TR_DUMP
}

int g_num_cu = 0;

int num_cu() {
  if (g_num_cu == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    g_num_cu = n;
  }
  return g_num_cu;
}

// Schedule settings of the persistent kernel (process-global, read once from the environment; idf_set_tuning may override).
struct BigSched { int walk, dephase, dephase_min_rounds, dephase_epi_cycles, epi_vmcnt; };
BigSched& big_sched() {
  static BigSched sc = [] {
    auto env = [](const char* name, int dflt, int lo, int hi) {
      const char* e = getenv(name);
      if (!e) return dflt;
      const int v = atoi(e);
      return (v < lo || v > hi) ? dflt : v;
    };
    BigSched c;
    // measured defaults (profiles/r04_big_sched_walk_dephase_epivm.log, all bit-identical): the chunked walk pays only when
    // an m-tile has >= 32 n-tiles (N = 10240: +6.7 %; -4 ... -18 % on the narrower launches) -> walk = 2 (automatic);
    // the counted wait behind the epilogue is +0.3 ... +1.6 % on the K <= 640 launches and +-0 elsewhere -> on;
    // the de-phased start is +0 ... +4 % on two launches and -4 ... -20 % on the long-K ones -> off.
    c.walk = env("IDF_BIG_WALK", 2, 0, 2);
    c.dephase = env("IDF_BIG_DEPHASE", 0, 0, 16);
    c.dephase_min_rounds = env("IDF_BIG_DEPHASE_MIN_ROUNDS", 8, 1, 1 << 20);
    c.dephase_epi_cycles = env("IDF_BIG_DEPHASE_EPI", 6000, 0, 1 << 20);
    c.epi_vmcnt = env("IDF_BIG_EPIVM", 1, 0, 1);
    return c;
  }();
  return sc;
}

template <int DT, int BN, int NSTG, bool CONV, bool SPLIT = false, bool LNS = false, bool VT = false, bool STATS = false, bool GLU = false,
          bool GST = false>
int launch_big_cfg(const CoreParams& p, hipStream_t s, int splitk = 1) {
  constexpr int BM = 256, BKT = 64;
  if constexpr (!SPLIT && !LNS && !VT && !STATS && !GLU && !GST && NSTG == 2) {
    if (splitk > 1) return launch_big_cfg<DT, BN, NSTG, CONV, true>(p, s, splitk);
  }
  if constexpr (!SPLIT && !LNS && !CONV && !STATS && !GST && NSTG == 2) {
    if ((p.epi & IDF_EPI_LN_ROW) && !p.ln_stats) return launch_big_cfg<DT, BN, NSTG, CONV, false, true, VT, false, GLU>(p, s, 1);
  }
  void (*kern)(const CoreParams, const int) = gemm_kernel_big<DT, BM, BN, BKT, NSTG, CONV, SPLIT, LNS, VT, STATS, GLU, GST>;
  constexpr int smem = NSTG * (BM + BN) * BKT * 2 + NWAVES * 2048;      // ring + one 2-KB slot per wave (160 KB at BN = 320)
  static std::atomic<unsigned long long> attr_done{0};
  if (const int e = idf_lds_optin(reinterpret_cast<const void*>(kern), smem, attr_done)) return e;
  CoreParams q = p;
  q.splitk = splitk; q.kt_per_slice = p.K / BKT / splitk; q.kt_full = p.K / BKT;
  const int tiles_all = (p.N / BN) * ((p.M + BM - 1) / BM);
  if (!SPLIT) { q.full_items = tiles_all; q.tail_m0 = 0; q.tail_rows = p.M; }
  // (SPLIT: full_items / tail_m0 / tail_rows were set by the dispatcher; uniform split-K = 0 / 0 / M)
  const int tiles = q.full_items + (tiles_all - q.full_items) * splitk;   // work items
  const int slots = num_cu();
  const int grid = tiles < slots ? tiles : slots;
  {                                                       // schedule knobs (all bit-identical; see the kernel)
    const BigSched& sc = big_sched();
    const int rounds = (tiles + slots - 1) / slots;
    q.tile_walk = (!SPLIT && (sc.walk == 1 || (sc.walk == 2 && p.N / BN >= 32))) ? 1 : 0;
    q.epi_vmcnt = sc.epi_vmcnt;
    q.dephase = 0; q.dephase_units = 0;
    if (!SPLIT && sc.dephase > 1 && rounds >= sc.dephase_min_rounds) {
      q.dephase = sc.dephase;
      q.dephase_units = (q.kt_full * 3400 + sc.dephase_epi_cycles) / 1024;
    }
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NWAVES), smem, s, q, tiles);
  return idf_launch_status();
}

}  // namespace

std::atomic<long long> idf_stat_big_launches{0};
int idf_num_cu() { return num_cu(); }

// Occupancy bar of the automatic rule: the persistent kernel takes a launch when its tile grid fills at least this share of the
// workgroup slots of its last round (idf_launch_big).  set >= 0: new value, returns the previous one; set < 0: query.
#ifndef IDF_BIG_MIN_EFF_DEFAULT
#define IDF_BIG_MIN_EFF_DEFAULT 50
#endif
int idf_big_min_eff_pct(int set) {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("IDF_BIG_MIN_EFF");
    const int x = e ? atoi(e) : IDF_BIG_MIN_EFF_DEFAULT;
    v = (x < 1 || x > 100) ? IDF_BIG_MIN_EFF_DEFAULT : x;
  }
  const int prev = v;
  if (set >= 0) v = set;
  return prev;
}

// Shape gate + tile-width choice.  `force` skips the occupancy heuristic, not the shape rules.
int idf_launch_big(const CoreParams& p, int dtype, bool conv, bool force, hipStream_t s, int* splitk_out, int* parts_out,
                   int* tail_m0_out, int* gst_out) {
  if (gst_out) *gst_out = 0;
  if (splitk_out) *splitk_out = 1;
  if (parts_out) *parts_out = 0;
  if (tail_m0_out) *tail_m0_out = 0;
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
  // self-normalising LN_ROW (no statistics passed): computed in the K loop of the dense kernel
  const bool self_ln = (p.epi & IDF_EPI_LN_ROW) && !p.ln_stats;
  if (self_ln && conv) return IDF_BIG_UNSUPPORTED;
  // fused q | k | v projection: the transposed tiles are 320 wide and start on a tile boundary; their epilogue takes
  // LN_ROW / BIAS only and stores 16 tokens per lane
  const bool vt = p.vt_out != nullptr;
  if (vt && (conv || geglu || (p.N % 320) || (p.vt_col0 % 320) || p.vt_col0 <= 0 || p.vt_col0 >= p.N || (p.M % 16) || (p.ld_vt % 8) ||
             !aligned16(p.vt_out) || (p.epi & ~(IDF_EPI_BIAS | IDF_EPI_LN_ROW)))) return IDF_BIG_UNSUPPORTED;
  int bn = 0;
  if (geglu && (p.epi & IDF_EPI_GEGLU_P32)) bn = (p.N % 320 == 0) ? 320 : ((p.N % 256 == 0) ? 256 : 0);
  else if (geglu) bn = (p.N % 256 == 0) ? 256 : 0;
  else if (p.N % 320 == 0) bn = 320;
  else if (p.N % 256 == 0) bn = 256;
  else if (p.N % 128 == 0 && !self_ln) bn = 128;
  if (!bn) return IDF_BIG_UNSUPPORTED;
  const int bm = 256;
  const long long slots = (long long)num_cu();
  const long long tiles = (long long)(p.N / bn) * ((p.M + bm - 1) / bm);
  // split-K: when the tile grid leaves most CUs idle and K is long (the 8x8-level convs and ff-out GEMMs: 64 tiles of
  // 180..360 K-tiles), S slices per tile (S | K-tiles, >= 16 K-tiles each) leave fp32 partials in the caller's workspace
  int splitk = 1;
  if (splitk_out && bn != 128 && !geglu && !self_ln && !vt && p.ws && tiles * 2 <= slots) {
    const int nkt = p.K / BK;
    for (int cand = (int)(slots / tiles); cand >= 2; --cand) {
      if (nkt % cand || nkt / cand < 16) continue;
      if ((size_t)cand * p.M * p.N * sizeof(float) > p.ws_bytes) continue;
      splitk = cand;
      break;
    }
  }
  // full_items > 0: hybrid -- whole tiles for the leading full rounds, K-slices for the tail rows [tail_m0, M)
  long long full_items = 0;
  int tail_m0 = 0;
  if (!force) {
    // tile quantisation: a persistent workgroup slot processes ceil(items / slots) work items
    const long long items = tiles * splitk;
    const long long rounds = (items + slots - 1) / slots;
    const double eff = (double)items / (double)(rounds * slots);
    if (eff * 100.0 < (double)idf_big_min_eff_pct(-1)) {
      // Hybrid (round 3): at least one full round of whole tiles, and the tiles beyond the last full round (whole m-tiles
      // only, so that the tail is a row range) cut into S | K-tiles slices, one per workgroup of the last round.
      const int tiles_n = p.N / bn;
      const long long whole = ((tiles / slots) * slots / tiles_n) * tiles_n;
      const long long rem = tiles - whole;
      const int nkt = p.K / BK;
      int S = 0;
      if (splitk == 1 && splitk_out && tail_m0_out && whole > 0 && rem > 0 && bn != 128 && !geglu && !self_ln && !vt && p.ws)
        for (int cand = (int)(slots / rem); cand >= 2; --cand) {
          // slices of >= 4 K-tiles: with shorter ones the slice prologues and the fp32 partial traffic cost more than the idle
          // CUs they fill (K = 320, five one-K-tile slices: 200 -> 169 TF at 18 rows; K = 1280: 409 -> 500, 3x3 conv
          // 536 -> 709 TF; profiles/r03_shape_profile_B18_hybrid.log)
          if (nkt % cand || nkt / cand < 4) continue;
          S = cand;
          break;
        }
      tail_m0 = (int)(whole / tiles_n) * bm;
      if (S < 2 || (size_t)S * (size_t)(p.M - tail_m0) * p.N * sizeof(float) > p.ws_bytes) return IDF_BIG_UNSUPPORTED;
      // worth it only if the tail round is short: its slices run nkt / S K-tiles plus a reducer pass over the tail rows
      full_items = whole;
      splitk = S;
    }
  }
  if (splitk_out) *splitk_out = splitk;
  if (tail_m0_out) *tail_m0_out = tail_m0;
  {                                                       // the loader uses 32-bit element offsets
    const unsigned long long rows = conv ? (unsigned long long)(p.M / (p.Ho * p.Wo)) * p.Hin * p.Win : (unsigned long long)p.M;
    if (rows * (unsigned long long)p.lda >= (1ull << 31) || (unsigned long long)p.N * p.ldw >= (1ull << 31)) return IDF_BIG_UNSUPPORTED;
  }
  ++idf_stat_big_launches;
  // output-row statistics from the epilogue registers: unsplit dense 16-bit-output GEMMs whose epilogue is the plain one
  // ... and the workspace holds one (mean, M2) slot per row and wave column half: [M][2 * N / bn][2] floats
  const bool stats = parts_out && p.stat_parts && !conv && !geglu && !vt && !self_ln && splitk == 1 && bn != 128 &&
                     !(p.epi & IDF_EPI_OUT_F32) && (((uintptr_t)p.stat_parts) & 7u) == 0 &&
                     (size_t)p.M * (size_t)(2 * (p.N / bn)) * 2 * sizeof(float) <= p.ws_bytes;
  CoreParams ps = p;
  if (stats) { ps.parts = 2 * (p.N / bn); *parts_out = ps.parts; }
  // GroupNorm partials of the output from the epilogue (GST): unsplit 3x3 convs on 320-wide tiles whose rows are whole 64-row
  // chunks of one sample and whose wave column halves (160) hold whole groups
  const bool gst = gst_out && p.gn_partial && conv && bn == 320 && splitk == 1 && full_items == 0 && !(p.epi & IDF_EPI_OUT_F32) &&
                   p.gn_hw > 0 && (p.gn_hw % 64) == 0 && (p.M % p.gn_hw) == 0 && (p.N % 32) == 0 && (160 % (p.N / 32)) == 0 &&
                   (((uintptr_t)p.gn_partial) & 7u) == 0;
  if (gst) *gst_out = 1;
  CoreParams pq = p;                                      // split launches: uniform split-K (full 0) or hybrid
  pq.full_items = (int)full_items; pq.tail_m0 = tail_m0; pq.tail_rows = p.M - tail_m0;
  const bool glu320 = geglu && (p.epi & IDF_EPI_GEGLU_P32) && bn == 320;
#define IDF_BIG_DISPATCH(DT)                                                                                              \
  if (glu320) return launch_big_cfg<DT, 320, 2, false, false, false, false, false, true>(p, s);                           \
  if (stats) return bn == 320 ? launch_big_cfg<DT, 320, 2, false, false, false, false, true>(ps, s)                        \
                              : launch_big_cfg<DT, 256, 2, false, false, false, false, true>(ps, s);                       \
  if (vt) return launch_big_cfg<DT, 320, 2, false, false, false, true>(p, s);                                             \
  if (bn == 128) {                       /* 128-wide tiles (the VAE's 128-channel convs at 512^2): 3 stages of 48 KB */  \
    if (conv) return launch_big_cfg<DT, 128, 3, true>(p, s);                                                              \
    return launch_big_cfg<DT, 128, 3, false>(p, s);                                                                       \
  }                                                                                                                       \
  if (gst) return launch_big_cfg<DT, 320, 2, true, false, false, false, false, false, true>(p, s);                         \
  if (conv) return bn == 320 ? launch_big_cfg<DT, 320, 2, true>(pq, s, splitk) : launch_big_cfg<DT, 256, 2, true>(pq, s, splitk);   \
  return bn == 320 ? launch_big_cfg<DT, 320, 2, false>(pq, s, splitk) : launch_big_cfg<DT, 256, 2, false>(pq, s, splitk);
  if (dtype == IDF_BF16) { IDF_BIG_DISPATCH(IDF_BF16) }
  if (dtype == IDF_F16) { IDF_BIG_DISPATCH(IDF_F16) }
#undef IDF_BIG_DISPATCH
  return IDF_E_UNSUPPORTED;
}
