// mlp_fused.hip -- the GEGLU feed-forward of a C = 320 transformer block as ONE kernel (gfx950):
//     out = x + [gate *] ( W2 . ( value * gelu(gate) ) + b2 ),   (value | gate) = LN(x) . W1^T + b1
// (attention.py:36-63 GEGLU / FeedForward, call sites :309 `x + tanh(alpha_dense) * ff(norm2(x))` and :337 `ff(norm3(x)) + x`).
//
// Why.  As two GEMMs the 4C-wide activated intermediate H is written to HBM and read back -- 2 x 1.34 GB per MLP at 128 rows
// of 64 x 64 latents -- the first GEMM (K = 320: five K-tiles per output tile) spends 40 % of its time in an epilogue that
// cannot overlap its own K loop (GELU on the VALU, then a store burst at the CU's ~12 B/clk store rate: rounds 3-4 traces,
// profiles/r04_big_stage_epilogue.log), and the second one is HBM-bound on reading H.  Here H never leaves the CU:
//
//   * one persistent 512-thread workgroup (8 wave64, 2 per SIMD) per CU walks 128-row tiles; wave (wm, wn) owns rows
//     32 wm .. + 31 and the output columns 160 wn .. + 159;
//   * the tile's activation rows live in REGISTERS for the whole tile: 20 MFMA B-operand fragments (80 VGPRs) per lane, read
//     once from global memory -- X never touches LDS;
//   * the weights stream through a 2-slot LDS ring by LDS-DMA in 40 chunks of 32 intermediate columns: W1 rows
//     [64 j, 64 j + 64) (period-32 packing: [16 value | 16 gate] per 32 rows), the matching 32 columns of W2, and the 64 + 64
//     LayerNorm-fold constants (c, d) of those rows -- 61 KB per chunk, the same 31 B/clk per CU at full MFMA rate as the
//     256 x 320 GEMM tile;
//   * per chunk, wave (wm, wn) computes the 32 x 32 pre-activation fragment of W1 rows 64 j + 32 wn .. (20 MFMAs, K = 320, B
//     operand from registers), applies LayerNorm fold + bias + GEGLU in registers and packs the 16 activated columns of its
//     32 rows to the 16-bit type -- which IS an MFMA B-operand fragment of the second GEMM (lane = row, 8 consecutive
//     registers = 8 k values; the k order inside a 16-group is a fixed permutation that W2's packed image carries too).  The
//     two waves of a row group exchange their fragments through 1 KB of LDS each, and both run the second GEMM's two k-steps
//     for their own 160 output columns (10 MFMAs, accumulators 80 VGPRs, live across the 40 chunks);
//   * epilogue once per tile: + b2, gate, + residual, 16-bit store (80 KB per 314 MFLOP instead of 720 KB of H and output).
//
// Numerics: as the two-GEMM path -- fp32 accumulation, H rounded to the 16-bit type before the second product -- up to the
// summation order of the second GEMM.  LDS: 2 x 61 KB ring + 8 KB exchange + 16 KB epilogue staging = 146 KB.
// Roofline: MFMA-bound, 2 * M * (2560 * 320 + 320 * 1280) flops; HBM: 2 B in + 2 B out per element of x (+ residual read).
#include "gemm_core.h"
#include "mw_prims.h"
#include <cstdlib>
#include <atomic>
#include <type_traits>
#include <utility>

using namespace idfcore;
using namespace idfmw;

namespace {

constexpr int MLP_C = 320, MLP_H = 1280, MLP_CH = 32, MLP_NCH = MLP_H / MLP_CH;     // 40 chunks of 32 intermediate columns
constexpr int MLP_BM = 128;
constexpr int W1_BYTES = 5 * 64 * 128;          // 5 K-tiles x [64 rows][64 k]: 128-B rows, 16-B slot ^= (row >> 1) & 7
constexpr int W2_BYTES = 320 * 64;              // [320 rows][32 k]: 64-B rows, 16-B slot ^= (row >> 2) & 3
constexpr int CD_BYTES = 1024;                  // c[64] | d[64] fp32 (+ 512 B the DMA instruction fills with a copy)
constexpr int SLOT_BYTES = W1_BYTES + W2_BYTES + CD_BYTES;
constexpr int XCH_OFF = 2 * SLOT_BYTES, STG_OFF = XCH_OFF + 8 * 1024, MLP_SMEM = STG_OFF + 8 * 2048;

struct MlpParams {
  const unsigned short* x; int ldx;             // [M][320] 16-bit: the LayerNorm input AND the residual
  const float* ln_stats;                        // [M][2] (mu, rstd)
  const unsigned short* w1; int ldw1;           // [2560][320] gamma-folded, rows packed [16 value | 16 gate] per 32
  const float* cd;                              // [40][128]: per chunk c[64] | d[64] of its 64 packed rows
  const unsigned short* w2p; int ldw2;          // [320][1280], k permuted inside every 16-group (mlp_w2_perm)
  const float* b2; const float* gate;           // [320]; device scalar or nullptr
  unsigned short* out; int ldo;
  int M;
};

// LDS-DMA as inline assembly (see gemm_big.hip): lds = LDS byte address of lane 0's 16-B slot, through M0
__device__ __forceinline__ void mlp_dma16(const void* sbase /* wave-uniform */, unsigned voff_bytes, unsigned lds) {
  lds = __builtin_amdgcn_readfirstlane(lds);
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds), "v"(voff_bytes), "s"(sbase) : "memory");
}

// Schedule details, each measured on its own in profiles/r04_mlp_variants.log (all variants bit-identical):
//   * skewed fill (as in gemm_big.hip): an LDS-DMA piece holds the issuing wave ~60 cycles, 7-8 pieces per wave and chunk; the
//     older waves 0-3 enqueue right behind the chunk's barrier, the younger waves 4-7 one piece per two MFMAs of their first
//     product, so the two waves of a SIMD are not blocked at the same time (-2.6 %);
//   * fragment reads ahead of their MFMAs: W1 fragments 4 steps, the fold constants under the last MFMAs of the first product,
//     all ten W2 fragments before the exchange barrier (+-0: the compiler's own schedule already covered the LDS latency);
//   * the k-step of the second product on the wave's own fragment is issued before the exchange barrier, and the chunk's two
//     waits are counted (W1 + constants before the first product, W2 at the exchange barrier): +-0 (r04_mlp_split_waits.log);
//   * NOT kept: running the two waves of a SIMD out of phase, so that one wave's VALU step (fold + GELU, ~850 cycles per
//     chunk with the pipe idle: cycle trace r04_mlp_trace_*.log) meets the other's MFMAs -- as two barrier intervals per chunk
//     (group B = group A delayed by one interval; ring refilled in two parts): 7 % SLOWER; as three balanced intervals
//     (first product | GELU + own k-step | partner's k-step): 14 % SLOWER (r04_mlp_phase_shift_*.log,
//     r04_mlp_three_interval_rejected.log; both bit-exact).  Two accumulator chains in the first product: spills, -12 %.
//     Round 5: the overlap INSIDE each wave, across chunks (first product of chunk g + 1 under the activation of chunk g, two
//     pre-activation accumulators, no spills, bit-identical): 5-6 % SLOWER (profiles/archive_rejected_kernels/mlp_pipelined_r05.hip,
//     profiles/r05_mlp_pipelined_rejected.log) -- with two waves per SIMD the activation's VALU issue competes with both waves' MFMAs.
//     Lock step is what this pipe likes: both waves of a SIMD in the same MFMA phase interleave perfectly, and a barrier
//     costs least when everybody arrives together.
// Optional per-segment cycle trace (a second library build with -DIDF_MLP_TRACE, read through idf_mlp_trace_read by
// tools/ubench/mlp_harness.hip; the shipped library has none of it): s_memtime deltas of waves 0 and 4 of the first and of a
// middle workgroup, summed over all chunks.  Segments: 0 vmcnt wait, 1 chunk barrier, 2 early LDS-DMA enqueue, 3 first
// product (+ late enqueue), 4 fold + GEGLU + publish + W2 fragment reads, 5 exchange barrier, 6 second product, 7 tile
// epilogue, 8 tile load, 9 number of chunks.
#ifdef IDF_MLP_TRACE
__device__ unsigned long long idf_mlp_trace_buf[4][10];
#define MTR_DECL unsigned long long tr_last = __builtin_readcyclecounter(), tr_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define MTR(i) { const unsigned long long tr_now = __builtin_readcyclecounter(); tr_acc[i] += tr_now - tr_last; tr_last = tr_now; }
#define MTR_COUNT(i) { tr_acc[i] += 1; }
#define MTR_DUMP { const int trb = blockIdx.x == 0 ? 0 : ((int)blockIdx.x == (int)gridDim.x / 2 ? 1 : -1);                     \
    if (trb >= 0 && lane == 0 && (wave == 0 || wave == 4)) { for (int i = 0; i < 10; ++i) idf_mlp_trace_buf[trb * 2 + (wave ? 1 : 0)][i] = tr_acc[i]; } }
#else
#define MTR_DECL
#define MTR(i) {}
#define MTR_COUNT(i) {}
#define MTR_DUMP {}
#endif

template <int DT>
__global__ __launch_bounds__(512, 2) void mlp320_kernel(const MlpParams p, const int tiles) {
  constexpr bool PF = true, SK = true;
  extern __shared__ __attribute__((aligned(128))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int wn = wave & 1, wm = wave >> 1;
  const int G = gridDim.x;

  // ---- LDS-DMA roles (per-lane source offsets are the same for every chunk; the chunk moves the wave-uniform base)
  // W1: piece (t, wave) = rows 8 wave .. + 7 of K-tile t: lane -> row 8 wave + lane / 8, slot lane % 8
  unsigned w1_voff, w2_voff[3];
  {
    const int row = 8 * wave + (lane >> 3);
    const int src = (lane & 7) ^ ((row >> 1) & 7);
    w1_voff = (unsigned)(row * p.ldw1 + src * 8) * 2u;
  }
  // W2: piece i = rows 16 i .. + 15: lane -> row 16 i + lane / 4, slot lane % 4; pieces wave, wave + 8, (wave + 16 < 20)
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    const int row = 16 * (wave + 8 * t) + (lane >> 2);
    const int src = (lane & 3) ^ ((row >> 2) & 3);
    w2_voff[t] = (unsigned)(row * p.ldw2 + src * 8) * 2u;
  }
  // piece `idx` of this wave for chunk j, in the order they are enqueued (= the order they land): 0 = the fold constants
  // (wave 7 only), 1..5 = W1 K-tiles, 6..8 = W2 row groups (8 only on waves 0-3).  The W2 pieces are the YOUNGEST: the chunk's
  // first wait (W1 + constants landed) lets them stay in flight, they are waited for at the exchange barrier.
  auto issue_piece = [&](int idx, int j, int slot) {
    char* const base = smem + slot * SLOT_BYTES;
    if (idx == 0) {
      if (wave == 7) mlp_dma16(p.cd + (size_t)j * 128, (unsigned)((lane & 31) * 16), lds_u32(base + W1_BYTES + W2_BYTES));
    } else if (idx < 6) {
      mlp_dma16(p.w1 + (size_t)j * 64 * p.ldw1 + (idx - 1) * 64, w1_voff, lds_u32(base + (idx - 1) * 8192 + wave * 1024));
    } else {
      const int t = idx - 6;
      if (wave + 8 * t < 20) mlp_dma16(p.w2p + j * 32, w2_voff[t], lds_u32(base + W1_BYTES + (wave + 8 * t) * 1024));
    }
  };
  auto issue_chunk = [&](int j, int slot) {
#pragma unroll
    for (int idx = 0; idx < 9; ++idx) issue_piece(idx, j, slot);
  };
  // counted waits (vector memory operations retire in order): "W1 + constants of this chunk landed" leaves this wave's W2
  // pieces in flight; "W2 of this chunk landed" leaves the pieces of the NEXT chunk, all enqueued by then, in flight
  auto wait_w1 = [&]() {
    if (wave < 4) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  };
  auto wait_w2 = [&](bool next_in_flight) {
    if (!next_in_flight) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (wave < 4 || wave == 7) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
  };

  // ---- fragment addressing
  const int sw1 = (l31 >> 1) & 7, sw2 = (l31 >> 2) & 3;
  const int w1_row = (32 * wn + l31) * 128;                    // byte offset of the lane's W1 row inside a K-tile image
  const int w2_row = W1_BYTES + (160 * wn + l31) * 64;         // ... of its W2 row for output fragment 0 (+ 2048 per fragment)
  char* const xch_mine = smem + XCH_OFF + wave * 1024 + lane * 16;
  char* const xch_peer = smem + XCH_OFF + (wave ^ 1) * 1024 + lane * 16;
  char* const stg = smem + STG_OFF + wave * 2048;

  int tile = ((G & 7) == 0) ? (int)(blockIdx.x & 7) * (G >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  if (tile >= tiles) return;
  const float gate = p.gate ? p.gate[0] : 1.0f;

  issue_chunk(0, 0);
  int g = 0;                                                   // chunks consumed so far (ring slot = g & 1)
  MTR_DECL
  for (; tile < tiles; tile += G) {
    const int m = tile * MLP_BM + wm * 32 + l31;               // the lane's row
    // activation rows: 20 B-operand fragments, elements 16 ks + 8 hi .. + 7 of row m
    u32x4 xf[20];
    {
      const unsigned short* xr = p.x + (size_t)m * p.ldx + 8 * hi;
#pragma unroll
      for (int ks = 0; ks < 20; ++ks) xf[ks] = *reinterpret_cast<const u32x4*>(xr + 16 * ks);
    }
    const f32x2 st = *reinterpret_cast<const f32x2*>(p.ln_stats + 2 * (size_t)m);
    const float nmu = -st[0], rstd = st[1];
    f32x16 acc2[5];
#pragma unroll
    for (int a = 0; a < 5; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[a][r] = 0.0f;

    MTR(8)
    for (int j = 0; j < MLP_NCH; ++j, ++g) {
      // chunk g has landed (this wave's pieces; the barrier publishes everyone's) and every wave is done with the other slot
      wait_w1();
      MTR(0)
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      MTR(1) MTR_COUNT(9)
      const bool last = (j == MLP_NCH - 1) && (tile + G >= tiles);
      const int jn = j + 1 == MLP_NCH ? 0 : j + 1;
      const bool late = SK && wave >= 4 && !last;                // this wave enqueues from inside its first product
      if (!last && !late) issue_chunk(jn, (g + 1) & 1);
      MTR(2)
      const char* const sl = smem + (g & 1) * SLOT_BYTES;

      // ---- GEMM 1: the 32 x 32 pre-activation fragment of packed W1 rows 64 j + 32 wn ..: 20 k-steps, B operand = xf
      f32x16 acc1;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc1[r] = 0.0f;
      const float* cdp = reinterpret_cast<const float*>(sl + W1_BYTES + W2_BYTES) + 32 * wn + 4 * hi;
      f32x4 cv[2], cg[2], dv[2], dg[2];
      {
        const char* wb = sl + w1_row;
        constexpr int DEPTH = PF ? 4 : 1;
        u32x4 wf[DEPTH + 1];
        auto rd = [&](int ks) {
          const int kt = ks >> 2, c = 2 * (ks & 3) + hi;
          return *reinterpret_cast<const u32x4*>(wb + kt * 8192 + ((c ^ sw1) << 4));
        };
#pragma unroll
        for (int ks = 0; ks < DEPTH; ++ks) wf[ks] = rd(ks);
#pragma unroll
        for (int ks = 0; ks < 20; ++ks) {
          if (ks + DEPTH < 20) wf[(ks + DEPTH) % (DEPTH + 1)] = rd(ks + DEPTH);
          if (PF && ks == 14) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
              cv[q] = *reinterpret_cast<const f32x4*>(cdp + 8 * q); cg[q] = *reinterpret_cast<const f32x4*>(cdp + 16 + 8 * q);
              dv[q] = *reinterpret_cast<const f32x4*>(cdp + 64 + 8 * q); dg[q] = *reinterpret_cast<const f32x4*>(cdp + 80 + 8 * q);
            }
          }
          acc1 = Elem<DT>::mfma32(wf[ks % (DEPTH + 1)], xf[ks], acc1);
          if (SK && (ks & 1) && (ks >> 1) < 9) {
            if (late) issue_piece(ks >> 1, jn, (g + 1) & 1);
          }
        }
      }
      MTR(3)
      if (!PF) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          cv[q] = *reinterpret_cast<const f32x4*>(cdp + 8 * q); cg[q] = *reinterpret_cast<const f32x4*>(cdp + 16 + 8 * q);
          dv[q] = *reinterpret_cast<const f32x4*>(cdp + 64 + 8 * q); dg[q] = *reinterpret_cast<const f32x4*>(cdp + 80 + 8 * q);
        }
      }
      // ---- LayerNorm fold + bias + GEGLU in registers: acc1[4 q + e] = pre[packed row 8 q + 4 hi + e][row l31]; rows 0..15 of
      // the fragment are the values, 16..31 the gates of intermediate columns 0..15: the lane ends up with columns
      // {4 hi + e, 8 + 4 hi + e}, which (in this order) are the 8 k values of its half of the second GEMM's 16-wide k-step
      u32x4 hmine;
      {
        float o[8];
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float val = fmaf(rstd, fmaf(nmu, cv[q][e], acc1[4 * q + e]), dv[q][e]);
            const float gat = fmaf(rstd, fmaf(nmu, cg[q][e], acc1[4 * (q + 2) + e]), dg[q][e]);
            o[4 * q + e] = val * gelu_erf_f(gat);
          }
        hmine = pack8<DT>(o);
      }
      // ---- exchange with the other wave of this row group (it holds the other 16 intermediate columns of the chunk)
      *reinterpret_cast<u32x4*>(xch_mine) = hmine;
      const char* w2b = sl + w2_row;
      // ---- GEMM 2: two k-steps x 5 output fragments.  The k-step of the wave's OWN fragment (k-step wn) needs nothing from the
      // other wave: its 5 MFMAs are issued BEFORE the exchange barrier and run while the wave waits there (cycle trace,
      // profiles/r04_mlp_trace_*.log: ~800 cycles per chunk at that barrier on the waves that enqueue early); the peer's
      // k-step follows behind it.  (The two waves of a row group therefore add their two k-steps in opposite orders.)
      u32x4 w2f[2][5];
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int a = 0; a < 5; ++a) w2f[kk][a] = *reinterpret_cast<const u32x4*>(w2b + a * 2048 + (((2 * (kk ^ wn) + hi) ^ sw2) << 4));
#pragma unroll
      for (int a = 0; a < 5; ++a) acc2[a] = Elem<DT>::mfma32(w2f[0][a], hmine, acc2[a]);       // w2f[0] = k-step wn
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      wait_w2(!last);
      MTR(4)
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      MTR(5)
      const u32x4 hpeer = *reinterpret_cast<const u32x4*>(xch_peer);
#pragma unroll
      for (int a = 0; a < 5; ++a) acc2[a] = Elem<DT>::mfma32(w2f[1][a], hpeer, acc2[a]);       // w2f[1] = k-step wn ^ 1
      MTR(6)
    }

    // ---- tile epilogue: + b2, gate, + residual; 32 x 32 fragments through the wave's 2-KB slot so that a store instruction
    // covers 16 rows x 64 contiguous bytes (gemm_big.hip "coalesced epilogue accesses"; same slot image)
    const int sl_row = lane >> 2, sl_pc = lane & 3;
    auto stg_f = [](int row) { return ((((row >> 2) ^ (row >> 3)) & 1) << 1) | (((row >> 1) ^ (row >> 3) ^ (row >> 4)) & 1); };
    auto stg_at = [&](int row, int pc) { return reinterpret_cast<u32x4*>(stg + row * 64 + ((pc ^ stg_f(row)) << 4)); };
    const int m_base = tile * MLP_BM + wm * 32;
#pragma unroll
    for (int a = 0; a < 5; ++a) {
      const int n = 160 * wn + 32 * a;
      // residual rows, coalesced
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = sl_row + 16 * i;
        *stg_at(row, sl_pc) = *reinterpret_cast<const u32x4*>(p.x + (size_t)(m_base + row) * p.ldx + n + sl_pc * 8);
      }
      float v[16];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const auto s02 = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc2[a][e]), __float_as_uint(acc2[a][8 + e]), false, false);
        const auto s13 = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc2[a][4 + e]), __float_as_uint(acc2[a][12 + e]), false, false);
        v[e] = __uint_as_float(s02[0]); v[4 + e] = __uint_as_float(s02[1]);
        v[8 + e] = __uint_as_float(s13[0]); v[12 + e] = __uint_as_float(s13[1]);
      }
      const float* bp = p.b2 + n + 16 * hi;
#pragma unroll
      for (int jq = 0; jq < 4; ++jq) {
        const f32x4 bq = *reinterpret_cast<const f32x4*>(bp + 4 * jq);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[4 * jq + e] += bq[e];
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      float r[16];
      unpack8<DT>(*stg_at(l31, 2 * hi), r);
      unpack8<DT>(*stg_at(l31, 2 * hi + 1), r + 8);
#pragma unroll
      for (int jq = 0; jq < 16; ++jq) v[jq] = fmaf(gate, v[jq], r[jq]);
      asm volatile("" ::: "memory");
      *stg_at(l31, 2 * hi) = pack8<DT>(v);
      *stg_at(l31, 2 * hi + 1) = pack8<DT>(v + 8);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = sl_row + 16 * i;
        *reinterpret_cast<u32x4*>(p.out + (size_t)(m_base + row) * p.ldo + n + sl_pc * 8) = *stg_at(row, sl_pc);
      }
    }
    MTR(7)
  }
  MTR_DUMP
}

template <int DT>
int launch_mlp320(const MlpParams& p, hipStream_t s) {
  void (*kern)(const MlpParams, const int) = mlp320_kernel<DT>;
  static std::atomic<unsigned long long> attr_done{0};
  if (const int e = idf_lds_optin(reinterpret_cast<const void*>(kern), MLP_SMEM, attr_done)) return e;
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
  const int tiles = p.M / MLP_BM;
  const int grid = tiles < cus ? tiles : cus;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), MLP_SMEM, s, p, tiles);
  return idf_launch_status();
}


// ====================================================================================================================
// mlp320w_kernel -- the same feed-forward as ONE INSTRUCTION STREAM PER SIMD (round 6).
//
// mlp320_kernel's cycle trace (profiles/r04_mlp_trace_*.log): 4300-4450 cycles per chunk for 1920 cycles of matrix pipe -- both
// waves of a SIMD run fold + GELU together with the pipe idle (~900), and ~1500 go to the chunk barrier, the exchange barrier
// between the two waves of a row group, the LDS-DMA issue and the wait for pieces that were enqueued late.  Overlapping the
// VALU work ACROSS the two waves (round 4) or inside each wave with compiler scheduling (round 5) lost 5-14 %: the 256-register
// budget of two waves per SIMD forced the fold constants into just-in-time scalar LDS reads, which put LDS latency and a
// 14-instruction dependent chain between two MFMAs of an in-order wave.  Here:
//   * a workgroup is FOUR waves, one per SIMD, 512 registers each; wave w owns rows 32 w .. + 31 of the 128-row tile and ALL
//     320 output columns -- no exchange of activated fragments, no second barrier, ONE barrier per chunk;
//   * the x rows (80 registers, MFMA B operands) and the 10 output accumulators (160) live in AGPRs NAMED in the asm text
//     (attention4w.hip's technique: the register allocator knows them only as clobbers);
//   * software pipeline over chunks, one iteration j = { second product of chunk j - 1 (20 MFMAs) | first product of chunk j + 1
//     (40 MFMAs, both 32 x 32 fragments, two independent chains) | fold + GEGLU of chunk j (144 VALU instructions, packed-fp32
//     where the operation exists) }: the VALU statements are STAGE-ORDERED over the four register pairs of a fragment (a
//     dependent instruction is >= 4 issue slots behind its producer) and spread ~3 per MFMA gap; LDS fragment / constant reads sit
//     3 gaps ahead of their consumers with counted lgkmcnt waits; the iteration's 15-16 LDS-DMA pieces ride in the first gaps
//     behind the barrier (W1 two chunks ahead, W2 / constants one -- the ring is the 8-wave kernel's: same LDS image, same
//     packed operands);
//   * the stream is generated (tools/gen_mlpw_stream.py -> mlpw_stream.inc): every statement is `asm volatile`, source order =
//     issue order.
// Same operations on the same operands in the same order per output element as mlp320_kernel: BIT-IDENTICAL results (the
// harness' whole-output checksums).  The residual comes out of the x fragments in the AGPRs (two v_permlane32_swap per dword
// turn the B-operand layout into the output layout): x is read from HBM once.
// ====================================================================================================================
constexpr int MW_NAGPR = 240;
constexpr int MW_B2_OFF = 2 * SLOT_BYTES, MW_STG_OFF = MW_B2_OFF + MLP_C * 4, MW_SMEM = MW_STG_OFF + 4 * 2 * 2048;

struct MwCtx {
  unsigned w1a[4], w2a[2], cda;                 // LDS byte addresses of this iteration's fragment / constant reads (per lane)
  f32x2 nmu2, rstd2, k1, k2, k3, one2;
  float lo8, hi8;
  const char* w1src; const char* w2src; const char* cdsrc;      // wave-uniform sources of this iteration's LDS-DMA pieces
  unsigned w1_ustride, w2_tstride;              // bytes: 32 rows of W1, 64 rows of W2
  unsigned w1_voff, w2_voff, cd_voff;           // per-lane source byte offsets
  const char* w1b[2]; const char* w2b[5]; const char* cdb;       // (MW_DMA_IMM) static sources: W1 rows 32 u .., W2 rows 64 t ..
  unsigned w1_vj, w2_vj, cd_vj;                 // (MW_DMA_IMM) per-lane offsets of this iteration: + chunk offset
  unsigned w1dst, w2dst, cddst;                 // LDS byte addresses of this wave's first piece
  int wave;
#ifdef IDF_MLPW_TRACE
  mutable unsigned long long tr[12];            // cycle sums: [0..5] segments of mw_body_11, [6] its calls, [7] tile load, [8] epilogue, [9] pro + 01 + 10 + drain
#endif
};

// Optional cycle trace (tools/build_mlpw_variant.sh <name> MW_TRACE=1 -- -DIDF_MLPW_TRACE; read through idf_mlpw_trace_read by
// tools/ubench/mlp_harness.hip; the shipped library has none of it).  The marks are s_memtime into separate SGPR pairs, read
// only at the end of the body: a wait for one of them inside the stream would also wait for the stream's LDS reads.
#ifdef IDF_MLPW_TRACE
__device__ unsigned long long idf_mlpw_trace_buf[4][12];
#define MW_TR_DECL unsigned long long trm[7];
#define MW_TR_MARK(i) asm volatile("s_memtime %0" : "=s"(trm[i]));
#define MW_TR_END { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(trm[0]), "+s"(trm[1]), "+s"(trm[2]), "+s"(trm[3]), "+s"(trm[4]), "+s"(trm[5]), "+s"(trm[6])); \
    for (int i = 0; i < 6; ++i) c.tr[i] += trm[i + 1] - trm[i]; c.tr[6] += 1; }
#else
#define MW_TR_DECL
#define MW_TR_MARK(i)
#define MW_TR_END
#endif

#define MW_TOP asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
#ifndef MLPW_STREAM_INC
#define MLPW_STREAM_INC "mlpw_stream.inc"
#endif
#include MLPW_STREAM_INC

template <int DT>
__global__ __launch_bounds__(256, 1) void mlp320w_kernel(const MlpParams p, const int tiles) {
  asm volatile("" ::: "a0", "a239");               // the asm-owned AGPR block: this is where the kernel descriptor learns its size
  extern __shared__ __attribute__((aligned(128))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int G = gridDim.x;
  const unsigned smem_lds = lds_u32(smem);

  MwCtx c;
  c.wave = wave;
  c.k1 = f32x2{1.0142652e-3f, 1.0142652e-3f}; c.k2 = f32x2{-1.0677574e-1f, -1.0677574e-1f}; c.k3 = f32x2{-2.3011213f, -2.3011213f};
  c.one2 = f32x2{1.0f, 1.0f};
  c.lo8 = -8.0f; c.hi8 = 8.0f;
  asm volatile("" : "+v"(c.k1), "+v"(c.k2), "+v"(c.k3), "+v"(c.one2), "+v"(c.hi8));
  // LDS-DMA roles.  W1: piece (kt, u) = rows 8 (wave + 4 u) .. + 7 of K-tile kt (lane -> row + lane / 8, 16-B slot lane % 8; the
  // swizzle (row >> 1) & 7 does not depend on u); W2: piece t = rows 16 (wave + 4 t) .. + 15 (lane -> row + lane / 4, slot lane % 4)
  {
    const int row = 8 * wave + (lane >> 3);
    c.w1_voff = (unsigned)(row * p.ldw1 + (((lane & 7) ^ ((row >> 1) & 7)) << 3)) * 2u;
    const int row2 = 16 * wave + (lane >> 2);
    c.w2_voff = (unsigned)(row2 * p.ldw2 + (((lane & 3) ^ ((row2 >> 2) & 3)) << 3)) * 2u;
    c.cd_voff = (unsigned)((lane & 31) * 16);
    c.w1_ustride = (unsigned)(32 * p.ldw1 * 2);
    c.w2_tstride = (unsigned)(64 * p.ldw2 * 2);
  }
  // fragment addressing (the 8-wave kernel's LDS image): W1 row 32 f + l31 of a K-tile, 16-B slot (2 (ks & 3) + hi) ^ sw1;
  // W2 row 32 a + l31, slot (2 kk + hi) ^ sw2
  const int sw1 = (l31 >> 1) & 7, sw2 = (l31 >> 2) & 3;
  unsigned w1o[4], w2o[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) w1o[i] = smem_lds + (unsigned)(l31 * 128 + (((2 * i + hi) ^ sw1) << 4));
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) w2o[kk] = smem_lds + (unsigned)(W1_BYTES + l31 * 64 + (((2 * kk + hi) ^ sw2) << 4));
  const unsigned cdo = smem_lds + (unsigned)(W1_BYTES + W2_BYTES + 16 * hi);
  const char* const w1g = reinterpret_cast<const char*>(p.w1);
  const char* const w2g = reinterpret_cast<const char*>(p.w2p);
  const char* const cdg = reinterpret_cast<const char*>(p.cd);
  const unsigned w1_chunk = (unsigned)(64 * p.ldw1 * 2);         // bytes of 64 packed W1 rows
#pragma unroll
  for (int u = 0; u < 2; ++u) c.w1b[u] = w1g + (size_t)u * c.w1_ustride;
#pragma unroll
  for (int t = 0; t < 5; ++t) c.w2b[t] = w2g + (size_t)t * c.w2_tstride;
  c.cdb = cdg;

  // iteration j of a tile: reads W1(j + 1) [slot (j + 1) & 1], the constants of chunk j [slot j & 1] and W2(j - 1) [slot (j + 1) & 1];
  // its LDS-DMA pieces bring W1(j + 2) [slot j & 1], the constants of chunk j + 1 [slot (j + 1) & 1] and W2(j) [slot j & 1]
  auto set_iter = [&](int j) {
    const unsigned sj = (unsigned)((j & 1) * SLOT_BYTES), sn = (unsigned)(((j + 1) & 1) * SLOT_BYTES);
#pragma unroll
    for (int i = 0; i < 4; ++i) c.w1a[i] = w1o[i] + sn;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) c.w2a[kk] = w2o[kk] + sn;
    c.cda = cdo + sj;
    const int j2 = j + 2 >= MLP_NCH ? j + 2 - MLP_NCH : j + 2, j1 = j + 1 >= MLP_NCH ? j + 1 - MLP_NCH : j + 1;
    c.w1src = w1g + (size_t)j2 * w1_chunk;
    c.w2src = w2g + (size_t)j * 64;
    c.cdsrc = cdg + (size_t)j1 * 512;
    c.w1_vj = c.w1_voff + (unsigned)j2 * w1_chunk; c.w2_vj = c.w2_voff + (unsigned)j * 64u; c.cd_vj = c.cd_voff + (unsigned)j1 * 512u;
    c.w1dst = smem_lds + sj + (unsigned)(wave * 1024);
    c.w2dst = smem_lds + sj + (unsigned)(W1_BYTES + wave * 1024);
    c.cddst = smem_lds + sn + (unsigned)(W1_BYTES + W2_BYTES);
  };

  int tile = ((G & 7) == 0) ? (int)(blockIdx.x & 7) * (G >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  if (tile >= tiles) return;
  const float gate = p.gate ? p.gate[0] : 1.0f;

  // kernel prologue: W1 of chunks 0 and 1 and the constants of chunk 0 (iteration 0 brings W2(0), W1(2), constants(1))
  {
#pragma unroll
    for (int ch = 0; ch < 2; ++ch)
#pragma unroll
      for (int kt = 0; kt < 5; ++kt)
#pragma unroll
        for (int u = 0; u < 2; ++u)
          mlp_dma16(w1g + (size_t)ch * w1_chunk + u * c.w1_ustride + kt * 128, c.w1_voff,
                    smem_lds + (unsigned)(ch * SLOT_BYTES + wave * 1024 + kt * 8192 + u * 4096));
    if (wave == 3) mlp_dma16(cdg, c.cd_voff, smem_lds + (unsigned)(W1_BYTES + W2_BYTES));
  }

  f32x16 acc1[2][2];
  u32x4 hh[2][2];
  // b2 once into LDS: the epilogue reads it with LDS latency and without a vector-memory operation of the compiler's own
  if (tid < MLP_C / 4) reinterpret_cast<f32x4*>(smem + MW_B2_OFF)[tid] = reinterpret_cast<const f32x4*>(p.b2)[tid];
#ifdef IDF_MLPW_TRACE
  for (int i = 0; i < 12; ++i) c.tr[i] = 0;
  unsigned long long trt = __builtin_readcyclecounter();
#define MW_TRT(i) { const unsigned long long now = __builtin_readcyclecounter(); c.tr[i] += now - trt; trt = now; }
#else
#define MW_TRT(i)
#endif
  // the first tile's rows; every later tile's arrive during the epilogue of the tile before (below)
  auto row_ptr = [&](int t) { return p.x + (size_t)(t * MLP_BM + wave * 32 + l31) * p.ldx + 8 * hi; };
  {
    const unsigned short* xr = row_ptr(tile);
    mw_static_for<20>([&](auto kc) { mw_load_x<decltype(kc)::value>(xr); });
  }
  f32x2 st = *reinterpret_cast<const f32x2*>(p.ln_stats + 2 * (size_t)(tile * MLP_BM + wave * 32 + l31));
  asm volatile("" : "+v"(st));                     // (arrived before the loop: see the pin inside the epilogue)
  bool first = true;
  for (;;) {
    c.nmu2 = f32x2{-st[0], -st[0]};
    c.rstd2 = f32x2{st[1], st[1]};
    asm volatile("" : "+v"(c.nmu2), "+v"(c.rstd2));
    mw_static_for<160>([&](auto rc) { mw_agpr_write<MW_OA + decltype(rc)::value>(0u); });
    if (first) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                              // the kernel prologue's pieces (and b2) are visible
      first = false;
    } else {
      // the x loads went out inside the epilogue; only the stores of its last two fragments (4) are younger -- vector-memory
      // operations retire in order: no wait for the store round trip
#ifdef MW_DIRECT_STORES
      asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
#else
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
#endif
    }
    asm volatile("" ::: "memory");

    MW_TRT(7)
    set_iter(-1);
    mw_pro<DT>(acc1[1], acc1[0], hh[1], hh[0], c);                 // first product of chunk 0 -> acc1[0]
    set_iter(0);
    mw_body_01<DT>(acc1[0], acc1[1], hh[1], hh[0], c);
    MW_TRT(9)
    for (int j = 1; j < MLP_NCH - 1; j += 2) {
      set_iter(j);
      mw_body_11<DT>(acc1[1], acc1[0], hh[0], hh[1], c);
      set_iter(j + 1);
      mw_body_11<DT>(acc1[0], acc1[1], hh[1], hh[0], c);
    }
    MW_TRT(10)
    set_iter(MLP_NCH - 1);
    mw_body_10<DT>(acc1[1], acc1[0], hh[0], hh[1], c);
    set_iter(MLP_NCH);
    mw_drain<DT>(acc1[0], acc1[1], hh[1], hh[0], c);
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 7" ::: "memory");  // the last MFMAs' results reach the AGPRs
    MW_TRT(9)

    // ---- tile epilogue: + b2, gate, + residual (out of the x fragments), 16-bit store: a lane holds 16 consecutive columns of
    // its row (32 bytes).  As soon as a pair of x fragments has given its residual, the NEXT tile's rows are fetched into it.
    const int next = tile + G;
    const bool has_next = next < tiles;
    const unsigned short* const xn = row_ptr(has_next ? next : tile);
    if (has_next) st = *reinterpret_cast<const f32x2*>(p.ln_stats + 2 * (size_t)(next * MLP_BM + wave * 32 + l31));
#ifdef MW_DIRECT_STORES
    unsigned short* const orow = p.out + (size_t)(tile * MLP_BM + wave * 32 + l31) * p.ldo + 16 * hi;
#else
    // through the wave's own LDS slots (two, alternating; LDS operations of one wave execute in order: no wait between the
    // write and the read) so that a store instruction covers 16 rows x 64 contiguous bytes
    char* const stg = smem + MW_STG_OFF + wave * 4096;
    const int sl_row = lane >> 2, sl_pc = lane & 3;
    auto stg_f = [](int row) { return ((((row >> 2) ^ (row >> 3)) & 1) << 1) | (((row >> 1) ^ (row >> 3) ^ (row >> 4)) & 1); };
    auto stg_at = [&](int slot, int row, int pc) { return reinterpret_cast<u32x4*>(stg + slot * 2048 + row * 64 + ((pc ^ stg_f(row)) << 4)); };
    unsigned short* const orow = p.out + (size_t)(tile * MLP_BM + wave * 32 + sl_row) * p.ldo + sl_pc * 8;
    u32x4 po0, po1;
#endif
    const char* const b2l = smem + MW_B2_OFF + 64 * hi;
    mw_static_for<10>([&](auto ac) {
      constexpr int a = decltype(ac)::value;
      // (the next tile's statistics are pinned HERE, where 18 of this epilogue's stores are younger than their load: carried
      // into the next iteration as a pending load, the compiler waits for them -- and for every store -- with vmcnt(0) at the loop head)
      if constexpr (a == 9) asm volatile("" : "+v"(st));
      float acc[16];
      mw_static_for<16>([&](auto rc) { acc[decltype(rc)::value] = __uint_as_float(mw_agpr_read<MW_OA + 16 * a + decltype(rc)::value>()); });
      float v[16];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const auto s02 = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[e]), __float_as_uint(acc[8 + e]), false, false);
        const auto s13 = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[4 + e]), __float_as_uint(acc[12 + e]), false, false);
        v[e] = __uint_as_float(s02[0]); v[4 + e] = __uint_as_float(s02[1]);
        v[8 + e] = __uint_as_float(s13[0]); v[12 + e] = __uint_as_float(s13[1]);
      }
#pragma unroll
      for (int jq = 0; jq < 4; ++jq) {
        const f32x4 bq = *reinterpret_cast<const f32x4*>(b2l + (32 * a + 4 * jq) * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[4 * jq + e] += bq[e];
      }
      // residual columns 32 a + 16 hi .. + 15 of the lane's row: x fragments 2 a (hi = 0 lanes) / 2 a + 1 (hi = 1 lanes), whose
      // second / first 8 elements sit in the other half-wave's registers
      u32x4 r0, r1;
      mw_static_for<4>([&](auto dc) {
        constexpr int d = decltype(dc)::value;
        const unsigned xa = mw_agpr_read<MW_XA + 4 * (2 * a) + d>(), xb = mw_agpr_read<MW_XA + 4 * (2 * a + 1) + d>();
        const auto sw = __builtin_amdgcn_permlane32_swap(xa, xb, false, false);
        r0[d] = sw[0]; r1[d] = sw[1];
      });
#ifndef MW_NO_XLOAD              /* (timing experiment: wrong results) */
      if (has_next) { mw_load_x<2 * a>(xn); mw_load_x<2 * a + 1>(xn); }
#endif
      float r[16];
      unpack8<DT>(r0, r);
      unpack8<DT>(r1, r + 8);
#pragma unroll
      for (int jq = 0; jq < 16; ++jq) v[jq] = fmaf(gate, v[jq], r[jq]);
#ifdef MW_DIRECT_STORES
      *reinterpret_cast<u32x4*>(orow + 32 * a) = pack8<DT>(v);
      *reinterpret_cast<u32x4*>(orow + 32 * a + 8) = pack8<DT>(v + 8);
#else
      // (the read-back of fragment a is stored one fragment later, behind fragment a + 1's register traffic: the LDS round
      // trip of every fragment was exposed otherwise)
      if constexpr (a > 0) {
        *reinterpret_cast<u32x4*>(orow + 32 * (a - 1)) = po0;
        *reinterpret_cast<u32x4*>(orow + (size_t)16 * p.ldo + 32 * (a - 1)) = po1;
      }
      *stg_at(a & 1, l31, 2 * hi) = pack8<DT>(v);
      *stg_at(a & 1, l31, 2 * hi + 1) = pack8<DT>(v + 8);
      asm volatile("" ::: "memory");
      po0 = *stg_at(a & 1, sl_row, sl_pc); po1 = *stg_at(a & 1, sl_row + 16, sl_pc);
      if constexpr (a == 9) {
        *reinterpret_cast<u32x4*>(orow + 32 * a) = po0;
        *reinterpret_cast<u32x4*>(orow + (size_t)16 * p.ldo + 32 * a) = po1;
      }
#endif
    });
    MW_TRT(8)
    if (!has_next) break;
    tile = next;
  }
#ifdef IDF_MLPW_TRACE
  if ((int)blockIdx.x == (int)gridDim.x / 2 && lane == 0) { for (int i = 0; i < 12; ++i) idf_mlpw_trace_buf[wave][i] = c.tr[i]; }
#endif
}

template <int DT>
int launch_mlp320w(const MlpParams& p, hipStream_t s) {
  void (*kern)(const MlpParams, const int) = mlp320w_kernel<DT>;
  static std::atomic<unsigned long long> attr_done{0};
  if (const int e = idf_lds_optin(reinterpret_cast<const void*>(kern), MW_SMEM, attr_done)) return e;
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
  const int tiles = p.M / MLP_BM;
  const int grid = tiles < cus ? tiles : cus;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), MW_SMEM, s, p, tiles);
  return idf_launch_status();
}

// IDF_MLP_MODE: 0 = mlp320_kernel (two waves per SIMD), 1 = mlp320w_kernel (one instruction stream per SIMD)
#ifndef IDF_MLP_MODE_DEFAULT
#define IDF_MLP_MODE_DEFAULT 1
#endif
int g_mlp_mode = -1;
inline int mlp_mode() {
  if (g_mlp_mode < 0) { const char* e = getenv("IDF_MLP_MODE"); g_mlp_mode = e ? (e[0] == '0' ? 0 : 1) : IDF_MLP_MODE_DEFAULT; }
  return g_mlp_mode;
}

}  // namespace

// idf_set_tuning(IDF_TUNE_MLP, v): returns the previous mode
int idf_mlp_set_mode(int v) {
  const int prev = mlp_mode();
  g_mlp_mode = v;
  return prev;
}

#ifdef IDF_MLPW_TRACE
extern "C" int idf_mlpw_trace_read(unsigned long long* host /* [4][12] */) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(idf_mlpw_trace_buf), sizeof(idf_mlpw_trace_buf));
}
#endif

#ifdef IDF_MLP_TRACE
extern "C" int idf_mlp_trace_read(unsigned long long* host /* [4][10] */) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(idf_mlp_trace_buf), sizeof(idf_mlp_trace_buf));
}
#endif

extern "C" int idf_mlp_geglu(const idf_mlp_args* a, void* stream) {
  if (!a) return IDF_E_ARG;
  if (!a->x || !a->ln_stats || !a->w1 || !a->cd || !a->w2p || !a->b2 || !a->out) return IDF_E_ARG;
  if (a->M <= 0 || a->C <= 0) return IDF_E_ARG;
  if (a->dtype != IDF_BF16 && a->dtype != IDF_F16) return IDF_E_ARG;
  if (a->C != MLP_C || (a->M % MLP_BM) != 0) return IDF_E_UNSUPPORTED;
  if (a->ldx < MLP_C || a->ldo < MLP_C || a->ldw1 < MLP_C || a->ldw2 < MLP_H) return IDF_E_ARG;
  if ((a->ldx % 8) || (a->ldo % 8) || (a->ldw1 % 8) || (a->ldw2 % 8)) return IDF_E_ALIGN;
  if (!aligned16(a->x) || !aligned16(a->out) || !aligned16(a->w1) || !aligned16(a->w2p) || !aligned16(a->cd) || !aligned16(a->b2) ||
      (((uintptr_t)a->ln_stats) & 7u))
    return IDF_E_ALIGN;
  // 32-bit per-lane DMA offsets
  if ((long long)64 * a->ldw1 * 2 >= (1ll << 31) || (long long)320 * a->ldw2 * 2 >= (1ll << 31)) return IDF_E_UNSUPPORTED;
  MlpParams p;
  p.x = static_cast<const unsigned short*>(a->x); p.ldx = a->ldx; p.ln_stats = a->ln_stats;
  p.w1 = static_cast<const unsigned short*>(a->w1); p.ldw1 = a->ldw1; p.cd = a->cd;
  p.w2p = static_cast<const unsigned short*>(a->w2p); p.ldw2 = a->ldw2; p.b2 = a->b2; p.gate = a->gate;
  p.out = static_cast<unsigned short*>(a->out); p.ldo = a->ldo; p.M = a->M;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (mlp_mode() == 1) return a->dtype == IDF_BF16 ? launch_mlp320w<IDF_BF16>(p, s) : launch_mlp320w<IDF_F16>(p, s);
  return a->dtype == IDF_BF16 ? launch_mlp320<IDF_BF16>(p, s) : launch_mlp320<IDF_F16>(p, s);
}
