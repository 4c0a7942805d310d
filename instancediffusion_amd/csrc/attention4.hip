// attention4.hip -- attention forward, variant 4 (gfx950): the 64x64-latent self / gated self-attention (d = 40).
//
// Same data layout as variant 2 (attention2.hip): 64 queries per wave in two groups of 32, swapped K.Q^T so a lane owns
// a query column, K / V^T tiles staged by LDS-DMA, permuted K fragment rows so the packed P feeds P.V without lane
// exchanges, softmax denominator from an all-ones V^T row.  What changed, and why (ISA notes in profiles/DESIGN_r01_r05_full.md on variant 2:
// 253 VGPRs, 7 of 14 ds_read_b128 directly followed by s_waitcnt lgkmcnt(0), a 34-deep v_max3 chain and 18 v_mov per
// tile, K/V re-fetched ~8x because the 16 query blocks of a (batch, head) were spread over the 8 XCD L2s):
//
//   * NO running max.  Q is pre-multiplied by scale*log2(e); the reference value m of a query (fixed by an exact pass
//     over its first tile) is subtracted BY THE MFMA through the spare half K-step that d = 40 leaves in the third
//     16-wide K-step: lanes hi = 1 of that step feed K = {1, 0, ..., 0} (one constant LDS fragment) against
//     Q = {-m, 0, ..., 0}, so the accumulator starts from the inline constant 0 (no -m register block, no v_mov) and comes
//     out as (score - m) in log2 units.  m is rounded to the 16-bit storage type, which is harmless: softmax is shift
//     invariant and the SAME m enters every P of the query and its denominator.
//   * Overflow guard without a max: m is chosen so that the largest P of the first tile is 2^-SHIFT; after packing, the
//     32 packed P words of a lane are OR-ed (16 v_or3) and bit 14 of either half (<=> some P >= 2, in bf16 and fp16
//     alike) sends the WAVE to the exact path, which recomputes the tile from the K tile still in LDS, raises m for the
//     queries that grew and rescales their O rows.  The common tile costs {64 v_exp, 32 v_cvt_pk, 16 v_or3} of VALU.
//   * No tail masking arithmetic: K rows beyond n are clamped duplicates of the last valid key (a valid score), their V^T
//     columns come from a page of zeros and, for a tile with a tail, the all-ones row is re-staged with zeros in the
//     invalid columns -- so the tail keys add nothing to O or to the denominator.
//   * LDS latency off the critical path: the V^T fragments of tile t are read right behind the barrier, under the K.Q^T
//     MFMAs and the exponentials; the K fragments of tile t+1 are read under the P.V MFMAs of tile t (K is fetched TWO
//     tiles ahead into a 3-stage ring, V^T one tile ahead into a 2-stage ring), so K.Q^T starts from registers.
//   * One s_barrier per tile, DMA issued right behind it (scalar base + constant per-lane offset, no per-tile address
//     VALU): a tile's loads have a whole tile of compute to land.
//   * 1-D grid, XCD-aware: the query blocks of one (batch, head) are consecutive on ONE XCD, so its K / V^T
//     (685 KB at 4280 keys) stay in that XCD's L2 while its 16 blocks run.
// Numerics contract as before: fp32 scores / accumulators, P rounded to the 16-bit type before P.V, denominator sums the
// rounded P.  Requirements (else IDF_ATTN2_UNSUPPORTED and the caller falls back): d in {24, 40, 56}, n % 8 == 0, aligned.
#include "attn_core.h"
#include <cstdlib>

using namespace idfattn;

namespace {

__device__ __attribute__((aligned(128))) unsigned short idf_attn4_zero_page[64];
__device__ __attribute__((aligned(16))) unsigned short idf_attn4_ones_page[2][8] = {
    {0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80},      // bf16 1.0
    {0x3c00, 0x3c00, 0x3c00, 0x3c00, 0x3c00, 0x3c00, 0x3c00, 0x3c00}};     // fp16 1.0

constexpr int KVT = 64;

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// LDS-DMA issued through inline asm: the compiler's waitcnt pass otherwise puts s_waitcnt vmcnt(0) in front of the first
// ds_read that follows ANY pending global_load_lds (it cannot tell the ring stages apart), which would serialise the
// prefetch with the tile's own LDS reads.  Ordering is ours: `s_waitcnt vmcnt(0)` + s_barrier at the end of every tile.
// lds = LDS byte address of lane 0's 16-B slot (lane i lands at lds + 16 i); it goes through M0.
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(size_t)p; }
__device__ __forceinline__ void dma16_sv(const void* sbase /* wave-uniform */, unsigned voff, unsigned lds) {
  lds = __builtin_amdgcn_readfirstlane(lds);         // wave-uniform by construction; make it provably so (an SGPR for M0)
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds), "v"(voff), "s"(sbase) : "memory");   // M0 is ours here: nothing else in this kernel uses it (no movrel / GWS / sendmsg)
}
__device__ __forceinline__ void dma16_v(const void* addr /* per lane */, unsigned lds) {
  lds = __builtin_amdgcn_readfirstlane(lds);
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds), "v"(addr) : "memory");   // M0 is ours here: nothing else in this kernel uses it (no movrel / GWS / sendmsg)
}

struct TrueT { static constexpr bool value = true; };
struct FalseT { static constexpr bool value = false; };

template <int DT> struct RefShift;           // after a re-base the largest P of a query is 2^-SHIFT (trigger: P >= 2)
template <> struct RefShift<IDF_BF16> { static constexpr float v = 7.0f; };    // bf16: 8 exponent bits, shift is free
// fp16 (round 5): 7, as bf16, instead of 1.  With 1 a wave left the common path whenever some score of a tile exceeded the
// reference by 2 log2 units -- routine on real score distributions, and the reason the fp16 leg of the bench ran 5.6 % behind bf16
// (harness at q / k amplitude 5: 641 TF with shift 1, 741 / 801 / 840 with 3 / 5 / 7, profiles/r05_attn8_first.log).
// With 7 the largest P after a re-base is 2^-7: P below 2^-14 (2^-7 of the largest) become fp16 denormals, which the conversion
// produces and the MFMA consumes exactly; their absolute rounding error (<= 2^-25) is far below the 2^-11 relative rounding of
// the dominant P (2^-18 absolute), and the measured error against the fp32 reference is the same for every shift
// (rel-RMS 4.6e-4 .. 4.9e-4 at amplitude 5, 5.0e-4 .. 5.4e-4 at 8).
#ifndef IDF_ATTN4_SHIFT_F16
#define IDF_ATTN4_SHIFT_F16 7.0f
#endif
template <> struct RefShift<IDF_F16> { static constexpr float v = IDF_ATTN4_SHIFT_F16; };

// VA = how many tiles ahead V^T is fetched (1: 2-stage ring, wait for everything at the end of a tile; 2: 3-stage ring like K,
// and the end-of-tile wait leaves the loads issued in THIS tile in flight -- they have two tiles to land).
// NW = waves per workgroup (64 queries each).  4: two independent workgroups per CU (their phases drift apart, which overlaps
// one's MFMAs with the other's exponentials).  8: ONE workgroup per CU sharing every K / V^T tile among twice as many waves --
// an LDS-DMA instruction holds its issuing wave 55-72 cycles (tools/ubench/dma_mix.hip), and with 11 KB per tile that is
// 2.75 instructions per wave and tile at NW = 4, 1.4 at NW = 8 (mode 2: +1 % in isolation, -3 % inside the forward; NW = 4 stays
// the default -- profiles/r03_attn_ab*_B64.log, r03_shape_profile_B64_attn{1,2}.log).
// Optional per-segment cycle trace (a second library build with -DIDF_ATTN_TRACE, read through idf_attn_trace_read by
// tools/ubench/attn_harness.hip; the shipped library has none of it): s_memtime deltas of wave 0 of the first and of a middle
// workgroup, summed over the common-path tiles.  Segments: 0 LDS-DMA issue, 1 K.Q^T MFMA issue (both query groups),
// 2 exp + pack of group 0, 3 P.V of group 0, 4 exp + pack of group 1, 5 next tile's K fragment reads, 6 P.V of group 1,
// 7 vmcnt wait + workgroup barrier, 8 overflow check, 9 number of tiles.
#ifdef IDF_ATTN_TRACE
__device__ unsigned long long idf_attn_trace_buf[2][10];
#define ATR_DECL unsigned long long tr_last = __builtin_readcyclecounter(), tr_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
// -DIDF_ATTN_TRACE=2: two time stamps per tile only (segments 1 = top of the tile .. K.Q^T MFMAs issued, 7 = the rest): the full
// set costs the kernel 46 % (every stamp drains lgkmcnt), this one a few per cent
#define ATR_ON(i) (IDF_ATTN_TRACE != 2 || (i) == 1 || (i) == 7)
#define ATR(i) { if constexpr (ATR_ON(i)) { const unsigned long long tr_now = __builtin_readcyclecounter(); tr_acc[i] += tr_now - tr_last; tr_last = tr_now; } }
#define ATR_RESET { if constexpr (IDF_ATTN_TRACE != 2) tr_last = __builtin_readcyclecounter(); }
#define ATR_COUNT(i) { tr_acc[i] += 1; }
#define ATR_DUMP { const int trb = blockIdx.x == 0 ? 0 : ((int)blockIdx.x == (int)gridDim.x / 2 ? 1 : -1);                     \
    if (trb >= 0 && tid == 0) { for (int i = 0; i < 10; ++i) idf_attn_trace_buf[trb][i] = tr_acc[i]; } }
#else                      // (empty BLOCKS, not empty macros: `if constexpr (..) ATR(1)` must not swallow the next statement)
#define ATR_DECL
#define ATR(i) {}
#define ATR_RESET {}
#define ATR_COUNT(i) {}
#define ATR_DUMP {}
#endif

template <int DT, int NKS, int NMT, int VA, int NW = 4>
__global__ __launch_bounds__(NW * 64, 2) void attn4_kernel(const AttnParams p, const int nqb, const int xcd_order) {
  constexpr int DCH = 2 * NKS - 1;                 // 16-B chunks per K row
  constexpr int D = 8 * DCH;                       // head dim
  constexpr int KSZ = KVT * D;                     // K stage (elements), linear rows of D*2 bytes (D/8 odd: conflict-free)
  constexpr int VROWS = NMT * 32;
  constexpr int VSZ = VROWS * KVT;                 // V^T stage (elements), 128-B rows, 16-B slot ^= (row >> 1) & 7
  constexpr int K_INST = DCH;                      // LDS-DMA instructions per K tile (64 chunks each)
  constexpr int V_INST = D / 8;                    // per V^T tile (8 rows each); instruction V_INST = the ones-row group
  constexpr int NT = NW * 64;                      // threads per workgroup
  constexpr int K_PER_WAVE = (K_INST + NW - 1) / NW, V_PER_WAVE = (V_INST + NW - 1) / NW;
  static_assert(D < 32 * NMT && (D % 8) == 0 && D + 8 <= VROWS, "needs a spare 8-row group for the softmax denominator");
  constexpr int VST = VA + 1;                      // V^T ring stages
  constexpr int RING = 3 * KSZ + VST * VSZ + 8, OSTAGE = NW * 64 * D;        // K / V^T rings; O staging block of the epilogue
  __shared__ __attribute__((aligned(128))) unsigned short smem[RING > OSTAGE ? RING : OSTAGE];
  __shared__ int redo_flag;                        // some wave met an inf / nan P: redo the block with the exact per-tile max
  unsigned short* const Ks = smem;
  unsigned short* const Vs = smem + 3 * KSZ;
  unsigned short* const ones_frag = smem + 3 * KSZ + VST * VSZ;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;

  // ---- XCD-aware block order: hardware block L runs on XCD L % 8; give every XCD a contiguous range of logical blocks
  int L = blockIdx.x;
  {
    const int total = gridDim.x;
    if ((xcd_order & 1) && (total & 7) == 0) L = (L & 7) * (total >> 3) + (L >> 3);
  }
  const int qb = L % nqb;
  const int h = (L / nqb) % p.H;
  const int b = L / (nqb * p.H);
#ifdef IDF_ATTN_EXP
  // experiment (tools/ubench/attn_harness.hip, env IDF_ATTN_EXP = sleep_units | prio << 8): the second workgroup of every CU
  // (hardware blocks [256, 512) of the first dispatch round; later blocks inherit the slot phase of the one they replace) starts
  // `sleep_units` x 64 cycles late and / or runs at s_setprio 1
  {
    const int exp = xcd_order >> 8;
    if (((blockIdx.x >> 8) & 1) != 0) {
      for (int i = 0; i < (exp & 0xff); ++i) __builtin_amdgcn_s_sleep(1);
      if ((exp >> 8) & 1) __builtin_amdgcn_s_setprio(1);
    }
  }
#endif

  // zero the V^T ring once (pad rows of the O^T tile must be finite zeros), then the ones row and the ones fragment
  for (int i = tid; i < VST * VSZ / 2; i += NT) reinterpret_cast<unsigned*>(Vs)[i] = 0u;
  __syncthreads();
  {
    const unsigned short one = Elem<DT>::from_f32(1.0f);
    for (int i = tid; i < VST * KVT; i += NT) Vs[(i / KVT) * VSZ + D * KVT + (i % KVT)] = one;
    if (tid < 8) ones_frag[tid] = tid == 0 ? one : (unsigned short)0;
    if (tid == 0) redo_flag = 0;
  }

  // ---- Q fragments (B operand) of the two query groups, pre-multiplied by scale*log2(e): lane holds q = l31,
  // e = 16*ks + 8*hi .. +7.  Element D (first element of the hi = 1 half of the last K-step) carries -m.
  u32x4 qf[2][NKS];
  int qrow[2];
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    qrow[g] = qb * NT + wave * 64 + g * 32 + l31;
    const int qr = min(qrow[g], p.nq - 1);
    const unsigned short* qp = p.q + (size_t)b * p.sQ + (size_t)qr * p.ldq + h * D;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      const int e0 = ks * 16 + hi * 8;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (e0 < D) {
        v = *reinterpret_cast<const u32x4*>(qp + e0);
        float f[8];
        unpack8<DT>(v, f);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] *= p.scale_log2;
        v = pack8<DT>(f);
      }
      qf[g][ks] = v;
    }
  }

  const int T0 = (p.n[0] + KVT - 1) / KVT;
  const int T1 = (p.n[1] + KVT - 1) / KVT;
  const int T = T0 + T1;

  // ---- DMA roles.  K: instruction i moves linear chunks 64 i .. 64 i + 63 of the tile: chunk c -> row c / DCH, column chunk
  // c % DCH.  V^T: instruction i moves rows 8 i .. 8 i + 7: lane -> row 8 i + (lane >> 3), LDS slot lane & 7 (holding the
  // global 8-key chunk slot ^ ((row >> 1) & 7)).  K instruction i is issued by wave i % 4, V^T instruction i by wave
  // (i + 1) % 4 (3 / 3 / 2 / 2 per wave at d = 40).  Full tiles: uniform base (SGPR) + a per-lane byte offset that only
  // depends on the segment -- no per-tile address arithmetic on the VALU.
  const int vwave = (wave + NW - 1) % NW;            // this wave issues V^T instructions vwave, vwave + NW
  // Only the segment-0 full-tile offsets of the steady-state loop are kept in registers.  Everything the other tiles need
  // (first tiles, segment change, tails, the ones row) is RECOMPUTED inside issue_k / issue_v from an opaque copy of the lane
  // id: left to the compiler, those address computations were hoisted out of the loops and kept alive across the hot loop
  // -- 72 spilled VGPRs, i.e. 288 B of scratch per lane written by every wave's prologue: 600 MB per launch at batch 64,
  // which was the "4.6x write amplification" of profiles/r02_rocprof/pmc_traffic_b64.json (not the store width).
  unsigned koff0[K_PER_WAVE], voff0[V_PER_WAVE];
#pragma unroll
  for (int j = 0; j < K_PER_WAVE; ++j) {
    const int c = (wave + NW * j) * 64 + lane;
    const int row = c / DCH, col = (c - row * DCH) * 8;
    koff0[j] = (unsigned)(row * p.ldk[0] + col) * 2u;
  }
#pragma unroll
  for (int j = 0; j < V_PER_WAVE; ++j) {
    const int row = (vwave + NW * j) * 8 + (lane >> 3);
    voff0[j] = (unsigned)(row * p.ldv[0] + ((lane & 7) ^ ((row >> 1) & 7)) * 8) * 2u;
  }
  auto cold_lane = [&]() { int l = lane; asm volatile("" : "+v"(l)); return l; };

  auto issue_k = [&](int t) {
    const int ln = cold_lane();
    const int seg = (t < T0) ? 0 : 1;
    const int kv0 = (seg ? (t - T0) : t) * KVT;
    const int n = p.n[seg];
    const int ldk = p.ldk[seg];
    const char* kb = reinterpret_cast<const char*>(p.k[seg] + (size_t)b * p.sK[seg] + h * D);
    unsigned short* dst = Ks + (t % 3) * KSZ;
    const bool full = kv0 + KVT <= n;
#pragma unroll
    for (int j = 0; j < K_PER_WAVE; ++j)
      if (wave + NW * j < K_INST) {
        const int c = (wave + NW * j) * 64 + ln;
        const int row = c / DCH, col = (c - row * DCH) * 8;
        // tail tile: rows beyond n are clamped to the last valid key
        const int kr = full ? kv0 + row : min(kv0 + row, n - 1);
        dma16_v(kb + ((size_t)kr * ldk + col) * 2, lds_addr(dst + (wave + NW * j) * 512));
      }
  };
  // the ones-row group (rows D .. D+7 of the V^T image): row D = ones in the valid columns, zeros elsewhere
  auto issue_ones = [&](int stage, int nvalid) {
    if (wave == NW / 2) {
      const int ln = cold_lane();
      const int row = D + (ln >> 3);
      const int chunk = (ln & 7) ^ ((row >> 1) & 7);
      const bool one = (row == D) && (chunk * 8 < nvalid);
      const unsigned short* src = one ? idf_attn4_ones_page[DT == IDF_BF16 ? 0 : 1] : idf_attn4_zero_page + (ln & 7) * 8;
      dma16_v(src, lds_addr(Vs + stage * VSZ + V_INST * 512));
    }
  };
  auto issue_v = [&](int t) {
    const int ln = cold_lane();
    const int seg = (t < T0) ? 0 : 1;
    const int kv0 = (seg ? (t - T0) : t) * KVT;
    const int n = p.n[seg];
    const int ldv = p.ldv[seg];
    const char* vb = reinterpret_cast<const char*>(p.vt[seg] + (size_t)b * p.sV[seg] + (size_t)(h * D) * ldv);
    unsigned short* dst = Vs + (t % VST) * VSZ;
    const char* base = vb + (size_t)kv0 * 2;
#pragma unroll
    for (int j = 0; j < V_PER_WAVE; ++j)
      if (vwave + NW * j < V_INST) {
        const int row = (vwave + NW * j) * 8 + (ln >> 3);
        const int chunk = (ln & 7) ^ ((row >> 1) & 7);
        // tail tile: 8-key chunks beyond n (n % 8 == 0) come from the zero page
        const bool valid = (kv0 + chunk * 8) < n;
        const char* src = valid ? base + ((size_t)row * ldv + chunk * 8) * 2
                                : reinterpret_cast<const char*>(idf_attn4_zero_page + (ln & 7) * 8);
        dma16_v(src, lds_addr(dst + (vwave + NW * j) * 512));
      }
    // the ones row of this stage: restrict it for a tail tile, restore it when the stage last held a tail tile (tile t-VST)
    const bool tail = (kv0 + KVT > n);
    bool prev_tail = false;
    if (t >= VST) {
      const int t2 = t - VST;
      const int s2 = (t2 < T0) ? 0 : 1;
      prev_tail = ((s2 ? (t2 - T0) : t2) + 1) * KVT > p.n[s2];
    }
    if (tail || prev_tail) issue_ones(t % VST, tail ? n - kv0 : KVT);
  };

  f32x16 o[2][NMT];
  float m_run[2];                                   // the reference value m of the lane's query, 16-bit representable
  const int v_sw = (l31 >> 1) & 7;                  // V^T fragment rows are mt*32 + l31
  // K fragment row permutation (see attention2.hip): MFMA row i of a 32-key half carries key (i with bits 2 and 3
  // swapped), so the 8 S^T registers of a lane-half per 16-key step are 8 CONSECUTIVE keys = the k order of P.V.
  const int kperm = (l31 & 0x13) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);
  const int kfoff = kperm * D + hi * 8;             // element offset of the lane's K fragment inside a 32-key half
  const int vfoff = l31 * KVT;                      // V^T fragment row offset

  // `hi_o`: an opaque copy of `hi` (made once per tile, in front of its MFMAs): the hi ? ones_frag : row select below is then
  // made per call (2 v_cndmask per tile) instead of being hoisted as six per-stage address registers that live across the
  // whole kernel (and were spilled)
  auto load_kf = [&](u32x4 (&dst)[2][NKS], int stage, int hi_o) {
    const unsigned short* Kc = Ks + stage * KSZ;
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      const unsigned short* base = Kc + st * 32 * D + kfoff;
#pragma unroll
      for (int ks = 0; ks < NKS - 1; ++ks) dst[st][ks] = *reinterpret_cast<const u32x4*>(base + ks * 16);
      // last K-step: hi = 0 lanes read elements 16*(NKS-1) .. +7 of the row, hi = 1 lanes the constant {1, 0, .., 0}
      const unsigned short* last = hi_o ? ones_frag : base + (NKS - 1) * 16;
      dst[st][NKS - 1] = *reinterpret_cast<const u32x4*>(last);
    }
  };
  ATR_DECL
  f32x16 s[2][2];                                    // [query group][kv half]
  u32x4 pk[2][4];                                    // packed P: [group][16-key step]
  u32x4 kf[2][NKS];                                  // K fragments of the CURRENT tile, read one tile ahead
  auto qk = [&](const int g) {
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
      for (int st = 0; st < 2; ++st) s[g][st] = Elem<DT>::mfma32(kf[st][ks], qf[g][ks], ks == 0 ? zero : s[g][st]);
  };
  // raise the reference value of group g's queries by (want > 0 ? want : 0) [first: by want], rounded so that m stays
  // 16-bit representable; rescale O (its row D is the denominator) and rewrite the -m element of Q.  Returns the shift.
  auto raise_m = [&](const int g, const float want, const bool first) -> float {
    const float delta = first ? want : fmaxf(want, 0.0f);
    const float m_new = Elem<DT>::to_f32(Elem<DT>::from_f32(m_run[g] + delta));
    const float d_eff = m_new - m_run[g];
    m_run[g] = m_new;
    const float al = first ? 1.0f : __builtin_amdgcn_exp2f(-d_eff);
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[g][mt][r] *= al;
    const unsigned neg_m = pack2<DT>(-m_new, 0.0f);
    qf[g][NKS - 1][0] = hi ? neg_m : qf[g][NKS - 1][0];
    return d_eff;
  };
  auto half_max = [&](float mx) -> float {           // max over the two lane halves that share a query
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
    return fmaxf(mx, __uint_as_float(hi ? sw[0] : sw[1]));
  };
  // exact pass over the scores of group g (already relative to the current m): the tile's max maps to 2^-SHIFT
  auto rebase_scores = [&](const int g, const bool first) {
    float m0 = fmaxf(s[g][0][0], s[g][0][1]), m1 = fmaxf(s[g][1][0], s[g][1][1]);
#pragma unroll
    for (int r = 2; r < 16; r += 2) {
      m0 = fmaxf(fmaxf(m0, s[g][0][r]), s[g][0][r + 1]);
      m1 = fmaxf(fmaxf(m1, s[g][1][r]), s[g][1][r + 1]);
    }
    const float d_eff = raise_m(g, half_max(fmaxf(m0, m1)) + RefShift<DT>::v, first);
#pragma unroll
    for (int st = 0; st < 2; ++st)
#pragma unroll
      for (int r = 0; r < 16; ++r) s[g][st][r] -= d_eff;
  };
  auto exp_pack = [&](const int g) -> unsigned {     // P = 2^s, packed; returns the OR of the packed words
#pragma unroll
    for (int st = 0; st < 2; ++st)
#pragma unroll
      for (int r = 0; r < 16; ++r) s[g][st][r] = __builtin_amdgcn_exp2f(s[g][st][r]);
    unsigned acc = 0u;
#pragma unroll
    for (int st = 0; st < 2; ++st)
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          const unsigned v = pack2<DT>(s[g][st][8 * k2 + 2 * w], s[g][st][8 * k2 + 2 * w + 1]);
          pk[g][st * 2 + k2][w] = v;
          acc |= v;
        }
    return acc;
  };
  // O^T(g) += V^T P^T(g): the V^T fragments stream through two register sets, one 16-key step ahead of their MFMAs
  auto pv = [&](const int g, const int stage) {
    const unsigned short* Vc = Vs + stage * VSZ + vfoff;
    u32x4 a[2][NMT];
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt) a[0][mt] = *reinterpret_cast<const u32x4*>(Vc + mt * 32 * KVT + ((hi ^ v_sw) * 8));
#pragma unroll
    for (int step = 0; step < 4; ++step) {
      if (step + 1 < 4) {
        const int chunk = (step + 1) * 2 + hi;               // 8-key chunk of the tile
#pragma unroll
        for (int mt = 0; mt < NMT; ++mt)
          a[(step + 1) & 1][mt] = *reinterpret_cast<const u32x4*>(Vc + mt * 32 * KVT + ((chunk ^ v_sw) * 8));
      }
#pragma unroll
      for (int mt = 0; mt < NMT; ++mt) o[g][mt] = Elem<DT>::mfma32(a[step & 1][mt], pk[g][step], o[g][mt]);
    }
  };
  // One tile.  EXACT: per-tile exact max (the classic online softmax; tile 0 and the fallback pass).  Otherwise the
  // common path: no max, no branch between the K.Q^T MFMAs and the last P.V MFMA; returns the OR of all packed P words.
  auto tile = [&](auto exact_tag, const int t) -> unsigned {
    constexpr bool EXACT = decltype(exact_tag)::value;
    int hi_o = hi;
    asm volatile("" : "+v"(hi_o));
    qk(0);
    qk(1);
    if constexpr (!EXACT) ATR(1)
    if constexpr (EXACT) {
      // keys beyond n in a tail tile are clamped duplicates of a valid key: they cannot raise the max
      rebase_scores(0, t == 0);
      rebase_scores(1, t == 0);
    }
    unsigned acc = exp_pack(0);
    if constexpr (!EXACT) ATR(2)
    pv(0, t % VST);
    if constexpr (!EXACT) ATR(3)
    acc |= exp_pack(1);
    if constexpr (!EXACT) ATR(4)
    load_kf(kf, (t + 1) % 3, hi_o);                  // next tile's K fragments (a stale stage after the last tile: unused)
    if constexpr (!EXACT) ATR(5)
    pv(1, t % VST);
    if constexpr (!EXACT) ATR(6)
#pragma unroll
    for (int st = 0; st < 2; ++st)
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) asm volatile("" : "+v"(kf[st][ks]));       // landed here, under the MFMAs
    return acc;
  };
  // End of a tile: the loads the NEXT tile needs have landed (this wave's share; the barrier publishes everyone's) and the
  // tile is retired.  VA == 1: everything this wave issued is needed next -> vmcnt(0).  VA == 2 and `counted`: the loads
  // issued at the top of THIS tile (this wave's n_mine K and V^T instructions for tile t+2: 3 on waves 0 / 1, 2 on waves
  // 2 / 3 at d = 40) may stay in flight; everything older -- tile t+1's -- must be there.
  int n_mine = 0;                                    // LDS-DMA instructions this wave issues per full tile (wave-uniform)
#pragma unroll
  for (int j = 0; j < K_PER_WAVE; ++j) n_mine += (wave + NW * j < K_INST) ? 1 : 0;
#pragma unroll
  for (int j = 0; j < V_PER_WAVE; ++j) n_mine += (vwave + NW * j < V_INST) ? 1 : 0;
  auto end_tile = [&](const bool counted) {
    if (VA == 2 && counted) {
      if (n_mine == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
      else if (n_mine == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      else if (n_mine == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
  };
  constexpr unsigned EXP_MASK = DT == IDF_BF16 ? 0x7f80u : 0x7c00u;     // all-ones exponent of a 16-bit half: inf / nan

  // rare path behind a tile of the common path: some P of this wave reached 2 (bit 14 of a half: bf16 exponent >= 128 /
  // fp16 exponent field >= 16).  Finite: the tile went into O correctly -- raise m for the queries that grew, using the
  // packed P still in pk.  Inf / nan: this wave's O is spoilt; flag the workgroup to redo the block with the exact max.
  auto after_tile = [&](const unsigned acc) {
    if (__builtin_amdgcn_ballot_w64((acc & 0x40004000u) != 0u) != 0) {
      bool bad = false;
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        float mx = 0.0f;
#pragma unroll
        for (int step = 0; step < 4; ++step)
#pragma unroll
          for (int w = 0; w < 4; ++w) {
            const unsigned v = pk[g][step][w];
            bad |= ((v & EXP_MASK) == EXP_MASK) | (((v >> 16) & EXP_MASK) == EXP_MASK);
            mx = fmaxf(mx, fmaxf(Elem<DT>::to_f32((unsigned short)(v & 0xffffu)), Elem<DT>::to_f32((unsigned short)(v >> 16))));
          }
        mx = half_max(mx);
        // growth beyond 2^40 in one tile is left to the exact pass as well: the rescale factor 2^-(log2 mx + SHIFT) must stay
        // a normal fp32 number (v_exp_f32 flushes denormal results to 0, which would wipe O AND its denominator row)
        bad |= !(mx <= 0x1p40f);
        raise_m(g, __builtin_amdgcn_logf(mx) + RefShift<DT>::v, false);           // v_log_f32 = log2; log2(0) = -inf: no raise
      }
      if (__builtin_amdgcn_ballot_w64(bad) != 0 && lane == 0) redo_flag = 1;
    }
  };

  const int F0 = p.n[0] / KVT;                       // full tiles of segment 0
  bool exact_all = false;
  for (;;) {
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      m_run[g] = 0.0f;
      qf[g][NKS - 1][0] = hi ? 0u : qf[g][NKS - 1][0];
#pragma unroll
      for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[g][mt][r] = 0.0f;
    }
    if (exact_all) {                                // the abandoned pass may have left a tail-restricted ones row behind
      const unsigned short one = Elem<DT>::from_f32(1.0f);
      for (int i = tid; i < VST * KVT; i += NT) Vs[(i / KVT) * VSZ + D * KVT + (i % KVT)] = one;
    }
    __syncthreads();                                // zero fill, ones row, ones fragment (or the abandoned pass) complete
    issue_k(0);
    issue_v(0);
    if (T > 1) issue_k(1);
    if (VA == 2 && T > 1) issue_v(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    load_kf(kf, 0, hi);

    // ---- tile 0: exact (fixes the reference value m of every query)
    if (T > 2) issue_k(2);
    if (T > VA) issue_v(VA);
    tile(TrueT{}, 0);
    end_tile(false);

    int t = 1;
    if (!exact_all) {
      // ---- steady state: the loads issued here (K(t+2), V^T(t+VA)) are full tiles of segment 0: running scalar bases
      const char* kptr = reinterpret_cast<const char*>(p.k[0] + (size_t)b * p.sK[0] + h * D) + (size_t)3 * KVT * p.ldk[0] * 2;
      const char* vptr = reinterpret_cast<const char*>(p.vt[0] + (size_t)b * p.sV[0] + (size_t)(h * D) * p.ldv[0]) + (size_t)(1 + VA) * KVT * 2;
      const size_t kstep = (size_t)KVT * p.ldk[0] * 2;
      for (; t + 2 < F0; ++t) {
        // Every wave passed the barrier that ended tile t-1: K(t+1) and V^T(t) are visible, K(t-1) / V^T(t-1) are dead.
        ATR_RESET
        unsigned short* kdst = Ks + ((t + 2) % 3) * KSZ;
        unsigned short* vdst = Vs + ((t + VA) % VST) * VSZ;
#pragma unroll
        for (int j = 0; j < K_PER_WAVE; ++j)
          if (wave + NW * j < K_INST)
            dma16_sv(kptr, koff0[j], lds_addr(kdst + (wave + NW * j) * 512));
#pragma unroll
        for (int j = 0; j < V_PER_WAVE; ++j)
          if (vwave + NW * j < V_INST)
            dma16_sv(vptr, voff0[j], lds_addr(vdst + (vwave + NW * j) * 512));
        kptr += kstep;
        vptr += KVT * 2;
        ATR(0)
        const unsigned acc = tile(FalseT{}, t);
        end_tile(true);
        ATR(7)
        after_tile(acc);
        ATR(8) ATR_COUNT(9)
      }
    }
    // ---- remaining tiles (segment change, tail tiles, end of the key range; every tile of the fallback pass)
    for (; t < T; ++t) {
      if (t + 2 < T) issue_k(t + 2);
      if (t + VA < T) issue_v(t + VA);
      if (exact_all) {
        tile(TrueT{}, t);
        end_tile(false);
      } else {
        const unsigned acc = tile(FalseT{}, t);
        end_tile(false);
        after_tile(acc);
      }
    }
    if (exact_all) break;
    __syncthreads();                                // every wave's redo_flag store is visible
    if (redo_flag == 0) break;
    exact_all = true;                               // workgroup-uniform: all four waves redo the block
  }

  ATR_DUMP
  // ---- normalise and store.  o[g][mt][r]: e = mt*32 + (r&3) + 8*(r>>2) + 4*hi, q = l31 of group g.
  // row e = D of O^T holds the denominator: tile D/32, register 4*((D%32)/8) of the hi = 0 lanes.
  // A lane owns ONE query row in 8-byte pieces: stored directly that is DCH 8-B stores per lane at a 2*ldo-byte lane
  // stride -- store-issue-bound, and every 8-B piece is a partial 32-B sector (PMC: 4.6x the algorithmic write bytes,
  // profiles/r02_rocprof/pmc_traffic_b64.json).  The K / V^T rings are dead here (every wave passed the last tile's
  // barrier), so each wave transposes its 64 x D block through its own LDS slice and writes 16 B per lane with consecutive
  // lanes on consecutive chunks of a row: a wave store instruction covers whole 2*D-byte row segments.
  constexpr int sel = (D & 31) >> 3;
  unsigned short* const ow = smem + wave * (64 * D);       // wave-private [64 queries][D]
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const float lv = o[g][NMT - 1][4 * sel];
    const float l_tot = __shfl(lv, l31, 64);               // broadcast from the hi = 0 lane of this query
    const float inv = 1.0f / l_tot;
    unsigned short* orow = ow + (g * 32 + l31) * D;
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const int e = mt * 32 + 8 * qd + 4 * hi;
        if (e < D) {
          u32x2 pkd = {pack2<DT>(o[g][mt][4 * qd] * inv, o[g][mt][4 * qd + 1] * inv),
                       pack2<DT>(o[g][mt][4 * qd + 2] * inv, o[g][mt][4 * qd + 3] * inv)};
          *reinterpret_cast<u32x2*>(orow + e) = pkd;
        }
      }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the wave's own LDS writes, in order, before its reads
  __builtin_amdgcn_wave_barrier();
  {
    const int q0 = qb * NT + wave * 64;
    unsigned short* const obase = p.out + (size_t)b * p.sO + h * D;
#pragma unroll
    for (int j = 0; j < DCH; ++j) {
      const int c = lane + 64 * j;                          // 16-B chunk of the block, row-major: byte offset 16 c
      const int row = c / DCH, col = c - row * DCH;
      const u32x4 v = *reinterpret_cast<const u32x4*>(ow + c * 8);
      if (q0 + row < p.nq) *reinterpret_cast<u32x4*>(obase + (size_t)(q0 + row) * p.ldo + col * 8) = v;
    }
  }
}

inline int attn_exp() {
#ifdef IDF_ATTN_EXP
  static int v = -1;
  if (v < 0) { const char* e = getenv("IDF_ATTN_EXP"); v = e ? atoi(e) & 0xffff : 0; }
  return v;
#else
  return 0;
#endif
}

template <int DT>
int launch_attn4(const AttnParams& p, int B, hipStream_t s) {
  const int mode = idf_attn2_mode();
  const int nw = mode == 2 ? 8 : 4;
  const int nqb = (p.nq + nw * 64 - 1) / (nw * 64);
  dim3 grid(nqb * p.H * B), block(nw * 64);
#define IDF_ATTN4_CASE(KS, MT) \
  if (p.d == 8 * (2 * KS - 1)) { \
    if (nw == 8) hipLaunchKernelGGL((attn4_kernel<DT, KS, MT, 1, 8>), grid, block, 0, s, p, nqb, 1); \
    else hipLaunchKernelGGL((attn4_kernel<DT, KS, MT, 1, 4>), grid, block, 0, s, p, nqb, (mode == 3 ? 0 : 1) | (attn_exp() << 8)); \
    return idf_launch_status(); }
  IDF_ATTN4_CASE(2, 1)    // d = 24
  IDF_ATTN4_CASE(3, 2)    // d = 40
  IDF_ATTN4_CASE(4, 2)    // d = 56
#undef IDF_ATTN4_CASE
  return IDF_ATTN2_UNSUPPORTED;
}

}  // namespace

#ifdef IDF_ATTN_TRACE
extern "C" int idf_attn_trace_read(unsigned long long* host /* [2][10] */) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(idf_attn_trace_buf), sizeof(idf_attn_trace_buf));
}
#endif

int idf_launch_attn4(const AttnParams& p, int B, int dtype, hipStream_t s) {
  if (p.d != 24 && p.d != 40 && p.d != 56) return IDF_ATTN2_UNSUPPORTED;
  if ((p.n[0] % 8) || (p.n[1] % 8)) return IDF_ATTN2_UNSUPPORTED;
  if ((p.ldk[0] % 8) || (p.ldv[0] % 8) || (p.n[1] > 0 && ((p.ldk[1] % 8) || (p.ldv[1] % 8)))) return IDF_ATTN2_UNSUPPORTED;
  if (!aligned16(p.k[0]) || !aligned16(p.vt[0]) || !aligned16(p.k[1]) || !aligned16(p.vt[1])) return IDF_ATTN2_UNSUPPORTED;
  if ((p.sK[0] % 8) || (p.sV[0] % 8) || (p.sK[1] % 8) || (p.sV[1] % 8)) return IDF_ATTN2_UNSUPPORTED;
  // the LDS-transposed epilogue stores O (and reads Q) as 16-B vectors: rows and batch strides must keep that alignment, else
  // the 32-query kernel (8-B stores, idf_attention's own ldo % 4 contract) takes the launch
  if (!aligned16(p.out) || (p.ldo % 8) || (p.sO % 8) || !aligned16(p.q) || (p.ldq % 8) || (p.sQ % 8)) return IDF_ATTN2_UNSUPPORTED;
  // per-lane DMA offsets are 32-bit: a (batch, head) slice of K / V^T must stay below 4 GB
  if ((long long)KVT * p.ldk[0] * 2 >= (1ll << 31) || (long long)(p.d + 8) * p.ldv[0] * 2 >= (1ll << 31)) return IDF_ATTN2_UNSUPPORTED;
  if (p.n[1] > 0 && ((long long)KVT * p.ldk[1] * 2 >= (1ll << 31) || (long long)(p.d + 8) * p.ldv[1] * 2 >= (1ll << 31)))
    return IDF_ATTN2_UNSUPPORTED;
  if (dtype == IDF_BF16) return launch_attn4<IDF_BF16>(p, B, s);
  if (dtype == IDF_F16) return launch_attn4<IDF_F16>(p, B, s);
  return IDF_ATTN2_UNSUPPORTED;
}
