// attention5.hip -- attention forward, variant 5 (gfx950): 8-wave "ping-pong" form of variant 4 for d = 40.
//
// Measured on variants 2 and 4 (profiles/r02_rocprof/attn_pmc_*.csv): with two INDEPENDENT 4-wave workgroups per CU the two
// waves that share a SIMD drift into the same phase -- both in their MFMA burst (one matrix pipe per SIMD: they serialise)
// or both in their exponentials (one VALU port) -- and MFMA time (896 cycles per 64x64 wave-tile) and VALU time (~900 cycles
// at the one-wave issue rate of v_exp_f32) ADD: ~1900 cycles per wave-tile, MFMA pipe ~45 % busy, whatever is done to either
// instruction count.  Here the two waves of a SIMD belong to ONE 512-thread workgroup (wave w and w + 4) and are held in
// OPPOSITE phases by a workgroup barrier after every phase:
//     M phase (matrix):  O^T += V^T P^T of tile t-1 (16 MFMAs)  then  S^T = K Q^T of tile t (12 MFMAs)      -- no VALU work
//     S phase (scalar):  P = 2^(S^T) (64 v_exp), pack (32 v_cvt_pk), overflow guard (16 v_or3), LDS reads of the next
//                        fragments                                                                            -- no MFMA
// waves 0-3 run M(t) while waves 4-7 run S(t-1), then swap: the matrix pipe and the VALU port of a SIMD are both busy all the
// time and a tile costs max(M, S) per wave instead of M + S.
// Everything else is variant 4 (attention4.hip): 64 queries per wave in two groups of 32, swapped K.Q^T, permuted K rows,
// the reference value m folded into the MFMA through the spare half K-step, max-free softmax with the OR-bit overflow
// guard, all-ones V^T row for the denominator, no tail masking arithmetic, inline-asm LDS-DMA, XCD-aware 1-D grid.
// Differences forced by the phase split:
//   * the guard fires in S(t), BEFORE P(t) enters O: a finite hit only schedules an exact max pass for tile t+1 (classic
//     online-softmax step: O then holds tiles <= t); inf / nan or P >= 2^60 flags the workgroup to redo the block exactly;
//   * K and V^T rings are 4 stages deep; at the END of its S(t) -- behind its VALU stream, while the partner runs nothing
//     but MFMAs -- a wave waits for the loads it issued one tile ago (s_waitcnt vmcnt(0): they have had a whole tile
//     period) and issues the next batch: waves 0-3 K(t+3), waves 4-7 V^T(t+2); the barrier that follows publishes the
//     landed batch at least one phase before its first reader.  ALL LDS fragment reads of an M phase (V^T(t-1), K(t))
//     are issued in the S phase before it: the M phase is 28 MFMAs out of registers.  A wave runs its M phases at
//     s_setprio 1 (see the main loop for why);
//   * 512 queries per workgroup: K / V^T tiles are staged once for 8 waves (half the L2 -> LDS traffic of variant 4).
#include "attn_core.h"

using namespace idfattn;

namespace {

__device__ __attribute__((aligned(128))) unsigned short idf_attn5_zero_page[64];
__device__ __attribute__((aligned(16))) unsigned short idf_attn5_ones_page[2][8] = {
    {0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80},      // bf16 1.0
    {0x3c00, 0x3c00, 0x3c00, 0x3c00, 0x3c00, 0x3c00, 0x3c00, 0x3c00}};     // fp16 1.0

constexpr int KVT = 64;
constexpr int RING = 4;                               // K and V^T ring stages

// LDS-DMA through inline asm (see attention4.hip: keeps the compiler's waitcnt pass from putting vmcnt(0) in front of
// every LDS read); lds = LDS byte address of lane 0's 16-B slot, through M0.
__device__ __forceinline__ unsigned lds_addr5(const void* p) { return (unsigned)(size_t)p; }
__device__ __forceinline__ void dma16_sv5(const void* sbase /* wave-uniform */, unsigned voff, unsigned lds) {
  lds = __builtin_amdgcn_readfirstlane(lds);
#ifdef IDF_ATTN5_BUFFER_DMA
  // A/B: the same transfer as buffer_load ... lds through a 128-bit resource descriptor (base, no stride, no bound)
  const unsigned long long a = (unsigned long long)sbase;
  u32x4 srd;
  srd[0] = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a);
  srd[1] = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xffffu));
  srd[2] = 0xffffffffu;
  srd[3] = 0x00020000u;
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds), "v"(voff), "s"(srd) : "memory");
#else
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds), "v"(voff), "s"(sbase) : "memory");
#endif
}
__device__ __forceinline__ void dma16_v5(const void* addr /* per lane */, unsigned lds) {
  lds = __builtin_amdgcn_readfirstlane(lds);
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds), "v"(addr) : "memory");
}
// phase boundary: nothing may be scheduled across it (an MFMA moved into the other phase would collide with the partner's)
__device__ __forceinline__ void phase_barrier() {
  __builtin_amdgcn_sched_barrier(0);
  __syncthreads();
  __builtin_amdgcn_sched_barrier(0);
}

// Optional per-segment cycle trace (tools/ubench/attn5_trace.hip builds this file with -DIDF_ATTN5_TRACE): s_memtime deltas
// summed per segment over all tiles, written by waves 0 and 4 of block 0.  Costs ~10 % (s_memtime drains lgkmcnt).
#ifdef IDF_ATTN5_TRACE
__device__ unsigned long long idf_attn5_trace_buf[2][16];
#define TR_DECL unsigned long long tr_last = __builtin_readcyclecounter(), tr_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define TR(i) { const unsigned long long tr_now = __builtin_readcyclecounter(); tr_acc[i] += tr_now - tr_last; tr_last = tr_now; }
#define TR_DUMP if (blockIdx.x == 0 && lane == 0 && (wave == 0 || wave == 4)) { for (int i = 0; i < 12; ++i) idf_attn5_trace_buf[wave >> 2][i] = tr_acc[i]; }
#else
#define TR_DECL
#define TR(i)
#define TR_DUMP
#endif

template <int DT> struct RefShift5;          // after an exact pass the largest P of a query is 2^-SHIFT (guard trigger: P >= 2)
// bf16 (8 exponent bits): the first tile's max maps to 2^-40 -- 167 binades of headroom above, 86 below before P flushes
// to zero -- and NO per-tile guard: an overflow (inf in P or O) is caught once, at the end of the block.  fp16 (5 exponent
// bits): P must stay near the top of its range, so the per-tile OR-bit guard (P >= 2 -> exact pass on the next tile) stays.
template <> struct RefShift5<IDF_BF16> { static constexpr float v = 40.0f; };
template <> struct RefShift5<IDF_F16> { static constexpr float v = 1.0f; };

template <int DT, int NKS, int NMT>
__global__ __launch_bounds__(512, 2) void attn5_kernel(const AttnParams p, const int nqb, const int xcd_order, const int prio) {
  constexpr int DCH = 2 * NKS - 1;                 // 16-B chunks per K row
  constexpr int D = 8 * DCH;                       // head dim
  constexpr int KSZ = KVT * D;                     // K stage (elements), linear rows of D*2 bytes (D/8 odd: conflict-free)
  constexpr int VROWS = NMT * 32;
  constexpr int VSZ = VROWS * KVT;                 // V^T stage (elements), 128-B rows, 16-B slot ^= (row >> 1) & 7
  constexpr int K_INST = DCH;                      // LDS-DMA instructions per K tile (64 chunks each)
  constexpr int V_INST = D / 8;                    // per V^T tile (8 rows each); instruction V_INST = the ones-row group
  constexpr int PER_WAVE = (K_INST + 3) / 4;       // instructions of a tile per wave of the issuing group (K_INST == V_INST)
  static_assert(K_INST == V_INST, "K and V^T tiles are both D/8 LDS-DMA instructions");
  static_assert(D < 32 * NMT && (D % 8) == 0 && D + 8 <= VROWS, "needs a spare 8-row group for the softmax denominator");
  __shared__ __attribute__((aligned(128))) unsigned short smem[RING * KSZ + RING * VSZ + 8];
  __shared__ int redo_flag;                        // some wave met an inf / nan / enormous P: redo the block with the exact max
  unsigned short* const Ks = smem;
  unsigned short* const Vs = smem + RING * KSZ;
  unsigned short* const ones_frag = smem + RING * KSZ + RING * VSZ;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int grp = wave >> 2;                         // 0: waves 0-3 (lead), 1: waves 4-7 (one phase behind)
  const int wi = wave & 3;                           // wave inside its group: issues DMA instructions wi, wi + 4

  // ---- XCD-aware block order: hardware block L runs on XCD L % 8; give every XCD a contiguous range of logical blocks
  int L = blockIdx.x;
  {
    const int total = gridDim.x;
    if (xcd_order && (total & 7) == 0) L = (L & 7) * (total >> 3) + (L >> 3);
  }
  const int qb = L % nqb;
  const int h = (L / nqb) % p.H;
  const int b = L / (nqb * p.H);

  // zero the V^T ring once (pad rows of the O^T tile must be finite zeros), then the ones rows and the ones fragment
  for (int i = tid; i < RING * VSZ / 2; i += 512) reinterpret_cast<unsigned*>(Vs)[i] = 0u;
  __syncthreads();
  {
    const unsigned short one = Elem<DT>::from_f32(1.0f);
    for (int i = tid; i < RING * KVT; i += 512) Vs[(i / KVT) * VSZ + D * KVT + (i % KVT)] = one;
    if (tid < 8) ones_frag[tid] = tid == 0 ? one : (unsigned short)0;
    if (tid == 0) redo_flag = 0;
  }

  // ---- Q fragments (B operand) of the two query groups, pre-multiplied by scale*log2(e): lane holds q = l31,
  // e = 16*ks + 8*hi .. +7.  Element D (first element of the hi = 1 half of the last K-step) carries -m.
  u32x4 qf[2][NKS];
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const int qr = min(qb * 512 + wave * 64 + g * 32 + l31, p.nq - 1);
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

  // ---- DMA roles (per group of 4 waves).  K: instruction i moves linear chunks 64 i .. 64 i + 63 of the tile: chunk c -> row
  // c / DCH, column chunk c % DCH.  V^T: instruction i moves rows 8 i .. 8 i + 7: lane -> row 8 i + (lane >> 3), LDS slot
  // lane & 7 (holding the global 8-key chunk slot ^ ((row >> 1) & 7)).  Wave wi of the issuing group issues instructions wi
  // and wi + 4.  Full tiles: uniform base (SGPR) + a per-lane byte offset that only depends on the segment.
  unsigned koff[2][PER_WAVE], voff[2][PER_WAVE];
#pragma unroll
  for (int j = 0; j < PER_WAVE; ++j) {
    const int c = (wi + 4 * j) * 64 + lane;
    const int row = c / DCH, col = (c - row * DCH) * 8;
    koff[0][j] = (unsigned)(row * p.ldk[0] + col) * 2u;
    koff[1][j] = (unsigned)(row * p.ldk[1] + col) * 2u;
    const int vrow = (wi + 4 * j) * 8 + (lane >> 3);
    const int vch = (lane & 7) ^ ((vrow >> 1) & 7);
    voff[0][j] = (unsigned)(vrow * p.ldv[0] + vch * 8) * 2u;
    voff[1][j] = (unsigned)(vrow * p.ldv[1] + vch * 8) * 2u;
  }
  const char* const kbase0 = reinterpret_cast<const char*>(p.k[0] + (size_t)b * p.sK[0] + h * D);
  const char* const kbase1 = reinterpret_cast<const char*>(p.k[1] + (size_t)b * p.sK[1] + h * D);
  const char* const vbase0 = reinterpret_cast<const char*>(p.vt[0] + (size_t)b * p.sV[0] + (size_t)(h * D) * p.ldv[0]);
  const char* const vbase1 = reinterpret_cast<const char*>(p.vt[1] + (size_t)b * p.sV[1] + (size_t)(h * D) * p.ldv[1]);

  // issue this wave's share of K tile t; returns true when the counted end-of-phase wait applies (a full tile)
  // jsel = 0: instruction wi (the S-phase share), 1: instruction wi + 4 (the M-phase share), -1: both (prologue)
  auto issue_k = [&](int t, int jsel) -> bool {
    const int seg = (t < T0) ? 0 : 1;
    const int kv0 = (seg ? (t - T0) : t) * KVT;
    const int n = p.n[seg];
    const int ldk = p.ldk[seg];
    const char* kb = seg ? kbase1 : kbase0;
    unsigned short* dst = Ks + (t % RING) * KSZ;
    if (kv0 + KVT <= n) {
      const char* base = kb + (size_t)kv0 * ldk * 2;
#pragma unroll
      for (int j = 0; j < PER_WAVE; ++j)
        if (wi + 4 * j < K_INST && (jsel < 0 || jsel == j))
          dma16_sv5(base, seg ? koff[1][j] : koff[0][j], lds_addr5(dst + (wi + 4 * j) * 512));
      return true;
    }
#pragma unroll
    for (int j = 0; j < PER_WAVE; ++j)               // tail tile: rows beyond n are clamped to the last valid key
      if (wi + 4 * j < K_INST && (jsel < 0 || jsel == j)) {
        const int c = (wi + 4 * j) * 64 + lane;
        const int row = c / DCH, col = (c - row * DCH) * 8;
        const int kr = min(kv0 + row, n - 1);
        dma16_v5(kb + ((size_t)kr * ldk + col) * 2, lds_addr5(dst + (wi + 4 * j) * 512));
      }
    return false;
  };
  // the ones-row group (rows D .. D+7 of the V^T image): row D = ones in the valid columns, zeros elsewhere
  auto issue_ones = [&](int stage, int nvalid) {
    if (wi == 2) {
      const int row = D + (lane >> 3);
      const int chunk = (lane & 7) ^ ((row >> 1) & 7);
      const bool one = (row == D) && (chunk * 8 < nvalid);
      const unsigned short* src = one ? idf_attn5_ones_page[DT == IDF_BF16 ? 0 : 1] : idf_attn5_zero_page + (lane & 7) * 8;
      dma16_v5(src, lds_addr5(Vs + stage * VSZ + V_INST * 512));
    }
  };
  auto issue_v = [&](int t, int jsel) -> bool {
    const int seg = (t < T0) ? 0 : 1;
    const int kv0 = (seg ? (t - T0) : t) * KVT;
    const int n = p.n[seg];
    const char* vb = seg ? vbase1 : vbase0;
    unsigned short* dst = Vs + (t % RING) * VSZ;
    const char* base = vb + (size_t)kv0 * 2;
    const bool tail = (kv0 + KVT > n);
    bool prev_tail = false;                          // the stage last held tile t - RING: was its ones row restricted?
    if (t >= RING) {
      const int t2 = t - RING;
      const int s2 = (t2 < T0) ? 0 : 1;
      prev_tail = ((s2 ? (t2 - T0) : t2) + 1) * KVT > p.n[s2];
    }
    if (!tail) {
#pragma unroll
      for (int j = 0; j < PER_WAVE; ++j)
        if (wi + 4 * j < V_INST && (jsel < 0 || jsel == j))
          dma16_sv5(base, seg ? voff[1][j] : voff[0][j], lds_addr5(dst + (wi + 4 * j) * 512));
    } else {                                         // tail tile: 8-key chunks beyond n (n % 8 == 0) come from the zero page
#pragma unroll
      for (int j = 0; j < PER_WAVE; ++j)
        if (wi + 4 * j < V_INST && (jsel < 0 || jsel == j)) {
          const int vrow = (wi + 4 * j) * 8 + (lane >> 3);
          const int vch = (lane & 7) ^ ((vrow >> 1) & 7);
          const bool valid = (kv0 + vch * 8) < n;
          const char* src = valid ? base + (seg ? voff[1][j] : voff[0][j])
                                  : reinterpret_cast<const char*>(idf_attn5_zero_page + (lane & 7) * 8);
          dma16_v5(src, lds_addr5(dst + (wi + 4 * j) * 512));
        }
    }
    if ((tail || prev_tail) && jsel <= 0) issue_ones(t % RING, tail ? n - kv0 : KVT);
    return !tail && !prev_tail;
  };
  f32x16 o[2][NMT];
  float m_run[2];                                   // the reference value m of the lane's query, 16-bit representable
  const int v_sw = (l31 >> 1) & 7;                  // V^T fragment rows are mt*32 + l31
  // K fragment row permutation (see attention2.hip): MFMA row i of a 32-key half carries key (i with bits 2 and 3
  // swapped), so the 8 S^T registers of a lane-half per 16-key step are 8 CONSECUTIVE keys = the k order of P.V.
  const int kperm = (l31 & 0x13) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);
  const int kfoff = kperm * D + hi * 8;             // element offset of the lane's K fragment inside a 32-key half
  const int vfoff = l31 * KVT;                      // V^T fragment row offset

  f32x16 s[2][2];                                    // [query group][kv half]
  u32x4 pk[2][4];                                    // packed P: [group][16-key step]
  u32x4 kf[2][NKS];                                  // K fragments of the next K.Q^T, read in the S phase before it
  u32x4 vf[4][NMT];                                  // V^T fragments of the next P.V, read in the S phase before it
  auto load_kf = [&](int stage) {
    const unsigned short* Kc = Ks + stage * KSZ;
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      const unsigned short* base = Kc + st * 32 * D + kfoff;
#pragma unroll
      for (int ks = 0; ks < NKS - 1; ++ks) kf[st][ks] = *reinterpret_cast<const u32x4*>(base + ks * 16);
      // last K-step: hi = 0 lanes read elements 16*(NKS-1) .. +7 of the row, hi = 1 lanes the constant {1, 0, .., 0}
      const unsigned short* last = hi ? ones_frag : base + (NKS - 1) * 16;
      kf[st][NKS - 1] = *reinterpret_cast<const u32x4*>(last);
    }
  };
  auto load_vf = [&](int stage) {
    const unsigned short* Vc = Vs + stage * VSZ + vfoff;
#pragma unroll
    for (int step = 0; step < 4; ++step)
#pragma unroll
      for (int mt = 0; mt < NMT; ++mt)
        vf[step][mt] = *reinterpret_cast<const u32x4*>(Vc + mt * 32 * KVT + (((step * 2 + hi) ^ v_sw) * 8));
  };
  auto qk = [&]() {
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
      for (int st = 0; st < 2; ++st)
#pragma unroll
        for (int g = 0; g < 2; ++g) s[g][st] = Elem<DT>::mfma32(kf[st][ks], qf[g][ks], ks == 0 ? zero : s[g][st]);
  };
  // O^T += V^T P^T for both query groups (every V^T fragment feeds two MFMAs), straight from registers: hipcc sinks LDS
  // reads placed in the M phase next to their MFMAs (one exposed LDS latency per fragment), so they all live in the S phase.
  auto pv = [&]() {
#pragma unroll
    for (int step = 0; step < 4; ++step)
#pragma unroll
      for (int mt = 0; mt < NMT; ++mt) {
        o[0][mt] = Elem<DT>::mfma32(vf[step][mt], pk[0][step], o[0][mt]);
        o[1][mt] = Elem<DT>::mfma32(vf[step][mt], pk[1][step], o[1][mt]);
      }
  };
  auto half_max = [&](float mx) -> float {           // max over the two lane halves that share a query
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
    return fmaxf(mx, __uint_as_float(hi ? sw[0] : sw[1]));
  };
  // exact pass over the scores of group g (already relative to the current m): the tile's max maps to 2^-SHIFT (first
  // tile) or m is raised when the max exceeds that; O (tiles before this one; its row D is the denominator) is rescaled,
  // the -m element of Q rewritten.  m stays 16-bit representable.
  auto rebase_scores = [&](const int g, const bool first) {
    float m0 = fmaxf(s[g][0][0], s[g][0][1]), m1 = fmaxf(s[g][1][0], s[g][1][1]);
#pragma unroll
    for (int r = 2; r < 16; r += 2) {
      m0 = fmaxf(fmaxf(m0, s[g][0][r]), s[g][0][r + 1]);
      m1 = fmaxf(fmaxf(m1, s[g][1][r]), s[g][1][r + 1]);
    }
    const float want = half_max(fmaxf(m0, m1)) + RefShift5<DT>::v;
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
#pragma unroll
    for (int st = 0; st < 2; ++st)
#pragma unroll
      for (int r = 0; r < 16; ++r) s[g][st][r] -= d_eff;
  };
  auto exp_pack = [&]() -> unsigned {                // P = 2^s, packed; returns the OR of the packed words
    unsigned acc = 0u;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
#pragma unroll
      for (int st = 0; st < 2; ++st)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[g][st][r] = __builtin_amdgcn_exp2f(s[g][st][r]);
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
    }
    return acc;
  };
  constexpr unsigned EXP_MASK = DT == IDF_BF16 ? 0x7f80u : 0x7c00u;     // all-ones exponent of a 16-bit half: inf / nan

  constexpr bool GUARD = DT == IDF_F16;            // per-tile overflow guard (see RefShift5)
  // S phase of tile t.  `exact`: max pass before the exponentials (tile 0, the tile after a guard hit, the fallback pass).
  // Returns whether the NEXT tile must be exact (guard hit: some P of this wave reached 2).
  auto s_phase = [&](const int t, const bool exact) -> bool {
    if (exact) {
      // keys beyond n in a tail tile are clamped duplicates of a valid key: they cannot raise the max
      rebase_scores(0, t == 0);
      rebase_scores(1, t == 0);
    }
    const unsigned acc = exp_pack();
    // pin: P is packed HERE (LLVM otherwise sinks the exponentials towards their consumer, the P.V MFMAs behind the next
    // barrier -- i.e. behind the loads below -- and the phase order this kernel is built on is gone)
#pragma unroll
    for (int g = 0; g < 2; ++g) asm volatile("" ::"v"(pk[g][0]), "v"(pk[g][1]), "v"(pk[g][2]), "v"(pk[g][3]));
    __builtin_amdgcn_sched_barrier(0);               // the reads below stay behind the exponentials (the scores are dead by then)
    load_vf(t % RING);                               // V^T(t), for the P.V of this tile in the next M phase
    load_kf((t + 1) % RING);                         // K(t+1), for the K.Q^T of that M phase (a stale stage after the last tile)
    bool hit = false;
    if (GUARD && __builtin_amdgcn_ballot_w64((acc & 0x40004000u) != 0u) != 0) {
      // rare: bit 14 of a half (fp16 exponent field >= 16), i.e. some P >= 2.  Finite: P(t) is valid and enters O as it is;
      // the next tile takes the exact pass (which raises m if the level stays high).  Inf / nan: flag the workgroup to
      // redo the block with the exact max on every tile.
      hit = true;
      bool bad = false;
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int step = 0; step < 4; ++step)
#pragma unroll
          for (int w = 0; w < 4; ++w) {
            const unsigned v = pk[g][step][w];
            bad |= ((v & EXP_MASK) == EXP_MASK) | (((v >> 16) & EXP_MASK) == EXP_MASK);
          }
      if (__builtin_amdgcn_ballot_w64(bad) != 0 && lane == 0) redo_flag = 1;
    }
    return hit;
  };

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
      for (int i = tid; i < RING * KVT; i += 512) Vs[(i / KVT) * VSZ + D * KVT + (i % KVT)] = one;
    }
    __syncthreads();                                // zero fill, ones rows, ones fragment (or the abandoned pass) complete
    // prologue loads: K(0..2) by group 0, V^T(0..1) by group 1
    if (grp == 0) {
      issue_k(0, -1);
      if (T > 1) issue_k(1, -1);
      if (T > 2) issue_k(2, -1);
    } else {
      issue_v(0, -1);
      if (T > 1) issue_v(1, -1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    load_kf(0);

    // Priorities (prio == 1): a wave holds s_setprio 1 through its M phases and 0 through its S phases (switched just
    // before the barrier that ends a phase).  Without it the YOUNGER wave of a SIMD (waves 4-7) loses every VALU / SMEM /
    // VMEM arbitration against its older partner: ~450 cycles stalled at the start of each of its M phases and its LDS-DMA
    // instructions wait for a gap in the partner's exponentials (profiles/r02_attn5_trace*.log).
    if (prio == 1) __builtin_amdgcn_s_setprio(1);    // prio == 2 (A/B): the other way round, S phases at priority 1
    if (grp == 1) phase_barrier();                   // waves 4-7 run one phase behind waves 0-3
    bool exact_next = true;                          // tile 0 fixes the reference value of every query
    TR_DECL
    for (int t = 0; t < T; ++t) {
      // ---- M phase: P.V of tile t-1, K.Q^T of tile t -- 28 MFMAs out of registers.  No VALU, no LDS reads.  Behind the
      // MFMAs: the PARTNER group's fifth LDS-DMA instruction of this step (see below).
      TR(0)
      if (t > 0) pv();
      TR(2)
      qk();
#pragma unroll
      for (int g = 0; g < 2; ++g) asm volatile("" ::"v"(s[g][0]), "v"(s[g][1]));      // pin: the MFMAs are issued in this phase
      if (K_INST > 4) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (grp == 1) { if (t + 3 < T) issue_k(t + 3, 1); }          // step 2t+1: group 0 is in S(t) issuing K(t+3)[0..3]
        else if (t > 0) { if (t + 1 < T) issue_v(t + 1, 1); }        // step 2t: group 1 is in S(t-1) issuing V^T(t+1)[0..3]
      }
      if (prio == 1) __builtin_amdgcn_s_setprio(0);
      if (prio == 2) __builtin_amdgcn_s_setprio(1);
      TR(3)
      phase_barrier();
      TR(5)
      // ---- S phase: exponentials, packing, the LDS reads of the next M phase (+ guard); then, behind the VALU stream and
      // while the partner wave runs nothing but MFMAs, ONE LDS-DMA instruction per wave: an instruction costs its wave
      // 200 - 280 cycles of issue (profiles/r02_attn5_trace8.log), so a tile's five K (V^T) instructions go out as four from
      // the four waves of the group in its S phase plus one from a wave of the other group behind its MFMAs in the same
      // step.  Before issuing, a wave waits for what it issued earlier (s_waitcnt vmcnt(0): at least a phase old); the
      // barrier that follows publishes it, at least one phase before its first reader.
      exact_next = s_phase(t, exact_next || exact_all);
      TR(6)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      TR(4)
      if (grp == 0) { if (t + 3 < T) issue_k(t + 3, 0); }
      else { if (t + 2 < T) issue_v(t + 2, 0); }
      if (prio == 1) __builtin_amdgcn_s_setprio(1);
      if (prio == 2) __builtin_amdgcn_s_setprio(0);
      TR(1)
      phase_barrier();
      TR(7)
    }
    TR_DUMP
    pv();                                            // the last tile's P.V (the other group is in its last S phase / done)
    if (grp == 0) phase_barrier();
    if (exact_all) break;
    {
      // the common path never looked at P (bf16) / only looked for P >= 2 (fp16): an overflow anywhere shows here, as a
      // non-finite accumulator or denominator
      bool bad = false;
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
          for (int r = 0; r < 16; ++r) bad |= !(__builtin_fabsf(o[g][mt][r]) < INFINITY);
      if (__builtin_amdgcn_ballot_w64(bad) != 0 && lane == 0) redo_flag = 1;
    }
    __syncthreads();                                // every wave's redo_flag store is visible
    if (redo_flag == 0) break;
    exact_all = true;                               // workgroup-uniform: all eight waves redo the block
  }

  // ---- normalise and store.  o[g][mt][r]: e = mt*32 + (r&3) + 8*(r>>2) + 4*hi, q = l31 of group g.
  // row e = D of O^T holds the denominator: tile D/32, register 4*((D%32)/8) of the hi = 0 lanes
  constexpr int sel = (D & 31) >> 3;
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const float lv = o[g][NMT - 1][4 * sel];
    const float l_tot = __shfl(lv, l31, 64);               // broadcast from the hi = 0 lane of this query
    const float inv = 1.0f / l_tot;
    const int qrow = qb * 512 + wave * 64 + g * 32 + l31;
    if (qrow < p.nq) {
      unsigned short* op = p.out + (size_t)b * p.sO + (size_t)qrow * p.ldo + h * D;
#pragma unroll
      for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          const int e = mt * 32 + 8 * qd + 4 * hi;
          if (e < D) {
            u32x2 pkd = {pack2<DT>(o[g][mt][4 * qd] * inv, o[g][mt][4 * qd + 1] * inv),
                         pack2<DT>(o[g][mt][4 * qd + 2] * inv, o[g][mt][4 * qd + 3] * inv)};
            *reinterpret_cast<u32x2*>(op + e) = pkd;
          }
        }
    }
  }
}

template <int DT>
int launch_attn5(const AttnParams& p, int B, hipStream_t s) {
  const int nqb = (p.nq + 511) / 512;
  dim3 grid(nqb * p.H * B), block(512);
  // modes 9 .. 14: (mode - 9) & 1 = plain block order instead of the XCD-aware one; (mode - 9) >> 1 = priorities: 0 none,
  // 1 s_setprio 1 through the M phases, 2 through the S phases
  const int mode = idf_attn2_mode() - 9;
  const int xcd = (mode & 1) ? 0 : 1, prio = (mode >> 1) & 3;
#define IDF_ATTN5_CASE(KS, MT) \
  if (p.d == 8 * (2 * KS - 1)) { \
    hipLaunchKernelGGL((attn5_kernel<DT, KS, MT>), grid, block, 0, s, p, nqb, xcd, prio); \
    return idf_launch_status(); }
  IDF_ATTN5_CASE(2, 1)    // d = 24
  IDF_ATTN5_CASE(3, 2)    // d = 40
  IDF_ATTN5_CASE(4, 2)    // d = 56
#undef IDF_ATTN5_CASE
  return IDF_ATTN2_UNSUPPORTED;
}

}  // namespace

int idf_launch_attn5(const AttnParams& p, int B, int dtype, hipStream_t s) {
  if (p.d != 24 && p.d != 40 && p.d != 56) return IDF_ATTN2_UNSUPPORTED;
  if ((p.n[0] % 8) || (p.n[1] % 8)) return IDF_ATTN2_UNSUPPORTED;
  if ((p.ldk[0] % 8) || (p.ldv[0] % 8) || (p.n[1] > 0 && ((p.ldk[1] % 8) || (p.ldv[1] % 8)))) return IDF_ATTN2_UNSUPPORTED;
  if (!aligned16(p.k[0]) || !aligned16(p.vt[0]) || !aligned16(p.k[1]) || !aligned16(p.vt[1])) return IDF_ATTN2_UNSUPPORTED;
  if ((p.sK[0] % 8) || (p.sV[0] % 8) || (p.sK[1] % 8) || (p.sV[1] % 8)) return IDF_ATTN2_UNSUPPORTED;
  // per-lane DMA offsets are 32-bit: a (batch, head) slice of K / V^T must stay below 4 GB
  if ((long long)KVT * p.ldk[0] * 2 >= (1ll << 31) || (long long)(p.d + 8) * p.ldv[0] * 2 >= (1ll << 31)) return IDF_ATTN2_UNSUPPORTED;
  if (p.n[1] > 0 && ((long long)KVT * p.ldk[1] * 2 >= (1ll << 31) || (long long)(p.d + 8) * p.ldv[1] * 2 >= (1ll << 31)))
    return IDF_ATTN2_UNSUPPORTED;
  if (dtype == IDF_BF16) return launch_attn5<IDF_BF16>(p, B, s);
  if (dtype == IDF_F16) return launch_attn5<IDF_F16>(p, B, s);
  return IDF_ATTN2_UNSUPPORTED;
}
