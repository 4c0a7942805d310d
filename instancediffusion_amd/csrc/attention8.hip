// attention8.hip -- attention forward, variant 8 (gfx950): the 32x32 / 16x16 / 8x8-latent self and gated self-attention
// (head dim d = 80 at C = 640, d = 160 at C = 1280; reference attention.py:120-157,257-282).
//
// Round 5.  These launches ran on the round-1 kernel of attention.hip (32 queries per wave, tiles staged global -> registers
// -> ds_write, online softmax with a rescale of O per tile, one workgroup per CU at d = 160 because of its 89 KB of LDS):
// 537-548 TF at d = 80, 235-280 TF at d = 160, and 5.5x the algorithmic bytes fetched at d = 80 because the 8 query blocks
// of a (batch, head) ran on 8 different XCD L2s (profiles/r04_rocprof/pmc_traffic_b128.json).  This kernel carries the
// d = 40 scheme of attention4.hip over to head dims that are multiples of 16:
//   * K and V^T tiles arrive by LDS-DMA (global_load_lds_dwordx4 issued through inline assembly, ordered by our own
//     s_waitcnt vmcnt(0) + s_barrier at the end of a tile) into rings; a tile's loads have a whole tile of compute to land.
//     K rows are linear; d / 8 is EVEN here, so a K row is padded to an odd number of 16-B slots (11 at d = 80, 21 at d = 160;
//     the pad slot is fetched like slot 0 and never read) -- with an odd slot count the 16 lanes of a ds_read_b128 group,
//     which read 16 rows that are distinct mod 16, hit 16 disjoint 4-bank windows.  V^T rows are 128 B with the 16-B slot
//     ^= (row >> 1) & 7, applied on the global source address and on the fragment read.
//   * 32 queries per wave (ONE query group: two groups, as at d = 40, would need 272+ registers at d = 80), swapped K.Q^T so a
//     lane owns a query column, K fragment rows permuted so that the packed P feeds P.V without lane exchanges.
//   * Running max with a DEFERRED rescale: Q is pre-multiplied by scale * log2(e); a tile rescales O (and raises m) only when
//     some query's tile maximum exceeds its reference m by more than 2^DEFER -- P then stays <= 2^DEFER, which costs the
//     16-bit P nothing (relative rounding) -- so the common tile is {16 v_max3, 32 v_sub, 32 v_exp, 16 v_cvt_pk} of VALU per
//     lane and no pass over O.  There is no spare K-step to carry -m through the MFMA here (d = 80 fills its five K-steps).
//   * Softmax denominator: an all-ones V^T row where the O^T tile has spare rows (d = 80: rows 80..95), fp32 adds at d = 160.
//   * Tails: K rows beyond n are clamped duplicates of the last valid key, whole 8-key V^T chunks beyond n come from a page of
//     zeros, and a tile with a tail sets its invalid scores to -inf (wave-uniform branch) -- requires n % 8 == 0.
//   * 1-D grid, XCD-aware: the query blocks of one (batch, head) run on ONE XCD, so its K / V^T stay in that L2.
//   * Epilogue: the wave transposes its 32 x d block through its own slice of the (dead) rings and stores 16 B per lane.
// KPRE (d = 80): the K fragments of tile t+1 are read under the P.V MFMAs of tile t (K fetched two tiles ahead into a 3-stage
// ring); d = 160 has no registers for that and reads its K fragments inside the K.Q^T loop from a 2-stage ring.
// Numerics contract as the other variants: fp32 scores / accumulators, P rounded to the 16-bit type before P.V.
// Requirements (else IDF_ATTN2_UNSUPPORTED and the caller falls back): d in {80, 160}, n0 % 8 == n1 % 8 == 0, no mask.
#include "attn_core.h"
#include <cstdlib>

using namespace idfattn;

namespace {

__device__ __attribute__((aligned(128))) unsigned short idf_attn8_zero_page[64];

constexpr int KVT = 64;
constexpr float DEFER = 6.0f;                        // log2 units a tile maximum may exceed the reference without a rescale

__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(size_t)p; }
// lds = LDS byte address of lane 0's 16-B slot (lane i lands at lds + 16 i); it goes through M0.  M0 cannot be declared in the
// clobber list (hipcc: "reserved register"); it is ours here -- nothing else in this kernel uses it (no movrel / GWS / sendmsg,
// no LDS-DMA builtin) and every asm statement that reads it writes it first.
__device__ __forceinline__ const void* uniform_ptr(const void* p) {      // provably wave-uniform (an SGPR pair for the asm operand)
  const unsigned long long a = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  return (const void*)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ void dma16_sv(const void* sbase /* wave-uniform */, unsigned voff, unsigned lds) {
  lds = __builtin_amdgcn_readfirstlane(lds);
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds), "v"(voff), "s"(sbase) : "memory");
}
__device__ __forceinline__ void dma16_v(const void* addr /* per lane */, unsigned lds) {
  lds = __builtin_amdgcn_readfirstlane(lds);
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds), "v"(addr) : "memory");
}
__device__ __forceinline__ float max3f(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

template <int DT, int D, int NW, int MODE>
__global__ __launch_bounds__(NW * 64, 2) void attn8_kernel(const AttnParams p, const int nqb, const int xcd_order) {
  static_assert(D % 16 == 0, "head dim must fill whole K-steps");
  constexpr int NKS = D / 16;                      // K-steps of K.Q^T
  constexpr bool MFMASUM = (D % 32) != 0;          // spare rows in the O^T tile: row D of V^T = ones -> denominator from the MFMAs
  constexpr int NMT = (D + 31) / 32;               // 32-row tiles of O^T
  constexpr int DCH = D / 8;                       // 16-B chunks of data per K row
  constexpr int KCH = DCH | 1;                     // slots per K row in LDS (odd)
  constexpr int KSZ = KVT * KCH * 8;               // K stage (elements)
  constexpr int VROWS = NMT * 32;
  constexpr int VSZ = VROWS * KVT;                 // V^T stage (elements)
  // MODE 0: K fragments of tile t+1 read into registers under the P.V MFMAs of tile t (3-stage K ring); 1: no look-ahead, K fragments
  // read inside the K.Q^T loop (2-stage ring); 2: software-pipelined -- K.Q^T of tile t+1 is issued BEFORE the softmax of tile t
  // (two score blocks live, 3-stage ring), so every wave carries an MFMA stream that does not depend on its own exponentials
  constexpr bool KPRE = MODE == 0, PIPE = MODE == 2;
  constexpr int KST = MODE == 1 ? 2 : 3, KA = KST - 1;  // K ring stages; K is fetched KA tiles ahead, V^T one tile ahead
  constexpr int K_INST = KCH, V_INST = DCH;        // LDS-DMA instructions per K / V^T tile (64 slots = 1 KiB each)
  constexpr int N_INST = K_INST + V_INST;
  constexpr int PER_WAVE = (N_INST + NW - 1) / NW;
  constexpr int NT = NW * 64;
  constexpr int RING = KST * KSZ + 2 * VSZ, OSTAGE = NW * 32 * D;
  static_assert(OSTAGE <= RING, "the epilogue staging block must fit the dead rings");
  extern __shared__ __attribute__((aligned(128))) unsigned short smem[];
  unsigned short* const Ks = smem;
  unsigned short* const Vs = smem + KST * KSZ;

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

  // zero the pad rows of the V^T ring once (rows D.. of the O^T tile must be finite zeros), then the ones row
  if constexpr (VROWS > D) {
    for (int i = tid; i < 2 * (VROWS - D) * KVT / 2; i += NT) {
      const int st = i / ((VROWS - D) * KVT / 2), r = i - st * ((VROWS - D) * KVT / 2);
      reinterpret_cast<unsigned*>(Vs + st * VSZ + D * KVT)[r] = 0u;
    }
    __syncthreads();
    const unsigned short one = Elem<DT>::from_f32(1.0f);
    for (int i = tid; i < 2 * KVT; i += NT) Vs[(i / KVT) * VSZ + D * KVT + (i % KVT)] = one;      // swizzle-invariant: a whole row
  }

  // ---- Q fragments (B operand): lane holds q = l31, e = 16 ks + 8 hi .. +7, pre-multiplied by scale*log2(e) (as attention4.hip and
  // torch's math SDPA do).  -DIDF_ATTN8_FMA_SCALE keeps Q raw and scales in the exponent's fma instead (p = 2^(s c - m)): one
  // rounding of Q fewer -- rel-RMS 2.0e-3 instead of 3.4e-3 against the fp32 reference on wide score distributions, equal on
  // narrow ones -- but the 32 v_fma_f32 (VOP3, SGPR or VGPR scale alike) per tile in place of 32 v_sub_f32 cost 5-14 % of the
  // kernel on every box measured (profiles/r05_attn8_third.log, r05_attn8_fourth.log), so the default pre-multiplies.
  u32x4 qf[NKS];
  {
    const int qr = min(qb * (NW * 32) + wave * 32 + l31, p.nq - 1);
    const unsigned short* qp = p.q + (size_t)b * p.sQ + (size_t)qr * p.ldq + h * D;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      qf[ks] = *reinterpret_cast<const u32x4*>(qp + ks * 16 + hi * 8);
#ifndef IDF_ATTN8_FMA_SCALE
      float f[8];
      unpack8<DT>(qf[ks], f);
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] *= p.scale_log2;
      qf[ks] = pack8<DT>(f);
#endif
    }
  }
#ifndef IDF_ATTN8_FMA_SCALE
  const float c = 1.0f;
#else
  const float c = p.scale_log2;
#endif

  const int T0 = (p.n[0] + KVT - 1) / KVT;
  const int T1 = (p.n[1] + KVT - 1) / KVT;
  const int T = T0 + T1;
  const int F0 = p.n[0] / KVT;                       // full tiles of segment 0

  // ---- DMA roles.  The tile's N_INST instructions (K: instruction i moves LDS slots 64 i .. 64 i + 63 of the stage, slot s ->
  // row s / KCH, chunk s % KCH [the pad chunk repeats chunk 0]; V^T: instruction i moves rows 8 i .. 8 i + 7, lane -> row
  // 8 i + (lane >> 3), slot lane & 7 holding the global 8-key chunk slot ^ ((row >> 1) & 7)) are dealt round-robin to the waves.
  // Only the per-lane offsets of FULL tiles of segment 0 are kept in registers; everything else (first tiles, segment change,
  // tails) is recomputed from an opaque copy of the lane id so that nothing cold is hoisted across the hot loop.
  unsigned off0[PER_WAVE];
#pragma unroll
  for (int j = 0; j < PER_WAVE; ++j) {
    const int g = wave + NW * j;
    unsigned o = 0u;
    if (g < K_INST) {
      const int s = g * 64 + lane;
      const int row = s / KCH;
      int col = s - row * KCH;
      col = col == DCH ? 0 : col;
      o = (unsigned)(row * p.ldk[0] + col * 8) * 2u;
    } else if (g < N_INST) {
      const int row = (g - K_INST) * 8 + (lane >> 3);
      o = (unsigned)(row * p.ldv[0] + ((lane & 7) ^ ((row >> 1) & 7)) * 8) * 2u;
    }
    off0[j] = o;
  }
  auto cold_lane = [&]() { int l = lane; asm volatile("" : "+v"(l)); return l; };
  // generic (cold) issue of tile t's K (do_k) and / or V^T (do_v) loads
  auto issue_cold = [&](const int t, const bool do_k, const bool do_v) {
    const int ln = cold_lane();
    const int seg = (t < T0) ? 0 : 1;
    const int kv0 = (seg ? (t - T0) : t) * KVT;
    const int n = p.n[seg];
    const int ldk = p.ldk[seg], ldv = p.ldv[seg];
    const char* kb = reinterpret_cast<const char*>(p.k[seg] + (size_t)b * p.sK[seg] + h * D);
    const char* vb = reinterpret_cast<const char*>(p.vt[seg] + (size_t)b * p.sV[seg] + (size_t)(h * D) * ldv) + (size_t)kv0 * 2;
    unsigned short* kdst = Ks + (t % KST) * KSZ;
    unsigned short* vdst = Vs + (t & 1) * VSZ;
#pragma unroll
    for (int j = 0; j < PER_WAVE; ++j) {
      const int g = wave + NW * j;
      if (g < K_INST) {
        if (do_k) {
          const int s = g * 64 + ln;
          const int row = s / KCH;
          int col = s - row * KCH;
          col = col == DCH ? 0 : col;
          const int kr = min(kv0 + row, n - 1);        // tail tile: rows beyond n are clamped to the last valid key
          dma16_v(kb + ((size_t)kr * ldk + col * 8) * 2, lds_addr(kdst + g * 512));
        }
      } else if (g < N_INST) {
        if (do_v) {
          const int row = (g - K_INST) * 8 + (ln >> 3);
          const int chunk = (ln & 7) ^ ((row >> 1) & 7);
          const bool valid = (kv0 + chunk * 8) < n;    // n % 8 == 0: a chunk is valid or invalid as a whole
          const char* src = valid ? vb + ((size_t)row * ldv + chunk * 8) * 2
                                  : reinterpret_cast<const char*>(idf_attn8_zero_page + (ln & 7) * 8);
          dma16_v(src, lds_addr(vdst + (g - K_INST) * 512));
        }
      }
    }
  };
  const char* const kb0 = reinterpret_cast<const char*>(p.k[0] + (size_t)b * p.sK[0] + h * D);
  const char* const vb0 = reinterpret_cast<const char*>(p.vt[0] + (size_t)b * p.sV[0] + (size_t)(h * D) * p.ldv[0]);
  const size_t kstep = (size_t)KVT * p.ldk[0] * 2;
  // hot issue: K(tk) and V^T(tv) are full tiles of segment 0 -> scalar tile base + the resident per-lane offsets
  auto issue_hot = [&](const int tk, const int tv) {
    const void* kptr = uniform_ptr(kb0 + (size_t)tk * kstep);
    const void* vptr = uniform_ptr(vb0 + (size_t)tv * (KVT * 2));
    unsigned short* kdst = Ks + (tk % KST) * KSZ;
    unsigned short* vdst = Vs + (tv & 1) * VSZ;
#pragma unroll
    for (int j = 0; j < PER_WAVE; ++j) {
      const int g = wave + NW * j;
      if (g < K_INST) dma16_sv(kptr, off0[j], lds_addr(kdst + g * 512));
      else if (g < N_INST) dma16_sv(vptr, off0[j], lds_addr(vdst + (g - K_INST) * 512));
    }
  };

  f32x16 o[NMT];
#pragma unroll
  for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[mt][r] = 0.0f;
  float m_run = -INFINITY, l_run = 0.0f;
  const int v_sw = (l31 >> 1) & 7;                  // V^T fragment rows are mt*32 + l31
  // K fragment row permutation: MFMA row i of a 32-key half carries key (i with bits 2 and 3 swapped), so the 8 S^T registers
  // of a lane-half per 16-key step are 8 CONSECUTIVE keys = the k order of the P.V operands:
  //   s[st][r] of lane (q, hi)  <->  key st*32 + 16 (r >> 3) + 8 hi + (r & 7)
  const int kperm = (l31 & 0x13) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);
  const int kfoff = kperm * (KCH * 8) + hi * 8;     // element offset of the lane's K fragment inside a 32-key half
  const int vfoff = l31 * KVT;

  u32x4 kf[KPRE ? 2 : 1][KPRE ? NKS : 1];           // K fragments of the CURRENT tile (KPRE: read one tile ahead)
  auto load_kf = [&](const int stage) {
    if constexpr (KPRE) {
      const unsigned short* Kc = Ks + stage * KSZ + kfoff;
#pragma unroll
      for (int st = 0; st < 2; ++st)
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) kf[st][ks] = *reinterpret_cast<const u32x4*>(Kc + st * 32 * (KCH * 8) + ks * 16);
    }
  };

  u32x4 pk[4];
  auto qk = [&](f32x16 (&s)[2], const int stage) {
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if constexpr (KPRE) {
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
        for (int st = 0; st < 2; ++st) s[st] = Elem<DT>::mfma32(kf[st][ks], qf[ks], ks == 0 ? zero : s[st]);
    } else {
      const unsigned short* Kc = Ks + stage * KSZ + kfoff;
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
        for (int st = 0; st < 2; ++st) {
          const u32x4 a = *reinterpret_cast<const u32x4*>(Kc + st * 32 * (KCH * 8) + ks * 16);
          s[st] = Elem<DT>::mfma32(a, qf[ks], ks == 0 ? zero : s[st]);
        }
    }
  };
  auto half_max = [&](float mx) -> float {           // max over the two lane halves that share a query
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
    return fmaxf(mx, __uint_as_float(hi ? sw[0] : sw[1]));
  };
  float neg_m = INFINITY;                            // -m_run (log2 units, scaled)
  auto exp_pack = [&](f32x16 (&s)[2], const int st) {      // P = 2^(s c - m) of one 32-key half, packed; !MFMASUM: row sum
    float rs = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
#ifndef IDF_ATTN8_FMA_SCALE
      s[st][r] = __builtin_amdgcn_exp2f(s[st][r] + neg_m);
#else
      s[st][r] = __builtin_amdgcn_exp2f(fmaf(s[st][r], c, neg_m));
#endif
      if constexpr (!MFMASUM) rs += s[st][r];
    }
    if constexpr (!MFMASUM) l_run += rs;
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
      for (int w = 0; w < 4; ++w) pk[st * 2 + k2][w] = pack2<DT>(s[st][8 * k2 + 2 * w], s[st][8 * k2 + 2 * w + 1]);
  };
  // O^T += V^T P^T over the two 16-key steps of half st; the V^T fragments stream through two register sets
  auto pv = [&](const int st, const int stage) {
    const unsigned short* Vc = Vs + stage * VSZ + vfoff;
    u32x4 a[2][NMT];
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt) a[0][mt] = *reinterpret_cast<const u32x4*>(Vc + mt * 32 * KVT + (((st * 4 + hi) ^ v_sw) * 8));
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2) {
      if (k2 == 0) {
        const int chunk = st * 4 + 2 + hi;
#pragma unroll
        for (int mt = 0; mt < NMT; ++mt) a[1][mt] = *reinterpret_cast<const u32x4*>(Vc + mt * 32 * KVT + ((chunk ^ v_sw) * 8));
      }
#pragma unroll
      for (int mt = 0; mt < NMT; ++mt) o[mt] = Elem<DT>::mfma32(a[k2][mt], pk[st * 2 + k2], o[mt]);
    }
  };
  // tile t's scores (in s): tail mask, maximum, (rare) rescale of O -- everything in front of the exponentials
  auto decide = [&](f32x16 (&s)[2], const int t) {
    // tail tile (wave-uniform, rare): invalid keys -> -inf (their K rows are clamped duplicates: finite scores)
    {
      const int seg = (t < T0) ? 0 : 1;
      const int nvalid = p.n[seg] - (seg ? (t - T0) : t) * KVT;
      if (nvalid < KVT) {
        // s[st][r] <-> key st*32 + 16 (r >> 3) + (r & 7) + 8 hi: compared against a limit that carries the lane half, from an
        // opaque copy of `hi` (else the 32 key indices are hoisted and live in registers across the hot loop)
        int hi_o = hi;
        asm volatile("" : "+v"(hi_o));
        const int lim = nvalid - 8 * hi_o;
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
          for (int r = 0; r < 16; ++r) s[st][r] = (st * 32 + 16 * (r >> 3) + (r & 7) >= lim) ? -INFINITY : s[st][r];
      }
    }
    float m0 = max3f(s[0][0], s[0][1], s[0][2]), m1 = max3f(s[1][0], s[1][1], s[1][2]);
#pragma unroll
    for (int r = 3; r < 15; r += 2) {
      m0 = max3f(m0, s[0][r], s[0][r + 1]);
      m1 = max3f(m1, s[1][r], s[1][r + 1]);
    }
    m0 = max3f(m0, m1, s[0][15]);
    m0 = max3f(m0, m0, s[1][15]);
    const float mx = half_max(m0) * c;               // c > 0
    if (__builtin_amdgcn_ballot_w64(mx > m_run + DEFER) != 0) {       // first tile: m_run = -inf -> always
      const float m_new = fmaxf(m_run, mx);
      const float al = __builtin_amdgcn_exp2f(m_run - m_new);          // exp2(-inf) = 0 on the first tile (O = l = 0 anyway)
      m_run = m_new;
      neg_m = -m_new;
      l_run *= al;
#pragma unroll
      for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[mt][r] *= al;
    }
  };
  // exponentials of tile t and its P.V (one basic block; PIPE: together with the K.Q^T MFMAs of tile t+1 issued in front of it)
  auto exp_pv = [&](f32x16 (&s)[2], const int t) {
    exp_pack(s, 0);
    pv(0, t & 1);
    exp_pack(s, 1);
    if constexpr (KPRE) load_kf((t + 1) % KST);      // next tile's K fragments (a stale stage after the last tile: unused)
    pv(1, t & 1);
    if constexpr (KPRE) {
#pragma unroll
      for (int st = 0; st < 2; ++st)
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) asm volatile("" : "+v"(kf[st][ks]));     // landed here, under the MFMAs
    }
  };
  auto end_tile = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  };
  // top of tile t.  Every wave passed the barrier that ended tile t-1: K(t) [3-stage ring: and K(t+1)] and V^T(t) are visible;
  // the stages of K(t+KA) and V^T(t+1) were last read in tile t-1 or earlier.
  auto issue = [&](const int t) {
    const int tk = t + KA, tv = t + 1;
    if (tk < F0) {                                   // (tv <= tk): both are full tiles of segment 0
      issue_hot(tk, tv);
    } else {
      if (tk < T) issue_cold(tk, true, false);
      if (tv < T) issue_cold(tv, false, true);
    }
  };

  __syncthreads();                                  // pad rows / ones row complete
  issue_cold(0, true, true);
  if (KST == 3 && T > 1) issue_cold(1, true, false);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if constexpr (PIPE) {
    f32x16 sA[2], sB[2];
    qk(sA, 0);
    for (int t = 0; t < T; t += 2) {
      issue(t);
      decide(sA, t);
      qk(sB, (t + 1) % KST);                         // (behind the last tile: a stale stage, result unused)
      exp_pv(sA, t);
      end_tile();
      if (t + 1 >= T) break;
      issue(t + 1);
      decide(sB, t + 1);
      qk(sA, (t + 2) % KST);
      exp_pv(sB, t + 1);
      end_tile();
    }
  } else {
    f32x16 s[2];
    load_kf(0);
    for (int t = 0; t < T; ++t) {
      issue(t);
      qk(s, t % KST);
      decide(s, t);
      exp_pv(s, t);
      end_tile();
    }
  }

  // ---- normalise and store.  o[mt][r]: e = mt*32 + (r&3) + 8*(r>>2) + 4*hi, q = l31.
  float l_tot;
  if constexpr (MFMASUM) {
    constexpr int sel = (D & 31) >> 3;               // row D of O^T: tile D/32, register 4*sel of the hi = 0 lanes
    l_tot = __shfl(o[NMT - 1][4 * sel], l31, 64);
  } else {
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(l_run), __float_as_uint(l_run), false, false);
    l_tot = l_run + __uint_as_float(hi ? sw[0] : sw[1]);
  }
  const float inv = 1.0f / l_tot;
  unsigned short* const ow = smem + wave * (32 * D);        // wave-private [32 queries][D] (the rings are dead: last barrier passed)
  {
    unsigned short* orow = ow + l31 * D;
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const int e = mt * 32 + 8 * qd + 4 * hi;
        if (e < D) {
          u32x2 pkd = {pack2<DT>(o[mt][4 * qd] * inv, o[mt][4 * qd + 1] * inv),
                       pack2<DT>(o[mt][4 * qd + 2] * inv, o[mt][4 * qd + 3] * inv)};
          *reinterpret_cast<u32x2*>(orow + e) = pkd;
        }
      }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  {
    const int q0 = qb * (NW * 32) + wave * 32;
    unsigned short* const obase = p.out + (size_t)b * p.sO + h * D;
#pragma unroll
    for (int j = 0; j < DCH / 2; ++j) {
      const int c = lane + 64 * j;                    // 16-B chunk of the block, row-major
      const int row = c / DCH, col = c - row * DCH;
      const u32x4 v = *reinterpret_cast<const u32x4*>(ow + c * 8);
      if (q0 + row < p.nq) *reinterpret_cast<u32x4*>(obase + (size_t)(q0 + row) * p.ldo + col * 8) = v;
    }
  }
}

template <int DT, int D, int NW, int MODE>
int launch_one(const AttnParams& p, int B, int order, hipStream_t s) {
  constexpr int NMT = (D + 31) / 32, KCH = (D / 8) | 1;
  constexpr int LDS = ((MODE == 1 ? 2 : 3) * KVT * KCH * 8 + 2 * NMT * 32 * KVT) * 2;
  auto kern = attn8_kernel<DT, D, NW, MODE>;
  static std::atomic<unsigned long long> attr_done{0};
  //  an opt-in failure is reported as the HIP error it is, like the other launchers -- IDF_E_UNSUPPORTED hid it and, not
  // being IDF_ATTN2_UNSUPPORTED, did not even fall back to the register-staged kernel
  if (const int rc = idf_lds_optin(reinterpret_cast<const void*>(kern), LDS, attr_done)) return rc;
  const int nqb = (p.nq + NW * 32 - 1) / (NW * 32);
  hipLaunchKernelGGL(kern, dim3(nqb * p.H * B), dim3(NW * 64), LDS, s, p, nqb, order);
  return idf_launch_status();
}

template <int DT>
int launch_attn8(const AttnParams& p, int B, int mode, hipStream_t s) {
  // mode 1: default (d = 80: two 4-wave workgroups per CU, K fragments one tile ahead; d = 160: 8-wave workgroups from 256
  // queries, 4-wave below); 2..6: A/B variants (tools/ubench/attn_harness.hip, profiles/r05_attn8_*.log)
  const int order = mode == 3 ? 0 : 1;
  if (p.d == 80) {
    if (mode == 2) return launch_one<DT, 80, 8, 0>(p, B, order, s);
    if (mode == 5) return launch_one<DT, 80, 4, 2>(p, B, order, s);
    if (mode == 6) return launch_one<DT, 80, 8, 2>(p, B, order, s);
    return launch_one<DT, 80, 4, 0>(p, B, order, s);
  }
  if (p.d == 160) {
    if (mode == 4 || (mode != 2 && p.nq < 256)) return launch_one<DT, 160, 4, 1>(p, B, order, s);
    return launch_one<DT, 160, 8, 1>(p, B, order, s);
  }
  return IDF_ATTN2_UNSUPPORTED;
}

}  // namespace

std::atomic<long long> idf_stat_attn8_launches{0};

int g_attn8_mode = -2;
int idf_attn8_mode() {
  if (g_attn8_mode == -2) {
    const char* e = getenv("IDF_ATTN8");
    const int v = e ? atoi(e) : IDF_ATTN8_DEFAULT;
    g_attn8_mode = (v < 0 || v > 6) ? IDF_ATTN8_DEFAULT : v;
  }
  return g_attn8_mode;
}
int idf_attn8_set_mode(int v) { const int prev = idf_attn8_mode(); g_attn8_mode = v; return prev; }

int idf_launch_attn8(const AttnParams& p, int B, int dtype, hipStream_t s) {
  const int mode = idf_attn8_mode();
  if (mode == 0) return IDF_ATTN2_UNSUPPORTED;
  if (p.d != 80 && p.d != 160) return IDF_ATTN2_UNSUPPORTED;
  if ((p.n[0] % 8) || (p.n[1] % 8)) return IDF_ATTN2_UNSUPPORTED;
  if ((p.ldk[0] % 8) || (p.ldv[0] % 8) || (p.n[1] > 0 && ((p.ldk[1] % 8) || (p.ldv[1] % 8)))) return IDF_ATTN2_UNSUPPORTED;
  if (!aligned16(p.k[0]) || !aligned16(p.vt[0]) || !aligned16(p.k[1]) || !aligned16(p.vt[1])) return IDF_ATTN2_UNSUPPORTED;
  if ((p.sK[0] % 8) || (p.sV[0] % 8) || (p.sK[1] % 8) || (p.sV[1] % 8)) return IDF_ATTN2_UNSUPPORTED;
  if (!aligned16(p.out) || (p.ldo % 8) || (p.sO % 8) || !aligned16(p.q) || (p.ldq % 8) || (p.sQ % 8)) return IDF_ATTN2_UNSUPPORTED;
  // per-lane DMA offsets are 32-bit: a (batch, head) slice of K / V^T must stay below 2 GB
  if ((long long)KVT * p.ldk[0] * 2 >= (1ll << 31) || (long long)p.d * p.ldv[0] * 2 >= (1ll << 31)) return IDF_ATTN2_UNSUPPORTED;
  if (dtype == IDF_BF16) return launch_attn8<IDF_BF16>(p, B, mode, s);
  if (dtype == IDF_F16) return launch_attn8<IDF_F16>(p, B, mode, s);
  return IDF_ATTN2_UNSUPPORTED;
}
