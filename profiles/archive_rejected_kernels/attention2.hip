// attention2.hip -- attention forward, variant 2 (gfx950): 64 queries per wave, LDS-DMA staged K / V^T tiles.
//
// Why (measured, profiles/r01_shape_profile_B64.log): the 64x64-latent self / gated-self attention (d = 40, 4096-4280
// keys) is 29 % of the UNet forward and attention.hip runs it at ~590 TFLOP/s.  There every wave owns 32 queries, so
// each 64-key tile costs 14 ds_read_b128 per 14 MFMAs per wave plus a register round trip (global -> VGPR -> ds_write)
// of the tile: with 5 workgroups per CU the LDS pipe is ~80 % busy.
// This variant:
//   * a wave owns TWO groups of 32 queries and every K / V^T fragment read from LDS feeds two MFMAs
//     (14 reads per 28 MFMAs); a workgroup of 4 waves covers 256 queries, halving tile staging per flop as well;
//   * the tiles are staged by LDS-DMA (global_load_lds_dwordx4), no staging registers, no ds_write:
//       K   tile: 64 rows x d elements, linear rows of d*2 bytes -- conflict-free for ds_read_b128 because d/8 is ODD
//                 (the kernel is instantiated for d = 8*(2*NKS-1): 24, 40, 56 -- d = 40 is the SD-1.5 64x64 level);
//       V^T tile: d rows x 64 keys, 128-B rows, 16-B slot ^= (row >> 1) & 7 applied on the global source address;
//   * the V^T image is in NATURAL key order, and so is the packed P fragment: the K fragment rows are read in a permuted
//     order (the two middle 4-row blocks of every 16 keys exchanged) so that the S^T registers of a lane-half come out as
//     8 consecutive keys per 16-key step -- no lane exchange, no permuted V^T image;
//   * softmax denominator from an all-ones row d of the V^T image (d < 32*NMT always holds for these d);
//   * the O rescale is skipped when no lane of the wave raised its running max (alpha == 1 exactly).
// Same algorithm / numerics contract as attention.hip (online softmax in fp32, P rounded to 16 bit before P.V).
// Requirements checked by idf_launch_attn2: d as above, n0 % 8 == 0, n1 % 8 == 0 (a partially valid 16-B V^T chunk cannot
// be masked in flight; fully invalid chunks are redirected to a page of zeros).
#include "attn_core.h"

using namespace idfattn;

namespace {

__device__ __attribute__((aligned(128))) unsigned short idf_attn_zero_page[64];

constexpr int KVT = 64;

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// LAZY = true ("lazy rescaling"): the VALU work per score drops from {fma, exp2, 1/2 max, 1/2 cvt} to {exp2, 1/2 max, 1/2 cvt}:
//   * Q is pre-multiplied by scale*log2(e) when it is loaded (rounded to 16 bit once -- the same thing torch's math SDPA
//     does with q * sqrt(scale)), so the MFMA result is already the exponent;
//   * the running reference value m of each query is SUBTRACTED BY THE MFMA ITSELF: the accumulator of K.Q^T is initialised
//     with -m (16 registers per group holding -m) instead of 0;
//   * m is not the exact running max but a LAGGING one: it is only raised (and O rescaled, the -m registers rewritten,
//     the current tile re-based) when some score of the tile exceeds it by more than 2^LAZY_THR -- exact arithmetic either
//     way (softmax is shift invariant; P <= 2^LAZY_THR keeps the same relative precision in bf16 / fp32).  The first tile
//     always re-bases to its exact max.
constexpr float LAZY_THR = 8.0f;

template <int DT, int NKS, int NMT, bool LAZY>
__global__ __launch_bounds__(256, 2) void attn2_kernel(const AttnParams p) {
  constexpr int DCH = 2 * NKS - 1;                 // 16-B chunks per K row
  constexpr int D = 8 * DCH;                       // head dim
  constexpr int KSZ = KVT * D;                     // K stage (elements)
  constexpr int VROWS = NMT * 32;
  constexpr int VSZ = VROWS * KVT;                 // V^T stage (elements), 128-B rows
  constexpr int STG = KSZ + VSZ;
  constexpr int K_INST = DCH;                      // LDS-DMA instructions per K tile (64 chunks each)
  constexpr int V_INST = D / 8;                    // per V^T tile (8 rows each)
  constexpr int K_PER_WAVE = (K_INST + 3) / 4, V_PER_WAVE = (V_INST + 3) / 4;
  static_assert(D < 32 * NMT, "needs a spare O^T row for the softmax denominator");
  // The K fragment of the last K-step reads 8 elements past a row (zero Q columns multiply them): for the last row of
  // the tile that is the first V^T row of the same stage -- always finite data.
  __shared__ __attribute__((aligned(128))) unsigned short smem[2 * STG];

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y;

  // zero both stages once (pad rows of V^T must be finite zeros), then the ones row
  for (int i = tid; i < STG; i += 256) reinterpret_cast<unsigned*>(smem)[i] = 0u;
  __syncthreads();
  {
    const unsigned short one = Elem<DT>::from_f32(1.0f);
    for (int i = tid; i < 2 * KVT; i += 256) smem[(i / KVT) * STG + KSZ + D * KVT + (i % KVT)] = one;
  }

  // ---- Q fragments (B operand) of the two query groups: lane holds q = l31, e = 16*ks + 8*hi .. +7
  u32x4 qf[2][NKS];
  int qrow[2];
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    qrow[g] = blockIdx.x * 256 + wave * 64 + g * 32 + l31;
    const int qr = min(qrow[g], p.nq - 1);
    const unsigned short* qp = p.q + (size_t)b * p.sQ + (size_t)qr * p.ldq + h * D;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      const int e0 = ks * 16 + hi * 8;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (e0 < D) v = *reinterpret_cast<const u32x4*>(qp + e0);
      if constexpr (LAZY) {
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

  // ---- DMA roles.  K: instruction i (= wave + 4j) moves linear chunks 64 i .. 64 i + 63 of the tile: chunk g -> row g / DCH,
  // column chunk g % DCH.  V^T: instruction i moves rows 8 i .. 8 i + 7: lane -> row 8 i + (lane >> 3), slot lane & 7.
  int k_row[K_PER_WAVE], k_col[K_PER_WAVE];
#pragma unroll
  for (int j = 0; j < K_PER_WAVE; ++j) {
    const int g = (wave + 4 * j) * 64 + lane;
    k_row[j] = g / DCH;
    k_col[j] = (g - k_row[j] * DCH) * 8;
  }
  int v_row[V_PER_WAVE], v_chunk[V_PER_WAVE];
#pragma unroll
  for (int j = 0; j < V_PER_WAVE; ++j) {
    const int row = (wave + 4 * j) * 8 + (lane >> 3);
    v_row[j] = row;
    v_chunk[j] = (lane & 7) ^ ((row >> 1) & 7);      // global 8-key chunk that lands in LDS slot lane & 7
  }

  auto issue_dma = [&](int t, int stage) {
    const int seg = (t < T0) ? 0 : 1;
    const int kv0 = (seg ? (t - T0) : t) * KVT;
    const int n = p.n[seg];
    const int ldk = p.ldk[seg], ldv = p.ldv[seg];
    const unsigned short* kb = p.k[seg] + (size_t)b * p.sK[seg] + h * D;
    const unsigned short* vb = p.vt[seg] + (size_t)b * p.sV[seg] + (size_t)(h * D) * ldv + kv0;
    unsigned short* Ks = smem + stage * STG;
    unsigned short* Vs = Ks + KSZ;
#pragma unroll
    for (int j = 0; j < K_PER_WAVE; ++j) {
      if (wave + 4 * j < K_INST) {
        const int kr = min(kv0 + k_row[j], n - 1);                     // tail rows: clamped, their scores are masked
        __builtin_amdgcn_global_load_lds((gptr_t)(kb + (size_t)kr * ldk + k_col[j]),
                                         (lptr_t)(Ks + (wave + 4 * j) * 512), 16, 0, 0);
      }
    }
#pragma unroll
    for (int j = 0; j < V_PER_WAVE; ++j) {
      if (wave + 4 * j < V_INST) {
        const bool valid = (kv0 + v_chunk[j] * 8) < n;                 // n % 8 == 0: a chunk is all valid or all invalid
        const unsigned short* src = valid ? vb + (size_t)v_row[j] * ldv + v_chunk[j] * 8 : idf_attn_zero_page + (lane & 7) * 8;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(Vs + (wave + 4 * j) * 512), 16, 0, 0);
      }
    }
  };

  f32x16 o[2][NMT];
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[g][mt][r] = 0.0f;
  float m_run[2] = {LAZY ? 0.0f : -INFINITY, LAZY ? 0.0f : -INFINITY};
  f32x16 cinit[2];                                  // LAZY: -m_lag of the group's query in every register (MFMA C operand)
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int r = 0; r < 16; ++r) cinit[g][r] = 0.0f;
  const float c = p.scale_log2;
  const int v_sw = (l31 >> 1) & 7;                  // V^T fragment rows are mt*32 + l31
  // K fragment row permutation: MFMA row i = 16u + 8a + 4h + j of a 32-key half carries key 16u + 8h + 4a + j (bits a and h
  // swapped, i.e. the two middle 4-row blocks of every 16 exchanged).  S^T register r of lane-half `hi` sits in MFMA row
  // (r&3) + 8*(r>>2) + 4*hi, so it then holds key 16*(r>>3) + 8*hi + (r&7): the 8 registers of one 16-key step are 8
  // CONSECUTIVE keys in register order -- exactly the k order of the P.V operands, no lane exchange needed.
  const int kperm = (l31 & 0x13) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);

  __syncthreads();                                  // zero fill + ones rows complete before the first DMA lands on them
  issue_dma(0, 0);
  for (int t = 0; t < T; ++t) {
    // tile t landed (every wave waits for ITS OWN DMA, then the barrier publishes all of them) and every wave is done
    // with tile t-1, whose stage the next DMA overwrites
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t + 1 < T) issue_dma(t + 1, (t + 1) & 1);
    const unsigned short* Kc = smem + (t & 1) * STG;
    const unsigned short* Vc = Kc + KSZ;
    const int seg = (t < T0) ? 0 : 1;
    const int kv0 = (seg ? (t - T0) : t) * KVT;
    const int nvalid = p.n[seg] - kv0;              // >= 1

    // ---- S^T = K Q^T for both query groups; every K fragment feeds two MFMAs
    f32x16 s[2][2];                                 // [kv half][query group]
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      const unsigned short* kf = Kc + (st * 32 + kperm) * D + hi * 8;
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        const u32x4 a = *reinterpret_cast<const u32x4*>(kf + ks * 16);
        if (ks == 0) {
          if constexpr (LAZY) {
            s[st][0] = Elem<DT>::mfma32(a, qf[0][0], cinit[0]);
            s[st][1] = Elem<DT>::mfma32(a, qf[1][0], cinit[1]);
          } else {
            const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            s[st][0] = Elem<DT>::mfma32(a, qf[0][0], zero);
            s[st][1] = Elem<DT>::mfma32(a, qf[1][0], zero);
          }
        } else {
          s[st][0] = Elem<DT>::mfma32(a, qf[0][ks], s[st][0]);
          s[st][1] = Elem<DT>::mfma32(a, qf[1][ks], s[st][1]);
        }
      }
    }
    if constexpr (LAZY) {
      // ---- lazy-rescaling softmax: s already is (score*scale*log2e - m_lag) of its query
      float pm[2];
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        if (nvalid < KVT) {
          int nv = nvalid;
          asm volatile("" : "+s"(nv));
#pragma unroll
          for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int kv = st * 32 + 16 * (r >> 3) + 8 * hi + (r & 7);
              s[st][g][r] = (kv >= nv) ? -INFINITY : s[st][g][r];
            }
        }
        float m0 = fmaxf(s[0][g][0], s[0][g][1]), m1 = fmaxf(s[1][g][0], s[1][g][1]);
#pragma unroll
        for (int r = 2; r < 16; r += 2) {
          m0 = fmaxf(fmaxf(m0, s[0][g][r]), s[0][g][r + 1]);
          m1 = fmaxf(fmaxf(m1, s[1][g][r]), s[1][g][r + 1]);
        }
        pm[g] = fmaxf(m0, m1);
      }
      if (t == 0 || __builtin_amdgcn_ballot_w64((pm[0] > LAZY_THR) | (pm[1] > LAZY_THR)) != 0) {
        // re-base (rare after the first tiles): exact max of the tile per query, shift the tile, rescale O (its row D is l)
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(pm[g]), __float_as_uint(pm[g]), false, false);
          const float mx = fmaxf(pm[g], __uint_as_float(hi ? sw[0] : sw[1]));
          const float delta = (t == 0) ? mx : fmaxf(mx, 0.0f);
          const float al = __builtin_amdgcn_exp2f(-delta);
          m_run[g] += delta;                                  // m_run starts at 0 in this mode
#pragma unroll
          for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int r = 0; r < 16; ++r) s[st][g][r] -= delta;
#pragma unroll
          for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[g][mt][r] *= al;
#pragma unroll
          for (int r = 0; r < 16; ++r) cinit[g][r] = -m_run[g];
        }
      }
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
          for (int r = 0; r < 16; ++r) s[st][g][r] = __builtin_amdgcn_exp2f(s[st][g][r]);
    } else {
    // ---- online softmax per group.  s[st][g][r]: key = kv0 + st*32 + 16*(r>>3) + 8*hi + (r&7), query = l31 of group g
    float alpha[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      if (nvalid < KVT) {
        int nv = nvalid;
        asm volatile("" : "+s"(nv));                  // keep the 32 compares inside the (rare) tail branch
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int kv = st * 32 + 16 * (r >> 3) + 8 * hi + (r & 7);
            s[st][g][r] = (kv >= nv) ? -INFINITY : s[st][g][r];
          }
      }
      float mx = s[0][g][0];
#pragma unroll
      for (int st = 0; st < 2; ++st)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[st][g][r]);
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float m_new = fmaxf(m_run[g], mx * c);                    // c > 0
      alpha[g] = __builtin_amdgcn_exp2f(m_run[g] - m_new);            // first tile: exp2(-inf) = 0
      m_run[g] = m_new;
#pragma unroll
      for (int st = 0; st < 2; ++st)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[st][g][r] = __builtin_amdgcn_exp2f(fmaf(s[st][g][r], c, -m_new));
    }
    // rescale O only when some lane's running max moved (alpha == 1 exactly otherwise) -- wave-uniform branch
    if (__builtin_amdgcn_ballot_w64((alpha[0] != 1.0f) | (alpha[1] != 1.0f)) != 0) {
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[g][mt][r] *= alpha[g];
    }
    }

    // ---- O^T += V^T P^T.  K-step (st, k2) = keys st*32 + 16*k2 .. +15; lane-half `hi` supplies keys 8*hi .. 8*hi+7 of the
    // step from its registers 8*k2 .. 8*k2+7 (see kperm), the V^T fragment is a plain 16-B read of the same 8 keys.
#pragma unroll
    for (int st = 0; st < 2; ++st) {
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2) {
        u32x4 pf[2];
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
          for (int w = 0; w < 4; ++w) pf[g][w] = pack2<DT>(s[st][g][8 * k2 + 2 * w], s[st][g][8 * k2 + 2 * w + 1]);
        const int chunk = st * 4 + k2 * 2 + hi;            // 8-key chunk of the tile
#pragma unroll
        for (int mt = 0; mt < NMT; ++mt) {
          const u32x4 a = *reinterpret_cast<const u32x4*>(Vc + (mt * 32 + l31) * KVT + ((chunk ^ v_sw) * 8));
          o[0][mt] = Elem<DT>::mfma32(a, pf[0], o[0][mt]);
          o[1][mt] = Elem<DT>::mfma32(a, pf[1], o[1][mt]);
        }
      }
    }
  }

  // ---- normalise and store.  o[g][mt][r]: e = mt*32 + (r&3) + 8*(r>>2) + 4*hi, q = l31 of group g.
  // row e = D of O^T holds l: tile D/32, register 4*((D%32)/8) of the hi = 0 lanes
  constexpr int sel = (D & 31) >> 3;
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const float lv = o[g][NMT - 1][4 * sel];
    const float l_tot = __shfl(lv, l31, 64);               // broadcast from the hi = 0 lane of this query
    const float inv = 1.0f / l_tot;
    if (qrow[g] < p.nq) {
      unsigned short* op = p.out + (size_t)b * p.sO + (size_t)qrow[g] * p.ldo + h * D;
#pragma unroll
      for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          const int e = mt * 32 + 8 * qd + 4 * hi;
          if (e < D) {
            u32x2 pk = {pack2<DT>(o[g][mt][4 * qd] * inv, o[g][mt][4 * qd + 1] * inv),
                        pack2<DT>(o[g][mt][4 * qd + 2] * inv, o[g][mt][4 * qd + 3] * inv)};
            *reinterpret_cast<u32x2*>(op + e) = pk;
          }
        }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Variant 3: the same data layout, software-pipelined INSIDE the wave.  PMC on variant 2 (profiles/r01_pmc_attention.log):
// the MFMA pipe is busy 41 % and the wave spends its time in VALU (softmax) and MFMA phases one after the other -- with 2
// waves per SIMD that start every tile together behind the same kind of barrier, the phases of the two waves coincide
// instead of interleaving.  Here each half-iteration ("slot") pairs the softmax of one query group (VALU: ~100
// instructions, 33 of them v_exp_f32) with MFMAs that do not depend on it:
//   slot A(t):  softmax(a, t)   ||   P.V(b, t-1)  (8 MFMA)  +  K.Q^T(b, t)    (6 MFMA)
//   slot B(t):  softmax(b, t)   ||   P.V(a, t)    (8 MFMA)  +  K.Q^T(a, t+1)  (6 MFMA)
// so both pipes have ~450 cycles of independent work per slot and all dependencies run slot -> next slot.
// A 3-stage LDS ring (tile t-1 is still read in slot A(t) while tile t+1 lands); one barrier per tile, at the A|B boundary.
// The tail tile needs NO masking arithmetic: its K rows beyond n are clamped duplicates of the last valid key (a valid
// score, so the running max is unaffected), its V^T columns beyond n come from the page of zeros, and the all-ones row
// that produces the softmax denominator is re-staged per tile from a ones / zeros page pair with the same validity rule.
// ------------------------------------------------------------------------------------------------------------------
__device__ __attribute__((aligned(128))) unsigned short idf_attn_ones_page[2][64] = {
    {0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80,
     0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80,
     0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80,
     0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80},
    {0x3c00, 0x3c00, 0x3c00, 0x3c00, 0x3c00, 0x3c00, 0x3c00, 0x3c00, 0x3c00, 0x3c00, 0x3c00, 0x3c00, 0x3c00, 0x3c00, 0x3c00, 0x3c00,
     0x3c00, 0x3c00, 0x3c00, 0x3c00, 0x3c00, 0x3c00, 0x3c00, 0x3c00, 0x3c00, 0x3c00, 0x3c00, 0x3c00, 0x3c00, 0x3c00, 0x3c00, 0x3c00,
     0x3c00, 0x3c00, 0x3c00, 0x3c00, 0x3c00, 0x3c00, 0x3c00, 0x3c00, 0x3c00, 0x3c00, 0x3c00, 0x3c00, 0x3c00, 0x3c00, 0x3c00, 0x3c00,
     0x3c00, 0x3c00, 0x3c00, 0x3c00, 0x3c00, 0x3c00, 0x3c00, 0x3c00, 0x3c00, 0x3c00, 0x3c00, 0x3c00, 0x3c00, 0x3c00, 0x3c00, 0x3c00}};

template <int I> struct AIC { static constexpr int value = I; };
template <int I, int N, class F> __device__ __forceinline__ void static_for_a(F&& f) {
  if constexpr (I < N) { f(AIC<I>{}); static_for_a<I + 1, N>(f); }
}

template <int DT, int NKS, int NMT, bool LAZY>
__global__ __launch_bounds__(256, 2) void attn3_kernel(const AttnParams p) {
  constexpr int DCH = 2 * NKS - 1;
  constexpr int D = 8 * DCH;
  constexpr int KSZ = KVT * D;
  constexpr int VROWS = NMT * 32;
  constexpr int VSZ = VROWS * KVT;
  constexpr int STG = KSZ + VSZ;
  constexpr int NSTG = 3;
  constexpr int K_INST = DCH;
  constexpr int V_INST = D / 8 + 1;                // + the 8-row group that starts with the ones row (row D)
  constexpr int K_PER_WAVE = (K_INST + 3) / 4, V_PER_WAVE = (V_INST + 3) / 4;
  static_assert(D + 8 <= VROWS, "ones-row group must fit the V^T image");
  __shared__ __attribute__((aligned(128))) unsigned short smem[NSTG * STG];

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y;

  for (int i = tid; i < NSTG * STG / 2; i += 256) reinterpret_cast<unsigned*>(smem)[i] = 0u;

  u32x4 qf[2][NKS];
  int qrow[2];
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    qrow[g] = blockIdx.x * 256 + wave * 64 + g * 32 + l31;
    const int qr = min(qrow[g], p.nq - 1);
    const unsigned short* qp = p.q + (size_t)b * p.sQ + (size_t)qr * p.ldq + h * D;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      const int e0 = ks * 16 + hi * 8;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (e0 < D) v = *reinterpret_cast<const u32x4*>(qp + e0);
      if constexpr (LAZY) {
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

  int k_row[K_PER_WAVE], k_col[K_PER_WAVE];
#pragma unroll
  for (int j = 0; j < K_PER_WAVE; ++j) {
    const int g = (wave + 4 * j) * 64 + lane;
    k_row[j] = g / DCH;
    k_col[j] = (g - k_row[j] * DCH) * 8;
  }
  int v_row[V_PER_WAVE], v_chunk[V_PER_WAVE];
#pragma unroll
  for (int j = 0; j < V_PER_WAVE; ++j) {
    const int row = (wave + 4 * j) * 8 + (lane >> 3);
    v_row[j] = row;
    v_chunk[j] = (lane & 7) ^ ((row >> 1) & 7);
  }
  const unsigned short* ones = idf_attn_ones_page[DT == IDF_BF16 ? 0 : 1];

  auto issue_dma = [&](int t, int stage) {
    const int seg = (t < T0) ? 0 : 1;
    const int kv0 = (seg ? (t - T0) : t) * KVT;
    const int n = p.n[seg];
    const int ldk = p.ldk[seg], ldv = p.ldv[seg];
    const unsigned short* kb = p.k[seg] + (size_t)b * p.sK[seg] + h * D;
    const unsigned short* vb = p.vt[seg] + (size_t)b * p.sV[seg] + (size_t)(h * D) * ldv + kv0;
    unsigned short* Ks = smem + stage * STG;
    unsigned short* Vs = Ks + KSZ;
#pragma unroll
    for (int j = 0; j < K_PER_WAVE; ++j) {
      if (wave + 4 * j < K_INST) {
        const int kr = min(kv0 + k_row[j], n - 1);
        __builtin_amdgcn_global_load_lds((gptr_t)(kb + (size_t)kr * ldk + k_col[j]),
                                         (lptr_t)(Ks + (wave + 4 * j) * 512), 16, 0, 0);
      }
    }
#pragma unroll
    for (int j = 0; j < V_PER_WAVE; ++j) {
      if (wave + 4 * j < V_INST) {
        const bool valid = (kv0 + v_chunk[j] * 8) < n;
        const unsigned short* src = idf_attn_zero_page + (lane & 7) * 8;
        if (valid) src = (v_row[j] < D) ? vb + (size_t)v_row[j] * ldv + v_chunk[j] * 8
                                        : ((v_row[j] == D) ? ones + (lane & 7) * 8 : src);
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(Vs + (wave + 4 * j) * 512), 16, 0, 0);
      }
    }
  };

  f32x16 o[2][NMT], s[2][2];                        // s[group][kv half]
  u32x4 pk[2][2][2];                                // packed P: [group][kv half][16-key step]
#pragma unroll
  for (int g = 0; g < 2; ++g) {
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[g][mt][r] = 0.0f;
#pragma unroll
    for (int st = 0; st < 2; ++st)
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2) pk[g][st][k2] = u32x4{0u, 0u, 0u, 0u};
  }
  float m_run[2] = {LAZY ? 0.0f : -INFINITY, LAZY ? 0.0f : -INFINITY};
  f32x16 cinit[2];                                  // LAZY: -m_lag of the group's query in every register (MFMA C operand)
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int r = 0; r < 16; ++r) cinit[g][r] = 0.0f;
  const float c = p.scale_log2;
  const int v_sw = (l31 >> 1) & 7;
  const int kperm = (l31 & 0x13) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);
  const int koff = kperm * D + hi * 8;                        // K fragment element offset inside a 32-key half
  const int voff = l31 * KVT;                                 // V^T fragment row offset

  // The slot's MFMA list for group mg: i < 4*NMT: P.V step (st, k2, mt) out of the V^T image Vp; then K.Q^T step (st, ks)
  // out of the K image Kq.  frag_read / mfma_do are separate so the LDS reads run two steps ahead of their MFMAs.
  auto frag_read = [&](auto II, const unsigned short* Kq, const unsigned short* Vp) -> u32x4 {
    constexpr int i = decltype(II)::value;
    if constexpr (i < 4 * NMT) {
      constexpr int st = i / (2 * NMT), k2 = (i / NMT) & 1, mt = i % NMT;
      const int chunk = st * 4 + k2 * 2 + hi;
      return *reinterpret_cast<const u32x4*>(Vp + mt * 32 * KVT + voff + ((chunk ^ v_sw) * 8));
    } else {
      constexpr int j = i - 4 * NMT, st = j / NKS, ks = j % NKS;
      return *reinterpret_cast<const u32x4*>(Kq + st * 32 * D + koff + ks * 16);
    }
  };
  auto mfma_do = [&](auto II, auto MG, const u32x4 a) {
    constexpr int i = decltype(II)::value, mg = decltype(MG)::value;
    if constexpr (i < 4 * NMT) {
      constexpr int st = i / (2 * NMT), k2 = (i / NMT) & 1, mt = i % NMT;
      o[mg][mt] = Elem<DT>::mfma32(a, pk[mg][st][k2], o[mg][mt]);
    } else {
      constexpr int j = i - 4 * NMT, st = j / NKS, ks = j % NKS;
      if constexpr (ks == 0) {
        if constexpr (LAZY) {
          s[mg][st] = Elem<DT>::mfma32(a, qf[mg][0], cinit[mg]);
        } else {
          const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          s[mg][st] = Elem<DT>::mfma32(a, qf[mg][0], zero);
        }
      } else {
        s[mg][st] = Elem<DT>::mfma32(a, qf[mg][ks], s[mg][st]);
      }
    }
  };
  constexpr int NM = 4 * NMT + 2 * NKS;              // MFMAs per slot (14 for d = 40)
  constexpr int NM1 = 3;                             // of which issued beside the max chain

  // slot: VALU softmax of group vg (scores s[vg] -> packed pk[vg], running max, O rescale) interleaved in PROGRAM ORDER
  // with the NM independent MFMAs of group 1-vg.  sched_barrier(0) after every step pins "one MFMA + its share of the
  // softmax" together (hipcc otherwise sinks the whole softmax next to its consumer in the NEXT slot).
  auto slot = [&](auto VG, const unsigned short* Kq, const unsigned short* Vp, const bool first) {
    constexpr int vg = decltype(VG)::value, mg = 1 - vg;
    u32x4 fr[3];
    fr[0] = frag_read(AIC<0>{}, Kq, Vp);
    fr[1] = frag_read(AIC<1>{}, Kq, Vp);
    // ---- part 1: max chain beside the first MFMAs
    fr[2] = frag_read(AIC<2>{}, Kq, Vp);
    mfma_do(AIC<0>{}, AIC<mg>{}, fr[0]);
    float mx0 = fmaxf(s[vg][0][0], s[vg][0][1]), mx1 = fmaxf(s[vg][1][0], s[vg][1][1]);
#pragma unroll
    for (int r = 2; r < 16; r += 2) { mx0 = fmaxf(fmaxf(mx0, s[vg][0][r]), s[vg][0][r + 1]); }
    __builtin_amdgcn_sched_barrier(0);
    fr[0] = frag_read(AIC<3>{}, Kq, Vp);
    mfma_do(AIC<1>{}, AIC<mg>{}, fr[1]);
#pragma unroll
    for (int r = 2; r < 16; r += 2) { mx1 = fmaxf(fmaxf(mx1, s[vg][1][r]), s[vg][1][r + 1]); }
    float mx = fmaxf(mx0, mx1);
    float m_new = 0.0f;
    if constexpr (!LAZY) {
      const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
      mx = fmaxf(mx, __uint_as_float(hi ? sw[0] : sw[1]));
    }
    __builtin_amdgcn_sched_barrier(0);
    fr[1] = frag_read(AIC<4>{}, Kq, Vp);
    mfma_do(AIC<2>{}, AIC<mg>{}, fr[2]);
    if constexpr (LAZY) {
      // lazy rescaling (see attn2_kernel): s already is exponent - m_lag; re-base only on the first tile or when a score
      // of this tile exceeds the lagging reference by more than 2^LAZY_THR
      if (first || __builtin_amdgcn_ballot_w64(mx > LAZY_THR) != 0) {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
        mx = fmaxf(mx, __uint_as_float(hi ? sw[0] : sw[1]));
        const float delta = first ? mx : fmaxf(mx, 0.0f);
        const float al = __builtin_amdgcn_exp2f(-delta);
        m_run[vg] += delta;
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
          for (int r = 0; r < 16; ++r) s[vg][st][r] -= delta;
#pragma unroll
        for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[vg][mt][r] *= al;
#pragma unroll
        for (int r = 0; r < 16; ++r) cinit[vg][r] = -m_run[vg];
      }
    } else {
      m_new = fmaxf(m_run[vg], mx * c);
      const float alpha = __builtin_amdgcn_exp2f(m_run[vg] - m_new);
      m_run[vg] = m_new;
      if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {
#pragma unroll
        for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[vg][mt][r] *= alpha;
      }
    }
    // ---- part 2: 32 x (fma, exp2) + 16 cvt_pk spread over the remaining MFMAs
    constexpr int REST = NM - NM1;
    constexpr int PER = (32 + REST - 1) / REST;        // scores handled after each MFMA
    static_for_a<0, REST>([&](auto JJ) {
      constexpr int j = decltype(JJ)::value, i = NM1 + j;
      if constexpr (i + 2 < NM) fr[(i + 2) % 3] = frag_read(AIC<i + 2>{}, Kq, Vp);
      mfma_do(AIC<i>{}, AIC<mg>{}, fr[i % 3]);
#pragma unroll
      for (int e = j * PER; e < (j + 1) * PER && e < 32; ++e) {
        const int st = e >> 4, r = e & 15;
        if constexpr (LAZY) s[vg][st][r] = __builtin_amdgcn_exp2f(s[vg][st][r]);
        else s[vg][st][r] = __builtin_amdgcn_exp2f(fmaf(s[vg][st][r], c, -m_new));
        if ((e & 1) == 1) {
          const int k2 = r >> 3, w = (r & 7) >> 1;
          unsigned pw = pack2<DT>(s[vg][st][r - 1], s[vg][st][r]);
          asm volatile("" : "+v"(pw));                // opaque use: keeps the softmax HERE (LLVM otherwise sinks it to its
          pk[vg][st][k2][w] = pw;                     // consumer, the P.V MFMAs of the next slot)
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    });
  };

  __syncthreads();                                  // zero fill complete
  issue_dma(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (T > 1) issue_dma(1, 1);
  {                                                 // prologue: S_a of tile 0
    const unsigned short* Kq = smem;
#pragma unroll
    for (int st = 0; st < 2; ++st)
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        const u32x4 a = *reinterpret_cast<const u32x4*>(Kq + st * 32 * D + koff + ks * 16);
        if (ks == 0) {
          const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          s[0][st] = Elem<DT>::mfma32(a, qf[0][0], zero);
        } else {
          s[0][st] = Elem<DT>::mfma32(a, qf[0][ks], s[0][st]);
        }
      }
  }
  int st_cur = 0, st_prev = 2, st_next = 1;         // ring stage of tile t, t-1, t+1
  for (int t = 0; t < T; ++t) {
    // slot A: softmax(a, t) || P.V(b, t-1) [stage of t-1; all zeros for t = 0] + K.Q^T(b, t)
    slot(AIC<0>{}, smem + st_cur * STG, smem + st_prev * STG + KSZ, t == 0);
    // tile t+1 landed (own DMA, then the barrier publishes everyone's); every wave is past its last read of tile t-1
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t + 2 < T) issue_dma(t + 2, st_prev);
    // slot B: softmax(b, t) || P.V(a, t) + K.Q^T(a, t+1)   (for t = T-1 the K.Q^T reads a stale stage; its result is unused)
    slot(AIC<1>{}, smem + st_next * STG, smem + st_cur * STG + KSZ, t == 0);
    const int tmp = st_prev; st_prev = st_cur; st_cur = st_next; st_next = tmp;
  }
  // epilogue: P.V(b, T-1)
  {
    const unsigned short* Vp = smem + st_prev * STG + KSZ;
#pragma unroll
    for (int st = 0; st < 2; ++st)
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2) {
        const int chunk = st * 4 + k2 * 2 + hi;
#pragma unroll
        for (int mt = 0; mt < NMT; ++mt) {
          const u32x4 a = *reinterpret_cast<const u32x4*>(Vp + mt * 32 * KVT + voff + ((chunk ^ v_sw) * 8));
          o[1][mt] = Elem<DT>::mfma32(a, pk[1][st][k2], o[1][mt]);
        }
      }
  }

  constexpr int sel = (D & 31) >> 3;
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const float lv = o[g][NMT - 1][4 * sel];
    const float l_tot = __shfl(lv, l31, 64);
    const float inv = 1.0f / l_tot;
    if (qrow[g] < p.nq) {
      unsigned short* op = p.out + (size_t)b * p.sO + (size_t)qrow[g] * p.ldo + h * D;
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
int launch_attn2(const AttnParams& p, int B, hipStream_t s) {
  dim3 grid((p.nq + 255) / 256, p.H, B), block(256);
#define IDF_ATTN2_CASE(KS, MT) \
  if (p.d == 8 * (2 * KS - 1)) { \
    const int mode = idf_attn2_mode(); \
    if (mode == 2) hipLaunchKernelGGL((attn3_kernel<DT, KS, MT, false>), grid, block, 0, s, p); \
    else if (mode == 4) hipLaunchKernelGGL((attn3_kernel<DT, KS, MT, true>), grid, block, 0, s, p); \
    else if (mode == 3) hipLaunchKernelGGL((attn2_kernel<DT, KS, MT, true>), grid, block, 0, s, p); \
    else hipLaunchKernelGGL((attn2_kernel<DT, KS, MT, false>), grid, block, 0, s, p); \
    return idf_launch_status(); }
  IDF_ATTN2_CASE(2, 1)    // d = 24
  IDF_ATTN2_CASE(3, 2)    // d = 40
  IDF_ATTN2_CASE(4, 2)    // d = 56
#undef IDF_ATTN2_CASE
  return IDF_ATTN2_UNSUPPORTED;
}

}  // namespace

std::atomic<long long> idf_stat_attn2_launches{0};

int idf_launch_attn2(const AttnParams& p, int B, int dtype, hipStream_t s) {
  if (p.d != 24 && p.d != 40 && p.d != 56) return IDF_ATTN2_UNSUPPORTED;
  if ((p.n[0] % 8) || (p.n[1] % 8)) return IDF_ATTN2_UNSUPPORTED;
  if ((p.ldk[0] % 8) || (p.ldv[0] % 8) || (p.n[1] > 0 && ((p.ldk[1] % 8) || (p.ldv[1] % 8)))) return IDF_ATTN2_UNSUPPORTED;
  if (!aligned16(p.k[0]) || !aligned16(p.vt[0]) || !aligned16(p.k[1]) || !aligned16(p.vt[1])) return IDF_ATTN2_UNSUPPORTED;
  if ((p.sK[0] % 8) || (p.sV[0] % 8) || (p.sK[1] % 8) || (p.sV[1] % 8)) return IDF_ATTN2_UNSUPPORTED;
  int rc = IDF_ATTN2_UNSUPPORTED;
  if (dtype == IDF_BF16) rc = launch_attn2<IDF_BF16>(p, B, s);
  else if (dtype == IDF_F16) rc = launch_attn2<IDF_F16>(p, B, s);
  if (rc != IDF_ATTN2_UNSUPPORTED) ++idf_stat_attn2_launches;
  return rc;
}
