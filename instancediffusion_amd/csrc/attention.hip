// attention.hip -- fused multi-head attention forward for gfx950 (flash-style online softmax), head dims 8..160.
//
// One workgroup = 4 wave64 = 128 query rows of one (batch, head); each wave owns 32 query rows.  Per KV tile of 64:
//   S^T[kv][q] = K . Q^T     v_mfma_f32_32x32x16 with A = K tile (LDS, ds_read_b128), B = Q fragment (registers)
//   "swapped" product: each lane owns ONE query column (q = lane&31) and 2x16 kv rows, so the softmax row
//   reductions are lane-local + one cross-half exchange (lane ^ 32); P never touches LDS.
//   O^T[e][q] += V^T . P^T   A = V^T tile (LDS, 2x ds_read_b64 per fragment), B = P packed to 16-bit in registers.
//   The k-ordering of the packed P fragment and of the V^T fragment is the same self-chosen permutation
//   (kv = 16*step + 4*hi + {0..3} and +8), so no lane shuffles are needed.
// K/V of the NEXT tile are prefetched global->registers while the MFMAs of the current tile run.
// Two KV segments (visual tokens + grounding tokens) implement the gated self-attention without materialising
// the concatenation (reference attention.py:307).
//
// Roofline: MFMA-bound; algorithmic flops = 4 * nq * (n0+n1) * d per (b, h).
#include "attn_core.h"
#include <cstdlib>

using namespace idfattn;

namespace {

constexpr int KVT = 64;            // kv rows per tile
constexpr int VSTR = 72;           // V^T LDS row stride in elements (144 B = 9 x 16-B slots: conflict-free b128 reads)

// NKS = ceil(d/16) K-steps of the QK^T contraction, NMT = ceil(d/32) 32-row tiles of O^T.
// MFMASUM: d < 32*NMT, i.e. the O^T tile has spare rows -> row d of the V^T image is set to ONES so the softmax
// denominator l = sum_j P[q][j] falls out of the PV MFMAs for free (no 32 VALU adds per tile per lane) and is rescaled
// together with O.  (It then sums the 16-bit-rounded P, exactly the P that multiplies V.)
#ifndef IDF_ATTN_MIN_WAVES
#define IDF_ATTN_MIN_WAVES 1
#endif
// MASK: instance-visibility bit masks (see AttnParams): the tile's 64 key words ride along in LDS; a score whose
// (query word & key word) is zero -- and which is not the query's own token -- becomes -inf before the softmax.
// RES ("resident keys"): the whole key set fits the two LDS buffers (n0 + n1 tiles <= 2: cross-attention on the 77 text
// tokens).  K / V^T are staged ONCE per workgroup, which then walks `qpw` blocks of 128 queries with the next block's Q
// fragments prefetched -- the per-block staging, LDS clears and barriers made those launches latency-bound (190 us
// against a 42 us HBM bound at 64x64).
template <int DT, int NKS, int NMT, bool MFMASUM, bool MASK = false, bool RES = false>
__global__ __launch_bounds__(256, (NKS <= 3 ? IDF_ATTN_MIN_WAVES : 1)) void attn_kernel(const AttnParams p, const int qpw) {
  constexpr int KSTR = (2 * NKS + 1) * 8;          // K LDS row stride (elements): odd number of 16-B slots
  constexpr int KCH_MAX = (KVT * 2 * NKS + 255) / 256;
  constexpr int VCH_MAX = (NMT * 32 * 8 + 255) / 256;
  constexpr int KSZ = KVT * KSTR, VSZ = NMT * 32 * VSTR;
  __shared__ __attribute__((aligned(16))) unsigned short Kl[2 * KSZ];     // double-buffered: ONE barrier per KV tile
  __shared__ __attribute__((aligned(16))) unsigned short Vl[2 * VSZ];
  __shared__ __attribute__((aligned(16))) unsigned Bl[MASK ? 2 * KVT : 4];  // key mask words of the two staged tiles
  // O staging of the coalesced epilogue (round 5): wave-private [32 queries][d].  The streaming variants reuse the K ring (dead
  // behind the last tile's barrier); with resident keys the ring stays live across the workgroup's query blocks -> own buffer.
  __shared__ __attribute__((aligned(16))) unsigned short Ol[RES ? 4 * 32 * 16 * NKS : 8];
  static_assert(2 * KSZ >= 4 * 32 * 16 * NKS, "the K ring must hold the four waves' O staging blocks");

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  // XCD-aware block order (round 3), resident-key (cross-attention) launches only: hardware block L (x fastest) runs on XCD
  // L % 8 and every XCD has its own L2; giving every XCD a CONTIGUOUS range of the logical (b, h, query block) list keeps the
  // workgroups that stage the same 77 keys on one L2 (64^2 cross-attention 183 -> 128 us at batch 64).  The streaming
  // launches (d = 80 / 160 self-attention) measured 3-5 % SLOWER with the same mapping -- their over-fetch is absorbed by the
  // Infinity Cache, and eight CUs of one XCD then pull the same K / V^T lines through one L2 channel at the same time --
  // so they keep the plain order (profiles/r03_shape_profile_B64_a.log vs r02_shape_profile_B64_fill_middle.log).
  int bx = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  if (RES) {
    const int gx = gridDim.x, gy = gridDim.y;
    const int total = gx * gy * (int)gridDim.z;
    if ((total & 7) == 0) {
      int L = bx + gx * (h + gy * b);
      L = (L & 7) * (total >> 3) + (L >> 3);
      bx = L % gx;
      const int r = L / gx;
      h = r % gy;
      b = r / gy;
    }
  }
  const int d = p.d;
  const int dch = d >> 3;                            // 16-B chunks per K row
  const int nqb = (p.nq + 127) / 128;
  const int qb0 = RES ? bx * qpw : bx;
  const int qb1 = RES ? min(nqb, qb0 + qpw) : qb0 + 1;

  // zero LDS once: pad columns of K (d..16*NKS) and pad rows of V^T (d..32*NMT) must stay finite zeros
  for (int i = tid; i < KSZ; i += 256) reinterpret_cast<unsigned*>(Kl)[i] = 0u;
  for (int i = tid; i < VSZ; i += 256) reinterpret_cast<unsigned*>(Vl)[i] = 0u;
  if (MFMASUM) {                                      // "ones" row (element d) in both ring buffers; never overwritten
    __syncthreads();
    const unsigned short one = Elem<DT>::from_f32(1.0f);
    for (int i = tid; i < 2 * KVT; i += 256) Vl[(i / KVT) * VSZ + d * VSTR + (i % KVT)] = one;
  }

  // ---- Q fragments (B operand): lane holds q = l31, e = 16*ks + 8*hi .. +7
  auto load_q = [&](int qblk, u32x4* dst) {
    const int qr = min(qblk * 128 + wave * 32 + l31, p.nq - 1);
    const unsigned short* qp = p.q + (size_t)b * p.sQ + (size_t)qr * p.ldq + h * d;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      const int e0 = ks * 16 + hi * 8;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (e0 < d) v = *reinterpret_cast<const u32x4*>(qp + e0);
      dst[ks] = v;
    }
  };
  u32x4 qf[NKS];
  load_q(qb0, qf);

  const int T0 = (p.n[0] + KVT - 1) / KVT;
  const int T1 = (p.n[1] + KVT - 1) / KVT;
  const int T = T0 + T1;

  // ---- staging roles (tile-invariant): thread -> K chunks (row, 16-B chunk) and V^T chunks (e row, 8-kv chunk)
  int k_lds[KCH_MAX], k_row[KCH_MAX], k_col[KCH_MAX];
#pragma unroll
  for (int i = 0; i < KCH_MAX; ++i) {
    const int id = tid + i * 256;
    const int row = id / dch, ch = id - row * dch;
    k_row[i] = row < KVT ? row : -1;
    k_col[i] = ch * 8;
    k_lds[i] = row * KSTR + ch * 8;
  }
  // V^T LDS image: inside each group of 16 kv the order is [0-3, 8-11, 4-7, 12-15], i.e. the k-permutation of the
  // packed P fragment, so that the PV A-fragment of lane-half `hi` is ONE 16-B ds_read_b128 at group*32 B + hi*16 B.
  // A staged 16-B chunk (8 consecutive kv, chunk index ch) therefore lands as two 8-B halves:
  //   low half -> group*16 + (ch&1)*4 elements, high half -> +8 elements.
  int v_lds[VCH_MAX], v_row[VCH_MAX];
  const int v_ch8 = (tid & 7) * 8;
  const int v_dst = ((tid & 7) >> 1) * 16 + ((tid & 7) & 1) * 4;
#pragma unroll
  for (int i = 0; i < VCH_MAX; ++i) {
    const int row = (tid + i * 256) >> 3;
    v_row[i] = row < d ? row : -1;
    v_lds[i] = row * VSTR + v_dst;
  }

  u32x4 kreg[KCH_MAX], vreg[VCH_MAX];
  unsigned breg = 0u;
  auto prefetch = [&](int t) {
    const int seg = (t < T0) ? 0 : 1;
    const int kv0 = (seg ? (t - T0) : t) * KVT;
    const int n = p.n[seg];
    if (MASK && tid < KVT) breg = p.kbits[seg][(size_t)b * p.sKb[seg] + min(kv0 + tid, n - 1)];
    const int ldk = p.ldk[seg], ldv = p.ldv[seg];
    const unsigned short* kb = p.k[seg] + (size_t)b * p.sK[seg] + h * d;
    const unsigned short* vb = p.vt[seg] + (size_t)b * p.sV[seg] + (size_t)(h * d) * ldv + kv0 + v_ch8;
#pragma unroll
    for (int i = 0; i < KCH_MAX; ++i) {
      u32x4 v = {0u, 0u, 0u, 0u};
      if (k_row[i] >= 0) {
        const int kr = min(kv0 + k_row[i], n - 1);
        v = *reinterpret_cast<const u32x4*>(kb + (size_t)kr * ldk + k_col[i]);
      }
      kreg[i] = v;
    }
    const bool tail = (n - kv0) < KVT;               // wave-uniform
#pragma unroll
    for (int i = 0; i < VCH_MAX; ++i) {
      u32x4 v = {0u, 0u, 0u, 0u};
      if (v_row[i] >= 0) {
        v = *reinterpret_cast<const u32x4*>(vb + (size_t)v_row[i] * ldv);
        if (tail) {                                   // zero kv >= n (keeps 0 * garbage out of P.V)
          const int valid = n - (kv0 + v_ch8);
#pragma unroll
          for (int w = 0; w < 4; ++w) {
            const unsigned m = (valid >= 2 * w + 2) ? 0xffffffffu : ((valid == 2 * w + 1) ? 0x0000ffffu : 0u);
            v[w] &= m;
          }
        }
      }
      vreg[i] = v;
    }
  };
  auto commit = [&](int buf) {
    unsigned short* Kb = Kl + buf * KSZ;
    unsigned short* Vb = Vl + buf * VSZ;
    if (MASK && tid < KVT) Bl[buf * KVT + tid] = breg;
#pragma unroll
    for (int i = 0; i < KCH_MAX; ++i)
      if (k_row[i] >= 0) *reinterpret_cast<u32x4*>(Kb + k_lds[i]) = kreg[i];
#pragma unroll
    for (int i = 0; i < VCH_MAX; ++i)
      if (v_row[i] >= 0) {
        u32x2 lo = {vreg[i][0], vreg[i][1]}, hi2 = {vreg[i][2], vreg[i][3]};
        *reinterpret_cast<u32x2*>(Vb + v_lds[i]) = lo;
        *reinterpret_cast<u32x2*>(Vb + v_lds[i] + 8) = hi2;
      }
  };

  const float c = p.scale_log2;

  prefetch(0);
  __syncthreads();              // zero-fill of both LDS buffers complete
  commit(0);
  if (T > 1) prefetch(1);
  if (RES && T > 1) commit(1);  // resident keys: both tiles staged once, no staging inside the tile loop
  __syncthreads();
  for (int qblk = qb0; qblk < qb1; ++qblk) {
  const int qrow = qblk * 128 + wave * 32 + l31;
  unsigned qb = 0xffffffffu;
  if (MASK) qb = p.qbits[(size_t)b * p.sQb + min(qrow, p.nq - 1)];
  u32x4 qnext[RES ? NKS : 1];
  if (RES && qblk + 1 < qb1) load_q(qblk + 1, qnext);
  f32x16 o[NMT];
#pragma unroll
  for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[mt][r] = 0.0f;
  float m_run = -INFINITY, l_run = 0.0f;
  for (int t = 0; t < T; ++t) {
    // Invariant: buffer t&1 holds tile t (visible to all waves); registers hold tile t+1 (loads in flight).
    const unsigned short* Kc = Kl + (t & 1) * KSZ;
    const unsigned short* Vc = Vl + (t & 1) * VSZ;

    const int seg = (t < T0) ? 0 : 1;
    const int kv0 = (seg ? (t - T0) : t) * KVT;
    const int nvalid = p.n[seg] - kv0;            // >= 1
    const bool two = nvalid > 32;                 // a tail of <= 32 keys (the 13 of the 77 text tokens): only the first half-tile

    // ---- S^T = K Q^T  (2 tiles of 32 kv rows)
    f32x16 s[2];
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      if (st == 1 && !two) break;
      const unsigned short* kf = Kc + (st * 32 + l31) * KSTR + hi * 8;
      {
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        u32x4 a = *reinterpret_cast<const u32x4*>(kf);
        s[st] = Elem<DT>::mfma32(a, qf[0], zero);
      }
#pragma unroll
      for (int ks = 1; ks < NKS; ++ks) {
        u32x4 a = *reinterpret_cast<const u32x4*>(kf + ks * 16);
        s[st] = Elem<DT>::mfma32(a, qf[ks], s[st]);
      }
    }
    // ---- mask the tail, running max.  s[st][r]: kv = kv0 + st*32 + (r&3) + 8*(r>>2) + 4*hi
    if (nvalid < KVT) {
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        if (st == 1 && !two) break;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int kv = st * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          s[st][r] = (kv >= nvalid) ? -INFINITY : s[st][r];
        }
      }
    }
    if (MASK) {
      const unsigned* bw = Bl + (t & 1) * KVT;
      const int self_kv = (seg == 0) ? (qrow - kv0) : -1;          // tile-local index of the query's own token
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        if (st == 1 && !two) break;
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          const int kvb = st * 32 + 8 * q4 + 4 * hi;
          const u32x4 kw = *reinterpret_cast<const u32x4*>(bw + kvb);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const bool ok = ((qb & kw[e]) != 0u) | (kvb + e == self_kv);
            s[st][4 * q4 + e] = ok ? s[st][4 * q4 + e] : -INFINITY;
          }
        }
      }
    }
    float mx = s[0][0];
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      if (st == 1 && !two) break;
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[st][r]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float m_new = fmaxf(m_run, mx * c);            // c > 0
    // MASK: every key seen so far may be masked for this query (m_new = -inf): keep exp2 arguments finite
    const float m_use = (MASK && m_new == -INFINITY) ? 0.0f : m_new;
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_use);      // first tile: exp2(-inf) = 0
    m_run = m_new;
    m_new = m_use;
    float rs = 0.0f;
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      if (st == 1 && !two) break;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pv = __builtin_amdgcn_exp2f(fmaf(s[st][r], c, -m_new));
        s[st][r] = pv;
        if (!MFMASUM) rs += pv;
      }
    }
    if (!MFMASUM) l_run = l_run * alpha + rs;
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[mt][r] *= alpha;

    // ---- O^T += V^T P^T.  K-step (st, k2): P regs 8*k2..8*k2+7 of tile st  <->  kv = st*32 + 16*k2 + 4*hi + {0..3, 8..11}
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      if (st == 1 && !two) break;
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2) {
        u32x4 pf;
#pragma unroll
        for (int w = 0; w < 4; ++w) pf[w] = pack2<DT>(s[st][8 * k2 + 2 * w], s[st][8 * k2 + 2 * w + 1]);
        const int kvoff = st * 32 + 16 * k2 + 8 * hi;      // permuted image: lane-half hi owns 8 contiguous elements
#pragma unroll
        for (int mt = 0; mt < NMT; ++mt) {
          const u32x4 a = *reinterpret_cast<const u32x4*>(Vc + (mt * 32 + l31) * VSTR + kvoff);
          o[mt] = Elem<DT>::mfma32(a, pf, o[mt]);
        }
      }
    }
    // stage the next tile into the OTHER buffer (last read in iteration t-1; every wave passed that barrier), then
    // fetch tile t+2 into the registers; one barrier per tile.
    if (!RES) {
      if (t + 1 < T) {
        commit((t + 1) & 1);
        if (t + 2 < T) prefetch(t + 2);
      }
      __syncthreads();
    }
  }

  // ---- normalise and store.  o[mt][r]: e = mt*32 + (r&3) + 8*(r>>2) + 4*hi, q = l31
  float l_tot;
  if (MFMASUM) {
    // row e = d of O^T: tile d/32, register 4*((d%32)/8) of the hi = 0 lanes (d % 8 == 0)
    const int sel = (d & 31) >> 3;
    const f32x16& ol = o[NMT - 1];
    float lv = sel == 0 ? ol[0] : (sel == 1 ? ol[4] : (sel == 2 ? ol[8] : ol[12]));
    l_tot = __shfl(lv, l31, 64);                       // broadcast from lane l31 (hi = 0) to its hi = 1 partner
  } else {
    l_tot = l_run + __shfl_xor(l_run, 32, 64);
  }
  const float inv = 1.0f / l_tot;
  // A lane owns ONE query row in 8-byte pieces: stored directly that is up to 20 8-B stores per lane at a 2*ldo-byte lane stride --
  // 64 separate requests per store instruction, every piece a partial sector: the cross-attention launches (a few MFMAs per query
  // block) were bound by exactly this store tail (round 5; the d = 40 / 80 / 160 LDS-DMA kernels already store this way).  Now the
  // wave transposes its 32 x d block through LDS and writes 16 B per lane, consecutive lanes on consecutive chunks of a row.
  // Needs 16-B-aligned output rows; else (ldo % 8 == 4, legal for idf_attention) the direct 8-B stores.
  const bool wide = ((p.ldo & 7) == 0) && ((p.sO & 7) == 0) && ((reinterpret_cast<uintptr_t>(p.out) & 15u) == 0);
  if (wide) {
    // (streaming variants: every tile, the last one included, ends with the workgroup barrier -- the K ring is dead here)
    unsigned short* const ow = (RES ? Ol : Kl) + wave * (32 * 16 * NKS);
    unsigned short* orow = ow + l31 * d;
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const int e = mt * 32 + 8 * qd + 4 * hi;
        if (e < d) {
          u32x2 pk = {pack2<DT>(o[mt][4 * qd] * inv, o[mt][4 * qd + 1] * inv),
                      pack2<DT>(o[mt][4 * qd + 2] * inv, o[mt][4 * qd + 3] * inv)};
          *reinterpret_cast<u32x2*>(orow + e) = pk;
        }
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    const int q0 = qblk * 128 + wave * 32;
    unsigned short* const obase = p.out + (size_t)b * p.sO + h * d;
    const int nch = 4 * d;                            // 16-B chunks of the 32 x d block (dch per row)
    for (int c = lane; c < nch; c += 64) {
      const int row = c / dch, col = c - row * dch;
      const u32x4 v = *reinterpret_cast<const u32x4*>(ow + c * 8);
      if (q0 + row < p.nq) *reinterpret_cast<u32x4*>(obase + (size_t)(q0 + row) * p.ldo + col * 8) = v;
    }
    if (RES) {                                        // the staging block is rewritten by this wave's next query block
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
    }
  } else if (qrow < p.nq) {
    unsigned short* op = p.out + (size_t)b * p.sO + (size_t)qrow * p.ldo + h * d;
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const int e = mt * 32 + 8 * qd + 4 * hi;
        if (e < d) {   // d % 8 == 0 and e % 4 == 0 -> the 4 columns are all valid
          u32x2 pk = {pack2<DT>(o[mt][4 * qd] * inv, o[mt][4 * qd + 1] * inv),
                      pack2<DT>(o[mt][4 * qd + 2] * inv, o[mt][4 * qd + 3] * inv)};
          *reinterpret_cast<u32x2*>(op + e) = pk;
        }
      }
  }
  if (RES) {
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) qf[ks] = qnext[ks];
  }
  }   // query blocks
}

template <int DT>
int launch_attn(const AttnParams& p, int B, hipStream_t s) {
  const int nqb = (p.nq + 127) / 128;
  dim3 grid(nqb, p.H, B), block(256);
  const int nks = (p.d + 15) / 16, nmt = (p.d + 31) / 32;
  // resident-key path: <= 2 key tiles in total, no mask; enough query blocks per workgroup to amortise the staging while
  // the grid still covers the chip ~4x
  const int tiles = (p.n[0] + KVT - 1) / KVT + (p.n[1] + KVT - 1) / KVT;
  int qpw = 1;
  const bool res = !p.qbits && tiles <= 2 && nqb >= 2;
  if (res) {
    const long long blocks = (long long)nqb * p.H * B;
    qpw = (int)(blocks / 1024);
    qpw = qpw < 1 ? 1 : (qpw > 8 ? 8 : qpw);
  }
  dim3 grid_res((nqb + qpw - 1) / qpw, p.H, B);
#define IDF_ATTN_CASE(KS, MT) \
  if (nks == KS && nmt == MT) { \
    if (p.qbits) { \
      if (p.d < 32 * MT) hipLaunchKernelGGL((attn_kernel<DT, KS, MT, true, true>), grid, block, 0, s, p, 1); \
      else hipLaunchKernelGGL((attn_kernel<DT, KS, MT, false, true>), grid, block, 0, s, p, 1); \
    } else if (res && qpw > 1) { \
      if (p.d < 32 * MT) hipLaunchKernelGGL((attn_kernel<DT, KS, MT, true, false, true>), grid_res, block, 0, s, p, qpw); \
      else hipLaunchKernelGGL((attn_kernel<DT, KS, MT, false, false, true>), grid_res, block, 0, s, p, qpw); \
    } else { \
      if (p.d < 32 * MT) hipLaunchKernelGGL((attn_kernel<DT, KS, MT, true>), grid, block, 0, s, p, 1); \
      else hipLaunchKernelGGL((attn_kernel<DT, KS, MT, false>), grid, block, 0, s, p, 1); \
    } \
    return idf_launch_status(); }
  IDF_ATTN_CASE(1, 1)    // d = 8, 16
  IDF_ATTN_CASE(2, 1)    // d = 24, 32
  IDF_ATTN_CASE(3, 2)    // d = 40, 48
  IDF_ATTN_CASE(4, 2)    // d = 56, 64
  IDF_ATTN_CASE(5, 3)    // d = 72, 80
  IDF_ATTN_CASE(6, 3)    // d = 88, 96
  IDF_ATTN_CASE(8, 4)    // d = 120, 128
  IDF_ATTN_CASE(10, 5)   // d = 152, 160
#undef IDF_ATTN_CASE
  return IDF_E_UNSUPPORTED;
}

}  // namespace

std::atomic<long long> idf_stat_attn2_launches{0};

int g_attn2_mode = -2;
int idf_attn2_mode() {
  if (g_attn2_mode == -2) {
    // the environment may hold a value of an older ABI (modes 4..14 named kernels that left the library): out of range = default,
    // the same range idf_set_tuning accepts
    const char* e = getenv("IDF_ATTN2");
    const int v = e ? atoi(e) : IDF_ATTN2_DEFAULT;
    g_attn2_mode = (v < 0 || v > 6) ? IDF_ATTN2_DEFAULT : v;
  }
  return g_attn2_mode;
}
int idf_attn2_set_mode(int v) { const int prev = idf_attn2_mode(); g_attn2_mode = v; return prev; }

extern "C" int idf_attention(const idf_attn_args* a, void* stream) {
  if (!a || !a->q || !a->k0 || !a->vt0 || !a->out) return IDF_E_ARG;
  if (a->B <= 0 || a->H <= 0 || a->d <= 0 || (a->d % 8) || a->d > 160 || a->nq <= 0 || a->n0 <= 0 || a->n1 < 0) return IDF_E_ARG;
  if (a->n1 > 0 && (!a->k1 || !a->vt1)) return IDF_E_ARG;
  if ((a->ldq % 8) || (a->ldk0 % 8) || (a->ldv0 % 8) || (a->ldo % 4)) return IDF_E_ALIGN;
  if (a->n1 > 0 && ((a->ldk1 % 8) || (a->ldv1 % 8))) return IDF_E_ALIGN;
  // every KV tile reads 64 columns of V^T: the row length must cover the rounded-up segment
  if (a->ldv0 < ((a->n0 + 63) / 64) * 64) return IDF_E_ARG;
  if (a->n1 > 0 && a->ldv1 < ((a->n1 + 63) / 64) * 64) return IDF_E_ARG;
  if (!aligned16(a->q) || !aligned16(a->k0) || !aligned16(a->vt0)) return IDF_E_ALIGN;
  AttnParams p{};
  p.q = (const unsigned short*)a->q; p.ldq = a->ldq; p.sQ = a->strideQ; p.nq = a->nq;
  p.k[0] = (const unsigned short*)a->k0; p.ldk[0] = a->ldk0; p.sK[0] = a->strideK0;
  p.vt[0] = (const unsigned short*)a->vt0; p.ldv[0] = a->ldv0; p.sV[0] = a->strideV0; p.n[0] = a->n0;
  p.k[1] = (const unsigned short*)(a->n1 > 0 ? a->k1 : a->k0); p.ldk[1] = a->n1 > 0 ? a->ldk1 : a->ldk0; p.sK[1] = a->strideK1;
  p.vt[1] = (const unsigned short*)(a->n1 > 0 ? a->vt1 : a->vt0); p.ldv[1] = a->n1 > 0 ? a->ldv1 : a->ldv0; p.sV[1] = a->strideV1;
  p.n[1] = a->n1;
  p.out = (unsigned short*)a->out; p.ldo = a->ldo; p.sO = a->strideO;
  p.H = a->H; p.d = a->d;
  p.scale_log2 = a->scale * 1.4426950408889634f;
  if (a->qbits) {                                   // instance-visibility mask: 32-queries-per-wave kernel only
    if (!a->kbits0 || (a->n1 > 0 && !a->kbits1)) return IDF_E_ARG;
    if ((((uintptr_t)a->qbits) | ((uintptr_t)a->kbits0) | ((uintptr_t)a->kbits1)) & 3u) return IDF_E_ALIGN;
    p.qbits = (const unsigned*)a->qbits; p.sQb = a->strideQb;
    p.kbits[0] = (const unsigned*)a->kbits0; p.sKb[0] = a->strideKb0;
    p.kbits[1] = (const unsigned*)(a->n1 > 0 ? a->kbits1 : a->kbits0); p.sKb[1] = a->n1 > 0 ? a->strideKb1 : a->strideKb0;
  }
  hipStream_t s = (hipStream_t)stream;
  if (!a->qbits && idf_attn2_mode() >= 4) {
    const int rc = idf_launch_attn4w(p, a->B, a->dtype, idf_attn2_mode(), s);
    if (rc != IDF_ATTN2_UNSUPPORTED) { ++idf_stat_attn2_launches; return rc; }
  }
  if (!a->qbits && idf_attn2_mode() > 0) {
    const int rc = idf_launch_attn4(p, a->B, a->dtype, s);
    if (rc != IDF_ATTN2_UNSUPPORTED) { ++idf_stat_attn2_launches; return rc; }
  }
  if (!a->qbits) {
    const int rc = idf_launch_attn8(p, a->B, a->dtype, s);
    if (rc != IDF_ATTN2_UNSUPPORTED) { ++idf_stat_attn8_launches; return rc; }
  }
  if (a->dtype == IDF_BF16) return launch_attn<IDF_BF16>(p, a->B, s);
  if (a->dtype == IDF_F16) return launch_attn<IDF_F16>(p, a->B, s);
  return IDF_E_UNSUPPORTED;
}
