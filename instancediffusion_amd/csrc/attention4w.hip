// attention4w.hip -- attention forward, d = 40, ONE WAVE PER SIMD (gfx950): the 64x64-latent self / gated self-attention
// (reference ldm/modules/attention.py:257-267 self-attention, :304-311 GatedSelfAttentionDense).
//
// Same data layout, MFMA operand roles and numerics contract as attention4.hip (swapped K.Q^T so a lane owns a query column,
// permuted K fragment rows so the packed P feeds P.V directly, softmax denominator from an all-ones V^T row, the reference
// value m of a query subtracted BY THE MFMA through the spare half K-step of the d = 40 -> 48 padding, K / V^T tiles by
// inline-assembly LDS-DMA).  What is different, and why: attention4.hip runs two waves per SIMD (256 registers each), and every
// schedule tried on it in rounds 2-5 lands where a wave's MFMA issue (896 cycles per 64 x 64 tile) and its VALU issue (~690)
// ADD.  Here a workgroup is 4 waves -- one per SIMD, the whole 512-register file each -- and a wave owns 128 queries in four
// groups of 32, so that
//   * every K / V^T fragment read from LDS feeds FOUR MFMAs instead of two (half the LDS read traffic per flop);
//   * the softmax of one query group (16 v_exp + 8 v_cvt_pk per 32 x 32 score block) is issued BETWEEN the MFMAs of another
//     group inside one instruction stream: per 32-key block and group a step of 7 MFMAs {P.V of group g (4), K.Q^T of the NEXT
//     block for group g (3)} carries the exponentials of group g+1 -- 3-4 single-issue instructions per 32-cycle MFMA gap,
//     under the <= 5 that MI355X_MICROARCH.md prices as hidden for a one-wave-per-SIMD stream;
//   * no per-tile overflow guard: a P may grow to the storage type's largest finite value, O and the denominator are fp32;
//     the block checks ONCE, in its epilogue, that every denominator and output is finite and otherwise reruns with the classic
//     per-tile running max (the `exact` pass; also how tile 0 fixes m).
// The score block of a group is overwritten in place by the next block's K.Q^T (its exponentials were packed one step
// earlier), so the four groups need 64 score registers, not 128; K fragments are read two blocks ahead, V^T fragments one.
// K and V^T are both fetched TWO tiles ahead into 3-stage rings: one s_barrier per 64-key tile, nothing read from LDS right
// behind it.
#include "attn_core.h"
#include <cstdlib>
#include <atomic>
#include <type_traits>
#include <utility>

using namespace idfattn;

namespace {

__device__ __attribute__((aligned(128))) unsigned short idf_attn4w_zero_page[64];
__device__ __attribute__((aligned(16))) unsigned short idf_attn4w_ones_page[2][8] = {
    {0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80},      // bf16 1.0
    {0x3c00, 0x3c00, 0x3c00, 0x3c00, 0x3c00, 0x3c00, 0x3c00, 0x3c00}};     // fp16 1.0

constexpr int KVT = 64;        // keys per tile
constexpr int D = 40, DCH = 5, NKS = 3, NMT = 2;
constexpr int NW = 4;          // waves per workgroup
constexpr int KSTG = 4096;     // K stage stride (elements): 64 rows x 40 (5120 B), ones fragments at bytes 5120 and 7680
constexpr int VSZ = 64 * KVT;  // V^T stage (elements): 64 rows of 128 B, 16-B slot ^= (row >> 1) & 7
constexpr int NST = 3;         // ring stages (K and V^T)

__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(size_t)p; }
// LDS-DMA through inline asm (see attention4.hip): ordering is ours -- s_waitcnt vmcnt(0) + s_barrier at the end of every tile.
__device__ __forceinline__ void dma16_sv(const void* sbase /* wave-uniform */, unsigned voff, unsigned lds) {
  lds = __builtin_amdgcn_readfirstlane(lds);
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds), "v"(voff), "s"(sbase) : "memory");
}
__device__ __forceinline__ void dma16_v(const void* addr /* per lane */, unsigned lds) {
  lds = __builtin_amdgcn_readfirstlane(lds);
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds), "v"(addr) : "memory");
}


// ---- asm-owned accumulation registers.  The O^T accumulators (a[0:127]: tile (g, mt) at 16 (2 g + mt)) and the Q fragments
// (a[128:175]: (g, ks) at 128 + 4 (3 g + ks)) live in AGPRs NAMED in the asm text: the register allocator never sees them, so
// it cannot split, copy or spill them (left to it, `"+a"` operands were copied AGPR -> AGPR around every MFMA and the Q
// fragments reloaded from scratch inside the stream).  The compiler's own values of the common pass are all VGPRs; the
// listing is checked for compiler-generated v_accvgpr_* in that region (there must be none: it knows these registers only as
// clobbers).  The kernel descriptor's AGPR count comes from the clobber lists.
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

constexpr int OA_BASE = 0;      // Q fragments behind the O^T tiles: QA_BASE = 32 NG
template <int R> __device__ __forceinline__ void agpr_write(unsigned v) {
  asm volatile("v_accvgpr_write_b32 a%c1, %0" ::"v"(v), "n"(R));
}
template <int R> __device__ __forceinline__ unsigned agpr_read() {
  unsigned v;
  asm volatile("v_accvgpr_read_b32 %0, a%c1" : "=v"(v) : "n"(R));
  return v;
}
// O^T(g, mt) += V^T fragment (asm-owned AGPR slot VS of 4 registers, behind the Q fragments) . b (packed P, VGPRs)
template <int DT, int NG, int G, int MT, int VS> __device__ __forceinline__ void mf_o(const u32x4& b) {
  constexpr int lo = OA_BASE + 16 * (2 * G + MT), va = 44 * NG + 4 * VS;
  if constexpr (DT == IDF_BF16) asm volatile("v_mfma_f32_32x32x16_bf16 a[%c1:%c2], a[%c3:%c4], %0, a[%c1:%c2]" ::"v"(b), "n"(lo), "n"(lo + 15), "n"(va), "n"(va + 3));
  else asm volatile("v_mfma_f32_32x32x16_f16 a[%c1:%c2], a[%c3:%c4], %0, a[%c1:%c2]" ::"v"(b), "n"(lo), "n"(lo + 15), "n"(va), "n"(va + 3));
}
// V^T fragment slot VS <- 16 bytes per lane of LDS
template <int NG, int VS, int OFF> __device__ __forceinline__ void ds_read128_vslot(unsigned addr) {
  constexpr int va = 44 * NG + 4 * VS;
  asm volatile("ds_read_b128 a[%c1:%c2], %0 offset:%c3" ::"v"(addr), "n"(va), "n"(va + 3), "n"(OFF));
}
// S^T = k . Q(g, ks) (FIRST: C = 0, else accumulate); the scores stay in VGPRs
template <int DT, int NG, int G, int KS, bool FIRST> __device__ __forceinline__ void mf_s(f32x16& acc, const u32x4& k) {
  constexpr int lo = 32 * NG + 4 * (3 * G + KS);
  if constexpr (FIRST) {
    if constexpr (DT == IDF_BF16) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[%c2:%c3], 0" : "=&v"(acc) : "v"(k), "n"(lo), "n"(lo + 3));
    else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, a[%c2:%c3], 0" : "=&v"(acc) : "v"(k), "n"(lo), "n"(lo + 3));
  } else {
    if constexpr (DT == IDF_BF16) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[%c2:%c3], %0" : "+v"(acc) : "v"(k), "n"(lo), "n"(lo + 3));
    else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, a[%c2:%c3], %0" : "+v"(acc) : "v"(k), "n"(lo), "n"(lo + 3));
  }
}

// ---- the stream's VALU / LDS instructions as inline assembly too.  Pure operations (exp2, the conversions, LDS reads) carry no
// chain in the compiler's instruction DAG: fenced by sched_barrier or not, they are linearised in front of the first MFMA of
// the block (the first build of this kernel came out as 24 VALU instructions followed by 7 back-to-back MFMAs per step).
// `asm volatile` statements keep their program order, so the source order below IS the issue order.  What the compiler then
// cannot do for us: the s_waitcnt lgkmcnt in front of the first use of a fragment read (one per block, below), the TRANS ->
// VALU forwarding wait state of gfx940+ (a v_cvt_pk never directly follows the v_exp that feeds it), MFMA -> VALU (above).
__device__ __forceinline__ float exp2_asm(float x) {
  float y;
  asm volatile("v_exp_f32_e32 %0, %1" : "=v"(y) : "v"(x));
  return y;
}
template <int DT> __device__ __forceinline__ unsigned cvt_pk_asm(float lo, float hi) {
  unsigned r;
  if constexpr (DT == IDF_BF16) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  else asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}
template <int OFF> __device__ __forceinline__ u32x4 ds_read128_asm(unsigned addr) {
  u32x4 r;
  asm volatile("ds_read_b128 %0, %1 offset:%c2" : "=v"(r) : "v"(addr), "n"(OFF));
  return r;
}

// Optional cycle trace (a second build with -DIDF_ATTN4W_TRACE, read through idf_attn4w_trace_read by tools/ubench/attn_harness.hip;
// the shipped library has none of it): s_memtime deltas of the four waves of the middle workgroup, summed over the tiles of the
// stream: [0] first block (with the DMA issue), [1] second block, [2] vmcnt wait + barrier, [3] tiles.
#ifdef IDF_ATTN4W_TRACE
__device__ unsigned long long idf_attn4w_trace_buf[4][8];
#define TR_DECL unsigned long long tr_last = 0, tr_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define TR_START tr_last = __builtin_readcyclecounter();
#define TR(i) { const unsigned long long tr_now = __builtin_readcyclecounter(); tr_acc[i] += tr_now - tr_last; tr_last = tr_now; }
#define TR_DUMP { if ((int)blockIdx.x == (int)gridDim.x / 2 && lane == 0) { for (int i = 0; i < 8; ++i) idf_attn4w_trace_buf[wave][i] = tr_acc[i]; } }
#define TR_MARK(i) { const unsigned long long tr_now = __builtin_readcyclecounter(); tr_acc[i] += tr_now - tr_blk; tr_blk = tr_now; }
#else
#define TR_DECL
#define TR_START
#define TR(i) {}
#define TR_DUMP {}
#define TR_MARK(i) {}
#endif

template <int DT> struct RefShiftW { static constexpr float v = 7.0f; };   // after tile 0 the largest P of a query is 2^-7

// NG = query groups (of 32) per wave.  4: one wave per SIMD (345 registers); 2: two waves per SIMD (two 4-wave workgroups per
// CU, <= 256 registers) -- the same stream with 2 steps per block: a wave that stalls (LDS-DMA issue, barrier, prologue /
// epilogue of its workgroup) is covered by the other wave of its SIMD.
// PERSIST: the workgroup walks query blocks blockIdx.x, + gridDim.x, ... (one workgroup per CU) and, in the last two tiles of a
// block, fetches the NEXT block's first two K / V^T tiles into the ring stages that fall free (the ring simply keeps rotating
// across blocks) and its Q rows into registers: with one wave per SIMD nothing else hides a block's prologue -- global-memory
// latency of Q and of the first tiles, workgroup launch -- which was 16 % of the kernel's time (profiles/NOTES_r06.md).  The
// epilogue's staging block then has its own LDS region (the ring stays live).
template <int DT, int NG, bool PERSIST>
__global__ __launch_bounds__(NW * 64, NG == 4 ? 1 : 2) void attn4w_kernel(const AttnParams p, const int nqb, const int xcd_order, const int total) {
  constexpr int QB = NW * NG * 32;                 // queries per workgroup
  constexpr int QA_BASE = 32 * NG;                 // first AGPR of the Q fragments
  // the asm-owned AGPR block a[0 : 44 NG + 31] (O^T tiles, Q fragments, the two V^T fragment sets of 16 registers); this (empty) statement is where the kernel descriptor learns about it
  if constexpr (NG == 4) asm volatile("" ::: "a0", "a207"); else asm volatile("" ::: "a0", "a119");
  constexpr int RING = NST * KSTG + NST * VSZ, OSTAGE = NW * NG * 32 * D;
  constexpr int OW_OFF = PERSIST ? RING : 0;       // epilogue staging block: behind the rings, or aliasing them
  constexpr int LDS_ELEMS = PERSIST ? RING + OSTAGE : (RING > OSTAGE ? RING : OSTAGE);
  extern __shared__ __attribute__((aligned(128))) unsigned short smem[];
  int* const redo_flag = reinterpret_cast<int*>(smem + LDS_ELEMS);
  unsigned short* const Ks = smem;
  unsigned short* const Vs = smem + NST * KSTG;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;

  // hardware block L runs on XCD L % 8; every XCD gets a contiguous range of logical blocks (the query blocks of one (batch,
  // head) are consecutive: its K / V^T stay in that XCD's L2).  With PERSIST the stride gridDim.x is a multiple of 8.
  auto locate = [&](int blk, int& qb_, int& h_, int& b_) {
    int L = blk;
    if ((xcd_order & 1) && (total & 7) == 0) L = (L & 7) * (total >> 3) + (L >> 3);
    qb_ = L % nqb;
    h_ = (L / nqb) % p.H;
    b_ = L / (nqb * p.H);
  };

  // Pad rows of the V^T stages (rows D .. 63 of every O^T tile must be finite: row D = ones, the others zero) and the ones
  // fragments of the K stages.  Rows 0 .. D-1 are rewritten by every tile's LDS-DMA (tail tiles fetch their invalid chunks from
  // a zero page), so the DMA of the first tiles may already be in flight while this runs.
  auto init_lds = [&]() {
    const unsigned short one = Elem<DT>::from_f32(1.0f);
    for (int i = tid; i < NST * (64 - D) * KVT / 2; i += NW * 64) {
      const int st = i / ((64 - D) * KVT / 2), r = i % ((64 - D) * KVT / 2);
      const unsigned v = r < KVT / 2 ? ((unsigned)one | ((unsigned)one << 16)) : 0u;      // row D: ones
      reinterpret_cast<unsigned*>(Vs + st * VSZ + D * KVT)[r] = v;
    }
    if (tid < NST * 2 * 8) {                       // {1, 0, .., 0} at elements 2560 and 3840 of every K stage
      const int st = tid >> 4, which = (tid >> 3) & 1, e = tid & 7;
      Ks[st * KSTG + 2560 + which * 1280 + e] = e == 0 ? one : (unsigned short)0;
    }
  };
  if (tid == 0) *redo_flag = 0;

  const int T0 = (p.n[0] + KVT - 1) / KVT;
  const int T1 = (p.n[1] + KVT - 1) / KVT;
  const int T = T0 + T1;

  // ---- DMA roles (as attention4.hip): K instruction i (linear 16-B chunks 64 i .. 64 i + 63 of the tile) by wave i % 4,
  // V^T instruction i (rows 8 i .. 8 i + 7) by wave (i + 1) % 4: 3 / 3 / 2 / 2 per wave and tile.  `so` = byte offset of the
  // target stage inside either ring (0, 8192, 16384: the stages rotate, across blocks too under PERSIST).
  constexpr int K_INST = DCH, V_INST = D / 8;
  constexpr int K_PER_WAVE = (K_INST + NW - 1) / NW, V_PER_WAVE = (V_INST + NW - 1) / NW;
  const int vwave = (wave + NW - 1) % NW;
  auto cold_lane = [&]() { int l = lane; asm volatile("" : "+v"(l)); return l; };
  unsigned ones_mask = 0u;                           // bit s: stage s holds a tail-restricted ones row (wave-uniform)

  auto issue_k = [&](int t, int bb, int hh, unsigned so) {
    const int ln = cold_lane();
    const int seg = (t < T0) ? 0 : 1;
    const int kv0 = (seg ? (t - T0) : t) * KVT;
    const int n = p.n[seg];
    const int ldk = p.ldk[seg];
    const char* kb = reinterpret_cast<const char*>(p.k[seg] + (size_t)bb * p.sK[seg] + hh * D);
    unsigned short* dst = Ks + (so >> 1);
    const bool full = kv0 + KVT <= n;
#pragma unroll
    for (int j = 0; j < K_PER_WAVE; ++j)
      if (wave + NW * j < K_INST) {
        const int c = (wave + NW * j) * 64 + ln;
        const int row = c / DCH, col = (c - row * DCH) * 8;
        const int kr = full ? kv0 + row : min(kv0 + row, n - 1);     // tail: clamped duplicates of the last valid key
        dma16_v(kb + ((size_t)kr * ldk + col) * 2, lds_addr(dst + (wave + NW * j) * 512));
      }
  };
  auto issue_ones = [&](unsigned so, int nvalid) {
    if (wave == NW / 2) {
      const int ln = cold_lane();
      const int row = D + (ln >> 3);
      const int chunk = (ln & 7) ^ ((row >> 1) & 7);
      const bool one = (row == D) && (chunk * 8 < nvalid);
      const unsigned short* src = one ? idf_attn4w_ones_page[DT == IDF_BF16 ? 0 : 1] : idf_attn4w_zero_page + (ln & 7) * 8;
      dma16_v(src, lds_addr(Vs + (so >> 1) + V_INST * 512));
    }
  };
  auto issue_v = [&](int t, int bb, int hh, unsigned so) {
    const int ln = cold_lane();
    const int seg = (t < T0) ? 0 : 1;
    const int kv0 = (seg ? (t - T0) : t) * KVT;
    const int n = p.n[seg];
    const int ldv = p.ldv[seg];
    const char* vb = reinterpret_cast<const char*>(p.vt[seg] + (size_t)bb * p.sV[seg] + (size_t)(hh * D) * ldv);
    unsigned short* dst = Vs + (so >> 1);
    const char* base = vb + (size_t)kv0 * 2;
#pragma unroll
    for (int j = 0; j < V_PER_WAVE; ++j)
      if (vwave + NW * j < V_INST) {
        const int row = (vwave + NW * j) * 8 + (ln >> 3);
        const int chunk = (ln & 7) ^ ((row >> 1) & 7);
        const bool valid = (kv0 + chunk * 8) < n;                    // tail: 8-key chunks beyond n come from the zero page
        const char* src = valid ? base + ((size_t)row * ldv + chunk * 8) * 2
                                : reinterpret_cast<const char*>(idf_attn4w_zero_page + (ln & 7) * 8);
        dma16_v(src, lds_addr(dst + (vwave + NW * j) * 512));
      }
    // the ones row of this stage: restrict it for a tail tile, restore it when the stage last held a tail tile
    const bool tail = (kv0 + KVT > n);
    const unsigned bit = 1u << (so >> 13);
    if (tail || (ones_mask & bit)) issue_ones(so, tail ? n - kv0 : KVT);
    ones_mask = tail ? (ones_mask | bit) : (ones_mask & ~bit);
  };

  // K fragment row permutation (attention4.hip): MFMA row i of a 32-key half carries key (i with bits 2 and 3 swapped)
  const int kperm = (l31 & 0x13) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);
  // per-lane LDS byte offsets inside a stage.  K: K-steps 0 / 1 at kA + 32 ks, the last K-step at kB (hi = 0: elements 32..39
  // of the row; hi = 1: the constant {1, 0, .., 0} behind the rows; + 2560 bytes for the second 32-key half in both cases)
  const unsigned kA = (unsigned)(kperm * D + hi * 8) * 2u;
  const unsigned kB = hi ? 5120u : (unsigned)(kperm * D + 32) * 2u;
  // V^T: row mt*32 + l31 (+ 4096 bytes per mt), 8-key chunk (2 step + hi) ^ ((l31 >> 1) & 7)
  unsigned vA[4];
#pragma unroll
  for (int step = 0; step < 4; ++step) vA[step] = (unsigned)(l31 * KVT + (((step * 2 + hi) ^ ((l31 >> 1) & 7)) * 8)) * 2u;
  constexpr unsigned vs_base = (unsigned)(NST * KSTG * 2);      // byte offset of the V^T ring inside smem
  static_assert(KSTG == VSZ && KSTG * 2 == 8192, "one 8-KB stage stride for both rings");

  auto lds128 = [&](unsigned addr) -> u32x4 { return *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(smem) + addr); };
  auto load_kf = [&](u32x4 (&dst)[NKS], unsigned stage_off /* bytes */, int half) {
    dst[0] = lds128(stage_off + kA + half * 2560);
    dst[1] = lds128(stage_off + kA + half * 2560 + 32);
    dst[2] = lds128(stage_off + kB + half * 2560);
  };
  auto load_vf = [&](u32x4 (&dst)[NMT][2], unsigned stage_off, int half) {
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
      for (int st = 0; st < 2; ++st) dst[mt][st] = lds128(vs_base + stage_off + vA[half * 2 + st] + mt * 4096);
  };
  auto half_max = [&](float mx) -> float {
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
    return fmaxf(mx, __uint_as_float(hi ? sw[0] : sw[1]));
  };
  // raw Q fragment (B operand) of one query group and K-step: lane holds q = l31, e = 16*ks + 8*hi .. +7 (zero beyond D)
  auto load_q_raw = [&](const int g, const int ks, int qbb, int hh, int bb) -> u32x4 {
    const int qrow = qbb * QB + wave * (NG * 32) + g * 32 + l31;
    const int qr = min(qrow, p.nq - 1);
    const unsigned short* qp = p.q + (size_t)bb * p.sQ + (size_t)qr * p.ldq + hh * D;
    const int e0 = ks * 16 + hi * 8;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (e0 < D) v = *reinterpret_cast<const u32x4*>(qp + e0);
    return v;
  };
  // ... pre-multiplied by scale*log2(e).  Element D (first element of the hi = 1 half of the last K-step) carries -m.
  auto scale_q = [&](u32x4 v) -> u32x4 {
    float f[8];
    unpack8<DT>(v, f);
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] *= p.scale_log2;
    return pack8<DT>(f);
  };
  auto end_tile = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  };
  unsigned short* const ow = smem + OW_OFF + wave * (NG * 32 * D);      // epilogue: wave-private [32 NG queries][D]

  // ================================================================================================================
  // The common pass.  Every MFMA is inline assembly: scores in VGPRs (the exponentials read them), O^T, Q and the V^T fragments
  // in asm-owned AGPRs (only MFMAs and LDS reads touch them), and in the stream the VALU / LDS instructions are asm too, so the
  // source order is the issue order.  Hazards the compiler cannot see behind the asm (an MFMA result read by a VALU instruction
  // needs ~12 wait states): in the stream a score block is read one whole step (7 MFMAs) or more after its last MFMA; elsewhere
  // an explicit s_nop 15 follows.
  // ================================================================================================================
#define FENCE __builtin_amdgcn_sched_barrier(0);
#define MFMA_DRAIN asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");
#define MFMA_DRAIN2(x, y) asm volatile("s_nop 15\n\ts_nop 3" : "+v"(x), "+v"(y)::"memory");   // (the VALU readers of x, y now depend on it)
  f32x16 s[NG];
  u32x4 pkb[2][2];                                   // [group parity][16-key step]
  u32x4 kfr[2][NKS];                                 // (the V^T fragment sets are AGPR slots: set b, tile mt, step st -> slot (2 b + mt) 2 + st)
#define VSLOT(b, mt, st) (((b) * 2 + (mt)) * 2 + (st))
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  const int ndma = wave < 2 ? 3 : 2;                 // wave-uniform
  const unsigned smem_lds = lds_addr(smem);
  const int F0 = p.n[0] / KVT;                       // full tiles of segment 0
  const size_t kstep = (size_t)KVT * p.ldk[0] * 2;

  // ring stage byte offsets of the tiles t, t + 1, t + 2 (rotated once per tile; never reset under PERSIST)
  unsigned o_cur = 0u, o_next = 8192u, o_nn = 16384u;
  u32x4 qreg[NG][NKS];                               // raw Q rows of the block about to start
  bool have_pref = false;                            // its Q rows and its tiles 0 / 1 were fetched during the previous block
  bool pads_ready = false;

  // (a do-while whose condition is constant-false without PERSIST: a `for` over blocks made the compiler hoist the cold paths'
  // address arithmetic across the whole body -- 16 VGPRs more, spilled into a0..a15, i.e. INTO the accumulators; the ISA check
  // caught it)
  TR_DECL
#ifdef IDF_ATTN4W_TRACE
  unsigned long long tr_blk = __builtin_readcyclecounter();
#endif
  int blk = blockIdx.x;
  do {
    int qb, h, b;
    locate(blk, qb, h, b);
    const int blk_n = blk + (int)gridDim.x;          // (the loop's own increment is in its condition)
    const bool has_next = PERSIST && blk_n < total && T >= 3;
    int qbn = 0, hn = 0, bn = 0;
    if (has_next) locate(blk_n, qbn, hn, bn);

    // ---- block prologue.  Not prefetched (first block of the workgroup; after a rerun; T < 3): the K loads of tiles 0 / 1 go out
    // before anything else, then the Q rows; the V^T loads carry the ones-row group of a tail tile, which must land AFTER
    // init_lds' plain ones row: behind its barrier.
    if (!have_pref) {
      issue_k(0, b, h, o_cur);
      if (T > 1) issue_k(1, b, h, o_next);
#pragma unroll
      for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) qreg[g][ks] = load_q_raw(g, ks, qb, h, b);
      if (!pads_ready) { init_lds(); pads_ready = true; ones_mask = 0u; }
      __syncthreads();
      issue_v(0, b, h, o_cur);
      if (T > 1) issue_v(1, b, h, o_next);
    }
    static_for<NG * NMT * 16>([&](auto rc) { agpr_write<OA_BASE + decltype(rc)::value>(0u); });
    static_for<NG>([&](auto gc) {
      constexpr int G = decltype(gc)::value;
      static_for<NKS>([&](auto kc) {
        constexpr int KS = decltype(kc)::value;
        const u32x4 q = scale_q(qreg[G][KS]);
        static_for<4>([&](auto wc) { agpr_write<QA_BASE + 4 * (3 * G + KS) + decltype(wc)::value>(q[decltype(wc)::value]); });
      });
    });
    // tiles 0 and 1 (and Q) landed ...  Prefetched: those loads are OLDER than the previous block's output stores (NG * 5 per
    // wave), which retire in order behind them and need not be waited for -- a vmcnt(0) here cost the persistent form the whole
    // store round trip per block (it came out 2 % SLOWER than the one-block-per-workgroup launch, whose stores drain after s_endpgm)
    if (PERSIST && have_pref) {
      static_assert(NG * 32 * DCH / 64 == (NG == 4 ? 10 : 5), "output store instructions per wave");
      if constexpr (NG == 4) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();                                 // ... and visible; tile 2 goes out
    if (T > 2) { issue_k(2, b, h, o_nn); issue_v(2, b, h, o_nn); }

    // ---- tile 0: the exact maximum of every query over its first 64 keys fixes its reference value m (the -m slot of Q is
    // still 0); P = 2^(s - m) with m = max + SHIFT, rounded to the storage type (the SAME m enters every P and the denominator)
    {
      u32x4 kfl[2][NKS];
      load_kf(kfl[0], o_cur, 0);
      load_kf(kfl[1], o_cur, 1);
      {
        const unsigned vb = smem_lds + vs_base + o_cur;
        ds_read128_vslot<NG, VSLOT(0, 0, 0), 0>(vb + vA[0]); ds_read128_vslot<NG, VSLOT(0, 0, 1), 0>(vb + vA[1]);
        ds_read128_vslot<NG, VSLOT(0, 1, 0), 4096>(vb + vA[0]); ds_read128_vslot<NG, VSLOT(0, 1, 1), 4096>(vb + vA[1]);
        ds_read128_vslot<NG, VSLOT(1, 0, 0), 0>(vb + vA[2]); ds_read128_vslot<NG, VSLOT(1, 0, 1), 0>(vb + vA[3]);
        ds_read128_vslot<NG, VSLOT(1, 1, 0), 4096>(vb + vA[2]); ds_read128_vslot<NG, VSLOT(1, 1, 1), 4096>(vb + vA[3]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      static_for<NG>([&](auto gc) {
        constexpr int G = decltype(gc)::value;
        f32x16 s0, s1;
        FENCE
        mf_s<DT, NG, G, 0, true>(s0, kfl[0][0]); mf_s<DT, NG, G, 0, true>(s1, kfl[1][0]);
        mf_s<DT, NG, G, 1, false>(s0, kfl[0][1]); mf_s<DT, NG, G, 1, false>(s1, kfl[1][1]);
        mf_s<DT, NG, G, 2, false>(s0, kfl[0][2]); mf_s<DT, NG, G, 2, false>(s1, kfl[1][2]);
        MFMA_DRAIN2(s0, s1)
        FENCE
        float m0 = fmaxf(s0[0], s0[1]), m1 = fmaxf(s1[0], s1[1]);
#pragma unroll
        for (int r = 2; r < 16; r += 2) {
          m0 = fmaxf(fmaxf(m0, s0[r]), s0[r + 1]);
          m1 = fmaxf(fmaxf(m1, s1[r]), s1[r + 1]);
        }
        // keys beyond n in a tail tile are clamped duplicates of a valid key: they cannot raise the max
        const float m = Elem<DT>::to_f32(Elem<DT>::from_f32(half_max(fmaxf(m0, m1)) + RefShiftW<DT>::v));
        u32x4 pk[4];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          s0[r] = __builtin_amdgcn_exp2f(s0[r] - m);
          s1[r] = __builtin_amdgcn_exp2f(s1[r] - m);
        }
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
          for (int w = 0; w < 4; ++w) {
            pk[k2][w] = pack2<DT>(s0[8 * k2 + 2 * w], s0[8 * k2 + 2 * w + 1]);
            pk[2 + k2][w] = pack2<DT>(s1[8 * k2 + 2 * w], s1[8 * k2 + 2 * w + 1]);
          }
        FENCE
        mf_o<DT, NG, G, 0, VSLOT(0, 0, 0)>(pk[0]); mf_o<DT, NG, G, 1, VSLOT(0, 1, 0)>(pk[0]);
        mf_o<DT, NG, G, 0, VSLOT(0, 0, 1)>(pk[1]); mf_o<DT, NG, G, 1, VSLOT(0, 1, 1)>(pk[1]);
        mf_o<DT, NG, G, 0, VSLOT(1, 0, 0)>(pk[2]); mf_o<DT, NG, G, 1, VSLOT(1, 1, 0)>(pk[2]);
        mf_o<DT, NG, G, 0, VSLOT(1, 0, 1)>(pk[3]); mf_o<DT, NG, G, 1, VSLOT(1, 1, 1)>(pk[3]);
        // -m into the spare slot of the last K-step (hi = 1 lanes, element 0 of the fragment)
        constexpr int QR = QA_BASE + 4 * (3 * G + NKS - 1);
        const unsigned q20 = agpr_read<QR>();
        agpr_write<QR>(hi ? pack2<DT>(-m, 0.0f) : q20);
        FENCE
      });
    }
    end_tile();
    { const unsigned o = o_cur; o_cur = o_next; o_next = o_nn; o_nn = o; }

    if (T > 1) {
      // ---- pipeline fill for tile 1: its K fragments (both halves), the V^T fragments of its first half, the scores of its
      // first block for every group and group 0's exponentials.
      // State between 32-key blocks kb -> kb + 1 (kb = 2 t + half):
      //   s[1..3] = scores of block kb for groups 1..3, pkb[0] = packed P of block kb for group 0,
      //   kfr[(kb + 1) & 1] = K fragments of block kb + 1, V^T fragment set kb & 1 = those of block kb.
      {
        // (asm reads here too: a compiler-visible ds_read into these registers makes its waitcnt pass guard their first use in
        // the loop with s_waitcnt lgkmcnt(n) -- which then waits for the stream's own asm reads issued just before)
        const unsigned ko = smem_lds + o_cur, vo = smem_lds + vs_base + o_cur;
        kfr[0][0] = ds_read128_asm<0>(ko + kA);        kfr[0][1] = ds_read128_asm<32>(ko + kA);        kfr[0][2] = ds_read128_asm<0>(ko + kB);
        kfr[1][0] = ds_read128_asm<2560>(ko + kA);     kfr[1][1] = ds_read128_asm<2592>(ko + kA);     kfr[1][2] = ds_read128_asm<2560>(ko + kB);
        ds_read128_vslot<NG, VSLOT(0, 0, 0), 0>(vo + vA[0]);    ds_read128_vslot<NG, VSLOT(0, 0, 1), 0>(vo + vA[1]);
        ds_read128_vslot<NG, VSLOT(0, 1, 0), 4096>(vo + vA[0]); ds_read128_vslot<NG, VSLOT(0, 1, 1), 4096>(vo + vA[1]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        FENCE
        static_for<NG>([&](auto gc) {
          constexpr int G = decltype(gc)::value;
          mf_s<DT, NG, G, 0, true>(s[G], kfr[0][0]); mf_s<DT, NG, G, 1, false>(s[G], kfr[0][1]); mf_s<DT, NG, G, 2, false>(s[G], kfr[0][2]);
        });
        MFMA_DRAIN2(s[0], s[1])
        FENCE
#pragma unroll
        for (int r = 0; r < 16; ++r) s[0][r] = __builtin_amdgcn_exp2f(s[0][r]);
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
          for (int w = 0; w < 4; ++w) pkb[0][k2][w] = pack2<DT>(s[0][8 * k2 + 2 * w], s[0][8 * k2 + 2 * w + 1]);
        FENCE
      }

      // per-wave DMA slots of the steady state (full tiles of segment 0): slot j of wave w -- j = 0: K instruction w; j = 1: K
      // instruction 4 (wave 0) / V^T instruction w - 1; j = 2: V^T instruction 3 (wave 0) / 4 (wave 1), none on waves 2 / 3
      unsigned doff[3];                                  // per-lane byte offset from the tile's K / V^T base
      unsigned dlds[3];                                  // LDS byte offset of the slot inside smem, stage 0 (wave-uniform)
      bool disk[3];                                      // slot reads K (else V^T); wave-uniform
      {
        const int ki[3] = {wave, wave == 0 ? 4 : -1, -1};
        const int vi[3] = {-1, wave == 0 ? -1 : wave - 1, wave == 0 ? 3 : (wave == 1 ? 4 : -1)};
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          disk[j] = ki[j] >= 0;
          if (ki[j] >= 0) {
            const int c = ki[j] * 64 + lane;
            const int row = c / DCH, col = (c - row * DCH) * 8;
            doff[j] = (unsigned)(row * p.ldk[0] + col) * 2u;
            dlds[j] = (unsigned)__builtin_amdgcn_readfirstlane(ki[j] * 1024);
          } else {
            const int vv = vi[j] >= 0 ? vi[j] : 0;
            const int row = vv * 8 + (lane >> 3);
            doff[j] = (unsigned)(row * p.ldv[0] + ((lane & 7) ^ ((row >> 1) & 7)) * 8) * 2u;
            dlds[j] = (unsigned)__builtin_amdgcn_readfirstlane(NST * KSTG * 2 + vv * 1024);
          }
        }
      }
      // one block: HALF = which 32-key half of tile t.  The K fragments of block kb + 2 and the V^T fragments of block kb + 1
      // are read here (stages t and t + 1: visible since the barrier that ended tile t - 1).  `dma(j)` issues the wave's j-th
      // LDS-DMA instruction of the tile.
      auto block = [&](auto half_tag, const unsigned ko_next, const unsigned vo_cur, const unsigned vo_next, auto&& dma) {
        constexpr int HALF = decltype(half_tag)::value;
        u32x4 (&kfn)[NKS] = kfr[(HALF + 1) & 1];       // K fragments of block kb + 1
        constexpr int VC = HALF, VW = (HALF + 1) & 1;  // V^T fragment sets: block kb's / <- block kb + 1's (over block kb - 1's: dead)
        u32x4 (&kfw)[NKS] = kfr[HALF];                 // <- K fragments of block kb + 2 (over block kb's: dead)
        const unsigned vo_w = HALF == 0 ? vo_cur : vo_next;
        constexpr int vh = HALF == 0 ? 1 : 0;          // which half of that stage
        const unsigned ka = smem_lds + ko_next + kA + HALF * 2560, kb2 = smem_lds + ko_next + kB + HALF * 2560;
        const unsigned va0 = smem_lds + vs_base + vo_w + vA[vh * 2 + 0], va1 = smem_lds + vs_base + vo_w + vA[vh * 2 + 1];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the fragments read (by asm) in the previous block
        static_for<NG>([&](auto gc) {
          constexpr int g = decltype(gc)::value;
          u32x4 (&pc)[2] = pkb[g & 1];
          u32x4 (&pn)[2] = pkb[(g + 1) & 1];
          constexpr int gn = (g + 1) & (NG - 1);              // the group whose exponentials ride in this step
          // e[i] = 2^score i of group gn (fresh registers: the score tuple itself is overwritten by the group's next K.Q^T)
#ifdef IDF_A4W_NOVALU      /* timing experiment: the stream without its exponentials / conversions (results are garbage) */
#define EX(i) e[i] = s[gn][i];
#define CV(i) pn[(i) >> 2][(i) & 3] = __float_as_uint(e[2 * (i)]);
#elif defined(IDF_A4W_NOEXP)
#define EX(i) e[i] = s[gn][i];
#define CV(i) pn[(i) >> 2][(i) & 3] = cvt_pk_asm<DT>(e[2 * (i)], e[2 * (i) + 1]);
#elif defined(IDF_A4W_NOCVT)
#define EX(i) e[i] = exp2_asm(s[gn][i]);
#define CV(i) pn[(i) >> 2][(i) & 3] = __float_as_uint(e[2 * (i)]) ^ __float_as_uint(e[2 * (i) + 1]);
#else
#define EX(i) e[i] = exp2_asm(s[gn][i]);
#define CV(i) pn[(i) >> 2][(i) & 3] = cvt_pk_asm<DT>(e[2 * (i)], e[2 * (i) + 1]);
#endif
#ifdef IDF_A4W_NOMFMA      /* timing experiment: the stream without its MFMAs */
#define SMF(...) [](auto&&...) {}
#define OMF(...) [](auto&&...) {}
#else
#define SMF(...) mf_s<__VA_ARGS__>
#define OMF(...) mf_o<__VA_ARGS__>
#endif
          float e[16];
          // Measured on variants of this stream (tools/build_attn_variant.sh; cycles per 28-MFMA block of one wave): MFMAs alone
          // 924, VALU / LDS work alone 908 (its 64 v_exp_f32 alone 760: ~12 cycles each, the 32 v_cvt_pk ride in their shadow),
          // both 1056-1068 -- the transcendental pipe is as loaded as the matrix pipe, and the stream overlaps them to within 15 %.
          // Spreading the exponentials evenly over the gaps (3 / 2 / 2 / 2 / 3 / 2 / 2) measured 2 % SLOWER than bunching them at
          // the head of the step as here.
          SMF(DT, NG, g, 0, true)(s[g], kfn[0]);
          EX(0) EX(1) EX(2) EX(3)
          OMF(DT, NG, g, 0, VSLOT(VC, 0, 0))(pc[0]);
          EX(4) EX(5) EX(6) CV(0)
          SMF(DT, NG, g, 1, false)(s[g], kfn[1]);
          EX(7) EX(8) CV(1) EX(9)
          OMF(DT, NG, g, 1, VSLOT(VC, 1, 0))(pc[0]);
          CV(2)
          if constexpr (NG == 4) { if (HALF == 0 && g < 3) dma(g); }
          else { if (HALF == 0) dma(g); if (HALF == 1 && g == 0) dma(2); }
          EX(10) EX(11) CV(3)
          SMF(DT, NG, g, 2, false)(s[g], kfn[2]);
          EX(12) EX(13) CV(4) CV(5)
          OMF(DT, NG, g, 0, VSLOT(VC, 0, 1))(pc[1]);
          EX(14) EX(15) CV(6)
          if constexpr (NG == 4) {
            if (g == 0) ds_read128_vslot<NG, VSLOT(VW, 0, 0), 0>(va0);
            if (g == 1) ds_read128_vslot<NG, VSLOT(VW, 0, 1), 0>(va1);
            if (g == 2) kfw[1] = ds_read128_asm<32>(ka);
          } else {
            if (g == 0) { ds_read128_vslot<NG, VSLOT(VW, 0, 0), 0>(va0); ds_read128_vslot<NG, VSLOT(VW, 0, 1), 0>(va1); kfw[0] = ds_read128_asm<0>(ka); kfw[1] = ds_read128_asm<32>(ka); }
          }
          OMF(DT, NG, g, 1, VSLOT(VC, 1, 1))(pc[1]);
          CV(7)
          if constexpr (NG == 4) {
            if (g == 0) ds_read128_vslot<NG, VSLOT(VW, 1, 0), 4096>(va0);
            if (g == 1) { ds_read128_vslot<NG, VSLOT(VW, 1, 1), 4096>(va1); kfw[0] = ds_read128_asm<0>(ka); }
            if (g == 2) kfw[2] = ds_read128_asm<0>(kb2);
          } else {
            if (g == 0) { ds_read128_vslot<NG, VSLOT(VW, 1, 0), 4096>(va0); ds_read128_vslot<NG, VSLOT(VW, 1, 1), 4096>(va1); kfw[2] = ds_read128_asm<0>(kb2); }
          }
#undef EX
#undef CV
#undef SMF
#undef OMF
        });
      };
      using H0 = std::integral_constant<int, 0>;
      using H1 = std::integral_constant<int, 1>;

      TR_MARK(4)
      const char* kptr = reinterpret_cast<const char*>(p.k[0] + (size_t)b * p.sK[0] + h * D) + (size_t)3 * KVT * p.ldk[0] * 2;
      const char* vptr = reinterpret_cast<const char*>(p.vt[0] + (size_t)b * p.sV[0] + (size_t)(h * D) * p.ldv[0]) + (size_t)3 * KVT * 2;
      for (int t = 1; t < T; ++t) {
        TR_START
        // Every wave passed the barrier that ended tile t-1: K(t+1), V^T(t+1) are visible, the stage o_nn is dead.
        const unsigned ko_next = o_next, vo_cur = o_cur, vo_next = o_next;
        // the loads for tile t + 2: in the steady state (a full tile of segment 0 into a stage with a plain ones row) they ride in
        // the stream; otherwise they go out here -- this block's tile, or (PERSIST) tile 0 / 1 of the NEXT block once this one
        // has no more, together with its Q rows (two tiles before the end: landed long before anybody waits for them)
        const bool hot = t + 2 < F0 && !(ones_mask & (1u << (o_nn >> 13)));
        if (!hot) {
          if (t + 2 < T) {
            issue_k(t + 2, b, h, o_nn);
            issue_v(t + 2, b, h, o_nn);
          } else if (has_next) {
            issue_k(t + 2 - T, bn, hn, o_nn);
            issue_v(t + 2 - T, bn, hn, o_nn);
            if (t + 2 == T) {
#pragma unroll
              for (int g = 0; g < NG; ++g)
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) qreg[g][ks] = load_q_raw(g, ks, qbn, hn, bn);
            }
          }
        }
        const int ndma_t = hot ? ndma : 0;             // wave-uniform
        const unsigned kst = smem_lds + o_nn, vst = kst;
        block(H0{}, ko_next, vo_cur, vo_next, [&](const int j) {
          if (j < ndma_t) dma16_sv(disk[j] ? kptr : vptr, doff[j], (disk[j] ? kst : vst) + dlds[j]);
        });
        TR(0)
        block(H1{}, ko_next, vo_cur, vo_next, [&](const int j) {
          if (j < ndma_t) dma16_sv(disk[j] ? kptr : vptr, doff[j], (disk[j] ? kst : vst) + dlds[j]);
        });
        if (hot) {
          kptr += kstep;
          vptr += KVT * 2;
        }
        { const unsigned o = o_cur; o_cur = o_next; o_next = o_nn; o_nn = o; }
        TR(1)
        end_tile();
        TR(2)
#ifdef IDF_ATTN4W_TRACE
        tr_acc[3] += 1;
#endif
      }
      TR_MARK(5)
      MFMA_DRAIN                                       // the last asm MFMAs have retired before anything reads O
    }
    have_pref = has_next;

    // ---- normalise, check.  O^T tile (g, mt), register r: e = mt*32 + (r&3) + 8*(r>>2) + 4*hi, q = l31 of group g; row e = D
    // (tile 1, register 4 of the hi = 0 lanes) is the denominator.  Each wave transposes its 32 NG x D block through its own LDS
    // slice (which aliases the rings unless PERSIST: every wave passed the last tile's barrier, the rings are dead).
    bool bad = false;
    static_for<NG>([&](auto gc) {
      constexpr int g = decltype(gc)::value;
      float o0[16], o1[5];
      static_for<16>([&](auto rc) { o0[decltype(rc)::value] = __uint_as_float(agpr_read<OA_BASE + 32 * g + decltype(rc)::value>()); });
      static_for<5>([&](auto rc) { o1[decltype(rc)::value] = __uint_as_float(agpr_read<OA_BASE + 32 * g + 16 + decltype(rc)::value>()); });
      const float l_tot = __shfl(o1[4], l31, 64);
      const float inv = 1.0f / l_tot;
      bad |= !(l_tot > 0.0f && l_tot < 0x1p120f);
      unsigned short* orow = ow + (g * 32 + l31) * D;
#pragma unroll
      for (int qd = 0; qd < 5; ++qd) {                 // e = 8 qd + 4 hi < 40
        const float v0 = (qd < 4 ? o0[4 * qd] : o1[0]) * inv, v1 = (qd < 4 ? o0[4 * qd + 1] : o1[1]) * inv;
        const float v2 = (qd < 4 ? o0[4 * qd + 2] : o1[2]) * inv, v3 = (qd < 4 ? o0[4 * qd + 3] : o1[3]) * inv;
        bad |= !(fabsf(v0) < 0x1p120f) | !(fabsf(v1) < 0x1p120f) | !(fabsf(v2) < 0x1p120f) | !(fabsf(v3) < 0x1p120f);
        u32x2 pkd = {pack2<DT>(v0, v1), pack2<DT>(v2, v3)};
        *reinterpret_cast<u32x2*>(orow + 8 * qd + 4 * hi) = pkd;
      }
    });
    // a non-finite denominator or output (a P overflowed the storage type: scores ~2^7 [bf16] / 23 [fp16] log2 units above the
    // first tile's maximum): the WORKGROUP reruns the block with the classic per-tile running max, one query group at a time
    if (__builtin_amdgcn_ballot_w64(bad) != 0 && lane == 0) *redo_flag = 1;
    __syncthreads();
    if (*redo_flag != 0) {
      // (any prefetch for the next block is abandoned: the rerun uses the ring from stage 0 and the next block starts afresh)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      for (int g = 0; g < NG; ++g) {                  // NOT unrolled: a small, register-lean loop (rare path)
        __syncthreads();                              // (every wave is done with the previous group's stages)
        issue_k(0, b, h, 0u);
        if (T > 1) issue_k(1, b, h, 8192u);
        init_lds();
        ones_mask = 0u;
        __syncthreads();
        issue_v(0, b, h, 0u);
        if (T > 1) issue_v(1, b, h, 8192u);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (T > 2) { issue_k(2, b, h, 16384u); issue_v(2, b, h, 16384u); }
        u32x4 q[NKS];
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) q[ks] = scale_q(load_q_raw(g, ks, qb, h, b));
        f32x16 ox[NMT] = {zero16, zero16};
        float m_run = 0.0f;
        for (int t = 0; t < T; ++t) {
          if (t > 0 && t + 2 < T) { issue_k(t + 2, b, h, (unsigned)(((t + 2) % NST) * 8192)); issue_v(t + 2, b, h, (unsigned)(((t + 2) % NST) * 8192)); }
          const unsigned ko = (unsigned)((t % NST) * KSTG * 2), vo = (unsigned)((t % NST) * VSZ * 2);
          f32x16 sx[2] = {zero16, zero16};
#pragma unroll
          for (int half = 0; half < 2; ++half) {
            u32x4 kf[NKS];
            load_kf(kf, ko, half);
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) sx[half] = Elem<DT>::mfma32(kf[ks], q[ks], sx[half]);
          }
          float m0 = fmaxf(sx[0][0], sx[0][1]), m1 = fmaxf(sx[1][0], sx[1][1]);
#pragma unroll
          for (int r = 2; r < 16; r += 2) {
            m0 = fmaxf(fmaxf(m0, sx[0][r]), sx[0][r + 1]);
            m1 = fmaxf(fmaxf(m1, sx[1][r]), sx[1][r + 1]);
          }
          // scores are relative to the current m (through the -m slot); raise it so that the tile's maximum maps to 2^-SHIFT
          const float want = half_max(fmaxf(m0, m1)) + RefShiftW<DT>::v;
          const float delta = t == 0 ? want : fmaxf(want, 0.0f);
          const float m_new = Elem<DT>::to_f32(Elem<DT>::from_f32(m_run + delta));
          const float d_eff = m_new - m_run;
          m_run = m_new;
          const float al = t == 0 ? 1.0f : __builtin_amdgcn_exp2f(-d_eff);
#pragma unroll
          for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) ox[mt][r] *= al;
          q[NKS - 1][0] = hi ? pack2<DT>(-m_new, 0.0f) : q[NKS - 1][0];
#pragma unroll
          for (int half = 0; half < 2; ++half) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sx[half][r] = __builtin_amdgcn_exp2f(sx[half][r] - d_eff);
            u32x4 vf[NMT][2];
            load_vf(vf, vo, half);
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
              u32x4 pk;
#pragma unroll
              for (int w = 0; w < 4; ++w) pk[w] = pack2<DT>(sx[half][8 * k2 + 2 * w], sx[half][8 * k2 + 2 * w + 1]);
#pragma unroll
              for (int mt = 0; mt < NMT; ++mt) ox[mt] = Elem<DT>::mfma32(vf[mt][k2], pk, ox[mt]);
            }
          }
          end_tile();
        }
        const float l_tot = __shfl(ox[1][4], l31, 64);
        const float inv = 1.0f / l_tot;
#pragma unroll
        for (int qd = 0; qd < 5; ++qd) {
          const float v0 = (qd < 4 ? ox[0][4 * qd] : ox[1][0]) * inv, v1 = (qd < 4 ? ox[0][4 * qd + 1] : ox[1][1]) * inv;
          const float v2 = (qd < 4 ? ox[0][4 * qd + 2] : ox[1][2]) * inv, v3 = (qd < 4 ? ox[0][4 * qd + 3] : ox[1][3]) * inv;
          u32x2 pkd = {pack2<DT>(v0, v1), pack2<DT>(v2, v3)};
          // the group's rows go straight to global memory (8-B pieces; the rare path)
          const int qrow = qb * QB + wave * (NG * 32) + g * 32 + l31;
          if (qrow < p.nq) *reinterpret_cast<u32x2*>(p.out + (size_t)b * p.sO + (size_t)qrow * p.ldo + h * D + 8 * qd + 4 * hi) = pkd;
        }
      }
      __syncthreads();
      if (tid == 0) *redo_flag = 0;
      have_pref = false;                              // the next block starts from stage 0 with a plain prologue
      o_cur = 0u; o_next = 8192u; o_nn = 16384u;
      continue;
    }

    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    {
      const int q0 = qb * QB + wave * (NG * 32);
      unsigned short* const obase = p.out + (size_t)b * p.sO + h * D;
#pragma unroll
      for (int j = 0; j < NG * 32 * DCH / 64; ++j) {
        const int c = lane + 64 * j;
        const int row = c / DCH, col = c - row * DCH;
        const u32x4 v = *reinterpret_cast<const u32x4*>(ow + c * 8);
        if (q0 + row < p.nq) *reinterpret_cast<u32x4*>(obase + (size_t)(q0 + row) * p.ldo + col * 8) = v;
      }
    }
    TR_MARK(6)
#ifdef IDF_ATTN4W_TRACE
    tr_acc[7] += 1;
#endif
    TR_DUMP
  } while (PERSIST && (blk += (int)gridDim.x) < total);
#undef FENCE
#undef MFMA_DRAIN
}

template <int DT, int NG, bool PERSIST>
int launch_attn4w_cfg(const AttnParams& p, int B, hipStream_t s) {
  constexpr int QB = NW * NG * 32;
  constexpr int RING = NST * KSTG + NST * VSZ, OSTAGE = NW * NG * 32 * D;
  constexpr int LDS = (PERSIST ? RING + OSTAGE : (RING > OSTAGE ? RING : OSTAGE)) * 2 + 16;
  const int nqb = (p.nq + QB - 1) / QB;
  const int total = nqb * p.H * B;
  auto kern = attn4w_kernel<DT, NG, PERSIST>;
  static std::atomic<unsigned long long> attr_done{0};
  if (LDS > 65536) {
    if (const int rc = idf_lds_optin(reinterpret_cast<const void*>(kern), LDS, attr_done)) return rc;
  }
  int grid = total;
  if (PERSIST) {
    static int ncu = 0;
    if (ncu == 0) {
      int dev = 0, v = 0;
      if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ncu = v;
      else ncu = 256;
    }
    int g = ncu & ~7;                               // a multiple of 8: a workgroup's blocks stay on one XCD
    if (g < 8) g = 8;
    if (total > g && (total & 7) == 0) grid = g;
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), LDS, s, p, nqb, 1, total);
  return idf_launch_status();
}

template <int DT>
int launch_attn4w(const AttnParams& p, int B, int variant, hipStream_t s) {
  if (variant == 4) return launch_attn4w_cfg<DT, 4, false>(p, B, s);
#ifdef IDF_ATTN4W_PERSIST_EXPERIMENT
  // The persistent form (PERSIST) is NOT in the shipped library: built, bit-identical, and 1-2 % SLOWER than mode 4 (profiles/
  // r06_attn4w_persist*.log: the prologue shrinks from 18 k to 10.5 k cycles per block, but the stream's DMA-carrying half-tile
  // goes from 1478 to 1761 cycles -- 256 workgroups in lock step issue their LDS-DMA at the same moments).  A variant build
  // (tools/build_attn_variant.sh persist -DIDF_ATTN4W_PERSIST_EXPERIMENT) serves it as mode 6.
  if (variant == 6) return launch_attn4w_cfg<DT, 4, true>(p, B, s);
#endif
  if (variant == 6) return IDF_ATTN2_UNSUPPORTED;
  return launch_attn4w_cfg<DT, 2, false>(p, B, s);
}

}  // namespace

#ifdef IDF_ATTN4W_TRACE
extern "C" int idf_attn4w_trace_read(unsigned long long* host /* [4][4] */) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(idf_attn4w_trace_buf), sizeof(idf_attn4w_trace_buf));
}
#endif

int idf_launch_attn4w(const AttnParams& p, int B, int dtype, int variant /* IDF_TUNE_ATTN2 value: 4, 5, 6 */, hipStream_t s) {
  if (p.d != 40) return IDF_ATTN2_UNSUPPORTED;
  if ((p.n[0] % 8) || (p.n[1] % 8)) return IDF_ATTN2_UNSUPPORTED;
  if ((p.ldk[0] % 8) || (p.ldv[0] % 8) || (p.n[1] > 0 && ((p.ldk[1] % 8) || (p.ldv[1] % 8)))) return IDF_ATTN2_UNSUPPORTED;
  if (!aligned16(p.k[0]) || !aligned16(p.vt[0]) || !aligned16(p.k[1]) || !aligned16(p.vt[1])) return IDF_ATTN2_UNSUPPORTED;
  if ((p.sK[0] % 8) || (p.sV[0] % 8) || (p.sK[1] % 8) || (p.sV[1] % 8)) return IDF_ATTN2_UNSUPPORTED;
  if (!aligned16(p.out) || (p.ldo % 8) || (p.sO % 8) || !aligned16(p.q) || (p.ldq % 8) || (p.sQ % 8)) return IDF_ATTN2_UNSUPPORTED;
  if ((long long)KVT * p.ldk[0] * 2 >= (1ll << 31) || (long long)(p.d + 8) * p.ldv[0] * 2 >= (1ll << 31)) return IDF_ATTN2_UNSUPPORTED;
  if (p.n[1] > 0 && ((long long)KVT * p.ldk[1] * 2 >= (1ll << 31) || (long long)(p.d + 8) * p.ldv[1] * 2 >= (1ll << 31)))
    return IDF_ATTN2_UNSUPPORTED;
  if (dtype == IDF_BF16) return launch_attn4w<IDF_BF16>(p, B, variant, s);
  if (dtype == IDF_F16) return launch_attn4w<IDF_F16>(p, B, variant, s);
  return IDF_ATTN2_UNSUPPORTED;
}
