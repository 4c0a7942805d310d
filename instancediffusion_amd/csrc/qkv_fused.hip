// qkv_fused.hip -- the fused q | k | v projection of a C = 320 transformer block with the activation rows RESIDENT IN REGISTERS
// (gfx950): out[m][0 .. 639] = LN(x) . [Wq; Wk]^T, vt[c][m] = (LN(x) . Wv^T)^T   (attention.py:168-172 to_q / to_k / to_v of
// CrossAttention / SelfAttention; call site engine._self_attn; C ABI: idf_gemm with vt_out, include/idf.h).
//
// Why.  On the persistent GEMM kernel (gemm_big.hip, 256 x 320 tiles) this launch runs at 534 TF = 0.60 ms at 128 rows of 64 x 64
// latents: K = 320 is five K-tiles per output tile, so prologue, epilogue and the accumulator turn-around of EVERY tile are
// exposed, against 0.30-0.35 ms for its 1.34 GB of HBM traffic.  Here the roles are those of mlp320w_kernel (mlp_fused.hip):
//   * a workgroup is 4 waves, one per SIMD; wave w owns rows 32 w .. + 31 of a 128-row tile, whose 320 elements per row are
//     loaded ONCE into 20 MFMA operand fragments in asm-owned AGPRs and serve all 960 output columns;
//   * the [960][320] weight image streams through a 2-slot LDS ring in 15 chunks of 64 rows by LDS-DMA (the fused MLP's W1
//     chunk geometry: same piece roles, same swizzle, same fragment reads), one barrier per chunk;
//   * per pipeline step: 40 MFMAs of chunk i + 1 (two independent 32 x 32 chains) carry the epilogue of chunk i in their gaps --
//     LayerNorm fold + bias in fp32, 16-bit, through a wave-private LDS staging slot so that a store instruction writes whole
//     lines (8 rows x 128 B of q | k, 16 rows x 64 B of V^T), 4 stores per chunk -- as a GENERATED `asm volatile` stream
//     (tools/gen_qkvw_stream.py -> qkvw_stream.inc; primitives mw_prims.h);
//   * V chunks (columns 640 ..) run with the MFMA operands swapped, so a lane owns a channel and its registers the wave's 32
//     tokens: after two v_permlane32_swap per register pair a lane holds 16 consecutive tokens of its channel;
//   * the next tile's rows are fetched in the last step of a tile BEFORE its stores, and every top-of-step wait is a counted
//     vmcnt: the stream never waits for a store round trip.
// Taken by idf_gemm when K = 320, N = 960, vt_col0 = 640, M % 128 == 0, M >= 128 x (number of CUs) / 2, LN_ROW with the
// statistics handed in (ln_stats != NULL) and BIAS; everything else stays on gemm_big.hip.  Same arithmetic per output element
// (fp32 accumulation over k = 0 .. 319 in the same order, then rstd * (acc - mu c) + d), up to the fma contraction of the fold.
// LDS: 2 x 40 KB ring + 7.5 KB (c | d of all 960 columns) + 1 KB (the waves' (-mu, rstd) tables) + 4 x 4 KB staging = 105 KB.
#include "gemm_core.h"
#include "mw_prims.h"
#include <cstdlib>
#include <atomic>

using namespace idfcore;
using namespace idfmw;

namespace {

constexpr int QW_BM = 128, QW_C = 320, QW_N = 960, QW_NCH = 15, QW_VCH0 = 10;      // chunks 10 .. 14 are V columns
constexpr int QW_SLOT = 5 * 64 * 128;                                               // one W chunk: 5 K-tiles x [64 rows][64 k]
constexpr int QW_CD_OFF = 2 * QW_SLOT, QW_ST_OFF = QW_CD_OFF + 2 * QW_N * 4, QW_STG_OFF = QW_ST_OFF + 4 * 256;
constexpr int QW_SMEM = QW_STG_OFF + 4 * 4096;
constexpr int QW_NAGPR = 242;                                                        // x fragments a0..a79, a240:241 = next tile's (mu, rstd)

struct QwParams {
  const unsigned short* x; int ldx;
  const float* ln_stats;                 // [M][2] (mu, rstd)
  const unsigned short* w; int ldw;      // [960][320] gamma-folded
  const float* c; const float* d;        // [960] row sums of w; beta term + bias
  unsigned short* out; int ldo;          // [M][>= 640]
  unsigned short* vt; int ld_vt;         // [320][>= M]
  int M;
};

struct QwCtx {
  unsigned w1a[4];                       // LDS byte addresses of the W fragment reads of the chunk whose MFMAs run (per lane, by ks & 3)
  unsigned cdq, cdv, stt;                // constants of the chunk in its epilogue (q | k: + 16 hi; V: + 4 l31); the wave's (-mu, rstd) table + 32 hi
  unsigned qw[8], qr[4], vw[2], vr[2];   // staging slot: q | k write (by 16-B slot) / read-back addresses, V^T write / read-back
  unsigned qst[4], vst[2];               // per-lane store offsets (bytes)
  const void* obase; const void* vtb[2]; // uniform store bases of this step
  float nmu, rstd;                       // of the lane's token row
  unsigned w1dst, w1_vj; const char* w1b[2];
  const unsigned short* xnext; const float* snext; bool has_next;
};

#ifndef QKVW_STREAM_INC
#define QKVW_STREAM_INC "qkvw_stream.inc"
#endif
#include QKVW_STREAM_INC

template <int DT>
__global__ __launch_bounds__(256, 1) void qkv320w_kernel(const QwParams p, const int tiles) {
  asm volatile("" ::: "a0", "a241");               // the asm-owned AGPR block (this is where the kernel descriptor learns its size)
  extern __shared__ __attribute__((aligned(128))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int G = gridDim.x;
  const unsigned smem_lds = lds_u32(smem);

  QwCtx c;
  // LDS-DMA roles (mlp_fused.hip): piece (kt, u) = rows 8 (wave + 4 u) .. + 7 of K-tile kt: lane -> row + lane / 8, 16-B slot lane % 8
  unsigned w1_voff;
  {
    const int row = 8 * wave + (lane >> 3);
    w1_voff = (unsigned)(row * p.ldw + (((lane & 7) ^ ((row >> 1) & 7)) << 3)) * 2u;
  }
  const char* const wg = reinterpret_cast<const char*>(p.w);
  const unsigned w_chunk = (unsigned)(64 * p.ldw * 2);
#pragma unroll
  for (int u = 0; u < 2; ++u) c.w1b[u] = wg + (size_t)u * (unsigned)(32 * p.ldw * 2);
  const int sw1 = (l31 >> 1) & 7;
  unsigned w1o[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) w1o[i] = smem_lds + (unsigned)(l31 * 128 + (((2 * i + hi) ^ sw1) << 4));
  // staging slot of the wave (4 KB).  q | k image: [32 tokens][128 B], 16-B slot ^= (row >> 1) & 7; a lane writes 8 B of
  // slot s = 4 f + q of its token row (+ 8 hi inside the slot), reads back rows lane / 8 + 8 i, slot lane % 8.
  // V^T image per fragment (2 KB): [32 channels][64 B], slot ^= (row >> 2) & 3; a lane writes slots 2 hi, 2 hi + 1 of its
  // channel row, reads back rows lane / 4 + 16 i, slot lane % 4.
  {
    const unsigned stg = smem_lds + (unsigned)(QW_STG_OFF + wave * 4096);
    const unsigned base = stg + (unsigned)(l31 * 128 + 8 * hi + (sw1 << 4));
#pragma unroll
    for (int s = 0; s < 8; ++s) c.qw[s] = base ^ (unsigned)(16 * s);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = (lane >> 3) + 8 * i;
      c.qr[i] = stg + (unsigned)(row * 128 + (((lane & 7) ^ ((row >> 1) & 7)) << 4));
      c.qst[i] = (unsigned)(row * p.ldo * 2 + (lane & 7) * 16);
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) c.vw[s] = stg + (unsigned)(l31 * 64 + (((2 * hi + s) ^ ((l31 >> 2) & 3)) << 4));
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = (lane >> 2) + 16 * i;
      c.vr[i] = stg + (unsigned)(row * 64 + (((lane & 3) ^ ((row >> 2) & 3)) << 4));
      c.vst[i] = (unsigned)(row * p.ld_vt * 2 + (lane & 3) * 16);
    }
  }
  c.stt = smem_lds + (unsigned)(QW_ST_OFF + wave * 256 + 32 * hi);
  const unsigned cd_lds = smem_lds + (unsigned)QW_CD_OFF;

  int tile = ((G & 7) == 0) ? (int)(blockIdx.x & 7) * (G >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  if (tile >= tiles) return;

  // kernel prologue: c | d of all 960 columns into LDS, W chunk 0 into ring slot 0, the first tile's rows and statistics
  for (int i = tid; i < QW_N / 4; i += 256) {
    reinterpret_cast<f32x4*>(smem + QW_CD_OFF)[i] = reinterpret_cast<const f32x4*>(p.c)[i];
    reinterpret_cast<f32x4*>(smem + QW_CD_OFF + QW_N * 4)[i] = reinterpret_cast<const f32x4*>(p.d)[i];
  }
#pragma unroll
  for (int kt = 0; kt < 5; ++kt)
#pragma unroll
    for (int u = 0; u < 2; ++u)
      mw_dma_rt(c.w1b[u] + kt * 128, w1_voff, smem_lds + (unsigned)(wave * 1024 + kt * 8192 + u * 4096));
  auto row_ptr = [&](int t) { return p.x + (size_t)(t * QW_BM + wave * 32 + l31) * p.ldx + 8 * hi; };
  auto st_ptr = [&](int t) { return p.ln_stats + 2 * (size_t)(t * QW_BM + wave * 32 + l31); };
  {
    const unsigned short* xr = row_ptr(tile);
    mw_static_for<20>([&](auto kc) { mw_load_x<decltype(kc)::value>(xr); });
    const float* sp = st_ptr(tile);
    asm volatile("global_load_dwordx2 a[240:241], %0, off" ::"v"(sp) : "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

#if defined(QW_DBG) && QW_DBG == 1
  return;
#endif
  f32x16 acc[2][2];
  int g = 0;                                         // chunks streamed so far: chunk j of this tile sits in ring slot (g + j) & 1
  // step i of a tile: MFMAs read chunk i + 1 [slot (g + i + 1) & 1]; its LDS-DMA pieces bring chunk i + 2 [slot (g + i) & 1]
  auto set_step = [&](int i) {
    const unsigned sn = (unsigned)(((g + i + 1) & 1) * QW_SLOT), sj = (unsigned)(((g + i) & 1) * QW_SLOT);
#pragma unroll
    for (int k = 0; k < 4; ++k) c.w1a[k] = w1o[k] + sn;
    int j2 = i + 2;
    if (j2 >= QW_NCH) j2 -= QW_NCH;
    c.w1_vj = w1_voff + (unsigned)j2 * w_chunk;
    c.w1dst = smem_lds + sj + (unsigned)(wave * 1024);
    // epilogue of chunk i
    c.cdq = cd_lds + (unsigned)(i * 256 + 16 * hi);
    c.cdv = cd_lds + (unsigned)(i * 256 + 4 * l31);
    const size_t m0 = (size_t)tile * QW_BM + wave * 32;
    c.obase = reinterpret_cast<const char*>(p.out) + m0 * p.ldo * 2 + (size_t)i * 128;
#pragma unroll
    for (int f = 0; f < 2; ++f)
      c.vtb[f] = reinterpret_cast<const char*>(p.vt) + ((size_t)(64 * (i - QW_VCH0) + 32 * f) * p.ld_vt + m0) * 2;
  };

  for (;;) {
    // the rows (a0..a79) and the statistics (a240:241) of this tile have landed (the caller of this point waited for them)
    {
      const float mu = __uint_as_float(mw_agpr_read<240>()), rs = __uint_as_float(mw_agpr_read<241>());
      c.nmu = -mu; c.rstd = rs;
      asm volatile("" : "+v"(c.nmu), "+v"(c.rstd));
      // the wave's table for the V chunks: token t -> (-mu, rstd) at 8 t (both half-waves hold the token; one writes)
      if (hi == 0) *reinterpret_cast<f32x2*>(smem + QW_ST_OFF + wave * 256 + 8 * l31) = f32x2{-mu, rs};
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const int next = tile + G;
    c.has_next = next < tiles;
    c.xnext = row_ptr(c.has_next ? next : tile);
    c.snext = st_ptr(c.has_next ? next : tile);

    set_step(-1);
    qw_pro<DT, 0>(acc[1], acc[0], c);                              // MFMAs of chunk 0 -> acc[0]; pieces of chunk 1
#if defined(QW_DBG) && QW_DBG == 2
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); return;
#endif
    set_step(0);
    qw_qq<DT, 0>(acc[0], acc[1], c);                               // (no stores behind pro's pieces)
#if defined(QW_DBG) && QW_DBG == 3
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); return;
#endif
    for (int i = 1; i < QW_VCH0 - 1; i += 2) {                     // steps 1 .. 8
      set_step(i);
      qw_qq<DT, 4>(acc[1], acc[0], c);
      set_step(i + 1);
      qw_qq<DT, 4>(acc[0], acc[1], c);
    }
    set_step(QW_VCH0 - 1);
    qw_qv<DT, 4>(acc[1], acc[0], c);                               // epilogue of q | k chunk 9, MFMAs of V chunk 10
    for (int i = QW_VCH0; i < QW_NCH - 1; i += 2) {                // steps 10 .. 13
      set_step(i);
      qw_vv<DT, 4>(acc[0], acc[1], c);
      set_step(i + 1);
      qw_vv<DT, 4>(acc[1], acc[0], c);
    }
    set_step(QW_NCH - 1);
    qw_v_<DT, 4>(acc[0], acc[1], c);                               // epilogue of chunk 14; the next tile's rows go out before its stores
    g += QW_NCH;
    if (!c.has_next) break;
    tile = next;
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");               // rows + statistics landed; the 4 stores behind them may fly
  }
}

template <int DT>
int launch_qkv320w(const QwParams& p, hipStream_t s) {
  void (*kern)(const QwParams, const int) = qkv320w_kernel<DT>;
  static std::atomic<unsigned long long> attr_done{0};
  if (const int e = idf_lds_optin(reinterpret_cast<const void*>(kern), QW_SMEM, attr_done)) return e;
  const int cus = idf_num_cu();
  const int tiles = p.M / QW_BM;
  const int grid = tiles < cus ? tiles : cus;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), QW_SMEM, s, p, tiles);
  return idf_launch_status();
}

int g_qkvw_mode = -1;
inline int qkvw_mode() {
  if (g_qkvw_mode < 0) { const char* e = getenv("IDF_QKV_ROW"); g_qkvw_mode = e ? (e[0] == '0' ? 0 : 1) : 1; }
  return g_qkvw_mode;
}

}  // namespace

int idf_qkvw_set_mode(int v) {
  const int prev = qkvw_mode();
  g_qkvw_mode = v;
  return prev;
}

// idf_gemm's fused q | k | v branch tries this first; IDF_BIG_UNSUPPORTED = the shape / epilogue is not this kernel's
int idf_launch_qkv320w(const idfcore::CoreParams& p, int dtype, hipStream_t s) {
  if (qkvw_mode() == 0) return IDF_BIG_UNSUPPORTED;
  if (p.K != QW_C || p.N != QW_N || p.vt_col0 != 2 * QW_C || !p.vt_out || !p.out) return IDF_BIG_UNSUPPORTED;
  if ((p.M % QW_BM) || p.M < QW_BM * (idf_num_cu() / 2)) return IDF_BIG_UNSUPPORTED;
  if (p.epi != (IDF_EPI_BIAS | IDF_EPI_LN_ROW) || !p.ln_stats || p.stride_ln_stats || !p.ln_c || !p.bias) return IDF_BIG_UNSUPPORTED;
  if (dtype != IDF_BF16 && dtype != IDF_F16) return IDF_BIG_UNSUPPORTED;
  if (p.lda < QW_C || p.ldw < QW_C || p.ldo < 2 * QW_C || p.ld_vt < p.M) return IDF_BIG_UNSUPPORTED;
  if ((p.lda % 8) || (p.ldw % 8) || (p.ldo % 8) || (p.ld_vt % 8)) return IDF_BIG_UNSUPPORTED;
  if (!aligned16(p.A) || !aligned16(p.W) || !aligned16(p.out) || !aligned16(p.vt_out) || !aligned16(p.ln_c) || !aligned16(p.bias)) return IDF_BIG_UNSUPPORTED;
  // 32-bit per-lane offsets: W image, a tile's rows of out, 32 channel rows of V^T
  if ((long long)QW_N * p.ldw * 2 >= (1ll << 31) || (long long)QW_BM * p.ldo * 2 >= (1ll << 31) || (long long)32 * p.ld_vt * 2 >= (1ll << 31)) return IDF_BIG_UNSUPPORTED;
  QwParams q;
  q.x = p.A; q.ldx = p.lda; q.ln_stats = p.ln_stats; q.w = p.W; q.ldw = p.ldw; q.c = p.ln_c; q.d = p.bias;
  q.out = static_cast<unsigned short*>(p.out); q.ldo = p.ldo; q.vt = p.vt_out; q.ld_vt = p.ld_vt; q.M = p.M;
  return dtype == IDF_BF16 ? launch_qkv320w<IDF_BF16>(q, s) : launch_qkv320w<IDF_F16>(q, s);
}
