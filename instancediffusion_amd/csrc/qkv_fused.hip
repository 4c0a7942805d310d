// qkv_fused.hip -- the fused q | k | v projection of a C = 320 transformer block with the activation rows RESIDENT IN REGISTERS
// (gfx950): out[m][0 .. 639] = LN(x) . [Wq; Wk]^T, vt[c][m] = (LN(x) . Wv^T)^T   (attention.py:168-172 to_q / to_k / to_v of
// CrossAttention / SelfAttention; call site engine._self_attn; C ABI: idf_gemm with vt_out, include/idf.h).
//
// Why.  On the persistent GEMM kernel (gemm_big.hip, 256 x 320 tiles) this launch runs at 534 TF = 0.60 ms at 128 rows of 64 x 64
// latents: K = 320 is five K-tiles per output tile, so prologue, epilogue and the accumulator turn-around of EVERY tile are
// exposed, against 0.30-0.35 ms for its 1.34 GB of HBM traffic.  Here the roles are those of mlp320w_kernel (mlp_fused.hip):
//   * a workgroup is 4 waves, one per SIMD; wave w owns rows 64 w .. + 63 of a 256-row tile as TWO groups of 32 rows, whose 320
//     elements per row are loaded ONCE into 2 x 20 MFMA operand fragments in asm-owned AGPRs and serve all 960 output columns;
//     the work items (W chunk, row group) run chunk-major, so a chunk staged in LDS serves both groups: half the LDS-DMA pieces
//     and half the barriers per flop of the first, one-group build (2 400 issue cycles per 1 280 of matrix pipe, 600 of them
//     LDS-DMA issue: profiles/NOTES_r06.md);
//   * the [960][320] weight image streams through a 2-slot LDS ring in 15 chunks of 64 rows by LDS-DMA (the fused MLP's W1
//     chunk geometry: same piece roles, same swizzle, same fragment reads), one barrier per chunk, all ten pieces of a wave
//     in the step that opens the chunk;
//   * per pipeline step: the 42 MFMAs of an item (two independent 32 x 32 chains of 20 k-steps + the mean term's k-step) carry
//     the epilogue of the item before in their gaps -- rstd . acc + d in fp32, 16-bit, through a wave-private LDS staging slot so
//     that a store instruction writes whole 128-B lines (8 token rows of q | k, 8 channel rows of V^T) -- as a GENERATED
//     `asm volatile` stream (tools/gen_qkvw_stream.py -> qkvw_stream.inc; primitives mw_prims.h);
//   * V chunks (columns 640 ..) run with the MFMA operands swapped, so a lane owns a channel and its registers 32 tokens of a row
//     group: after two v_permlane32_swap per register pair a lane holds 16 consecutive tokens of its channel; the chunk's first
//     item leaves its half of the V^T rows in the staging image, the second stores the 64-token rows;
//   * the next tile's rows are fetched in the last step of a tile BEFORE its stores, and every top-of-step wait is a counted
//     vmcnt: the stream never waits for a store round trip.
// What bounds it (profiles/NOTES_r06.md section 5): 395 us per launch at 128 rows against 296 with the stores compiled out and
// 530-557 on the persistent kernel -- 1.0 GB written at 2.56 TB/s next to 0.34 GB read.
// Taken by idf_gemm when K = 320, N = 960, vt_col0 = 640, M % 256 == 0, M >= 2 x 256 x (number of CUs), LN_ROW with the
// statistics handed in (ln_stats != NULL) and BIAS; everything else stays on gemm_big.hip.  Same arithmetic per output element
// (fp32 accumulation over k = 0 .. 319 in the same order), except that the mean term -mu c[n] rides the MFMAs as a 21st k-step of
// four 16-bit products (c and -mu split hi + lo, 2^-17 each) instead of an fp32 fma per element: measured error against fp64
// identical to the persistent kernel's (1.66e-3 bf16 / 2.07e-4 fp16 rel-RMS), 0.1 % of the outputs differ by one 16-bit ulp.
// LDS: 2 x 40 KB ring + 15.5 KB (the c table as 16-bit hi | lo pairs) + 3.75 KB (d) + 1 KB (the waves' rstd tables) + 4 x 8 KB staging = 132 KB.
#include "gemm_core.h"
#include "mw_prims.h"
#include <cstdlib>
#include <atomic>

using namespace idfcore;
using namespace idfmw;

namespace {

constexpr int QW_BM = 256, QW_C = 320, QW_N = 960, QW_NCH = 15, QW_VCH0 = 10;      // chunks 10 .. 14 are V columns
constexpr int QW_SLOT = 5 * 64 * 128;                                               // one W chunk: 5 K-tiles x [64 rows][64 k]
constexpr int QW_CX_OFF = 2 * QW_SLOT, QW_CD_OFF = QW_CX_OFF + (QW_N + 33) * 16, QW_ST_OFF = QW_CD_OFF + QW_N * 4;
constexpr int QW_STG_OFF = (QW_ST_OFF + 4 * 256 + 127) & ~127;                       // (the staging swizzle XORs address bits 4..6)
constexpr int QW_SMEM = QW_STG_OFF + 4 * 8192;
constexpr int QW_NAGPR = 172;   // x fragments a0..a159 (row group r at 80 r), a160..167 the two groups' mean fragments, a168:171 the next tile's (mu, rstd)

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
  unsigned cdq, cdv, stt;                // d of the item in its epilogue (q | k: + 16 hi; V: + 4 l31); its row group's rstd table + 16 hi
  unsigned cxa;                          // the lane's row of the c table for the chunk whose MFMAs run (hi = 1 lanes: the zero row)
  unsigned qw[8], qr[4], vw[4];          // staging slot: q | k write (by 16-B slot) / read-back addresses (both images), V^T write (row group, half)
  unsigned qst[4], vstw[4];              // per-lane store offsets (bytes)
  const void* obase; const void* vtb[2]; // uniform store bases of this step
  float rstd;                            // of the lane's token row in the row group whose item is in its epilogue
  unsigned w1dst, w1_vj; const char* w1b[2];
  const unsigned short* xnext; const unsigned short* xnext2; const float* snext; bool has_next;
};

#ifndef QKVW_STREAM_INC
#define QKVW_STREAM_INC "qkvw_stream.inc"
#endif
#include QKVW_STREAM_INC

template <int DT>
__global__ __launch_bounds__(256, 1) void qkv320w_kernel(const QwParams p, const int tiles) {
  asm volatile("" ::: "a0", "a171");               // the asm-owned AGPR block (this is where the kernel descriptor learns its size)
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
  // V^T image per fragment (4 KB, the slot is 8 KB): [32 channels][128 B = the wave's 64 tokens], same swizzle; a lane of row
  // group r writes slots 4 r + 2 hi, + 1 of its channel row; read back (by the chunk's second item) as the q | k image.
  {
    const unsigned stg = smem_lds + (unsigned)(QW_STG_OFF + wave * 8192);
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
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int s = 0; s < 2; ++s) c.vw[2 * r + s] = stg + (unsigned)(l31 * 128 + (((4 * r + 2 * hi + s) ^ sw1) << 4));
#pragma unroll
    for (int i = 0; i < 4; ++i) c.vstw[i] = (unsigned)(((lane >> 3) + 8 * i) * p.ld_vt * 2 + (lane & 7) * 16);
  }
  const unsigned cd_lds = smem_lds + (unsigned)QW_CD_OFF;

  int tile = ((G & 7) == 0) ? (int)(blockIdx.x & 7) * (G >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  if (tile >= tiles) return;

  // kernel prologue: c | d of all 960 columns into LDS, W chunk 0 into ring slot 0, the first tile's rows and statistics
  // d of all 960 columns; the c table: row n = {c_hi, c_lo, c_hi, c_lo, 0, 0, 0, 0} (16-bit; c = c_hi + c_lo to 2^-17), rows 960.. = 0:
  // the W-row side of the mean term's k-step (its token side is {-mu_hi, -mu_hi, -mu_lo, -mu_lo, 0 ..}, hi = 0 lanes only)
  for (int i = tid; i < QW_N / 4; i += 256) reinterpret_cast<f32x4*>(smem + QW_CD_OFF)[i] = reinterpret_cast<const f32x4*>(p.d)[i];
  for (int n = tid; n < QW_N + 33; n += 256) {                  // (rows 960 .. 992 = 0: what the hi = 1 lanes read, + 512 for fragment 1)
    u32x4 row = {0u, 0u, 0u, 0u};
    if (n < QW_N) {
      const float cf = p.c[n];
      const unsigned short ch = Elem<DT>::from_f32(cf), cl = Elem<DT>::from_f32(cf - Elem<DT>::to_f32(ch));
      row[0] = row[1] = (unsigned)ch | ((unsigned)cl << 16);
    }
    reinterpret_cast<u32x4*>(smem + QW_CX_OFF)[n] = row;
  }
#pragma unroll
  for (int kt = 0; kt < 5; ++kt)
#pragma unroll
    for (int u = 0; u < 2; ++u)
      mw_dma_rt(c.w1b[u] + kt * 128, w1_voff, smem_lds + (unsigned)(wave * 1024 + kt * 8192 + u * 4096));
  auto row_ptr = [&](int t, int rg) { return p.x + (size_t)(t * QW_BM + wave * 64 + rg * 32 + l31) * p.ldx + 8 * hi; };
  auto st_ptr = [&](int t) { return p.ln_stats + 2 * (size_t)(t * QW_BM + wave * 64 + l31); };
  {
    const unsigned short* xr = row_ptr(tile, 0);
    const unsigned short* xr2 = row_ptr(tile, 1);
    mw_static_for<20>([&](auto kc) { mw_load_x2<decltype(kc)::value, decltype(kc)::value>(xr); mw_load_x2<20 + decltype(kc)::value, decltype(kc)::value>(xr2); });
    const float* sp = st_ptr(tile);
    asm volatile("global_load_dwordx2 a[168:169], %0, off\n\tglobal_load_dwordx2 a[170:171], %0, off offset:256" ::"v"(sp) : "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

#if defined(QW_DBG) && QW_DBG == 1
  return;
#endif
  f32x16 acc[2][2];
  float rstd2[2];
  const unsigned cx_lds = smem_lds + (unsigned)QW_CX_OFF;
  int g = 0;                                         // chunks streamed so far: chunk j of this tile sits in ring slot (g + j) & 1
  // step s of a tile: epilogue of item s = (chunk s / 2, row group s % 2); MFMAs of item s + 1, which read chunk cm = (s + 1) / 2
  // [slot (g + cm) & 1]; the step's LDS-DMA pieces bring half of chunk cm + 1 [the other slot]
  auto set_step = [&](int s) {
    const int cm = (s + 1) >> 1, ce = s >> 1, rg = s & 1;
    const unsigned sm = (unsigned)(((g + cm) & 1) * QW_SLOT), sd = (unsigned)(((g + cm + 1) & 1) * QW_SLOT);
#pragma unroll
    for (int k = 0; k < 4; ++k) c.w1a[k] = w1o[k] + sm;
    int j2 = cm + 1;
    if (j2 >= QW_NCH) j2 -= QW_NCH;
    c.w1_vj = w1_voff + (unsigned)j2 * w_chunk;
    c.w1dst = smem_lds + sd + (unsigned)(wave * 1024);
    // epilogue of item s
    c.cxa = hi ? cx_lds + (unsigned)(QW_N * 16) : cx_lds + (unsigned)((64 * cm + l31) * 16);
    c.cdq = cd_lds + (unsigned)(ce * 256 + 16 * hi);
    c.cdv = cd_lds + (unsigned)(ce * 256 + 4 * l31);
    c.rstd = rg ? rstd2[1] : rstd2[0];
    c.stt = smem_lds + (unsigned)(QW_ST_OFF + wave * 256 + rg * 128 + 16 * hi);
    const size_t m0 = (size_t)tile * QW_BM + wave * 64 + rg * 32;
    c.obase = reinterpret_cast<const char*>(p.out) + m0 * p.ldo * 2 + (size_t)ce * 128;
    const size_t m0w = (size_t)tile * QW_BM + wave * 64;           // (V^T rows are stored for both row groups at once)
#pragma unroll
    for (int f = 0; f < 2; ++f)
      c.vtb[f] = reinterpret_cast<const char*>(p.vt) + ((size_t)(64 * (ce - QW_VCH0) + 32 * f) * p.ld_vt + m0w) * 2;
  };

  for (;;) {
    // the rows (a0..a159) and the statistics (a168:171) of this tile have landed (the caller of this point waited for them)
    {
      const float mu0 = __uint_as_float(mw_agpr_read<168>()), rs0 = __uint_as_float(mw_agpr_read<169>());
      const float mu1 = __uint_as_float(mw_agpr_read<170>()), rs1 = __uint_as_float(mw_agpr_read<171>());
      rstd2[0] = rs0; rstd2[1] = rs1;
      asm volatile("" : "+v"(rstd2[0]), "+v"(rstd2[1]));
      // the token side of the mean term's k-step, fragments 40 / 41 of the AGPR block
      auto mean_frag = [&](float mu, auto base) {
        const unsigned short mh = Elem<DT>::from_f32(-mu), ml = Elem<DT>::from_f32(-mu - Elem<DT>::to_f32(mh));
        const unsigned d0 = hi ? 0u : ((unsigned)mh | ((unsigned)mh << 16)), d1 = hi ? 0u : ((unsigned)ml | ((unsigned)ml << 16));
        constexpr int R = decltype(base)::value;
        mw_agpr_write<R>(d0); mw_agpr_write<R + 1>(d1); mw_agpr_write<R + 2>(0u); mw_agpr_write<R + 3>(0u);
      };
      mean_frag(mu0, std::integral_constant<int, 160>{});
      mean_frag(mu1, std::integral_constant<int, 164>{});
      // the wave's rstd tables for the V items: token t of row group r at 128 r + 4 t (one half-wave writes)
      if (hi == 0) {
        *reinterpret_cast<float*>(smem + QW_ST_OFF + wave * 256 + 4 * l31) = rs0;
        *reinterpret_cast<float*>(smem + QW_ST_OFF + wave * 256 + 128 + 4 * l31) = rs1;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const int next = tile + G;
    c.has_next = next < tiles;
    c.xnext = row_ptr(c.has_next ? next : tile, 0);
    c.xnext2 = row_ptr(c.has_next ? next : tile, 1);
    c.snext = st_ptr(c.has_next ? next : tile);

    constexpr int NIT = 2 * QW_NCH, QIT = 2 * QW_VCH0;             // 30 items per tile, the first 20 of q | k chunks
    set_step(-1);
    qw_pro<DT, 0, 0>(acc[1], acc[0], c);                           // MFMAs of item 0 -> acc[0]; first half of chunk 1's pieces
    set_step(0);
    qw_qq<DT, 0, 1>(acc[0], acc[1], c);
    // (VMC of a chunk-opening step = the stores issued since ITS pieces went out two steps earlier: 4 per q | k item)
    set_step(1);
    qw_qq<DT, 4, 0>(acc[1], acc[0], c);
    set_step(2);
    qw_qq<DT, 4, 1>(acc[0], acc[1], c);
    for (int s = 3; s < QIT - 1; s += 2) {                         // steps 3 .. 18
      set_step(s);
      qw_qq<DT, 8, 0>(acc[1], acc[0], c);
      set_step(s + 1);
      qw_qq<DT, 8, 1>(acc[0], acc[1], c);
    }
    set_step(QIT - 1);
    qw_qv<DT, 8, 0>(acc[1], acc[0], c);                            // epilogue of the last q | k item, MFMAs of the first V item
    set_step(QIT);
    qw_vv1<DT, 4, 1>(acc[0], acc[1], c);                           // (a V chunk's first item leaves its half of the V^T rows in LDS: no stores)
    set_step(QIT + 1);
    qw_vv0<DT, 4, 0>(acc[1], acc[0], c);                           // (behind chunk 11's pieces: the 4 stores of step 19, none of step 20)
    set_step(QIT + 2);
    qw_vv1<DT, 0, 1>(acc[0], acc[1], c);
    for (int s = QIT + 3; s < NIT - 1; s += 2) {                   // steps 23 .. 28
      set_step(s);
      qw_vv0<DT, 8, 0>(acc[1], acc[0], c);                         // (the 8 stores of the V chunk before)
      set_step(s + 1);
      qw_vv1<DT, 0, 1>(acc[0], acc[1], c);
    }
    set_step(NIT - 1);
    qw_v_<DT, 8, 0>(acc[1], acc[0], c);                            // epilogue of item 29 (behind a barrier: the next tile's first pieces
    g += QW_NCH;                                                   // overwrite chunk 14's slot); the next tile's rows go out before its stores
    if (!c.has_next) break;
    tile = next;
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");               // rows + statistics landed; the 8 stores behind them may fly
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
  idf_qkv640w_set_mode(v);                         // one knob for both levels
  return prev;
}

// idf_gemm's fused q | k | v branch tries this first; IDF_BIG_UNSUPPORTED = the shape / epilogue is not this kernel's
int idf_launch_qkv320w(const idfcore::CoreParams& p, int dtype, hipStream_t s) {
  if (qkvw_mode() == 0) return IDF_BIG_UNSUPPORTED;
  if (p.K != QW_C || p.N != QW_N || p.vt_col0 != 2 * QW_C || !p.vt_out || !p.out) return IDF_BIG_UNSUPPORTED;
  // (from two tiles per CU: below, the 256-row tiles quantise badly on 256 CUs and the persistent kernel's 256 x 320 tiles, three
  // per row block, spread better)
  if ((p.M % QW_BM) || p.M < QW_BM * 2 * idf_num_cu()) return IDF_BIG_UNSUPPORTED;
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
