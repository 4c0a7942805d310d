// geglu_fused.hip -- the GEGLU projection of a C = 640 transformer block with the activation rows RESIDENT IN REGISTERS (gfx950):
//     out[m][j] = value * gelu(gate),   (value | gate) = LN(x) . W1^T + b1        (attention.py:36-47 GEGLU; K = 640, N = 5120,
// weight rows packed [16 value | 16 gate] per 32, LayerNorm folded: C ABI idf_gemm with IDF_EPI_BIAS | GEGLU | GEGLU_P32 | LN_ROW).
//
// Why.  On the persistent GEMM kernel (gemm_big.hip) this launch runs at 860-875 TF (0.98 ms at 128 rows of 32 x 32 latents): the
// LayerNorm fold + GELU epilogue of a 256 x 320 tile is ~27 % of a K = 640 tile with the matrix pipe idle (both waves of a SIMD are
// in it together).  Here -- the machinery of mlp320w_kernel / qkv320w_kernel (mw_prims.h, a generated `asm volatile` stream,
// tools/gen_gegluw_stream.py -> gegluw_stream.inc):
//   * a workgroup is 4 waves, one per SIMD; wave w owns rows 32 w .. + 31 of a 128-row tile, whose 640 elements per row stay in
//     40 MFMA operand fragments in asm-owned AGPRs (a0..a159) and serve all 5120 packed weight rows;
//   * the weight image streams through a 2-slot LDS ring in 160 chunks of 32 packed rows (= 16 output columns; 40 KB: ten K-tiles
//     of [32 rows][64 k], 128-B rows, 16-B slot ^= (row >> 1) & 7) by LDS-DMA, ten pieces per wave and chunk, one barrier per chunk;
//   * per pipeline step: the 40 MFMAs of chunk i + 1 (two accumulators, even / odd k-steps) carry the epilogue of chunk i in their
//     gaps: sum of the two accumulators, LayerNorm fold + bias, GEGLU in the fused MLP's x sigmoid(p(x)) form, 16-bit, 8 B per lane
//     into a wave-private staging image; every fourth chunk the image (32 rows x 128 B) is stored as whole lines;
//   * counted vmcnt everywhere; the next tile's rows go out in the tile's last step BEFORE its stores.
// Taken by idf_gemm when K = 640, N = 5120, the epilogue is exactly BIAS | GEGLU | GEGLU_P32 | LN_ROW with the statistics handed
// in, M % 128 == 0 and M >= two tiles per CU; everything else stays on gemm_big.hip.  Same arithmetic per output element as the
// persistent kernel up to the order of the K sum (two partial sums) and the fma contraction of the fold.
// LDS: 2 x 40 KB ring + 40 KB (c | d of all 5120 packed rows) + 4 x 4 KB staging = 136 KB.
#include "gemm_core.h"
#include "mw_prims.h"
#include <cstdlib>
#include <atomic>

using namespace idfcore;
using namespace idfmw;

namespace {

constexpr int GW_BM = 128, GW_K = 640, GW_N = 5120, GW_NCH = GW_N / 32;          // 160 chunks of 32 packed rows
constexpr int GW_SLOT = 10 * 32 * 128;                                            // one W chunk: 10 K-tiles x [32 rows][64 k]
constexpr int GW_CD_OFF = 2 * GW_SLOT, GW_STG_OFF = GW_CD_OFF + 2 * GW_N * 4, GW_SMEM = GW_STG_OFF + 4 * 4096;

struct GwParams {
  const unsigned short* x; int ldx;
  const float* ln_stats;                 // [M][2] (mu, rstd)
  const unsigned short* w; int ldw;      // [5120][640] gamma-folded, rows packed [16 value | 16 gate] per 32
  const float* c; const float* d;        // [5120] row sums of w; beta term + bias (packed order)
  unsigned short* out; int ldo;          // [M][>= 2560]
  int M;
};

struct GwCtx {
  unsigned w1a[4];                       // LDS byte addresses of the W fragment reads of the chunk whose MFMAs run (per lane, by ks & 3)
  unsigned cda;                          // c of the chunk in its epilogue (+ 16 hi); d at + 5120 floats
  unsigned qwj[2], qr[4], qst[4];        // staging image: this chunk's two 8-B write addresses, read-back addresses, store offsets
  const void* obase;
  float nmu, rstd, k1, k2, k3, one, lo8, hi8;
  unsigned w1dst, w1_vj; const char* wb;
  const unsigned short* xnext; const float* snext; bool has_next;
};

#ifndef GEGLUW_STREAM_INC
#define GEGLUW_STREAM_INC "gegluw_stream.inc"
#endif
#include GEGLUW_STREAM_INC

template <int DT>
__global__ __launch_bounds__(256, 1) void geglu640w_kernel(const GwParams p, const int tiles) {
  asm volatile("" ::: "a0", "a161");               // the asm-owned AGPR block: x fragments a0..a159, a160:161 the next tile's (mu, rstd)
  extern __shared__ __attribute__((aligned(128))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int G = gridDim.x;
  const unsigned smem_lds = lds_u32(smem);

  GwCtx c;
  c.k1 = 1.0142652e-3f; c.k2 = -1.0677574e-1f; c.k3 = -2.3011213f; c.one = 1.0f; c.lo8 = -8.0f; c.hi8 = 8.0f;
  asm volatile("" : "+v"(c.k1), "+v"(c.k2), "+v"(c.k3), "+v"(c.one), "+v"(c.hi8));
  // LDS-DMA roles: piece kt of wave w = rows 8 w .. + 7 of K-tile kt of the chunk: lane -> row + lane / 8, 16-B slot lane % 8
  unsigned w1_voff;
  {
    const int row = 8 * wave + (lane >> 3);
    w1_voff = (unsigned)(row * p.ldw + (((lane & 7) ^ ((row >> 1) & 7)) << 3)) * 2u;
  }
  c.wb = reinterpret_cast<const char*>(p.w);
  const unsigned w_chunk = (unsigned)(32 * p.ldw * 2);
  const int sw1 = (l31 >> 1) & 7;
  unsigned w1o[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) w1o[i] = smem_lds + (unsigned)(l31 * 128 + (((2 * i + hi) ^ sw1) << 4));
  // staging image of the wave: [32 rows][128 B = the 64 output columns of four chunks], 16-B slot ^= (row >> 1) & 7; a lane
  // writes 8 B of slots 2 jj + q (+ 8 hi inside the slot) of its row, reads back rows lane / 8 + 8 i, slot lane % 8
  const unsigned stg = smem_lds + (unsigned)(GW_STG_OFF + wave * 4096);
  const unsigned qwb = stg + (unsigned)(l31 * 128 + 8 * hi + (sw1 << 4));
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (lane >> 3) + 8 * i;
    c.qr[i] = stg + (unsigned)(row * 128 + (((lane & 7) ^ ((row >> 1) & 7)) << 4));
    c.qst[i] = (unsigned)(row * p.ldo * 2 + (lane & 7) * 16);
  }
  const unsigned cd_lds = smem_lds + (unsigned)GW_CD_OFF;

  int tile = ((G & 7) == 0) ? (int)(blockIdx.x & 7) * (G >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  if (tile >= tiles) return;

  // kernel prologue: c | d of all 5120 packed rows into LDS, W chunk 0 into ring slot 0, the first tile's rows and statistics
  for (int i = tid; i < GW_N / 4; i += 256) {
    reinterpret_cast<f32x4*>(smem + GW_CD_OFF)[i] = reinterpret_cast<const f32x4*>(p.c)[i];
    reinterpret_cast<f32x4*>(smem + GW_CD_OFF + GW_N * 4)[i] = reinterpret_cast<const f32x4*>(p.d)[i];
  }
#pragma unroll
  for (int kt = 0; kt < 10; ++kt) mw_dma_rt(c.wb + kt * 128, w1_voff, smem_lds + (unsigned)(wave * 1024 + kt * 4096));
  auto row_ptr = [&](int t) { return p.x + (size_t)(t * GW_BM + wave * 32 + l31) * p.ldx + 8 * hi; };
  auto st_ptr = [&](int t) { return p.ln_stats + 2 * (size_t)(t * GW_BM + wave * 32 + l31); };
  {
    const unsigned short* xr = row_ptr(tile);
    mw_static_for<40>([&](auto kc) { mw_load_x2<decltype(kc)::value, decltype(kc)::value>(xr); });
    const float* sp = st_ptr(tile);
    asm volatile("global_load_dwordx2 a[160:161], %0, off" ::"v"(sp) : "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  f32x16 acc[2][2];
  // step i of a tile: epilogue of chunk i; MFMAs of chunk i + 1 [ring slot (i + 1) & 1: 160 chunks per tile, the parity carries
  // over]; its LDS-DMA pieces bring chunk i + 2 [slot i & 1]
  auto set_step = [&](int i) {
    const unsigned sn = (unsigned)(((i + 1) & 1) * GW_SLOT), sj = (unsigned)((i & 1) * GW_SLOT);
#pragma unroll
    for (int k = 0; k < 4; ++k) c.w1a[k] = w1o[k] + sn;
    int j2 = i + 2;
    if (j2 >= GW_NCH) j2 -= GW_NCH;
    c.w1_vj = w1_voff + (unsigned)j2 * w_chunk;
    c.w1dst = smem_lds + sj + (unsigned)(wave * 1024);
    c.cda = cd_lds + (unsigned)((32 * i + 4 * hi) * 4);
    const int jj = i & 3;
#pragma unroll
    for (int q = 0; q < 2; ++q) c.qwj[q] = qwb ^ (unsigned)(16 * (2 * jj + q));
    c.obase = reinterpret_cast<const char*>(p.out) + ((size_t)tile * GW_BM + wave * 32) * p.ldo * 2 + (size_t)(i >> 2) * 128;
  };

  for (;;) {
    {
      const float mu = __uint_as_float(mw_agpr_read<160>()), rs = __uint_as_float(mw_agpr_read<161>());
      c.nmu = -mu; c.rstd = rs;
      asm volatile("" : "+v"(c.nmu), "+v"(c.rstd));
    }
    const int next = tile + G;
    c.has_next = next < tiles;
    c.xnext = row_ptr(c.has_next ? next : tile);
    c.snext = st_ptr(c.has_next ? next : tile);

    set_step(-1);
    gw_pro<DT, 0>(acc[1], acc[0], c);                              // MFMAs of chunk 0 -> acc[0]; the pieces of chunk 1
    // groups of four steps share a staging image; VMC of a step = the stores issued behind the pieces it waits for
    set_step(0);
    gw_step<DT, 0>(acc[0], acc[1], c);
    set_step(1);
    gw_step<DT, 0>(acc[1], acc[0], c);
    set_step(2);
    gw_step<DT, 0>(acc[0], acc[1], c);
    set_step(3);
    gw_step_st<DT, 0>(acc[1], acc[0], c);
    for (int i = 4; i < GW_NCH - 4; i += 4) {                      // steps 4 .. 155
      set_step(i);
      gw_step<DT, 4>(acc[0], acc[1], c);                           // (behind the four stores of step i - 1)
      set_step(i + 1);
      gw_step<DT, 0>(acc[1], acc[0], c);
      set_step(i + 2);
      gw_step<DT, 0>(acc[0], acc[1], c);
      set_step(i + 3);
      gw_step_st<DT, 0>(acc[1], acc[0], c);
    }
    set_step(GW_NCH - 4);
    gw_step<DT, 4>(acc[0], acc[1], c);
    set_step(GW_NCH - 3);
    gw_step<DT, 0>(acc[1], acc[0], c);
    set_step(GW_NCH - 2);
    gw_step<DT, 0>(acc[0], acc[1], c);
    set_step(GW_NCH - 1);
    gw_last<DT, 0>(acc[1], acc[0], c);                             // epilogue of chunk 159 + the store group; the next tile's rows first
    if (!c.has_next) break;
    tile = next;
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");               // rows + statistics landed; the 4 stores behind them may fly
  }
}

template <int DT>
int launch_geglu640w(const GwParams& p, hipStream_t s) {
  void (*kern)(const GwParams, const int) = geglu640w_kernel<DT>;
  static std::atomic<unsigned long long> attr_done{0};
  if (const int e = idf_lds_optin(reinterpret_cast<const void*>(kern), GW_SMEM, attr_done)) return e;
  const int cus = idf_num_cu();
  const int tiles = p.M / GW_BM;
  const int grid = tiles < cus ? tiles : cus;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), GW_SMEM, s, p, tiles);
  return idf_launch_status();
}

int g_gegluw_mode = -1;
inline int gegluw_mode() {
  if (g_gegluw_mode < 0) { const char* e = getenv("IDF_GEGLU_ROW"); g_gegluw_mode = e ? (e[0] == '0' ? 0 : 1) : 1; }
  return g_gegluw_mode;
}

}  // namespace

int idf_gegluw_set_mode(int v) {
  const int prev = gegluw_mode();
  g_gegluw_mode = v;
  return prev;
}

// idf_gemm tries this first for GEGLU launches; IDF_BIG_UNSUPPORTED = the shape / epilogue is not this kernel's
int idf_launch_geglu640w(const idfcore::CoreParams& p, int dtype, hipStream_t s) {
  if (gegluw_mode() == 0) return IDF_BIG_UNSUPPORTED;
  if (p.K != GW_K || p.N != GW_N || !p.out || p.vt_out) return IDF_BIG_UNSUPPORTED;
  if ((p.M % GW_BM) || p.M < GW_BM * 2 * idf_num_cu()) return IDF_BIG_UNSUPPORTED;
  if (p.epi != (IDF_EPI_BIAS | IDF_EPI_GEGLU | IDF_EPI_GEGLU_P32 | IDF_EPI_LN_ROW) || !p.ln_stats || p.stride_ln_stats || !p.ln_c || !p.bias) return IDF_BIG_UNSUPPORTED;
  if (dtype != IDF_BF16 && dtype != IDF_F16) return IDF_BIG_UNSUPPORTED;
  if (p.lda < GW_K || p.ldw < GW_K || p.ldo < GW_N / 2 || (p.lda % 8) || (p.ldw % 8) || (p.ldo % 8)) return IDF_BIG_UNSUPPORTED;
  if (!aligned16(p.A) || !aligned16(p.W) || !aligned16(p.out) || !aligned16(p.ln_c) || !aligned16(p.bias) || p.stat_parts || p.ln_stats_out) return IDF_BIG_UNSUPPORTED;
  if ((long long)GW_N * p.ldw * 2 >= (1ll << 31) || (long long)GW_BM * p.ldo * 2 >= (1ll << 31)) return IDF_BIG_UNSUPPORTED;
  GwParams q;
  q.x = p.A; q.ldx = p.lda; q.ln_stats = p.ln_stats; q.w = p.W; q.ldw = p.ldw; q.c = p.ln_c; q.d = p.bias;
  q.out = static_cast<unsigned short*>(p.out); q.ldo = p.ldo; q.M = p.M;
  return dtype == IDF_BF16 ? launch_geglu640w<IDF_BF16>(q, s) : launch_geglu640w<IDF_F16>(q, s);
}
