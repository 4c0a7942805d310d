// qkv640_fused.hip -- the fused q | k | v projection of a C = 640 transformer block with the activation rows RESIDENT IN REGISTERS
// (gfx950): out[m][0 .. 1279] = LN(x) . [Wq; Wk]^T, vt[c][m] = (LN(x) . Wv^T)^T   (attention.py:168-172; C ABI idf_gemm with vt_out).
// The skeleton of geglu640w_kernel (geglu_fused.hip: 4 waves x 32 rows, 40 x fragments in asm-owned AGPRs, the [1920][640] weight
// image through a 2-slot LDS ring in 60 chunks of 32 rows, 40 MFMAs per chunk on two accumulators, one barrier per chunk, counted
// vmcnt) with the epilogues of qkv320w_kernel (qkv_fused.hip): q | k chunks -- fold + bias, 16-bit, 8 B per lane into a staging
// image [32 tokens][128 B] of two chunks, stored as whole lines; V chunks with the MFMA operands swapped -- a lane owns a channel
// and the wave's 32 tokens -- staged [32 channels][64 B], 2 stores per chunk.  Stream: tools/gen_qkv640w_stream.py.
// Taken by idf_gemm when K = 640, N = 1920, vt_col0 = 1280, M % 128 == 0, M >= two tiles per CU, LN_ROW with the statistics
// handed in and BIAS (on the persistent kernel: 843 TF at 128 rows of 32 x 32 latents).
// LDS: 2 x 40 KB ring + 15 KB (c | d) + 1 KB (the waves' (-mu, rstd) tables) + 4 x 4 KB staging = 112 KB.
#include "gemm_core.h"
#include "mw_prims.h"
#include <cstdlib>
#include <atomic>

using namespace idfcore;
using namespace idfmw;

namespace {

constexpr int GW_BM = 128, GW_K = 640, GW_N = 1920, GW_NCH = GW_N / 32, GW_VCH0 = 40;    // 60 chunks of 32 rows; chunks 40 .. 59 are V columns
constexpr int GW_SLOT = 10 * 32 * 128;                                            // one W chunk: 10 K-tiles x [32 rows][64 k]
constexpr int GW_CD_OFF = 2 * GW_SLOT, GW_ST_OFF = GW_CD_OFF + 2 * GW_N * 4, GW_STG_OFF = GW_ST_OFF + 4 * 256, GW_SMEM = GW_STG_OFF + 4 * 4096;

struct QmParams {
  const unsigned short* x; int ldx;
  const float* ln_stats;                 // [M][2] (mu, rstd)
  const unsigned short* w; int ldw;      // [1920][640] gamma-folded
  const float* c; const float* d;        // [1920] row sums of w; beta term + bias
  unsigned short* out; int ldo;          // [M][>= 1280]
  unsigned short* vt; int ld_vt;         // [640][>= M]
  int M;
};

struct QmCtx {
  unsigned w1a[4];                       // LDS byte addresses of the W fragment reads of the chunk whose MFMAs run (per lane, by ks & 3)
  unsigned cdq, cdv, stt;                // c of the chunk in its epilogue (q | k: + 16 hi; V: + 4 l31), d at + 1920 floats; the wave's (-mu, rstd) table + 32 hi
  unsigned qwj[4], qr[4], qst[4];        // q | k staging image: this chunk's four 8-B write addresses, read-back addresses, store offsets
  unsigned vw[2], vr[2], vst[2];         // V^T staging image / store offsets
  const void* obase; const void* vtb;
  float nmu, rstd;
  unsigned w1dst, w1_vj; const char* wb;
  const unsigned short* xnext; const float* snext; bool has_next;
};

#ifndef QKV640W_STREAM_INC
#define QKV640W_STREAM_INC "qkv640w_stream.inc"
#endif
#include QKV640W_STREAM_INC

template <int DT>
__global__ __launch_bounds__(256, 1) void qkv640w_kernel(const QmParams p, const int tiles) {
  asm volatile("" ::: "a0", "a161");               // the asm-owned AGPR block: x fragments a0..a159, a160:161 the next tile's (mu, rstd)
  extern __shared__ __attribute__((aligned(128))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int G = gridDim.x;
  const unsigned smem_lds = lds_u32(smem);

  QmCtx c;
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
  // staging slot of the wave (4 KB).  q | k image: [32 tokens][128 B = the 64 columns of two chunks], 16-B slot ^= (row >> 1) & 7; a
  // lane writes 8 B of slots 4 jj + q (+ 8 hi inside the slot) of its row, reads back rows lane / 8 + 8 i, slot lane % 8.
  // V^T image (2 KB): [32 channels][64 B], slot ^= (row >> 2) & 3; a lane writes slots 2 hi, 2 hi + 1 of its channel row, reads back
  // rows lane / 4 + 16 i, slot lane % 4.
  const unsigned stg = smem_lds + (unsigned)(GW_STG_OFF + wave * 4096);
  const unsigned qwb = stg + (unsigned)(l31 * 128 + 8 * hi + (sw1 << 4));
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
  c.stt = smem_lds + (unsigned)(GW_ST_OFF + wave * 256 + 32 * hi);
  const unsigned cd_lds = smem_lds + (unsigned)GW_CD_OFF;

  int tile = ((G & 7) == 0) ? (int)(blockIdx.x & 7) * (G >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  if (tile >= tiles) return;

  // kernel prologue: c | d of all 1920 rows into LDS, W chunk 0 into ring slot 0, the first tile's rows and statistics
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
    c.cdq = cd_lds + (unsigned)((32 * i + 4 * hi) * 4);
    c.cdv = cd_lds + (unsigned)((32 * i + l31) * 4);
    const int jj = i & 1;
#pragma unroll
    for (int q = 0; q < 4; ++q) c.qwj[q] = qwb ^ (unsigned)(16 * (4 * jj + q));
    const size_t m0 = (size_t)tile * GW_BM + wave * 32;
    c.obase = reinterpret_cast<const char*>(p.out) + m0 * p.ldo * 2 + (size_t)(i >> 1) * 128;
    c.vtb = reinterpret_cast<const char*>(p.vt) + ((size_t)(32 * (i - GW_VCH0)) * p.ld_vt + m0) * 2;
  };

  for (;;) {
    {
      const float mu = __uint_as_float(mw_agpr_read<160>()), rs = __uint_as_float(mw_agpr_read<161>());
      c.nmu = -mu; c.rstd = rs;
      asm volatile("" : "+v"(c.nmu), "+v"(c.rstd));
      // the wave's table for the V chunks: token t -> (-mu, rstd) at 8 t (both half-waves hold the token; one writes)
      if (hi == 0) *reinterpret_cast<f32x2*>(smem + GW_ST_OFF + wave * 256 + 8 * l31) = f32x2{-mu, rs};
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const int next = tile + G;
    c.has_next = next < tiles;
    c.xnext = row_ptr(c.has_next ? next : tile);
    c.snext = st_ptr(c.has_next ? next : tile);

    set_step(-1);
    qm_pro<DT, 0>(acc[1], acc[0], c);                              // MFMAs of chunk 0 -> acc[0]; the pieces of chunk 1
    // VMC of a step = the stores the step before issued behind its pieces: 4 after a q | k store step, 2 after a V step
    set_step(0);
    qm_qq<DT, 0>(acc[0], acc[1], c);
    set_step(1);
    qm_qq_st<DT, 0>(acc[1], acc[0], c);
    for (int i = 2; i < GW_VCH0 - 2; i += 2) {                     // steps 2 .. 37
      set_step(i);
      qm_qq<DT, 4>(acc[0], acc[1], c);
      set_step(i + 1);
      qm_qq_st<DT, 0>(acc[1], acc[0], c);
    }
    set_step(GW_VCH0 - 2);
    qm_qq<DT, 4>(acc[0], acc[1], c);
    set_step(GW_VCH0 - 1);
    qm_qv_st<DT, 0>(acc[1], acc[0], c);                            // epilogue of the last q | k chunk (+ store group), MFMAs of the first V chunk
    set_step(GW_VCH0);
    qm_vv<DT, 4>(acc[0], acc[1], c);
    for (int i = GW_VCH0 + 1; i < GW_NCH - 1; i += 2) {            // steps 41 .. 58
      set_step(i);
      qm_vv<DT, 2>(acc[1], acc[0], c);
      set_step(i + 1);
      qm_vv<DT, 2>(acc[0], acc[1], c);
    }
    set_step(GW_NCH - 1);
    qm_last<DT, 2>(acc[1], acc[0], c);                             // epilogue of chunk 59; the next tile's rows go out before its 2 stores
    if (!c.has_next) break;
    tile = next;
    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");               // rows + statistics landed; the 2 stores behind them may fly
  }
}

template <int DT>
int launch_qkv640w(const QmParams& p, hipStream_t s) {
  void (*kern)(const QmParams, const int) = qkv640w_kernel<DT>;
  static std::atomic<unsigned long long> attr_done{0};
  if (const int e = idf_lds_optin(reinterpret_cast<const void*>(kern), GW_SMEM, attr_done)) return e;
  const int cus = idf_num_cu();
  const int tiles = p.M / GW_BM;
  const int grid = tiles < cus ? tiles : cus;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), GW_SMEM, s, p, tiles);
  return idf_launch_status();
}

int g_qkv640w_mode = -1;
inline int qkv640w_mode() {
  if (g_qkv640w_mode < 0) { const char* e = getenv("IDF_QKV_ROW"); g_qkv640w_mode = e ? (e[0] == '0' ? 0 : 1) : 1; }
  return g_qkv640w_mode;
}

}  // namespace

int idf_qkv640w_set_mode(int v) {
  const int prev = qkv640w_mode();
  g_qkv640w_mode = v;
  return prev;
}

// idf_gemm's fused q | k | v branch tries this for the C = 640 level; IDF_BIG_UNSUPPORTED = the shape / epilogue is not this kernel's
int idf_launch_qkv640w(const idfcore::CoreParams& p, int dtype, hipStream_t s) {
  if (qkv640w_mode() == 0) return IDF_BIG_UNSUPPORTED;
  if (p.K != GW_K || p.N != GW_N || p.vt_col0 != 2 * GW_K || !p.vt_out || !p.out) return IDF_BIG_UNSUPPORTED;
  if ((p.M % GW_BM) || p.M < GW_BM * 2 * idf_num_cu()) return IDF_BIG_UNSUPPORTED;
  if (p.epi != (IDF_EPI_BIAS | IDF_EPI_LN_ROW) || !p.ln_stats || p.stride_ln_stats || !p.ln_c || !p.bias) return IDF_BIG_UNSUPPORTED;
  if (dtype != IDF_BF16 && dtype != IDF_F16) return IDF_BIG_UNSUPPORTED;
  if (p.lda < GW_K || p.ldw < GW_K || p.ldo < 2 * GW_K || p.ld_vt < p.M || (p.lda % 8) || (p.ldw % 8) || (p.ldo % 8) || (p.ld_vt % 8)) return IDF_BIG_UNSUPPORTED;
  if (!aligned16(p.A) || !aligned16(p.W) || !aligned16(p.out) || !aligned16(p.vt_out) || !aligned16(p.ln_c) || !aligned16(p.bias)) return IDF_BIG_UNSUPPORTED;
  if ((long long)GW_N * p.ldw * 2 >= (1ll << 31) || (long long)GW_BM * p.ldo * 2 >= (1ll << 31) || (long long)32 * p.ld_vt * 2 >= (1ll << 31)) return IDF_BIG_UNSUPPORTED;
  QmParams q;
  q.x = p.A; q.ldx = p.lda; q.ln_stats = p.ln_stats; q.w = p.W; q.ldw = p.ldw; q.c = p.ln_c; q.d = p.bias;
  q.out = static_cast<unsigned short*>(p.out); q.ldo = p.ldo; q.vt = p.vt_out; q.ld_vt = p.ld_vt; q.M = p.M;
  return dtype == IDF_BF16 ? launch_qkv640w<IDF_BF16>(q, s) : launch_qkv640w<IDF_F16>(q, s);
}
