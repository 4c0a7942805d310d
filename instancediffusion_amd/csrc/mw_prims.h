// mw_prims.h -- instruction-level primitives of the one-instruction-stream-per-SIMD kernels (mlp_fused.hip mlp320w_kernel,
// qkv_fused.hip qkv320w_kernel): every one is a single `asm volatile` statement, so a stream written with them issues in source
// order (the compiler's scheduler clusters pure VALU / LDS operations in front of the MFMAs otherwise), and the registers named
// in the asm text (AGPR blocks) are invisible to the register allocator.  What the compiler then cannot do for us: s_waitcnt
// lgkmcnt in front of the first use of an LDS read (mw_wait_lgkm, placed by the stream generators), MFMA -> VALU / AGPR-read
// wait states (the streams read an accumulator one whole pipeline stage after its last MFMA; elsewhere an explicit s_nop).
#pragma once
#include "gemm_core.h"
#include <type_traits>
#include <utility>

namespace idfmw {
using namespace idfcore;

constexpr int MW_XA = 0;      // the x fragments of the lane's row: a[4 ks : 4 ks + 3], ks = 0..19 (K = 320)
constexpr int MW_OA = 80;     // (mlp320w) the ten output accumulators: a[80 + 16 a : ..]

__device__ __forceinline__ unsigned lds_u32(const void* p) { return (unsigned)(size_t)p; }

template <int R> __device__ __forceinline__ void mw_agpr_write(unsigned v) { asm volatile("v_accvgpr_write_b32 a%c1, %0" ::"v"(v), "n"(R)); }
template <int R> __device__ __forceinline__ unsigned mw_agpr_read() {
  unsigned v;
  asm volatile("v_accvgpr_read_b32 %0, a%c1" : "=v"(v) : "n"(R));
  return v;
}
template <class F, int... I>
__device__ __forceinline__ void mw_static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void mw_static_for(F&& f) { mw_static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// first product: acc (VGPRs; the activation reads them) (+)= W1 fragment (A, VGPRs) . x fragment KS (B, asm-owned AGPRs)
template <int DT, int KS, bool FIRST> __device__ __forceinline__ void mw_mf1(f32x16& acc, const u32x4& w) {
  constexpr int lo = MW_XA + 4 * KS;
  if constexpr (FIRST) {
    if constexpr (DT == IDF_BF16) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[%c2:%c3], 0" : "=&v"(acc) : "v"(w), "n"(lo), "n"(lo + 3));
    else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, a[%c2:%c3], 0" : "=&v"(acc) : "v"(w), "n"(lo), "n"(lo + 3));
  } else {
    if constexpr (DT == IDF_BF16) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[%c2:%c3], %0" : "+v"(acc) : "v"(w), "n"(lo), "n"(lo + 3));
    else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, a[%c2:%c3], %0" : "+v"(acc) : "v"(w), "n"(lo), "n"(lo + 3));
  }
}
// second product: output accumulator A (asm-owned AGPRs) += W2 fragment (A, VGPRs) . activated fragment (B, VGPRs)
template <int DT, int A> __device__ __forceinline__ void mw_mf2(const u32x4& w, const u32x4& h) {
  constexpr int lo = MW_OA + 16 * A;
  if constexpr (DT == IDF_BF16) asm volatile("v_mfma_f32_32x32x16_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(w), "v"(h), "n"(lo), "n"(lo + 15));
  else asm volatile("v_mfma_f32_32x32x16_f16 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(w), "v"(h), "n"(lo), "n"(lo + 15));
}
template <int OFF> __device__ __forceinline__ u32x4 mw_lds128(unsigned addr) {
  u32x4 r;
  asm volatile("ds_read_b128 %0, %1 offset:%c2" : "=v"(r) : "v"(addr), "n"(OFF));
  return r;
}
template <int OFF> __device__ __forceinline__ f32x4 mw_lds128f(unsigned addr) {
  f32x4 r;
  asm volatile("ds_read_b128 %0, %1 offset:%c2" : "=v"(r) : "v"(addr), "n"(OFF));
  return r;
}
template <int N> __device__ __forceinline__ void mw_wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(%c0)" ::"n"(N)); }
// Two scalar fp32 instructions per pair, NOT v_pk_*_f32: in the shadow of an MFMA a packed-fp32 instruction costs the wave ~12
// cycles, a scalar one ~5 (profiles/NOTES_r06.md: 144 packed instructions per chunk 3860 cycles per iteration, 232 scalar ones
// 3230); -DMW_PACKED_VALU builds the packed form for A/B runs.
#ifdef MW_PACKED_VALU
__device__ __forceinline__ f32x2 mw_pk_fma(f32x2 a, f32x2 b, f32x2 c) {
  f32x2 d;
  asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
__device__ __forceinline__ f32x2 mw_pk_mul(f32x2 a, f32x2 b) {
  f32x2 d;
  asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ f32x2 mw_pk_add(f32x2 a, f32x2 b) {
  f32x2 d;
  asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
#else
__device__ __forceinline__ f32x2 mw_pk_fma(f32x2 a, f32x2 b, f32x2 c) {
  float x, y;
  asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(x) : "v"(a.x), "v"(b.x), "v"(c.x));
  asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(y) : "v"(a.y), "v"(b.y), "v"(c.y));
  return f32x2{x, y};
}
__device__ __forceinline__ f32x2 mw_pk_mul(f32x2 a, f32x2 b) {
  float x, y;
  asm volatile("v_mul_f32_e32 %0, %1, %2" : "=v"(x) : "v"(a.x), "v"(b.x));
  asm volatile("v_mul_f32_e32 %0, %1, %2" : "=v"(y) : "v"(a.y), "v"(b.y));
  return f32x2{x, y};
}
__device__ __forceinline__ f32x2 mw_pk_add(f32x2 a, f32x2 b) {
  float x, y;
  asm volatile("v_add_f32_e32 %0, %1, %2" : "=v"(x) : "v"(a.x), "v"(b.x));
  asm volatile("v_add_f32_e32 %0, %1, %2" : "=v"(y) : "v"(a.y), "v"(b.y));
  return f32x2{x, y};
}
#endif
__device__ __forceinline__ float mw_med3(float x, float lo /* uniform */, float hi) {
  float d;
  asm volatile("v_med3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(x), "s"(lo), "v"(hi));
  return d;
}
__device__ __forceinline__ float mw_exp2(float x) {
  float d;
  asm volatile("v_exp_f32_e32 %0, %1" : "=v"(d) : "v"(x));
  return d;
}
__device__ __forceinline__ float mw_rcp(float x) {
  float d;
  asm volatile("v_rcp_f32_e32 %0, %1" : "=v"(d) : "v"(x));
  return d;
}
template <int DT> __device__ __forceinline__ unsigned mw_cvt_pk(float lo, float hi) {
  unsigned r;
  if constexpr (DT == IDF_BF16) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  else asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}
template <int H> __device__ __forceinline__ f32x2 mw_half(const f32x4& v) { return __builtin_shufflevector(v, v, 2 * H, 2 * H + 1); }
template <int I> __device__ __forceinline__ f32x2 mw_pair(const f32x16& v) { return __builtin_shufflevector(v, v, I, I + 1); }
// x fragment KS of the lane's row <- 16 bytes of global memory, straight into the AGPRs
template <int KS> __device__ __forceinline__ void mw_load_x(const unsigned short* rowp) {
  asm volatile("global_load_dwordx4 a[%c1:%c2], %0, off offset:%c3" ::"v"(rowp), "n"(MW_XA + 4 * KS), "n"(MW_XA + 4 * KS + 3), "n"(32 * KS) : "memory");
}

// one LDS-DMA piece as ONE statement: M0 = LDS base + LDSOFF straight from the add (no scalar temporaries), the global source =
// sbase + voff + SRCOFF through the instruction's offset field -- which moves the LDS address as well (measured: with M0 =
// base + LDSOFF the harness' fp64 check fails at 2.3e-1, with LDSOFF - SRCOFF the output is bit-identical; profiles/r06_mlpw_imm.log).
// Same ~60 cycles per piece as the s_mov / s_add form, but no scalar temporaries: the kernel's 17 SGPR spills are gone.
template <int LDSOFF, int SRCOFF> __device__ __forceinline__ void mw_dma(unsigned ldsbase /* uniform */, unsigned voff, const void* sbase /* uniform */) {
  constexpr int L = LDSOFF - SRCOFF;
  // (s_add_i32 writes SCC: without the clobber the compiler keeps a compare result alive across this statement -- qkv320w_kernel's
  // first build selected its ring slot with the carry of this add and faulted)
  asm volatile("s_add_i32 m0, %0, %c1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3 offset:%c4" ::"s"(ldsbase), "n"(L), "v"(voff), "s"(sbase), "n"(SRCOFF) : "memory", "scc");
}


// fragment KA of the AGPR block <- 16 bytes at byte offset 32 KO of the lane's row pointer (KA != KO: a second row group)
template <int KA, int KO> __device__ __forceinline__ void mw_load_x2(const unsigned short* rowp) {
  asm volatile("global_load_dwordx4 a[%c1:%c2], %0, off offset:%c3" ::"v"(rowp), "n"(MW_XA + 4 * KA), "n"(MW_XA + 4 * KA + 3), "n"(32 * KO) : "memory");
}

// ---- additions of qkv_fused.hip
// transposed first product: acc (+)= x fragment KS (A, asm-owned AGPRs) . W fragment (B, VGPRs): a lane then owns an output
// COLUMN (W row) and its registers the 32 tokens of the wave
template <int DT, int KS, bool FIRST> __device__ __forceinline__ void mw_mf1t(f32x16& acc, const u32x4& w) {
  constexpr int lo = MW_XA + 4 * KS;
  if constexpr (FIRST) {
    if constexpr (DT == IDF_BF16) asm volatile("v_mfma_f32_32x32x16_bf16 %0, a[%c2:%c3], %1, 0" : "=&v"(acc) : "v"(w), "n"(lo), "n"(lo + 3));
    else asm volatile("v_mfma_f32_32x32x16_f16 %0, a[%c2:%c3], %1, 0" : "=&v"(acc) : "v"(w), "n"(lo), "n"(lo + 3));
  } else {
    if constexpr (DT == IDF_BF16) asm volatile("v_mfma_f32_32x32x16_bf16 %0, a[%c2:%c3], %1, %0" : "+v"(acc) : "v"(w), "n"(lo), "n"(lo + 3));
    else asm volatile("v_mfma_f32_32x32x16_f16 %0, a[%c2:%c3], %1, %0" : "+v"(acc) : "v"(w), "n"(lo), "n"(lo + 3));
  }
}
__device__ __forceinline__ float mw_fma(float a, float b, float c) {
  float d;
  asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
__device__ __forceinline__ float mw_add(float a, float b) {
  float d;
  asm volatile("v_add_f32_e32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ float mw_mul(float a, float b) {
  float d;
  asm volatile("v_mul_f32_e32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
template <int OFF> __device__ __forceinline__ float mw_lds32f(unsigned addr) {
  float r;
  asm volatile("ds_read_b32 %0, %1 offset:%c2" : "=v"(r) : "v"(addr), "n"(OFF));
  return r;
}
template <int OFF> __device__ __forceinline__ void mw_lds_write64(unsigned addr, unsigned lo, unsigned hi) {
  typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
  const u32x2_t v = {lo, hi};
  asm volatile("ds_write_b64 %0, %1 offset:%c2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
}
template <int OFF> __device__ __forceinline__ void mw_lds_write128(unsigned addr, const u32x4& v) {
  asm volatile("ds_write_b128 %0, %1 offset:%c2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
}
// v_permlane32_swap: x of lanes 32..63 <-> y of lanes 0..31
__device__ __forceinline__ void mw_swap32(unsigned& x, unsigned& y) { asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(y)); }
// 16-byte store: sbase (uniform 64-bit) + voff (per lane, 32-bit) -- an asm statement so that the stream owns the vmcnt count
__device__ __forceinline__ void mw_store128(unsigned voff, const u32x4& v, const void* sbase) {
  asm volatile("global_store_dwordx4 %0, %1, %2" ::"v"(voff), "v"(v), "s"(sbase) : "memory");
}
// one LDS-DMA piece with run-time addresses (prologues): lds = LDS byte address of lane 0's 16-B slot (uniform), through M0
__device__ __forceinline__ void mw_dma_rt(const void* sbase /* uniform */, unsigned voff, unsigned lds) {
  lds = __builtin_amdgcn_readfirstlane(lds);
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds), "v"(voff), "s"(sbase) : "memory");
}
template <int N> __device__ __forceinline__ void mw_wait_vm_barrier() { asm volatile("s_waitcnt vmcnt(%c0)\n\ts_barrier" ::"n"(N) : "memory"); }

}  // namespace idfmw
