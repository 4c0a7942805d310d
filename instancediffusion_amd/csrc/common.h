// common.h -- shared device helpers for the gfx950 (CDNA4, wave64) kernels.  Written for MI355X only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/idf.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;

#define WAVE 64

// ---- 16-bit storage <-> f32 ------------------------------------------------------------------------------
template <int DT> struct Elem;   // DT = IDF_BF16 / IDF_F16

template <> struct Elem<IDF_BF16> {
  static __device__ __forceinline__ float to_f32(unsigned short u) { return __uint_as_float(((unsigned)u) << 16); }
  // hardware round-to-nearest-even conversions (gfx950 v_cvt_pk_bf16_f32)
  static __device__ __forceinline__ unsigned short from_f32(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }
  static __device__ __forceinline__ unsigned pack(float lo, float hi) {
    f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
  }
  static __device__ __forceinline__ f32x16 mfma32(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  }
  // The same MFMA as inline assembly with the accumulator pinned to the AGPR (AG) or the VGPR file: the 4-wave experiment of
  // gemm_big.hip holds 320 accumulator registers per lane, more than either file, and the register allocator left to itself
  // shuttles them between the files inside the K loop (640 v_accvgpr moves per K-tile).
  template <bool AG> static __device__ __forceinline__ void mfma32_pin(u32x4 a, u32x4 b, f32x16& c) {
    if constexpr (AG) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
  }
  static constexpr unsigned ONES2 = 0x3f803f80u;                 // (1.0, 1.0)
  // acc += a.lo * b.lo + a.hi * b.hi on the packed pair (one VALU op per two elements)
  static __device__ __forceinline__ void dot2c(float& acc, unsigned a, unsigned b) {
    asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(acc) : "v"(a), "v"(b));
  }
};

template <> struct Elem<IDF_F16> {
  static __device__ __forceinline__ float to_f32(unsigned short u) { return (float)__builtin_bit_cast(_Float16, u); }
  static __device__ __forceinline__ unsigned short from_f32(float f) { return __builtin_bit_cast(unsigned short, (_Float16)f); }
  static __device__ __forceinline__ unsigned pack(float lo, float hi) {
    f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2_t));
  }
  static __device__ __forceinline__ f32x16 mfma32(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
  }
  template <bool AG> static __device__ __forceinline__ void mfma32_pin(u32x4 a, u32x4 b, f32x16& c) {
    if constexpr (AG) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
  }
  static constexpr unsigned ONES2 = 0x3c003c00u;
  static __device__ __forceinline__ void dot2c(float& acc, unsigned a, unsigned b) {
    asm("v_dot2c_f32_f16 %0, %1, %2" : "+v"(acc) : "v"(a), "v"(b));
  }
};

template <int DT> __device__ __forceinline__ unsigned pack2(float lo, float hi) { return Elem<DT>::pack(lo, hi); }
template <int DT> __device__ __forceinline__ void unpack8(u32x4 v, float* f) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f[2 * i] = Elem<DT>::to_f32((unsigned short)(v[i] & 0xffffu));
    f[2 * i + 1] = Elem<DT>::to_f32((unsigned short)(v[i] >> 16));
  }
}
template <int DT> __device__ __forceinline__ u32x4 pack8(const float* f) {
  u32x4 v;
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = pack2<DT>(f[2 * i], f[2 * i + 1]);
  return v;
}

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }
// erf-GELU (F.gelu default, attention.py:43) as x * sigmoid(p(x)) with p the odd degree-5 minimax fit of logit(Phi(x)):
//   |gelu_erf_f(x) - x Phi(x)| <= 2.6e-5 for every x (fit on [-8, 8], tools/fit_gelu.py; outside, Phi is 0 / 1 to 1e-15 and
//   the argument is clamped), i.e. ~1 % of the 2^-9 relative rounding of a 16-bit result of magnitude 1.
// 9 VALU ops (2 transcendental) instead of the ~18 of an Abramowitz-Stegun erf: the GEGLU epilogue of the K = 320 layers
// was 2.6x as long as their MFMA loop (profiles/DESIGN_r01_r05_full.md section 4).  The coefficients carry the factor -log2(e) of exp -> exp2.
__device__ __forceinline__ float gelu_erf_f(float x) {
  const float xc = __builtin_amdgcn_fmed3f(x, -8.0f, 8.0f);
  const float x2 = xc * xc;
  float q = fmaf(x2, 1.0142652e-3f, -1.0677574e-1f);        // -log2(e) * (-7.03035068e-4, 7.40113019e-2, 1.59501576)
  q = fmaf(q, x2, -2.3011213f);
  const float e = __builtin_amdgcn_exp2f(q * xc);            // e^(-p(x))
  return x * __builtin_amdgcn_rcpf(1.0f + e);
}

// wave64 butterfly reductions
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

static inline int idf_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}
// One-time opt-in of a kernel to more than 64 KB of dynamic LDS (hipFuncAttributeMaxDynamicSharedMemorySize), PER DEVICE and
// thread-safe: `done` is the kernel's own static bit mask of devices already served (the process-wide `static bool`
// this replaces was neither -- a second device of a multi-device process launched without the attribute).
#include <atomic>
static inline int idf_lds_optin(const void* kern, int bytes, std::atomic<unsigned long long>& done) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = -1;
  if (dev >= 0 && ((done.load(std::memory_order_relaxed) >> dev) & 1ull)) return 0;
  const hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) return (int)e;
  if (dev >= 0) done.fetch_or(1ull << dev, std::memory_order_relaxed);
  return 0;
}
static inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }
