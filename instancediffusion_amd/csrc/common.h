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
// exact-erf GELU (F.gelu default) with a branch-free erf: Abramowitz-Stegun 7.1.26, |err| <= 1.5e-7 (far below the
// 2^-9 relative rounding of the 16-bit result), ~12 VALU ops instead of ocml erff's ~40 with branches.
__device__ __forceinline__ float erf_as(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * ax * ax);
  const float r = fmaf(-p * t, e, 1.0f);
  return copysignf(r, x);
}
__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.0f + erf_as(x * 0.70710678118654752440f)); }

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
static inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }
