// Shared device helpers for libprediff_hip (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/prediff_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

extern "C" void pd_set_error(const char* fmt, ...);

#define PD_CHECK_ARG(cond, ...)                       \
  do {                                                \
    if (!(cond)) {                                    \
      pd_set_error(__VA_ARGS__);                      \
      return PD_ERR_ARG;                              \
    }                                                 \
  } while (0)

// hipFuncSetAttribute is per device: one process may drive modules on several GPUs, so the "already raised the dynamic-LDS limit"
// flags of the launch helpers are kept per device (indexed by the current device of the calling thread).
#define PD_MAX_DEVICES 64
static inline int pd_cur_device() {
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= PD_MAX_DEVICES) d = 0;
  return d;
}

#define PD_CHECK_LAUNCH()                                                 \
  do {                                                                    \
    hipError_t e__ = hipGetLastError();                                   \
    if (e__ != hipSuccess) {                                              \
      pd_set_error("%s:%d launch failed: %s", __FILE__, __LINE__, hipGetErrorString(e__)); \
      return PD_ERR_LAUNCH;                                               \
    }                                                                     \
  } while (0)

// fp32 -> bf16 bits, round to nearest even (NaN kept quiet).
__device__ __forceinline__ uint16_t f2bf(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
// two fp32 -> packed bf16x2 (lo | hi << 16), round to nearest even, ONE instruction (gfx950 v_cvt_pk_bf16_f32; no builtin)
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  uint32_t r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}
__device__ __forceinline__ float bf2f(uint16_t h) { return __uint_as_float(((uint32_t)h) << 16); }

// hi/lo bf16 decomposition: x ~= hi + lo with ~16 mantissa bits.
__device__ __forceinline__ void f2bf_split(float f, uint16_t& hi, uint16_t& lo) {
  hi = f2bf(f);
  lo = f2bf(f - bf2f(hi));
}

// erf with |abs error| <= 1.5e-7 (Abramowitz-Stegun 7.1.26): 1 rcp + 1 exp2 + 5 fma instead of the ~40-instruction libm erff;
// the epilogue of the FFN GEMM evaluates it 64x per lane.
__device__ __forceinline__ float fast_erf(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
  float y = fmaf(1.061405429f, t, -1.453152027f);
  y = fmaf(y, t, 1.421413741f);
  y = fmaf(y, t, -0.284496736f);
  y = fmaf(y, t, 0.254829592f);
  y = y * t * __builtin_amdgcn_exp2f(-ax * ax * 1.4426950408889634f);
  return copysignf(1.0f - y, x);
}

// GELU as x * sigmoid(x (a + b x^2 + c x^4)) -- 9 VALU instructions (2 transcendental) instead of the 16 of the erf form above; minimax fit
// against 0.5 x (1 + erf(x / sqrt 2)) on [-9, 9]: max abs error 2.5e-5 (x^2 is clamped where the sigmoid is saturated: the quartic term
// would turn the polynomial over at |x| ~ 11).  Used where the result is rounded to bf16 anyway (bf16 engine: the FFN-1 epilogue of
// pd_igemm and pd_attn_ffn_pair); the hi/lo (fp32-class) engine keeps the erf form.
__device__ __forceinline__ float gelu_sigmoid_arg(float x) {   // the argument of exp2 (first stage of the software-pipelined form)
  constexpr float L2E = -1.4426950408889634f;
  const float x2 = fminf(x * x, 52.0f);
  float pl = fmaf(x2, L2E * -7.03034059e-04f, L2E * 7.40112943e-02f);
  pl = fmaf(x2, pl, L2E * 1.59501577f);
  return x * pl;
}
__device__ __forceinline__ float gelu_sigmoid(float x) {
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(gelu_sigmoid_arg(x)));
}

__device__ __forceinline__ float act_apply(float v, int act) {
  switch (act) {
    case PD_ACT_GELU: return 0.5f * v * (1.0f + fast_erf(v * 0.70710678118654752440f));
    case PD_ACT_SILU: return v / (1.0f + expf(-v));
    case PD_ACT_LEAKY: return v > 0.f ? v : 0.1f * v;
    case PD_ACT_RELU: return v > 0.f ? v : 0.f;
    default: return v;
  }
}

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
