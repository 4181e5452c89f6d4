// Shared device helpers for libprediff_hip (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/prediff_hip.h"

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

// ---- the 16-bit MFMA operand type of this translation unit ---------------------------------------------------------------------------
// Every kernel source that touches 16-bit operands is compiled TWICE (Makefile): as is -- bfloat16 operands, namespace pdk_bf16, the public
// pd_* entry points -- and with -DPD_BUILD_F16 -- IEEE half operands (10-bit mantissa: what the reference's float32_matmul_precision
// "high" = TF32 keeps on its GPUs, prediff_sevirlr_v1.yaml:63; same MFMA rate as bf16), namespace pdk_f16, entry points pd_f16_* that the
// public ones forward to when the caller's pd_call_opts / args struct says operand = PD_OPERAND_F16.  The two builds share every line;
// the differences are the typedefs and the helpers below (conversion, the three MFMA shapes).  fp16's range is 65504: the packers
// saturate (activations behind a norm / softmax / GELU are O(1..30), the residual stream stays fp32).
#ifdef PD_BUILD_F16
#define PD_NS pdk_f16
#define PD_ENTRY(name) pd_f16_##name
#define PD_IS_F16 1
typedef _Float16 op_t;
#else
#define PD_NS pdk_bf16
#define PD_ENTRY(name) pd_##name
#define PD_IS_F16 0
typedef __bf16 op_t;
#endif
typedef __attribute__((ext_vector_type(8))) op_t op8;      // one lane's operand of a K = 32 MFMA
typedef __attribute__((ext_vector_type(4))) op_t op4v;
typedef __attribute__((ext_vector_type(2))) op_t op2v;
// bf16 TU: forward to the fp16 build of the same entry point
#if PD_IS_F16
#define PD_FORWARD_F16(is_f16, call)
#else
#define PD_FORWARD_F16(is_f16, call) do { if (is_f16) return call; } while (0)
#endif
#define PD_OPTS_F16(opts) ((opts) && (opts)->operand == PD_OPERAND_F16)

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

// compute units of the current device (MI355X: 256), asked once per device: grid sizes and form heuristics derive from it
static inline int pd_num_cus() {
  static int n_dev[PD_MAX_DEVICES];
  int& n = n_dev[pd_cur_device()];
  if (n <= 0) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, pd_cur_device()) != hipSuccess || v <= 0) v = 256;
    n = v;
  }
  return n;
}

#define PD_CHECK_LAUNCH()                                                 \
  do {                                                                    \
    hipError_t e__ = hipGetLastError();                                   \
    if (e__ != hipSuccess) {                                              \
      pd_set_error("%s:%d launch failed: %s", __FILE__, __LINE__, hipGetErrorString(e__)); \
      return PD_ERR_LAUNCH;                                               \
    }                                                                     \
  } while (0)

// fp32 -> bf16 bits, round to nearest even (NaN kept quiet).  (bf16 whatever the operand type: the hi/lo split below is bf16 only)
__device__ __forceinline__ uint16_t f2bf(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
__device__ __forceinline__ float bf2f(uint16_t h) { return __uint_as_float(((uint32_t)h) << 16); }

namespace PD_NS {
// four fp32 -> operand type (RNE): v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32 through the compiler (MFMA hazards handled).  NOT saturating: this
// is the register-to-register packer of pd_attn_ffn_pair, whose every input sits behind a LayerNorm (|y| <= sqrt(C) |gamma| + |beta|), a
// softmax ([0, 1]), a GELU of such a product or a product of two such values -- nowhere near 65504 for any weights that are not already
// broken, and its chunk loops are VALU-issue bound (two v_pk_min/max_f16 per pair of values measured 5.5 % of the fp16 step).  An overflow
// would come out as inf / NaN in the result, not silently: tests assert finiteness.  Everything that converts UNnormalised values (the
// GEMM epilogues, pd_cast_rows on the residual stream) goes through pack_op2 / f2op below, which saturate.
__device__ __forceinline__ op4v cvt_op4(const f32x4& a) { return __builtin_convertvector(a, op4v); }
// two fp32 -> packed operand pair (lo | hi << 16), round to nearest even
__device__ __forceinline__ uint32_t pack_op2(float lo, float hi) {
#if PD_IS_F16
  op2v h = __builtin_convertvector(f32x2{lo, hi}, op2v);
  h = __builtin_elementwise_min(h, (op2v)(_Float16)65504.f);
  h = __builtin_elementwise_max(h, (op2v)(_Float16)(-65504.f));
  return __builtin_bit_cast(uint32_t, h);
#else
  uint32_t r;                                   // ONE instruction (gfx950 v_cvt_pk_bf16_f32; no scalar builtin)
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
#endif
}
// one fp32 -> operand bits / back
__device__ __forceinline__ uint16_t f2op(float f) {
#if PD_IS_F16
  return __builtin_bit_cast(uint16_t, (_Float16)fminf(fmaxf(f, -65504.f), 65504.f));
#else
  return f2bf(f);
#endif
}
__device__ __forceinline__ float op2f(uint16_t h) {
#if PD_IS_F16
  return (float)__builtin_bit_cast(_Float16, h);
#else
  return bf2f(h);
#endif
}
// the three MFMA shapes in use, fp32 accumulate.  16x16x16 takes its operands as raw 16-bit lanes (s16x4) in both builds.
__device__ __forceinline__ f32x4 mfma_16x16x32(const op8& a, const op8& b, const f32x4& c) {
#if PD_IS_F16
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
#else
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
#endif
}
__device__ __forceinline__ f32x4 mfma_16x16x16(const s16x4& a, const s16x4& b, const f32x4& c) {
#if PD_IS_F16
  return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(op4v, a), __builtin_bit_cast(op4v, b), c, 0, 0, 0);
#else
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
#endif
}
__device__ __forceinline__ f32x16 mfma_32x32x16(const op8& a, const op8& b, const f32x16& c) {
#if PD_IS_F16
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
#else
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
#endif
}
}  // namespace PD_NS
using namespace PD_NS;

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

// The activation of a producer whose result is rounded to 16 bits right away (every single-pass 16-bit-operand producer: pd_igemm's
// 16-bit-only epilogues, the split-K reduce, pd_ffn_fused, pd_attn_ffn_pair): GELU in the sigmoid form -- ONE form in every such producer,
// whatever instantiation or launch path a shape / batch size selects (a layer's bits must not depend on the path: ADVICE r4).  The hi/lo
// (fp32-class) engine and fp32 outputs keep act_apply's erf form.
__device__ __forceinline__ float act_apply(float v, int act);
__device__ __forceinline__ float act_apply16(float v, int act) { return act == PD_ACT_GELU ? gelu_sigmoid(v) : act_apply(v, act); }

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
