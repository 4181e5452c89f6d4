// Small HBM-bound glue kernels of the sampling path + error plumbing of the C ABI.
#include <stdarg.h>
#include "common.h"

static thread_local char g_err[512] = "";
extern "C" void pd_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* pd_last_error(void) { return g_err; }
extern "C" int pd_abi_version(void) { return 4; }   // 2: per-call options (pd_call_opts), IEEE-half operand builds, no exported data symbols; 3: pd_call_opts.small_grid; 4: pd_cuboid_attn_args.qkv_fp8_log2 (was reserved)
extern "C" int pd_sizeof_igemm_args(void) { return (int)sizeof(pd_igemm_args); }
extern "C" int pd_sizeof_cuboid_attn_args(void) { return (int)sizeof(pd_cuboid_attn_args); }
extern "C" int pd_sizeof_call_opts(void) { return (int)sizeof(pd_call_opts); }

static inline unsigned grid_for(int64_t n) { return (unsigned)min((int64_t)8192, (n + 255) / 256); }

// ---- cat([cond, x], T) + observation indicator channel (cuboid_transformer_unet.py:425-428) ----
__global__ void __launch_bounds__(256) build_input_kernel(const float* __restrict__ x, const float* __restrict__ cond,
                                                          float* __restrict__ out, int B, int T_in, int T_out, int HW, int C, int ld_out) {
  const int T = T_in + T_out;
  const int64_t total = (int64_t)B * T * HW * ld_out;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % ld_out);
    const int64_t pos = i / ld_out;
    const int s = (int)(pos % HW);
    const int t = (int)((pos / HW) % T);
    const int64_t b = pos / ((int64_t)HW * T);
    float v = 0.f;
    if (c < C)
      v = t < T_in ? cond[((b * T_in + t) * HW + s) * C + c] : x[((b * T_out + (t - T_in)) * HW + s) * C + c];
    else if (c == C)
      v = t < T_in ? 1.f : 0.f;
    out[i] = v;
  }
}
extern "C" int pd_unet_build_input(const float* x, const float* cond, float* out, int B, int T_in, int T_out, int HW, int C, int ld_out,
                                   pd_stream_t stream) {
  PD_CHECK_ARG(x && cond && out && ld_out >= C + 1, "pd_unet_build_input: bad args");
  const int64_t total = (int64_t)B * (T_in + T_out) * HW * ld_out;
  hipLaunchKernelGGL(build_input_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, cond, out, B, T_in, T_out, HW, C, ld_out);
  PD_CHECK_LAUNCH();
  return PD_OK;
}

// ---- sinusoidal timestep embedding, [cos | sin] (models/utils.py:68-88) ----
__global__ void timestep_embedding_kernel(const int64_t* __restrict__ t, const float* __restrict__ freqs, float* __restrict__ out, int B, int dim) {
  const int half = dim / 2;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * dim) return;
  const int b = i / dim, c = i - b * dim;
  float v = 0.f;
  if (c < 2 * half) {
    const int k = c < half ? c : c - half;
    const float arg = (float)t[b] * freqs[k];
    v = c < half ? cosf(arg) : sinf(arg);
  }
  out[i] = v;
}
extern "C" int pd_timestep_embedding(const int64_t* t, const float* freqs, float* out, int B, int dim, pd_stream_t stream) {
  PD_CHECK_ARG(t && freqs && out && B > 0 && dim > 0, "pd_timestep_embedding: bad args");
  hipLaunchKernelGGL(timestep_embedding_kernel, dim3((B * dim + 255) / 256), dim3(256), 0, (hipStream_t)stream, t, freqs, out, B, dim);
  PD_CHECK_LAUNCH();
  return PD_OK;
}

// ---- small dense layer (M <= 64 rows): one wave per (row, output column), fp32, float4 loads ----
__global__ void __launch_bounds__(256) linear_small_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                           const float* __restrict__ bias, float* __restrict__ out, int M, int K, int N,
                                                           int act_in, int act_out) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int m = blockIdx.y;
  if (n >= N) return;
  const float* w = W + (int64_t)n * K;
  const float* xr = x + (int64_t)m * K;
  float a = 0.f;
  if ((K & 3) == 0) {
    for (int k = lane * 4; k < K; k += 256) {
      const float4 xv = *(const float4*)(xr + k), wv = *(const float4*)(w + k);
      a += act_apply(xv.x, act_in) * wv.x + act_apply(xv.y, act_in) * wv.y + act_apply(xv.z, act_in) * wv.z + act_apply(xv.w, act_in) * wv.w;
    }
  } else {
    for (int k = lane; k < K; k += 64) a += act_apply(xr[k], act_in) * w[k];
  }
  a = wave_sum(a);
  if (lane == 0) out[(int64_t)m * N + n] = act_apply(a + (bias ? bias[n] : 0.f), act_out);
}
extern "C" int pd_linear_small(const float* x, const float* W, const float* b, float* out, int M, int K, int N, int act_in, int act_out,
                               pd_stream_t stream) {
  PD_CHECK_ARG(x && W && out && M > 0 && M <= 65535 && K > 0 && N > 0, "pd_linear_small: bad args (M=%d)", M);
  hipLaunchKernelGGL(linear_small_kernel, dim3((N + 3) / 4, M), dim3(256), 0, (hipStream_t)stream, x, W, b, out, M, K, N, act_in, act_out);
  PD_CHECK_LAUNCH();
  return PD_OK;
}

// ---- x[b, s, :] += table[s, :] ----
__global__ void __launch_bounds__(256) add_rowtable_kernel(float* __restrict__ x, const float* __restrict__ table, int64_t total, int64_t per) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) x[i] += table[i % per];
}
extern "C" int pd_add_rowtable(float* x, const float* table, int64_t n_samples, int rows_per_sample, int C, pd_stream_t stream) {
  PD_CHECK_ARG(x && table, "pd_add_rowtable: null");
  const int64_t per = (int64_t)rows_per_sample * C, total = per * n_samples;
  hipLaunchKernelGGL(add_rowtable_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, table, total, per);
  PD_CHECK_LAUNCH();
  return PD_OK;
}

__global__ void __launch_bounds__(256) add_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) out[i] = a[i] + b[i];
}
extern "C" int pd_add(const float* a, const float* b, float* out, int64_t n, pd_stream_t stream) {
  PD_CHECK_ARG(a && b && out, "pd_add: null");
  hipLaunchKernelGGL(add_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, a, b, out, n);
  PD_CHECK_LAUNCH();
  return PD_OK;
}

// ---- DDPM ancestral step (latent_diffusion.py:553-566, 592-596, 620-631) ----
__global__ void __launch_bounds__(256) ddpm_step_kernel(const float* __restrict__ zt, const float* __restrict__ eps,
                                                        const float* __restrict__ noise, const float* __restrict__ shift,
                                                        const int64_t* __restrict__ t, const float* __restrict__ coef, int T,
                                                        float* __restrict__ out, int64_t per, float temperature, int clip) {
  const int b = blockIdx.y;
  const int tt = (int)t[b];
  const float c_recip = coef[tt], c_recipm1 = coef[T + tt], c1 = coef[2 * T + tt], c2 = coef[3 * T + tt];
  const float sigma = expf(0.5f * coef[4 * T + tt]);
  const float nz = tt != 0 ? 1.f : 0.f;
  const int64_t base = (int64_t)b * per;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < per; i += (int64_t)gridDim.x * 256) {
    const float z = zt[base + i];
    float z0 = c_recip * z - c_recipm1 * eps[base + i];
    if (clip) z0 = fminf(1.f, fmaxf(-1.f, z0));
    float mean = c1 * z0 + c2 * z;
    if (shift) mean = mean - sigma * shift[base + i];
    out[base + i] = mean + nz * sigma * (noise[base + i] * temperature);
  }
}
extern "C" int pd_ddpm_step(const float* zt, const float* eps, const float* noise, const float* mean_shift, const int64_t* t,
                            const float* coef, int T, float* out, int B, int64_t per_sample, float temperature, int clip_denoised,
                            pd_stream_t stream) {
  PD_CHECK_ARG(zt && eps && noise && t && coef && out && B > 0, "pd_ddpm_step: bad args");
  hipLaunchKernelGGL(ddpm_step_kernel, dim3(grid_for(per_sample), B), dim3(256), 0, (hipStream_t)stream, zt, eps, noise, mean_shift, t, coef,
                     T, out, per_sample, temperature, clip_denoised);
  PD_CHECK_LAUNCH();
  return PD_OK;
}

// ---- DDIM step (parity-unpinned; see header) ----
__global__ void __launch_bounds__(256) ddim_step_kernel(const float* __restrict__ zt, const float* __restrict__ eps,
                                                        const float* __restrict__ noise, const float* __restrict__ coef,
                                                        float* __restrict__ out, int64_t per) {
  const int b = blockIdx.y;
  const float a_t = coef[b * 3], a_prev = coef[b * 3 + 1], sigma = coef[b * 3 + 2];
  const float s1 = sqrtf(1.f - a_t), r = 1.f / sqrtf(a_t), sp = sqrtf(a_prev);
  const float dir = sqrtf(fmaxf(0.f, 1.f - a_prev - sigma * sigma));
  const int64_t base = (int64_t)b * per;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < per; i += (int64_t)gridDim.x * 256) {
    const float e = eps[base + i];
    const float z0 = (zt[base + i] - s1 * e) * r;
    out[base + i] = sp * z0 + dir * e + (noise ? sigma * noise[base + i] : 0.f);
  }
}
extern "C" int pd_ddim_step(const float* zt, const float* eps, const float* noise, const float* coef, float* out, int B,
                            int64_t per_sample, pd_stream_t stream) {
  PD_CHECK_ARG(zt && eps && coef && out && B > 0, "pd_ddim_step: bad args");
  hipLaunchKernelGGL(ddim_step_kernel, dim3(grid_for(per_sample), B), dim3(256), 0, (hipStream_t)stream, zt, eps, noise, coef, out, per_sample);
  PD_CHECK_LAUNCH();
  return PD_OK;
}

// ---- NCHW <-> NHWC (fp32) ----
__global__ void __launch_bounds__(256) nchw_to_nhwc_kernel(const float* __restrict__ x, float* __restrict__ out, int N, int C, int HW, int ld) {
  const int64_t total = (int64_t)N * HW * ld;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % ld);
    const int64_t ns = i / ld;
    const int s = (int)(ns % HW);
    const int64_t n = ns / HW;
    out[i] = c < C ? x[(n * C + c) * HW + s] : 0.f;
  }
}
__global__ void __launch_bounds__(256) nhwc_to_nchw_kernel(const float* __restrict__ x, float* __restrict__ out, int N, int C, int HW, int ld) {
  const int64_t total = (int64_t)N * C * HW;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int s = (int)(i % HW);
    const int64_t nc = i / HW;
    const int c = (int)(nc % C);
    const int64_t n = nc / C;
    out[i] = x[(n * HW + s) * ld + c];
  }
}
extern "C" int pd_nchw_to_nhwc(const float* x, float* out, int N, int C, int HW, int ld_out, pd_stream_t stream) {
  PD_CHECK_ARG(x && out && ld_out >= C, "pd_nchw_to_nhwc: bad args");
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(grid_for((int64_t)N * HW * ld_out)), dim3(256), 0, (hipStream_t)stream, x, out, N, C, HW, ld_out);
  PD_CHECK_LAUNCH();
  return PD_OK;
}
extern "C" int pd_nhwc_to_nchw(const float* x, float* out, int N, int C, int HW, int ld_in, pd_stream_t stream) {
  PD_CHECK_ARG(x && out && ld_in >= C, "pd_nhwc_to_nchw: bad args");
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(grid_for((int64_t)N * C * HW)), dim3(256), 0, (hipStream_t)stream, x, out, N, C, HW, ld_in);
  PD_CHECK_LAUNCH();
  return PD_OK;
}
