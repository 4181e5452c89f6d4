// Normalisation / cast kernels: fp32 residual stream in, bf16 (hi[/lo]) GEMM operands out.  All HBM-bound:
// one pass over the row (LayerNorm) or two passes over the sample (GroupNorm: stats, then apply+SiLU+cast).
#include "common.h"

namespace PD_NS {

// -------------------------------------------------------------------------------------------------
// LayerNorm over C (affine), one wave per row, two-pass in registers (exact mean / centered variance).
// cuboid_transformer.py:813 / :197 (nn.LayerNorm eps 1e-5, models/utils.py:192-221)
// -------------------------------------------------------------------------------------------------
constexpr int LN_MAXV = 16;   // up to 64*4*16 = 4096 channels per row

// NV = float4 vectors per lane per row (ceil(C/256)), R = rows handled concurrently by one wave (independent load /
// reduce chains keep R x NV 16 B loads in flight per lane: the kernel is pure HBM streaming).
// F8: the output is OCP e4m3, value * fp8_scale, saturating (v_cvt_pk_fp8_f32 rounds to nearest even), one byte per channel.
template <bool GATHER, int NV, int R, bool F8 = false>
__global__ void __launch_bounds__(256) layernorm_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, pd_bf16* __restrict__ out,
                                                        pd_bf16* __restrict__ out_lo, int64_t rows, int C, int ld_out,
                                                        float eps, float fp8_scale,
                                                        // patch-merge gather geometry (GATHER only)
                                                        int T, int H, int W, int Cs, int dt, int dh, int dw, int nearest) {
  const int lane = threadIdx.x & 63;
  const int64_t row0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * R;
  if (row0 >= rows) return;
  float4 v[R][NV];
  float s[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int64_t row = row0 + r;
    s[r] = 0.f;
    // source row decode for the patch-merge gather: row = ((b*To + to)*Ho + ho)*Wo + wo
    int64_t gb = 0;
    int to = 0, ho = 0, wo = 0;
    if (GATHER) {
      const int To = (T + dt - 1) / dt, Ho = (H + dh - 1) / dh, Wo = (W + dw - 1) / dw;
      int64_t q = row;
      wo = (int)(q % Wo); q /= Wo;
      ho = (int)(q % Ho); q /= Ho;
      to = (int)(q % To); gb = q / To;
    }
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int c = j * 256 + lane * 4;
      float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
      if (c < C && row < rows) {
        if (GATHER) {
          // element e = ((it*dh + ih)*dw + iw)*Cs + cs   (cuboid_transformer.py:286-292)
          const int sub = c / Cs, cs = c - sub * Cs;
          const int iw = sub % dw, ih = (sub / dw) % dh, it = sub / (dw * dh);
          int tt = to * dt + it, hh = ho * dh + ih, ww = wo * dw + iw;
          if (nearest) {
            // padding_type "nearest" (models/utils.py:228-256): the padded grid is F.interpolate(x, size = padded size), i.e. padded
            // coordinate p reads a source coordinate inside the tensor
            // -- with torch's own arithmetic: min(floor(p * float32(size / padded size)), size - 1), which is not the integer floor
            // (22 padded to 26: position 13 reads 10, not 11)
            const int Tp = ((T + dt - 1) / dt) * dt, Hp = ((H + dh - 1) / dh) * dh, Wp = ((W + dw - 1) / dw) * dw;
            tt = min((int)floorf((float)tt * ((float)T / (float)Tp)), T - 1);
            hh = min((int)floorf((float)hh * ((float)H / (float)Hp)), H - 1);
            ww = min((int)floorf((float)ww * ((float)W / (float)Wp)), W - 1);
          }
          if (tt < T && hh < H && ww < W)
            t = *(const float4*)(x + ((((gb * T + tt) * H + hh) * W + ww) * (int64_t)Cs + cs));
        } else {
          t = *(const float4*)(x + row * (int64_t)C + c);
        }
      }
      v[r][j] = t;
      s[r] += (t.x + t.y) + (t.z + t.w);
    }
  }
  float mean[R], rstd[R];
#pragma unroll
  for (int r = 0; r < R; ++r) mean[r] = wave_sum(s[r]) / (float)C;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int c = j * 256 + lane * 4;
      if (c < C) {
        const float a = v[r][j].x - mean[r], b = v[r][j].y - mean[r], cc = v[r][j].z - mean[r], d = v[r][j].w - mean[r];
        q += (a * a + b * b) + (cc * cc + d * d);
      }
    }
    s[r] = q;
  }
#pragma unroll
  for (int r = 0; r < R; ++r) rstd[r] = rsqrtf(wave_sum(s[r]) / (float)C + eps);
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int c = j * 256 + lane * 4;
    if (c >= ld_out) continue;
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f), be = g;
    if (c < C) { g = *(const float4*)(gamma + c); be = *(const float4*)(beta + c); }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int64_t row = row0 + r;
      if (row >= rows) continue;
      float y[4] = {0.f, 0.f, 0.f, 0.f};
      if (c < C) {
        y[0] = (v[r][j].x - mean[r]) * rstd[r] * g.x + be.x;
        y[1] = (v[r][j].y - mean[r]) * rstd[r] * g.y + be.y;
        y[2] = (v[r][j].z - mean[r]) * rstd[r] * g.z + be.z;
        y[3] = (v[r][j].w - mean[r]) * rstd[r] * g.w + be.w;
      }
      if (F8) {
#pragma unroll
        for (int k = 0; k < 4; ++k) y[k] = fminf(fmaxf(y[k] * fp8_scale, -448.f), 448.f);
        int w = __builtin_amdgcn_cvt_pk_fp8_f32(y[0], y[1], 0, false);
        w = __builtin_amdgcn_cvt_pk_fp8_f32(y[2], y[3], w, true);
        *(int*)((uint8_t*)out + row * (int64_t)ld_out + c) = w;
      } else if (out_lo) {
        uint16_t hi[4], lo[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) f2bf_split(y[k], hi[k], lo[k]);
        *(uint2*)(out + row * (int64_t)ld_out + c) = make_uint2(hi[0] | ((uint32_t)hi[1] << 16), hi[2] | ((uint32_t)hi[3] << 16));
        *(uint2*)(out_lo + row * (int64_t)ld_out + c) = make_uint2(lo[0] | ((uint32_t)lo[1] << 16), lo[2] | ((uint32_t)lo[3] << 16));
      } else {
        *(uint2*)(out + row * (int64_t)ld_out + c) =
            make_uint2(pack_op2(y[0], y[1]), pack_op2(y[2], y[3]));
      }
    }
  }
}

template <bool GATHER, bool F8 = false>
static void launch_layernorm(const float* x, const float* gamma, const float* beta, pd_bf16* out, pd_bf16* out_lo, int64_t rows, int C,
                             int ld_out, float eps, int T, int H, int W, int Cs, int dt, int dh, int dw, hipStream_t s, float fp8_scale = 0.f,
                             int nearest = 0) {
  const int nv = (C + 255) / 256;
#define PD_LN(NV, R)                                                                                                              \
  hipLaunchKernelGGL((layernorm_kernel<GATHER, NV, R, F8>), dim3((unsigned)((rows + 4 * (R) - 1) / (4 * (R)))), dim3(256), 0, s, x, gamma, \
                     beta, out, out_lo, rows, C, ld_out, eps, fp8_scale, T, H, W, Cs, dt, dh, dw, nearest)
  if (nv <= 1) PD_LN(1, 4);
  else if (nv <= 2) PD_LN(2, 2);
  else if (nv <= 4) PD_LN(4, 1);
  else if (nv <= 8) PD_LN(8, 1);
  else PD_LN(16, 1);
#undef PD_LN
}

#if !PD_IS_F16
extern "C" int pd_f16_layernorm(const float*, const float*, const float*, pd_bf16*, pd_bf16*, int64_t, int, int, float, const pd_call_opts*, pd_stream_t);
extern "C" int pd_f16_patch_merge_layernorm_ex(const float*, const float*, const float*, pd_bf16*, pd_bf16*, int, int, int, int, int, int, int, int, int,
                                               float, int, const pd_call_opts*, pd_stream_t);
extern "C" int pd_f16_groupnorm_silu(const float*, const float*, const float*, const float*, const float*, int, double*, pd_bf16*, pd_bf16*, int, int,
                                     int, int, int, float, int, const pd_call_opts*, pd_stream_t);
extern "C" int pd_f16_cast_rows(const float*, pd_bf16*, pd_bf16*, int64_t, int, int, int, int, int, int, const pd_call_opts*, pd_stream_t);
#endif

extern "C" int PD_ENTRY(layernorm)(const float* x, const float* gamma, const float* beta, pd_bf16* out, pd_bf16* out_lo,
                                   int64_t rows, int C, int ld_out, float eps, const pd_call_opts* opts, pd_stream_t stream) {
  PD_FORWARD_F16(PD_OPTS_F16(opts), pd_f16_layernorm(x, gamma, beta, out, out_lo, rows, C, ld_out, eps, opts, stream));
  PD_CHECK_ARG(!PD_IS_F16 || !out_lo, "pd_layernorm: the hi/lo split exists for bfloat16 operands only");
  PD_CHECK_ARG(x && gamma && beta && out, "pd_layernorm: null pointer");
  PD_CHECK_ARG(C > 0 && (C & 3) == 0 && C <= 256 * LN_MAXV, "pd_layernorm: C=%d must be a multiple of 4 and <= %d", C, 256 * LN_MAXV);
  PD_CHECK_ARG(ld_out >= C && (ld_out & 3) == 0 && ld_out <= ((C + 255) / 256) * 256,
               "pd_layernorm: ld_out=%d must be >= C, multiple of 4 and within the last 256-column block", ld_out);
  if (rows <= 0) return PD_OK;
  launch_layernorm<false>(x, gamma, beta, out, out_lo, rows, C, ld_out, eps, 0, 0, 0, 0, 1, 1, 1, (hipStream_t)stream);
  PD_CHECK_LAUNCH();
  return PD_OK;
}

#if !PD_IS_F16
extern "C" int pd_layernorm_fp8(const float* x, const float* gamma, const float* beta, uint8_t* out, int64_t rows, int C, int ld_out,
                                float eps, float fp8_scale, pd_stream_t stream) {
  PD_CHECK_ARG(x && gamma && beta && out && fp8_scale > 0.f, "pd_layernorm_fp8: null pointer / bad scale");
  PD_CHECK_ARG(C > 0 && (C & 3) == 0 && C <= 256 * LN_MAXV, "pd_layernorm_fp8: C=%d must be a multiple of 4 and <= %d", C, 256 * LN_MAXV);
  PD_CHECK_ARG(ld_out >= C && (ld_out & 3) == 0 && ld_out <= ((C + 255) / 256) * 256,
               "pd_layernorm_fp8: ld_out=%d must be >= C, multiple of 4 and within the last 256-column block", ld_out);
  if (rows <= 0) return PD_OK;
  launch_layernorm<false, true>(x, gamma, beta, (pd_bf16*)out, nullptr, rows, C, ld_out, eps, 0, 0, 0, 0, 1, 1, 1, (hipStream_t)stream,
                                fp8_scale);
  PD_CHECK_LAUNCH();
  return PD_OK;
}

#endif

extern "C" int PD_ENTRY(patch_merge_layernorm_ex)(const float* x, const float* gamma, const float* beta, pd_bf16* out, pd_bf16* out_lo,
                                                  int B, int T, int H, int W, int C, int dt, int dh, int dw, int ld_out, float eps,
                                                  int pad_nearest, const pd_call_opts* opts, pd_stream_t stream) {
  PD_FORWARD_F16(PD_OPTS_F16(opts), pd_f16_patch_merge_layernorm_ex(x, gamma, beta, out, out_lo, B, T, H, W, C, dt, dh, dw, ld_out, eps,
                                                                    pad_nearest, opts, stream));
  PD_CHECK_ARG(!PD_IS_F16 || !out_lo, "pd_patch_merge_layernorm: the hi/lo split exists for bfloat16 operands only");
  PD_CHECK_ARG(x && gamma && beta && out, "pd_patch_merge_layernorm: null pointer");
  const int Cm = C * dt * dh * dw;
  PD_CHECK_ARG((C & 3) == 0 && Cm <= 256 * LN_MAXV, "pd_patch_merge_layernorm: C=%d (merged %d) unsupported", C, Cm);
  PD_CHECK_ARG(ld_out >= Cm && (ld_out & 3) == 0 && ld_out <= ((Cm + 255) / 256) * 256, "pd_patch_merge_layernorm: bad ld_out=%d", ld_out);
  const int64_t rows = (int64_t)B * ((T + dt - 1) / dt) * ((H + dh - 1) / dh) * ((W + dw - 1) / dw);
  launch_layernorm<true>(x, gamma, beta, out, out_lo, rows, Cm, ld_out, eps, T, H, W, C, dt, dh, dw, (hipStream_t)stream, 0.f, pad_nearest ? 1 : 0);
  PD_CHECK_LAUNCH();
  return PD_OK;
}

#if !PD_IS_F16
extern "C" int pd_patch_merge_layernorm(const float* x, const float* gamma, const float* beta, pd_bf16* out, pd_bf16* out_lo,
                                        int B, int T, int H, int W, int C, int dt, int dh, int dw, int ld_out, float eps,
                                        pd_stream_t stream) {
  return pd_patch_merge_layernorm_ex(x, gamma, beta, out, out_lo, B, T, H, W, C, dt, dh, dw, ld_out, eps, 0, nullptr, stream);
}
#endif

// -------------------------------------------------------------------------------------------------
// GroupNorm on channels-last (B, S, C): pass 1 partial (sum, sumsq) per (sample, chunk of positions, group),
// pass 2 finalises mean/rstd per group in the block prologue (double), then normalise + affine [+ scale/shift]
// [+ SiLU] + bf16 cast.  Deterministic (no atomics).
// -------------------------------------------------------------------------------------------------
constexpr int GN_ROWS = 64;   // positions per stats block

#if !PD_IS_F16
extern "C" int pd_groupnorm_nchunk(int S, int C) { return (S + GN_ROWS - 1) / GN_ROWS; }
#else
extern "C" int pd_groupnorm_nchunk(int S, int C);
#endif

__global__ void __launch_bounds__(256) gn_stats_kernel(const float* __restrict__ x, double* __restrict__ partials, int S, int C,
                                                       int G) {
  // grid (nchunk, B).  Thread t walks the chunk's [rows x C] elements with stride 256; channel-fastest, so a
  // wave reads 256 contiguous bytes.  Per-group accumulation goes through LDS atomics on doubles? -> no: each thread
  // keeps (sum, sumsq) per *its* channel only when 256 % C == 0 or C % 256 == 0; the general case uses the slow path.
  extern __shared__ double sred[];   // [256][2] then [G][2]
  const int b = blockIdx.y, chunk = blockIdx.x, nchunk = gridDim.x;
  const int r0 = chunk * GN_ROWS, r1 = min(S, r0 + GN_ROWS);
  const int cpg = C / G;
  const float* xb = x + ((int64_t)b * S + r0) * C;
  const int n = (r1 - r0) * C;
  const int tid = threadIdx.x;
  double* out = partials + ((int64_t)b * nchunk + chunk) * G * 2;
  if ((C <= 256 && 256 % C == 0) || (C % 256 == 0)) {
    // fast path: a thread always sees channels == tid (mod 256) -> one group per thread when cpg divides nicely
    float s = 0.f, q = 0.f;
    if (C <= 256) {
      for (int i = tid; i < n; i += 256) { const float v = xb[i]; s += v; q += v * v; }
      sred[tid * 2] = s; sred[tid * 2 + 1] = q;
      __syncthreads();
      if (tid < G) {
        double ss = 0, qq = 0;
        for (int k = 0; k < 256; ++k)
          if ((k % C) / cpg == tid) { ss += sred[k * 2]; qq += sred[k * 2 + 1]; }
        out[tid * 2] = ss; out[tid * 2 + 1] = qq;
      }
    } else {
      // C multiple of 256: thread handles channels tid, tid+256, ... ; groups differ per 256-column block when cpg < 256
      const int nblk = C / 256;
      for (int k = tid; k < G * 2; k += 256) sred[512 + k] = 0.0;
      __syncthreads();
      for (int cb = 0; cb < nblk; ++cb) {
        float s2 = 0.f, q2 = 0.f;
        const int c = cb * 256 + tid;
        for (int r = 0; r < r1 - r0; ++r) { const float v = xb[(int64_t)r * C + c]; s2 += v; q2 += v * v; }
        sred[tid * 2] = s2; sred[tid * 2 + 1] = q2;
        __syncthreads();
        // groups covered by this 256-column block: [cb*256/cpg, (cb*256+255)/cpg]
        const int g0 = (cb * 256) / cpg, g1 = (cb * 256 + 255) / cpg;
        if (tid <= g1 - g0) {
          const int g = g0 + tid;
          double ss = 0, qq = 0;
          for (int k = 0; k < 256; ++k)
            if ((cb * 256 + k) / cpg == g) { ss += sred[k * 2]; qq += sred[k * 2 + 1]; }
          sred[512 + g * 2] += ss; sred[512 + g * 2 + 1] += qq;
        }
        __syncthreads();
      }
      for (int k = tid; k < G * 2; k += 256) out[k] = sred[512 + k];
    }
  } else {
    // general path (e.g. C = 65 with 65 groups): one thread per (group) walks its channels
    for (int g = tid; g < G; g += 256) {
      double ss = 0, qq = 0;
      for (int r = 0; r < r1 - r0; ++r)
        for (int c = g * cpg; c < (g + 1) * cpg; ++c) { const double v = xb[(int64_t)r * C + c]; ss += v; qq += v * v; }
      out[g * 2] = ss; out[g * 2 + 1] = qq;
    }
  }
}

__global__ void __launch_bounds__(256) gn_apply_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, const float* __restrict__ ss_scale,
                                                       const float* __restrict__ ss_shift, int ld_ss,
                                                       const double* __restrict__ partials, pd_bf16* __restrict__ out,
                                                       pd_bf16* __restrict__ out_lo, int S, int C, int G, int ld_out, float eps,
                                                       int silu, int nchunk) {
  // grid (nchunk, B): same chunking as the stats pass.
  extern __shared__ float smr[];   // [G][2] mean, rstd
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int cpg = C / G;
  const int tid = threadIdx.x;
  for (int g = tid; g < G; g += 256) {
    double ss = 0, qq = 0;
    const double* pp = partials + (int64_t)b * nchunk * G * 2 + g * 2;
    for (int k = 0; k < nchunk; ++k) { ss += pp[(int64_t)k * G * 2]; qq += pp[(int64_t)k * G * 2 + 1]; }
    const double cnt = (double)S * cpg;
    const double mean = ss / cnt;
    double var = qq / cnt - mean * mean;
    if (var < 0) var = 0;
    smr[g * 2] = (float)mean;
    smr[g * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
  }
  __syncthreads();
  const int r0 = chunk * GN_ROWS, r1 = min(S, r0 + GN_ROWS);
  const float* xb = x + ((int64_t)b * S + r0) * C;
  pd_bf16* ob = out + ((int64_t)b * S + r0) * ld_out;
  pd_bf16* obl = out_lo ? out_lo + ((int64_t)b * S + r0) * ld_out : nullptr;
  const int n = (r1 - r0) * ld_out;
  for (int i = tid; i < n; i += 256) {
    const int r = i / ld_out, c = i - r * ld_out;
    float y = 0.f;
    if (c < C) {
      const int g = c / cpg;
      y = (xb[(int64_t)r * C + c] - smr[g * 2]) * smr[g * 2 + 1] * gamma[c] + beta[c];
      if (ss_scale) y = y * (1.f + ss_scale[(int64_t)b * ld_ss + c]) + ss_shift[(int64_t)b * ld_ss + c];
      if (silu) y = y / (1.f + expf(-y));
    }
    if (obl) {
      uint16_t hi, lo;
      f2bf_split(y, hi, lo);
      ob[i] = hi; obl[i] = lo;
    } else {
      ob[i] = f2op(y);
    }
  }
}

// ---- fast path: C % 4 == 0, (C/4) divides 256, channels-per-group % 4 == 0, ld_out == C.  A thread owns one float4
// column (4 channels of ONE group) and walks rows; loads are 16 B/lane, stores 8 B/lane (bf16 x4). ----
__global__ void __launch_bounds__(256) gn_stats_vec_kernel(const float* __restrict__ x, double* __restrict__ partials, int S, int C, int G) {
  __shared__ float sred[256 * 2];
  const int b = blockIdx.y, chunk = blockIdx.x, nchunk = gridDim.x;
  const int CV = C >> 2, RP = 256 / CV;                 // float4 columns, rows per pass
  const int tid = threadIdx.x, cv = tid % CV, rr = tid / CV;
  const int r0 = chunk * GN_ROWS, r1 = min(S, r0 + GN_ROWS);
  const float4* xb = (const float4*)(x + ((int64_t)b * S + r0) * C) + cv;
  float s = 0.f, q = 0.f;
#pragma unroll 4
  for (int r = rr; r < r1 - r0; r += RP) {
    const float4 v = xb[(int64_t)r * CV];
    s += (v.x + v.y) + (v.z + v.w);
    q += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
  }
  sred[tid * 2] = s;
  sred[tid * 2 + 1] = q;
  __syncthreads();
  if (tid < G) {
    const int cvg = (C / G) >> 2;                       // float4 columns per group
    double ss = 0, qq = 0;
    for (int r = 0; r < RP; ++r)
      for (int k = 0; k < cvg; ++k) {
        const int t = r * CV + tid * cvg + k;
        ss += sred[t * 2];
        qq += sred[t * 2 + 1];
      }
    double* out = partials + ((int64_t)b * nchunk + chunk) * G * 2;
    out[tid * 2] = ss;
    out[tid * 2 + 1] = qq;
  }
}

// F8: the output is OCP e4m3, value * fp8_scale, saturating (v_cvt_pk_fp8_f32 rounds to nearest even), one byte per channel
template <bool F8>
__global__ void __launch_bounds__(256) gn_apply_vec_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const float* __restrict__ ss_scale,
                                                           const float* __restrict__ ss_shift, int ld_ss,
                                                           const double* __restrict__ partials, pd_bf16* __restrict__ out,
                                                           pd_bf16* __restrict__ out_lo, int S, int C, int G, float eps, int silu,
                                                           int nchunk, float fp8_scale) {
  __shared__ float smr[2 * 256];
  __shared__ double spart[2 * 256];
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int cpg = C / G;
  const int tid = threadIdx.x;
  // the sample's per-chunk partial sums: 256 / G threads per group take every LP-th chunk (a single thread walking all the
  // chunks is a chain of dependent L2 round trips -- 10+ us at 52 chunks, most of this kernel's time at small batches), then one
  // thread per group adds the LP strided sums in a fixed order (deterministic)
  const int LP = 256 / G;
  if (tid < LP * G) {
    const int g = tid % G, j = tid / G;
    double ss = 0, qq = 0;
    const double* pp = partials + (int64_t)b * nchunk * G * 2 + g * 2;
    for (int k = j; k < nchunk; k += LP) { ss += pp[(int64_t)k * G * 2]; qq += pp[(int64_t)k * G * 2 + 1]; }
    spart[tid * 2] = ss;
    spart[tid * 2 + 1] = qq;
  }
  __syncthreads();
  if (tid < G) {
    double ss = 0, qq = 0;
    for (int j = 0; j < LP; ++j) { ss += spart[(j * G + tid) * 2]; qq += spart[(j * G + tid) * 2 + 1]; }
    const double cnt = (double)S * cpg, mean = ss / cnt;
    double var = qq / cnt - mean * mean;
    if (var < 0) var = 0;
    smr[tid * 2] = (float)mean;
    smr[tid * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
  }
  __syncthreads();
  const int CV = C >> 2, RP = 256 / CV;
  const int cv = tid % CV, rr = tid / CV, c = cv * 4, g = c / cpg;
  const float mean = smr[g * 2], rstd = smr[g * 2 + 1];
  float4 ga = *(const float4*)(gamma + c), be = *(const float4*)(beta + c);
  // fold: y = x * a + d  with a = rstd*gamma, d = beta - mean*rstd*gamma  [then (1+scale)*y + shift]
  float a[4] = {rstd * ga.x, rstd * ga.y, rstd * ga.z, rstd * ga.w};
  float d[4] = {be.x - mean * a[0], be.y - mean * a[1], be.z - mean * a[2], be.w - mean * a[3]};
  if (ss_scale) {
    const float4 sc = *(const float4*)(ss_scale + (int64_t)b * ld_ss + c), sh = *(const float4*)(ss_shift + (int64_t)b * ld_ss + c);
    const float scv[4] = {sc.x, sc.y, sc.z, sc.w}, shv[4] = {sh.x, sh.y, sh.z, sh.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) { a[k] *= (1.f + scv[k]); d[k] = d[k] * (1.f + scv[k]) + shv[k]; }
  }
  const int r0 = chunk * GN_ROWS, r1 = min(S, r0 + GN_ROWS);
  const float4* xb = (const float4*)(x + ((int64_t)b * S + r0) * C) + cv;
  uint2* ob = (uint2*)(out + ((int64_t)b * S + r0) * C) + cv;
  uint2* obl = out_lo ? (uint2*)(out_lo + ((int64_t)b * S + r0) * C) + cv : nullptr;
  uint32_t* ob8 = (uint32_t*)((uint8_t*)out + ((int64_t)b * S + r0) * C) + cv;
#pragma unroll 4
  for (int r = rr; r < r1 - r0; r += RP) {
    const float4 v = xb[(int64_t)r * CV];
    float y[4] = {v.x * a[0] + d[0], v.y * a[1] + d[1], v.z * a[2] + d[2], v.w * a[3] + d[3]};
    if (silu) {
#pragma unroll
      for (int k = 0; k < 4; ++k) y[k] = y[k] / (1.f + __expf(-y[k]));
    }
    if (F8) {
#pragma unroll
      for (int k = 0; k < 4; ++k) y[k] = fminf(fmaxf(y[k] * fp8_scale, -448.f), 448.f);
      int w = __builtin_amdgcn_cvt_pk_fp8_f32(y[0], y[1], 0, false);
      w = __builtin_amdgcn_cvt_pk_fp8_f32(y[2], y[3], w, true);
      ob8[(int64_t)r * CV] = (uint32_t)w;
    } else if (obl) {
      uint16_t hi[4], lo[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) f2bf_split(y[k], hi[k], lo[k]);
      ob[(int64_t)r * CV] = make_uint2(hi[0] | ((uint32_t)hi[1] << 16), hi[2] | ((uint32_t)hi[3] << 16));
      obl[(int64_t)r * CV] = make_uint2(lo[0] | ((uint32_t)lo[1] << 16), lo[2] | ((uint32_t)lo[3] << 16));
    } else {
      ob[(int64_t)r * CV] = make_uint2(pack_op2(y[0], y[1]), pack_op2(y[2], y[3]));
    }
  }
}

// ---- one pass (round 4): a workgroup owns (sample, 32 or 16 channels) and keeps its S x 32 / 16 values in registers -- statistics (two-pass in
// registers: mean, then the centred sum of squares) and the normalised bf16 rows from ONE read of x instead of two (the statistics
// kernel's pass over the tensor is gone: 109 -> 0 MB per level-0 call at 32 trajectories).  NT threads, CH = 32 or 16 channels per
// workgroup: CH / 4 threads per row segment (a float4 each), RMAX sweeps over the rows (<16, 512, 32> up to 1024 rows, <26, 512, 16> up
// to 3328: 104 data registers per thread want the 256-register budget of 8 waves); the groups of the chunk (C / G = 4, 8, 16 or 32
// channels) are reduced over the row lanes of a wave by shuffles and over the waves through LDS in a fixed order.  The choice of
// this kernel depends on the shape only, never on the batch (the engine's batch-split-reproducible mode).
template <int RMAX, int NT, int CH>
__global__ void __launch_bounds__(NT) gn_onepass_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, const float* __restrict__ ss_scale,
                                                         const float* __restrict__ ss_shift, int ld_ss, double* __restrict__ partials,
                                                         int nchunk, pd_bf16* __restrict__ out, int S, int C, int G, float eps, int silu) {
  constexpr int TPR = CH / 4, NWV = NT / 64, RP = NT / TPR;          // threads per row segment, waves, rows per sweep
  __shared__ float sred[2][NWV][TPR];
  // (sample, chunk) of this workgroup.  With 16-channel chunks a 128-byte line of a row is shared by TWO workgroups: they get ids 8
  // apart -- the same XCD (workgroups go to the XCDs round robin), next to each other in its queue -- so that the line is fetched into
  // one L2 once
  int b, chunk;
  {
    const int per_row = C / CH;
    int id = blockIdx.x;
    if (CH == 16 && id < ((int)gridDim.x & ~15)) {                     // (whole blocks of 16 ids; a tail keeps its order)
      const int h = (id >> 3) & 1, p = ((id >> 4) << 3) | (id & 7);    // pair index p, half h
      id = 2 * p + h;
    }
    b = id / per_row;
    chunk = id - b * per_row;
  }
  const int tid = threadIdx.x, slot = tid % TPR, rl = tid / TPR, wave = tid >> 6;
  const int c = chunk * CH + slot * 4, cpg = C / G, spg = cpg >> 2;   // float4 slots per group: 1, 2, 4 (or 8 with 32-channel chunks)
  // buffer addressing: the sample's rows behind one descriptor, lane offset in a VGPR, the sweep's offset in an SGPR -- no per-load
  // address registers (35 in-flight 64-bit pointers were what spilled); rows >= S get an out-of-range lane offset: zeros / dropped stores
  const auto rX = __builtin_amdgcn_make_buffer_rsrc((void*)(x + (int64_t)b * S * C), 0, (uint32_t)((int64_t)S * C * 4), 0x00020000);
  const auto rO = __builtin_amdgcn_make_buffer_rsrc((void*)(out + (int64_t)b * S * C), 0, (uint32_t)((int64_t)S * C * 2), 0x00020000);
  const uint32_t voff = (uint32_t)(rl * C + c) * 4u, sstep = (uint32_t)(RP * C) * 4u;
  constexpr uint32_t GN_OOB = 0xFFFFF000u;
  typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
  typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;
  float4 v[RMAX];
#pragma unroll
  for (int i = 0; i < RMAX; ++i) {
    // rows >= S: an out-of-range LANE offset (whether the hardware's range check also counts the scalar offset is not relied upon)
    const u32x4_t w = __builtin_amdgcn_raw_buffer_load_b128(rX, rl + RP * i < S ? voff : GN_OOB, (uint32_t)i * sstep, 0);
    v[i] = make_float4(__uint_as_float(w[0]), __uint_as_float(w[1]), __uint_as_float(w[2]), __uint_as_float(w[3]));
  }
  // sum over the workgroup's values of this thread's group: row lanes of the wave (lane bits 3..5), the group's slots (lane bits
  // 0..log2(spg)-1), then the 8 waves in order
  auto group_sum = [&](float t, int buf) {
#pragma unroll
    for (int m = TPR; m < 64; m <<= 1) t += __shfl_xor(t, m);
    for (int m = 1; m < spg; m <<= 1) t += __shfl_xor(t, m);
    if ((tid & 63) < TPR) sred[buf][wave][slot] = t;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < NWV; ++w) tot += sred[buf][w][slot];
    return tot;
  };
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < RMAX; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  const float cnt = (float)S * (float)cpg;
  const float mean = group_sum(s, 0) / cnt;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < RMAX; ++i) {
    if (rl + RP * i < S) {
      const float d0 = v[i].x - mean, d1 = v[i].y - mean, d2 = v[i].z - mean, d3 = v[i].w - mean;
      q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
    }
  }
  const float var = group_sum(q, 1) / cnt;
  const float rstd = 1.0f / sqrtf(var + eps);
  // `partials` keeps its contract -- the (sample, group)'s statistics as fp64 partial sums [b][chunk][g][sum, sum of squares], which
  // pd_groupnorm_silu_bwd reduces again: everything in chunk 0 (mean * n and (var + mean^2) * n reproduce this kernel's mean and
  // variance exactly in fp64), zeros in the others
  if (rl == 0 && slot % spg == 0) {
    double* pp = partials + (int64_t)b * nchunk * G * 2 + (c / cpg) * 2;
    pp[0] = (double)mean * (double)cnt;
    pp[1] = ((double)var + (double)mean * (double)mean) * (double)cnt;
    for (int k = 1; k < nchunk; ++k) { pp[(int64_t)k * G * 2] = 0.0; pp[(int64_t)k * G * 2 + 1] = 0.0; }
  }
  const float4 ga = *(const float4*)(gamma + c), be = *(const float4*)(beta + c);
  float a[4] = {rstd * ga.x, rstd * ga.y, rstd * ga.z, rstd * ga.w};
  float d[4] = {be.x - mean * a[0], be.y - mean * a[1], be.z - mean * a[2], be.w - mean * a[3]};
  if (ss_scale) {
    const float4 sc = *(const float4*)(ss_scale + (int64_t)b * ld_ss + c), sh = *(const float4*)(ss_shift + (int64_t)b * ld_ss + c);
    const float scv[4] = {sc.x, sc.y, sc.z, sc.w}, shv[4] = {sh.x, sh.y, sh.z, sh.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) { a[k] *= (1.f + scv[k]); d[k] = d[k] * (1.f + scv[k]) + shv[k]; }
  }
#pragma unroll
  for (int i = 0; i < RMAX; ++i) {
    float y[4] = {v[i].x * a[0] + d[0], v[i].y * a[1] + d[1], v[i].z * a[2] + d[2], v[i].w * a[3] + d[3]};
    if (silu) {
#pragma unroll
      for (int k = 0; k < 4; ++k) y[k] = y[k] / (1.f + __expf(-y[k]));
    }
    const u32x2_t o = {pack_op2(y[0], y[1]), pack_op2(y[2], y[3])};
    __builtin_amdgcn_raw_buffer_store_b64(o, rO, rl + RP * i < S ? voff >> 1 : GN_OOB, ((uint32_t)i * sstep) >> 1, 0);
    if ((i & 1) == 1) __builtin_amdgcn_sched_barrier(0);      // (two rows at a time: an unbounded interleave of the 35 rows spilled)
  }
}

// Shapes of the one-pass kernel: 16-channel chunks holding whole groups, at most 26 x 128 rows (everything of a (sample, chunk) in registers)
static bool gn_onepass_fits(int S, int C, int G) {
  const int cpg = C / G;
  return C % 32 == 0 && cpg % 4 == 0 && 16 % cpg == 0 && S <= 128 * 26;
}
// `fine` (pd_call_opts.small_grid: the caller may choose kernels by launch size -- the engine's small-batch mode) at <= 1024 rows, when the
// 32-channel chunks would leave more than half of the CUs without a workgroup: 16-channel chunks, twice the workgroups, half the rows in
// flight per thread (level-1 rows of the SEVIR-LR denoiser at 4 trajectories: 7.3 vs 9.3 us).  Its statistics are summed in another order
// than the coarse kernel's (not bit-identical to it).  Not for the longer rows: 8-channel chunks there read 32 B of every 128-B line
// (measured 17.8 vs 16.7 us at level 0).
static void gn_onepass_launch(const float* x, const float* gamma, const float* beta, const float* ss_scale, const float* ss_shift, int ld_ss,
                              double* partials, int nchunk, pd_bf16* out, int B, int S, int C, int G, float eps, int silu, bool fine,
                              hipStream_t s) {
  const int cpg = C / G;
  const bool small = S <= 64 * 16;
  fine = fine && small && 16 % cpg == 0 && (int64_t)B * (C / 32) * 2 <= pd_num_cus();
#define GN1P(RM, CHN) hipLaunchKernelGGL((gn_onepass_kernel<RM, 512, CHN>), dim3(B * (C / CHN)), dim3(512), 0, s, x, gamma, beta, ss_scale, ss_shift, \
                                         ld_ss, partials, nchunk, out, S, C, G, eps, silu)
  if (small) { if (fine) GN1P(8, 16); else GN1P(16, 32); }
  else GN1P(26, 16);
#undef GN1P
}

extern "C" int PD_ENTRY(groupnorm_silu)(const float* x, const float* gamma, const float* beta, const float* ss_scale,
                                        const float* ss_shift, int ld_ss, double* partials, pd_bf16* out, pd_bf16* out_lo, int B, int S,
                                        int C, int G, int ld_out, float eps, int silu, const pd_call_opts* opts, pd_stream_t stream) {
  PD_FORWARD_F16(PD_OPTS_F16(opts), pd_f16_groupnorm_silu(x, gamma, beta, ss_scale, ss_shift, ld_ss, partials, out, out_lo, B, S, C, G, ld_out,
                                                          eps, silu, opts, stream));
  PD_CHECK_ARG(!PD_IS_F16 || !out_lo, "pd_groupnorm_silu: the hi/lo split exists for bfloat16 operands only");
  const bool onepass = !(opts && opts->groupnorm_two_launches);      // A/B: the statistics + apply pair of launches instead of the one-pass kernel
  PD_CHECK_ARG(x && gamma && beta && partials && out, "pd_groupnorm_silu: null pointer");
  PD_CHECK_ARG(G > 0 && C % G == 0 && ld_out >= C, "pd_groupnorm_silu: bad C/G/ld_out (%d,%d,%d)", C, G, ld_out);
  PD_CHECK_ARG((ss_scale == nullptr) == (ss_shift == nullptr), "pd_groupnorm_silu: scale/shift must come together");
  PD_CHECK_ARG(G <= 4096, "pd_groupnorm_silu: too many groups");
  const int nchunk = pd_groupnorm_nchunk(S, C);
  hipStream_t s = (hipStream_t)stream;
  const int CV = C / 4, cpg = C / G;
  const bool vec = (C % 4 == 0) && CV <= 256 && (256 % CV == 0) && (cpg % 4 == 0) && ld_out == C && G <= 256 &&
                   (((uintptr_t)x | (uintptr_t)gamma | (uintptr_t)beta) & 15) == 0 && (((uintptr_t)out | (uintptr_t)out_lo) & 7) == 0 &&
                   (!ss_scale || ((ld_ss % 4 == 0) && (((uintptr_t)ss_scale | (uintptr_t)ss_shift) & 15) == 0));
  // bf16 engine (no lo half), 16-channel chunks holding whole groups, at most 26 x 128 rows: everything of a (sample, chunk) in registers
  if (vec && onepass && !out_lo && gn_onepass_fits(S, C, G)) {
    gn_onepass_launch(x, gamma, beta, ss_scale, ss_shift, ld_ss, partials, nchunk, out, B, S, C, G, eps, silu,
                      opts && opts->small_grid, s);
    PD_CHECK_LAUNCH();
    return PD_OK;
  }
  if (vec) {
    hipLaunchKernelGGL(gn_stats_vec_kernel, dim3(nchunk, B), dim3(256), 0, s, x, partials, S, C, G);
    PD_CHECK_LAUNCH();
    hipLaunchKernelGGL(gn_apply_vec_kernel<false>, dim3(nchunk, B), dim3(256), 0, s, x, gamma, beta, ss_scale, ss_shift, ld_ss, partials,
                       out, out_lo, S, C, G, eps, silu, nchunk, 1.f);
    PD_CHECK_LAUNCH();
    return PD_OK;
  }
  hipLaunchKernelGGL(gn_stats_kernel, dim3(nchunk, B), dim3(256), (512 + 2 * G) * sizeof(double), s, x, partials, S, C, G);
  PD_CHECK_LAUNCH();
  hipLaunchKernelGGL(gn_apply_kernel, dim3(nchunk, B), dim3(256), 2 * G * sizeof(float), s, x, gamma, beta, ss_scale, ss_shift,
                     ld_ss, partials, out, out_lo, S, C, G, ld_out, eps, silu, nchunk);
  PD_CHECK_LAUNCH();
  return PD_OK;
}

// Statistics only: mean / rstd per (sample, group) as fp32 pairs, from the same fp64 partial sums and the same reduction order as the
// apply kernels' prologue -- for consumers that normalise on the fly (pd_conv2d_gn_silu: the VAE's fused ResBlock convolution).
__global__ void __launch_bounds__(256) gn_finalize_kernel(const double* __restrict__ partials, float* __restrict__ stats, int S, int C, int G,
                                                          int nchunk, float eps) {
  const int b = blockIdx.x;
  const int cpg = C / G;
  for (int g = threadIdx.x; g < G; g += 256) {
    double ss = 0, qq = 0;
    const double* pp = partials + (int64_t)b * nchunk * G * 2 + g * 2;
    for (int k = 0; k < nchunk; ++k) { ss += pp[(int64_t)k * G * 2]; qq += pp[(int64_t)k * G * 2 + 1]; }
    const double cnt = (double)S * cpg;
    const double mean = ss / cnt;
    double var = qq / cnt - mean * mean;
    if (var < 0) var = 0;
    stats[((int64_t)b * G + g) * 2] = (float)mean;
    stats[((int64_t)b * G + g) * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
  }
}

#if !PD_IS_F16
extern "C" int pd_groupnorm_stats(const float* x, double* partials, float* stats, int B, int S, int C, int G, float eps, pd_stream_t stream) {
  PD_CHECK_ARG(x && partials && stats, "pd_groupnorm_stats: null pointer");
  PD_CHECK_ARG(G > 0 && C % G == 0 && G <= 4096 && B > 0 && S > 0, "pd_groupnorm_stats: bad B/S/C/G (%d,%d,%d,%d)", B, S, C, G);
  const int nchunk = pd_groupnorm_nchunk(S, C);
  hipStream_t s = (hipStream_t)stream;
  const int CV = C / 4, cpg = C / G;
  const bool vec = (C % 4 == 0) && CV <= 256 && (256 % CV == 0) && (cpg % 4 == 0) && G <= 256 && (((uintptr_t)x) & 15) == 0;
  if (vec) hipLaunchKernelGGL(gn_stats_vec_kernel, dim3(nchunk, B), dim3(256), 0, s, x, partials, S, C, G);
  else hipLaunchKernelGGL(gn_stats_kernel, dim3(nchunk, B), dim3(256), (512 + 2 * G) * sizeof(double), s, x, partials, S, C, G);
  PD_CHECK_LAUNCH();
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(B), dim3(256), 0, s, partials, stats, S, C, G, nchunk, eps);
  PD_CHECK_LAUNCH();
  return PD_OK;
}

// GroupNorm [-> SiLU] -> e4m3 rows (the operand of an fp8 pd_igemm launch): out[b, s, c] = e4m3(y * fp8_scale), ld_out == C
extern "C" int pd_groupnorm_silu_fp8(const float* x, const float* gamma, const float* beta, const float* ss_scale,
                                     const float* ss_shift, int ld_ss, double* partials, uint8_t* out, int B, int S, int C, int G,
                                     float eps, int silu, float fp8_scale, pd_stream_t stream) {
  PD_CHECK_ARG(x && gamma && beta && partials && out, "pd_groupnorm_silu_fp8: null pointer");
  PD_CHECK_ARG(G > 0 && C % G == 0 && fp8_scale > 0.f, "pd_groupnorm_silu_fp8: bad C/G/scale (%d,%d,%g)", C, G, (double)fp8_scale);
  PD_CHECK_ARG((ss_scale == nullptr) == (ss_shift == nullptr), "pd_groupnorm_silu_fp8: scale/shift must come together");
  const int nchunk = pd_groupnorm_nchunk(S, C);
  const int CV = C / 4, cpg = C / G;
  const bool vec = (C % 4 == 0) && CV <= 256 && (256 % CV == 0) && (cpg % 4 == 0) && G <= 256 &&
                   (((uintptr_t)x | (uintptr_t)gamma | (uintptr_t)beta) & 15) == 0 && ((uintptr_t)out & 3) == 0 &&
                   (!ss_scale || ((ld_ss % 4 == 0) && (((uintptr_t)ss_scale | (uintptr_t)ss_shift) & 15) == 0));
  if (!vec) {
    pd_set_error("pd_groupnorm_silu_fp8: needs C %% 4 == 0, C/4 dividing 256 and 4 | C/G (C = %d, G = %d)", C, G);
    return PD_ERR_UNSUPPORTED;
  }
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(gn_stats_vec_kernel, dim3(nchunk, B), dim3(256), 0, s, x, partials, S, C, G);
  PD_CHECK_LAUNCH();
  hipLaunchKernelGGL(gn_apply_vec_kernel<true>, dim3(nchunk, B), dim3(256), 0, s, x, gamma, beta, ss_scale, ss_shift, ld_ss, partials,
                     (pd_bf16*)out, (pd_bf16*)nullptr, S, C, G, eps, silu, nchunk, fp8_scale);
  PD_CHECK_LAUNCH();
  return PD_OK;
}

// -------------------------------------------------------------------------------------------------
// Data gradient of y = SiLU(GroupNorm(x)) on channels-last rows (guidance network, frozen affine): with xh = (x - mean) rstd,
// n = gamma xh + beta, dn = dy SiLU'(n), dxh = dn gamma and N = S * C/G elements per (sample, group):
//   dx = rstd (dxh - mean_N(dxh) - xh mean_N(dxh xh)).
// Two passes like the forward: per-chunk fp64 partial sums of (dxh, dxh xh), then the apply pass reduces them in a fixed order
// (deterministic).  mean / rstd are re-derived from the forward's partial sums.  A thread owns one channel; C divides 256.
// -------------------------------------------------------------------------------------------------
__device__ __forceinline__ void gn_reduce_partials(const double* __restrict__ pb, int nchunk, int G, double* spart, double* sout) {
  // pb: this sample's (nchunk, G, 2) partial sums -> sout[2 g], sout[2 g + 1]; 256 / G threads per group stride the chunks
  const int tid = threadIdx.x, LP = 256 / G;
  if (tid < LP * G) {
    const int g = tid % G, j = tid / G;
    double a = 0, b = 0;
    for (int k = j; k < nchunk; k += LP) { a += pb[((int64_t)k * G + g) * 2]; b += pb[((int64_t)k * G + g) * 2 + 1]; }
    spart[tid * 2] = a;
    spart[tid * 2 + 1] = b;
  }
  __syncthreads();
  if (tid < G) {
    double a = 0, b = 0;
    for (int j = 0; j < LP; ++j) { a += spart[(j * G + tid) * 2]; b += spart[(j * G + tid) * 2 + 1]; }
    sout[tid * 2] = a;
    sout[tid * 2 + 1] = b;
  }
  __syncthreads();
}

template <bool APPLY>
__global__ void __launch_bounds__(256) gn_silu_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          const double* __restrict__ fwd_partials, double* __restrict__ bwd_partials,
                                                          float* __restrict__ dx, int S, int C, int G, float eps, int silu, int nchunk) {
  __shared__ double spart[2 * 256];
  __shared__ double sfwd[2 * 256];
  __shared__ double sbwd[2 * 256];
  __shared__ float sred[2 * 256];
  const int b = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
  const int cpg = C / G;
  const double cnt = (double)S * cpg;
  gn_reduce_partials(fwd_partials + (int64_t)b * nchunk * G * 2, nchunk, G, spart, sfwd);
  if (APPLY) gn_reduce_partials(bwd_partials + (int64_t)b * nchunk * G * 2, nchunk, G, spart, sbwd);
  const int c = tid % C, rr = tid / C, RP = 256 / C, g = c / cpg;
  const double mean_d = sfwd[g * 2] / cnt;
  double var = sfwd[g * 2 + 1] / cnt - mean_d * mean_d;
  if (var < 0) var = 0;
  const float mean = (float)mean_d, rstd = (float)(1.0 / sqrt(var + (double)eps));
  const float ga = gamma[c], a = rstd * ga, d = beta[c] - mean * a;
  const float m1 = APPLY ? (float)(sbwd[g * 2] / cnt) : 0.f, m2 = APPLY ? (float)(sbwd[g * 2 + 1] / cnt) : 0.f;
  const int r0 = chunk * GN_ROWS, r1 = min(S, r0 + GN_ROWS);
  const int64_t base = ((int64_t)b * S + r0) * C + c;
  float s1 = 0.f, s2 = 0.f;
#pragma unroll 4
  for (int r = rr; r < r1 - r0; r += RP) {
    const float xv = x[base + (int64_t)r * C], gy = dy[base + (int64_t)r * C];
    const float n = xv * a + d;
    float dn = gy;
    if (silu) {
      const float sig = 1.f / (1.f + __expf(-n));
      dn = gy * sig * (1.f + n * (1.f - sig));
    }
    const float dxh = dn * ga, xh = (xv - mean) * rstd;
    if (APPLY) {
      dx[base + (int64_t)r * C] = rstd * (dxh - m1 - xh * m2);
    } else {
      s1 += dxh;
      s2 += dxh * xh;
    }
  }
  if (!APPLY) {
    sred[tid * 2] = s1;
    sred[tid * 2 + 1] = s2;
    __syncthreads();
    if (tid < G) {
      double a1 = 0, a2 = 0;
      for (int r = 0; r < RP; ++r)
        for (int k = 0; k < cpg; ++k) {
          const int t = r * C + tid * cpg + k;
          a1 += sred[t * 2];
          a2 += sred[t * 2 + 1];
        }
      double* out = bwd_partials + ((int64_t)b * nchunk + chunk) * G * 2;
      out[tid * 2] = a1;
      out[tid * 2 + 1] = a2;
    }
  }
}

extern "C" int pd_groupnorm_silu_bwd(const float* x, const float* dy, const float* gamma, const float* beta,
                                     const double* fwd_partials, double* bwd_partials, float* dx, int B, int S, int C, int G,
                                     float eps, int silu, pd_stream_t stream) {
  PD_CHECK_ARG(x && dy && gamma && beta && fwd_partials && bwd_partials && dx, "pd_groupnorm_silu_bwd: null pointer");
  PD_CHECK_ARG(G > 0 && C % G == 0, "pd_groupnorm_silu_bwd: bad C/G (%d,%d)", C, G);
  if (C > 256 || 256 % C != 0 || G > 256) {
    pd_set_error("pd_groupnorm_silu_bwd: C = %d must divide 256 (G = %d <= 256)", C, G);
    return PD_ERR_UNSUPPORTED;
  }
  if (B <= 0 || S <= 0) return PD_OK;
  const int nchunk = pd_groupnorm_nchunk(S, C);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(gn_silu_bwd_kernel<false>, dim3(nchunk, B), dim3(256), 0, s, x, dy, gamma, beta, fwd_partials, bwd_partials, dx, S, C, G,
                     eps, silu, nchunk);
  PD_CHECK_LAUNCH();
  hipLaunchKernelGGL(gn_silu_bwd_kernel<true>, dim3(nchunk, B), dim3(256), 0, s, x, dy, gamma, beta, fwd_partials, bwd_partials, dx, S, C, G,
                     eps, silu, nchunk);
  PD_CHECK_LAUNCH();
  return PD_OK;
}

#endif   // !PD_IS_F16 (statistics-only, e4m3 and backward entry points: one copy, in the bf16 build)

// -------------------------------------------------------------------------------------------------
// fp32 -> bf16 row cast with row-slice gather and zero column padding
// -------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) cast_rows_kernel(const float* __restrict__ x, pd_bf16* __restrict__ out,
                                                        pd_bf16* __restrict__ out_lo, int64_t n_out_rows, int rows_in, int row_off,
                                                        int rows_out, int C, int ld_in, int ld_out) {
  const int64_t total = n_out_rows * ld_out;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t ro = i / ld_out;
    const int c = (int)(i - ro * ld_out);
    const int64_t smp = ro / rows_out;
    const int r = (int)(ro - smp * rows_out);
    const float v = c < C ? x[(smp * rows_in + row_off + r) * (int64_t)ld_in + c] : 0.f;
    if (out_lo) {
      uint16_t hi, lo;
      f2bf_split(v, hi, lo);
      out[i] = hi; out_lo[i] = lo;
    } else {
      out[i] = f2op(v);
    }
  }
}

extern "C" int PD_ENTRY(cast_rows)(const float* x, pd_bf16* out, pd_bf16* out_lo, int64_t n_samples, int rows_per_sample_in, int row_off,
                                   int rows_per_sample_out, int C, int ld_in, int ld_out, const pd_call_opts* opts, pd_stream_t stream) {
  PD_FORWARD_F16(PD_OPTS_F16(opts), pd_f16_cast_rows(x, out, out_lo, n_samples, rows_per_sample_in, row_off, rows_per_sample_out, C, ld_in,
                                                     ld_out, opts, stream));
  PD_CHECK_ARG(!PD_IS_F16 || !out_lo, "pd_cast_rows: the hi/lo split exists for bfloat16 operands only");
  PD_CHECK_ARG(x && out, "pd_cast_rows: null pointer");
  PD_CHECK_ARG(row_off >= 0 && row_off + rows_per_sample_out <= rows_per_sample_in && ld_in >= C && ld_out >= C, "pd_cast_rows: bad geometry");
  const int64_t rows = n_samples * rows_per_sample_out;
  if (rows <= 0) return PD_OK;
  const int64_t total = rows * ld_out;
  const unsigned grid = (unsigned)min((int64_t)4096, (total + 255) / 256);
  hipLaunchKernelGGL(cast_rows_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, out, out_lo, rows, rows_per_sample_in, row_off,
                     rows_per_sample_out, C, ld_in, ld_out);
  PD_CHECK_LAUNCH();
  return PD_OK;
}

}  // namespace PD_NS
