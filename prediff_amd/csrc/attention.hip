// Cuboid self-attention core: softmax(scale q k^T + rel-pos bias [mask]) v for every (sample, cuboid, head).
//
// Replaces cuboid_transformer.py:839-861 (reorder, q*scale, q@k^T, +bias), :947-949 (masked_softmax, @v, head merge) and
// :956-962 (reverse reorder / un-shift / un-pad) of the reference.  The cuboid decomposition (local / dilated strategy,
// window shift, zero/ignore padding) never materialises: a per-layer int32 table tok_index[cuboid][slot] gives the flat
// token id of every slot (or -1 for a padded slot), q/k/v rows are gathered straight from the (B, ntok, 3C) QKV GEMM
// output and the result is scattered back to natural token order -- "shift/pad done as coalesced HBM gathers".
//
//  * MFMA path (vol <= 16, bf16 qkv, hd % 32 == 0): one wave per (sample, cuboid, head).
//      S^T = K Q^T   with v_mfma_f32_16x16x32_bf16   (A = K rows, B = Q rows: both 16 B/lane row loads from HBM/L2)
//      softmax over keys: every lane owns one query column and 4 key rows -> 4-register + 2-shuffle reduction
//      O^T = V^T P^T with v_mfma_f32_16x16x16_bf16   (P^T is already in B-operand layout: no LDS, no transpose)
//    Per work item: 3*vol*hd*2 B in, vol*hd*2 B out, 4*vol^2*hd flop (~8 flop/B): HBM/latency bound by construction.
//  * generic path (vol <= 64, any hd <= 128, bf16 or fp32 qkv): LDS-staged fp32 VALU kernel; used for the non-axial
//    patterns (video_swin / spatial_lg / dilated / default cuboids) and for the fp32-accurate mode.
#include "common.h"

// ------------------------------------------------------------------------------------------------- MFMA path
__global__ void __launch_bounds__(256) cuboid_attn_mfma_kernel(const pd_cuboid_attn_args p) {
  const int lane = threadIdx.x & 63;
  const int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);   // ((b * nc) + c) * heads + h
  const int64_t nitems = (int64_t)p.B * p.nc * p.heads;
  if (item >= nitems) return;
  const int h = (int)(item % p.heads);
  const int c = (int)((item / p.heads) % p.nc);
  const int b = (int)(item / ((int64_t)p.heads * p.nc));
  const int hd = p.C / p.heads;
  const int q = lane & 15, g = lane >> 4;

  const int tok = q < p.vol ? p.tok_index[c * p.vol + q] : -1;   // token of row/column (lane & 15)
  const pd_bf16* row = p.qkv_bf16 + ((int64_t)b * p.ntok + (tok >= 0 ? tok : 0)) * p.ld_qkv + h * hd;

  // ---- S^T[key][query] ----
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  for (int d0 = 0; d0 < hd; d0 += 32) {
    bf16x8 kf = {0, 0, 0, 0, 0, 0, 0, 0}, qf = {0, 0, 0, 0, 0, 0, 0, 0};
    if (tok >= 0) {
      qf = *(const bf16x8*)(row + d0 + 8 * g);
      kf = *(const bf16x8*)(row + p.C + d0 + 8 * g);
    }
    s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf, s, 0, 0, 0);
  }
  // lane: query q, keys 4g..4g+3
  float sc[4];
  float mx = -3.0e38f;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int key = 4 * g + r;
    float v = -INFINITY;
    if (key < p.vol && q < p.vol) {
      v = s[r] * p.scale + p.bias[((int64_t)h * p.vol + q) * p.vol + key];
      if (p.mask && !p.mask[((int64_t)c * p.vol + q) * p.vol + key]) v = -1e18f;
    }
    sc[r] = v;
    mx = fmaxf(mx, v);
  }
  mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  float pr[4], sum = 0.f;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    pr[r] = expf(sc[r] - mx);   // exp(-inf) = 0 for non-existent keys
    sum += pr[r];
  }
  sum += __shfl_xor(sum, 16, 64);
  sum += __shfl_xor(sum, 32, 64);
  const float inv = sum > 0.f ? 1.f / sum : 0.f;
  s16x4 pf;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float v = pr[r] * inv;
    if (sc[r] <= -1e18f) v = 0.f;   // masked_softmax multiplies by the mask after the softmax
    pf[r] = (short)f2bf(v);
  }

  // ---- O^T[d][query] = sum_key V[key][d] P[query][key] ----
  // A = V^T: lane holds V[key = 4g + jj][d0 + (lane & 15)], jj = 0..3;  B = P^T: pf.
  int vtok[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) vtok[r] = (4 * g + r) < p.vol ? p.tok_index[c * p.vol + 4 * g + r] : -1;
  const pd_bf16* vbase = p.qkv_bf16 + (int64_t)b * p.ntok * p.ld_qkv + 2 * p.C + h * hd + q;
  pd_bf16* orow = (tok >= 0) ? p.out_bf16 + ((int64_t)b * p.ntok + tok) * p.ld_out + h * hd + 4 * g : nullptr;
  for (int d0 = 0; d0 < hd; d0 += 16) {
    s16x4 vf;
#pragma unroll
    for (int r = 0; r < 4; ++r) vf[r] = vtok[r] >= 0 ? (short)vbase[(int64_t)vtok[r] * p.ld_qkv + d0] : (short)0;
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
    o = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(vf, pf, o, 0, 0, 0);
    // the result feeds inline asm (v_cvt_pk_bf16_f32) directly: hipcc does not insert the XDL-write -> VALU-read wait states for asm
    asm volatile("s_nop 15" : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3]));
    // lane: query q, d = d0 + 4g + r
    if (orow) {
      const uint32_t lo = pack_bf16x2(o[0], o[1]);
      const uint32_t hi = pack_bf16x2(o[2], o[3]);
      *(uint2*)(orow + d0) = make_uint2(lo, hi);
    }
  }
}

// ------------------------------------------------------------------------------------------------- generic path
template <typename QT>
__device__ __forceinline__ float ldq(const QT* p);
template <>
__device__ __forceinline__ float ldq<float>(const float* p) { return *p; }
template <>
__device__ __forceinline__ float ldq<pd_bf16>(const pd_bf16* p) { return bf2f(*p); }

constexpr int GA_MAXVOL = 64;

template <typename QT>
__global__ void __launch_bounds__(256) cuboid_attn_generic_kernel(const pd_cuboid_attn_args p, const QT* __restrict__ qkv) {
  extern __shared__ float sm[];
  const int hd = p.C / p.heads, vol = p.vol;
  float* sq = sm;                       // [vol][hd]
  float* sk = sq + vol * hd;            // [vol][hd+1]
  float* sv = sk + vol * (hd + 1);      // [vol][hd]
  float* ss = sv + vol * hd;            // [vol][vol+1]
  __shared__ int stok[GA_MAXVOL];
  const int64_t item = blockIdx.x;
  const int h = (int)(item % p.heads);
  const int c = (int)((item / p.heads) % p.nc);
  const int b = (int)(item / ((int64_t)p.heads * p.nc));
  const int tid = threadIdx.x;
  if (tid < vol) stok[tid] = p.tok_index[c * vol + tid];
  __syncthreads();
  for (int i = tid; i < vol * hd; i += 256) {
    const int r = i / hd, d = i - r * hd;
    const int tok = stok[r];
    float qv = 0.f, kv = 0.f, vv = 0.f;
    if (tok >= 0) {
      const QT* row = qkv + ((int64_t)b * p.ntok + tok) * p.ld_qkv + h * hd + d;
      qv = ldq<QT>(row); kv = ldq<QT>(row + p.C); vv = ldq<QT>(row + 2 * p.C);
    }
    sq[r * hd + d] = qv * p.scale;            // q * scale before q k^T (cuboid_transformer.py:852)
    sk[r * (hd + 1) + d] = kv;
    sv[r * hd + d] = vv;
  }
  __syncthreads();
  for (int i = tid; i < vol * vol; i += 256) {
    const int qi = i / vol, kj = i - qi * vol;
    float a = 0.f;
    for (int d = 0; d < hd; ++d) a += sq[qi * hd + d] * sk[kj * (hd + 1) + d];
    a += p.bias[((int64_t)h * vol + qi) * vol + kj];
    if (p.mask && !p.mask[((int64_t)c * vol + qi) * vol + kj]) a = -1e18f;
    ss[qi * (vol + 1) + kj] = a;
  }
  __syncthreads();
  // softmax per query row: one wave per row
  const int lane = tid & 63, wv = tid >> 6;
  for (int qi = wv; qi < vol; qi += 4) {
    const float v = lane < vol ? ss[qi * (vol + 1) + lane] : -INFINITY;
    const float mx = wave_max(v);
    const float e = lane < vol ? expf(v - mx) : 0.f;
    const float sum = wave_sum(e);
    if (lane < vol) ss[qi * (vol + 1) + lane] = (v <= -1e18f) ? 0.f : e / sum;
  }
  __syncthreads();
  for (int i = tid; i < vol * hd; i += 256) {
    const int qi = i / hd, d = i - qi * hd;
    const int tok = stok[qi];
    if (tok < 0) continue;
    float a = 0.f;
    for (int kj = 0; kj < vol; ++kj) a += ss[qi * (vol + 1) + kj] * sv[kj * hd + d];
    const int64_t o = ((int64_t)b * p.ntok + tok) * p.ld_out + h * hd + d;
    if (p.out_f32) p.out_f32[o] = a;
    if (p.out_bf16) {
      if (p.out_bf16_lo) {
        uint16_t hi, lo;
        f2bf_split(a, hi, lo);
        p.out_bf16[o] = hi; p.out_bf16_lo[o] = lo;
      } else {
        p.out_bf16[o] = f2bf(a);
      }
    }
  }
}

extern "C" int pd_cuboid_attention(const pd_cuboid_attn_args* pa, pd_stream_t stream) {
  PD_CHECK_ARG(pa != nullptr, "pd_cuboid_attention: null args");
  const pd_cuboid_attn_args a = *pa;
  PD_CHECK_ARG((a.qkv_bf16 != nullptr) != (a.qkv_f32 != nullptr), "pd_cuboid_attention: exactly one of qkv_bf16 / qkv_f32");
  PD_CHECK_ARG(a.tok_index && a.bias && (a.out_bf16 || a.out_f32), "pd_cuboid_attention: null pointer");
  PD_CHECK_ARG(a.heads > 0 && a.C % a.heads == 0 && a.vol > 0 && a.nc > 0, "pd_cuboid_attention: bad geometry");
  const int hd = a.C / a.heads;
  hipStream_t s = (hipStream_t)stream;
  const int64_t nitems = (int64_t)a.B * a.nc * a.heads;
  const bool mfma_ok = a.qkv_bf16 && a.vol <= 16 && (hd % 32) == 0 && a.out_bf16 && !a.out_f32 && !a.out_bf16_lo &&
                       (a.ld_qkv % 8) == 0 && (a.ld_out % 4) == 0 && !a.force_generic;
  if (mfma_ok) {
    hipLaunchKernelGGL(cuboid_attn_mfma_kernel, dim3((unsigned)((nitems + 3) / 4)), dim3(256), 0, s, a);
    PD_CHECK_LAUNCH();
    return PD_OK;
  }
  if (a.vol > GA_MAXVOL || hd > 128) {
    pd_set_error("pd_cuboid_attention: cuboid volume %d (max %d) / head_dim %d (max 128) not supported by the HIP path", a.vol,
                 GA_MAXVOL, hd);
    return PD_ERR_UNSUPPORTED;
  }
  const size_t lds = sizeof(float) * ((size_t)a.vol * hd * 2 + (size_t)a.vol * (hd + 1) + (size_t)a.vol * (a.vol + 1));
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)cuboid_attn_generic_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512);
    (void)hipFuncSetAttribute((const void*)cuboid_attn_generic_kernel<pd_bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512);
    attr_set = true;
  }
  if (a.qkv_f32)
    hipLaunchKernelGGL((cuboid_attn_generic_kernel<float>), dim3((unsigned)nitems), dim3(256), lds, s, a, a.qkv_f32);
  else
    hipLaunchKernelGGL((cuboid_attn_generic_kernel<pd_bf16>), dim3((unsigned)nitems), dim3(256), lds, s, a, a.qkv_bf16);
  PD_CHECK_LAUNCH();
  return PD_OK;
}

// ------------------------------------------------------------------------------------------------- row softmax (VAE mid attention)
__global__ void __launch_bounds__(256) softmax_rows_kernel(const float* __restrict__ x, pd_bf16* __restrict__ out,
                                                           pd_bf16* __restrict__ out_lo, int64_t rows, int n, int ld_in, int ld_out) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + row * ld_in;
  float mx = -INFINITY;
  for (int i = lane; i < n; i += 64) mx = fmaxf(mx, xr[i]);
  mx = wave_max(mx);
  float sum = 0.f;
  for (int i = lane; i < n; i += 64) sum += expf(xr[i] - mx);
  sum = wave_sum(sum);
  const float inv = 1.f / sum;
  for (int i = lane; i < ld_out; i += 64) {
    const float v = i < n ? expf(xr[i] - mx) * inv : 0.f;
    if (out_lo) {
      uint16_t hi, lo;
      f2bf_split(v, hi, lo);
      out[row * ld_out + i] = hi; out_lo[row * ld_out + i] = lo;
    } else {
      out[row * ld_out + i] = f2bf(v);
    }
  }
}

extern "C" int pd_softmax_rows(const float* x, pd_bf16* out, pd_bf16* out_lo, int64_t rows, int n, int ld_in, int ld_out,
                               pd_stream_t stream) {
  PD_CHECK_ARG(x && out && n > 0 && ld_in >= n && ld_out >= n, "pd_softmax_rows: bad args");
  if (rows <= 0) return PD_OK;
  hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, out, out_lo, rows, n,
                     ld_in, ld_out);
  PD_CHECK_LAUNCH();
  return PD_OK;
}
