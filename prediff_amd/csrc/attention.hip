// Cuboid self-attention core: softmax(scale q k^T + rel-pos bias [mask]) v for every (sample, cuboid, head).
//
// Replaces cuboid_transformer.py:839-861 (reorder, q*scale, q@k^T, +bias), :947-949 (masked_softmax, @v, head merge) and
// :956-962 (reverse reorder / un-shift / un-pad) of the reference.  The cuboid decomposition (local / dilated strategy,
// window shift, zero/ignore padding) never materialises: a per-layer int32 table tok_index[cuboid][slot] gives the flat
// token id of every slot (or -1 for a padded slot), q/k/v rows are gathered straight from the (B, ntok, 3C) QKV GEMM
// output and the result is scattered back to natural token order -- "shift/pad done as coalesced HBM gathers".
//
//  * MFMA path (vol <= 64, bf16 qkv, hd % 32 == 0): one wave per (sample, cuboid, head, 16-query tile), 1-4 key tiles of 16.
//      S^T = K Q^T   with v_mfma_f32_16x16x32_bf16   (A = K rows, B = Q rows: both 16 B/lane row loads from HBM/L2)
//      softmax over keys: every lane owns one query column and 4 key rows -> 4-register + 2-shuffle reduction
//      O^T = V^T P^T with v_mfma_f32_16x16x16_bf16   (P^T is already in B-operand layout: no LDS, no transpose)
//    Per work item: 3*vol*hd*2 B in, vol*hd*2 B out, 4*vol^2*hd flop (~8 flop/B): HBM/latency bound by construction.
//    (Tried in round 2: V rows staged through per-wave LDS with 16 B row loads instead of the 2-byte V^T gathers below -- no faster:
//     24.7 / 29.4 us against 24.7 / 29.7 us at the v1 level-1 shapes, 10 % slower at head_dim 128 with 2 key tiles; the kernel already
//     moves 3-4.4 TB/s of q, k, v, o bytes and the gathers hit lines the K loads just brought in.  scripts/bench_attn_core.py.)
//  * large cuboids (vol > 64, bf16 qkv, hd 32 / 64 / 128): online-softmax MFMA kernel, one wave per 16 queries.
//  * generic path (vol <= 64, any hd <= 128, bf16 or fp32 qkv): LDS-staged fp32 VALU kernel; used for the fp32-accurate mode and
//    odd head sizes.
#include "common.h"

namespace PD_NS {

// ------------------------------------------------------------------------------------------------- MFMA path
// combine over the four 16-lane rows of the wave (v_permlane16_swap / v_permlane32_swap: one VALU instruction per exchange)
__device__ __forceinline__ float attn_rows4_max(float v) {
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float attn_rows4_sum(float v) {
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// One wave per (sample, cuboid, head, tile of 16 queries); KT = key tiles of 16 (cuboid volume <= 16 KT: the axial cuboids of the
// SEVIR-LR grid need 1, those of the 48 x 48 full-resolution grid -- volumes 25 / 48 / 24 -- need 2 or 3).  All KT x 4 scores of a
// lane stay in registers, so the softmax is exact (no online rescaling).
template <int KT>
__global__ void __launch_bounds__(256) cuboid_attn_mfma_kernel(const pd_cuboid_attn_args p) {
  const int lane = threadIdx.x & 63;
  const int QT = (p.vol + 15) >> 4;
  const int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);   // (((b * nc) + c) * heads + h) * QT + qt
  const int64_t nitems = (int64_t)p.B * p.nc * p.heads * QT;
  if (item >= nitems) return;
  const int qt = (int)(item % QT);
  const int64_t it2 = item / QT;
  const int h = (int)(it2 % p.heads);
  const int c = (int)((it2 / p.heads) % p.nc);
  const int b = (int)(it2 / ((int64_t)p.heads * p.nc));
  const int hd = p.C / p.heads;
  const int l16 = lane & 15, g = lane >> 4;
  const int vol = p.vol;
  const int query = qt * 16 + l16;                                       // this lane's query column

  const int qtok = query < vol ? p.tok_index[c * vol + query] : -1;
  const pd_bf16* base = p.qkv_bf16 + (int64_t)b * p.ntok * p.ld_qkv + h * hd;
  const pd_bf16* qrow = base + (int64_t)(qtok >= 0 ? qtok : 0) * p.ld_qkv;

  // ---- S^T[key][query], one 16 x 16 tile per key tile ----
  f32x4 s[KT];
  int ktok[KT];                                                          // token of key row kt*16 + l16 (A operand rows)
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) {
    const int key = kt * 16 + l16;
    ktok[kt] = key < vol ? p.tok_index[c * vol + key] : -1;
    s[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  for (int d0 = 0; d0 < hd; d0 += 32) {
    op8 qf = {0, 0, 0, 0, 0, 0, 0, 0};
    if (qtok >= 0) qf = *(const op8*)(qrow + d0 + 8 * g);
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      op8 kf = {0, 0, 0, 0, 0, 0, 0, 0};
      if (ktok[kt] >= 0) kf = *(const op8*)(base + (int64_t)ktok[kt] * p.ld_qkv + p.C + d0 + 8 * g);
      s[kt] = mfma_16x16x32(kf, qf, s[kt]);
    }
  }
  // lane: query `query`, keys kt*16 + 4g .. +3
  float sc[KT][4];
  float mx = -3.0e38f;
#pragma unroll
  for (int kt = 0; kt < KT; ++kt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int key = kt * 16 + 4 * g + r;
      float v = -INFINITY;
      if (key < vol && query < vol) {
        v = s[kt][r] * p.scale + p.bias[((int64_t)h * vol + query) * vol + key];
        if (p.mask && !p.mask[((int64_t)c * vol + query) * vol + key]) v = -1e18f;
      }
      sc[kt][r] = v;
      mx = fmaxf(mx, v);
    }
  mx = attn_rows4_max(mx);
  float sum = 0.f;
#pragma unroll
  for (int kt = 0; kt < KT; ++kt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float e = expf(sc[kt][r] - mx);   // exp(-inf) = 0 for non-existent keys
      sum += e;
      s[kt][r] = e;
    }
  sum = attn_rows4_sum(sum);
  const float inv = sum > 0.f ? 1.f / sum : 0.f;
  s16x4 pf[KT];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v = s[kt][r] * inv;
      if (sc[kt][r] <= -1e18f) v = 0.f;   // masked_softmax multiplies by the mask after the softmax
      pf[kt][r] = (short)f2op(v);
    }

  // ---- O^T[d][query] = sum_key V[key][d] P[query][key] ----
  // A = V^T: lane holds V[key = kt*16 + 4g + jj][d0 + l16], jj = 0..3;  B = P^T: pf[kt].
  int vtok[KT][4];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int key = kt * 16 + 4 * g + r;
      vtok[kt][r] = key < vol ? p.tok_index[c * vol + key] : -1;
    }
  const pd_bf16* vbase = base + 2 * p.C + l16;
  pd_bf16* orow = (qtok >= 0) ? p.out_bf16 + ((int64_t)b * p.ntok + qtok) * p.ld_out + h * hd + 4 * g : nullptr;
  for (int d0 = 0; d0 < hd; d0 += 16) {
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      s16x4 vf;
#pragma unroll
      for (int r = 0; r < 4; ++r) vf[r] = vtok[kt][r] >= 0 ? (short)vbase[(int64_t)vtok[kt][r] * p.ld_qkv + d0] : (short)0;
      o = mfma_16x16x16(vf, pf[kt], o);
    }
    // the result feeds inline asm (v_cvt_pk_bf16_f32) directly: hipcc does not insert the XDL-write -> VALU-read wait states for asm
    asm volatile("s_nop 15" : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3]));
    // lane: query, d = d0 + 4g + r
    if (orow) {
      if (p.out_fp8_log2 > 0) {         // e4m3 bytes, value * 2^k, saturating: the A operand of an fp8 proj launch
        const float f8s = __builtin_amdgcn_ldexpf(1.0f, p.out_fp8_log2);
        int w = __builtin_amdgcn_cvt_pk_fp8_f32(fminf(fmaxf(o[0] * f8s, -448.f), 448.f), fminf(fmaxf(o[1] * f8s, -448.f), 448.f), 0, false);
        w = __builtin_amdgcn_cvt_pk_fp8_f32(fminf(fmaxf(o[2] * f8s, -448.f), 448.f), fminf(fmaxf(o[3] * f8s, -448.f), 448.f), w, true);
        *(int*)((uint8_t*)p.out_bf16 + ((int64_t)b * p.ntok + qtok) * p.ld_out + h * hd + 4 * g + d0) = w;
      } else {
        const uint32_t lo = pack_op2(o[0], o[1]);
        const uint32_t hi = pack_op2(o[2], o[3]);
        *(uint2*)(orow + d0) = make_uint2(lo, hi);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------- e4m3 core (precision="fp8")
// The same core with OCP e4m3 q / k / v (bytes, value * 2^qk): S^T = K Q^T on v_mfma_f32_16x16x32_fp8_fp8 (8 bytes of one row per lane:
// the bf16 core's operand geometry at half the bytes), the probabilities as e4m3(P * 256) (P in [0, 1]: 2^8 keeps 2^-17 above the
// subnormal floor and 256 below the format's 448), and O^T = V^T P^T over 32 keys per MFMA.  BASELINE.json configs[4] ("fp8 MFMA
// attention"): the reference's q k^T and attn v products (cuboid_transformer.py:849-861, 947-952) on e4m3 operands.  A dot product
// does not care in which order it sums: MFMA position (g, j) of the 32-key block kb stands for key  kb * 32 + (j < 4 ? 4 g + j :
// 16 + 4 g + j - 4)  -- exactly the keys whose scores the lane already holds in the C layout of the two S^T tiles (no cross-lane
// movement of P); the V bytes are gathered with the same assignment.  fp32 scores, softmax and accumulation; q / k / v halve the
// core's HBM bytes (it is a 2 flop / B kernel).  Only in the bf16 translation unit: e4m3 does not depend on the 16-bit operand type.
#if !PD_IS_F16
template <int KT>
__global__ void __launch_bounds__(256) cuboid_attn_mfma_fp8_kernel(const pd_cuboid_attn_args p) {
  constexpr int KB = (KT + 1) / 2;                                       // 32-key blocks
  const int lane = threadIdx.x & 63;
  const int QT = (p.vol + 15) >> 4;
  const int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);   // (((b * nc) + c) * heads + h) * QT + qt
  const int64_t nitems = (int64_t)p.B * p.nc * p.heads * QT;
  if (item >= nitems) return;
  const int qt = (int)(item % QT);
  const int64_t it2 = item / QT;
  const int h = (int)(it2 % p.heads);
  const int c = (int)((it2 / p.heads) % p.nc);
  const int b = (int)(it2 / ((int64_t)p.heads * p.nc));
  const int hd = p.C / p.heads;
  const int l16 = lane & 15, g = lane >> 4;
  const int vol = p.vol;
  const int query = qt * 16 + l16;
  const int qtok = query < vol ? p.tok_index[c * vol + query] : -1;
  const uint8_t* base = (const uint8_t*)p.qkv_bf16 + (int64_t)b * p.ntok * p.ld_qkv + h * hd;
  const uint8_t* qrow = base + (int64_t)(qtok >= 0 ? qtok : 0) * p.ld_qkv;

  f32x4 s[2 * KB];
  int ktok[2 * KB];
#pragma unroll
  for (int kt = 0; kt < 2 * KB; ++kt) {
    const int key = kt * 16 + l16;
    ktok[kt] = (kt < KT && key < vol) ? p.tok_index[c * vol + key] : -1;
    s[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  for (int d0 = 0; d0 < hd; d0 += 32) {
    long qf = 0;
    if (qtok >= 0) qf = *(const long*)(qrow + d0 + 8 * g);
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      long kf = 0;
      if (ktok[kt] >= 0) kf = *(const long*)(base + (int64_t)ktok[kt] * p.ld_qkv + p.C + d0 + 8 * g);
      s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(kf, qf, s[kt], 0, 0, 0);
    }
  }
  // lane: query `query`, keys kt*16 + 4g .. +3; q and k both carry 2^qk
  const float sscale = p.scale * __builtin_amdgcn_ldexpf(1.0f, -2 * p.qkv_fp8_log2);
  float sc[2 * KB][4];
  float mx = -3.0e38f;
#pragma unroll
  for (int kt = 0; kt < 2 * KB; ++kt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int key = kt * 16 + 4 * g + r;
      float v = -INFINITY;
      if (kt < KT && key < vol && query < vol) {
        v = s[kt][r] * sscale + p.bias[((int64_t)h * vol + query) * vol + key];
        if (p.mask && !p.mask[((int64_t)c * vol + query) * vol + key]) v = -1e18f;
      }
      sc[kt][r] = v;
      mx = fmaxf(mx, v);
    }
  mx = attn_rows4_max(mx);
  float sum = 0.f;
#pragma unroll
  for (int kt = 0; kt < 2 * KB; ++kt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float e = expf(sc[kt][r] - mx);   // exp(-inf) = 0 for non-existent keys
      sum += e;
      s[kt][r] = e;
    }
  sum = attn_rows4_sum(sum);
  const float inv = sum > 0.f ? 256.f / sum : 0.f;                       // P * 2^8 -> e4m3 (<= 256 < 448: no saturation needed)
  long pf[KB];
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int kt = 2 * kb + (j >> 2), r = j & 3;
      v[j] = s[kt][r] * inv;
      if (sc[kt][r] <= -1e18f) v[j] = 0.f;   // masked_softmax multiplies by the mask after the softmax
    }
    int w0 = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], 0, false);
    w0 = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], w0, true);
    int w1 = __builtin_amdgcn_cvt_pk_fp8_f32(v[4], v[5], 0, false);
    w1 = __builtin_amdgcn_cvt_pk_fp8_f32(v[6], v[7], w1, true);
    pf[kb] = (long)(((unsigned long long)(uint32_t)w1 << 32) | (unsigned long long)(uint32_t)w0);
  }

  // ---- O^T[d][query] = sum_key V[key][d] P[query][key]: A = V^T, lane (l16, g) holds V[key(g, j)][d0 + l16], j = 0..7 ----
  int vtok[KB][8];
#pragma unroll
  for (int kb = 0; kb < KB; ++kb)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int kt = 2 * kb + (j >> 2), key = kt * 16 + 4 * g + (j & 3);
      vtok[kb][j] = (kt < KT && key < vol) ? p.tok_index[c * vol + key] : -1;
    }
  const uint8_t* vbase = base + 2 * p.C + l16;
  const float oscale = __builtin_amdgcn_ldexpf(1.0f, -8 - p.qkv_fp8_log2);          // P carries 2^8, v carries 2^qk
  for (int d0 = 0; d0 < hd; d0 += 16) {
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      unsigned long long vf = 0;
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (vtok[kb][j] >= 0) vf |= (unsigned long long)vbase[(int64_t)vtok[kb][j] * p.ld_qkv + d0] << (8 * j);
      o = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8((long)vf, pf[kb], o, 0, 0, 0);
    }
    // lane: query, d = d0 + 4g + r
    if (qtok >= 0) {
      if (p.out_fp8_log2 > 0) {         // e4m3 bytes, value * 2^k, saturating: the A operand of an fp8 proj launch
        const float f8s = oscale * __builtin_amdgcn_ldexpf(1.0f, p.out_fp8_log2);
        int w = __builtin_amdgcn_cvt_pk_fp8_f32(fminf(fmaxf(o[0] * f8s, -448.f), 448.f), fminf(fmaxf(o[1] * f8s, -448.f), 448.f), 0, false);
        w = __builtin_amdgcn_cvt_pk_fp8_f32(fminf(fmaxf(o[2] * f8s, -448.f), 448.f), fminf(fmaxf(o[3] * f8s, -448.f), 448.f), w, true);
        *(int*)((uint8_t*)p.out_bf16 + ((int64_t)b * p.ntok + qtok) * p.ld_out + h * hd + 4 * g + d0) = w;
      } else {
        pd_bf16* orow = p.out_bf16 + ((int64_t)b * p.ntok + qtok) * p.ld_out + h * hd + 4 * g;
        *(uint2*)(orow + d0) = make_uint2(pack_op2(o[0] * oscale, o[1] * oscale), pack_op2(o[2] * oscale, o[3] * oscale));
      }
    }
  }
}
#endif

// ------------------------------------------------------------------------------------------------- large cuboids (volume > 64)
// Patterns whose cuboids span the whole grid or a whole frame ("full": 13 x 16 x 16 = 3328 slots; "divided_st": 1 x 16 x 16 = 256;
// cuboid_transformer_patterns.py:11-16,53-58).  One wave per (sample, cuboid, head, tile of 16 queries) walks the key tiles of 16
// with an online softmax (running max m, running denominator l, accumulators rescaled by exp(m_old - m_new)): the score matrix
// never exists.  Same MFMA operand layouts as the kernel above; unnormalised bf16 probabilities, fp32 normalisation at the end.
// Masked entries (-1e18, cuboid_transformer.py:553-557) keep the reference's semantics: they take part in the running max and
// denominator exactly as in a whole-row softmax (a fully masked row ends as zeros), their probabilities are zeroed afterwards.
template <int NDT>   // head_dim / 16
__global__ void __launch_bounds__(256) cuboid_attn_flash_kernel(const pd_cuboid_attn_args p) {
  const int lane = threadIdx.x & 63;
  const int vol = p.vol;
  const int QT = (vol + 15) >> 4;
  const int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);   // (((b * nc) + c) * heads + h) * QT + qt
  const int64_t nitems = (int64_t)p.B * p.nc * p.heads * QT;
  if (item >= nitems) return;
  const int qt = (int)(item % QT);
  const int64_t it2 = item / QT;
  const int h = (int)(it2 % p.heads);
  const int c = (int)((it2 / p.heads) % p.nc);
  const int b = (int)(it2 / ((int64_t)p.heads * p.nc));
  constexpr int hd = NDT * 16;
  const int l16 = lane & 15, g = lane >> 4;
  const int query = qt * 16 + l16;
  const int* tokc = p.tok_index + (int64_t)c * vol;
  const int qtok = query < vol ? tokc[query] : -1;
  const pd_bf16* base = p.qkv_bf16 + (int64_t)b * p.ntok * p.ld_qkv + h * hd;
  op8 qf[hd / 32];
#pragma unroll
  for (int i = 0; i < hd / 32; ++i) {
    qf[i] = op8{0, 0, 0, 0, 0, 0, 0, 0};
    if (qtok >= 0) qf[i] = *(const op8*)(base + (int64_t)qtok * p.ld_qkv + 32 * i + 8 * g);
  }
  f32x4 o[NDT];
#pragma unroll
  for (int i = 0; i < NDT; ++i) o[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m = -3.0e38f, l = 0.f;
  const float* brow = p.bias + ((int64_t)h * vol + (query < vol ? query : 0)) * vol;
  const uint8_t* mrow = p.mask ? p.mask + ((int64_t)c * vol + (query < vol ? query : 0)) * vol : nullptr;
  const pd_bf16* vbase = base + 2 * p.C + l16;
  for (int kt = 0; kt < QT; ++kt) {
    const int krow = kt * 16 + l16;
    const int ktok = krow < vol ? tokc[krow] : -1;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < hd / 32; ++i) {
      op8 kf = {0, 0, 0, 0, 0, 0, 0, 0};
      if (ktok >= 0) kf = *(const op8*)(base + (int64_t)ktok * p.ld_qkv + p.C + 32 * i + 8 * g);
      s = mfma_16x16x32(kf, qf[i], s);
    }
    float sc[4], tmax = -3.0e38f;
    int vtok[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int key = kt * 16 + 4 * g + r;
      float v = -INFINITY;
      vtok[r] = -1;
      if (key < vol) {
        vtok[r] = tokc[key];
        if (query < vol) {
          v = s[r] * p.scale + brow[key];
          if (mrow && !mrow[key]) v = -1e18f;
        }
      }
      sc[r] = v;
      tmax = fmaxf(tmax, v);
    }
    tmax = attn_rows4_max(tmax);
    const float m_new = fmaxf(m, tmax);
    const float alpha = expf(m - m_new);
    float tsum = 0.f;
    s16x4 pf;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float e = expf(sc[r] - m_new);          // exp(-inf) = 0 for non-existent keys
      tsum += e;
      pf[r] = (short)f2op(sc[r] <= -1e18f ? 0.f : e);
    }
    l = l * alpha + attn_rows4_sum(tsum);
    m = m_new;
#pragma unroll
    for (int i = 0; i < NDT; ++i) {
      s16x4 vf;
#pragma unroll
      for (int r = 0; r < 4; ++r) vf[r] = vtok[r] >= 0 ? (short)vbase[(int64_t)vtok[r] * p.ld_qkv + 16 * i] : (short)0;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[i][r] *= alpha;
      o[i] = mfma_16x16x16(vf, pf, o[i]);
    }
  }
  if (qtok < 0) return;
  const float inv = l > 0.f ? 1.f / l : 0.f;
  pd_bf16* orow = p.out_bf16 + ((int64_t)b * p.ntok + qtok) * p.ld_out + h * hd + 4 * g;
#pragma unroll
  for (int i = 0; i < NDT; ++i) {
    // lane: query, d = 16 i + 4g + r  (the accumulators are read by ordinary VALU code here: hipcc pads the MFMA hazard itself)
    const float v0 = o[i][0] * inv, v1 = o[i][1] * inv, v2 = o[i][2] * inv, v3 = o[i][3] * inv;
    *(uint2*)(orow + 16 * i) = make_uint2((uint32_t)f2op(v0) | ((uint32_t)f2op(v1) << 16), (uint32_t)f2op(v2) | ((uint32_t)f2op(v3) << 16));
  }
}

// ------------------------------------------------------------------------------------------------- generic path
template <typename QT>
__device__ __forceinline__ float ldq(const QT* p);
template <>
__device__ __forceinline__ float ldq<float>(const float* p) { return *p; }
template <>
__device__ __forceinline__ float ldq<pd_bf16>(const pd_bf16* p) { return op2f(*p); }

constexpr int GA_MAXVOL = 64;

template <typename QT>
__global__ void __launch_bounds__(256) cuboid_attn_generic_kernel(const pd_cuboid_attn_args p, const QT* __restrict__ qkv) {
  extern __shared__ float sm[];
  const int hd = p.C / p.heads, vol = p.vol;
  float* sq = sm;                       // [vol][hd]
  float* sk = sq + vol * hd;            // [vol][hd+1]
  float* sv = sk + vol * (hd + 1);      // [vol][hd]
  float* ss = sv + vol * hd;            // [vol][vol+1]
  __shared__ int stok[GA_MAXVOL], stok_out[GA_MAXVOL];
  const int64_t item = blockIdx.x;
  const int h = (int)(item % p.heads);
  const int c = (int)((item / p.heads) % p.nc);
  const int b = (int)(item / ((int64_t)p.heads * p.nc));
  const int tid = threadIdx.x;
  if (tid < vol) {
    stok[tid] = p.tok_index[c * vol + tid];
    stok_out[tid] = p.tok_out ? p.tok_out[c * vol + tid] : stok[tid];    // (padding_type "nearest": who receives the slot's result)
  }
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
    const int tok = stok_out[qi];
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
        p.out_bf16[o] = f2op(a);
      }
    }
  }
}

#if !PD_IS_F16
extern "C" int pd_f16_cuboid_attention(const pd_cuboid_attn_args*, pd_stream_t);
extern "C" int pd_f16_softmax_rows(const float*, pd_bf16*, pd_bf16*, int64_t, int, int, int, const pd_call_opts*, pd_stream_t);
#endif

extern "C" int PD_ENTRY(cuboid_attention)(const pd_cuboid_attn_args* pa, pd_stream_t stream) {
  PD_CHECK_ARG(pa != nullptr, "pd_cuboid_attention: null args");
  PD_FORWARD_F16(pa->operand == PD_OPERAND_F16, pd_f16_cuboid_attention(pa, stream));
  const pd_cuboid_attn_args a = *pa;
  PD_CHECK_ARG(!PD_IS_F16 || !a.out_bf16_lo, "pd_cuboid_attention: the hi/lo split exists for bfloat16 operands only");
  PD_CHECK_ARG((a.qkv_bf16 != nullptr) != (a.qkv_f32 != nullptr), "pd_cuboid_attention: exactly one of qkv_bf16 / qkv_f32");
  PD_CHECK_ARG(a.tok_index && a.bias && (a.out_bf16 || a.out_f32), "pd_cuboid_attention: null pointer");
  PD_CHECK_ARG(a.heads > 0 && a.C % a.heads == 0 && a.vol > 0 && a.nc > 0, "pd_cuboid_attention: bad geometry");
  const int hd = a.C / a.heads;
  hipStream_t s = (hipStream_t)stream;
  const int64_t nitems = (int64_t)a.B * a.nc * a.heads;
  const bool mfma_ok = a.qkv_bf16 && a.vol <= 64 && (hd % 32) == 0 && a.out_bf16 && !a.out_f32 && !a.out_bf16_lo &&
                       (a.ld_qkv % 8) == 0 && (a.ld_out % 4) == 0 && !a.force_generic && !a.tok_out;
  PD_CHECK_ARG(!a.tok_out || a.vol <= GA_MAXVOL, "pd_cuboid_attention: a separate output token table (padding_type \"nearest\") runs on the "
               "generic core: cuboid volume %d > %d", a.vol, GA_MAXVOL);
  if (a.out_fp8_log2 > 0 && !mfma_ok) {
    pd_set_error("pd_cuboid_attention: an e4m3 output is built for the MFMA cores only (bf16 q/k/v, cuboid volume <= 64, head_dim %% 32 == 0, out_bf16 alone)");
    return PD_ERR_UNSUPPORTED;
  }
  if (a.qkv_fp8_log2 > 0) {
#if !PD_IS_F16
    // e4m3 q / k / v (precision="fp8"): qkv_bf16 points to BYTES, ld_qkv counts bytes; the 16-bit output (if any) is bfloat16
    PD_CHECK_ARG(mfma_ok && !a.qkv_f32, "pd_cuboid_attention: e4m3 q/k/v run on the MFMA core only (cuboid volume <= 64, head_dim %% 32 == 0, "
                 "out_bf16 alone, no separate output token table)");
    const int kt = (a.vol + 15) / 16;
    const dim3 grid((unsigned)((nitems * kt + 3) / 4));
    switch (kt) {
      case 1: hipLaunchKernelGGL(cuboid_attn_mfma_fp8_kernel<1>, grid, dim3(256), 0, s, a); break;
      case 2: hipLaunchKernelGGL(cuboid_attn_mfma_fp8_kernel<2>, grid, dim3(256), 0, s, a); break;
      case 3: hipLaunchKernelGGL(cuboid_attn_mfma_fp8_kernel<3>, grid, dim3(256), 0, s, a); break;
      default: hipLaunchKernelGGL(cuboid_attn_mfma_fp8_kernel<4>, grid, dim3(256), 0, s, a); break;
    }
    PD_CHECK_LAUNCH();
    return PD_OK;
#else
    pd_set_error("pd_cuboid_attention: e4m3 q/k/v belong to the bfloat16 engine (operand = PD_OPERAND_BF16)");
    return PD_ERR_UNSUPPORTED;
#endif
  }
  if (mfma_ok) {
    const int kt = (a.vol + 15) / 16;
    const int64_t nwaves = nitems * kt;                  // one wave per 16-query tile
    const dim3 grid((unsigned)((nwaves + 3) / 4));
    switch (kt) {
      case 1: hipLaunchKernelGGL(cuboid_attn_mfma_kernel<1>, grid, dim3(256), 0, s, a); break;
      case 2: hipLaunchKernelGGL(cuboid_attn_mfma_kernel<2>, grid, dim3(256), 0, s, a); break;
      case 3: hipLaunchKernelGGL(cuboid_attn_mfma_kernel<3>, grid, dim3(256), 0, s, a); break;
      default: hipLaunchKernelGGL(cuboid_attn_mfma_kernel<4>, grid, dim3(256), 0, s, a); break;
    }
    PD_CHECK_LAUNCH();
    return PD_OK;
  }
  if (a.vol > GA_MAXVOL && a.qkv_bf16 && (hd == 32 || hd == 64 || hd == 128) && a.out_bf16 && !a.out_f32 && !a.out_bf16_lo && (a.ld_qkv % 8) == 0 &&
      (a.ld_out % 4) == 0) {
    // large cuboids ("full", "divided_st"): online-softmax MFMA kernel
    const int qt = (a.vol + 15) / 16;
    const dim3 grid((unsigned)((nitems * qt + 3) / 4));
    if (hd == 32) hipLaunchKernelGGL(cuboid_attn_flash_kernel<2>, grid, dim3(256), 0, s, a);
    else if (hd == 64) hipLaunchKernelGGL(cuboid_attn_flash_kernel<4>, grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(cuboid_attn_flash_kernel<8>, grid, dim3(256), 0, s, a);
    PD_CHECK_LAUNCH();
    return PD_OK;
  }
  if (a.vol > GA_MAXVOL || hd > 128) {
    pd_set_error("pd_cuboid_attention: cuboid volume %d > %d needs bf16 q/k/v with head_dim 32 / 64 / 128 (precision=\"bf16\"); "
                 "head_dim %d (max 128)", a.vol, GA_MAXVOL, hd);
    return PD_ERR_UNSUPPORTED;
  }
  const size_t lds = sizeof(float) * ((size_t)a.vol * hd * 2 + (size_t)a.vol * (hd + 1) + (size_t)a.vol * (a.vol + 1));
  static bool attr_set_dev[PD_MAX_DEVICES];
  bool& attr_set = attr_set_dev[pd_cur_device()];
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

#if !PD_IS_F16
// ------------------------------------------------------------------------------------------------- data gradient (guidance network)
// d(loss)/d(q, k, v) of the cuboid attention above, fp32 throughout, one workgroup per (sample, cuboid, head).  The probabilities are
// recomputed from q, k and the bias (nothing but qkv is kept from the forward); with P = softmax(s q k^T + bias) (masked entries 0):
//   dV = P^T dO,   dP = dO V^T,   dS = P o (dP - rowsum(P o dP)),   dQ = s dS K,   dK = s dS^T Q.
// This is what autograd derives for cuboid_transformer.py:852-861,947-949 inside the reference's knowledge-alignment gradient
// (alignment.py:60-66 -> models.py:459-528); the relative-position table and the weights get no gradient here (sampling only).
__global__ void __launch_bounds__(256) cuboid_attn_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ d_out,
                                                              const int32_t* __restrict__ tok_index, const float* __restrict__ bias,
                                                              const uint8_t* __restrict__ mask, float* __restrict__ d_qkv, int B, int ntok,
                                                              int C, int heads, int nc, int vol, int ld_qkv, int ld_dout, int ld_dqkv,
                                                              float scale) {
  extern __shared__ float sm[];
  const int hd = C / heads;
  float* sq = sm;                        // [vol][hd]    q * scale
  float* sk = sq + vol * hd;             // [vol][hd+1]
  float* sv = sk + vol * (hd + 1);       // [vol][hd+1]
  float* sdo = sv + vol * (hd + 1);      // [vol][hd]
  float* sp = sdo + vol * hd;            // [vol][vol+1] P
  float* sds = sp + vol * (vol + 1);     // [vol][vol+1] dP, then dS
  __shared__ int stok[GA_MAXVOL];
  const int64_t item = blockIdx.x;
  const int h = (int)(item % heads);
  const int c = (int)((item / heads) % nc);
  const int b = (int)(item / ((int64_t)heads * nc));
  const int tid = threadIdx.x;
  if (tid < vol) stok[tid] = tok_index[c * vol + tid];
  __syncthreads();
  for (int i = tid; i < vol * hd; i += 256) {
    const int r = i / hd, d = i - r * hd;
    const int tok = stok[r];
    float qv = 0.f, kv = 0.f, vv = 0.f, gv = 0.f;
    if (tok >= 0) {
      const float* row = qkv + ((int64_t)b * ntok + tok) * ld_qkv + h * hd + d;
      qv = row[0]; kv = row[C]; vv = row[2 * C];
      gv = d_out[((int64_t)b * ntok + tok) * ld_dout + h * hd + d];
    }
    sq[r * hd + d] = qv * scale;
    sk[r * (hd + 1) + d] = kv;
    sv[r * (hd + 1) + d] = vv;
    sdo[r * hd + d] = gv;
  }
  __syncthreads();
  for (int i = tid; i < vol * vol; i += 256) {
    const int qi = i / vol, kj = i - qi * vol;
    float a = 0.f, g = 0.f;
    for (int d = 0; d < hd; ++d) {
      a += sq[qi * hd + d] * sk[kj * (hd + 1) + d];
      g += sdo[qi * hd + d] * sv[kj * (hd + 1) + d];
    }
    a += bias[((int64_t)h * vol + qi) * vol + kj];
    if (mask && !mask[((int64_t)c * vol + qi) * vol + kj]) a = -1e18f;
    sp[qi * (vol + 1) + kj] = a;
    sds[qi * (vol + 1) + kj] = g;
  }
  __syncthreads();
  const int lane = tid & 63, wv = tid >> 6;
  for (int qi = wv; qi < vol; qi += 4) {
    const float v = lane < vol ? sp[qi * (vol + 1) + lane] : -INFINITY;
    const float mx = wave_max(v);
    const float e = lane < vol ? expf(v - mx) : 0.f;
    const float sum = wave_sum(e);
    const float pr = (lane < vol && v > -1e18f) ? e / sum : 0.f;
    const float g = lane < vol ? sds[qi * (vol + 1) + lane] : 0.f;
    const float delta = wave_sum(pr * g);
    if (lane < vol) {
      sp[qi * (vol + 1) + lane] = pr;
      sds[qi * (vol + 1) + lane] = pr * (g - delta);
    }
  }
  __syncthreads();
  for (int i = tid; i < vol * hd; i += 256) {
    const int r = i / hd, d = i - r * hd;
    const int tok = stok[r];
    if (tok < 0) continue;
    float dq = 0.f, dk = 0.f, dv = 0.f;
    for (int j = 0; j < vol; ++j) {
      dq += sds[r * (vol + 1) + j] * sk[j * (hd + 1) + d];
      dk += sds[j * (vol + 1) + r] * sq[j * hd + d];      // sq already carries the scale
      dv += sp[j * (vol + 1) + r] * sdo[j * hd + d];
    }
    float* row = d_qkv + ((int64_t)b * ntok + tok) * ld_dqkv + h * hd + d;
    row[0] = dq * scale; row[C] = dk; row[2 * C] = dv;
  }
}

extern "C" int pd_cuboid_attention_bwd(const float* qkv, const float* d_out, const int32_t* tok_index, const float* bias,
                                       const uint8_t* mask, float* d_qkv, int B, int ntok, int C, int heads, int nc, int vol,
                                       int ld_qkv, int ld_dout, int ld_dqkv, float scale, pd_stream_t stream) {
  PD_CHECK_ARG(qkv && d_out && tok_index && bias && d_qkv, "pd_cuboid_attention_bwd: null pointer");
  PD_CHECK_ARG(heads > 0 && C % heads == 0 && vol > 0 && nc > 0 && B >= 0, "pd_cuboid_attention_bwd: bad geometry");
  const int hd = C / heads;
  if (vol > GA_MAXVOL || hd > 128) {
    pd_set_error("pd_cuboid_attention_bwd: cuboid volume %d (max %d) / head_dim %d (max 128) not supported", vol, GA_MAXVOL, hd);
    return PD_ERR_UNSUPPORTED;
  }
  if (B == 0) return PD_OK;
  const size_t lds = sizeof(float) * ((size_t)vol * hd * 2 + (size_t)vol * (hd + 1) * 2 + (size_t)vol * (vol + 1) * 2);
  if (lds > 160 * 1024 - 512) {     // e.g. volume 64 with head_dim 128: 164,864 B
    pd_set_error("pd_cuboid_attention_bwd: volume %d x head_dim %d needs %zu B of LDS (max %d)", vol, hd, lds, 160 * 1024 - 512);
    return PD_ERR_UNSUPPORTED;
  }
  static bool attr_set_dev[PD_MAX_DEVICES];
  bool& attr_set = attr_set_dev[pd_cur_device()];
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)cuboid_attn_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512);
    attr_set = true;
  }
  const int64_t nitems = (int64_t)B * nc * heads;
  hipLaunchKernelGGL(cuboid_attn_bwd_kernel, dim3((unsigned)nitems), dim3(256), lds, (hipStream_t)stream, qkv, d_out, tok_index, bias, mask,
                     d_qkv, B, ntok, C, heads, nc, vol, ld_qkv, ld_dout, ld_dqkv, scale);
  PD_CHECK_LAUNCH();
  return PD_OK;
}

#endif   // !PD_IS_F16 (fp32 backward: one copy)

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
      out[row * ld_out + i] = f2op(v);
    }
  }
}

extern "C" int PD_ENTRY(softmax_rows)(const float* x, pd_bf16* out, pd_bf16* out_lo, int64_t rows, int n, int ld_in, int ld_out,
                                      const pd_call_opts* opts, pd_stream_t stream) {
  PD_FORWARD_F16(PD_OPTS_F16(opts), pd_f16_softmax_rows(x, out, out_lo, rows, n, ld_in, ld_out, opts, stream));
  PD_CHECK_ARG(!PD_IS_F16 || !out_lo, "pd_softmax_rows: the hi/lo split exists for bfloat16 operands only");
  PD_CHECK_ARG(x && out && n > 0 && ld_in >= n && ld_out >= n, "pd_softmax_rows: bad args");
  if (rows <= 0) return PD_OK;
  hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, out, out_lo, rows, n,
                     ld_in, ld_out);
  PD_CHECK_LAUNCH();
  return PD_OK;
}

}  // namespace PD_NS
