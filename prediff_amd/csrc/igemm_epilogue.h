// Epilogue shared by the pd_igemm kernels: a wave's fp32 accumulator slab (WM rows x WNS columns, row major in LDS) ->
// alpha, bias, per-sample row vector (timestep embedding), activation, gate multiply, fp32 residual add -> coalesced
// 16 B row segments of the fp32 and/or bf16 (hi[/lo]) outputs.
#pragma once
#include <type_traits>
#include "common.h"

// The row loop is instantiated per (columns-per-lane, activation, operand presence) so that the hot call sites get
// straight-line code; tag value 2 = "decide at run time" (generic instantiation).
//   CW columns per lane: 8 when only bf16 is stored (16 B stores: the 8 B/lane form is store-issue bound), else 4.
template <int WM, int WNS, int CW, int ACT, int RV, int MU, int RS, int OF, int OB, int OL>
__device__ __forceinline__ void igemm_epilogue_rows(const pd_igemm_args& p, const float* sC, int lane, int m_base, int m_end, int n_base,
                                                    float* outf, pd_bf16* outb, pd_bf16* outbl, const float* res) {
  auto on = [](int tag, bool rt) { return tag == 2 ? rt : (tag == 1); };
  constexpr int LPR = WNS / CW;                    // lanes per row
  constexpr int RPP = 64 / LPR;                    // rows per pass
  const int c0 = (lane % LPR) * CW;
  const int n = n_base + c0;
  const bool vec = p.vec_epilogue && (n + CW - 1 < p.N);
  const bool has_rv = on(RV, p.rowvec != nullptr), has_mu = on(MU, p.mul != nullptr), has_rs = on(RS, res != nullptr);
  const bool has_of = on(OF, outf != nullptr), has_ob = on(OB, outb != nullptr), has_ol = on(OL, outbl != nullptr);
  const int act = ACT >= 0 ? ACT : ((p.debug_flags & 4) ? 0 : p.act);
  float bias_v[CW];
#pragma unroll
  for (int e = 0; e < CW; ++e) bias_v[e] = (p.bias && n + e < p.N) ? p.bias[n + e] : 0.f;
  if (n >= p.N || (p.debug_flags & 2)) return;
#pragma unroll 1
  for (int pass = 0; pass < WM / RPP; ++pass) {
    const int row = pass * RPP + lane / LPR;
    const int m = m_base + row;
    if (m >= m_end) continue;
    float v[CW];
#pragma unroll
    for (int q = 0; q < CW / 4; ++q) {
      const float4 a4 = *(const float4*)(sC + row * WNS + c0 + 4 * q);
      v[4 * q] = a4.x; v[4 * q + 1] = a4.y; v[4 * q + 2] = a4.z; v[4 * q + 3] = a4.w;
    }
    const float* rv = has_rv ? p.rowvec + (int64_t)(m / p.rows_per_sample) * p.ld_rowvec + n : nullptr;
    const float* mu = has_mu ? p.mul + (int64_t)m * p.ld_mul + n : nullptr;
    const float* rs = has_rs ? res + (int64_t)(p.res_period ? m % p.res_period : m) * p.ld_res + n : nullptr;
    if (vec) {
#pragma unroll
      for (int e = 0; e < CW; ++e) v[e] = v[e] * p.alpha + bias_v[e];
      if (has_rv) {
#pragma unroll
        for (int q = 0; q < CW / 4; ++q) { const float4 t4 = *(const float4*)(rv + 4 * q); v[4 * q] += t4.x; v[4 * q + 1] += t4.y; v[4 * q + 2] += t4.z; v[4 * q + 3] += t4.w; }
      }
      if (act != 0) {
        // a 16-bit-only producer (FFN-1 of the single-pass engines; compile-time in the hot instantiation, the same rule at run time in
        // the generic ones): the 9-instruction sigmoid-form GELU, 2.5e-5 from the erf form and rounded to 16 bits right below; an fp32
        // or hi/lo output (the fp32-class engine) keeps act_apply's erf form
        const bool only16 = (ACT == PD_ACT_GELU && OL == 0 && OF == 0) || (has_ob && !has_ol && !has_of);
#pragma unroll
        for (int e = 0; e < CW; ++e) v[e] = only16 ? act_apply16(v[e], act) : act_apply(v[e], act);
      }
      if (has_mu) {
#pragma unroll
        for (int q = 0; q < CW / 4; ++q) { const float4 t4 = *(const float4*)(mu + 4 * q); v[4 * q] *= t4.x; v[4 * q + 1] *= t4.y; v[4 * q + 2] *= t4.z; v[4 * q + 3] *= t4.w; }
      }
      if (has_rs) {
#pragma unroll
        for (int q = 0; q < CW / 4; ++q) { const float4 t4 = *(const float4*)(rs + 4 * q); v[4 * q] += t4.x; v[4 * q + 1] += t4.y; v[4 * q + 2] += t4.z; v[4 * q + 3] += t4.w; }
      }
      if (has_of) {
#pragma unroll
        for (int q = 0; q < CW / 4; ++q)
          *(float4*)(outf + (int64_t)m * p.ld_out + n + 4 * q) = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
      }
      if (CW == 8 && has_ob && p.out_fp8_log2 > 0) {
        // e4m3 bytes, value * 2^k, round to nearest even, saturating: the A operand of a following fp8 launch
        const float f8s = __builtin_amdgcn_ldexpf(1.0f, p.out_fp8_log2);
        int w[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          float y[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) y[k] = fminf(fmaxf(v[4 * q + k] * f8s, -448.f), 448.f);
          w[q] = __builtin_amdgcn_cvt_pk_fp8_f32(y[0], y[1], 0, false);
          w[q] = __builtin_amdgcn_cvt_pk_fp8_f32(y[2], y[3], w[q], true);
        }
        *(uint2*)((uint8_t*)outb + (int64_t)m * p.ld_outb + n) = make_uint2((uint32_t)w[0], (uint32_t)w[1]);
      } else if (has_ob) {
        uint32_t hi[CW / 2], lo[CW / 2];
#pragma unroll
        for (int e = 0; e < CW / 2; ++e) {
          if (has_ol) {
            uint16_t h0, l0, h1, l1;
            f2bf_split(v[2 * e], h0, l0);
            f2bf_split(v[2 * e + 1], h1, l1);
            hi[e] = h0 | ((uint32_t)h1 << 16);
            lo[e] = l0 | ((uint32_t)l1 << 16);
          } else {
            hi[e] = pack_op2(v[2 * e], v[2 * e + 1]);
          }
        }
        if constexpr (CW == 8) {
          *(uint4*)(outb + (int64_t)m * p.ld_outb + n) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
          if (has_ol) *(uint4*)(outbl + (int64_t)m * p.ld_outb + n) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
        } else {
          *(uint2*)(outb + (int64_t)m * p.ld_outb + n) = make_uint2(hi[0], hi[1]);
          if (has_ol) *(uint2*)(outbl + (int64_t)m * p.ld_outb + n) = make_uint2(lo[0], lo[1]);
        }
      }
    } else {
      for (int e = 0; e < CW; ++e) {
        if (n + e >= p.N) break;
        float x = v[e] * p.alpha + bias_v[e];
        if (rv) x += rv[e];
        x = (has_ob && !has_ol && !has_of) ? act_apply16(x, act) : act_apply(x, act);
        if (mu) x *= mu[e];
        if (rs) x += rs[e];
        if (has_of) outf[(int64_t)m * p.ld_out + n + e] = x;
        if (has_ob) {
          uint16_t h, l;
          f2bf_split(x, h, l);
          outb[(int64_t)m * p.ld_outb + n + e] = h;
          if (has_ol) outbl[(int64_t)m * p.ld_outb + n + e] = l;
        }
      }
    }
  }
}

// sC: this wave's WM x WNS fp32 slab; (m_base, n_base): its position in the output; rows >= m_end are not stored; bz: batched-GEMM index.
template <int WM, int WNS>
__device__ __forceinline__ void igemm_epilogue(const pd_igemm_args& p, const float* sC, int lane, int m_base, int m_end, int n_base, int bz) {
  float* outf = p.out_f32 ? p.out_f32 + (int64_t)bz * p.out_batch_stride : nullptr;
  pd_bf16* outb = p.out_bf16 ? p.out_bf16 + (int64_t)bz * p.outb_batch_stride : nullptr;
  pd_bf16* outbl = p.out_bf16_lo ? p.out_bf16_lo + (int64_t)bz * p.outb_batch_stride : nullptr;
  const float* res = p.residual ? p.residual + (int64_t)bz * p.res_batch_stride : nullptr;
  const bool rvp = p.rowvec != nullptr, mup = p.mul != nullptr, rsp = res != nullptr, ofp = outf != nullptr, obp = outb != nullptr,
             olp = outbl != nullptr;
  const int actv = (p.debug_flags & 4) ? 0 : p.act;
  constexpr int GELU = PD_ACT_GELU;
  if (p.vec_epilogue == 2 && !rvp && !mup && !rsp && !ofp && obp && !olp && (actv == 0 || actv == GELU)) {
    // bf16-only producers: QKV (no activation), FFN-1 (GELU)
    if (actv == 0) igemm_epilogue_rows<WM, WNS, 8, 0, 0, 0, 0, 0, 1, 0>(p, sC, lane, m_base, m_end, n_base, outf, outb, outbl, res);
    else igemm_epilogue_rows<WM, WNS, 8, GELU, 0, 0, 0, 0, 1, 0>(p, sC, lane, m_base, m_end, n_base, outf, outb, outbl, res);
  } else if (p.vec_epilogue && ofp && !obp && !mup && actv == 0 && (rsp != rvp)) {
    // fp32 residual-stream writers: proj / FFN-2 / conv-2 (+residual), conv-1 (+timestep embedding)
    if (rsp) igemm_epilogue_rows<WM, WNS, 4, 0, 0, 0, 1, 1, 0, 0>(p, sC, lane, m_base, m_end, n_base, outf, outb, outbl, res);
    else igemm_epilogue_rows<WM, WNS, 4, 0, 1, 0, 0, 1, 0, 0>(p, sC, lane, m_base, m_end, n_base, outf, outb, outbl, res);
  } else if (p.vec_epilogue == 2) {
    igemm_epilogue_rows<WM, WNS, 8, -1, 2, 2, 2, 2, 2, 2>(p, sC, lane, m_base, m_end, n_base, outf, outb, outbl, res);
  } else {
    igemm_epilogue_rows<WM, WNS, 4, -1, 2, 2, 2, 2, 2, 2>(p, sC, lane, m_base, m_end, n_base, outf, outb, outbl, res);
  }
}
