// LayerNorm of a BM-row block (16 rows per wave) straight into the swizzled bf16 A tile of the fused kernels (ffn.hip, attn_block.hip).
//
// A row is spread over the 16 lanes of one DPP row (C/64 float4 per lane), four rows per wave instruction: the two reductions
// of a row are 4 v_add_f32 with DPP modifiers each (quad_perm xor 1, xor 2, row_half_mirror, row_mirror) -- no LDS crossbar.
// The 64-lane form (2 x 6 dependent ds_bpermute per row, 16 rows per wave) was latency bound: 6 us of every 128-row block.
#pragma once
#include "common.h"


template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
// sum over the 16 lanes of a DPP row, result in every lane
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_move<0xB1>(v);     // quad_perm [1,0,3,2]
  v += dpp_move<0x4E>(v);     // quad_perm [2,3,0,1]
  v += dpp_move<0x141>(v);    // row_half_mirror
  v += dpp_move<0x140>(v);    // row_mirror
  return v;
}

// wave `wave` normalises rows wave*RPW .. wave*RPW+RPW-1 of the block (RPW = 16, or 8 when 8 waves share a 64-row block);
// row_of(r) gives the global row (or -1: the tile row is zero).
// sA: KS = C/64 slabs of [BM rows][64 k] bf16, 16 B chunk index XOR (row >> 1) & 7 (the layout the MFMA fragment reads expect).
// while_loading(): independent work done while the row loads are in flight (table set-up of the caller); default none.
struct ln_no_overlap { __device__ __forceinline__ void operator()() const {} };
template <int C, int BM = 128, int RPW = 16, typename RowOf, typename Overlap = ln_no_overlap>
__device__ __forceinline__ void ln_block_to_tile(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                 float eps, char* sA, int wave, int lane, bool skip_loads, RowOf row_of,
                                                 Overlap while_loading = Overlap()) {
  constexpr int NV = C / 64;                       // float4 per lane
  const int rsub = lane >> 4, j = lane & 15;
  float4 g4[NV], b4[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    g4[i] = *(const float4*)(gamma + (j + 16 * i) * 4);
    b4[i] = *(const float4*)(beta + (j + 16 * i) * 4);
  }
  constexpr int NB = RPW / 4;                      // four rows per wave instruction
  float4 xv[NB][NV];                               // all rows of this wave in flight at once
  int mrow[NB];
#pragma unroll
  for (int bt = 0; bt < NB; ++bt) {
    mrow[bt] = row_of(wave * RPW + bt * 4 + rsub);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      xv[bt][i] = make_float4(0, 0, 0, 0);
      if (mrow[bt] >= 0 && !skip_loads) xv[bt][i] = *(const float4*)(x + (int64_t)mrow[bt] * C + (j + 16 * i) * 4);
    }
  }
  while_loading();
#pragma unroll
  for (int bt = 0; bt < NB; ++bt) {
    const int row = wave * RPW + bt * 4 + rsub;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) s += (xv[bt][i].x + xv[bt][i].y) + (xv[bt][i].z + xv[bt][i].w);
    const float mean = row16_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const float a = xv[bt][i].x - mean, b = xv[bt][i].y - mean, c = xv[bt][i].z - mean, d = xv[bt][i].w - mean;
      q += (a * a + b * b) + (c * c + d * d);
    }
    const float rstd = rsqrtf(row16_sum(q) / (float)C + eps);
    const bool ok = mrow[bt] >= 0;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      float y0 = (xv[bt][i].x - mean) * rstd * g4[i].x + b4[i].x, y1 = (xv[bt][i].y - mean) * rstd * g4[i].y + b4[i].y;
      float y2 = (xv[bt][i].z - mean) * rstd * g4[i].z + b4[i].z, y3 = (xv[bt][i].w - mean) * rstd * g4[i].w + b4[i].w;
      if (!ok) y0 = y1 = y2 = y3 = 0.f;
      // columns 4j + 64i .. +3: slab i, 16 B chunk j >> 1, half j & 1
      const int off = i * (BM * 128) + row * 128 + (((j >> 1) ^ ((row >> 1) & 7)) << 4) + ((j & 1) << 3);
      *(uint2*)(sA + off) = make_uint2(pack_op2(y0, y1), pack_op2(y2, y3));
    }
  }
}
