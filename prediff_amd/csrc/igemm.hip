// pd_igemm: implicit GEMM (Linear / Conv2d 3x3 / Conv3d 3x3x3) on gfx950 MFMA, bf16 inputs, fp32 accumulate.
//
// Replaces (see include/prediff_hip.h for the per-call-site list): nn.Linear, nn.Conv3d 3x3x3 pad 1 on channels-last
// (B,T,H,W,C), nn.Conv2d 3x3 with fused nearest x2 up-sampling or stride-2 asymmetric-pad down-sampling.
//
// Structure (one workgroup = 256 threads = 4 waves in a 2x2 grid, tile BM x BN x 64):
//   * operands are K-contiguous in HBM: A rows = C_in channels of one (shifted) input position, W rows = C_in of
//     one output channel for one filter tap  ->  both tiles are staged with `global_load_lds_dwordx4` (16 B/lane
//     DMA into LDS, no VGPR round trip).  The DMA writes LDS lane-linearly, so the bank-conflict swizzle is applied
//     to the per-lane SOURCE address (which k-chunk a lane fetches) and undone by the ds_read_b128 address
//     (cdna_hip_programming.md rule 21).  Out-of-image filter taps / M,N tails fetch from a zero page instead.
//   * double-buffered LDS, one barrier per K-step: loads of step k+1 are in flight while the MFMAs of step k run.
//   * v_mfma_f32_32x32x16_bf16, each wave owns (BM/2) x (BN/2) of the tile.
//   * epilogue in registers: alpha, bias, per-sample row vector (timestep embedding), activation, gate multiply,
//     fp32 residual add, fp32 and/or bf16 (hi[/lo]) stores.
//   * blockIdx -> tile mapping gives each XCD a contiguous range of tiles (A rows are then re-used out of that
//     XCD's L2 across the N tiles and across the overlapping filter taps).
//   * SPLIT: x = hi + lo bf16 decomposition of both operands, 3 MFMAs per fragment pair (drops lo*lo): fp32-class
//     accuracy (~2^-16 relative per product) at 1/3 of the bf16 MFMA rate instead of 1/16 for the f32 MFMA.
#include "common.h"

__device__ __attribute__((aligned(64))) uint32_t g_pd_zero_page[32];   // 128 B of zeros (never written)

#define GLDS16(gptr, ldsptr)                                                                   \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),      \
                                   (__attribute__((address_space(3))) void*)(ldsptr), 16, 0, 0)

// KIND only tags the instantiation (0 linear, 1 conv2d, 2 conv3d) so that profilers report the three uses separately.
template <int BM, int BN, bool SPLIT, int KIND>
__global__ void __launch_bounds__(256, 2) igemm_kernel(const pd_igemm_args p) {
  constexpr int BK = 64;
  constexpr int A_TILE = BM * BK * 2;   // bytes
  constexpr int B_TILE = BN * BK * 2;
  constexpr int NP = SPLIT ? 2 : 1;
  constexpr int STAGE = (A_TILE + B_TILE) * NP;
  constexpr int AI = BM / 32, BI = BN / 32;       // 16 B DMA instructions per thread per tile
  constexpr int TM = BM / 64, TN = BN / 64;       // 32x32 MFMA tiles per wave
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // ---- XCD-aware tile id (bijective for any tile count) ----
  const int tiles_n = (p.N + BN - 1) / BN;
  const int tiles_m = (p.M + BM - 1) / BM;
  const int nt = tiles_m * tiles_n;
  int t;
  {
    const int bid = blockIdx.x, xcd = bid & 7, q = nt >> 3, r = nt & 7;
    t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int m0 = (t / tiles_n) * BM;
  const int n0 = (t % tiles_n) * BN;
  const int bz = blockIdx.z;

  const pd_bf16* __restrict__ Ag = p.A + (int64_t)bz * p.a_batch_stride;
  const pd_bf16* __restrict__ Wg = p.W + (int64_t)bz * p.w_batch_stride;
  const pd_bf16* __restrict__ Alo = SPLIT ? p.A_lo + (int64_t)bz * p.a_batch_stride : nullptr;
  const pd_bf16* __restrict__ Wlo = SPLIT ? p.W_lo + (int64_t)bz * p.w_batch_stride : nullptr;
  const pd_bf16* zero = (const pd_bf16*)g_pd_zero_page;

  // ---- staging descriptors (fixed per thread) ----
  const int srow = tid >> 3;                      // row inside a 32-row slab
  const int schunk = (tid & 7) ^ ((tid >> 4) & 7);   // logical 16 B k-chunk fetched by this lane (source-side swizzle)
  const int hw_o = p.Ho * p.Wo, thw_o = p.To * hw_o;
  int vt0[AI], vh0[AI], vw0[AI];
  uint32_t abase[AI];
  bool mok[AI];
#pragma unroll
  for (int i = 0; i < AI; ++i) {
    const int m = m0 + i * 32 + srow;
    mok[i] = m < p.M;
    const int mm = mok[i] ? m : 0;
    const int b = mm / thw_o, r1 = mm - b * thw_o;
    const int ot = r1 / hw_o, r2 = r1 - ot * hw_o;
    const int oh = r2 / p.Wo, ow = r2 - oh * p.Wo;
    vt0[i] = ot * p.st - p.pt;
    vh0[i] = oh * p.sh - p.ph;
    vw0[i] = ow * p.sw - p.pw;
    abase[i] = (uint32_t)b * (uint32_t)(p.Ti * p.Hi * p.Wi);
  }
  uint32_t woff[BI];
#pragma unroll
  for (int i = 0; i < BI; ++i) {
    const int n = n0 + i * 32 + srow;
    woff[i] = n < p.N ? (uint32_t)n * (uint32_t)p.ldw + schunk * 8 : 0xffffffffu;
  }
  const int kchunks = p.Cin >> 6;
  const int nk = p.taps * kchunks;
  const int vT = p.vT > 0 ? p.vT : p.Ti * p.ut, vH = p.vH > 0 ? p.vH : p.Hi * p.uh, vW = p.vW > 0 ? p.vW : p.Wi * p.uw;
  const int khw = p.KH * p.KW;

  uint32_t aoff[AI];   // element offset of (row, this tap, chunk) or 0xffffffff
  auto set_tap = [&](int tap) {
    const int kt = tap / khw, r = tap - kt * khw;
    const int kh = r / p.KW, kw = r - kh * p.KW;
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const int vt = vt0[i] + kt, vh = vh0[i] + kh, vw = vw0[i] + kw;
      const bool ok = mok[i] && (unsigned)vt < (unsigned)vT && (unsigned)vh < (unsigned)vH && (unsigned)vw < (unsigned)vW;
      const int it = p.ut == 2 ? vt >> 1 : vt, ih = p.uh == 2 ? vh >> 1 : vh, iw = p.uw == 2 ? vw >> 1 : vw;
      aoff[i] = ok ? (abase[i] + (uint32_t)((it * p.Hi + ih) * p.Wi + iw)) * (uint32_t)p.lda + schunk * 8 : 0xffffffffu;
    }
  };

  auto issue = [&](int stage, int ks) {
    const int tap = ks / kchunks, kc = ks - tap * kchunks;
    if (kc == 0) set_tap(tap);
    char* sbase = smem + stage * STAGE;
    const uint32_t kofs = kc * 64;
    const int64_t wtap = (int64_t)tap * p.w_tap_stride + kofs;
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const bool ok = aoff[i] != 0xffffffffu;
      char* dst = sbase + (i * 256 + wave * 64) * 16;
      GLDS16(ok ? Ag + aoff[i] + kofs : zero, dst);
      if (SPLIT) GLDS16(ok ? Alo + aoff[i] + kofs : zero, dst + A_TILE);
    }
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      const bool ok = woff[i] != 0xffffffffu;
      char* dst = sbase + A_TILE * NP + (i * 256 + wave * 64) * 16;
      GLDS16(ok ? Wg + wtap + woff[i] : zero, dst);
      if (SPLIT) GLDS16(ok ? Wlo + wtap + woff[i] : zero, dst + B_TILE);
    }
  };

  // ---- accumulators ----
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int wr = wave >> 1, wc = wave & 1;
  const int lrow = lane & 31, lhalf = lane >> 5;
  const int swz = (lrow >> 1) & 7;
  const int a_row_b = (wr * (BM / 2) + lrow) * 128;   // byte offset of this lane's row in the A tile
  const int b_row_b = (wc * (BN / 2) + lrow) * 128;

  issue(0, 0);
  for (int ks = 0; ks < nk; ++ks) {
    __syncthreads();   // stage ks&1 has landed (vmcnt(0) precedes the barrier); stage (ks+1)&1 is free again
    if (ks + 1 < nk) issue((ks + 1) & 1, ks + 1);
    const char* sA = smem + (ks & 1) * STAGE;
    const char* sB = sA + A_TILE * NP;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int pos = ((kk * 2 + lhalf) ^ swz) * 16;
      bf16x8 a[TM], b[TN], al[TM], bl[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        a[i] = *(const bf16x8*)(sA + a_row_b + i * 32 * 128 + pos);
        if (SPLIT) al[i] = *(const bf16x8*)(sA + A_TILE + a_row_b + i * 32 * 128 + pos);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        b[j] = *(const bf16x8*)(sB + b_row_b + j * 32 * 128 + pos);
        if (SPLIT) bl[j] = *(const bf16x8*)(sB + B_TILE + b_row_b + j * 32 * 128 + pos);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          if (SPLIT) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], b[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], bl[j], acc[i][j], 0, 0, 0);
          }
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
  }

  // ---- epilogue: accumulators -> per-wave LDS slab (row major) -> coalesced 16 B row segments ----
  constexpr int WM = BM / 2, WN = BN / 2;
  __syncthreads();                                 // every wave is done reading the operand stages
  float* sC = (float*)smem + wave * (WM * WN);
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        sC[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhalf) * WN + j * 32 + lrow] = acc[i][j][r];
  __syncthreads();

  float* outf = p.out_f32 ? p.out_f32 + (int64_t)bz * p.out_batch_stride : nullptr;
  pd_bf16* outb = p.out_bf16 ? p.out_bf16 + (int64_t)bz * p.outb_batch_stride : nullptr;
  pd_bf16* outbl = p.out_bf16_lo ? p.out_bf16_lo + (int64_t)bz * p.outb_batch_stride : nullptr;
  const float* res = p.residual ? p.residual + (int64_t)bz * p.res_batch_stride : nullptr;
  constexpr int LPR = WN / 4;                      // lanes per row
  constexpr int RPP = 64 / LPR;                    // rows per pass
  const int c4 = (lane % LPR) * 4;
  const int n = n0 + wc * WN + c4;
  const bool vec = p.vec_epilogue && (n + 3 < p.N);
  float bias4[4] = {0.f, 0.f, 0.f, 0.f};
  if (p.bias) {
#pragma unroll
    for (int e = 0; e < 4; ++e) if (n + e < p.N) bias4[e] = p.bias[n + e];
  }
#pragma unroll 1
  for (int pass = 0; pass < WM / RPP; ++pass) {
    const int row = pass * RPP + lane / LPR;
    const int m = m0 + wr * WM + row;
    if (m >= p.M || n >= p.N) continue;
    const float4 a4 = *(const float4*)(sC + row * WN + c4);
    float v[4] = {a4.x, a4.y, a4.z, a4.w};
    const float* rv = p.rowvec ? p.rowvec + (int64_t)(m / p.rows_per_sample) * p.ld_rowvec + n : nullptr;
    const float* mu = p.mul ? p.mul + (int64_t)m * p.ld_mul + n : nullptr;
    const float* rs = res ? res + (int64_t)(p.res_period ? m % p.res_period : m) * p.ld_res + n : nullptr;
    if (vec) {
      float4 t;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = v[e] * p.alpha + bias4[e];
      if (rv) { t = *(const float4*)rv; v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w; }
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = act_apply(v[e], p.act);
      if (mu) { t = *(const float4*)mu; v[0] *= t.x; v[1] *= t.y; v[2] *= t.z; v[3] *= t.w; }
      if (rs) { t = *(const float4*)rs; v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w; }
      if (outf) *(float4*)(outf + (int64_t)m * p.ld_out + n) = make_float4(v[0], v[1], v[2], v[3]);
      if (outb) {
        uint16_t hi[4], lo[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) f2bf_split(v[e], hi[e], lo[e]);
        *(uint2*)(outb + (int64_t)m * p.ld_outb + n) = make_uint2(hi[0] | ((uint32_t)hi[1] << 16), hi[2] | ((uint32_t)hi[3] << 16));
        if (outbl)
          *(uint2*)(outbl + (int64_t)m * p.ld_outb + n) = make_uint2(lo[0] | ((uint32_t)lo[1] << 16), lo[2] | ((uint32_t)lo[3] << 16));
      }
    } else {
      for (int e = 0; e < 4; ++e) {
        if (n + e >= p.N) break;
        float x = v[e] * p.alpha + bias4[e];
        if (rv) x += rv[e];
        x = act_apply(x, p.act);
        if (mu) x *= mu[e];
        if (rs) x += rs[e];
        if (outf) outf[(int64_t)m * p.ld_out + n + e] = x;
        if (outb) {
          uint16_t hi, lo;
          f2bf_split(x, hi, lo);
          outb[(int64_t)m * p.ld_outb + n + e] = hi;
          if (outbl) outbl[(int64_t)m * p.ld_outb + n + e] = lo;
        }
      }
    }
  }
}

template <int BM, int BN, bool SPLIT, int KIND>
static int launch_igemm(const pd_igemm_args& a, hipStream_t s) {
  constexpr int lds = 2 * (BM + BN) * 64 * 2 * (SPLIT ? 2 : 1);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)igemm_kernel<BM, BN, SPLIT, KIND>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) {
      pd_set_error("pd_igemm: hipFuncSetAttribute(%d) failed: %s", lds, hipGetErrorString(e));
      return PD_ERR_LAUNCH;
    }
    attr_set = true;
  }
  const int tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
  dim3 grid(tiles, 1, a.nbatch > 0 ? a.nbatch : 1);
  hipLaunchKernelGGL((igemm_kernel<BM, BN, SPLIT, KIND>), grid, dim3(256), lds, s, a);
  PD_CHECK_LAUNCH();
  return PD_OK;
}

extern "C" int pd_igemm(const pd_igemm_args* pa, pd_stream_t stream) {
  PD_CHECK_ARG(pa != nullptr, "pd_igemm: null args");
  pd_igemm_args a = *pa;
  PD_CHECK_ARG(a.A && a.W, "pd_igemm: A/W null");
  PD_CHECK_ARG(a.M > 0 && a.N > 0 && a.taps > 0, "pd_igemm: bad M/N/taps (%d,%d,%d)", a.M, a.N, a.taps);
  PD_CHECK_ARG(a.Cin > 0 && (a.Cin & 63) == 0, "pd_igemm: Cin=%d must be a positive multiple of 64 (zero padded)", a.Cin);
  PD_CHECK_ARG((a.lda & 7) == 0 && (a.ldw & 7) == 0, "pd_igemm: lda/ldw must be multiples of 8 (16 B rows)");
  PD_CHECK_ARG(a.taps == a.KT * a.KH * a.KW, "pd_igemm: taps != KT*KH*KW");
  PD_CHECK_ARG((int64_t)a.B * a.To * a.Ho * a.Wo == a.M, "pd_igemm: M != B*To*Ho*Wo");
  PD_CHECK_ARG((a.ut == 1 || a.ut == 2) && (a.uh == 1 || a.uh == 2) && (a.uw == 1 || a.uw == 2), "pd_igemm: bad upsample");
  PD_CHECK_ARG((int64_t)a.B * a.Ti * a.Hi * a.Wi * (int64_t)a.lda < (1ll << 32) - 64, "pd_igemm: A too large for 32-bit offsets");
  PD_CHECK_ARG((int64_t)a.N * a.ldw < (1ll << 32) - 64, "pd_igemm: W tap too large for 32-bit offsets");
  PD_CHECK_ARG(!a.split || (a.A_lo && a.W_lo), "pd_igemm: split needs A_lo and W_lo");
  PD_CHECK_ARG(!a.rowvec || a.rows_per_sample > 0, "pd_igemm: rowvec needs rows_per_sample");
  PD_CHECK_ARG(a.out_f32 || a.out_bf16, "pd_igemm: no output");
  hipStream_t s = (hipStream_t)stream;
  a.vec_epilogue = ((a.N & 3) == 0) && (!a.out_f32 || (a.ld_out & 3) == 0) && (!a.out_bf16 || (a.ld_outb & 3) == 0) &&
                   (!a.residual || (a.ld_res & 3) == 0) && (!a.rowvec || (a.ld_rowvec & 3) == 0) &&
                   (!a.mul || (a.ld_mul & 3) == 0) && (((uintptr_t)a.out_f32 | (uintptr_t)a.residual | (uintptr_t)a.rowvec |
                   (uintptr_t)a.mul) & 15) == 0 && (((uintptr_t)a.out_bf16 | (uintptr_t)a.out_bf16_lo) & 7) == 0 &&
                   ((a.out_batch_stride | a.outb_batch_stride | a.res_batch_stride) & 3) == 0;
  int tile = a.tile;
  if (tile == 0) {
    const int64_t t128 = (int64_t)((a.M + 127) / 128) * ((a.N + 127) / 128) * (a.nbatch > 0 ? a.nbatch : 1);
    tile = t128 >= 192 ? 1 : 2;
  }
  const int kind = a.taps == 1 ? 0 : (a.KT == 1 ? 1 : 2);
#define PD_DISPATCH(KIND)                                                                                         \
  if (a.split) return tile == 1 ? launch_igemm<128, 128, true, KIND>(a, s) : launch_igemm<64, 64, true, KIND>(a, s); \
  return tile == 1 ? launch_igemm<128, 128, false, KIND>(a, s) : launch_igemm<64, 64, false, KIND>(a, s);
  if (kind == 0) { PD_DISPATCH(0) }
  if (kind == 1) { PD_DISPATCH(1) }
  PD_DISPATCH(2)
#undef PD_DISPATCH
}
