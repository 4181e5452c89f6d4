// pd_igemm, long-K variant: 256 x 256 x 64 tile, 8 waves (2 x 4), one workgroup per CU, bf16 MFMA 32x32x16, fp32 accumulate.
//
// The 128 x 128 two-barrier kernel in igemm.hip tops out near 0.9-1.0 PFLOP/s whatever its pipeline depth; this variant is
// built around the structure that removes that ceiling (cdna_hip_programming.md "256^2 8-phase"):
//   * each wave owns 128 x 64 of the tile (4 x 2 MFMA tiles, 128 accumulator registers) -> 2/3 of the LDS bytes per MFMA;
//   * every K-tile (64 deep) is consumed in 4 phases of 8 MFMAs (one 64 x 32 quadrant, K = 64 each); a phase is
//     [ds_read of the fragments it adds | DMA issue] s_barrier [8 MFMAs] s_barrier;
//   * the two wave rows run ONE barrier apart (wave row 1 takes an extra s_barrier before the loop, wave row 0 one after),
//     so the two waves that share a SIMD alternate: one is in its MFMA section while the other reads LDS / issues DMA;
//   * operands arrive by buffer_load ... lds DMA (16 B/lane) as 128-row x 128-B half tiles (A half 0 / 1 is read only by wave
//     row 0 / 1; W rows 0-127 / 128-255 by wave columns 0-1 / 2-3), two K-tile buffers, one counted s_waitcnt vmcnt per
//     K-tile, never 0 in the steady state, issue spread over the phases (2, 2, 0, 4 instructions):
//        phase 0 of K-tile kt: issue A half 0 of kt+1      phase 1: A half 1 of kt+1
//        phase 3:              issue both W halves of kt+2, then vmcnt(4) -> everything of kt+1 has landed
//     Hazards (S(n) = the interval after the n-th workgroup barrier; wave row 0 reads in S(2p), row 1 in S(2p+1)):
//        W of kt is last read in phase 1 (row 1: S(8kt+3), retired by its lgkmcnt(0) in S(8kt+4)); its buffer is refilled
//        from phase 3 (S(8kt+6)).  A half 0 / 1 of kt is last read in phase 2 (S(8kt+4) / S(8kt+5)) and refilled with kt+2
//        in phase 0 / 1 of kt+1 (S(8kt+8) / S(8kt+10)).  Data of kt+1 is waited for in phase 3 of kt by every wave BEFORE
//        a barrier that precedes any read of it (row 0 waits before barrier 8kt+7, row 1 before 8kt+8, first read S(8kt+8)).
//     (Issuing every half tile the moment its buffer is free -- 4 phases of lead instead of 2-3 -- measured 5-8 % SLOWER:
//      the lever is the even interleave of DMA issue with the MFMA sections, not the prefetch distance.)
//   * (tried and dropped: variable tile heights, h x 32 rows per round with absent row tiles skipping their reads and MFMAs,
//     to turn e.g. 416 tiles on 256 CUs into a 224-row + a 192-row tile per CU.  A phase is not MFMA-bound -- barrier pair
//     + LDS reads + DMA issue cost about as much as its 8 MFMAs -- so shorter tiles were not faster, and the wave-uniform
//     branches cost 10 % everywhere.)
//   * (tried and dropped: rotating the temporal tap order per tile so that the three output slices reading one input slice do
//     so in the same third of their loops -- FETCH_SIZE went UP (L1 convs 109 -> 260 MB per launch); the un-rotated order
//     already re-uses a slice one third-of-a-loop after the neighbour tile touched it.)
//   * the swizzle, descriptors, zero-fill of out-of-image taps, XCD-contiguous tile order and epilogue are those of igemm.hip.
// Supported: bf16 (non-split) operands, Cin % 64 == 0, KIND 0 (row-wise linear) and KIND 2 (stride-1, un-upsampled
// Conv2d/Conv3d gather, any padding); pd_igemm picks it for long-K launches with enough rows (see igemm.hip).
#include "common.h"
#include "igemm_epilogue.h"

#define BLDS16(rsrc, ldsptr, voff, soff) \
  __builtin_amdgcn_raw_ptr_buffer_load_lds((rsrc), (__attribute__((address_space(3))) void*)(ldsptr), 16, (voff), (soff), 0, 0)
#define PD_OOB 0xffffff00u
#define VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define PHASE_SYNC()                   \
  __builtin_amdgcn_sched_barrier(0);   \
  __builtin_amdgcn_s_barrier();        \
  __builtin_amdgcn_sched_barrier(0)

namespace {
constexpr int HT = 128 * 128;   // bytes of one half tile: 128 rows x 64 bf16
constexpr int KBUF = 4 * HT;    // one K-tile buffer: A half 0, A half 1, W half 0, W half 1

}

// 8 MFMAs of one quadrant: 2 row tiles x 4 k-substeps, accumulators alternate so that no MFMA waits on its predecessor
#define QUAD_MFMA(accA, accB, bfrag)                                                       \
  _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) {                                       \
    accA = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][kk], bfrag[kk], accA, 0, 0, 0);    \
    accB = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][kk], bfrag[kk], accB, 0, 0, 0);    \
  }

template <int KIND>
__global__ void __launch_bounds__(512) igemm256_kernel(const pd_igemm_args p) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;

  // ---- XCD-aware tile id (bijective for any tile count) ----
  const int tiles_n = (p.N + 255) >> 8;
  const int tiles_m = (p.M + 255) >> 8;
  const int nt = tiles_m * tiles_n;
  int t;
  {
    const int bid = blockIdx.x, xcd = bid & 7, q = nt >> 3, r = nt & 7;
    t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int m0 = (t / tiles_n) << 8;
  const int n0 = (t % tiles_n) << 8;
  const int bz = blockIdx.z;

  const auto rA = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + (int64_t)bz * p.a_batch_stride), 0, p.a_bytes, 0x00020000);
  const auto rW = __builtin_amdgcn_make_buffer_rsrc((void*)(p.W + (int64_t)bz * p.w_batch_stride), 0, p.w_bytes, 0x00020000);

  // ---- staging: one DMA instruction covers 64 rows x 128 B; thread -> (row tid/8, 16 B slot tid%8), swizzled source chunk ----
  const int srow = tid >> 3, spos = tid & 7;
  const int schunk = spos ^ ((srow >> 1) & 7);
  uint32_t aoff[2][2];    // [half][i]  byte offset of (row, current tap, chunk) or PD_OOB
  uint32_t acoord[2][2];  // KIND 2: ot | oh << 10 | ow << 20 | invalid << 31
  uint32_t abase[2][2];   // KIND 2: first input row of the row's sample
#pragma unroll
  for (int hh = 0; hh < 2; ++hh)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int m = m0 + hh * 128 + i * 64 + srow;
      const bool ok = m < p.M;
      if (KIND == 0) {
        aoff[hh][i] = ok ? ((uint32_t)m * (uint32_t)p.lda + schunk * 8) * 2u : PD_OOB;
      } else {
        const int hw_o = p.Ho * p.Wo, thw_o = p.To * hw_o;
        const int mm = ok ? m : 0;
        const int b = mm / thw_o, r1 = mm - b * thw_o;
        const int ot = r1 / hw_o, r2 = r1 - ot * hw_o;
        const int oh = r2 / p.Wo, ow = r2 - oh * p.Wo;
        acoord[hh][i] = (uint32_t)ot | ((uint32_t)oh << 10) | ((uint32_t)ow << 20) | (ok ? 0u : 0x80000000u);
        abase[hh][i] = (uint32_t)b * (uint32_t)(p.Ti * p.Hi * p.Wi);
        aoff[hh][i] = PD_OOB;
      }
    }
  uint32_t woff[2][2];
#pragma unroll
  for (int hh = 0; hh < 2; ++hh)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int n = n0 + hh * 128 + i * 64 + srow;
      woff[hh][i] = n < p.N ? ((uint32_t)n * (uint32_t)p.ldw + schunk * 8) * 2u : PD_OOB;
    }
  const int kchunks = p.Cin >> 6;
  const int nk = p.taps * kchunks;
  const int khw = p.KH * p.KW;
  char* const dma_dst = smem + wave * (8 * 128);   // + half * HT + i * (64 * 128) + buffer * KBUF  (lane * 16 is implicit)

  auto set_tap = [&](int hh, int tap) {
    const int kt = tap / khw, r = tap - kt * khw;
    const int kh = r / p.KW, kw = r - kh * p.KW;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const uint32_t c = acoord[hh][i];
      const int vt = (int)(c & 1023u) - p.pt + kt, vh = (int)((c >> 10) & 1023u) - p.ph + kh, vw = (int)((c >> 20) & 1023u) - p.pw + kw;
      const bool ok = !(c >> 31) && (unsigned)vt < (unsigned)p.Ti && (unsigned)vh < (unsigned)p.Hi && (unsigned)vw < (unsigned)p.Wi;
      aoff[hh][i] = ok ? ((abase[hh][i] + (uint32_t)((vt * p.Hi + vh) * p.Wi + vw)) * (uint32_t)p.lda + schunk * 8) * 2u : PD_OOB;
    }
  };
  // K-tile counters of the DMA streams (A runs one K-tile ahead of the MFMAs, W two); all wave-uniform scalars
  int a_tap = 0, a_kc = 0, w_kc = 0;
  uint32_t w_tap_b = 0;                                   // byte offset of the current W tap
  const uint32_t w_tap_stride_b = (uint32_t)p.w_tap_stride * 2u;
  auto issue_a = [&](int hh, int buf) {    // A half hh of the A stream's current K-tile
    if (KIND != 0 && a_kc == 0) set_tap(hh, a_tap);
    const int ka = __builtin_amdgcn_readfirstlane(a_kc * 128);
    char* dst = dma_dst + buf * KBUF + hh * HT;
    BLDS16(rA, dst, aoff[hh][0], ka);
    BLDS16(rA, dst + 64 * 128, aoff[hh][1], ka);
  };
  auto next_a = [&]() { if (++a_kc == kchunks) { a_kc = 0; ++a_tap; } };
  auto issue_w = [&](int buf) {            // both W halves of the W stream's current K-tile
    const int kw = __builtin_amdgcn_readfirstlane((int)(w_tap_b + (uint32_t)w_kc * 128u));
    char* dst = dma_dst + buf * KBUF + 2 * HT;
    BLDS16(rW, dst, woff[0][0], kw);
    BLDS16(rW, dst + 64 * 128, woff[0][1], kw);
    BLDS16(rW, dst + HT, woff[1][0], kw);
    BLDS16(rW, dst + HT + 64 * 128, woff[1][1], kw);
    if (++w_kc == kchunks) { w_kc = 0; w_tap_b += w_tap_stride_b; }
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int lrow = lane & 31, lhalf = lane >> 5;
  const int swz = (lrow >> 1) & 7;
  // fragment addresses inside a K-tile buffer: + row-tile * (32 * 128); the 16 B slot of k-substep kk is ((kk*2 + lhalf) ^ swz)
  const int a_rd = wr * HT + lrow * 128;
  const int b_rd = (2 + (wc >> 1)) * HT + ((wc & 1) * 64 + lrow) * 128;

  // ---- prologue: K-tile 0 (A + W) and W of K-tile 1 ----
  if (nk > 0) {
    issue_a(0, 0);
    issue_a(1, 0);
    next_a();
    issue_w(0);
    if (nk > 1) {
      issue_w(1);
      VMCNT(4);
    } else {
      VMCNT(0);
    }
  }
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();   // wave row 1 runs one barrier behind wave row 0
  __builtin_amdgcn_sched_barrier(0);

  bf16x8 a[2][4], b0[4], b1[4];
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    const char* sA = smem + cur * KBUF + a_rd;
    const char* sB = smem + cur * KBUF + b_rd;
    const bool more1 = kt + 1 < nk, more2 = kt + 2 < nk;

    // ---------- phase 0: W tile 0, A row tiles 0-1; quadrant (A0, W0) ----------
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) b0[kk] = *(const bf16x8*)(sB + (((kk * 2 + lhalf) ^ swz) * 16));
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) a[i][kk] = *(const bf16x8*)(sA + i * (32 * 128) + (((kk * 2 + lhalf) ^ swz) * 16));
    if (more1) issue_a(0, cur ^ 1);
    PHASE_SYNC();
    __builtin_amdgcn_s_setprio(1);
    QUAD_MFMA(acc[0][0], acc[1][0], b0)
    __builtin_amdgcn_s_setprio(0);
    PHASE_SYNC();

    // ---------- phase 1: W tile 1; quadrant (A0, W1) ----------
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) b1[kk] = *(const bf16x8*)(sB + 32 * 128 + (((kk * 2 + lhalf) ^ swz) * 16));
    if (more1) { issue_a(1, cur ^ 1); next_a(); }
    PHASE_SYNC();
    __builtin_amdgcn_s_setprio(1);
    QUAD_MFMA(acc[0][1], acc[1][1], b1)
    __builtin_amdgcn_s_setprio(0);
    PHASE_SYNC();

    // ---------- phase 2: A row tiles 2-3; quadrant (A1, W1) ----------
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) a[i][kk] = *(const bf16x8*)(sA + (2 + i) * (32 * 128) + (((kk * 2 + lhalf) ^ swz) * 16));
    PHASE_SYNC();
    __builtin_amdgcn_s_setprio(1);
    QUAD_MFMA(acc[2][1], acc[3][1], b1)
    __builtin_amdgcn_s_setprio(0);
    PHASE_SYNC();

    // ---------- phase 3: nothing to read (W tile 0 is still in registers); quadrant (A1, W0); W of kt+2; wait for kt+1 ----------
    if (more2) {
      issue_w(cur);
      VMCNT(4);
    } else {
      VMCNT(0);
    }
    PHASE_SYNC();
    __builtin_amdgcn_s_setprio(1);
    QUAD_MFMA(acc[2][0], acc[3][0], b0)
    __builtin_amdgcn_s_setprio(0);
    PHASE_SYNC();
  }
  if (wr == 0) __builtin_amdgcn_s_barrier();   // re-join the two wave rows

  // ---- epilogue: two 32-column slabs per wave (8 waves x 128 x 32 fp32 = 128 KB = the operand buffers) ----
  float* sC = (float*)smem + wave * (128 * 32);
  const int m_base = m0 + wr * 128;
#pragma unroll
  for (int js = 0; js < 2; ++js) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) sC[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhalf) * 32 + lrow] = acc[i][js][r];
    __syncthreads();
    igemm_epilogue<128, 32>(p, sC, lane, m_base, p.M, n0 + wc * 64 + js * 32, bz);
  }
#endif
}

template <int KIND>
static int launch256(const pd_igemm_args& a, hipStream_t s) {
  constexpr int lds = 2 * KBUF;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)igemm256_kernel<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) {
      pd_set_error("pd_igemm: hipFuncSetAttribute(%d) failed: %s", lds, hipGetErrorString(e));
      return PD_ERR_LAUNCH;
    }
    attr_set = true;
  }
  const int tiles = ((a.M + 255) / 256) * ((a.N + 255) / 256);
  dim3 grid(tiles, 1, a.nbatch > 0 ? a.nbatch : 1);
  hipLaunchKernelGGL((igemm256_kernel<KIND>), grid, dim3(512), lds, s, a);
  PD_CHECK_LAUNCH();
  return PD_OK;
}

// true when the 256-tile kernel can run this (already validated) launch
bool pd_igemm256_supported(const pd_igemm_args& a, int kind) {
  if (a.split) return false;
  if (kind == 0) return true;
  return a.st == 1 && a.sh == 1 && a.sw == 1 && a.ut == 1 && a.uh == 1 && a.uw == 1 && a.vT <= 0 && a.vH <= 0 && a.vW <= 0 &&
         a.To < 1024 && a.Ho < 1024 && a.Wo < 1024;
}

int pd_igemm256_launch(const pd_igemm_args& a, int kind, hipStream_t s) {
  return kind == 0 ? launch256<0>(a, s) : launch256<2>(a, s);
}
