// pd_igemm, long-K variant: 256 x 256 x 64 tile, 8 waves (2 x 4), one workgroup per CU, bf16 MFMA 16x16x32, fp32 accumulate.
//
// The 128 x 128 two-barrier kernel in igemm.hip tops out near 0.9-1.0 PFLOP/s whatever its pipeline depth; this variant is
// built around the structure that removes that ceiling (cdna_hip_programming.md "256^2 8-phase"):
//   * each wave owns 128 x 64 of the tile (8 x 4 MFMA tiles of 16 x 16, 128 accumulator registers) -> 2/3 of the LDS bytes per
//     MFMA; 16x16x32 tiles measured 12 % faster than 32x32x16 here (388 vs 441 us on the level-0 Conv3d, 1.25 vs 1.09 PFLOP/s on
//     a 4096^3 GEMM): half-length MFMAs interleave better with the fragment reads and leave no dependent issue pairs;
//   * every K-tile (64 deep) is consumed in 4 phases of 16 MFMAs (one 64 x 32 quadrant, K = 64 each); a phase is
//     [ds_read of the fragments it adds | DMA issue] s_barrier [16 MFMAs] s_barrier;
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
#include <algorithm>
#include "common.h"
#include "igemm_epilogue.h"

namespace PD_NS {

#define BLDS16(rsrc, ldsptr, voff, soff) \
  __builtin_amdgcn_raw_ptr_buffer_load_lds((rsrc), (__attribute__((address_space(3))) void*)(ldsptr), 16, (voff), (soff), 0, 0)
#define PD_OOB 0xffffff00u
#define VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define PHASE_SYNC()                   \
  __builtin_amdgcn_sched_barrier(0);   \
  __builtin_amdgcn_s_barrier();        \
  __builtin_amdgcn_sched_barrier(0)

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4v __attribute__((ext_vector_type(4)));

namespace {
// one lane's operand fragments of a 16-row tile for a whole K-tile: two 16 B LDS slots, (0*4 + lg) ^ swz and (1*4 + lg) ^ swz.
// bf16: the operands of the two k-steps; fp8: together the 32 B operand of one scaled MFMA (loaded straight into the halves of an
// 8-register tuple so that no copies are needed)
template <bool F8>
struct Frag256 {
  op8 v[2];
  __device__ __forceinline__ void load(const char* base, int lg, int swz) {
    v[0] = *(const op8*)(base + ((lg ^ swz) * 16));
    v[1] = *(const op8*)(base + (((4 + lg) ^ swz) * 16));
  }
};
template <>
struct Frag256<true> {
  i32x8 v;
  __device__ __forceinline__ void load(const char* base, int lg, int swz) {
    const i32x4v l = *(const i32x4v*)(base + ((lg ^ swz) * 16)), h = *(const i32x4v*)(base + (((4 + lg) ^ swz) * 16));
    v = __builtin_shufflevector(l, h, 0, 1, 2, 3, 4, 5, 6, 7);
  }
};
__device__ __forceinline__ f32x4 mfma_f8(const Frag256<true>& a, const Frag256<true>& b, f32x4 c) {
  return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a.v, b.v, c, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);   // e4m3 x e4m3, block scales 2^0
}
__device__ __forceinline__ f32x4 mfma_f8(const Frag256<false>&, const Frag256<false>&, f32x4 c) { return c; }
__device__ __forceinline__ f32x4 mfma_op16(const Frag256<false>& a, const Frag256<false>& b, int ks, f32x4 c) {
  return mfma_16x16x32(a.v[ks], b.v[ks], c);
}
__device__ __forceinline__ f32x4 mfma_op16(const Frag256<true>&, const Frag256<true>&, int, f32x4 c) { return c; }
constexpr int HT = 128 * 128;   // bytes of one half tile: 128 rows x 64 bf16
constexpr int KBUF = 4 * HT;    // one K-tile buffer: A half 0, A half 1, W half 0, W half 1

}

// RT = row tiles (of 16) per wave row: 8 -> the 256-row tile.  RT = 7 gives a 208-row tile (wave row 1 = rows 96-207, the 16
// shared rows computed twice and stored once) that turns the 416 / 208 tiles of the SEVIR-LR convolutions at 32 trajectories
// (M = trajectories * 13 * H * W) into exactly 512 / 256 -- measured: identical launch times (388 vs 391 us, 318 vs 318 us), once
// more because a phase is bound by its barrier pair, LDS reads and the W stream, not by its MFMA count; only RT = 8 is built.
// SK: split-K launch (small grids).  blockIdx.y = K-slice; the slice's K-tiles [kt0, kt1) run through the unchanged main loop and
// the raw fp32 accumulators go to slab `slice` of p.splitk_ws ([ksplit][M][N]); igemm_splitk_reduce_kernel sums the slabs in slice
// order (deterministic) and applies the epilogue.
// F8: OCP e4m3 operands (one byte per element).  A K-tile is still 128 B of every row -- 128 elements instead of 64 -- so the
// staging, swizzle, DMA schedule and LDS fragment reads are byte for byte those of the bf16 kernel; the two 16 B fragments a lane
// reads per 16 x 16 tile become the 32 B operand of ONE v_mfma_scale_f32_16x16x128_f8f6f4 (unit E8M0 block scales; the tensor
// scales ride in p.alpha) where bf16 issues two 16x16x32 MFMAs: the same MFMA cycles per K-tile for twice the K.  Both operands
// use the same (lane, byte) -> k assignment, which is all a dot product needs.
// SP: the hi/lo split engine (precision="fp32": x = hi + lo in bfloat16, three products per fragment pair).  A K-tile is still 128 B of every
// row: 32 elements of the HIGH parts (16-B slots 0-3) followed by the same 32 elements of the LOW parts (slots 4-7), fetched by ONE DMA
// instruction from the two arrays -- hi and lo behind one buffer descriptor, the lane's offset picks the array (the engine allocates the
// two halves of an operand in one tensor; the launcher checks that they are less than 4 GB apart).  The fragment reads are unchanged: the
// "first k-step" registers hold the high parts, the "second k-step" registers the low parts, and a K-tile is  acc += lo.hi; acc += hi.lo;
// acc += hi.hi  per accumulator -- the products, their order and the 32-deep k-steps of igemm_kernel<.., SPLIT = true>, so the two
// kernels give the SAME bits (the fp32-class engine stays bit-identical across batch sizes whichever kernel a launch size selects).
// Same DMA bytes and LDS reads per K-tile as the bf16 kernel for 24 instead of 16 MFMAs per phase.
// P2 (default since round 5; debug_flags bit 64 = the four-phase form of rounds 2-4 for A/B): a K-tile in TWO phases of 32 MFMAs (quadrants
// (A0, W0) + (A0, W1), then (A1, W1) + (A1, W0)) instead of four of 16 -- the same fragments, registers, DMA schedule and hazards, the same
// products in the same order per accumulator (bit-identical), half the barriers.  Measured at 32 trajectories, interleaved on one box
// (profiles/r05_g_two_phase_ab.txt): Conv3d level 0 392 / 401 -> 385 / 377 us, level 1 320 / 325 -> 311 / 310 us, 4096^3 GEMM 108.4 -> 105.6 us,
// headline 1532 -> 1555 steps/s.
// WF (round 6): folded weights (pd_igemm_args.w_fold, precision="fp16x2") -- the weight slabs of W_lo walk the activation gather of the W_hi slabs.
// A template flag, not a run-time test: the one-product instantiations -- the headline's Conv3d kernel -- keep the instruction stream they had
// (the run-time form added 12 scalar instructions to every K-tile of the hot loop and cost the kernel 2-5 %).
template <int KIND, int RT, bool SK = false, bool F8 = false, bool SP = false, bool P2 = true, bool WF = false>
__global__ void __launch_bounds__(512) igemm256_kernel(const pd_igemm_args p) {
  static_assert(!WF || (!F8 && !SP), "folded weights: 16-bit operands, no hi/lo split");
  static_assert(!SP || (!SK && !F8), "the hi/lo form: bf16 operands, no K-slices");
  constexpr uint32_t EB = F8 ? 1u : 2u;           // bytes per operand element
  constexpr int KSH = F8 ? 7 : SP ? 5 : 6;        // log2(elements per 128 B K-tile row)
  constexpr uint32_t KB = SP ? 64u : 128u;        // bytes of ONE source array per K-tile row
  constexpr int BM = RT == 8 ? 256 : 16 * RT + 96;
  constexpr int ROW1 = BM - 16 * RT;              // first tile row of wave row 1 (128 for RT = 8, 96 for RT = 7)
  constexpr int RA0 = 4, RA1 = RT - 4;            // row tiles of the two A sub-halves
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;

  // ---- XCD-aware tile id (bijective for any tile count) ----
  const int tiles_n = (p.N + 255) >> 8;
  const int tiles_m = (p.M + BM - 1) / BM;
  const int nt = tiles_m * tiles_n;
  int t;
  {
    const int bid = blockIdx.x, xcd = bid & 7, q = nt >> 3, r = nt & 7;
    t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int m0 = (t / tiles_n) * BM;
  const int n0 = (t % tiles_n) << 8;
  const int bz = blockIdx.z;

  // SP: one descriptor over [min(hi, lo), max(hi, lo) + bytes); a_sel / w_sel = this lane's array offset + 16-B slot inside the array's 64 B
  const char* a_lo_p = SP ? (const char*)p.A_lo : (const char*)p.A;
  const char* w_lo_p = SP ? (const char*)p.W_lo : (const char*)p.W;
  const char* a_base = (const char*)p.A < a_lo_p ? (const char*)p.A : a_lo_p;
  const char* w_base = (const char*)p.W < w_lo_p ? (const char*)p.W : w_lo_p;
  const uint32_t a_span = (uint32_t)(((const char*)p.A < a_lo_p ? a_lo_p - (const char*)p.A : (const char*)p.A - a_lo_p));
  const uint32_t w_span = (uint32_t)(((const char*)p.W < w_lo_p ? w_lo_p - (const char*)p.W : (const char*)p.W - w_lo_p));
  const auto rA = __builtin_amdgcn_make_buffer_rsrc((void*)(a_base + (int64_t)bz * p.a_batch_stride * EB), 0, p.a_bytes + a_span, 0x00020000);
  const auto rW = __builtin_amdgcn_make_buffer_rsrc((void*)(w_base + (int64_t)bz * p.w_batch_stride * EB), 0, p.w_bytes + w_span, 0x00020000);

  // ---- staging: one DMA instruction covers 64 rows x 128 B; thread -> (row tid/8, 16 B slot tid%8), swizzled source chunk ----
  const int srow = tid >> 3, spos = tid & 7;
  const int schunk = spos ^ ((srow >> 1) & 7);
  // byte offset of this lane's 16-B slot inside a K-tile row of the source: plain = slot * 16; SP = the array (hi: slots 0-3, lo: 4-7) + slot % 4
  const uint32_t a_sel = !SP ? (uint32_t)schunk * 16u
                             : (uint32_t)(schunk & 3) * 16u + (uint32_t)(((schunk & 4) ? a_lo_p : (const char*)p.A) - a_base);
  const uint32_t w_sel = !SP ? (uint32_t)schunk * 16u
                             : (uint32_t)(schunk & 3) * 16u + (uint32_t)(((schunk & 4) ? w_lo_p : (const char*)p.W) - w_base);
  uint32_t aoff[2][2];    // [half][i]  byte offset of (row, current tap, chunk) or PD_OOB
  uint32_t acoord[2][2];  // KIND 2: ot | oh << 10 | ow << 20 | invalid << 31
  uint32_t abase[2][2];   // KIND 2: first input row of the row's sample
#pragma unroll
  for (int hh = 0; hh < 2; ++hh)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int rloc = i * 64 + srow;                                     // row inside the half (16 * RT rows are used)
      const int m = m0 + hh * ROW1 + rloc;
      const bool ok = m < p.M && rloc < 16 * RT && m < m0 + BM;
      if (KIND == 0) {
        aoff[hh][i] = ok ? ((uint32_t)m * (uint32_t)p.lda) * EB + a_sel : PD_OOB;
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
      woff[hh][i] = n < p.N ? ((uint32_t)n * (uint32_t)p.ldw) * EB + w_sel : PD_OOB;
    }
  const int kchunks = p.Cin >> KSH;
  const int khw = p.KH * p.KW;
  // ---- whole-tile tap skipping (KIND 2, un-split launches): when the input frame of a temporal filter tap lies outside [0, Ti) for
  // EVERY row of the tile -- a tile inside the first / last frame(s) of a sample: at the v1 level-0 grid a 256-row tile IS one frame,
  // so 2 of 13 tiles drop 9 of their 27 taps -- the zero-padding taps are left out of both DMA streams and of the MFMA loop instead of
  // being streamed as zero rows (bit `tap` of tap_skip; debug_flags bit 8 keeps the dense loop for A/B runs).
  uint32_t tap_skip = 0;
  if (KIND == 2 && !SK && p.KT > 1 && p.ut == 1 && p.vT <= 0 && p.taps < 32 && khw < 32 && !WF && !(p.debug_flags & 8)) {   // (< 32: `tap_skip >> taps` stays a defined shift)
    const int hw_o = p.Ho * p.Wo, thw_o = p.To * hw_o;
    const int m_last = min(p.M, m0 + BM) - 1;
    const int b_first = m0 / thw_o;
    if (b_first == m_last / thw_o) {                               // all rows belong to one sample
      const int ot_a = (m0 - b_first * thw_o) / hw_o, ot_b = (m_last - b_first * thw_o) / hw_o;
      for (int kt = 0; kt < p.KT; ++kt)
        if (ot_b - p.pt + kt < 0 || ot_a - p.pt + kt >= p.Ti) tap_skip |= ((1u << khw) - 1u) << (kt * khw);
    }
  }
  const int ntap = p.taps - __builtin_popcount(tap_skip);
  const int nk_all = ntap * kchunks;
  const int kslice = SK ? (int)blockIdx.y : 0;
  const int kt0 = SK ? (int)((int64_t)nk_all * kslice / p.ksplit) : 0;
  const int nk = (p.debug_flags & 1) ? 0 : SK ? (int)((int64_t)nk_all * (kslice + 1) / p.ksplit) - kt0 : nk_all;   // (bit 1: profiling, epilogue only)
  char* const dma_dst = smem + wave * (8 * 128);   // + half * HT + i * (64 * 128) + buffer * KBUF  (lane * 16 is implicit)

  const int vT = p.vT > 0 ? p.vT : p.Ti * p.ut, vH = p.vH > 0 ? p.vH : p.Hi * p.uh, vW = p.vW > 0 ? p.vW : p.Wi * p.uw;
  // (kt, kh, kw) of the A stream's current tap: kept as three wave-uniform counters that next_a() advances -- the two integer divisions of
  // tap -> (kt, kh, kw) by run-time divisors were ~70 scalar instructions in the hot loop at every tap change (round 6: the loop's time follows its
  // scalar instruction count more than its MFMA count suggests -- 12 added ones had cost the kernel 3 %)
  int a_kt = 0, a_kh = 0, a_kw = 0;
  auto set_tap = [&](int hh) {
    const int kt = a_kt, kh = a_kh, kw = a_kw;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const uint32_t c = acoord[hh][i];
      const int vt = (int)(c & 1023u) - p.pt + kt, vh = (int)((c >> 10) & 1023u) - p.ph + kh, vw = (int)((c >> 20) & 1023u) - p.pw + kw;
      // (nearest x2 up-sampling fused in the gather -- Upsample3DLayer / the VAE's Upsample2D: the taps are bounds-checked against the
      //  VIRTUAL, up-sampled size, the row read is the source pixel (v >> 1))
      const bool ok = !(c >> 31) && (unsigned)vt < (unsigned)vT && (unsigned)vh < (unsigned)vH && (unsigned)vw < (unsigned)vW;
      const int it = p.ut == 2 ? vt >> 1 : vt, ih = p.uh == 2 ? vh >> 1 : vh, iw = p.uw == 2 ? vw >> 1 : vw;
      aoff[hh][i] = ok ? ((abase[hh][i] + (uint32_t)((it * p.Hi + ih) * p.Wi + iw)) * (uint32_t)p.lda) * EB + a_sel : PD_OOB;
    }
  };
  // K-tile counters of the DMA streams (A runs one K-tile ahead of the MFMAs, W two); all wave-uniform scalars
  const uint32_t w_tap_stride_b = (uint32_t)p.w_tap_stride * EB;
  int a_tap = SK ? kt0 / kchunks : (tap_skip ? __builtin_ctz(~tap_skip) : 0);       // (first tap that is not skipped)
  int a_kc = SK ? kt0 - a_tap * kchunks : 0, w_kc = a_kc;
  int w_tap = a_tap;
  uint32_t w_tap_b = (uint32_t)a_tap * w_tap_stride_b;     // byte offset of the current W tap
  if (KIND != 0) {                                         // coordinates of the first tap (the only divisions: once per workgroup)
    int t0 = a_tap;
    if constexpr (WF) { if (t0 >= p.w_fold) t0 -= p.w_fold; }
    a_kt = t0 / khw;
    const int r0 = t0 - a_kt * khw;
    a_kh = r0 / p.KW;
    a_kw = r0 - a_kh * p.KW;
  }
  if (SK && KIND != 0 && a_kc != 0) {                      // a slice that starts inside a tap: issue_a only sets a tap up at its first chunk
    set_tap(0);
    set_tap(1);
  }
  auto issue_a = [&](int hh, int buf) {    // A half hh of the A stream's current K-tile
    if (KIND != 0 && a_kc == 0) set_tap(hh);
    const int ka = __builtin_amdgcn_readfirstlane(a_kc * (int)KB);
    char* dst = dma_dst + buf * KBUF + hh * HT;
    BLDS16(rA, dst, aoff[hh][0], ka);
    BLDS16(rA, dst + 64 * 128, aoff[hh][1], ka);
  };
  auto next_a = [&]() {
    if (++a_kc == kchunks) {
      a_kc = 0;
      do {                                                 // (a_tap <= taps < 32 and bit `taps` is never set: the loop stops at the end)
        ++a_tap;
        if (KIND != 0) {
          if (++a_kw == p.KW) {
            a_kw = 0;
            if (++a_kh == p.KH) {
              a_kh = 0;
              ++a_kt;
              if constexpr (WF) { if (a_kt == p.KT) a_kt = 0; }      // the W_lo slabs walk the same activation gather as the W_hi slabs
            }
          }
        }
      } while ((tap_skip >> a_tap) & 1u);
    }
  };
  auto issue_w = [&](int buf) {            // both W halves of the W stream's current K-tile
    const int kw = __builtin_amdgcn_readfirstlane((int)(w_tap_b + (uint32_t)w_kc * KB));
    char* dst = dma_dst + buf * KBUF + 2 * HT;
    BLDS16(rW, dst, woff[0][0], kw);
    BLDS16(rW, dst + 64 * 128, woff[0][1], kw);
    BLDS16(rW, dst + HT, woff[1][0], kw);
    BLDS16(rW, dst + HT + 64 * 128, woff[1][1], kw);
    if (++w_kc == kchunks) {
      w_kc = 0;
      do { ++w_tap; w_tap_b += w_tap_stride_b; } while ((tap_skip >> w_tap) & 1u);
    }
  };

  f32x4 acc[RT][4];
#pragma unroll
  for (int i = 0; i < RT; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;

  const int l16 = lane & 15, lg = lane >> 4;
  const int swz = (l16 >> 1) & 7;
  // fragment addresses inside a K-tile buffer: + tile * (16 * 128); the 16 B slot of k-step ks (32 deep) is ((ks*4 + lg) ^ swz)
  const int a_rd = wr * HT + l16 * 128;
  const int b_rd = (2 + (wc >> 1)) * HT + ((wc & 1) * 64 + l16) * 128;

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

  Frag256<F8> a[4], b0[2], b1[2];
  // one quadrant: NR row tiles x 2 column tiles x K = 64 (two k-steps); consecutive MFMAs hit different accumulators
#define QUAD16(NR, R0, C0, bfrag)                                                                                         \
  if constexpr (F8) {                                                                                                     \
    _Pragma("unroll") for (int i = 0; i < (NR); ++i)                                                                      \
      _Pragma("unroll") for (int c = 0; c < 2; ++c)                                                                       \
        acc[(R0) + i][(C0) + c] = mfma_f8(a[i], bfrag[c], acc[(R0) + i][(C0) + c]);                                       \
    /* pin the phase's MFMAs between its two barriers: without the anchors LLVM sinks the (side-effect free) intrinsic */  \
    /* calls of phases 0-2 past the barriers into phase 3 and the wave rows stop alternating (measured: 1.4x -> see DESIGN) */ \
    _Pragma("unroll") for (int i = 0; i < (NR); ++i)                                                                      \
      _Pragma("unroll") for (int c = 0; c < 2; ++c) asm volatile("" : "+v"(acc[(R0) + i][(C0) + c]));                     \
  } else if constexpr (SP) {                                                                                              \
    _Pragma("unroll") for (int i = 0; i < (NR); ++i)                                                                      \
      _Pragma("unroll") for (int c = 0; c < 2; ++c) {                                                                     \
        acc[(R0) + i][(C0) + c] = mfma_16x16x32(a[i].v[1], bfrag[c].v[0], acc[(R0) + i][(C0) + c]);   /* lo . hi */        \
        acc[(R0) + i][(C0) + c] = mfma_16x16x32(a[i].v[0], bfrag[c].v[1], acc[(R0) + i][(C0) + c]);   /* hi . lo */        \
        acc[(R0) + i][(C0) + c] = mfma_16x16x32(a[i].v[0], bfrag[c].v[0], acc[(R0) + i][(C0) + c]);   /* hi . hi */        \
      }                                                                                                                   \
  } else {                                                                                                                \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                                      \
      _Pragma("unroll") for (int i = 0; i < (NR); ++i)                                                                    \
        _Pragma("unroll") for (int c = 0; c < 2; ++c)                                                                     \
          acc[(R0) + i][(C0) + c] = mfma_op16(a[i], bfrag[c], ks, acc[(R0) + i][(C0) + c]);                               \
  }
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    const char* sA = smem + cur * KBUF + a_rd;
    const char* sB = smem + cur * KBUF + b_rd;
    const bool more1 = kt + 1 < nk, more2 = kt + 2 < nk;

    if constexpr (P2) {
      // ---------- phase A: all four W column tiles, A row tiles 0-3; quadrants (A0, W0), (A0, W1); A of kt+1 ----------
#pragma unroll
      for (int c = 0; c < 2; ++c) b0[c].load(sB + c * (16 * 128), lg, swz);
#pragma unroll
      for (int i = 0; i < RA0; ++i) a[i].load(sA + i * (16 * 128), lg, swz);
#pragma unroll
      for (int c = 0; c < 2; ++c) b1[c].load(sB + (2 + c) * (16 * 128), lg, swz);
      if (more1) { issue_a(0, cur ^ 1); issue_a(1, cur ^ 1); next_a(); }
      PHASE_SYNC();
      __builtin_amdgcn_s_setprio(1);
      QUAD16(RA0, 0, 0, b0)
      QUAD16(RA0, 0, 2, b1)
      __builtin_amdgcn_s_setprio(0);
      PHASE_SYNC();
      // ---------- phase B: A row tiles 4 .. RT-1; quadrants (A1, W1), (A1, W0); W of kt+2; wait for kt+1 ----------
#pragma unroll
      for (int i = 0; i < RA1; ++i) a[i].load(sA + (RA0 + i) * (16 * 128), lg, swz);
      if (more2) {
        issue_w(cur);
        VMCNT(4);
      } else {
        VMCNT(0);
      }
      PHASE_SYNC();
      __builtin_amdgcn_s_setprio(1);
      QUAD16(RA1, RA0, 2, b1)
      QUAD16(RA1, RA0, 0, b0)
      __builtin_amdgcn_s_setprio(0);
      PHASE_SYNC();
      continue;
    }
    // ---------- phase 0: W column tiles 0-1, A row tiles 0-3; quadrant (A0, W0) ----------
#pragma unroll
    for (int c = 0; c < 2; ++c) b0[c].load(sB + c * (16 * 128), lg, swz);
#pragma unroll
    for (int i = 0; i < RA0; ++i) a[i].load(sA + i * (16 * 128), lg, swz);
    if (more1) issue_a(0, cur ^ 1);
    PHASE_SYNC();
    __builtin_amdgcn_s_setprio(1);
    QUAD16(RA0, 0, 0, b0)
    __builtin_amdgcn_s_setprio(0);
    PHASE_SYNC();

    // ---------- phase 1: W column tiles 2-3; quadrant (A0, W1) ----------
#pragma unroll
    for (int c = 0; c < 2; ++c) b1[c].load(sB + (2 + c) * (16 * 128), lg, swz);
    if (more1) { issue_a(1, cur ^ 1); next_a(); }
    PHASE_SYNC();
    __builtin_amdgcn_s_setprio(1);
    QUAD16(RA0, 0, 2, b1)
    __builtin_amdgcn_s_setprio(0);
    PHASE_SYNC();

    // ---------- phase 2: A row tiles 4 .. RT-1; quadrant (A1, W1) ----------
#pragma unroll
    for (int i = 0; i < RA1; ++i) a[i].load(sA + (RA0 + i) * (16 * 128), lg, swz);
    PHASE_SYNC();
    __builtin_amdgcn_s_setprio(1);
    QUAD16(RA1, RA0, 2, b1)
    __builtin_amdgcn_s_setprio(0);
    PHASE_SYNC();

    // ---------- phase 3: nothing to read (W tiles 0-1 are still in registers); quadrant (A1, W0); W of kt+2; wait for kt+1 ----------
    if (more2) {
      issue_w(cur);
      VMCNT(4);
    } else {
      VMCNT(0);
    }
    PHASE_SYNC();
    __builtin_amdgcn_s_setprio(1);
    QUAD16(RA1, RA0, 0, b0)
    __builtin_amdgcn_s_setprio(0);
    PHASE_SYNC();
  }
#undef QUAD16
  if (wr == 0) __builtin_amdgcn_s_barrier();   // re-join the two wave rows
  if (p.debug_flags & 2) return;               // (profiling: main loop only, nothing is stored)

  // ---- epilogue: two 32-column slabs per wave (8 waves x 128 x 32 fp32 = 128 KB = the operand buffers) ----
  float* sC = (float*)smem + wave * (128 * 32);
  // wave row 1 starts ROW1 rows into the tile; when the two wave rows overlap (RT = 7) it leaves the shared rows to wave row 0
  constexpr int SKIP1 = 16 * RT - ROW1 > 0 ? 16 * RT - ROW1 : 0;
  const int skip = wr == 1 ? SKIP1 : 0;
  const int m_base = m0 + wr * ROW1 + skip;
  const int m_end = min(p.M, m0 + (wr == 0 ? 16 * RT : BM));
#pragma unroll
  for (int js = 0; js < 2; ++js) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) sC[(i * 16 + 4 * lg + r) * 32 + c * 16 + l16] = acc[i][js * 2 + c][r];
    __syncthreads();
    if (SK) {
      pd_igemm_args q = p;         // raw partial sums -> this slice's slab
      q.bias = nullptr; q.rowvec = nullptr; q.mul = nullptr; q.residual = nullptr; q.out_bf16 = nullptr; q.out_bf16_lo = nullptr;
      q.out_f32 = p.splitk_ws + (int64_t)kslice * p.M * p.N;
      q.ld_out = p.N; q.alpha = 1.f; q.act = 0; q.out_batch_stride = 0;
      q.vec_epilogue = (p.N & 3) == 0 ? 1 : 0;
      igemm_epilogue<128, 32>(q, sC + skip * 32, lane, m_base, m_end, n0 + wc * 64 + js * 32, 0);
    } else {
      igemm_epilogue<128, 32>(p, sC + skip * 32, lane, m_base, m_end, n0 + wc * 64 + js * 32, bz);
    }
  }
#endif
}


// out = epilogue(sum over K-slices of the slabs), four columns per thread; same arithmetic as igemm_epilogue_rows
__global__ void __launch_bounds__(256) igemm_splitk_reduce_kernel(const pd_igemm_args p) {
  const int n4 = p.N >> 2;
  const int64_t total = (int64_t)p.M * n4;
  const int64_t slab = (int64_t)p.M * p.N;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int m = (int)(i / n4), n = (int)(i - (int64_t)m * n4) * 4;
    const float* src = p.splitk_ws + (int64_t)m * p.N + n;
    float4 acc = *(const float4*)src;
    for (int k = 1; k < p.ksplit; ++k) {
      const float4 t = *(const float4*)(src + k * slab);
      acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
    }
    float v[4] = {acc.x, acc.y, acc.z, acc.w};
    if (p.vec_epilogue && !p.mul && p.act == 0 && ((uintptr_t)p.bias & 15) == 0) {
      // the Conv3d launches of the small-batch mode: every epilogue operand as ONE 16-B load issued beside the slab loads (the general path
      // below reads them as twelve dwords behind the sums); the same operations in the same order: bit-identical
      float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f), r4 = b4, s4 = b4;
      if (p.bias) b4 = *(const float4*)(p.bias + n);
      if (p.rowvec) r4 = *(const float4*)(p.rowvec + (int64_t)(m / p.rows_per_sample) * p.ld_rowvec + n);
      if (p.residual) s4 = *(const float4*)(p.residual + (int64_t)(p.res_period ? m % p.res_period : m) * p.ld_res + n);
      const float bb[4] = {b4.x, b4.y, b4.z, b4.w}, rr[4] = {r4.x, r4.y, r4.z, r4.w}, ss[4] = {s4.x, s4.y, s4.z, s4.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] = v[e] * p.alpha + bb[e];
        if (p.rowvec) v[e] += rr[e];
        if (p.residual) v[e] += ss[e];
      }
      if (p.out_f32) *(float4*)(p.out_f32 + (int64_t)m * p.ld_out + n) = make_float4(v[0], v[1], v[2], v[3]);
      if (p.out_bf16) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          uint16_t h, l;
          f2bf_split(v[e], h, l);
          p.out_bf16[(int64_t)m * p.ld_outb + n + e] = p.out_bf16_lo ? h : (uint16_t)f2op(v[e]);
          if (p.out_bf16_lo) p.out_bf16_lo[(int64_t)m * p.ld_outb + n + e] = l;
        }
      }
      continue;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = v[e] * p.alpha + (p.bias ? p.bias[n + e] : 0.f);
    if (p.rowvec) {
      const float* rv = p.rowvec + (int64_t)(m / p.rows_per_sample) * p.ld_rowvec + n;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] += rv[e];
    }
    if (p.act != 0) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = (p.out_bf16 && !p.out_bf16_lo && !p.out_f32) ? act_apply16(v[e], p.act) : act_apply(v[e], p.act);
    }
    if (p.mul) {
      const float* mu = p.mul + (int64_t)m * p.ld_mul + n;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] *= mu[e];
    }
    if (p.residual) {
      const float* rs = p.residual + (int64_t)(p.res_period ? m % p.res_period : m) * p.ld_res + n;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] += rs[e];
    }
    if (p.out_f32) {
      float* o = p.out_f32 + (int64_t)m * p.ld_out + n;
      if (p.vec_epilogue) *(float4*)o = make_float4(v[0], v[1], v[2], v[3]);
      else { o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3]; }
    }
    if (p.out_bf16) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        uint16_t h, l;
        f2bf_split(v[e], h, l);
        p.out_bf16[(int64_t)m * p.ld_outb + n + e] = p.out_bf16_lo ? h : (uint16_t)f2op(v[e]);
        if (p.out_bf16_lo) p.out_bf16_lo[(int64_t)m * p.ld_outb + n + e] = l;
      }
    }
  }
}

template <int KIND, bool F8 = false, bool WF = false>
static int launch256_splitk(const pd_igemm_args& a, hipStream_t s) {
  constexpr int lds = 2 * KBUF;
  static bool attr_set_dev[PD_MAX_DEVICES];
  bool& attr_set = attr_set_dev[pd_cur_device()];
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)igemm256_kernel<KIND, 8, true, F8, false, true, WF>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) {
      pd_set_error("pd_igemm: hipFuncSetAttribute(%d) failed: %s", lds, hipGetErrorString(e));
      return PD_ERR_LAUNCH;
    }
    attr_set = true;
  }
  const int tiles = ((a.M + 255) / 256) * ((a.N + 255) / 256);
  hipLaunchKernelGGL((igemm256_kernel<KIND, 8, true, F8, false, true, WF>), dim3(tiles, a.ksplit, 1), dim3(512), lds, s, a);
  PD_CHECK_LAUNCH();
  const int64_t total = (int64_t)a.M * (a.N >> 2);
  const unsigned blocks = (unsigned)std::min<int64_t>((total + 255) / 256, 4096);
  hipLaunchKernelGGL(igemm_splitk_reduce_kernel, dim3(blocks), dim3(256), 0, s, a);
  PD_CHECK_LAUNCH();
  return PD_OK;
}

bool pd_igemm256_supported(const pd_igemm_args& a, int kind);

// K-slices for a launch of `tiles` 256 x 256 tiles and nk K-tiles on the device's CUs (256 on the MI355X; 0 = do not split): split when the
// tiles cover at most half of the CUs, into as many slices as fit one round, each at least 8 K-tiles long, within the workspace.
int pd_igemm256_ksplit(const pd_igemm_args& a, int kind) {
  const int max_tiles = a.splitk_max_tiles > 0 ? a.splitk_max_tiles : a.splitk_max_tiles < 0 ? 0 : 128;   // (A/B switch of the caller)
  if (!a.splitk_ws || a.split || (a.N & 3) || (a.nbatch > 1) || !pd_igemm256_supported(a, kind)) return 0;
  const int64_t tiles = (int64_t)((a.M + 255) / 256) * ((a.N + 255) / 256);
  const int nk = a.taps * (a.Cin >> (a.fp8 ? 7 : 6));       // K-tiles of 128 B per row
  if (tiles > max_tiles || nk < 32) return 0;
  int64_t ks = std::min<int64_t>(pd_num_cus() / tiles, nk / 8);
  ks = std::min<int64_t>(ks, a.splitk_ws_elems / ((int64_t)a.M * a.N));
  return ks >= 2 ? (int)ks : 0;
}

int pd_igemm256_launch_splitk(const pd_igemm_args& a, int kind, hipStream_t s) {
#if !PD_IS_F16
  if (a.fp8) return kind == 0 ? launch256_splitk<0, true>(a, s) : launch256_splitk<2, true>(a, s);
#endif
  if (a.w_fold > 0) return kind == 0 ? launch256_splitk<0, false, true>(a, s) : launch256_splitk<2, false, true>(a, s);
  return kind == 0 ? launch256_splitk<0>(a, s) : launch256_splitk<2>(a, s);
}

template <int KIND, int RT, bool F8 = false, bool SP = false, bool P2 = true, bool WF = false>
static int launch256(const pd_igemm_args& a, hipStream_t s) {
  constexpr int lds = 2 * KBUF;
  constexpr int BM = RT == 8 ? 256 : 16 * RT + 96;
  static bool attr_set_dev[PD_MAX_DEVICES];
  bool& attr_set = attr_set_dev[pd_cur_device()];
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)igemm256_kernel<KIND, RT, false, F8, SP, P2, WF>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) {
      pd_set_error("pd_igemm: hipFuncSetAttribute(%d) failed: %s", lds, hipGetErrorString(e));
      return PD_ERR_LAUNCH;
    }
    attr_set = true;
  }
  const int tiles = ((a.M + BM - 1) / BM) * ((a.N + 255) / 256);
  dim3 grid(tiles, 1, a.nbatch > 0 ? a.nbatch : 1);
  hipLaunchKernelGGL((igemm256_kernel<KIND, RT, false, F8, SP, P2, WF>), grid, dim3(512), lds, s, a);
  PD_CHECK_LAUNCH();
  return PD_OK;
}

// true when the 256-tile kernel can run this (already validated) launch
bool pd_igemm256_supported(const pd_igemm_args& a, int kind) {
  if (a.split) {
    // the hi/lo form: both halves of an operand behind ONE buffer descriptor (bf16 build only; the engine allocates them in one tensor)
#if PD_IS_F16
    return false;
#else
    if (a.fp8 || a.nbatch > 1 || !a.A_lo || !a.W_lo) return false;
    const int64_t da = (const char*)a.A_lo - (const char*)a.A, dw = (const char*)a.W_lo - (const char*)a.W;
    if ((da < 0 ? -da : da) + (int64_t)a.a_bytes >= 0xfffffe00ll || (dw < 0 ? -dw : dw) + (int64_t)a.w_bytes >= 0xfffffe00ll) return false;
    if ((da | dw) & 15) return false;
#endif
  }
  if (kind == 0) return true;
  return a.st == 1 && a.sh == 1 && a.sw == 1 && a.To < 1024 && a.Ho < 1024 && a.Wo < 1024;      // (nearest x2 up-sampling: in the gather)
}

int pd_igemm256_launch(const pd_igemm_args& a, int kind, hipStream_t s) {
#if !PD_IS_F16
  if (a.split) return kind == 0 ? launch256<0, 8, false, true>(a, s) : launch256<2, 8, false, true>(a, s);
  if (a.fp8 && (a.debug_flags & 64)) return kind == 0 ? launch256<0, 8, true, false, false>(a, s) : launch256<2, 8, true, false, false>(a, s);
  if (a.fp8) return kind == 0 ? launch256<0, 8, true>(a, s) : launch256<2, 8, true>(a, s);
#endif
  if (a.w_fold > 0) return kind == 0 ? launch256<0, 8, false, false, true, true>(a, s) : launch256<2, 8, false, false, true, true>(a, s);
  if (a.debug_flags & 64) return kind == 0 ? launch256<0, 8, false, false, false>(a, s) : launch256<2, 8, false, false, false>(a, s);   // (A/B: four phases)
  return kind == 0 ? launch256<0, 8>(a, s) : launch256<2, 8>(a, s);
}

}  // namespace PD_NS
