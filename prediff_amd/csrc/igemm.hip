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
//   * v_mfma_f32_16x16x32_bf16, each wave owns (BM/2) x (BN/2) of the tile.
//   * epilogue in registers: alpha, bias, per-sample row vector (timestep embedding), activation, gate multiply,
//     fp32 residual add, fp32 and/or bf16 (hi[/lo]) stores.
//   * blockIdx -> tile mapping gives each XCD a contiguous range of tiles (A rows are then re-used out of that
//     XCD's L2 across the N tiles and across the overlapping filter taps).
//   * SPLIT: x = hi + lo bf16 decomposition of both operands, 3 MFMAs per fragment pair (drops lo*lo): fp32-class
//     accuracy (~2^-16 relative per product) at 1/3 of the bf16 MFMA rate instead of 1/16 for the f32 MFMA.
#include <type_traits>
#include "common.h"
#include "igemm_epilogue.h"

namespace PD_NS {

__device__ __attribute__((aligned(64))) uint32_t g_pd_zero_page[32];   // 128 B of zeros (never written)

#ifndef PD_BIG_TILE_DEFAULT
#define PD_BIG_TILE_DEFAULT 1
#endif

// 16 B/lane DMA HBM/L2 -> LDS through a buffer descriptor: per-lane byte offset in a VGPR, wave-uniform byte offset in an
// SGPR, hardware bounds check (offset >= num_records reads as zero: out-of-image taps and M/N tails cost no address math).
#define BLDS16(rsrc, ldsptr, voff, soff) \
  __builtin_amdgcn_raw_ptr_buffer_load_lds((rsrc), (__attribute__((address_space(3))) void*)(ldsptr), 16, (voff), (soff), 0, 0)
#define PD_OOB 0xffffff00u

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// KIND: 0 linear (no spatial decode), 1 conv2d, 2 conv3d -- also tags the instantiation so profilers report the uses separately.
// BK in {32, 64}; NS = LDS ring depth (NS-1 K-steps of operand traffic in flight while one is being consumed).
template <int BM, int BN, int BK, int NS, bool SPLIT, int KIND, int OCC = 2, int ESLAB = 1>
__global__ void __launch_bounds__(256, OCC) igemm_kernel(const pd_igemm_args p) {
#if defined(__HIP_DEVICE_COMPILE__)   // the host pass only needs the launch stub (buffer-resource types are device-only)
  constexpr int ROWB = BK * 2;                    // bytes per tile row
  constexpr int CPR = ROWB / 16;                  // 16 B chunks per row (4 or 8)
  constexpr int RPI = 256 / CPR;                  // tile rows covered by one 256-thread DMA instruction (64 or 32)
  constexpr int A_TILE = BM * ROWB;
  constexpr int B_TILE = BN * ROWB;
  constexpr int NP = SPLIT ? 2 : 1;
  constexpr int STAGE = (A_TILE + B_TILE) * NP;
  constexpr int AI = BM / RPI, BI = BN / RPI;     // DMA instructions per thread per tile
  constexpr int LPS = (AI + BI) * NP;             // DMA instructions per thread per stage
  constexpr int KSUB = BK / 32;                   // MFMA k-steps (32 deep) per stage
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

  // buffer descriptors (wave-uniform: kernel arguments + blockIdx only)
  const auto rA = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + (int64_t)bz * p.a_batch_stride), 0, p.a_bytes, 0x00020000);
  const auto rW = __builtin_amdgcn_make_buffer_rsrc((void*)(p.W + (int64_t)bz * p.w_batch_stride), 0, p.w_bytes, 0x00020000);
  const auto rAlo = __builtin_amdgcn_make_buffer_rsrc((void*)((SPLIT ? p.A_lo : p.A) + (int64_t)bz * p.a_batch_stride), 0, p.a_bytes, 0x00020000);
  const auto rWlo = __builtin_amdgcn_make_buffer_rsrc((void*)((SPLIT ? p.W_lo : p.W) + (int64_t)bz * p.w_batch_stride), 0, p.w_bytes, 0x00020000);

  // ---- staging descriptors (fixed per thread) ----
  // lane -> (row inside the RPI-row slab, 16 B position); the k-chunk it FETCHES is position ^ swizzle(row) so that the
  // lane-linear DMA image is the bank-conflict-free one the ds_read_b128 below expects.
  const int srow = tid / CPR;
  const int spos = tid % CPR;
  const int schunk = (CPR == 8) ? (spos ^ ((srow >> 1) & 7)) : (spos ^ ((srow >> 2) & 3));
  int vt0[AI], vh0[AI], vw0[AI];
  uint32_t abase[AI];
  uint32_t aoff[AI];   // BYTE offset of (row, current tap, chunk) or PD_OOB
#pragma unroll
  for (int i = 0; i < AI; ++i) {
    const int m = m0 + i * RPI + srow;
    if (KIND == 0) {
      aoff[i] = m < p.M ? ((uint32_t)m * (uint32_t)p.lda + schunk * 8) * 2u : PD_OOB;
    } else {
      const int hw_o = p.Ho * p.Wo, thw_o = p.To * hw_o;
      const bool ok = m < p.M;
      const int mm = ok ? m : 0;
      const int b = mm / thw_o, r1 = mm - b * thw_o;
      const int ot = r1 / hw_o, r2 = r1 - ot * hw_o;
      const int oh = r2 / p.Wo, ow = r2 - oh * p.Wo;
      vt0[i] = ok ? ot * p.st - p.pt : -(1 << 20);      // invalid rows fail every bounds check
      vh0[i] = oh * p.sh - p.ph;
      vw0[i] = ow * p.sw - p.pw;
      abase[i] = (uint32_t)b * (uint32_t)(p.Ti * p.Hi * p.Wi);
    }
  }
  uint32_t woff[BI];
#pragma unroll
  for (int i = 0; i < BI; ++i) {
    const int n = n0 + i * RPI + srow;
    woff[i] = n < p.N ? ((uint32_t)n * (uint32_t)p.ldw + schunk * 8) * 2u : PD_OOB;
  }
  const int kchunks = p.Cin / BK;
  const int nk = (p.debug_flags & 1) ? 0 : p.taps * kchunks;
  const int vT = p.vT > 0 ? p.vT : p.Ti * p.ut, vH = p.vH > 0 ? p.vH : p.Hi * p.uh, vW = p.vW > 0 ? p.vW : p.Wi * p.uw;
  const int khw = p.KH * p.KW;

  auto set_tap = [&](int tap) {
    if (p.w_fold > 0 && tap >= p.w_fold) tap -= p.w_fold;     // the W_lo slabs walk the same activation gather as the W_hi slabs
    const int kt = tap / khw, r = tap - kt * khw;
    const int kh = r / p.KW, kw = r - kh * p.KW;
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const int vt = vt0[i] + kt, vh = vh0[i] + kh, vw = vw0[i] + kw;
      const bool ok = (unsigned)vt < (unsigned)vT && (unsigned)vh < (unsigned)vH && (unsigned)vw < (unsigned)vW;
      const int it = p.ut == 2 ? vt >> 1 : vt, ih = p.uh == 2 ? vh >> 1 : vh, iw = p.uw == 2 ? vw >> 1 : vw;
      aoff[i] = ok ? ((abase[i] + (uint32_t)((it * p.Hi + ih) * p.Wi + iw)) * (uint32_t)p.lda + schunk * 8) * 2u : PD_OOB;
    }
  };

  auto issue = [&](int stage, int ks) {
    const int tap = ks / kchunks, kc = ks - tap * kchunks;
    if (KIND != 0 && kc == 0) set_tap(tap);
    char* sbase = smem + stage * STAGE;
    const int ka = kc * (BK * 2);                                             // wave-uniform byte offsets
    const int kw = (int)(((int64_t)tap * p.w_tap_stride + kc * BK) * 2);
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      char* dst = sbase + (i * 256 + wave * 64) * 16;
      BLDS16(rA, dst, aoff[i], ka);
      if (SPLIT) BLDS16(rAlo, dst + A_TILE, aoff[i], ka);
    }
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      char* dst = sbase + A_TILE * NP + (i * 256 + wave * 64) * 16;
      BLDS16(rW, dst, woff[i], kw);
      if (SPLIT) BLDS16(rWlo, dst + B_TILE, woff[i], kw);
    }
  };

  // ---- accumulators: 16 x 16 MFMA tiles (v_mfma_f32_16x16x32_bf16: half-length MFMAs interleave better with the fragment reads
  //      than 32x32x16 and leave no dependent issue pairs -- +12 % on the 256 x 256 kernel, same LDS traffic) ----
  constexpr int TM16 = BM / 32, TN16 = BN / 32;   // 16-row / 16-column tiles per wave (wave tile BM/2 x BN/2)
  f32x4 acc[TM16][TN16];
#pragma unroll
  for (int i = 0; i < TM16; ++i)
#pragma unroll
    for (int j = 0; j < TN16; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;

  const int wr = wave >> 1, wc = wave & 1;
  const int l16 = lane & 15, lg = lane >> 4;
  const int swz = (CPR == 8) ? ((l16 >> 1) & 7) : ((l16 >> 2) & 3);
  const int a_row_b = (wr * (BM / 2) + l16) * ROWB;   // byte offset of this lane's row in the A tile
  const int b_row_b = (wc * (BN / 2) + l16) * ROWB;

  // ---- software pipeline: stages ks+1 .. ks+NS-1 are in flight while stage ks is consumed ----
  constexpr int D = NS - 1;
#pragma unroll
  for (int s0 = 0; s0 < D; ++s0)
    if (s0 < nk) issue(s0, s0);
  int stage = 0;
  for (int ks = 0; ks < nk; ++ks) {
    // this wave's share of stage ks has landed once at most min(D-1, nk-ks-1) younger stages remain outstanding
    const int younger = min(D - 1, nk - ks - 1);
    if (D >= 3 && younger == 2) wait_vmcnt<2 * LPS>();
    else if (D >= 2 && younger == 1) wait_vmcnt<LPS>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();   // everyone's share landed; everyone finished reading the stage refilled below
    if (ks + D < nk) issue((stage + D) % NS, ks + D);
    const char* sA = smem + stage * STAGE;
    const char* sB = sA + A_TILE * NP;
#pragma unroll
    for (int kk = 0; kk < KSUB; ++kk) {                 // 32-deep k-steps
      const int pos = ((kk * 4 + lg) ^ swz) * 16;
      op8 a[TM16], b[TN16], al[TM16], bl[TN16];
#pragma unroll
      for (int i = 0; i < TM16; ++i) {
        a[i] = *(const op8*)(sA + a_row_b + i * 16 * ROWB + pos);
        if (SPLIT) al[i] = *(const op8*)(sA + A_TILE + a_row_b + i * 16 * ROWB + pos);
      }
#pragma unroll
      for (int j = 0; j < TN16; ++j) {
        b[j] = *(const op8*)(sB + b_row_b + j * 16 * ROWB + pos);
        if (SPLIT) bl[j] = *(const op8*)(sB + B_TILE + b_row_b + j * 16 * ROWB + pos);
      }
#pragma unroll
      for (int i = 0; i < TM16; ++i)
#pragma unroll
        for (int j = 0; j < TN16; ++j) {
          if (SPLIT) {
            acc[i][j] = mfma_16x16x32(al[i], b[j], acc[i][j]);
            acc[i][j] = mfma_16x16x32(a[i], bl[j], acc[i][j]);
          }
          acc[i][j] = mfma_16x16x32(a[i], b[j], acc[i][j]);
        }
    }
    stage = stage + 1 == NS ? 0 : stage + 1;
  }

  // ---- epilogue: accumulators -> per-wave LDS slab (row major, ESLAB column slabs in turn) -> coalesced 16 B row segments ----
  constexpr int WM = BM / 2, WN = BN / 2;
  constexpr int WNS = WN / ESLAB;                  // columns staged at a time
  constexpr int TNS = TN16 / ESLAB;                // 16-column tiles staged at a time
  float* sC = (float*)smem + wave * (WM * WNS);
#pragma unroll
  for (int js = 0; js < ESLAB; ++js) {
    __syncthreads();                               // operand stages (js == 0) / previous slab (js > 0) are no longer read
#pragma unroll
    for (int i = 0; i < TM16; ++i)
#pragma unroll
      for (int j = 0; j < TNS; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          sC[(i * 16 + 4 * lg + r) * WNS + j * 16 + l16] = acc[i][js * TNS + j][r];
    __syncthreads();
    igemm_epilogue<WM, WNS>(p, sC, lane, m0 + wr * WM, p.M, n0 + wc * WN + js * WNS, bz);
  }
#endif
}

template <int BM, int BN, int BK, int NS, bool SPLIT, int KIND, int OCC = 2, int ESLAB = 1>
static int launch_igemm(const pd_igemm_args& a, hipStream_t s) {
  constexpr int stage = (BM + BN) * BK * 2 * (SPLIT ? 2 : 1);
  constexpr int epi = 4 * (BM / 2) * (BN / 2) * 4 / ESLAB;
  constexpr int lds = NS * stage > epi ? NS * stage : epi;   // the accumulator slab of the epilogue re-uses the operand ring
  static bool attr_set_dev[PD_MAX_DEVICES];
  bool& attr_set = attr_set_dev[pd_cur_device()];
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)igemm_kernel<BM, BN, BK, NS, SPLIT, KIND, OCC, ESLAB>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) {
      pd_set_error("pd_igemm: hipFuncSetAttribute(%d) failed: %s", lds, hipGetErrorString(e));
      return PD_ERR_LAUNCH;
    }
    attr_set = true;
  }
  const int tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
  dim3 grid(tiles, 1, a.nbatch > 0 ? a.nbatch : 1);
  hipLaunchKernelGGL((igemm_kernel<BM, BN, BK, NS, SPLIT, KIND, OCC, ESLAB>), grid, dim3(256), lds, s, a);
  PD_CHECK_LAUNCH();
  return PD_OK;
}

// Tile / pipeline configurations (a.tile): 1 = 128x128, BK 64, 2-stage (2 workgroups/CU);  2 = 64x64, BK 64, 2-stage;  8 / 9 = 128x64 / 64x128;
// 3 = 128x128, BK 32, 4-stage ring (2 workgroups/CU, 3 K-steps in flight);  4 = 128x128, BK 64, 3-stage (1 workgroup/CU).
template <bool SPLIT, int KIND>
static int dispatch_igemm(const pd_igemm_args& a, int tile, hipStream_t s) {
  switch (tile) {
    case 1: return launch_igemm<128, 128, 64, 2, SPLIT, KIND>(a, s);
    case 2: return launch_igemm<64, 64, 64, 2, SPLIT, KIND>(a, s);
    case 3: return launch_igemm<128, 128, 32, 4, SPLIT, KIND>(a, s);
    case 4: return launch_igemm<128, 128, 64, 3, SPLIT, KIND>(a, s);
    case 5: return launch_igemm<128, 128, 32, 2, SPLIT, KIND, (SPLIT ? 2 : 4), 2>(a, s);   // 32 KB LDS, <=128 VGPR: 4 workgroups/CU
    case 6: return launch_igemm<128, 128, 32, 3, SPLIT, KIND, (SPLIT ? 2 : 3), 2>(a, s);   // 48 KB LDS: 3 workgroups/CU
    case 8: return launch_igemm<128, 64, 64, 2, SPLIT, KIND>(a, s);    // half-width tile for small grids (24 KB LDS per stage pair)
    case 9: return launch_igemm<64, 128, 64, 2, SPLIT, KIND>(a, s);
    default: pd_set_error("pd_igemm: unknown tile config %d", tile); return PD_ERR_ARG;
  }
}

// igemm256.hip: 256 x 256 x 64 eight-wave ping-pong variant for long-K launches
bool pd_igemm256_supported(const pd_igemm_args& a, int kind);
int pd_igemm256_launch(const pd_igemm_args& a, int kind, hipStream_t s);

int pd_igemm256_ksplit(const pd_igemm_args& a, int kind);
int pd_igemm256_launch_splitk(const pd_igemm_args& a, int kind, hipStream_t s);

#if !PD_IS_F16
extern "C" int pd_f16_igemm(const pd_igemm_args*, pd_stream_t);
#endif

extern "C" int PD_ENTRY(igemm)(const pd_igemm_args* pa, pd_stream_t stream) {
  PD_CHECK_ARG(pa != nullptr, "pd_igemm: null args");
  PD_FORWARD_F16(pa->operand == PD_OPERAND_F16, pd_f16_igemm(pa, stream));
  pd_igemm_args a = *pa;
  PD_CHECK_ARG(!PD_IS_F16 || (!a.split && !a.fp8 && !a.out_bf16_lo), "pd_igemm: IEEE-half operands: no hi/lo split, no e4m3 operands");
  // A/B switches of the caller (0 = the defaults)
  // hi/lo launches carry 1.5x the MFMA work per operand byte: the 256 x 256 form wins from K = 256 on (v1, 64 trajectories, precision="fp32":
  // 397 / 431 / 478 steps/s with 1024 / 512 / 256 -- profiles/r05_e_*)
  const int min_k_256 = a.min_k_256 > 0 ? a.min_k_256 : a.split ? 256 : 1024;   // shortest K (taps * Cin) the automatic choice gives to the 256 x 256 kernel (512 wins
                                                                // 25 % on stand-alone full-resolution level-1 launches and nothing end to end, two lanes running)
  PD_CHECK_ARG(a.A && a.W, "pd_igemm: A/W null");
  PD_CHECK_ARG(a.M > 0 && a.N > 0 && a.taps > 0, "pd_igemm: bad M/N/taps (%d,%d,%d)", a.M, a.N, a.taps);
  PD_CHECK_ARG(a.Cin > 0 && (a.Cin & 63) == 0, "pd_igemm: Cin=%d must be a positive multiple of 64 (zero padded)", a.Cin);
  PD_CHECK_ARG((a.lda & 7) == 0 && (a.ldw & 7) == 0, "pd_igemm: lda/ldw must be multiples of 8 (16 B rows)");
  PD_CHECK_ARG(a.w_fold >= 0 && (a.w_fold == 0 || (!a.split && !a.fp8 && a.taps == 2 * a.w_fold)),
               "pd_igemm: w_fold = %d needs taps = 2 w_fold (W_hi slabs then W_lo slabs), no hi/lo split, no e4m3 operands", a.w_fold);
  PD_CHECK_ARG((a.w_fold ? a.w_fold : a.taps) == a.KT * a.KH * a.KW, "pd_igemm: taps != KT*KH*KW");
  PD_CHECK_ARG((int64_t)a.B * a.To * a.Ho * a.Wo == a.M, "pd_igemm: M != B*To*Ho*Wo");
  PD_CHECK_ARG((a.ut == 1 || a.ut == 2) && (a.uh == 1 || a.uh == 2) && (a.uw == 1 || a.uw == 2), "pd_igemm: bad upsample");
  if (a.fp8) {
    PD_CHECK_ARG((a.Cin & 127) == 0 && (a.lda & 15) == 0 && (a.ldw & 15) == 0, "pd_igemm: fp8 operands need Cin %% 128 == 0 and lda/ldw %% 16 == 0");
    PD_CHECK_ARG(!a.split && a.nbatch <= 1, "pd_igemm: fp8 operands: no hi/lo split, no batch");
  }
  {
    const int eb = a.fp8 ? 1 : 2;
    const int64_t abytes = (int64_t)a.B * a.Ti * a.Hi * a.Wi * (int64_t)a.lda * eb;
    const int64_t wbytes = ((int64_t)(a.taps - 1) * a.w_tap_stride + (int64_t)a.N * a.ldw) * eb;
    PD_CHECK_ARG(abytes < 0xfffffe00ll && wbytes < 0xfffffe00ll, "pd_igemm: operand larger than a 4 GiB buffer descriptor");
    a.a_bytes = (uint32_t)abytes;
    a.w_bytes = (uint32_t)wbytes;
  }
  PD_CHECK_ARG(!a.split || (a.A_lo && a.W_lo), "pd_igemm: split needs A_lo and W_lo");
  PD_CHECK_ARG(!a.rowvec || a.rows_per_sample > 0, "pd_igemm: rowvec needs rows_per_sample");
  PD_CHECK_ARG(a.out_f32 || a.out_bf16, "pd_igemm: no output");
  hipStream_t s = (hipStream_t)stream;
  a.vec_epilogue = ((a.N & 3) == 0) && (!a.out_f32 || (a.ld_out & 3) == 0) && (!a.out_bf16 || (a.ld_outb & 3) == 0) &&
                   (!a.residual || (a.ld_res & 3) == 0) && (!a.rowvec || (a.ld_rowvec & 3) == 0) &&
                   (!a.mul || (a.ld_mul & 3) == 0) && (((uintptr_t)a.out_f32 | (uintptr_t)a.residual | (uintptr_t)a.rowvec |
                   (uintptr_t)a.mul) & 15) == 0 && (((uintptr_t)a.out_bf16 | (uintptr_t)a.out_bf16_lo) & 7) == 0 &&
                   ((a.out_batch_stride | a.outb_batch_stride | a.res_batch_stride) & 3) == 0;
  if (a.vec_epilogue && !a.out_f32 && a.out_bf16 && (a.N & 7) == 0 && (a.ld_outb & 7) == 0 && (a.outb_batch_stride & 7) == 0 &&
      (((uintptr_t)a.out_bf16 | (uintptr_t)a.out_bf16_lo) & 15) == 0)
    a.vec_epilogue = 2;   // 16 B bf16 stores
  if (a.out_fp8_log2 > 0) {
    PD_CHECK_ARG(a.vec_epilogue == 2 && !a.out_bf16_lo && !a.split && a.out_fp8_log2 <= 16,
                 "pd_igemm: an e4m3 output needs the 8-column vector epilogue (out_bf16 only, N %% 8 == 0, ld_outb %% 8 == 0, 16 B aligned), no split");
  }
  int tile = a.tile;
  // a 1-tap, stride-1, unpadded, un-upsampled "convolution" is a plain row-wise linear layer: row m reads A row m
  const bool pointwise = (a.taps == 1 || a.w_fold == 1) && a.st == 1 && a.sh == 1 && a.sw == 1 && a.pt == 0 && a.ph == 0 && a.pw == 0 && a.ut == 1 &&
                         a.uh == 1 && a.uw == 1 && a.Ti == a.To && a.Hi == a.Ho && a.Wi == a.Wo && a.vT <= 0 && a.vH <= 0 && a.vW <= 0;
  const int kind = pointwise ? 0 : ((a.KT == 1 && a.Ti == 1 && a.To == 1) ? 1 : 2);
  a.ksplit = 1;
#if !PD_IS_F16
  if (a.fp8) {
    if (!pd_igemm256_supported(a, kind) || kind == 1) {
      pd_set_error("pd_igemm: fp8 operands are built for row-wise linear layers and stride-1, un-upsampled Conv3d launches only");
      return PD_ERR_UNSUPPORTED;
    }
    // small grids: K-slices as extra workgroups -- never with an e4m3 OUTPUT: the split-K reduce kernel stores 2-byte bf16 rows, which
    // would overrun the caller's 1-byte e4m3 buffer (the un-split epilogue is the only e4m3 producer)
    const int ks = (a.tile == 0 && !a.disable_256 && a.out_fp8_log2 <= 0) ? pd_igemm256_ksplit(a, kind) : 0;
    if (ks >= 2) {
      a.ksplit = ks;
      return pd_igemm256_launch_splitk(a, kind, s);
    }
    return pd_igemm256_launch(a, kind, s);
  }
#endif
  if (tile == 0 && !a.disable_256) {
    // small grids (few trajectories per launch) with a long K loop: 256 x 256 tiles x K-slices fill the CUs (igemm256.hip)
    const int ks = pd_igemm256_ksplit(a, kind);
    if (ks >= 2) {
      a.ksplit = ks;
      return pd_igemm256_launch_splitk(a, kind, s);
    }
  }
  if (tile == 0) {
    const int64_t t128 = (int64_t)((a.M + 127) / 128) * ((a.N + 127) / 128) * (a.nbatch > 0 ? a.nbatch : 1);
    // measured on MI355X (scripts/bench_igemm.py): with <= 4 K-steps the 4-workgroups/CU variant (BK 32, 32 KB LDS) hides the
    // prologue/epilogue of its neighbours best; longer K prefers the BK 64 two-stage tile.
    tile = t128 >= 192 ? ((!a.split && (int64_t)a.taps * a.Cin <= 256) ? 5 : PD_BIG_TILE_DEFAULT) : 2;
    // long K: the 256 x 256 eight-wave kernel does a round of 256 tiles (one per CU of the MI355X) in ~1.65x the time the
    // 128 x 128 kernel needs for a round of 512 (two per CU) inside the sampling loop -- twice the work; take it when its whole
    // rounds are the cheaper ones (Conv3d at 32 trajectories: 317 us in 2 rounds against 385 us in 4)
    if (tile == PD_BIG_TILE_DEFAULT && !a.disable_256 && (int64_t)a.taps * a.Cin >= min_k_256 && pd_igemm256_supported(a, kind)) {
      const int64_t t256 = (int64_t)((a.M + 255) / 256) * ((a.N + 255) / 256) * (a.nbatch > 0 ? a.nbatch : 1);
      const int64_t ncu = pd_num_cus();
      // (the hi/lo form of the 128 x 128 kernel holds 128 KB of LDS: ONE workgroup per CU, a round is 256 tiles)
      const int64_t r128 = a.split ? (t128 + ncu - 1) / ncu : (t128 + 2 * ncu - 1) / (2 * ncu), r256 = (t256 + ncu - 1) / ncu;
      if (a.split ? r256 * 11 <= r128 * 5 : r256 * 33 <= r128 * 20) tile = 7;
    }
  }
  if (a.split && tile == 4) tile = 1;   // 3 x 64 KB stages do not fit
  if (tile == 7) {
    if (pd_igemm256_supported(a, kind)) return pd_igemm256_launch(a, kind, s);
    tile = PD_BIG_TILE_DEFAULT;
  }
#if !PD_IS_F16
  if (a.split) {
    if (kind == 0) return dispatch_igemm<true, 0>(a, tile, s);
    if (kind == 1) return dispatch_igemm<true, 1>(a, tile, s);
    return dispatch_igemm<true, 2>(a, tile, s);
  }
#endif
  if (kind == 0) return dispatch_igemm<false, 0>(a, tile, s);
  if (kind == 1) return dispatch_igemm<false, 1>(a, tile, s);
  return dispatch_igemm<false, 2>(a, tile, s);
}

}  // namespace PD_NS
