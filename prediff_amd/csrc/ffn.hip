// pd_ffn_fused: out = x + W2 * act(W1 * LayerNorm(x) + b1) + b2 in ONE kernel (PositionwiseFFN.forward, pre-norm,
// reference cuboid_transformer.py:182-208).
//
// Why: at the SEVIR-LR level-0 shapes (C = 256, hidden 1024, K only 256) the un-fused chain LN -> GEMM -> GEMM is HBM bound: the
// bf16 hidden tensor (tokens x 1024) is written and read back (2 x 218 MB at 32 trajectories), plus the LN output.  Here the
// hidden activations never leave the CU: algorithmic HBM traffic drops to one fp32 read + one fp32 write of x, and the kernel
// is MFMA bound again.
//
// One workgroup = 512 threads (8 waves) owns BM = 128 token rows:
//   phase 0   LayerNorm of the 128 rows (one wave per 16 rows, fp32, two-pass) -> bf16 A tile in LDS, K-slab swizzled exactly
//             like the igemm operand tiles (conflict-free ds_read_b128).
//   chunk j   (64 hidden units at a time, Hd/64 chunks), per GROUP of 4 waves = 64 of the 128 rows:
//       MFMA slot  GEMM-2 of chunk j-1: acc[64 x C] += H_{j-1}[64 x 64] * W2_{j-1}[C x 64]^T   (wave tile 32 x C/2, K = 64)
//                  GEMM-1 of chunk j:   H_j^T[64 x 64] = W1_j[64 x C] * A^T                     (one 32x32 tile per wave, K = C)
//       VALU slot  +b1, activation, bf16: the transposed product leaves 4 consecutive hidden units per lane -> 8 B LDS writes
//                  straight into the A-operand layout GEMM-2 reads.
//     The two groups run one slot apart (one workgroup barrier per slot): each SIMD holds one wave of each group, so its MFMA
//     pipe works on one group's GEMMs while its VALU evaluates the other group's activation (erf GELU costs more issue cycles
//     than the chunk's 32 MFMAs; serialised it was the largest term of the kernel).
//     W1_j and W2_j stream through two buffer sets by buffer-descriptor DMA (16 B/lane), issued two slots before first use.
//   epilogue  acc + b2 + x -> out (fp32), staged through LDS for 16 B row segments.
// LDS: A 128*C*2 + H 16 KB + W1 64*C*2 + W2 C*128  = 144 KB at C = 256 (one workgroup per CU, two waves per SIMD).
#include "common.h"
#include "ln_tile.h"

namespace PD_NS {

#define BLDS16(rsrc, ldsptr, voff, soff) \
  __builtin_amdgcn_raw_ptr_buffer_load_lds((rsrc), (__attribute__((address_space(3))) void*)(ldsptr), 16, (voff), (soff), 0, 0)

struct pd_ffn_args_k {
  const float* x;
  float* out;
  const float* gamma;
  const float* beta;
  const pd_bf16* W1;     // [Hd][C]
  const float* b1;
  const pd_bf16* W2;     // [C][Hd]
  const float* b2;
  int M, Hd, act;
  float eps;
  uint32_t w1_bytes, w2_bytes;
  unsigned long long* trace;   // profiling only: per-slot clock stamps of waves 0 and 4 of workgroup 300 (null in production)
  int dbg;   // profiling ablations: 1 no weight DMA after chunk 0, 2 no GEMM-1, 4 no activation + H store, 8 no GEMM-2,
             // 16 no LN loads, 32 no residual loads, 64 no stores
};

template <int C, int ACT>
__global__ void __launch_bounds__(512, 2) ffn_fused_kernel(const pd_ffn_args_k p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int BM = 128, HC = 64;
  constexpr int KS = C / 64;                       // 64-wide K slabs of the A tile / W1 chunk
  constexpr int A_BYTES = BM * C * 2;
  constexpr int H_BYTES = BM * HC * 2;
  constexpr int W1_BYTES = HC * C * 2;
  constexpr int W2_BYTES = C * HC * 2;
  constexpr int TN2 = (C / 2) / 32;                // 32x32 tiles per wave in GEMM-2 (wave tile 32 x C/2)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sA = smem;
  char* sH = sA + A_BYTES;
  char* sW1 = sH + H_BYTES;
  char* sW2 = sW1 + W1_BYTES;
  float* sB1 = (float*)(sW2 + W2_BYTES);            // whole b1 (Hd floats): an ordinary global load inside the chunk loop would make
                                                    // hipcc drain the weight DMA queue (vmcnt(0)) every chunk

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m0 = blockIdx.x * BM;
  const int NJ = p.Hd / HC;

  const auto rW1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.W1, 0, p.w1_bytes, 0x00020000);
  const auto rW2 = __builtin_amdgcn_make_buffer_rsrc((void*)p.W2, 0, p.w2_bytes, 0x00020000);

  // DMA lane mapping: one 512-thread instruction fills one [64 rows][64 k] slab (8 KB), lane-linear, source-side swizzle
  const int drow = tid >> 3, dpos = tid & 7;
  const int dchunk = dpos ^ ((drow >> 1) & 7);
  const uint32_t w1_voff = ((uint32_t)drow * C + dchunk * 8) * 2u;                 // + (j*64*C + s*64)*2 in the SGPR offset
  uint32_t w2_voff[KS];
#pragma unroll
  for (int i = 0; i < KS; ++i) w2_voff[i] = ((uint32_t)(i * 64 + drow) * (uint32_t)p.Hd + dchunk * 8) * 2u;   // + j*64*2

  // weight buffers: set 0 = {sW1, sW2}; set 1 = the A-tile region (A_BYTES == W1_BYTES + W2_BYTES), free once every wave
  // holds its A fragments in registers
  auto w1buf = [&](int b) { return b ? sA : sW1; };
  auto w2buf = [&](int b) { return b ? sA + W1_BYTES : sW2; };
  auto issue_w1 = [&](int j, int b) {
    char* d1 = w1buf(b);
#pragma unroll
    for (int s = 0; s < KS; ++s) BLDS16(rW1, d1 + s * 8192 + wave * 1024, w1_voff, (j * HC * C + s * 64) * 2);
  };
  auto issue_w2 = [&](int j, int b) {
    char* d2 = w2buf(b);
#pragma unroll
    for (int i = 0; i < KS; ++i) BLDS16(rW2, d2 + i * 8192 + wave * 1024, w2_voff[i], j * HC * 2);
  };

  for (int i = tid; i < p.Hd; i += 512) sB1[i] = p.b1[i];
  issue_w1(0, 0);

  // ---- phase 0: LayerNorm -> bf16 A tile (KS slabs of [128][64], chunk swizzle (row>>1)&7) ----
  ln_block_to_tile<C>(p.x, p.gamma, p.beta, p.eps, sA, wave, lane, (p.dbg & 16) != 0,
                      [&](int r) { const int m = m0 + r; return m < p.M ? m : -1; });

  // ---- wave roles: two groups of 4 waves (one wave of each group per SIMD), each owns 64 of the 128 token rows ----
  const int grp = wave >> 2, wg = wave & 3;
  const int lrow = lane & 31, lhalf = lane >> 5;
  const int swz = (lrow >> 1) & 7;
  // GEMM-1 (transposed): wave -> hidden tile tn (of 2) x row tile tq (of the group's 2)
  const int tn = wg & 1, tq = wg >> 1;
  const int g1_a_row = (tn * 32 + lrow) * 128;                 // W1 chunk row (hidden unit) inside a slab
  const int g1_b_row = (grp * 64 + tq * 32 + lrow) * 128;      // A tile row (token) inside a slab
  // GEMM-2: wave -> row tile wm (of the group's 2) x column half wn (of 2)
  const int wm = wg >> 1, wn = wg & 1;
  char* const sHg = sH + grp * (64 * 128);                     // the group's own H tile [64 rows][64 hidden]
  const int g2_a_row = (wm * 32 + lrow) * 128;                 // H row
  const int g2_b_row = (wn * (C / 2) + lrow) * 128;            // W2 chunk row (output channel)

  const uint32_t h_lds = (uint32_t)(uintptr_t)sHg;             // LDS byte address of the H tile (low half of the flat address)
  const uint32_t b1_lds = (uint32_t)(uintptr_t)sB1;
  f32x16 acc2[TN2];
#pragma unroll
  for (int t = 0; t < TN2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[t][r] = 0.f;

  // ---- this wave's GEMM-1 B operand (its 32 token rows, all of K) lives in registers for the whole kernel ----
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();                                        // A tile written by all waves; W1 of chunk 0 landed
  op8 areg[KS * 4];
#pragma unroll
  for (int s = 0; s < KS; ++s)
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) areg[s * 4 + kk] = *(const op8*)(sA + s * (BM * 128) + g1_b_row + (((kk * 2 + lhalf) ^ swz) * 16));
  __syncthreads();                                        // A-tile region is now free: it becomes weight-buffer set 1

  // ---- slot loop.  A chunk j of a group is: MFMA slot [GEMM-2 of chunk j-1, GEMM-1 of chunk j] then VALU slot [bias,
  //      activation, bf16 -> H].  Group 1 runs ONE slot behind group 0, so on every SIMD one wave is in its MFMA slot while the
  //      other is in its VALU slot: the activation (17 VALU + 2 transcendental ops per hidden value: more issue cycles than the
  //      chunk's MFMAs) is hidden instead of serialised.  One workgroup barrier per slot.
  //      Weights: global slot 2j issues W1_{j+1} and W2_j (both waited for at the end of slot 2j+1);  W1_j is read in slots
  //      2j (group 0) and 2j+1 (group 1), W2_j in slots 2j+2 and 2j+3; buffer sets alternate with the chunk parity. ----
  f32x16 acc1, acc1b;                                      // two accumulators: no back-to-back dependent MFMA chain
  const int nslots = 2 * NJ + 2;
  for (int s = 0; s < nslots; ++s) {
    if (!(s & 1) && !(p.dbg & 1)) {
      const int j = s >> 1;
      if (j + 1 < NJ) issue_w1(j + 1, (j + 1) & 1);
      if (j < NJ) issue_w2(j, j & 1);
    }
    const int sl = s - grp;                                // this group's own slot index
    const int j = sl >> 1;
    if (sl >= 0 && !(sl & 1)) {
      // ---- MFMA slot: GEMM-2 of chunk j-1 (acc2 += H_{j-1} * W2_{j-1}^T), then GEMM-1 of chunk j (H_j^T = W1_j * A^T) ----
      // explicit two-deep fragment pipeline: left to itself hipcc re-uses one register set per MFMA pair, so every pair waits for
      // a full LDS round trip (GEMM-2 took ~1600 clocks for 512 clocks of MFMA work)
      const bool do2 = j >= 1 && !(p.dbg & 8), do1 = j < NJ;
      const char* cW2 = w2buf((j - 1) & 1);
      const char* cW1 = w1buf(j & 1);
      op8 fa[2], fb[2][TN2], fw[2][4];
      auto load2 = [&](int kk, int buf) {
        const int pos = ((kk * 2 + lhalf) ^ swz) * 16;
        fa[buf] = *(const op8*)(sHg + g2_a_row + pos);
#pragma unroll
        for (int t = 0; t < TN2; ++t) fb[buf][t] = *(const op8*)(cW2 + g2_b_row + t * 32 * 128 + pos);
      };
      auto load1 = [&](int ks, int buf) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) fw[buf][kk] = *(const op8*)(cW1 + ks * 8192 + g1_a_row + (((kk * 2 + lhalf) ^ swz) * 16));
      };
      if (do2) load2(0, 0);
      if (do2) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          if (kk < 3) load2(kk + 1, (kk + 1) & 1);
          else if (do1) load1(0, 0);
#pragma unroll
          for (int t = 0; t < TN2; ++t) acc2[t] = mfma_32x32x16(fa[kk & 1], fb[kk & 1][t], acc2[t]);
        }
      } else if (do1) {
        load1(0, 0);
      }
      if (do1) {
        // GEMM-1 accumulates on top of the bias: lane element r = hidden unit tn*32 + 8 (r >> 2) + 4 lhalf + (r & 3) of the chunk.
        // (Read here, behind GEMM-2's MFMAs, the four LDS round trips cost nothing; read in the VALU slot they were a serial
        //  read -> wait per group of four values and made that slot the longer one of the two.)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 bb = *(const f32x4*)(sB1 + j * HC + tn * 32 + 8 * g + 4 * lhalf);
#pragma unroll
          for (int i = 0; i < 4; ++i) { acc1[4 * g + i] = bb[i]; acc1b[4 * g + i] = 0.f; }
        }
        if (!(p.dbg & 2))
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          if (ks + 1 < KS) load1(ks + 1, (ks + 1) & 1);
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) {
            if (kk & 1) acc1b = mfma_32x32x16(fw[ks & 1][kk], areg[ks * 4 + kk], acc1b);
            else acc1 = mfma_32x32x16(fw[ks & 1][kk], areg[ks * 4 + kk], acc1);
          }
        }
      }
    } else if (sl >= 1 && j < NJ && !(p.dbg & 4)) {
      // ---- VALU slot of chunk j.  lane: token row tq*32 + lrow, hidden units tn*32 + 8g + 4*lhalf + (0..3) for g = 0..3 ----
      const int hrow = tq * 32 + lrow;
      const int hswz = (hrow >> 1) & 7;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int nl = tn * 32 + 8 * g + 4 * lhalf;          // hidden unit inside the chunk (multiple of 4)
        // (the bias is already in the accumulator: GEMM-1 started from it)
        const float h0 = act_apply16(acc1[4 * g] + acc1b[4 * g], ACT), h1 = act_apply16(acc1[4 * g + 1] + acc1b[4 * g + 1], ACT);
        const float h2 = act_apply16(acc1[4 * g + 2] + acc1b[4 * g + 2], ACT), h3 = act_apply16(acc1[4 * g + 3] + acc1b[4 * g + 3], ACT);
        const int off = hrow * 128 + (((nl >> 3) ^ hswz) << 4) + ((nl & 7) << 1);
        // Written with an opaque ds_write: for a visible LDS store hipcc first drains the in-flight weight DMA (it cannot tell
        // that H and the DMA destinations are disjoint LDS regions), which would serialise the prefetch every chunk.
        const uint64_t pk = (uint64_t)(pack_op2(h0, h1)) | ((uint64_t)(pack_op2(h2, h3)) << 32);
        asm volatile("ds_write_b64 %0, %1" ::"v"(h_lds + (uint32_t)off), "v"(pk) : "memory");
      }
    }
    if (p.trace && blockIdx.x == 300 && (tid & 255) == 0) p.trace[256 + (tid >> 8) * 64 + s] = clock64();              // before the DMA wait
    if (s & 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the weights issued in slot s-1 (first read in slot s+1)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (p.trace && blockIdx.x == 300 && (tid & 255) == 0) p.trace[(tid >> 8) * 128 + 2 * s] = clock64();       // work of the slot done
    __builtin_amdgcn_s_barrier();
    if (p.trace && blockIdx.x == 300 && (tid & 255) == 0) p.trace[(tid >> 8) * 128 + 2 * s + 1] = clock64();   // barrier passed
  }

  // ---- epilogue: acc2 -> per-wave LDS slab [32][C/2] fp32 -> + b2 + x -> out ----
  constexpr int WN = C / 2;
  constexpr int LPR = WN / 4;                      // lanes per row (float4 each): 32 at C = 256
  constexpr int RPP = 64 / LPR;
  constexpr int NPASS = 32 / RPP;
  const int c0 = (lane % LPR) * 4;
  const int n = wn * WN + c0;
  const int mrow0 = m0 + (grp * 2 + wm) * 32 + lane / LPR;
  __syncthreads();
  float* sC = (float*)smem + wave * (32 * WN);
#pragma unroll
  for (int t = 0; t < TN2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) sC[((r & 3) + 8 * (r >> 2) + 4 * lhalf) * WN + t * 32 + lrow] = acc2[t][r];
  // every residual row of this lane in flight at once (the accumulators are dead now): one exposed HBM round trip instead of
  // NPASS / 4; the loads land while the slab is being synchronised
  float4 xr[NPASS];
#pragma unroll
  for (int u = 0; u < NPASS; ++u) {
    const int m = mrow0 + u * RPP;
    xr[u] = make_float4(0, 0, 0, 0);
    if (m < p.M && !(p.dbg & 32)) xr[u] = *(const float4*)(p.x + (int64_t)m * C + n);
  }
  const float4 bias = *(const float4*)(p.b2 + n);
  // (no workgroup barrier: the slab is private to the wave)
#pragma unroll
  for (int u = 0; u < NPASS; ++u) {
    const int m = mrow0 + u * RPP;
    if (m >= p.M || (p.dbg & 64)) continue;
    const float4 a4 = *(const float4*)(sC + (u * RPP + lane / LPR) * WN + c0);
    *(float4*)(p.out + (int64_t)m * C + n) =
        make_float4(a4.x + bias.x + xr[u].x, a4.y + bias.y + xr[u].y, a4.z + bias.z + xr[u].z, a4.w + bias.w + xr[u].w);
  }
#endif
}

// ================================================================================================================================
// 64-row variant for units = 256 (the SEVIR-LR level-0 blocks): 512 threads, 64 rows, 76 KB of LDS, <= 128 VGPRs -> TWO workgroups
// = 16 waves per CU (four per SIMD).  Same reasoning as csrc/attn_block.hip: the 128-row kernel above keeps one 144 KB workgroup per
// CU whose waves are latency chains (fragment round trips, MFMA -> activation -> LDS, a barrier per slot) with the MFMA pipe
// about a third busy, and its HBM phases (LayerNorm rows in, residual in, rows out: 55 of 235 us) overlap nothing.  Here a wave
// holds the A fragments of 16 rows and 32 accumulator registers; W1_j / W2_j ([64 x 256] / [256 x 64], 32 KB each) alternate
// through two LDS slots (slot 0 = the A-tile region), each requested one step ahead.
//   chunk j:  step A  H_j^T[64 hidden x 64 rows] = W1_j A^T + b1 (wave: 32 hidden x 16 rows, accumulator starts from the bias),
//                     activation, bf16 -> H tile (8 B stores: 4 consecutive hidden units of one token)
//             step B  acc[64 x 256] += H_j W2_j^T (wave: 32 rows x 32 columns of each 128-column half)
//   one workgroup barrier per step; the DMA of the next step's weights is in flight during the current one.
#define FFN64_WLD(dst, base_vgpr, ks, dt)                                                                               \
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(((ks) & 1) ? ((base_vgpr) ^ 64u) : (base_vgpr)),     \
               "n"(((ks) >> 1) * 8192 + (dt) * 2048))

// NS = weight slots: 2 (76 KB: two workgroups per CU) or 4 (140 KB, one workgroup per CU: grids of at most one workgroup per CU
// anyway) -- three chunks (96 KB) in flight instead of one, so that a step no longer waits for the L2 -> LDS latency of its successor.
template <int ACT, int NS = 2>
__global__ void __launch_bounds__(512, NS == 2 ? 4 : 2) ffn64_kernel(const pd_ffn_args_k p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int C = 256, BM = 64, HC = 64, KS = C / 64;
  constexpr int SLOT = 32768;                      // one weight chunk: W1_j [64 hidden][256 k] or W2_j [256 out][64 hidden]
  constexpr int PF = 2;                            // W1-fragment prefetch distance, k-steps of 32
  constexpr int NSTEP = 2 * KS;                    // k-steps of 32 over K = C
  static_assert(NS % 2 == 0, "W1 chunks (even) and W2 chunks (odd) keep their slot parity: chunk s lives in slot (s + 1) % NS");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sS0 = smem;                                // A tile, then weight slot 0
  char* sH = smem + NS * SLOT;                     // H tile [64 rows][64 hidden] bf16, 16 B chunk XOR (row >> 1) & 7
  float* sB1 = (float*)(sH + BM * HC * 2);         // whole b1 (Hd floats)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m0 = blockIdx.x * BM;
  const int NJ = p.Hd / HC;
  const int NCHUNK = 2 * NJ;                       // chunk s: even = W1_{s/2}, odd = W2_{s/2}; chunk s lives in slot (s + 1) & 1

  const auto rW1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.W1, 0, p.w1_bytes, 0x00020000);
  const auto rW2 = __builtin_amdgcn_make_buffer_rsrc((void*)p.W2, 0, p.w2_bytes, 0x00020000);
  // DMA lane mapping: one 512-thread instruction fills one [64 rows][64 k] slab (8 KB), lane-linear, source-side swizzle
  const int drow = tid >> 3, dpos = tid & 7;
  const int dchunk = dpos ^ ((drow >> 1) & 7);
  const uint32_t w1_voff = ((uint32_t)drow * C + dchunk * 8) * 2u;                       // + (j*64*C + i*64)*2: K slab i of W1_j
  const uint32_t w2_voff = ((uint32_t)drow * (uint32_t)p.Hd + dchunk * 8) * 2u;          // + (i*64*Hd + j*64)*2: output slab i of W2_j
  auto slot_of = [&](int s) { return sS0 + ((s + 1) % NS) * SLOT; };
  auto issue = [&](int s) {
    char* d = slot_of(s) + wave * 1024;
    const int j = s >> 1;
    if (!(s & 1)) {
#pragma unroll
      for (int i = 0; i < KS; ++i) BLDS16(rW1, d + i * 8192, w1_voff, (j * HC * C + i * 64) * 2);
    } else {
#pragma unroll
      for (int i = 0; i < KS; ++i) BLDS16(rW2, d + i * 8192, w2_voff, (i * 64 * p.Hd + j * HC) * 2);
    }
  };
#pragma unroll
  for (int s0 = 0; s0 < NS - 1; ++s0) issue(s0);   // chunks 0 .. NS-2 -> slots 1 .. NS-1 (slot 0 is still the A tile)

  // ---- phase 0: LayerNorm -> bf16 A tile (KS slabs of [64][64]); b1 -> LDS while the row loads are in flight ----
  ln_block_to_tile<C, BM, 8>(p.x, p.gamma, p.beta, p.eps, sS0, wave, lane, (p.dbg & 16) != 0,
                             [&](int r) { const int m = m0 + r; return m < p.M ? m : -1; },
                             [&]() { for (int i = tid; i < p.Hd; i += 512) sB1[i] = p.b1[i]; });

  // ---- wave roles ----
  const int l16 = lane & 15, lg = lane >> 4;
  const int swz16 = (l16 >> 1) & 7;
  const int lrow = lane & 31, lhalf = lane >> 5;
  const int swz = (lrow >> 1) & 7;
  const int tn = wave & 1, tq = wave >> 1;         // GEMM-1: 32-hidden half tn x 16-row tile tq
  const int wm = wave >> 2, wn = wave & 3;         // GEMM-2: 32-row tile wm x 32-column tile wn of every 128-column half
  f32x16 acc2[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[t][r] = 0.f;

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();                                 // A tile written by all waves; W1_0 landed
  op8 areg[NSTEP];                              // this wave's 16 token rows, all of K, for the whole kernel
#pragma unroll
  for (int ks = 0; ks < NSTEP; ++ks)
    areg[ks] = *(const op8*)(sS0 + (ks >> 1) * (BM * 128) + (tq * 16 + l16) * 128 + ((((ks & 1) * 4 + lg) ^ swz16) << 4));
  __syncthreads();                                 // the A-tile region is free: weight slot 0
  issue(NS - 1);                                   // -> slot 0

  const uint32_t w_lane_off = (uint32_t)((tn * 32 + l16) * 128 + ((lg ^ swz16) << 4));
  const uint32_t h_lds = (uint32_t)(uintptr_t)sH, b1_lds = (uint32_t)(uintptr_t)sB1;
  const int g2_a = (wm * 32 + lrow) * 128;                               // H row of GEMM-2's A operand
  const int g2_b = (wn >> 1) * 8192 + ((wn & 1) * 32 + lrow) * 128;      // W2 chunk: slab (64 output channels), row in it; + oh * 16384
  // end of a step: the next step's weights have landed (the only DMA in flight), everyone is done with this step's slot and H
  // (chunks s+2 .. s+NS-1 may stay in flight -- the counted wait is exact only when all of them were issued)
  auto step_end = [&](int s) {
    if (NS > 2 && s + NS - 1 < NCHUNK && !(p.dbg & 1)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * KS) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (s + NS < NCHUNK && !(p.dbg & 1)) issue(s + NS);
  };

  for (int j = 0; j < NJ; ++j) {
    // ---------------- step A: H_j^T = W1_j A^T + b1, activation -> H tile ----------------
    {
      // lane: token row tq*16 + l16 (column of the tile), hidden units tn*32 + 16 dt + 4 lg + (0..3)
      f32x4 acc1[2], bb[2];
      const uint32_t w_lane = (uint32_t)(uintptr_t)slot_of(2 * j) + w_lane_off;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
        asm volatile("ds_read_b128 %0, %1" : "=v"(bb[dt]) : "v"(b1_lds + (uint32_t)((j * HC + tn * 32 + dt * 16 + 4 * lg) * 4)));
      op8 w[PF + 1][2];
#pragma unroll
      for (int i = 0; i < PF; ++i) {
        FFN64_WLD(w[i][0], w_lane, i, 0);
        FFN64_WLD(w[i][1], w_lane, i, 1);
      }
      if (!(p.dbg & 2)) {
#pragma unroll
        for (int ks = 0; ks < NSTEP; ++ks) {
          if (ks + PF < NSTEP) {
            FFN64_WLD(w[(ks + PF) % (PF + 1)][0], w_lane, ks + PF, 0);
            FFN64_WLD(w[(ks + PF) % (PF + 1)][1], w_lane, ks + PF, 1);
            asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * PF) : "memory");
          } else {
            asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * (NSTEP - 1 - ks)) : "memory");
          }
          __builtin_amdgcn_sched_barrier(0);
          if (ks == 0) { acc1[0] = bb[0]; acc1[1] = bb[1]; }      // (the bias reads are older than every fragment read: landed)
#pragma unroll
          for (int dt = 0; dt < 2; ++dt)
            acc1[dt] = mfma_16x16x32(w[ks % (PF + 1)][dt], areg[ks], acc1[dt]);
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        acc1[0] = bb[0]; acc1[1] = bb[1];
      }
      if (!(p.dbg & 4)) {
        const int trow = tq * 16 + l16;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const int d = tn * 32 + dt * 16 + 4 * lg;
          const float h0 = act_apply16(acc1[dt][0], ACT), h1 = act_apply16(acc1[dt][1], ACT);
          const float h2 = act_apply16(acc1[dt][2], ACT), h3 = act_apply16(acc1[dt][3], ACT);
          const uint64_t pk = (uint64_t)(pack_op2(h0, h1)) | ((uint64_t)(pack_op2(h2, h3)) << 32);
          const int off = trow * 128 + (((d >> 3) ^ ((trow >> 1) & 7)) << 4) + ((d & 7) << 1);
          // opaque ds_write: a visible LDS store would make hipcc drain the in-flight weight DMA first
          asm volatile("ds_write_b64 %0, %1" ::"v"(h_lds + (uint32_t)off), "v"(pk) : "memory");
        }
      }
      step_end(2 * j);                             // W2_j landed, H_j visible, slot 1 free -> W1_{j+1}
    }
    // ---------------- step B: acc += H_j W2_j^T ----------------
    {
      if (!(p.dbg & 8)) {
        op8 fa[2], fb[2][2];                                         // [pipeline slot][output half]: two k-sub-steps in flight
        const uint32_t xs = (uint32_t)((lhalf ^ swz) << 4);              // 16 B slot of k-sub-step 0; sub-step kk: ^ (kk << 5)
        const uint32_t a2 = h_lds + (uint32_t)g2_a + xs;
        const uint32_t b2 = (uint32_t)(uintptr_t)slot_of(2 * j + 1) + (uint32_t)g2_b + xs;
        auto ld2 = [&](int kk, int slot) {
          asm volatile("ds_read_b128 %0, %1" : "=v"(fa[slot]) : "v"(a2 ^ (uint32_t)(kk << 5)));
          asm volatile("ds_read_b128 %0, %1" : "=v"(fb[slot][0]) : "v"(b2 ^ (uint32_t)(kk << 5)));
          asm volatile("ds_read_b128 %0, %1 offset:16384" : "=v"(fb[slot][1]) : "v"(b2 ^ (uint32_t)(kk << 5)));
        };
        ld2(0, 0);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          if (kk + 1 < 4) {
            ld2(kk + 1, (kk + 1) & 1);
            asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");
          } else {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          }
          __builtin_amdgcn_sched_barrier(0);
          acc2[0] = mfma_32x32x16(fa[kk & 1], fb[kk & 1][0], acc2[0]);
          acc2[1] = mfma_32x32x16(fa[kk & 1], fb[kk & 1][1], acc2[1]);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      step_end(2 * j + 1);                         // W1_{j+1} landed, slot 0 and H free -> W2_{j+1}
    }
  }

  // ---- epilogue: acc2 -> per-wave LDS slab [32][64] fp32 -> + b2 + x -> out ----
  constexpr int WN = 64;                           // columns per wave: 2 pieces of 32
  constexpr int LPR = WN / 4, RPP = 64 / LPR, NPASS = 32 / RPP;
  const int c0 = (lane % LPR) * 4;                 // slab column; output column = 128 (c0 / 32) + 32 wn + c0 % 32
  const int n = (c0 >> 5) * 128 + wn * 32 + (c0 & 31);
  float* sC = (float*)smem + wave * (32 * WN);     // (the last step_end left every wave past its weight / H reads)
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) sC[((r & 3) + 8 * (r >> 2) + 4 * lhalf) * WN + t * 32 + lrow] = acc2[t][r];
  const int mrow0 = m0 + wm * 32 + lane / LPR;
  float4 xr[NPASS];
#pragma unroll
  for (int u = 0; u < NPASS; ++u) {
    const int m = mrow0 + u * RPP;
    xr[u] = make_float4(0, 0, 0, 0);
    if (m < p.M && !(p.dbg & 32)) xr[u] = *(const float4*)(p.x + (int64_t)m * C + n);
  }
  const float4 bias = *(const float4*)(p.b2 + n);
  // (no workgroup barrier: the slab is private to the wave)
#pragma unroll
  for (int u = 0; u < NPASS; ++u) {
    const int m = mrow0 + u * RPP;
    if (m >= p.M || (p.dbg & 64)) continue;
    const float4 a4 = *(const float4*)(sC + (u * RPP + lane / LPR) * WN + c0);
    *(float4*)(p.out + (int64_t)m * C + n) =
        make_float4(a4.x + bias.x + xr[u].x, a4.y + bias.y + xr[u].y, a4.z + bias.z + xr[u].z, a4.w + bias.w + xr[u].w);
  }
#endif
}

template <int ACT, int NS = 2>
static int launch_ffn64(const pd_ffn_args_k& a, hipStream_t s) {
  const int bytes = NS * 32768 + 64 * 64 * 2 + a.Hd * 4;         // the weight slots (the epilogue slab re-uses them) + H tile + b1
  static int attr_set_dev[PD_MAX_DEVICES];
  int& attr_set = attr_set_dev[pd_cur_device()];
  if (attr_set < bytes) {
    hipError_t e = hipFuncSetAttribute((const void*)ffn64_kernel<ACT, NS>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) {
      pd_set_error("pd_ffn_fused: hipFuncSetAttribute(%d) failed: %s", bytes, hipGetErrorString(e));
      return PD_ERR_LAUNCH;
    }
    attr_set = bytes;
  }
  hipLaunchKernelGGL((ffn64_kernel<ACT, NS>), dim3((a.M + 63) / 64), dim3(512), bytes, s, a);
  PD_CHECK_LAUNCH();
  return PD_OK;
}

template <int C, int ACT>
static int launch_ffn(const pd_ffn_args_k& a, hipStream_t s) {
  const int lds = 128 * C * 2 + 128 * 64 * 2 + 64 * C * 2 + C * 64 * 2 + a.Hd * 4;
  constexpr int epi = 8 * 32 * (C / 2) * 4;
  const int bytes = lds > epi ? lds : epi;
  static int attr_set_dev[PD_MAX_DEVICES];
  int& attr_set = attr_set_dev[pd_cur_device()];
  if (attr_set < bytes) {
    hipError_t e = hipFuncSetAttribute((const void*)ffn_fused_kernel<C, ACT>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) {
      pd_set_error("pd_ffn_fused: hipFuncSetAttribute(%d) failed: %s", bytes, hipGetErrorString(e));
      return PD_ERR_LAUNCH;
    }
    attr_set = bytes;
  }
  hipLaunchKernelGGL((ffn_fused_kernel<C, ACT>), dim3((a.M + 127) / 128), dim3(512), bytes, s, a);
  PD_CHECK_LAUNCH();
  return PD_OK;
}

#if !PD_IS_F16
extern "C" int pd_ffn_fused_supported(int C, int Hd) {
  return (C == 64 || C == 128 || C == 256) && Hd > 0 && Hd % 64 == 0 && Hd <= 3072;   // LDS: 144 KB tiles + 4*Hd bytes of bias
}
extern "C" int pd_f16_ffn_fused(const float*, float*, const float*, const float*, const pd_bf16*, const float*, const pd_bf16*, const float*, int64_t, int,
                                int, int, float, const pd_call_opts*, pd_stream_t);
#else
extern "C" int pd_ffn_fused_supported(int C, int Hd);
#endif

extern "C" int PD_ENTRY(ffn_fused)(const float* x, float* out, const float* gamma, const float* beta, const pd_bf16* W1, const float* b1,
                                   const pd_bf16* W2, const float* b2, int64_t M, int C, int Hd, int act, float eps, const pd_call_opts* opts,
                                   pd_stream_t stream) {
  PD_FORWARD_F16(PD_OPTS_F16(opts), pd_f16_ffn_fused(x, out, gamma, beta, W1, b1, W2, b2, M, C, Hd, act, eps, opts, stream));
  PD_CHECK_ARG(x && out && gamma && beta && W1 && b1 && W2 && b2, "pd_ffn_fused: null pointer");
  PD_CHECK_ARG(pd_ffn_fused_supported(C, Hd), "pd_ffn_fused: unsupported units=%d hidden=%d (units in {64,128,256}, hidden %% 64 == 0)", C, Hd);
  PD_CHECK_ARG(M > 0 && M < (1ll << 31), "pd_ffn_fused: bad M");
  pd_ffn_args_k a;
  a.x = x; a.out = out; a.gamma = gamma; a.beta = beta; a.W1 = W1; a.b1 = b1; a.W2 = W2; a.b2 = b2;
  a.M = (int)M; a.Hd = Hd; a.act = act; a.eps = eps;
  a.w1_bytes = (uint32_t)((int64_t)Hd * C * 2);
  a.w2_bytes = (uint32_t)((int64_t)C * Hd * 2);
  a.dbg = opts ? opts->ffn_debug_flags : 0;           // (profiling ablations / per-phase clock stamps: scripts/bench_ffn.py)
  a.trace = opts ? opts->trace : nullptr;
  const bool use_64 = !(opts && opts->ffn_rows128);     // units 256: the 64-row, two-workgroups-per-CU kernel (A/B switch: the 128-row kernel)
  hipStream_t s = (hipStream_t)stream;
#define PD_FFN(ACT)                                  \
  if (C == 256 && use_64 && Hd * 4 + 2 * 32768 + 8192 <= 80 * 1024) return launch_ffn64<ACT>(a, s);   \
  if (C == 256) return launch_ffn<256, ACT>(a, s);   \
  if (C == 128) return launch_ffn<128, ACT>(a, s);   \
  return launch_ffn<64, ACT>(a, s);
  switch (act) {
    case PD_ACT_GELU: PD_FFN(PD_ACT_GELU)
    case PD_ACT_LEAKY: PD_FFN(PD_ACT_LEAKY)
    case PD_ACT_RELU: PD_FFN(PD_ACT_RELU)
    case PD_ACT_SILU: PD_FFN(PD_ACT_SILU)
    case PD_ACT_NONE: PD_FFN(PD_ACT_NONE)
    default: pd_set_error("pd_ffn_fused: unknown activation %d", act); return PD_ERR_ARG;
  }
#undef PD_FFN
}

}  // namespace PD_NS
