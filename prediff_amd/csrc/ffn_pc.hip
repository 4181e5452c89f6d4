// pd_ffn_fused_pc: out = x + W2 * act(W1 * LayerNorm(x) + b1) + b2 for units = 256 (PositionwiseFFN.forward, pre-norm, reference
// cuboid_transformer.py:182-208) with the two GEMMs on DIFFERENT waves of the workgroup -- round 3's answer to what the round-2
// measurements of ffn64_kernel (csrc/ffn.hip) showed:
//   * all eight waves of a workgroup ran the same step at the same time (GEMM-1 + GELU, barrier, GEMM-2, barrier): the two waves a
//     workgroup has on a SIMD want the MFMA pipe together and the VALU together, and 32 barrier-separated steps per tile each pay an
//     LDS round trip, a DMA wait and the barrier skew -- MFMA pipe 30-34 % busy with one OR two workgroups per CU;
//   * every W1 fragment read from LDS fed ONE 16-row MFMA (each weight byte crossed the LDS four times), and both weight matrices went
//     through the LDS ring: LDS time (reads + DMA writes) exceeded the MFMA time of a step.
// Here:  waves 0-3 = PRODUCERS   H_j = act(LN(x) W1_j^T + b1_j)   (64 rows x 64 hidden units per chunk j, bf16 -> LDS, double buffered)
//        waves 4-7 = CONSUMERS   acc[64 x 256] += H_{j-1} W2_{j-1}^T, one chunk behind the producers
// so that every SIMD holds one producer and one consumer of each workgroup (MFMA of one beside the GELU VALU of the other), with ONE
// workgroup barrier per chunk (Hd/64 + 1 per tile instead of 2 Hd/64).
//   producer  wave (tq, tn): 32 rows x 32 hidden units; the LN output of its 32 rows lives in registers for the whole kernel (64 VGPRs),
//             a W1 fragment [16 hidden x 32 k] read from LDS feeds TWO MFMAs (both 16-row tiles): half the LDS reads of ffn64.
//             W1_j [64 x 256] (32 KB) streams through two LDS slots by buffer-descriptor DMA issued by the producers, one chunk ahead.
//   consumer  wave cw: all 64 rows x output columns 64 cw .. 64 cw + 63 (four 32 x 32 accumulators).  Nobody else needs its W2
//             fragments, so they never touch the LDS: the host stores W2 in FRAGMENT ORDER (pd_ffn_pack_w2_frag layout below) and the
//             wave loads each 1 KB fragment with one fully coalesced global_load_dwordx4, a whole chunk (8 fragments) ahead, into the
//             registers of the fragment it has just consumed (inline asm + counted vmcnt: the consumers issue no DMA, so their vmcnt
//             counts only these loads).
// LDS: 2 x 32 KB W1 slots (slot 0 = the LN tile first) + 2 x 8 KB H tiles = 80 KB -> two workgroups per CU, <= 128 VGPRs.
// Numerics are ffn64's: bf16 LN output / weights / hidden activations, fp32 accumulation, same accumulation order per output element
// within a chunk (k ascending), chunks ascending.
#include "common.h"
#include "ln_tile.h"

#define BLDS16(rsrc, ldsptr, voff, soff) \
  __builtin_amdgcn_raw_ptr_buffer_load_lds((rsrc), (__attribute__((address_space(3))) void*)(ldsptr), 16, (voff), (soff), 0, 0)

struct pd_ffn_pc_args_k {
  const float* x;
  float* out;
  const float* gamma;
  const float* beta;
  const pd_bf16* W1;      // [Hd][256] bf16, K contiguous
  const float* b1;
  const pd_bf16* W2f;     // fragment order: [Hd/64 chunk j][4 consumer cw][4 kk][2 ct][64 lanes][8]: element e of lane (kh = lane >> 5, n = lane & 31)
                          //                 = W2[64 cw + 32 ct + n][64 j + 16 kk + 8 kh + e]
  const float* b2;
  int M, Hd;
  float eps;
  uint32_t w1_bytes;
  int dbg;                // ablations (scripts/bench_ffn.py): 1 no W1 DMA after the first two chunks, 2 no GEMM-1, 4 no activation, 8 no GEMM-2
};

// W1 fragment of k-step ks (32 deep), hidden tile dt of the producer's 32-unit half: slab ks >> 1, row tn*32 + dt*16 + l16,
// 16 B chunk ((ks & 1) * 4 + lg) ^ swz16 (the lane part sits in a VGPR, an odd k-step flips bit 6, the rest is the immediate)
#define PC_WLD(dst, base_vgpr, ks, dt)                                                                                  \
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(((ks) & 1) ? ((base_vgpr) ^ 64u) : (base_vgpr)),     \
               "n"(((ks) >> 1) * 8192 + (dt) * 2048))

template <int ACT>
__global__ void __launch_bounds__(512, 4) ffn_pc_kernel(const pd_ffn_pc_args_k p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int C = 256, BM = 64, HC = 64, KS = C / 64;
  constexpr int SLOT = 32768;                      // W1_j [64 hidden][256 k] bf16
  constexpr int NSTEP = 2 * KS;                    // k-steps of 32 over K = C
  constexpr int PF = 2;                            // W1-fragment prefetch distance, k-steps
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sS0 = smem;                                // LN tile, then W1 slot 0
  char* sS1 = smem + SLOT;                         // W1 slot 1
  char* sH = smem + 2 * SLOT;                      // two H tiles [64 rows][64 hidden] bf16, 16 B chunk XOR (row >> 1) & 7

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m0 = blockIdx.x * BM;
  const int NJ = p.Hd / HC;
  const bool producer = wave < 4;

  const auto rW1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.W1, 0, p.w1_bytes, 0x00020000);
  // W1 DMA, issued by the four producer waves: one 256-thread instruction fills half a slab = [32 rows][64 k] (4 KB), lane-linear,
  // source-side swizzle.  Chunk j = 4 K-slabs x 2 row halves = 8 instructions; chunk j lives in slot (j + 1) & 1.
  const int drow = (tid & 255) >> 3, dpos = tid & 7;
  const int dchunk = dpos ^ ((drow >> 1) & 7);     // ((drow + 32) >> 1) & 7 is the same: one swizzle for both row halves
  const uint32_t w1_voff = ((uint32_t)drow * C + dchunk * 8) * 2u;
  auto issue_w1 = [&](int j) {
    char* d = ((j + 1) & 1 ? sS1 : sS0) + wave * 1024;
#pragma unroll
    for (int i = 0; i < KS; ++i)
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) BLDS16(rW1, d + i * 8192 + hh * 4096, w1_voff, ((j * HC + hh * 32) * C + i * 64) * 2);
  };
  if (producer) issue_w1(0);                       // -> slot 1 (slot 0 still receives the LN tile)

  // ---- phase 0: LayerNorm of the 64 rows (8 per wave, all waves) -> bf16 tile in slot 0 ----
  ln_block_to_tile<C, BM, 8>(p.x, p.gamma, p.beta, p.eps, sS0, wave, lane, false,
                             [&](int r) { const int m = m0 + r; return m < p.M ? m : -1; });

  const int l16 = lane & 15, lg = lane >> 4;
  const int swz16 = (l16 >> 1) & 7;
  const int lrow = lane & 31, lhalf = lane >> 5;
  const uint32_t h_lds = (uint32_t)(uintptr_t)sH;

  if (producer) {
    // =========================================================================================================== producers
    const int tn = wave & 1, tq = wave >> 1;       // 32-hidden half tn x 32-row half tq
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                               // (1) LN tile complete; this wave's part of W1_0 landed
    bf16x8 areg[2][NSTEP];                         // the wave's 32 token rows (two 16-row tiles), all of K, for the whole kernel
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int ks = 0; ks < NSTEP; ++ks)
        areg[rt][ks] = *(const bf16x8*)(sS0 + (ks >> 1) * (BM * 128) + (tq * 32 + rt * 16 + l16) * 128 + ((((ks & 1) * 4 + lg) ^ swz16) << 4));
    __syncthreads();                               // (2) slot 0 is free
    const uint32_t w_lane_off = (uint32_t)((tn * 32 + l16) * 128 + ((lg ^ swz16) << 4));
    // bias of chunk j for this lane: hidden units tn*32 + dt*16 + 4 lg + (0..3).  Loaded with opaque loads one chunk ahead, in front of
    // the end-of-step vmcnt(0) (a compiler-visible load next to the LDS DMA would drain the DMA queue at its first use).
    f32x4 bnext[2];
    const float* b1_lane = p.b1 + tn * 32 + 4 * lg;
    auto load_bias = [&](int j) {
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(bnext[dt]) : "v"(b1_lane + j * HC + dt * 16) : "memory");
    };
    load_bias(0);
    if (NJ > 1) issue_w1(1);                       // -> slot 0 (8 DMA instructions, younger than the bias loads: they stay in flight)
    asm volatile("s_waitcnt vmcnt(8)" : "+v"(bnext[0]), "+v"(bnext[1]) : : "memory");

    for (int j = 0; j < NJ; ++j) {
      f32x4 acc[2][2];                             // [row tile rt][hidden tile dt]: H^T tile, lane = token row rt*16 + l16, hidden 4 lg + (0..3)
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) acc[rt][dt] = bnext[dt];       // GEMM-1 accumulates on top of the bias
      const uint32_t w_lane = (uint32_t)(uintptr_t)((j + 1) & 1 ? sS1 : sS0) + w_lane_off;
      if (!(p.dbg & 2)) {
        bf16x8 w[PF + 1][2];
#pragma unroll
        for (int i = 0; i < PF; ++i) {
          PC_WLD(w[i][0], w_lane, i, 0);
          PC_WLD(w[i][1], w_lane, i, 1);
        }
#pragma unroll
        for (int ks = 0; ks < NSTEP; ++ks) {
          if (ks + PF < NSTEP) {
            PC_WLD(w[(ks + PF) % (PF + 1)][0], w_lane, ks + PF, 0);
            PC_WLD(w[(ks + PF) % (PF + 1)][1], w_lane, ks + PF, 1);
            asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * PF) : "memory");
          } else {
            asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * (NSTEP - 1 - ks)) : "memory");
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
              acc[rt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[ks % (PF + 1)][dt], areg[rt][ks], acc[rt][dt], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (j + 1 < NJ) load_bias(j + 1);
      // activation -> bf16 -> H tile j & 1 (8 B stores: 4 consecutive hidden units of one token)
      const uint32_t hb = h_lds + (uint32_t)((j & 1) * (BM * HC * 2));
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        const int trow = tq * 32 + rt * 16 + l16;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const int d = tn * 32 + dt * 16 + 4 * lg;
          float h0 = acc[rt][dt][0], h1 = acc[rt][dt][1], h2 = acc[rt][dt][2], h3 = acc[rt][dt][3];
          if (!(p.dbg & 4)) { h0 = act_apply(h0, ACT); h1 = act_apply(h1, ACT); h2 = act_apply(h2, ACT); h3 = act_apply(h3, ACT); }
          const uint64_t pk = (uint64_t)(pack_bf16x2(h0, h1)) | ((uint64_t)(pack_bf16x2(h2, h3)) << 32);
          const int off = trow * 128 + (((d >> 3) ^ ((trow >> 1) & 7)) << 4) + ((d & 7) << 1);
          // opaque ds_write: a visible LDS store would make hipcc drain the in-flight weight DMA first
          asm volatile("ds_write_b64 %0, %1" ::"v"(hb + (uint32_t)off), "v"(pk) : "memory");
        }
      }
      // end of the step: W1_{j+1} (issued a step ago) and the next bias have landed, H_j is written
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(bnext[0]), "+v"(bnext[1]) : : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                // (3 + j)
      if (j + 2 < NJ && !(p.dbg & 1)) issue_w1(j + 2);        // into the slot W1_j has just left
    }
    __builtin_amdgcn_s_barrier();                  // (3 + NJ): the consumers' last chunk
  } else {
    // =========================================================================================================== consumers
    const int cw = wave - 4;                       // output columns 64 cw .. 64 cw + 63
    const pd_bf16* wbase = p.W2f + ((int64_t)cw * 8 * 64 + lane) * 8;      // + (j * 4 * 8 * 64 + f * 64) * 8 for fragment f = kk * 2 + ct of chunk j
    bf16x8 fb[8];
    auto load_frag = [&](int j, int f) {
      const pd_bf16* src = wbase + ((int64_t)j * 4 * 8 * 64 + f * 64) * 8;
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(fb[f]) : "v"(src) : "memory");
    };
#pragma unroll
    for (int f = 0; f < 8; ++f) load_frag(0, f);   // chunk 0, consumed in the step after the producers' first
    f32x16 acc[2][2];                              // [row tile rt of 32][column tile ct of 32]
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[rt][ct][r] = 0.f;
    __syncthreads();                               // (1)
    __syncthreads();                               // (2)
    const int swz = (lrow >> 1) & 7;
    const uint32_t xs = (uint32_t)((lhalf ^ swz) << 4);     // 16 B slot of k-sub-step 0 of this lane's H row; sub-step kk: ^ (kk << 5)
    for (int j = 0; j <= NJ; ++j) {
      if (j >= 1) {
        // chunk jc = j - 1: acc += H_jc W2_jc^T.  Fragment f = 2 kk + ct is the oldest load in flight when it is needed: the seven
        // younger ones are the rest of this chunk and the part of the next chunk already requested -> vmcnt(7).
        const int jc = j - 1;
        const bool more = jc + 1 < NJ;             // the last chunk requests nothing (a load into a dead register could land in whatever
                                                   // the compiler puts there next) and counts down instead
        const uint32_t hb = h_lds + (uint32_t)((jc & 1) * (BM * HC * 2));
        bf16x8 fa[2][2];
        auto ld_a = [&](int kk, int slot) {
#pragma unroll
          for (int rt = 0; rt < 2; ++rt)
            asm volatile("ds_read_b128 %0, %1" : "=v"(fa[slot][rt]) : "v"((hb + (uint32_t)((rt * 32 + lrow) * 128) + xs) ^ (uint32_t)(kk << 5)));
        };
        ld_a(0, 0);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          if (kk + 1 < 4) {
            ld_a(kk + 1, (kk + 1) & 1);
            asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
          } else {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          }
#pragma unroll
          for (int ct = 0; ct < 2; ++ct) {
            const int f = 2 * kk + ct;
            if (more) asm volatile("s_waitcnt vmcnt(7)" : "+v"(fb[f]) : : "memory");
            else asm volatile("s_waitcnt vmcnt(%1)" : "+v"(fb[f]) : "n"(7 - f) : "memory");
            __builtin_amdgcn_sched_barrier(0);
            if (!(p.dbg & 8)) {
              acc[0][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[kk & 1][0], fb[f], acc[0][ct], 0, 0, 0);
              acc[1][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[kk & 1][1], fb[f], acc[1][ct], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            // the MFMAs have read fb[f] by the time the reload can land (>= an L2 round trip away); the "+v" tie orders the asm after them
            asm volatile("" : "+v"(fb[f]), "+v"(acc[0][ct][0]), "+v"(acc[1][ct][0]));
            if (more) load_frag(jc + 1, f);
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                // (3 + j)
    }
    // ---- accumulators -> LDS slab of this consumer: [64 rows][64 columns] fp32 (16 KB) in the W1 slots (idle now) ----
    float* sC = (float*)smem + cw * (64 * 64);
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) sC[(rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhalf) * 64 + ct * 32 + lrow] = acc[rt][ct][r];
  }
  __syncthreads();                                 // slabs complete
  // ---- epilogue, all eight waves: slab + b2 + x -> out; wave w: rows 8 w .. 8 w + 7, a lane = one float4 of a row ----
  {
    const int c4 = lane * 4;                       // output column
    const float* sC = (const float*)smem + (c4 >> 6) * (64 * 64) + (c4 & 63);
    const float4 bias = *(const float4*)(p.b2 + c4);
    float4 xr[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int m = m0 + wave * 8 + u;
      xr[u] = make_float4(0, 0, 0, 0);
      if (m < p.M) xr[u] = *(const float4*)(p.x + (int64_t)m * C + c4);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int m = m0 + wave * 8 + u;
      if (m >= p.M) continue;
      const float4 a4 = *(const float4*)(sC + (wave * 8 + u) * 64);
      *(float4*)(p.out + (int64_t)m * C + c4) =
          make_float4(a4.x + bias.x + xr[u].x, a4.y + bias.y + xr[u].y, a4.z + bias.z + xr[u].z, a4.w + bias.w + xr[u].w);
    }
  }
#endif
}

template <int ACT>
static int launch_ffn_pc(const pd_ffn_pc_args_k& a, hipStream_t s) {
  constexpr int bytes = 2 * 32768 + 2 * 64 * 64 * 2;            // 80 KB: two workgroups per CU
  static bool attr_set_dev[PD_MAX_DEVICES];
  bool& attr_set = attr_set_dev[pd_cur_device()];
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)ffn_pc_kernel<ACT>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) {
      pd_set_error("pd_ffn_fused_pc: hipFuncSetAttribute(%d) failed: %s", bytes, hipGetErrorString(e));
      return PD_ERR_LAUNCH;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL((ffn_pc_kernel<ACT>), dim3((a.M + 63) / 64), dim3(512), bytes, s, a);
  PD_CHECK_LAUNCH();
  return PD_OK;
}

extern "C" int pd_ffn_pc_debug_flags = 0;

extern "C" int pd_ffn_fused_pc_supported(int C, int Hd) { return C == 256 && Hd >= 128 && Hd % 64 == 0; }

extern "C" int pd_ffn_fused_pc(const float* x, float* out, const float* gamma, const float* beta, const pd_bf16* W1, const float* b1,
                               const pd_bf16* W2_frag, const float* b2, int64_t M, int C, int Hd, int act, float eps, pd_stream_t stream) {
  PD_CHECK_ARG(x && out && gamma && beta && W1 && b1 && W2_frag && b2, "pd_ffn_fused_pc: null pointer");
  PD_CHECK_ARG(pd_ffn_fused_pc_supported(C, Hd), "pd_ffn_fused_pc: unsupported units=%d hidden=%d (units 256, hidden %% 64 == 0, >= 128)", C, Hd);
  PD_CHECK_ARG(M > 0 && M < (1ll << 31), "pd_ffn_fused_pc: bad M");
  pd_ffn_pc_args_k a;
  a.x = x; a.out = out; a.gamma = gamma; a.beta = beta; a.W1 = W1; a.b1 = b1; a.W2f = W2_frag; a.b2 = b2;
  a.M = (int)M; a.Hd = Hd; a.eps = eps;
  a.w1_bytes = (uint32_t)((int64_t)Hd * C * 2);
  a.dbg = pd_ffn_pc_debug_flags;
  hipStream_t s = (hipStream_t)stream;
  switch (act) {
    case PD_ACT_GELU: return launch_ffn_pc<PD_ACT_GELU>(a, s);
    case PD_ACT_LEAKY: return launch_ffn_pc<PD_ACT_LEAKY>(a, s);
    case PD_ACT_RELU: return launch_ffn_pc<PD_ACT_RELU>(a, s);
    case PD_ACT_SILU: return launch_ffn_pc<PD_ACT_SILU>(a, s);
    case PD_ACT_NONE: return launch_ffn_pc<PD_ACT_NONE>(a, s);
    default: pd_set_error("pd_ffn_fused_pc: unknown activation %d", act); return PD_ERR_ARG;
  }
}
