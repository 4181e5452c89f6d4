// pd_attn_block_fused: x += proj(cuboid_attention(qkv(LayerNorm(x)))) in ONE kernel for head_dim 64, cuboid volume <= 64
// (CuboidSelfAttentionLayer.forward + the residual of StackCuboidSelfAttentionBlock, reference cuboid_transformer.py:812-966,
// :1151) -- every axial pattern of the SEVIR-LR denoiser at level 0 (units 256, 4 heads).
//
// Why: un-fused the block is LN -> QKV GEMM -> attention core -> proj GEMM with K = 256 GEMMs and a 2 flop/B core, i.e. four
// HBM-bound launches that write and re-read the bf16 LN output, QKV (3 x) and attention output.  Here a workgroup owns 4 whole
// cuboids (4 x 16 slots = 64 rows, gathered through tok_index -- the cuboid reorder / un-shift never materialises), and
// nothing but x goes to HBM: one gathered fp32 read, one scattered fp32 write (+ the residual re-read).
//
// One workgroup = 512 threads (8 waves), 64 rows, < 80 KB of LDS, <= 128 VGPRs: TWO workgroups = 16 waves per CU.  Round 1's kernel (128 rows, 8 waves,
// 152 KB: one workgroup per CU) spent 58 % of its time in the HBM phases (row gather, residual re-read, scatter: 87 of 148 us
// with every GEMM and the core ablated, profiles/r02_b_attn_block_ablations.log) with idle MFMA pipes; all CUs ran those phases
// in lock step and nothing could overlap them (a persistent variant that requested the next tile's rows early had nowhere to
// put them before the last head: no gain).  Two independent workgroups per CU overlap one's HBM phases with the other's GEMMs
// by construction; the price -- every 64 rows stream the 512 KB of weights instead of every 128 -- is affordable: the LDS DMA
// sustains 114 GB/s per CU with 64 KB in flight (profiles/r02_b_ubench_dma_rate.log), this kernel needs < 50.
// A 4-wave version of the 64-row tile (same 2 waves per SIMD as round 1) ran no faster: every wave's instruction stream is a
// latency chain (LDS round trips, MFMA -> VALU -> LDS epilogues, barriers) with the MFMA pipe 25 % busy, so throughput follows the
// number of resident waves.  Eight waves per 64 rows halve the per-wave state (A fragments of 16 rows, 32 accumulator registers
// of the proj GEMM) and fit four waves on every SIMD.
//   phase 0   LayerNorm of the 64 gathered rows (8 per wave) -> bf16 A tile in LDS -> each wave keeps the fragments of its 16 rows
//             (all of K = C) in registers for the whole kernel; the A region becomes weight buffers.
//   head h    weight chunks of 16 KB through a 3-slot LDS ring, DMA two chunks ahead:
//             Wq_h, Wk_h, Wv_h [64 x C] as C/128 K-halves [64 x 128]; Wp[:, 64h:64h+64] [C x 64] as C/128 output halves [128 x 64]
//       q, k   Q_h^T, K_h^T [64 x 64] = W * A^T (transposed product: a lane ends up with 4 consecutive d of one token -> 8 B
//              writes into row-major [row][d] tiles, the operand layout of S^T = K Q^T), accumulated over the K-halves
//       v      V_h [64 x 64] = A * Wv^T (plain product: a lane ends up with 4 consecutive tokens of one d -> 8 B writes into
//              the [d][row] tile, the A-operand layout of O^T = V^T P^T)
//       core   waves 0-3, wave w = cuboid w: S^T = K Q^T (2 x mfma 16x16x32), scale, + relative-position bias, mask, softmax over keys in
//              registers (4 values + 2 row swaps), O^T = V^T P^T (4 x mfma 16x16x16) -> O tile (over the Q tile: same rows, same wave)
//       proj   acc[64 x C] += O_h[64 x 64] * Wp_h^T, one output half per chunk
//   epilogue  acc + b_proj + x -> out rows (scatter through the same token table).
// LDS: weight ring 3 x 16 KB (two slots = the A region) + Q/O, K, V^T tiles 3 x 8 KB + bias tables + token ids = 79 KB.
// Numerics are those of the un-fused bf16 path (bf16 LN output, bf16 q/k/v/P/O, fp32 accumulation, fp32 softmax).
#include <algorithm>
#include "common.h"
// A/B switch (compile time): 1 runs the cuboid-volume <= 16 kernel through the multi-key-tile core as well (all 8 waves, bias from
// L2, one more workgroup barrier per head).  Measured at the v1 level-0 shapes, 32 trajectories: 161 / 139 us against 154 / 130 us
// for the 4-wave core with its LDS bias table (core phase 2.0 k -> 4.0 k clocks per head) -- so the tuned core stays the default.
#ifndef PD_AB_GENERIC_CORE
#define PD_AB_GENERIC_CORE 0
#endif
#include "ln_tile.h"

namespace PD_NS {

#define BLDS16(rsrc, ldsptr, voff, soff) \
  __builtin_amdgcn_raw_ptr_buffer_load_lds((rsrc), (__attribute__((address_space(3))) void*)(ldsptr), 16, (voff), (soff), 0, 0)

// combine a value over the four 16-lane rows of the wave (lanes l, l^16, l^32, l^48), result in every lane: v_permlane16_swap /
// v_permlane32_swap exchange odd and even rows / the two halves in one VALU instruction each (no LDS crossbar round trip)
__device__ __forceinline__ float rows4_max(float v) {
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float rows4_sum(float v) {
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// W fragment of k-step ks (32 deep, NSTEP = 4 per chunk), feature tile dt of a 16 KB [64 d][128 k] weight chunk: slab (ks >> 1),
// row tn*32 + dt*16 + l16, 16 B chunk ((ks & 1) * 4 + lg) ^ swz16.  The lane part (row, chunk of an even k-step) sits in a VGPR,
// an odd k-step flips bit 6 of it, the rest is the instruction's immediate offset.
#define WFRAG_LD(dst, base_vgpr, ks, dt)                                                                                          \
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(((ks) & 1) ? ((base_vgpr) ^ 64u) : (base_vgpr)),               \
               "n"(((ks) >> 1) * 8192 + (dt) * 2048))
#define WFRAG_PROLOGUE(w, wb)                                                      \
  _Pragma("unroll") for (int i_ = 0; i_ < PF; ++i_) {                              \
    WFRAG_LD(w[i_][0], wb, i_, 0);                                                 \
    WFRAG_LD(w[i_][1], wb, i_, 1);                                                 \
  }
// issue the reads of k-step ks + PF, then wait until those of k-step ks have landed (LDS returns in order: 2 reads per step)
#define WFRAG_STEP(w, wb, ks)                                                      \
  if ((ks) + PF < NSTEP) {                                                         \
    WFRAG_LD(w[((ks) + PF) % (PF + 1)][0], wb, (ks) + PF, 0);                      \
    WFRAG_LD(w[((ks) + PF) % (PF + 1)][1], wb, (ks) + PF, 1);                      \
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * PF) : "memory");               \
  } else {                                                                         \
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * (NSTEP - 1 - (ks))) : "memory"); \
  }                                                                                \
  __builtin_amdgcn_sched_barrier(0);

struct pd_attn_block_args_k {
  const float* x;
  float* out;
  const float* gamma;
  const float* beta;
  const pd_bf16* Wqkv;   // [3C][C]  (q rows, k rows, v rows; head h = rows 64h .. 64h+63 of each)
  const float* bqkv;     // [3C] or null
  const pd_bf16* Wp;     // [C][C]
  const float* bp;       // [C] or null
  const int32_t* tok_index;   // [nc][vol]
  const float* bias;     // [heads][vol][vol]
  const uint8_t* mask;   // [nc][vol][vol] or null
  int B, ntok, nc, vol;
  float scale, eps;
  uint32_t wqkv_bytes, wp_bytes;
  // token of (cuboid c, slot s) = (c / aff_ninner) * aff_outer + (c % aff_ninner) * aff_inner + s * aff_slot when aff_on (un-shifted,
  // un-padded axial cuboids: the host checked the table against this form) -- no dependent table load in front of the row gather
  int aff_on, aff_ninner, aff_outer, aff_inner, aff_slot;
  unsigned long long* trace;   // profiling only: per-phase clock stamps of wave 0 of workgroup 600 (null in production)
  int dbg;   // profiling ablations: 1 no weight DMA after chunk 2, 2 no q/k/v GEMMs, 4 no attention core, 8 no proj GEMM,
             // 16 no LN loads, 32 no residual loads, 64 no stores
};

// KT = key tiles of 16 per cuboid: a workgroup's 64 rows are 4 / KT whole cuboids of up to 16 KT slots each (KT = 1: the axial
// cuboids of the SEVIR-LR grid; KT = 2 / 4: cuboid volumes up to 32 / 64, e.g. 25 and 48 on the 48 x 48 full-resolution grid --
// the slots beyond the volume are empty rows, 22-25 % of the tile there).
// NS = slots of the weight ring: 3 for two workgroups per CU (79 KB each).  (A 6-slot ring for grids of at most one workgroup per
// CU was measured in round 3 and gained nothing: the DMA depth is not what a lone tile waits for.)
template <int C, int KT = 1, int RPC_ = 16 * KT, int NS = 3>
__global__ void __launch_bounds__(512, NS == 3 ? 4 : 2) attn_block_kernel(const pd_attn_block_args_k p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int BM = 64, HD = 64, HEADS = C / HD;
  constexpr int RPC = RPC_, CPW = 64 / RPC, QPC = RPC / 16;   // rows per cuboid (16 KT, or 64 for 3 key tiles: 33-48 slots), cuboids per
                                                              // workgroup, query tiles per cuboid
  constexpr int KS = C / 64;                       // 64-wide K slabs of the A tile
  constexpr int KH = C / 128;                      // K halves of a head's Wq / Wk / Wv slice: chunks [64 d][128 k]
  constexpr int OH = C / 128;                      // output halves of a head's Wp slice: chunks [128 out][64 k]
  constexpr int NSTEP = 4;                         // k-steps of 32 in a q/k/v chunk
  constexpr int CHUNK = 16384;                     // bytes of every weight chunk = two [64 rows][64 k] bf16 slabs
  constexpr int A_BYTES = BM * C * 2;
  static_assert(A_BYTES <= 2 * CHUNK, "the A region becomes ring slots 0 and 1");
  constexpr int TILE = BM * HD * 2;                // 8 KB
  constexpr int NCH = 3 * KH + OH;                 // weight chunks per head
  constexpr int NCHUNK = HEADS * NCH;
  constexpr int NDMA = CHUNK / 8192;               // DMA instructions per chunk (512 threads x 16 B each)
  constexpr int PF = 1;                            // weight-fragment prefetch distance of the q/k/v GEMMs, in k-steps of 32
  extern __shared__ __attribute__((aligned(16))) char smem[];
  static_assert(NS >= 3, "ring: chunk s lives in slot (s + 2) % NS; slots 0 and 1 are the A region");
  char* sA = smem;                                 // A tile, then ring slots 0 and 1
  char* sR2 = sA + 2 * CHUNK;                      // ring slots 2 .. NS-1
  char* sQ = sR2 + (NS - 2) * CHUNK;               // Q tile [row][d], re-used for O
  char* sK = sQ + TILE;                            // K tile [row][d]
  char* sVT = sK + TILE;                           // V^T tile [d][row]
  float* sBias = (float*)(sVT + TILE);             // [HEADS][16][16]
  int* sTok = (int*)(sBias + HEADS * 256);         // [64] global row of every slot, -1 = no token
  float* sBq = (float*)(sTok + BM);                // [3C] qkv bias (zeros without one): an ordinary global load inside the head loop
                                                   // would make hipcc drain the weight DMA queue (vmcnt(0)) at every use

  if (p.dbg & 128) return;   // ablation: workgroup dispatch cost only
  int tr_n = 0;
#define TRACE() do { if (p.trace && blockIdx.x == 600 && threadIdx.x == 0) p.trace[tr_n] = clock64(); ++tr_n; } while (0)
  TRACE();
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int vol = p.vol;

  const auto rWqkv = __builtin_amdgcn_make_buffer_rsrc((void*)p.Wqkv, 0, p.wqkv_bytes, 0x00020000);
  const auto rWp = __builtin_amdgcn_make_buffer_rsrc((void*)p.Wp, 0, p.wp_bytes, 0x00020000);
  const auto rBias = __builtin_amdgcn_make_buffer_rsrc((void*)p.bias, 0, HEADS * p.vol * p.vol * 4, 0x00020000);

  // DMA lane mapping: one 512-thread instruction fills one slab = [64 rows][64 k] (8 KB), lane-linear, source-side swizzle.
  // Every chunk is 2 slabs: ONE per-lane offset serves all chunks, the rest is the instruction's scalar offset.
  const int drow = tid >> 3, dpos = tid & 7;
  const int dchunk = dpos ^ ((drow >> 1) & 7);
  const uint32_t w_voff = ((uint32_t)drow * C + dchunk * 8) * 2u;
  auto ring = [&](int s) {                          // chunk s lives in ring slot (s + 2) % NS (slots are contiguous from sA)
    return sA + ((s + 2) % NS) * CHUNK;
  };
  // chunk s = NCH * head + j:  j < 3 KH: kind j / KH (0 q, 1 k, 2 v), K half j % KH;  else proj output half j - 3 KH
  auto issue_chunk = [&](int s) {
    char* d = ring(s) + wave * 1024;
    const int h = s / NCH, j = s - h * NCH;
    if (j < 3 * KH) {
      const int kind = j / KH, kh = j - kind * KH;
#pragma unroll
      for (int i = 0; i < NDMA; ++i)
        BLDS16(rWqkv, d + i * 8192, w_voff, ((kind * C + h * HD) * C + (2 * kh + i) * 64) * 2);
    } else {
      const int oh = j - 3 * KH;
#pragma unroll
      for (int i = 0; i < NDMA; ++i)
        BLDS16(rWp, d + i * 8192, w_voff, ((oh * 128 + i * 64) * C + h * HD) * 2);
    }
  };
#pragma unroll
  for (int s0 = 0; s0 < NS - 2; ++s0) issue_chunk(s0);        // slots 2 .. NS-1 (slots 0 and 1 are still the A tile)

  // ---- token table of the 4 cuboids, relative-position bias (padded to 16 x 16 per head) ----
  if (tid < BM) {
    const int cl = tid / RPC, slot = tid % RPC;
    const int64_t gc = (int64_t)blockIdx.x * CPW + cl;
    int row = -1;
    if (gc < (int64_t)p.B * p.nc && slot < vol) {
      const int b = (int)(gc / p.nc), c = (int)(gc - (int64_t)b * p.nc);
      const int tok = p.aff_on ? (c / p.aff_ninner) * p.aff_outer + (c % p.aff_ninner) * p.aff_inner + slot * p.aff_slot
                               : p.tok_index[c * vol + slot];
      if (tok >= 0) row = b * p.ntok + tok;
    }
    sTok[tid] = row;
  }
  __syncthreads();
  TRACE();

  // ---- phase 0: LayerNorm of the gathered rows -> bf16 A tile (KS slabs of [64][64], chunk swizzle (row>>1)&7); the qkv-bias and
  //      relative-position-bias tables are built while the row loads are in flight ----
  ln_block_to_tile<C, BM, 8>(p.x, p.gamma, p.beta, p.eps, sA, wave, lane, (p.dbg & 16) != 0, [&](int r) { return sTok[r]; }, [&]() {
    for (int i = tid; i < 3 * C; i += 512) sBq[i] = p.bqkv ? p.bqkv[i] : 0.f;
    if (KT == 1 && !PD_AB_GENERIC_CORE)           // (larger cuboids read their bias rows from L2 inside the core: the table would not fit next to a second workgroup)
      for (int i = tid; i < HEADS * 256; i += 512) {
        const int h = i >> 8, q = (i >> 4) & 15, k = i & 15;
        sBias[i] = (q < vol && k < vol) ? p.bias[((int64_t)h * vol + q) * vol + k] : 0.f;
      }
  });

  TRACE();
  // ---- wave roles ----
  const int lrow = lane & 31, lhalf = lane >> 5;
  const int swz = (lrow >> 1) & 7;
  // q/k/v GEMMs: wave -> d tile tn (of 2) x 16-row tile tq (of 4)
  const int tn = wave & 1, tq = wave >> 1;
  // proj GEMM: wave -> 32-row tile wm (of 2) x 32-column tile wn (of 4) of every 128-column output half
  const int wm = wave >> 2, wn = wave & 3;
  const int g2_a_row = (wm * 32 + lrow) * 128;                       // O row
  const int g2_b_row = (wn >> 1) * 8192 + ((wn & 1) * 32 + lrow) * 128;   // Wp chunk: slab wn >> 1 (64 output channels), row in it

  f32x16 acc2[OH];                                 // [output half]: columns oh*128 + wn*32 + (lane & 31)
#pragma unroll
  for (int t = 0; t < OH; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[t][r] = 0.f;

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();                                        // A tile written by all waves; chunk 0 landed
  // 16x16x32 fragments of the wave's 16 rows: [k-step of 32 (C/32)]; lane = (row l16, 8-element k group lg)
  const int l16 = lane & 15, lg = lane >> 4;
  const int swz16 = (l16 >> 1) & 7;
  op8 areg[KS * 2];
#pragma unroll
  for (int ks = 0; ks < KS * 2; ++ks)
    areg[ks] = *(const op8*)(sA + (ks >> 1) * (BM * 128) + (tq * 16 + l16) * 128 + ((((ks & 1) * 4 + lg) ^ swz16) << 4));
  // validity of this lane's slots: q/k tiles = row tq*16 + l16; v tile = rows tq*16 + 4 lg + (0..3)
  const bool row_ok = sTok[tq * 16 + l16] >= 0;
  uint32_t vrow_ok = 0;
#pragma unroll
  for (int r = 0; r < 4; ++r) vrow_ok |= (sTok[tq * 16 + 4 * lg + r] >= 0 ? 1u : 0u) << r;
  __syncthreads();                                        // A-tile region is now free: ring slots 0 and 1
  issue_chunk(NS - 2);
  issue_chunk(NS - 1);
  TRACE();

  const uint32_t w_lane_off = (uint32_t)((tn * 32 + l16) * 128 + ((lg ^ swz16) << 4));
  const uint32_t q_lds = (uint32_t)(uintptr_t)sQ, k_lds = (uint32_t)(uintptr_t)sK, vt_lds = (uint32_t)(uintptr_t)sVT;
  const uint32_t bq_lds = (uint32_t)(uintptr_t)sBq, bias_lds = (uint32_t)(uintptr_t)sBias;
  const int cub_of_wave = (int)(((int64_t)blockIdx.x * CPW + (wave & 3) / QPC) % p.nc);  // cuboid (mask table row) of this wave's core
  // end of a weight-chunk step: chunk s+1 has landed (chunks s+2 .. s+NS-1 may stay in flight -- only when all of them were issued
  // is the counted wait exact), everyone is done with chunk s, refill its slot
  auto step_end = [&](int s) {
    if (s + NS - 1 < NCHUNK && !(p.dbg & 1)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * NDMA) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (s + NS < NCHUNK && !(p.dbg & 1)) issue_chunk(s + NS);
  };

  for (int h = 0; h < HEADS; ++h) {
    TRACE();
    // (opaque copy of the lane id per head: the per-lane addresses of the tile epilogues and of the core derived from it cannot be
    //  hoisted out of the head loop, where two dozen of them would sit in registers through every phase -- the kernel has to stay
    //  within 128 VGPRs for four waves per SIMD, and a spill reload is a VMEM load that drains the weight DMA queue)
    int lane_h = lane;
    asm volatile("" : "+v"(lane_h));
    const int l16h = lane_h & 15, lgh = lane_h >> 4;
    // ---------------- q and k: transposed products, accumulated over the K halves ----------------
#pragma unroll
    for (int kind = 0; kind < 2; ++kind) {
      // 2 tiles of 16 x 16 (feature tile dt) on the wave's 16 slots: two independent accumulator chains
      f32x4 acc1[2];
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc1[dt][r] = 0.f;
#pragma unroll
      for (int kh = 0; kh < KH; ++kh) {
        const int s = NCH * h + kind * KH + kh;
        if (!(p.dbg & 2)) {
          // Hand-placed fragment pipeline (PF k-steps ahead).  Left to itself hipcc keeps ONE fragment register set and waits
          // lgkmcnt(0) in front of every MFMA pair (it re-serialises a source-level double buffer, too).  Opaque ds_reads +
          // counted waits + sched_barrier pin the order.
          op8 w[PF + 1][2];
          const uint32_t wb = (uint32_t)(uintptr_t)ring(s) + w_lane_off;
          WFRAG_PROLOGUE(w, wb)
#pragma unroll
          for (int ks = 0; ks < NSTEP; ++ks) {
            WFRAG_STEP(w, wb, ks)
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
              acc1[dt] = mfma_16x16x32(w[ks % (PF + 1)][dt], areg[kh * NSTEP + ks], acc1[dt]);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        if (kh + 1 < KH) step_end(s);
      }
      // tile dt: lane = slot row tq*16 + l16, features tn*32 + 16 dt + 4 lg + (0..3)
      const uint32_t t_lds = kind == 0 ? q_lds : k_lds;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const int d = tn * 32 + dt * 16 + 4 * lgh;
        f32x4 bb;   // opaque LDS read (+ its wait): see the note at the ds_write below
        asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(bb) : "v"(bq_lds + (uint32_t)((kind * C + h * HD + d) * 4)) : "memory");
        {
          const int trow = tq * 16 + l16h;
          float v0 = acc1[dt][0] + bb[0], v1 = acc1[dt][1] + bb[1], v2 = acc1[dt][2] + bb[2], v3 = acc1[dt][3] + bb[3];
          if (!row_ok) v0 = v1 = v2 = v3 = 0.f;                 // a padded slot is a zero token (as in the un-fused path)
          const uint64_t pk = (uint64_t)(pack_op2(v0, v1)) | ((uint64_t)(pack_op2(v2, v3)) << 32);
          const int off = trow * 128 + (((d >> 3) ^ ((trow >> 1) & 7)) << 4) + ((d & 7) << 1);
          // opaque ds_write: for a visible LDS store hipcc first drains the in-flight weight DMA (it cannot tell that the tiles and
          // the DMA destinations are disjoint LDS regions), which would serialise the prefetch at every step
          asm volatile("ds_write_b64 %0, %1" ::"v"(t_lds + (uint32_t)off), "v"(pk) : "memory");
        }
      }
      step_end(NCH * h + kind * KH + KH - 1);
    }
    // ---------------- v: plain product, stored transposed ----------------
    TRACE();
    {
      f32x4 acc1[2];                                           // [feature tile dt]
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc1[dt][r] = 0.f;
#pragma unroll
      for (int kh = 0; kh < KH; ++kh) {
        const int s = NCH * h + 2 * KH + kh;
        if (!(p.dbg & 2)) {
          op8 w[PF + 1][2];
          const uint32_t wb = (uint32_t)(uintptr_t)ring(s) + w_lane_off;
          WFRAG_PROLOGUE(w, wb)
#pragma unroll
          for (int ks = 0; ks < NSTEP; ++ks) {
            WFRAG_STEP(w, wb, ks)
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
              acc1[dt] = mfma_16x16x32(areg[kh * NSTEP + ks], w[ks % (PF + 1)][dt], acc1[dt]);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        if (kh + 1 < KH) step_end(s);
      }
      // tile dt: lane = feature d = tn*32 + 16 dt + l16, slot rows tq*16 + 4 lg + (0..3)
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const int d = tn * 32 + dt * 16 + l16h;
        float bv;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(bv) : "v"(bq_lds + (uint32_t)((2 * C + h * HD + d) * 4)) : "memory");
        {
          float v0 = acc1[dt][0] + bv, v1 = acc1[dt][1] + bv, v2 = acc1[dt][2] + bv, v3 = acc1[dt][3] + bv;
          if (!(vrow_ok & 1u)) v0 = 0.f;
          if (!(vrow_ok & 2u)) v1 = 0.f;
          if (!(vrow_ok & 4u)) v2 = 0.f;
          if (!(vrow_ok & 8u)) v3 = 0.f;
          const uint64_t pk = (uint64_t)(pack_op2(v0, v1)) | ((uint64_t)(pack_op2(v2, v3)) << 32);
          const int c4 = tq * 4 + lgh;                                             // 4-row group of the [d][64 rows] tile
          const int off = d * 128 + ((c4 ^ (d & 15)) << 3);                        // group XOR: conflict-free 8 B reads by (d, 4-row group)
          asm volatile("ds_write_b64 %0, %1" ::"v"(vt_lds + (uint32_t)off), "v"(pk) : "memory");
        }
      }
      if constexpr (KT == 1 && !PD_AB_GENERIC_CORE) step_end(NCH * h + 3 * KH - 1);
    }
    // larger cuboids: the core waves fetch their bias entries (query qi of the cuboid, keys kt*16 + 4g .. +3) from L2 BEFORE the
    // next weight chunk is issued, so that the counted vmcnt in front of the core covers them without draining that chunk
    float bkg[KT > 1 ? KT : 1][4];
    if constexpr (KT > 1 || PD_AB_GENERIC_CORE) {
      static_assert(KT == 1 || NS == 3, "the deep ring is built for the 16-slot cuboids only");
      const int s = NCH * h + 3 * KH - 1;
      if (s + 2 < NCHUNK && !(p.dbg & 1)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      {
        // one per-lane offset + immediates; keys beyond the table's end read 0 through the descriptor's bounds check
        const int qi = ((wave & 3) % QPC) * 16 + l16h;
        const uint32_t boff = (uint32_t)((((h * vol) + (qi < vol ? qi : 0)) * vol + 4 * lgh) * 4);
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            bkg[kt][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rBias, boff + (uint32_t)((kt * 16 + r) * 4), 0, 0));
      }
      if (s + 3 < NCHUNK && !(p.dbg & 1)) issue_chunk(s + 3);
    }
    TRACE();
    if constexpr (KT > 1 || PD_AB_GENERIC_CORE) {
      // ---------------- attention core, cuboids of up to 16 KT slots: waves w and w + 4 = query tile w of the workgroup's 64 rows;
      //                  both form the probabilities, each then takes half of the head's 64 output features ----------------
      if (!(p.dbg & 4)) {
        const int q = l16h, g = lgh;
        const int wq = wave & 3, dhalf = wave >> 2;
        const int row = wq * 16 + q;                      // this lane's query row
        const int cbase = (wq / QPC) * RPC;               // first row of its cuboid
        const int qi = (wq % QPC) * 16 + q;               // query index inside the cuboid
        const int rswz = (row >> 1) & 7;
        op8 qf[2];
#pragma unroll
        for (int st = 0; st < 2; ++st)
          asm volatile("ds_read_b128 %0, %1" : "=v"(qf[st]) : "v"(q_lds + (uint32_t)(row * 128 + (((g + 4 * st) ^ rswz) << 4))));
        // S^T[key][query] per key tile, turned into this lane's logits in place (query qi, keys kt*16 + 4g .. +3); masked entries are
        // remembered as bits so that nothing but the 4 KT logits stays live (the core runs inside the kernel's 128-VGPR budget)
        f32x4 sc4[KT];
        uint32_t masked = 0;
        float mx = -3.0e38f;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
          const int krow = cbase + kt * 16 + q;
          const int kswz = (krow >> 1) & 7;
          op8 kf[2];
#pragma unroll
          for (int st = 0; st < 2; ++st)
            asm volatile("ds_read_b128 %0, %1" : "=v"(kf[st]) : "v"(k_lds + (uint32_t)(krow * 128 + (((g + 4 * st) ^ kswz) << 4))));
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
          const f32x4 z = {0.f, 0.f, 0.f, 0.f};
          f32x4 t = mfma_16x16x32(kf[0], qf[0], z);
          t = mfma_16x16x32(kf[1], qf[1], t);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int key = kt * 16 + 4 * g + r;
            float v = -INFINITY;
            if (key < vol && qi < vol) {
              v = t[r] * p.scale + bkg[kt][r];
              if (p.mask && !p.mask[((int64_t)cub_of_wave * vol + qi) * vol + key]) { v = -1e18f; masked |= 1u << (kt * 4 + r); }
            }
            sc4[kt][r] = v;
            mx = fmaxf(mx, v);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        mx = rows4_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float e = expf(sc4[kt][r] - mx);   // exp(-inf) = 0 for non-existent keys
            sum += e;
            sc4[kt][r] = e;
          }
        sum = rows4_sum(sum);
        const float inv = sum > 0.f ? 1.f / sum : 0.f;
        s16x4 pf[KT];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float v = sc4[kt][r] * inv;
            if (masked & (1u << (kt * 4 + r))) v = 0.f;       // masked_softmax multiplies by the mask after the softmax
            pf[kt][r] = (short)f2op(v);
          }
        // O^T[d][query] = sum over the key tiles of V^T (lane: d = 16 i + q, keys of 4-row group cbase/4 + 4 kt + g) x P^T
        f32x4 o[2];
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
          const int d = 16 * (2 * dhalf + ii) + q;
          s16x4 vf[KT];
#pragma unroll
          for (int kt = 0; kt < KT; ++kt)
            asm volatile("ds_read_b64 %0, %1" : "=v"(vf[kt]) : "v"(vt_lds + (uint32_t)(d * 128 + ((((cbase >> 2) + 4 * kt + g) ^ (d & 15)) << 3))));
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
          o[ii] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int kt = 0; kt < KT; ++kt) o[ii] = mfma_16x16x16(vf[kt], pf[kt], o[ii]);
        }
        // the waves of a pair read the same Q rows above and overwrite them with O below: everyone is past its Q reads first
        asm volatile("s_nop 15" : "+v"(o[0][0]), "+v"(o[0][1]), "+v"(o[0][2]), "+v"(o[0][3]), "+v"(o[1][0]), "+v"(o[1][1]), "+v"(o[1][2]), "+v"(o[1][3]));
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
          const int dd = 16 * (2 * dhalf + ii) + 4 * g;
          const uint64_t pk = (uint64_t)(pack_op2(o[ii][0], o[ii][1])) | ((uint64_t)(pack_op2(o[ii][2], o[ii][3])) << 32);
          const int off = row * 128 + (((dd >> 3) ^ rswz) << 4) + ((dd & 7) << 1);
          asm volatile("ds_write_b64 %0, %1" ::"v"(q_lds + (uint32_t)off), "v"(pk) : "memory");
        }
      } else {
        __builtin_amdgcn_s_barrier();     // (ablation flag: keep the barrier count of the core)
      }
    }
    // ---------------- attention core: waves 0-3, wave w = cuboid w of this workgroup ----------------
    if (KT == 1 && !PD_AB_GENERIC_CORE && wave < 4 && !(p.dbg & 4)) {
      const int q = l16h, g = lgh;
      const int row = wave * 16 + q;
      const int rswz = (row >> 1) & 7;
      // every fragment of the core in flight at once, through opaque reads (a visible LDS load makes hipcc drain the weight DMA
      // queue first -- vmcnt(0) -- because it cannot tell the tiles from the DMA destinations); consumed behind counted waits
      op8 kf[2], qf[2];
      f32x4 bq4;
      s16x4 vf[HD / 16];
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        const uint32_t pos = (uint32_t)(row * 128 + (((g + 4 * st) ^ rswz) << 4));
        asm volatile("ds_read_b128 %0, %1" : "=v"(kf[st]) : "v"(k_lds + pos));
        asm volatile("ds_read_b128 %0, %1" : "=v"(qf[st]) : "v"(q_lds + pos));
      }
      asm volatile("ds_read_b128 %0, %1" : "=v"(bq4) : "v"(bias_lds + (uint32_t)((h * 256 + q * 16 + 4 * g) * 4)));
      f32x4 sc4 = {0.f, 0.f, 0.f, 0.f};
      asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      sc4 = mfma_16x16x32(kf[0], qf[0], sc4);
      asm volatile("s_waitcnt lgkmcnt(1)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      sc4 = mfma_16x16x32(kf[1], qf[1], sc4);
      __builtin_amdgcn_sched_barrier(0);
      // V^T fragments (lane: d = 16 i + q, keys 4g..4g+3) do not depend on the softmax: requested once the K / Q fragments are
      // dead, their latency hides behind the softmax
#pragma unroll
      for (int i = 0; i < HD / 16; ++i) {
        const int d = 16 * i + q;
        asm volatile("ds_read_b64 %0, %1" : "=v"(vf[i]) : "v"(vt_lds + (uint32_t)(d * 128 + (((wave * 4 + g) ^ (d & 15)) << 3))));
      }
      asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(HD / 16) : "memory");      // the bias row
      __builtin_amdgcn_sched_barrier(0);
      // lane: query q, keys 4g .. 4g+3
      const float bk[4] = {bq4[0], bq4[1], bq4[2], bq4[3]};
      float sc[4];
      float mx = -3.0e38f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = 4 * g + r;
        float v = -INFINITY;
        if (key < vol && q < vol) {
          v = sc4[r] * p.scale + bk[r];
          if (p.mask && !p.mask[((int64_t)cub_of_wave * vol + q) * vol + key]) v = -1e18f;
        }
        sc[r] = v;
        mx = fmaxf(mx, v);
      }
      // reductions over the four 16-lane rows (the key groups of one query): gfx950 row / half swaps instead of two ds_bpermute
      // round trips each
      mx = rows4_max(mx);
      float pr[4], sum = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        pr[r] = expf(sc[r] - mx);   // exp(-inf) = 0 for non-existent keys
        sum += pr[r];
      }
      sum = rows4_sum(sum);
      const float inv = sum > 0.f ? 1.f / sum : 0.f;
      s16x4 pf;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = pr[r] * inv;
        if (sc[r] <= -1e18f) v = 0.f;   // masked_softmax multiplies by the mask after the softmax
        pf[r] = (short)f2op(v);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                      // the V^T fragments
      __builtin_amdgcn_sched_barrier(0);
      // O^T[d][query] = sum_key V[key][d] P[query][key]: A = V^T (lane: d = d0 + q, keys 4g..4g+3), B = P^T.
      // The four MFMAs are independent: issue them back to back, wait once, then pack and store (one latency chain, not four).
      f32x4 o[HD / 16];
#pragma unroll
      for (int i = 0; i < HD / 16; ++i) {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        o[i] = mfma_16x16x16(vf[i], pf, z);
      }
      // the MFMA results go straight into inline asm (v_cvt_pk_bf16_f32): hipcc's hazard recogniser does not look inside asm, so
      // the XDL-write -> VALU-read wait states have to be spelled out (the "+v" ties keep this between the MFMAs and the packs)
      asm volatile("s_nop 15" : "+v"(o[0][0]), "+v"(o[0][1]), "+v"(o[0][2]), "+v"(o[0][3]), "+v"(o[1][0]), "+v"(o[1][1]), "+v"(o[1][2]),
                   "+v"(o[1][3]), "+v"(o[2][0]), "+v"(o[2][1]), "+v"(o[2][2]), "+v"(o[2][3]), "+v"(o[3][0]), "+v"(o[3][1]), "+v"(o[3][2]),
                   "+v"(o[3][3]));
#pragma unroll
      for (int i = 0; i < HD / 16; ++i) {
        // lane: query q, features 16 i + 4g + (0..3) -> O tile (the Q tile: these 16 rows belong to this wave only)
        const int dd = 16 * i + 4 * g;
        const uint64_t pk = (uint64_t)(pack_op2(o[i][0], o[i][1])) | ((uint64_t)(pack_op2(o[i][2], o[i][3])) << 32);
        const int off = row * 128 + (((dd >> 3) ^ rswz) << 4) + ((dd & 7) << 1);
        asm volatile("ds_write_b64 %0, %1" ::"v"(q_lds + (uint32_t)off), "v"(pk) : "memory");
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                           // O_h visible to every wave (weight DMA stays in flight)
    TRACE();
    // ---------------- proj: acc2 += O_h * Wp_h^T, one 128-column output half per chunk ----------------
#pragma unroll
    for (int oh = 0; oh < OH; ++oh) {
      const int s = NCH * h + 3 * KH + oh;
      const char* cW = ring(s);
      if (!(p.dbg & 8)) {
        op8 fa[2], fb[2];                       // two-deep fragment pipeline
        // (row / swizzle terms from the head's opaque lane id: derived from `lane` they are loop invariant, hipcc parks the eight
        //  fragment addresses in registers across the whole head loop and -- at KT >= 2, where the core needs the registers -- spills
        //  them: reloads are VMEM loads in front of which the weight DMA queue drains)
        const int lrow_h = lane_h & 31, lhalf_h = lane_h >> 5, swz_h = (lrow_h >> 1) & 7;
        const int a_row_h = (wm * 32 + lrow_h) * 128, b_row_h = (wn >> 1) * 8192 + ((wn & 1) * 32 + lrow_h) * 128;
        auto ldp = [&](int kk, int slot) {
          const int pos = ((kk * 2 + lhalf_h) ^ swz_h) * 16;
          fa[slot] = *(const op8*)(sQ + a_row_h + pos);
          fb[slot] = *(const op8*)(cW + b_row_h + pos);
        };
        ldp(0, 0);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          if (kk + 1 < 4) ldp(kk + 1, (kk + 1) & 1);
          acc2[oh] = mfma_32x32x16(fa[kk & 1], fb[kk & 1], acc2[oh]);
        }
      }
      step_end(s);
    }
  }

  TRACE();
  // ---- epilogue: acc2 -> per-wave LDS slab [32][OH * 32] fp32 -> + b_proj + x -> out rows of the token table ----
  constexpr int WN = OH * 32;                      // columns per wave: OH pieces of 32
  constexpr int LPR = WN / 4;                      // lanes per row (float4 each): 16 at C = 256
  constexpr int RPP = 64 / LPR;
  constexpr int NPASS = 32 / RPP;
  const int c0 = (lane % LPR) * 4;                 // slab column; output column = 128 * (c0 / 32) + 32 wn + c0 % 32
  const int n = (c0 >> 5) * 128 + wn * 32 + (c0 & 31);
  float* sC = (float*)smem + wave * (32 * WN);     // (the last step_end left every wave past its weight / tile reads)
#pragma unroll
  for (int t = 0; t < OH; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) sC[((r & 3) + 8 * (r >> 2) + 4 * lhalf) * WN + t * 32 + lrow] = acc2[t][r];
  // every residual row of this lane in flight at once (the accumulators are dead now): one exposed HBM round trip
  int mrow[NPASS];
  float4 xr[NPASS];
#pragma unroll
  for (int u = 0; u < NPASS; ++u) {
    mrow[u] = sTok[wm * 32 + u * RPP + lane / LPR];
    xr[u] = make_float4(0, 0, 0, 0);
    if (mrow[u] >= 0 && !(p.dbg & 32)) xr[u] = *(const float4*)(p.x + (int64_t)mrow[u] * C + n);
  }
  float4 bias = make_float4(0, 0, 0, 0);
  if (p.bp) bias = *(const float4*)(p.bp + n);
  // (no workgroup barrier: the slab is private to the wave)
#pragma unroll
  for (int u = 0; u < NPASS; ++u) {
    if (mrow[u] < 0 || (p.dbg & 64)) continue;
    const float4 a4 = *(const float4*)(sC + (u * RPP + lane / LPR) * WN + c0);
    *(float4*)(p.out + (int64_t)mrow[u] * C + n) =
        make_float4(a4.x + bias.x + xr[u].x, a4.y + bias.y + xr[u].y, a4.z + bias.z + xr[u].z, a4.w + bias.w + xr[u].w);
  }
  TRACE();
#undef TRACE
#endif
}

template <int C, int KT = 1, int RPC = 16 * KT, int NS = 3>
static int launch_attn_block(const pd_attn_block_args_k& a, hipStream_t s) {
  constexpr int heads = C / 64;
  constexpr int work = NS * 16384 + 3 * 64 * 64 * 2;                // weight ring + Q, K, V^T tiles: re-used by the epilogue slab
  constexpr int lds = work + heads * 256 * 4 + 64 * 4 + 3 * C * 4;
  constexpr int epi = 8 * 32 * ((C / 128) * 32) * 4;
  static_assert(epi <= work, "the epilogue slab must not reach the bias / token tables");
  static_assert((NS == 3 ? 2 : 1) * lds <= 160 * 1024, "two workgroups per CU (one with the deep ring)");
  static bool attr_set_dev[PD_MAX_DEVICES];
  bool& attr_set = attr_set_dev[pd_cur_device()];
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)attn_block_kernel<C, KT, RPC, NS>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) {
      pd_set_error("pd_attn_block_fused: hipFuncSetAttribute(%d) failed: %s", lds, hipGetErrorString(e));
      return PD_ERR_LAUNCH;
    }
    attr_set = true;
  }
  const int64_t cuboids = (int64_t)a.B * a.nc;
  constexpr int CPW = 64 / RPC;
  hipLaunchKernelGGL((attn_block_kernel<C, KT, RPC, NS>), dim3((unsigned)((cuboids + CPW - 1) / CPW)), dim3(512), lds, s, a);
  PD_CHECK_LAUNCH();
  return PD_OK;
}

#if !PD_IS_F16
extern "C" int pd_attn_block_fused_supported(int C, int heads, int vol) {
  return (C == 256 || C == 128) && heads * 64 == C && vol >= 1 && vol <= 64;
}
extern "C" int pd_f16_attn_block_fused_ex(const float*, float*, const float*, const float*, const pd_bf16*, const float*, const pd_bf16*, const float*,
                                          const int32_t*, const float*, const uint8_t*, int, int, int, int, int, int, float, float, const int32_t*,
                                          const pd_call_opts*, pd_stream_t);
#else
extern "C" int pd_attn_block_fused_supported(int C, int heads, int vol);
#endif

// opts->attn_block_table_ids (A/B and tests): load the token table even when the affine form of the cuboid table is given (arithmetic
// token ids are the default: neutral in time, one dependent load less).  The atomic in-place epilogue and the deep weight ring for small
// grids of round 3 were measured slower / neutral (profiles/r03_b_fused_opts_ab.log: 122 -> 159 us, 25.6 -> 27.3 us) and are gone.
extern "C" int PD_ENTRY(attn_block_fused_ex)(const float* x, float* out, const float* gamma, const float* beta, const pd_bf16* Wqkv,
                                             const float* bqkv, const pd_bf16* Wp, const float* bp, const int32_t* tok_index, const float* bias,
                                             const uint8_t* mask, int B, int ntok, int C, int heads, int nc, int vol, float scale, float eps,
                                             const int32_t* tok_affine, const pd_call_opts* opts, pd_stream_t stream) {
  PD_FORWARD_F16(PD_OPTS_F16(opts), pd_f16_attn_block_fused_ex(x, out, gamma, beta, Wqkv, bqkv, Wp, bp, tok_index, bias, mask, B, ntok, C, heads, nc,
                                                               vol, scale, eps, tok_affine, opts, stream));
  PD_CHECK_ARG(x && out && gamma && beta && Wqkv && Wp && tok_index && bias, "pd_attn_block_fused: null pointer");
  PD_CHECK_ARG(pd_attn_block_fused_supported(C, heads, vol), "pd_attn_block_fused: unsupported units=%d heads=%d cuboid volume=%d "
               "(units in {128,256}, head_dim 64, volume <= 64)", C, heads, vol);
  PD_CHECK_ARG(B > 0 && ntok > 0 && nc > 0 && (int64_t)B * ntok < (1ll << 31), "pd_attn_block_fused: bad sizes");
  pd_attn_block_args_k a;
  a.x = x; a.out = out; a.gamma = gamma; a.beta = beta; a.Wqkv = Wqkv; a.bqkv = bqkv; a.Wp = Wp; a.bp = bp;
  a.tok_index = tok_index; a.bias = bias; a.mask = mask;
  a.B = B; a.ntok = ntok; a.nc = nc; a.vol = vol; a.scale = scale; a.eps = eps;
  a.wqkv_bytes = (uint32_t)((int64_t)3 * C * C * 2);
  a.wp_bytes = (uint32_t)((int64_t)C * C * 2);
  a.dbg = opts ? opts->attn_block_debug_flags : 0;     // (profiling ablations / per-phase clock stamps: scripts/bench_attn_block.py)
  a.trace = opts ? opts->trace : nullptr;
  a.aff_on = (tok_affine && !(opts && opts->attn_block_table_ids) && tok_affine[0] > 0) ? 1 : 0;
  a.aff_ninner = a.aff_on ? tok_affine[0] : 1;
  a.aff_outer = a.aff_on ? tok_affine[1] : 0;
  a.aff_inner = a.aff_on ? tok_affine[2] : 0;
  a.aff_slot = a.aff_on ? tok_affine[3] : 0;
  hipStream_t s = (hipStream_t)stream;
  const int kt = (vol + 15) / 16;     // key tiles of 16 per cuboid; 3 and 4 tiles: one cuboid per 64-row workgroup
  if (C == 256) {
    if (kt == 1) return launch_attn_block<256, 1>(a, s);
    if (kt == 2) return launch_attn_block<256, 2>(a, s);
    return kt == 3 ? launch_attn_block<256, 3, 64>(a, s) : launch_attn_block<256, 4>(a, s);
  }
  if (kt == 1) return launch_attn_block<128, 1>(a, s);
  if (kt == 2) return launch_attn_block<128, 2>(a, s);
  return kt == 3 ? launch_attn_block<128, 3, 64>(a, s) : launch_attn_block<128, 4>(a, s);
}

#if !PD_IS_F16
extern "C" int pd_attn_block_fused(const float* x, float* out, const float* gamma, const float* beta, const pd_bf16* Wqkv,
                                   const float* bqkv, const pd_bf16* Wp, const float* bp, const int32_t* tok_index, const float* bias,
                                   const uint8_t* mask, int B, int ntok, int C, int heads, int nc, int vol, float scale, float eps,
                                   pd_stream_t stream) {
  return pd_attn_block_fused_ex(x, out, gamma, beta, Wqkv, bqkv, Wp, bp, tok_index, bias, mask, B, ntok, C, heads, nc, vol, scale, eps,
                                nullptr, nullptr, stream);
}
#endif

}  // namespace PD_NS
