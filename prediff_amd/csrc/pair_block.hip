// pd_attn_ffn_pair:  x += proj(cuboid_attention(qkv(LayerNorm(x))));  x += W2 gelu(W1 LayerNorm(x) + b1) + b2   in ONE kernel
// for the level-0 blocks of the SEVIR-LR denoiser (units 256, 4 heads of 64, hidden 1024, cuboid volume <= 16):
// one (CuboidSelfAttentionLayer, PositionwiseFFN) pair of StackCuboidSelfAttentionBlock.forward -- reference
// cuboid_transformer.py:812-966 (attention), :182-208 (FFN), :1147-1156 (the pair and its residuals).
//
// Why a new structure (round 4).  The round-1..3 kernels (attn_block.hip, ffn.hip) keep the activations of 16 rows per wave and
// stream the weights through LDS once per 64 rows; q, k, v, P, O and the FFN hidden tile make LDS round trips between the GEMMs,
// every weight chunk (8 MFMAs per wave) ends in a workgroup barrier, and every MFMA needs its own ds_read_b128: 12.6 non-MFMA
// instructions per MFMA, MFMA pipe 19 % busy (profiles/r03_e_pmc_fused.log).  Here:
//   * a wave owns 32 rows = TWO whole cuboids and EVERYTHING about them stays in its registers for the whole pair (one wave per
//     SIMD, up to 512 VGPRs): the fp32 residual rows x[32][256] are at the same time the LayerNorm input, the accumulator of the proj
//     GEMM and of FFN-2 and the output -- x is read ONCE and written ONCE, nothing else touches HBM;
//   * every GEMM is a TRANSPOSED product D[feature][token] = W[feature][k] * act[token][k]: the MFMA C layout then has lane =
//     (token, 4 consecutive features), which after bf16 packing IS the B-operand layout of the next GEMM provided the k index of the
//     next weight fragment is permuted accordingly -- a dot product does not care in which order it sums.  The permutation is done
//     once, at weight-pack time (packing.pack_pair_block).  So LayerNorm -> qkv, q/k -> S, P/V -> O, O -> proj, LayerNorm -> FFN-1,
//     gelu -> FFN-2 are all register-to-register: NO activation ever goes through LDS, there is no transposition and no barrier
//     between the stages;
//   * LDS holds nothing but the weight stream: 32 KB chunks, pre-packed in MFMA-fragment order (a fragment = 1 KB, lane-linear:
//     the DMA is a plain copy, the ds_read_b128 of a fragment is conflict free with ONE address register and an immediate), 4-slot
//     ring, ONE barrier per chunk (= per 64 MFMAs of a wave) placed in the middle of the previous chunk so that the fragment
//     pipeline never drains; every fragment feeds two MFMAs (the two cuboids): half the LDS bytes per MFMA of the old kernels;
//   * persistent workgroups: the weight stream runs on across tiles, tile t+1's first chunks are in LDS before tile t ends.
// 256 threads = 4 waves; tile = 128 rows = 8 cuboids; LDS 16 KB of tables + 128 KB ring.
// Numerics: those of the bf16 engine (bf16 LayerNorm output, q, k, v, P, O, hidden; fp32 accumulation, softmax, residual).  GELU is
// evaluated as x * sigmoid(x (a + b x^2 + c x^4)) (max abs deviation from the erf form 2.5e-5, below the bf16 rounding of the hidden
// activations that follows; 9 VALU instructions instead of 16).
#include <algorithm>
#include <type_traits>
#include "common.h"

#define BLDS16(rsrc, ldsptr, voff, soff) \
  __builtin_amdgcn_raw_ptr_buffer_load_lds((rsrc), (__attribute__((address_space(3))) void*)(ldsptr), 16, (voff), (soff), 0, 0)

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

namespace pairk {
constexpr int C = 256, HEADS = 4, HID = 1024;
#ifndef PD_PAIR_PF
#define PD_PAIR_PF 4
#endif
#ifndef PD_PAIR_DMA_SPREAD
#define PD_PAIR_DMA_SPREAD 0
#endif
#ifndef PD_PAIR_DMA_EARLY
#define PD_PAIR_DMA_EARLY 0          // 1: a second barrier at the START of a chunk, behind which chunk c + 3 is requested (three chunks of lead
#endif                              //    instead of two and a half); the landing check of chunk c + 1 stays at the middle.  Measured SLOWER
                                    //    (259-265 vs 250-258 us at 32 trajectories, 53 vs 50 us at 4): the stream is not DMA-latency bound
constexpr int PF = PD_PAIR_PF, PFN = 8;                     // weight fragments in flight (LDS latency ~ 4 x 32 MFMA clocks) / register slots: the slot of
                                                   // fragment i is i % PFN in EVERY chunk, so PFN must divide the 32 fragments of a chunk
constexpr int CHUNK = 32768, NSLOT = 4, NFRAG = 32;
static_assert(NFRAG % PFN == 0 && PF + 2 <= PFN && PF % 2 == 0 && PF + 4 <= 15, "fragment pipeline geometry");
constexpr int DMA_PER_WAVE = CHUNK / 4 / 1024;     // 8 x 1 KB per wave per chunk
// fp32 tables in LDS (float offsets) == layout of the `vecs` argument
constexpr int T_LN1G = 0, T_LN1B = 256, T_BP = 512, T_LN2G = 768, T_LN2B = 1024, T_B2 = 1280, T_B1 = 1536, T_RB = 2560, T_FLOATS = 3584;
constexpr int RING_OFF = 16384;
constexpr int LDS_BYTES = RING_OFF + NSLOT * CHUNK;   // 147456
constexpr int CH_ALL = 48;                         // chunks per tile: 4 heads x (q, k, v, proj) + 16 x (W1_j, W2_j)
}  // namespace pairk

struct pd_pair_args_k {
  const float* x;
  float* out;
  const void* wstream;        // CH_ALL chunks of 32 KB, packing.pack_pair_block
  const float* vecs;          // T_FLOATS floats
  const int32_t* tok_index;   // [nc][vol] or null with aff_on
  int B, ntok, nc, vol;
  float scale, eps1, eps2;
  int aff_on, aff_ninner, aff_outer, aff_inner, aff_slot;
  int ntiles;
  uint32_t wbytes, xbytes;
  unsigned long long* trace;
  float* dbg_buf;             // PD_PAIR_DEBUG builds only: [rows][256] dump of one intermediate of head 0 (dbg_stage)
  int dbg_stage;
};

__device__ __forceinline__ float pk_rows4_max(float v) {
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float pk_rows4_sum(float v) {
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
// fp32 -> bf16 (RNE) through the compiler's own v_cvt_pk_bf16_f32 (NOT common.h's inline-asm form: hipcc pads no hazard wait states
// around an asm statement, and here the conversions sit directly between MFMAs -- an asm v_cvt reading a fresh MFMA result, or an MFMA
// reading a fresh asm v_cvt result, gets stale registers; measured: wrong O tiles / NaNs in the first build of this kernel)
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
__device__ __forceinline__ bf16x8 pk_pack8(const f32x4& a, const f32x4& b) {
  const bf16x4 lo = __builtin_convertvector(a, bf16x4), hi = __builtin_convertvector(b, bf16x4);
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
__device__ __forceinline__ s16x4 pk_pack4(const f32x4& a) {
  return __builtin_bit_cast(s16x4, __builtin_convertvector(a, bf16x4));
}
// GELU: common.h gelu_sigmoid / gelu_sigmoid_arg (x * sigmoid(x (a + b x^2 + c x^4)), 2.5e-5 from the erf form)
#define pk_gelu_arg gelu_sigmoid_arg

#define PK_WLD(dst, addr, imm) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(imm))
// wait until at most N LDS operations are outstanding; the fragment about to be consumed is tied to the wait (in / out operand), so
// that no MFMA reading it can be moved in front of the wait
#define PK_WAIT_FRAG(frag, N) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(frag) : "n"(N))
#define PK_LDS_F4(dst, addr, imm) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(imm))
#define PK_LANDED(v) asm volatile("" : "+v"(v))
#define PK_DRAIN()                                                                                                        \
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4]), "+v"(w[5]), "+v"(w[6]), \
               "+v"(w[7]))

#ifndef PD_PAIR_DEBUG
#define PD_PAIR_DEBUG 0
#endif
// compile-time ablations for profiling builds (-DPD_PAIR_ABLATE=bits, scripts/ablate_pair.sh): 1 no weight DMA after the prologue,
// 2 no fragment reads, 4 no MFMAs in the chunk bodies, 8 no GELU work in the chunk hooks, 16 no row loads / stores in the chunk hooks
#ifndef PD_PAIR_ABLATE
#define PD_PAIR_ABLATE 0
#endif
#ifndef PD_PAIR_GELU_BOTH
#define PD_PAIR_GELU_BOTH 1           // 1: gelu(h_j) split over the W2 and the W1 chunk between W1_j and W2_j; 0: all of it beside the W1 chunk
                                      // (A/B on MI355X, 32 trajectories, two rounds: 258-273 us either way -- the FFN iteration's time is a
                                      // property of the chunk pair, the four waves meet at every chunk's barrier)
#endif

template <int NC>
__global__ void __launch_bounds__(256, 1) pair_kernel(const pd_pair_args_k p) {
#if defined(__HIP_DEVICE_COMPILE__)
  using namespace pairk;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane & 15, g = lane >> 4;

  // ---- tables -> LDS (before any DMA is in flight: a compiler-visible LDS store behind a DMA would drain it) ----
  for (int i = tid; i < T_FLOATS; i += 256) ((float*)smem)[i] = p.vecs[i];
  __syncthreads();

  // ---- weight stream: chunk ids 0 .. CH_ALL-1 cyclically, chunk number n -> ring slot n & 3 ----
  const auto rW = __builtin_amdgcn_make_buffer_rsrc((void*)p.wstream, 0, p.wbytes, 0x00020000);
  // Every wave copies a quarter of every chunk, as 8 pieces of 1 KB (one DMA instruction each); piece k of a wave belongs to chunk
  // k / 8.  Default (PD_PAIR_DMA_SPREAD 0): the prologue issues chunks 0..2, then the 8 pieces of chunk c + 3 go out in one burst
  // behind the mid-chunk barrier of chunk c (everybody is past chunk c - 1, whose slot this is).  PD_PAIR_DMA_SPREAD 1 issues ONE
  // piece every second fragment group instead (prologue: chunks 0, 1 and half of 2; groups 0..6 of chunk c: second half of chunk c + 2,
  // groups 8..14: first half of chunk c + 3), the "even interleave" of the 256^2 GEMM kernel -- measured here 3 % SLOWER (275-279 vs
  // 264-270 us at 32 trajectories, two rounds): with one wave per SIMD every piece costs the wave its own issue slots either way.
  int n_piece = 0, kid = 0;                        // pieces issued by this wave; stream id of the chunk the next piece belongs to
  const uint32_t dma_voff = (uint32_t)lane * 16u;
  auto issue_piece = [&]() {
#if PD_PAIR_ABLATE & 1
    if (n_piece >= (PD_PAIR_DMA_SPREAD ? 20 : 24)) { ++n_piece; return; }
#endif
    const int sub = n_piece & (DMA_PER_WAVE - 1);
    char* d = smem + RING_OFF + ((n_piece >> 3) & (NSLOT - 1)) * CHUNK + wave * (DMA_PER_WAVE * 1024) + sub * 1024;
    const uint32_t so = (uint32_t)kid * CHUNK + (uint32_t)wave * (DMA_PER_WAVE * 1024) + (uint32_t)sub * 1024u;
    BLDS16(rW, d, dma_voff, so);
    ++n_piece;
    if (sub == DMA_PER_WAVE - 1) kid = (kid + 1 == CH_ALL) ? 0 : kid + 1;
  };
#pragma unroll
  for (int i = 0; i < (PD_PAIR_DMA_SPREAD ? 20 : 24); ++i) issue_piece();

  const uint32_t vbase = (uint32_t)(uintptr_t)(smem + RING_OFF) + (uint32_t)lane * 16u;   // fragment reads: lane-linear 16 B
  const uint32_t vtab = (uint32_t)(uintptr_t)smem + (uint32_t)g * 16u;                     // fp32 tables: 4 floats at column 4 g
  const uint32_t vrb = (uint32_t)(uintptr_t)smem + (uint32_t)((T_RB + q * 16 + 4 * g) * 4);

#if PD_PAIR_DEBUG
  int tr_n = 0;
#define PK_TRACE() do { if (p.trace && blockIdx.x == 7 && tid == 0 && tr_n < 256) p.trace[tr_n] = __builtin_amdgcn_s_memtime(); ++tr_n; } while (0)
#else
#define PK_TRACE() do {} while (0)
#endif
  int cc = 0;                                       // chunks consumed by this workgroup
  bf16x8 w[PFN] = {};                               // fragment pipeline (runs on across chunks, tiles and phases)

  // One chunk = 32 fragments = 16 groups of two.  Per group: two fragment reads three groups ahead, ONE counted wait, the four MFMAs
  // of the group's two fragments (BODY_STMT, once per fragment: `i`, `wf`) and HOOK_STMT (`gi`): independent work for their shadow
  // (GELU stages, ONE row load or store).
  // In the middle of chunk cc (group 8): chunk cc+1 has landed for everybody and everybody is past chunk cc-1, whose slot takes chunk
  // cc+3.  The last piece of chunk cc+1 was issued just before the previous mid-chunk sync, so VMC = the VMEM instructions issued
  // since then that may stay in flight: 8 DMA pieces (4 of chunk cc+2 in the second half of chunk cc-1, 4 in the first half of
  // chunk cc) plus the row loads / stores the hooks issued in those two half chunks (a fixed schedule: the constants are derived at
  // the tile loop).  Loads and stores retire in order (one vmcnt queue on gfx9-class hardware), and every hook instruction is
  // issued unconditionally.
  // EXTRA_STMT: NEXTRA other LDS reads issued at the start; they have landed at group PF / 2, where LANDED_STMT re-defines their
  // destinations (PK_LANDED).
  // RULE for every asynchronous (inline-asm) LDS read in this kernel: its destination must not live long BEFORE its wait -- the
  // compiler believes the register is valid from the asm statement on and may park a long-lived value in an AGPR at once, i.e. copy
  // it before the data has arrived.  So: fragments are consumed three groups after their read; table values are re-defined
  // (PK_LANDED, no instruction) right after the wait that covers them and only that new value lives on; and wherever more than a few
  // instructions separate two chunk loops the fragments in flight are drained first (PK_DRAIN).  scripts/check_async_lds.py replays
  // the LDS queue over the generated ISA and fails the build if any instruction touches a destination that is still in flight.
#define PK_VMC(BURST, SPREAD) (PD_PAIR_DMA_SPREAD ? (SPREAD) : (BURST))
#define PK_VMC0 (PD_PAIR_DMA_EARLY ? 16 : 8)   /* chunks without hook traffic around them: the DMA pieces younger than chunk c + 1 */
#define PK_RD_ON (!(PD_PAIR_ABLATE & 2))
#define PK_MFMA_ON (!(PD_PAIR_ABLATE & 4))
#define PK_RD(i_)                                                                                 \
  if (PK_RD_ON) {                                                                                 \
    if ((i_) < NFRAG) PK_WLD(w[(i_) % PFN], va_, (i_) * 1024);                                     \
    else PK_WLD(w[(i_) % PFN], vn_, ((i_) - NFRAG) * 1024);                                        \
  }
#define PK_CHUNK(VMC, NEXTRA, EXTRA_STMT, LANDED_STMT, BODY_STMT, HOOK_STMT)                      \
  {                                                                                               \
    const uint32_t va_ = vbase + (uint32_t)(cc & (NSLOT - 1)) * CHUNK;                            \
    const uint32_t vn_ = vbase + (uint32_t)((cc + 1) & (NSLOT - 1)) * CHUNK;                      \
    _Pragma("unroll") for (int gi = 0; gi < NFRAG / 2; ++gi) {                                    \
      if (gi == 0) { EXTRA_STMT; }                                                                \
      if (PD_PAIR_DMA_EARLY && gi == 0) {                                                         \
        asm volatile("s_barrier" ::: "memory");                                                   \
        _Pragma("unroll") for (int i_ = 0; i_ < DMA_PER_WAVE; ++i_) issue_piece();                \
      }                                                                                           \
      if (gi == NFRAG / 4) {                                                                      \
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(VMC) : "memory");                   \
        if (!PD_PAIR_DMA_SPREAD && !PD_PAIR_DMA_EARLY) { _Pragma("unroll") for (int i_ = 0; i_ < DMA_PER_WAVE; ++i_) issue_piece(); } \
      }                                                                                           \
      if (PD_PAIR_DMA_SPREAD && (gi & 1) == 0) issue_piece();                                     \
      PK_RD(2 * gi + PF);                                                                         \
      PK_RD(2 * gi + PF + 1);                                                                     \
      asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(w[(2 * gi) % PFN]), "+v"(w[(2 * gi + 1) % PFN]) \
                   : "n"(PF + (2 * gi < PF ? (NEXTRA) : 0)));                                     \
      if (gi == PF / 2) { LANDED_STMT; }                                                          \
      __builtin_amdgcn_sched_barrier(0);                                                          \
      { const int i = 2 * gi; const bf16x8 wf = w[i % PFN]; BODY_STMT; }                          \
      { const int i = 2 * gi + 1; const bf16x8 wf = w[i % PFN]; BODY_STMT; }                      \
      { HOOK_STMT; }                                                                              \
      __builtin_amdgcn_sched_barrier(0);                                                          \
    }                                                                                             \
    ++cc;                                                                                         \
  }
#define PK_MFMA_T(ACC, AF) /* transposed product on a [64 features x 256 k] chunk: fragment i = 4 ks + dt */ \
  if (PK_MFMA_ON) {                                                                                        \
    _Pragma("unroll") for (int c_ = 0; c_ < NC; ++c_)                                                      \
      ACC[c_][i & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, AF[c_][i >> 2], ACC[c_][i & 3], 0, 0, 0); \
  }
#define PK_MFMA_OUT(OF) /* x^T += W[256 outputs x 64 k] act^T: fragment i = 16 st + nt */                      \
  if (PK_MFMA_ON) {                                                                                            \
    _Pragma("unroll") for (int c_ = 0; c_ < NC; ++c_)                                                          \
      acc[c_][i & 15] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, OF[c_][i >> 4], acc[c_][i & 15], 0, 0, 0); \
  }

  // rows of a tile's cuboids for this lane = (slot q, column group g), as byte offsets into x / out.  An invalid slot gets an offset
  // beyond the buffers: its loads return 0 and its stores are dropped by the descriptor's bounds check -- ALWAYS exactly 32 load and
  // 32 store instructions per tile, no lane ever branches.
  const auto rX = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.xbytes, 0x00020000);
  const auto rO = __builtin_amdgcn_make_buffer_rsrc((void*)p.out, 0, p.xbytes, 0x00020000);
  constexpr uint32_t OOB = 0xFFFFF000u;
  auto tile_rows = [&](int tile, uint32_t (&off)[NC]) {
    // (the lane id is re-derived here, opaquely: q and g kept alive across the whole tile loop were the two registers this kernel
    //  did not have -- they went to scratch, and a scratch reload drains every DMA piece in flight)
    uint32_t l2;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l2));
    const int q = (int)(l2 & 15u), g = (int)(l2 >> 4);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int64_t gc = (int64_t)tile * (4 * NC) + wave * NC + c;
      int row = -1;
      if (gc < (int64_t)p.B * p.nc && q < p.vol) {
        const int b = (int)(gc / p.nc), cu = (int)(gc - (int64_t)b * p.nc);
        const int tok = p.aff_on ? (cu / p.aff_ninner) * p.aff_outer + (cu % p.aff_ninner) * p.aff_inner + q * p.aff_slot
                                 : p.tok_index[cu * p.vol + q];
        if (tok >= 0 && tok < p.ntok) row = b * p.ntok + tok;
      }
      off[c] = row < 0 ? OOB : (uint32_t)row * (uint32_t)(C * 4) + (uint32_t)g * 16u;
    }
  };
#ifndef PD_PAIR_AUX_LD
#define PD_PAIR_AUX_LD 0
#endif
#ifndef PD_PAIR_AUX_ST
#define PD_PAIR_AUX_ST 0
#endif
#define PK_ROW_LD(OFF, NT) __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rX, (OFF) + (uint32_t)((NT) * 64), 0, PD_PAIR_AUX_LD))
#define PK_ROW_ST(V, OFF, NT) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, V), rO, (OFF) + (uint32_t)((NT) * 64), 0, PD_PAIR_AUX_ST)
#define PK_HOOK_IO (!(PD_PAIR_ABLATE & 16))

  // acc: the rows in flight (x -> x + attn -> x + attn + ffn), lane = (token q, columns 16 nt + 4 g .. +3): the MFMA C layout of every
  // transposed product.  xn: the rows of the NEXT tile.  The tile boundary is software pipelined: xn is requested one row-instruction
  // per fragment group during the last two chunks of a tile, the finished rows (acc) leave one row-instruction per group during the
  // first two chunks of the next tile (whose LayerNorm and q / k / v products read xn), and only then acc <- xn + b_proj.
  f32x4 acc[NC][16], xn[NC][16];
  uint32_t roff[NC], noff[NC], ooff[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) roff[c] = ooff[c] = OOB;
  tile_rows(blockIdx.x, noff);
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int nt = 0; nt < 16; ++nt) {
      xn[c][nt] = PK_ROW_LD(noff[c], nt);
      acc[c][nt] = f32x4{0.f, 0.f, 0.f, 0.f};       // (the first tile has no predecessor: its hook stores go to OOB offsets)
    }
  // chunks 0, 1 and half of 2 landed (the row loads above are younger: this wait covers both), visible to every wave; fragment prologue
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int i = 0; i < PF; ++i) PK_WLD(w[i], vbase, i * 1024);
  PK_DRAIN();

  for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
    PK_TRACE();   // tile start
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      ooff[c] = roff[c];
      roff[c] = noff[c];
      noff[c] = OOB;
    }
    if (tile + (int)gridDim.x < p.ntiles) tile_rows(tile + gridDim.x, noff);
#if PD_PAIR_DEBUG
    auto dump4 = [&](int stage, int c, const f32x4& a, const f32x4& b, const f32x4& cc4, const f32x4& d) {
      if (p.dbg_buf && p.dbg_stage == stage && roff[c] != OOB) {
        float* o = p.dbg_buf + roff[c] / 4;
        *(f32x4*)(o) = a; *(f32x4*)(o + 16) = b; *(f32x4*)(o + 32) = cc4; *(f32x4*)(o + 48) = d;
      }
    };
#endif
    bf16x8 af[NC][8];                                // LayerNorm output as B-operand fragments: [cuboid][k-step of 32]
    // LayerNorm over the 256 columns of a row (64 in this lane, the rest in lanes q + 16 g'), -> af
    auto layer_norm = [&](const f32x4 (&src)[NC][16], int t_gamma, int t_beta, float eps) {
      float mean[NC], rstd[NC];
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        float s = 0.f;
#pragma unroll
        for (int nt = 0; nt < 16; ++nt) s += (src[c][nt][0] + src[c][nt][1]) + (src[c][nt][2] + src[c][nt][3]);
        mean[c] = pk_rows4_sum(s) * (1.0f / C);
        float v = 0.f;
#pragma unroll
        for (int nt = 0; nt < 16; ++nt) {
          const float d0 = src[c][nt][0] - mean[c], d1 = src[c][nt][1] - mean[c], d2 = src[c][nt][2] - mean[c], d3 = src[c][nt][3] - mean[c];
          v += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
        }
        rstd[c] = rsqrtf(pk_rows4_sum(v) * (1.0f / C) + eps);
      }
      // gamma / beta of the k-step's two column tiles: opaque reads, one k-step ahead
      f32x4 gb[2][4];
      auto ld_gb = [&](int ks, int slot) {
        PK_LDS_F4(gb[slot][0], vtab, (t_gamma + 32 * ks) * 4);
        PK_LDS_F4(gb[slot][1], vtab, (t_gamma + 32 * ks + 16) * 4);
        PK_LDS_F4(gb[slot][2], vtab, (t_beta + 32 * ks) * 4);
        PK_LDS_F4(gb[slot][3], vtab, (t_beta + 32 * ks + 16) * 4);
      };
      ld_gb(0, 0);
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        if (ks + 1 < 8) {
          ld_gb(ks + 1, (ks + 1) & 1);
          asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(gb[ks & 1][0]), "+v"(gb[ks & 1][1]), "+v"(gb[ks & 1][2]), "+v"(gb[ks & 1][3]));
        } else {
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(gb[ks & 1][0]), "+v"(gb[ks & 1][1]), "+v"(gb[ks & 1][2]), "+v"(gb[ks & 1][3]));
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          f32x4 y[2];
#pragma unroll
          for (int hf = 0; hf < 2; ++hf)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              y[hf][r] = (src[c][2 * ks + hf][r] - mean[c]) * rstd[c] * gb[ks & 1][hf][r] + gb[ks & 1][2 + hf][r];
          af[c][ks] = pk_pack8(y[0], y[1]);
        }
      }
    };
    // acc[c][nt] = src[c][nt] + table[16 nt + 4 g .. +3]  (proj / FFN-2 bias: the accumulator starts from residual + bias)
    auto add_vec = [&](const f32x4 (&src)[NC][16], int t_off) {
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        f32x4 bv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) PK_LDS_F4(bv[i], vtab, (t_off + 16 * (half * 8 + i)) * 4);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bv[0]), "+v"(bv[1]), "+v"(bv[2]), "+v"(bv[3]), "+v"(bv[4]), "+v"(bv[5]), "+v"(bv[6]), "+v"(bv[7]));
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int c = 0; c < NC; ++c) acc[c][half * 8 + i] = src[c][half * 8 + i] + bv[i];
      }
    };

    // ================= attention: x += proj(attn(LN1(x))) =================
    layer_norm(xn, T_LN1G, T_LN1B, p.eps1);
    PK_TRACE();   // LN1 done
    // VMEM schedule around a tile boundary (chunk c = consumption order; hooks issue one row instruction per group):
    //   W2_14 (c46): 16 loads, W2_15 (c47): 16 loads, Q_0 (c48): 16 stores, K_0 (c49): 16 stores; everything else none.
    //   pieces in one burst behind the mid-chunk barrier (PD_PAIR_DMA_SPREAD 0): chunk c+1's pieces were issued at the sync of chunk c-2,
    //     VMC(c) = 8 + hook instructions issued in [second half of c-2, first half of c]:
    //     c46: 8 + 8 = 16;  c47: 8 + 16 + 8 = 32;  c48: 8 + 8 + 16 + 8 = 40;  c49: 40;  c50 (V_0): 8 + 8 + 16 = 32;  c51 (P_0): 8 + 8 = 16.
    //   one piece every second group (PD_PAIR_DMA_SPREAD 1): VMC(c) = 8 + hook instructions issued in [second half of c-1, first half of c]:
    //     c46: 8 + 0 + 8 = 16;  c47: 8 + 8 + 8 = 24;  c48: 8 + 8 + 8 = 24;  c49: 24;  c50 (V_0): 8 + 8 + 0 = 16;  c51 (P_0): 8.
    //   with ONE cuboid per wave (NC = 1) only W2_14 carries loads (16) and Q_0 stores (16): c46 16, c47 8 + 16 = 24, c48 8 + 8 + 8 = 24,
    //   c49 8 + 16 = 24, c50 8 + 8 = 16, c51 8  (burst form; the spread form is built for NC = 2 only).
    static_assert(NC == 2 || !PD_PAIR_DMA_SPREAD, "the one-piece-per-group DMA schedule is derived for two cuboids per wave");
    //   PD_PAIR_DMA_EARLY: chunk c + 1's pieces go out at the START of chunk c - 2, so VMC(c) = 16 (chunks c + 2, c + 3) + the hook instructions of
    //   chunks c - 2, c - 1 and the first half of c:  NC = 2: c46 24, c47 40, c48 56, c49 56, c50 48, c51 32;  NC = 1: 24, 32, 40, 32, 32, 16.
    constexpr int VMC_W2_14 = PD_PAIR_DMA_EARLY ? 24 : 16;
    constexpr int VMC_W2_15 = PD_PAIR_DMA_EARLY ? (NC == 2 ? 40 : 32) : (NC == 2 ? PK_VMC(32, 24) : 24);
    constexpr int VMC_Q0 = PD_PAIR_DMA_EARLY ? (NC == 2 ? 56 : 40) : (NC == 2 ? PK_VMC(40, 24) : 24);
    constexpr int VMC_K0 = PD_PAIR_DMA_EARLY ? (NC == 2 ? 56 : 32) : (NC == 2 ? PK_VMC(40, 24) : 24);
    constexpr int VMC_V0 = PD_PAIR_DMA_EARLY ? (NC == 2 ? 48 : 32) : (NC == 2 ? PK_VMC(32, 16) : 16);
    constexpr int VMC_P0 = PD_PAIR_DMA_EARLY ? (NC == 2 ? 32 : 16) : (NC == 2 ? PK_VMC(16, 8) : 8);
    static_assert(!(PD_PAIR_DMA_EARLY && PD_PAIR_DMA_SPREAD), "one DMA schedule at a time");
    auto head = [&](auto first_tag, int h) __attribute__((always_inline)) {
      constexpr bool FIRST = decltype(first_tag)::value;
      const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
      f32x4 t[NC][4];
      bf16x8 qf[NC][2], kf[NC][2];
      f32x4 rb;                                     // relative-position bias of (head h, query q, keys 4 g .. 4 g + 3)
      const uint32_t vrb_h = vrb + (uint32_t)h * 1024u;
      // ---------------- q^T = Wq_h a^T  (head 0: the previous tile's rows of cuboid 0 leave in its shadow) ----------------
#pragma unroll
      for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) t[c][dt] = z4;
      PK_CHUNK(FIRST ? VMC_Q0 : PK_VMC0, 0, (void)0, (void)0, PK_MFMA_T(t, af), { if constexpr (FIRST && PK_HOOK_IO) PK_ROW_ST(acc[0][gi], ooff[0], gi); })
#if PD_PAIR_DEBUG
      if (h == 0) { for (int c = 0; c < NC; ++c) dump4(1, c, t[c][0], t[c][1], t[c][2], t[c][3]); }
#endif
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        qf[c][0] = pk_pack8(t[c][0], t[c][1]);
        qf[c][1] = pk_pack8(t[c][2], t[c][3]);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) t[c][dt] = z4;
      }
      PK_TRACE();   // q done
      // ---------------- k^T = Wk_h a^T  (head 0: ... and those of cuboid 1) ----------------
      PK_CHUNK(FIRST ? VMC_K0 : PK_VMC0, 1, PK_LDS_F4(rb, vrb_h, 0), PK_LANDED(rb), PK_MFMA_T(t, af), { if constexpr (FIRST && PK_HOOK_IO && NC == 2) PK_ROW_ST(acc[NC - 1][gi], ooff[NC - 1], gi); })
      PK_DRAIN();
      PK_TRACE();   // k done
      // ---------------- S^T = K Q^T, softmax over the keys (registers + two row swaps) ----------------
      s16x4 pf[NC];
#pragma unroll
      for (int c = 0; c < NC; ++c) {
#if PD_PAIR_DEBUG
        if (h == 0) dump4(2, c, t[c][0], t[c][1], t[c][2], t[c][3]);
#endif
        kf[c][0] = pk_pack8(t[c][0], t[c][1]);
        kf[c][1] = pk_pack8(t[c][2], t[c][3]);
        f32x4 s4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[c][0], qf[c][0], z4, 0, 0, 0);
        s4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[c][1], qf[c][1], s4, 0, 0, 0);
        float sc[4], mx = -3.0e38f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = -INFINITY;
          if (4 * g + r < p.vol && q < p.vol) v = s4[r] * p.scale + rb[r];
          sc[r] = v;
          mx = fmaxf(mx, v);
        }
#if PD_PAIR_DEBUG
        if (h == 0) dump4(3, c, s4, f32x4{sc[0], sc[1], sc[2], sc[3]}, rb, z4);
#endif
        mx = pk_rows4_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          sc[r] = __builtin_amdgcn_exp2f((sc[r] - mx) * 1.4426950408889634f);   // exp(-inf) = 0 for non-existent keys
          sum += sc[r];
        }
        sum = pk_rows4_sum(sum);
        const float inv = sum > 0.f ? __builtin_amdgcn_rcpf(sum) : 0.f;
        pf[c] = pk_pack4(f32x4{sc[0] * inv, sc[1] * inv, sc[2] * inv, sc[3] * inv});
#if PD_PAIR_DEBUG
        if (h == 0) dump4(4, c, f32x4{sc[0] * inv, sc[1] * inv, sc[2] * inv, sc[3] * inv}, f32x4{mx, sum, inv, 0.f}, z4, z4);
#endif
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) t[c][dt] = z4;
      }
      PK_TRACE();   // softmax done
      // ---------------- v = a Wv_h^T (plain product: lane = feature, 4 consecutive tokens -> the A operand of O^T = V^T P^T) -------
      PK_CHUNK(FIRST ? VMC_V0 : PK_VMC0, 0, (void)0, (void)0, {
        if (PK_MFMA_ON) {
          _Pragma("unroll") for (int c_ = 0; c_ < NC; ++c_) t[c_][i & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[c_][i >> 2], wf, t[c_][i & 3], 0, 0, 0);
        }
      }, (void)0)
      PK_DRAIN();
      bf16x8 of[NC][2];
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        f32x4 o[4];
#if PD_PAIR_DEBUG
        if (h == 0) dump4(5, c, t[c][0], t[c][1], t[c][2], t[c][3]);
#endif
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(pk_pack4(t[c][dt]), pf[c], z4, 0, 0, 0);
#if PD_PAIR_DEBUG
        if (h == 0) dump4(6, c, o[0], o[1], o[2], o[3]);
#endif
        of[c][0] = pk_pack8(o[0], o[1]);
        of[c][1] = pk_pack8(o[2], o[3]);
      }
      if constexpr (FIRST) add_vec(xn, T_BP);       // the finished rows have left: acc <- x + b_proj, the accumulator of every head's proj
      PK_TRACE();   // v + PV done
      // ---------------- x^T += Wp[:, head h] O_h^T ----------------
      PK_CHUNK(FIRST ? VMC_P0 : PK_VMC0, 0, (void)0, (void)0, PK_MFMA_OUT(of), (void)0)
    };
    head(std::true_type{}, 0);
#pragma unroll 1
    for (int h = 1; h < HEADS; ++h) head(std::false_type{}, h);
    PK_DRAIN();                                     // (a LayerNorm follows)
    PK_TRACE();   // attention done

    // ================= FFN: x += W2 gelu(W1 LN2(x) + b1) + b2 =================
    layer_norm(acc, T_LN2G, T_LN2B, p.eps2);
    add_vec(acc, T_B2);
    PK_TRACE();   // LN2 done
    // Chunk order: W1_0, W1_1, (W2_j, W1_{j+2}) for j = 0..13, W2_14, W2_15.  gelu(h_j) has the two chunks between W1_j and W2_j to
    // itself, as three software-pipelined stages (polynomial | exp, +1 | rcp, mul) of one value per fragment group: independent
    // short chains beside the MFMAs instead of one 9-deep dependent chain per value (a lone wave hides no VALU latency).
    const uint32_t vb1 = vtab + (uint32_t)(T_B1 * 4);
    f32x4 hc[NC][4], hn[NC][4], b1n[4];
    float ga[32], gd[32];
#define PK_HV(H, v) H[(v) >> 4][((v) >> 2) & 3][(v) & 3]
#define PK_GELU_GROUP(H, VB, NPER, GI)                                                                                    \
  if (!(PD_PAIR_ABLATE & 8)) _Pragma("unroll") for (int u_ = 0; u_ < (NPER); ++u_) {                                      \
    if ((GI) < 16) { const int v_ = (VB) + (GI) * (NPER) + u_; ga[v_] = pk_gelu_arg(PK_HV(H, v_)); }                      \
    if ((GI) >= 1 && (GI) < 17) { const int v_ = (VB) + ((GI) - 1) * (NPER) + u_; gd[v_] = 1.0f + __builtin_amdgcn_exp2f(ga[v_]); } \
    if ((GI) >= 2 && (GI) < 18) { const int v_ = (VB) + ((GI) - 2) * (NPER) + u_; PK_HV(H, v_) = PK_HV(H, v_) * __builtin_amdgcn_rcpf(gd[v_]); } \
  }
#define PK_B1_FETCH(ADDR) { PK_LDS_F4(b1n[0], ADDR, 0); PK_LDS_F4(b1n[1], ADDR, 64); PK_LDS_F4(b1n[2], ADDR, 128); PK_LDS_F4(b1n[3], ADDR, 192); }
#define PK_B1_LANDED() { PK_LANDED(b1n[0]); PK_LANDED(b1n[1]); PK_LANDED(b1n[2]); PK_LANDED(b1n[3]); }
    // b1 of chunk 0: plain wait (the fragment prologue in flight is older and simply lands first)
    PK_B1_FETCH(vb1)
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b1n[0]), "+v"(b1n[1]), "+v"(b1n[2]), "+v"(b1n[3]));
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int ht = 0; ht < 4; ++ht) hc[c][ht] = b1n[ht];
    const uint32_t vb1_1 = vb1 + 256u;
    // ---------------- h_0^T = W1_0 a^T + b1 (in its shadow: b1 of chunk 1) ----------------
    PK_CHUNK(PK_VMC0, 4, PK_B1_FETCH(vb1_1), PK_B1_LANDED(), PK_MFMA_T(hc, af), (void)0)
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int ht = 0; ht < 4; ++ht) hn[c][ht] = b1n[ht];
    PK_TRACE();   // W1_0 done
    // ---------------- h_1 beside the whole of gelu(h_0) (two values per group); b1 of chunk 2 ----------------
    const uint32_t vb1_2 = vb1 + 512u;
    PK_CHUNK(PK_VMC0, 4, PK_B1_FETCH(vb1_2), PK_B1_LANDED(), PK_MFMA_T(hn, af), PK_GELU_GROUP(hc, 0, NC, gi))
    PK_GELU_GROUP(hc, 0, NC, 16)
    PK_GELU_GROUP(hc, 0, NC, 17)
    bf16x8 hfr[NC][2];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      hfr[c][0] = pk_pack8(hc[c][0], hc[c][1]);
      hfr[c][1] = pk_pack8(hc[c][2], hc[c][3]);
    }
    PK_TRACE();   // W1_1 done
    // invariant: hfr = gelu(h_j) as fragments, hn = h_{j+1} (pre-activation), b1n = b1 of chunk j + 2
#pragma unroll 1
    for (int j = 0; j < HID / 64 - 2; ++j) {
#pragma unroll
      for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int ht = 0; ht < 4; ++ht) hc[c][ht] = b1n[ht];
      const uint32_t vb1n = vb1 + (uint32_t)(j + 3 < HID / 64 ? j + 3 : 0) * 256u;
      if constexpr (NC == 2 && PD_PAIR_GELU_BOTH) {
      // x^T += W2[:, chunk j] gelu(h_j)^T   beside the first half of gelu(h_{j+1}) (cuboid 0)
      PK_CHUNK(PK_VMC0, 0, (void)0, (void)0, PK_MFMA_OUT(hfr), PK_GELU_GROUP(hn, 0, 1, gi))
      PK_GELU_GROUP(hn, 0, 1, 16)
      PK_GELU_GROUP(hn, 0, 1, 17)
      // h_{j+2}^T = W1_{j+2} a^T + b1   beside the second half of gelu(h_{j+1}); b1 of chunk j + 3
      PK_CHUNK(PK_VMC0, 4, PK_B1_FETCH(vb1n), PK_B1_LANDED(), PK_MFMA_T(hc, af), PK_GELU_GROUP(hn, 16, 1, gi))
      PK_GELU_GROUP(hn, 16, 1, 16)
      PK_GELU_GROUP(hn, 16, 1, 17)
      } else {
      // x^T += W2[:, chunk j] gelu(h_j)^T
      PK_CHUNK(PK_VMC0, 0, (void)0, (void)0, PK_MFMA_OUT(hfr), (void)0)
      // h_{j+2}^T = W1_{j+2} a^T + b1   beside the whole of gelu(h_{j+1}) (two values per group); b1 of chunk j + 3.
      PK_CHUNK(PK_VMC0, 4, PK_B1_FETCH(vb1n), PK_B1_LANDED(), PK_MFMA_T(hc, af), PK_GELU_GROUP(hn, 0, NC, gi))
      PK_GELU_GROUP(hn, 0, NC, 16)
      PK_GELU_GROUP(hn, 0, NC, 17)
      }
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        hfr[c][0] = pk_pack8(hn[c][0], hn[c][1]);
        hfr[c][1] = pk_pack8(hn[c][2], hn[c][3]);
#pragma unroll
        for (int ht = 0; ht < 4; ++ht) hn[c][ht] = hc[c][ht];
      }
    }
    PK_TRACE();   // FFN loop done
    // W2_14 beside the whole of gelu(h_15) and the next tile's rows of cuboid 0, then W2_15 beside those of cuboid 1
    // (the LayerNorm fragments are dead: xn takes their registers)
    PK_CHUNK(VMC_W2_14, 0, (void)0, (void)0, PK_MFMA_OUT(hfr), { PK_GELU_GROUP(hn, 0, NC, gi) if (PK_HOOK_IO) xn[0][gi] = PK_ROW_LD(noff[0], gi); })
    PK_GELU_GROUP(hn, 0, NC, 16)
    PK_GELU_GROUP(hn, 0, NC, 17)
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      hfr[c][0] = pk_pack8(hn[c][0], hn[c][1]);
      hfr[c][1] = pk_pack8(hn[c][2], hn[c][3]);
    }
    PK_CHUNK(VMC_W2_15, 0, (void)0, (void)0, PK_MFMA_OUT(hfr), { if constexpr (NC == 2) { if (PK_HOOK_IO) xn[NC - 1][gi] = PK_ROW_LD(noff[NC - 1], gi); } })
    PK_DRAIN();                                     // (the next tile's LayerNorm follows)
    PK_TRACE();   // FFN done
  }
  // ---- the last tile's rows ----
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int nt = 0; nt < 16; ++nt) PK_ROW_ST(acc[c][nt], roff[c], nt);
  // nothing of this workgroup may still be writing LDS when its allocation is handed to the next one
  asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
#endif
}

template <int NC>
static int launch_pair(const pd_pair_args_k& a, hipStream_t s) {
  using namespace pairk;
  static bool attr_set_dev[PD_MAX_DEVICES];
  bool& attr_set = attr_set_dev[pd_cur_device()];
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)pair_kernel<NC>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e != hipSuccess) {
      pd_set_error("pd_attn_ffn_pair: hipFuncSetAttribute(%d) failed: %s", LDS_BYTES, hipGetErrorString(e));
      return PD_ERR_LAUNCH;
    }
    attr_set = true;
  }
  // persistent: every workgroup takes the same number of tiles (the last few one less), one workgroup per CU
  const int per_wg = (a.ntiles + 255) / 256;
  const int grid = (a.ntiles + per_wg - 1) / per_wg;
  hipLaunchKernelGGL(pair_kernel<NC>, dim3((unsigned)grid), dim3(256), LDS_BYTES, s, a);
  PD_CHECK_LAUNCH();
  return PD_OK;
}

extern "C" unsigned long long* pd_pair_trace = nullptr;
extern "C" int pd_pair_force_nc = 0;             // A/B: 1 / 2 = cuboids per wave whatever the grid; 0 = automatic
#if PD_PAIR_DEBUG
extern "C" float* pd_pair_dbg_buf = nullptr;    // (profiling / debugging builds only: scripts/debug_pair.py)
extern "C" int pd_pair_dbg_stage = 0;
#endif

extern "C" int pd_attn_ffn_pair_supported(int C, int heads, int hidden, int vol, int act) {
  return C == 256 && heads == 4 && hidden == 1024 && vol >= 1 && vol <= 16 && act == PD_ACT_GELU;
}

extern "C" int pd_attn_ffn_pair(const float* x, float* out, const void* wstream, const float* vecs, const int32_t* tok_index,
                                const int32_t* tok_affine, int B, int ntok, int nc, int vol, float scale, float eps_attn, float eps_ffn,
                                pd_stream_t stream) {
  using namespace pairk;
  PD_CHECK_ARG(x && out && wstream && vecs, "pd_attn_ffn_pair: null pointer");
  PD_CHECK_ARG(B > 0 && ntok > 0 && (int64_t)B * ntok < (1ll << 31), "pd_attn_ffn_pair: bad sizes");
  PD_CHECK_ARG(nc > 0 && vol >= 1 && vol <= 16, "pd_attn_ffn_pair: cuboid volume %d not in 1..16", vol);
  PD_CHECK_ARG(tok_index || (tok_affine && tok_affine[0] > 0), "pd_attn_ffn_pair: neither a token table nor its affine form");
  PD_CHECK_ARG((int64_t)B * ntok * (C * 4) < 0xFFFFF000ll, "pd_attn_ffn_pair: x larger than a 4 GiB buffer descriptor");
  pd_pair_args_k a;
  a.x = x; a.out = out; a.wstream = wstream; a.vecs = vecs; a.tok_index = tok_index;
  a.scale = scale; a.eps1 = eps_attn; a.eps2 = eps_ffn;
  a.wbytes = (uint32_t)(CH_ALL * CHUNK);
  a.xbytes = (uint32_t)((int64_t)B * ntok * (C * 4));
  a.trace = pd_pair_trace;
#if PD_PAIR_DEBUG
  a.dbg_buf = pd_pair_dbg_buf;
  a.dbg_stage = pd_pair_dbg_stage;
#else
  a.dbg_buf = nullptr;
  a.dbg_stage = 0;
#endif
  a.B = B; a.ntok = ntok; a.nc = nc; a.vol = vol;
  a.aff_on = (tok_affine && tok_affine[0] > 0) ? 1 : 0;
  a.aff_ninner = a.aff_on ? tok_affine[0] : 1;
  a.aff_outer = a.aff_on ? tok_affine[1] : 0;
  a.aff_inner = a.aff_on ? tok_affine[2] : 0;
  a.aff_slot = a.aff_on ? tok_affine[3] : 0;
  // two cuboids per wave (128-row tiles: every weight fragment feeds two MFMAs) once that leaves no CU idle; below that ONE cuboid per
  // wave (64-row tiles): twice the workgroups, half the MFMAs per streamed chunk -- the small-batch form
  const int64_t cuboids = (int64_t)B * nc;
  const int nc_wave = pd_pair_force_nc ? pd_pair_force_nc : ((cuboids + 7) / 8 > 128 ? 2 : 1);
  a.ntiles = (int)((cuboids + 4 * nc_wave - 1) / (4 * nc_wave));
  return nc_wave == 2 ? launch_pair<2>(a, (hipStream_t)stream) : launch_pair<1>(a, (hipStream_t)stream);
}
