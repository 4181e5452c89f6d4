// pd_attn_ffn_pair:  x += proj(cuboid_attention(qkv(LayerNorm(x))));  x += W2 gelu(W1 LayerNorm(x) + b1) + b2   in ONE kernel
// for the blocks of the SEVIR-LR denoiser -- level 0: units 256, 4 heads of 64, hidden 1024 (CW = 1); level 1: units 512, 4 heads of
// 128, hidden 2048 (CW = 2) -- with cuboid volume <= 16 (two cuboids of volume <= 8 share a 16-slot group):
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
// Forms: <NC = 2, CW = 1> 4 waves x 32 rows (described above), <1, 1> 4 waves x 16 rows (64-row tiles: small grids), <1, 1, 8> 8 waves x 16
// rows (two waves per SIMD, 226 registers: the 128-row form in use), <1, 2> 4 waves x 16 rows of 512 (units 512).  LDS 16 / 24 KB of
// tables + 128 KB ring.
// Numerics: those of the bf16 engine (bf16 LayerNorm output, q, k, v, P, O, hidden; fp32 accumulation, softmax, residual).  GELU is
// evaluated as x * sigmoid(x (a + b x^2 + c x^4)) (max abs deviation from the erf form 2.5e-5, below the bf16 rounding of the hidden
// activations that follows; 9 VALU instructions instead of 16).
#include <algorithm>
#include <type_traits>
#include "common.h"

namespace PD_NS {

// No implicit mul + add contraction in this file: the instantiations of the kernel (one / two groups per wave, units 256 / 512) must
// compute a row with the SAME fp32 operations -- the batch-split-reproducible mode of the engine runs a trajectory through whichever
// instantiation its launch size selects and promises bit-identical results (tests/test_hip_configs.py::test_v1_lane_split_tolerance;
// with the default fp-contract=fast hipcc fused different pairs in different instantiations and one row in a thousand differed by a
// bf16 rounding).  Every fused multiply-add below is written as one.
#pragma clang fp contract(off)

#define BLDS16I(rsrc, ldsptr, voff, soff, imm) \
  __builtin_amdgcn_raw_ptr_buffer_load_lds((rsrc), (__attribute__((address_space(3))) void*)(ldsptr), 16, (voff), (soff), (imm), 0)

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

namespace pairk {
constexpr int HEADS = 4;
#ifndef PD_PAIR_PF
#define PD_PAIR_PF 4
#endif
// Two other DMA schedules were built behind compile-time switches, measured on MI355X and removed again (profiles/r04_*):
// one piece every second fragment group instead of a burst of 8 behind the mid-chunk barrier (3 % slower: with one wave per SIMD every
// piece costs the wave its own issue slots either way), and a second barrier at the START of a chunk behind which chunk c + 3 is
// requested (259-265 vs 250-258 us at 32 trajectories, 53 vs 50 us at 4: the stream is not DMA-latency bound).
#ifndef PD_PAIR_PF1
#define PD_PAIR_PF1 4               // (4 / 6 / 8 in flight measured alike for both forms, two rounds on one box: 297-303 / 253-259 / 242-248 us at units
#endif                              //  512, 254-258 / 241-249 / 230-248 us at units 256, 46-48 us at 4 trajectories: LDS latency is covered at 4)
// weight fragments in flight (PD_PAIR_PF with two groups per wave, PD_PAIR_PF1 with one).  Register slots: the slot of fragment i is
// i % PFN in EVERY chunk, so PFN must divide the 32 fragments of a chunk.
#ifndef PD_PAIR_PF8
#define PD_PAIR_PF8 4
#endif
template <int NC, int NWV> struct FragPipe {
  static constexpr int PF = NWV == 8 ? PD_PAIR_PF8 : NC == 2 ? PD_PAIR_PF : PD_PAIR_PF1, PFN = PF + 2 <= 4 ? 4 : PF + 2 <= 8 ? 8 : 16;
};
constexpr int CHUNK = 32768, NSLOT = 4, NFRAG = 32;
constexpr int DMA_PER_WAVE = CHUNK / 4 / 1024;     // 8 x 1 KB per wave per chunk
// geometry of one block width: CW = units / 256
template <int CW>
struct G {
  static constexpr int C = 256 * CW, HID = 1024 * CW, HD = C / HEADS;
  static constexpr int CT = C / 16;                // 16-column tiles of a row
  static constexpr int KS = C / 32;                // k-steps of 32 over the units
  static constexpr int DT = HD / 16;               // 16-feature tiles of a head
  static constexpr int HS = HD / 32;               // k-steps over a head
  static constexpr int NQ = CW * CW;               // chunks of one head's Wq / Wk / Wv ([HD x C]) and of its proj slice ([C x HD])
  static constexpr int NW = CW;                    // chunks of W1_j ([64 x C]) and of W2_j ([C x 64])
  static constexpr int NJ = HID / 64;
  // fp32 tables in LDS (float offsets) == layout of the `vecs` argument
  static constexpr int T_LN1G = 0, T_LN1B = C, T_BP = 2 * C, T_LN2G = 3 * C, T_LN2B = 4 * C, T_B2 = 5 * C, T_B1 = 6 * C, T_RB = 6 * C + HID,
                       T_FLOATS = T_RB + HEADS * 256;
  static constexpr int RING_OFF = (T_FLOATS * 4 + 8191) / 8192 * 8192;
  static constexpr int LDS_BYTES = RING_OFF + NSLOT * CHUNK;       // 147456 / 155648
  static constexpr int CH_ALL = HEADS * 4 * NQ + 2 * NJ * NW;      // chunks per tile: 48 / 192
  static_assert(LDS_BYTES <= 160 * 1024, "LDS");
};
}  // namespace pairk

struct pd_pair_args_k {
  const float* x;
  float* out;
  const void* wstream;        // CH_ALL chunks of 32 KB, packing.pack_pair_block
  const float* vecs;          // T_FLOATS floats
  const int32_t* tok_index;   // [nc][vol] or null with aff_on
  int B, ntok, nc, vol;
  int pack, gps;              // cuboids per 16-slot group (2 when 2 vol <= 16); groups per sample = ceil(nc / pack)
  float scale, eps1, eps2;
  int aff_on, aff_ninner, aff_outer, aff_inner, aff_slot;
  int ntiles;
  uint32_t wbytes, xbytes;
  unsigned long long* trace;
  float* dbg_buf;             // PD_PAIR_DEBUG builds only: [rows][256] dump of one intermediate of head 0 (dbg_stage)
  int dbg_stage;
  // split forms (MODE 1 / 2, small grids): blockIdx.y = slice (head / quarter of the hidden units)
  const float* slab_in;       // MODE 2: the NSPL attention partials [slice][B * ntok][C] of MODE 1, summed in slice order onto x + b_proj
  float* slab_out;            // MODE 1 / 2: this slice's partial result, [slice][B * ntok][C]
  const void* wstream2;       // MODE 2: the FFN chunks in quarter-major order (packing.pack_pair_ffn_split)
  uint32_t w2bytes;
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
// fp32 -> operand type (RNE) through the compiler's own v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32 (common.h cvt_op4; NOT the inline-asm form: hipcc pads no hazard wait states
// around an asm statement, and here the conversions sit directly between MFMAs -- an asm v_cvt reading a fresh MFMA result, or an MFMA
// reading a fresh asm v_cvt result, gets stale registers; measured: wrong O tiles / NaNs in the first build of this kernel)
__device__ __forceinline__ op8 pk_pack8(const f32x4& a, const f32x4& b) {
  const op4v lo = cvt_op4(a), hi = cvt_op4(b);
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
__device__ __forceinline__ s16x4 pk_pack4(const f32x4& a) {
  return __builtin_bit_cast(s16x4, cvt_op4(a));
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
  {                                                                                                                       \
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]));                                 \
    if constexpr (PFN >= 8) asm volatile("" : "+v"(w[4]), "+v"(w[5]), "+v"(w[6]), "+v"(w[7]));                             \
    if constexpr (PFN == 16)                                                                                              \
      asm volatile("" : "+v"(w[8]), "+v"(w[9]), "+v"(w[10]), "+v"(w[11]), "+v"(w[12]), "+v"(w[13]), "+v"(w[14]), "+v"(w[15])); \
  }

#ifndef PD_PAIR_DEBUG
#define PD_PAIR_DEBUG 0
#endif
#ifndef PD_PAIR_TRACE
#define PD_PAIR_TRACE 0
#endif
// compile-time ablations for profiling builds (-DPD_PAIR_ABLATE=bits, scripts/ablate_pair.sh): 1 no weight DMA after the prologue,
// 2 no fragment reads, 4 no MFMAs in the chunk bodies, 8 no GELU work in the chunk hooks, 16 no row loads / stores in the chunk hooks,
// 32 no mid-chunk wait + barrier (timing only: the results are garbage)
#ifndef PD_PAIR_ABLATE
#define PD_PAIR_ABLATE 0
#endif
#ifndef PD_PAIR_GELU_SPREAD
#define PD_PAIR_GELU_SPREAD 0         // one group per wave: 1 = gelu(h_j) one value every 2 / 4 fragment groups across all the chunks between W1_j and
#endif                                // W2_j, 0 = all of it beside the first W1 chunk (one value per group).  A/B on one MI355X box, two
                                      // rounds: units 512 at 32 trajectories 305 / 261 / 249 us spread vs 299 / 258 / 247 us; units 256 at
                                      // 4 trajectories 48.4 / 46.1 / 46.8 vs 48.4 / 46.1 / 45.6 -- no gain: the groups are not issue bound
#ifndef PD_PAIR_GELU_BOTH
#define PD_PAIR_GELU_BOTH 1           // 1: gelu(h_j) split over the W2 and the W1 chunk between W1_j and W2_j; 0: all of it beside the W1 chunk
                                      // (A/B on MI355X, 32 trajectories, two rounds: 258-273 us either way -- the FFN iteration's time is a
                                      // property of the chunk pair, the four waves meet at every chunk's barrier)
#endif

// MODE (small grids: a 64-row tile per CU leaves most of the chip idle and every tile streams ALL the weights -- 6.3 MB at units 512):
//   0  the whole pair in one workgroup per tile (everything above);
//   1  grid (tiles, 4): workgroup (tile, h) runs LayerNorm-1 and head h only (a quarter of the attention chunks) and stores its proj
//      partial  Wp[:, head h] O_h  (no bias, no residual) to slab_out[h];
//   2  grid (tiles, 4): workgroup (tile, s) forms  x' = ((((x + b_proj) + slab_in[0]) + slab_in[1]) + slab_in[2]) + slab_in[3]  (fixed order:
//      deterministic), LayerNorm-2, and the FFN over hidden units [s HID / 4, (s + 1) HID / 4) from the quarter-major stream wstream2;
//      slice 0's partial starts from x' + b_2, the others from zero; -> slab_out[s].  pair_split_sum_kernel adds the four.
// Four times the workgroups, a quarter of the weight stream each.  The split forms keep no cross-tile pipelining of the rows (plain
// bursts at the tile boundaries); their fp32 summation order differs from MODE 0's (partials added instead of one running accumulator),
// so the engine uses them only where it may choose kernels by launch size (split_k: the small-batch mode).
constexpr int PAIR_NSPL = 4;
// WP (round 6, precision="fp16x2"): weight products per k-step.  2 = every matrix of the stream is followed by its LOW part (W = W_hi + W_lo in
// the 16-bit operand type, packing.pack_pair_block(fold=True)): each group of chunks (a head's Wq / Wk / Wv / proj slice, a hidden slice of W1 /
// W2) comes twice, and the second pass multiplies the SAME activation fragments into the SAME accumulators -- the chunk loops simply run
// NQ = WP * CW^2 (NW = WP * CW) chunks per matrix with the operand index folded (SUB mod the single-product count); nothing else changes:
// no new registers, the same tile-boundary schedule (its constants count row instructions and DMA pieces, not what a chunk holds).
// NSPL (MODE 2): hidden slices the FFN is cut into.  4 = the split form of the pair; **1 = the FFN alone** (pd_ffn_rows, round 6): LayerNorm ->
// W1 -> GELU -> W2 -> + x over ALL hidden units of arbitrary rows (any 16 consecutive rows make a group), straight to `out` -- for blocks
// whose attention the pair kernel cannot take (cuboid volumes > 16: the full-resolution grid), where the FFN alone is 2/3 of the pair's FLOPs.
template <int NC, int CW, int NWV = 4, int MODE = 0, int WP = 1, int NSPL = 4>
__global__ void __launch_bounds__(NWV * 64, 1) pair_kernel(const pd_pair_args_k p) {
#if defined(__HIP_DEVICE_COMPILE__)
  using namespace pairk;
  using GG = G<CW>;
  constexpr int C = GG::C, CT = GG::CT, KS = GG::KS, DT = GG::DT, HS = GG::HS, NJ = GG::NJ;
  constexpr int NQ1 = GG::NQ, NW1 = GG::NW;         // chunks of one product of a head matrix / of a hidden slice: the operand index wraps here
  constexpr int NQ = NQ1 * WP, NW = NW1 * WP;       // chunks the loops run per head matrix / hidden slice
  static_assert(WP == 1 || (WP == 2 && NC == 1), "folded weights: one 16-slot group per wave (the two-group FFN tail is written for one chunk per slice)");
  constexpr int T_LN1G = GG::T_LN1G, T_LN1B = GG::T_LN1B, T_BP = GG::T_BP, T_LN2G = GG::T_LN2G, T_LN2B = GG::T_LN2B, T_B2 = GG::T_B2,
                T_B1 = GG::T_B1, T_RB = GG::T_RB, T_FLOATS = GG::T_FLOATS, RING_OFF = GG::RING_OFF, CH_ALL = GG::CH_ALL * WP;
  static_assert(NC * CW <= 2, "a wave holds 32 x 256 or 16 x 512 fp32 row values");
  static_assert(NWV == 4 || (NWV == 8 && NC == 1 && CW == 1), "eight waves (two per SIMD, 256 registers each): 16 x 256 rows per wave only");
  static_assert(MODE == 0 || (NC == 1 && (NWV == 4 || (MODE == 2 && NSPL == 1)) && NJ % NSPL == 0 && HEADS == PAIR_NSPL),
                "split forms: one group per wave, four waves (the FFN-alone form also with eight)");
  static_assert(MODE == 2 || NSPL == PAIR_NSPL, "NSPL belongs to MODE 2");
  // the slice of a split form and its window of the weight stream
  constexpr int CH_HEAD = 4 * NQ, CH_FQ = 2 * (NJ / NSPL) * NW, NJL = MODE == 2 ? NJ / NSPL : NJ;
  const int slice = MODE ? (int)blockIdx.y : 0;
  const int kid_base = MODE == 1 ? slice * CH_HEAD : MODE == 2 ? slice * CH_FQ : 0;
  const int kid_end = MODE == 1 ? kid_base + CH_HEAD : MODE == 2 ? kid_base + CH_FQ : CH_ALL;
  constexpr int DPW = NFRAG / NWV;                  // 1 KB DMA pieces per wave per chunk: 8 / 4
  constexpr int PF = FragPipe<NC, NWV>::PF, PFN = FragPipe<NC, NWV>::PFN;
  static_assert(NFRAG % PFN == 0 && PF + 2 <= PFN && PF % 2 == 0 && PF + 4 <= 15, "fragment pipeline geometry");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane & 15, g = lane >> 4;

  // ---- tables -> LDS (before any DMA is in flight: a compiler-visible LDS store behind a DMA would drain it) ----
  for (int i = tid; i < T_FLOATS; i += NWV * 64) ((float*)smem)[i] = p.vecs[i];
  __syncthreads();

  // ---- weight stream: chunk ids 0 .. CH_ALL-1 cyclically, chunk number n -> ring slot n & 3 ----
  const auto rW = MODE == 2 ? __builtin_amdgcn_make_buffer_rsrc((void*)p.wstream2, 0, p.w2bytes, 0x00020000)
                            : __builtin_amdgcn_make_buffer_rsrc((void*)p.wstream, 0, p.wbytes, 0x00020000);
  // Every wave copies a quarter of every chunk, as 8 pieces of 1 KB (one DMA instruction each); piece k of a wave belongs to chunk
  // k / 8.  The prologue issues chunks 0..2, then the 8 pieces of chunk c + 3 go out in one burst behind the mid-chunk barrier of
  // chunk c (everybody is past chunk c - 1, whose slot this is).
  // The instruction's immediate offset advances BOTH the global and the LDS address, so a burst needs two scalar address pairs (pieces
  // 0..3 and 4..7 with immediates 0 / 1024 / 2048 / 3072), not eight: ~10 scalar instructions per chunk instead of ~110 (they were a
  // fifth of all instructions the level-1 instantiation issued in its FFN loop).
  int n_chunk = 0, kid = kid_base;                 // chunks requested by this wave; stream id of the next one
  const uint32_t dma_voff = (uint32_t)lane * 16u;
  auto issue_chunk = [&]() {
#if PD_PAIR_ABLATE & 1
    if (n_chunk >= 3) { ++n_chunk; return; }
#endif
    char* d = smem + RING_OFF + (n_chunk & (NSLOT - 1)) * CHUNK + wave * (DPW * 1024);
    const uint32_t so = (uint32_t)kid * CHUNK + (uint32_t)wave * (DPW * 1024);
    BLDS16I(rW, d, dma_voff, so, 0);
    BLDS16I(rW, d, dma_voff, so, 1024);
    BLDS16I(rW, d, dma_voff, so, 2048);
    BLDS16I(rW, d, dma_voff, so, 3072);
    if constexpr (DPW == 8) {
      BLDS16I(rW, d + 4096, dma_voff, so + 4096u, 0);
      BLDS16I(rW, d + 4096, dma_voff, so + 4096u, 1024);
      BLDS16I(rW, d + 4096, dma_voff, so + 4096u, 2048);
      BLDS16I(rW, d + 4096, dma_voff, so + 4096u, 3072);
    }
    ++n_chunk;
    kid = (kid + 1 == kid_end) ? kid_base : kid + 1;
  };
  static_assert(DPW == 8 || DPW == 4, "issue_chunk is written out for 8 or 4 pieces per wave");
  issue_chunk();
  issue_chunk();
  issue_chunk();

  const uint32_t vbase = (uint32_t)(uintptr_t)(smem + RING_OFF) + (uint32_t)lane * 16u;   // fragment reads: lane-linear 16 B
  const uint32_t vtab = (uint32_t)(uintptr_t)smem + (uint32_t)g * 16u;                     // fp32 tables: 4 floats at column 4 g
  const uint32_t vrb = (uint32_t)(uintptr_t)smem + (uint32_t)((T_RB + q * 16 + 4 * g) * 4);

#if PD_PAIR_DEBUG
  int tr_n = 0;
#define PK_TRACE() do { if (p.trace && blockIdx.x == 7 && tid == 0 && tr_n < 256) p.trace[tr_n] = __builtin_amdgcn_s_memtime(); ++tr_n; } while (0)
#else
#define PK_TRACE() do {} while (0)
#endif
  // PD_PAIR_TRACE (libprediff_hip_trace.so, bench.py's `phases`): THREE clock stamps per tile -- tile start, attention done, FFN done -- kept in
  // scalar registers and written by one lane at the end of the tile: the light form of the PD_PAIR_DEBUG stamps (24 per tile + the dump
  // hooks cost the units-512 instantiation its registers: that build ran 3x slower than the product kernel), within a few
  // percent of the product kernel's duration (bench.py compares the two and drops the split otherwise).
#if PD_PAIR_TRACE
  int trl_n = 0;
  unsigned long long trl_a = 0, trl_b = 0;
#define PK_STAMP_START() trl_a = __builtin_amdgcn_s_memtime()
#define PK_STAMP_ATT() trl_b = __builtin_amdgcn_s_memtime()
#define PK_STAMP_END()                                                                       \
  do {                                                                                       \
    const unsigned long long e_ = __builtin_amdgcn_s_memtime();                              \
    if (p.trace && blockIdx.x == 7 && tid == 0 && trl_n < 85) {                              \
      p.trace[3 * trl_n] = trl_a; p.trace[3 * trl_n + 1] = trl_b; p.trace[3 * trl_n + 2] = e_; \
    }                                                                                        \
    ++trl_n;                                                                                 \
  } while (0)
#else
#define PK_STAMP_START() do {} while (0)
#define PK_STAMP_ATT() do {} while (0)
#define PK_STAMP_END() do {} while (0)
#endif
  int cc = 0;                                       // chunks consumed by this workgroup
  op8 w[PFN] = {};                               // fragment pipeline (runs on across chunks, tiles and phases)

  // One chunk = 32 fragments = 16 groups of two.  Per group: two fragment reads three groups ahead, ONE counted wait, the MFMAs of the
  // group's two fragments (BODY_STMT, once per fragment: `i`, `wf`) and HOOK_STMT (`gi`): independent work for their shadow
  // (GELU stages, ONE row load or store).
  // In the middle of chunk cc (group 8, SYNC_STMT): chunk cc+1 has landed for everybody and everybody is past chunk cc-1, whose slot
  // takes chunk cc+3.  The last piece of chunk cc+1 was issued just before the previous mid-chunk sync, so the wait is vmcnt(VMC) with
  // VMC = the VMEM instructions issued since then that may stay in flight: the 8 DMA pieces of chunk cc+2 plus the row loads / stores
  // the hooks issued in the second half of chunk cc-1 and the first half of chunk cc (a fixed schedule: the constants are derived at
  // the tile loop).  Loads and stores retire in order (one vmcnt queue on gfx9-class hardware).  A VMC SMALLER than the true count is
  // merely a stronger wait; a larger one would let a piece of chunk cc+1 stay in flight.
  // EXTRA_STMT: NEXTRA other LDS reads issued at the start; they have landed at group PF / 2, where LANDED_STMT re-defines their
  // destinations (PK_LANDED).
  // RULE for every asynchronous (inline-asm) LDS read in this kernel: its destination must not live long BEFORE its wait -- the
  // compiler believes the register is valid from the asm statement on and may park a long-lived value in an AGPR at once, i.e. copy
  // it before the data has arrived.  So: fragments are consumed three groups after their read; table values are re-defined
  // (PK_LANDED, no instruction) right after the wait that covers them and only that new value lives on; and wherever more than a few
  // instructions separate two chunk loops the fragments in flight are drained first (PK_DRAIN).  scripts/check_async_lds.py replays
  // the LDS queue over the generated ISA and fails the build if any instruction touches a destination that is still in flight.
#define PK_VMC0 DPW   /* chunks without hook traffic around them: the DMA pieces younger than chunk c + 1 */
#if PD_PAIR_ABLATE & 32
#define PK_SYNC(VMC) asm volatile("" ::"n"(VMC) : "memory")
#else
#define PK_SYNC(VMC) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(VMC) : "memory")
#endif
#define PK_RD_ON (!(PD_PAIR_ABLATE & 2))
#define PK_MFMA_ON (!(PD_PAIR_ABLATE & 4))
#define PK_RD(i_)                                                                                 \
  if (PK_RD_ON) {                                                                                 \
    if ((i_) < NFRAG) PK_WLD(w[(i_) % PFN], va_, (i_) * 1024);                                     \
    else PK_WLD(w[(i_) % PFN], vn_, ((i_) - NFRAG) * 1024);                                        \
  }
#define PK_CHUNK(SYNC_STMT, NEXTRA, EXTRA_STMT, LANDED_STMT, BODY_STMT, HOOK_STMT)                \
  {                                                                                               \
    const uint32_t va_ = vbase + (uint32_t)(cc & (NSLOT - 1)) * CHUNK;                            \
    const uint32_t vn_ = vbase + (uint32_t)((cc + 1) & (NSLOT - 1)) * CHUNK;                      \
    _Pragma("unroll") for (int gi = 0; gi < NFRAG / 2; ++gi) {                                    \
      if (gi == 0) { EXTRA_STMT; }                                                                \
      if (gi == NFRAG / 4) {                                                                      \
        SYNC_STMT;                                                                                \
        issue_chunk();                                                                            \
      }                                                                                           \
      PK_RD(2 * gi + PF);                                                                         \
      PK_RD(2 * gi + PF + 1);                                                                     \
      asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(w[(2 * gi) % PFN]), "+v"(w[(2 * gi + 1) % PFN]) \
                   : "n"(PF + (2 * gi < PF ? (NEXTRA) : 0)));                                     \
      if (gi == PF / 2) { LANDED_STMT; }                                                          \
      __builtin_amdgcn_sched_barrier(0);                                                          \
      { const int i = 2 * gi; const op8 wf = w[i % PFN]; BODY_STMT; }                          \
      { const int i = 2 * gi + 1; const op8 wf = w[i % PFN]; BODY_STMT; }                      \
      { HOOK_STMT; }                                                                              \
      __builtin_amdgcn_sched_barrier(0);                                                          \
    }                                                                                             \
    ++cc;                                                                                         \
  }
  // fragment i of a chunk of ...
#define PK_MFMA_T(ACC, AF, SUB) /* ... a head's Wq / Wk ([HD x C], transposed product): feature tile i % DT, k-step SUB * 32 / DT + i / DT */ \
  if (PK_MFMA_ON) {                                                                                                  \
    _Pragma("unroll") for (int c_ = 0; c_ < NC; ++c_)                                                                \
      ACC[c_][i % DT] = mfma_16x16x32(wf, AF[c_][((SUB) % NQ1) * (32 / DT) + i / DT], ACC[c_][i % DT]); \
  }
#define PK_MFMA_V(ACC, AF, SUB) /* ... a head's Wv (plain product: lane = feature, 4 consecutive tokens) */             \
  if (PK_MFMA_ON) {                                                                                                  \
    _Pragma("unroll") for (int c_ = 0; c_ < NC; ++c_)                                                                \
      ACC[c_][i % DT] = mfma_16x16x32(AF[c_][((SUB) % NQ1) * (32 / DT) + i / DT], wf, ACC[c_][i % DT]); \
  }
#define PK_MFMA_H(ACC, AF, SUB) /* ... W1_j ([64 x C]): hidden tile i & 3, k-step 8 SUB + i / 4 */                      \
  if (PK_MFMA_ON) {                                                                                                  \
    _Pragma("unroll") for (int c_ = 0; c_ < NC; ++c_)                                                                \
      ACC[c_][i & 3] = mfma_16x16x32(wf, AF[c_][((SUB) % NW1) * 8 + (i >> 2)], ACC[c_][i & 3]); \
  }
#define PK_MFMA_OUT(OF, SUB, N1) /* ... a [C outputs x k] slice of Wproj / W2 (x^T += W act^T): column tile i % CT, k-step (SUB mod N1) * 32 / CT + i / CT */ \
  if (PK_MFMA_ON) {                                                                                                  \
    _Pragma("unroll") for (int c_ = 0; c_ < NC; ++c_)                                                                \
      acc[c_][i % CT] = mfma_16x16x32(wf, OF[c_][((SUB) % (N1)) * (32 / CT) + i / CT], acc[c_][i % CT]); \
  }

  // rows of a tile's 16-slot groups for this lane = (slot q, column group g), as byte offsets into x / out.  A group holds ONE cuboid,
  // or TWO of volume <= 8 (p.pack == 2: slots [0, vol) and [vol, 2 vol); the relative-position table keeps their scores apart).  An
  // invalid slot gets an offset beyond the buffers: its loads return 0 and its stores are dropped by the descriptor's bounds check --
  // ALWAYS exactly NC * CT load and store instructions per tile, no lane ever branches.
  const auto rX = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.xbytes, 0x00020000);
  // MODE 1 / 2: the rows leave to this slice's slab (same row layout as x: the same lane offsets address it)
  const auto rO = MODE ? __builtin_amdgcn_make_buffer_rsrc((void*)((char*)p.slab_out + (size_t)slice * p.xbytes), 0, p.xbytes, 0x00020000)
                       : __builtin_amdgcn_make_buffer_rsrc((void*)p.out, 0, p.xbytes, 0x00020000);
  constexpr uint32_t OOB = 0xFFFFF000u;
  auto tile_rows = [&](int tile, uint32_t (&off)[NC]) {
    // (the lane id is re-derived here, opaquely: q and g kept alive across the whole tile loop were the two registers this kernel
    //  did not have -- they went to scratch, and a scratch reload drains every DMA piece in flight)
    uint32_t l2;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l2));
    const int q = (int)(l2 & 15u), g = (int)(l2 >> 4);
    const int sub = q >= p.vol ? 1 : 0, slot = q - sub * p.vol;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      // group -> (sample, cuboids): groups never straddle samples, so where a cuboid sits inside its group -- and with it the order
      // in which its keys are summed -- does not depend on the batch the sample is launched in
      const int64_t gi = (int64_t)tile * (NWV * NC) + wave * NC + c;
      const int b = (int)(gi / p.gps), cu = (int)(gi - (int64_t)b * p.gps) * p.pack + sub;
      int row = -1;
      if (b < p.B && cu < p.nc && sub < p.pack && slot < p.vol) {
        const int tok = p.aff_on ? (cu / p.aff_ninner) * p.aff_outer + (cu % p.aff_ninner) * p.aff_inner + slot * p.aff_slot
                                 : p.tok_index[cu * p.vol + slot];
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
#define PK_HOOK_IO (!(PD_PAIR_ABLATE & 16) && MODE == 0)      /* (the split forms move their rows in plain bursts) */
  // hook of fragment group gi in the TP-th chunk of a tile: row instruction 16 TP + gi of the NC * CT stores of the previous tile's rows
#define PK_ST_HOOK(TP)                                                                                                     \
  { const int fi_ = 16 * (TP) + gi; if (PK_HOOK_IO && fi_ < NC * CT) PK_ROW_ST(acc[fi_ / CT][fi_ % CT], ooff[fi_ / CT], fi_ % CT); }
  // ... in the chunk EP chunks before the tile's last (EP = 1, 0): row instruction 16 (1 - EP) + gi of the NC * CT loads of the next tile's rows
#define PK_LD_HOOK(EP)                                                                                                     \
  { const int fi_ = 16 * (1 - (EP)) + gi; if (PK_HOOK_IO && fi_ < NC * CT) xn[fi_ / CT][fi_ % CT] = PK_ROW_LD(noff[fi_ / CT], fi_ % CT); }

  // acc: the rows in flight (x -> x + attn -> x + attn + ffn), lane = (token q, columns 16 nt + 4 g .. +3): the MFMA C layout of every
  // transposed product.  xn: the rows of the NEXT tile.  The tile boundary is software pipelined: xn is requested one row-instruction
  // per fragment group during the last two chunks of a tile, the finished rows (acc) leave one row-instruction per group during the
  // first two chunks of the next tile (whose LayerNorm and q products read xn), and only then acc <- xn + b_proj.
  f32x4 acc[NC][CT], xn[NC][CT];
  uint32_t roff[NC], noff[NC], ooff[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) roff[c] = ooff[c] = OOB;
  if constexpr (MODE == 0) {
    tile_rows(blockIdx.x, noff);
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int nt = 0; nt < CT; ++nt) {
        xn[c][nt] = PK_ROW_LD(noff[c], nt);
        acc[c][nt] = f32x4{0.f, 0.f, 0.f, 0.f};     // (the first tile has no predecessor: its hook stores go to OOB offsets)
      }
  }
  // MODE 2: x' = ((((x + b_proj) + P_0) + P_1) + P_2) + P_3 from the attention partials of MODE 1 -- always in this order.  One slab per
  // round trip into acc (nothing else is alive yet; two slabs at a time -- a third 128-register array -- spilled).
  auto sum_partials = [&](const uint32_t (&off)[NC], f32x4 (&dst)[NC][CT], f32x4 (&tmp)[NC][CT]) {
    const auto rS = __builtin_amdgcn_make_buffer_rsrc((void*)p.slab_in, 0, p.xbytes * (uint32_t)PAIR_NSPL, 0x00020000);
#pragma unroll 1
    for (int k = 0; k < PAIR_NSPL; ++k) {
      const uint32_t so = (uint32_t)k * p.xbytes;
#pragma unroll
      for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int nt = 0; nt < CT; ++nt)     // (an invalid row keeps its out-of-range LANE offset: zeros, whatever the scalar offset adds)
          tmp[c][nt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rS, off[c] + (uint32_t)(nt * 64), so, 0));
      if (k == 0) {                          // x + b_proj while the first slab is in flight
#pragma unroll
        for (int blk = 0; blk < CT / 8; ++blk) {
          f32x4 bv[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) PK_LDS_F4(bv[i], vtab, (T_BP + 16 * (blk * 8 + i)) * 4);
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bv[0]), "+v"(bv[1]), "+v"(bv[2]), "+v"(bv[3]), "+v"(bv[4]), "+v"(bv[5]), "+v"(bv[6]), "+v"(bv[7]));
#pragma unroll
          for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int c = 0; c < NC; ++c) dst[c][blk * 8 + i] = dst[c][blk * 8 + i] + bv[i];
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int nt = 0; nt < CT; ++nt) dst[c][nt] = dst[c][nt] + tmp[c][nt];
    }
  };
  // chunks 0..2 landed (the row loads above are younger: this wait covers both), visible to every wave; fragment prologue
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int i = 0; i < PF; ++i) PK_WLD(w[i], vbase, i * 1024);
  PK_DRAIN();

  for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
    PK_TRACE();   // tile start
    PK_STAMP_START();
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      ooff[c] = roff[c];
      roff[c] = noff[c];
      noff[c] = OOB;
    }
    if (MODE == 0 && tile + (int)gridDim.x < p.ntiles) tile_rows(tile + gridDim.x, noff);
    if constexpr (MODE != 0) {
      // split forms: a tile's rows arrive in one burst at its start (nothing is carried from tile to tile: no row registers live
      // across the chunk loops beyond the one array in use)
      tile_rows(tile, roff);
      if constexpr (MODE == 1) {
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
          for (int nt = 0; nt < CT; ++nt) xn[c][nt] = PK_ROW_LD(roff[c], nt);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else {                                       // MODE 2: x' is built in acc (where MODE 0 has it at this point), xn is the scratch array
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
          for (int nt = 0; nt < CT; ++nt) acc[c][nt] = PK_ROW_LD(roff[c], nt);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (p.slab_in) sum_partials(roff, acc, xn);      // (null: p.x already IS x' -- pair_split_xsum_kernel formed it once for the four slices)
      }
    }
#if PD_PAIR_DEBUG
    auto dump4 = [&](int stage, int c, const f32x4& a, const f32x4& b, const f32x4& cc4, const f32x4& d) {
      if (p.dbg_buf && p.dbg_stage == stage && roff[c] != OOB) {
        float* o = p.dbg_buf + roff[c] / (4 * CW);
        *(f32x4*)(o) = a; *(f32x4*)(o + 16) = b; *(f32x4*)(o + 32) = cc4; *(f32x4*)(o + 48) = d;
      }
    };
#endif
    op8 af[NC][KS];                               // LayerNorm output as B-operand fragments: [group][k-step of 32]
    // LayerNorm over the C columns of a row (C / 4 in this lane, the rest in lanes q + 16 g'), -> af
    auto layer_norm = [&](const f32x4 (&src)[NC][CT], int t_gamma, int t_beta, float eps) {
      float mean[NC], rstd[NC];
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        float s = 0.f;
#pragma unroll
        for (int nt = 0; nt < CT; ++nt) s += (src[c][nt][0] + src[c][nt][1]) + (src[c][nt][2] + src[c][nt][3]);
        mean[c] = pk_rows4_sum(s) * (1.0f / C);
        float v = 0.f;
#pragma unroll
        for (int nt = 0; nt < CT; ++nt) {
          const float d0 = src[c][nt][0] - mean[c], d1 = src[c][nt][1] - mean[c], d2 = src[c][nt][2] - mean[c], d3 = src[c][nt][3] - mean[c];
          v += __builtin_fmaf(d0, d0, d1 * d1) + __builtin_fmaf(d2, d2, d3 * d3);
        }
        rstd[c] = rsqrtf(__builtin_fmaf(pk_rows4_sum(v), 1.0f / C, eps));
        mean[c] = -mean[c] * rstd[c];              // y = (x rstd - mean rstd) gamma + beta: two fused multiply-adds per element
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
      for (int ks = 0; ks < KS; ++ks) {
        if (ks + 1 < KS) {
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
              y[hf][r] = __builtin_fmaf(__builtin_fmaf(src[c][2 * ks + hf][r], rstd[c], mean[c]), gb[ks & 1][hf][r], gb[ks & 1][2 + hf][r]);
          af[c][ks] = pk_pack8(y[0], y[1]);
        }
      }
    };
    // acc[c][nt] = src[c][nt] + table[16 nt + 4 g .. +3]  (proj / FFN-2 bias: the accumulator starts from residual + bias)
    auto add_vec = [&](const f32x4 (&src)[NC][CT], int t_off) {
#pragma unroll
      for (int blk = 0; blk < CT / 8; ++blk) {
        f32x4 bv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) PK_LDS_F4(bv[i], vtab, (t_off + 16 * (blk * 8 + i)) * 4);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bv[0]), "+v"(bv[1]), "+v"(bv[2]), "+v"(bv[3]), "+v"(bv[4]), "+v"(bv[5]), "+v"(bv[6]), "+v"(bv[7]));
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int c = 0; c < NC; ++c) acc[c][blk * 8 + i] = src[c][blk * 8 + i] + bv[i];
      }
    };

    // ================= attention: x += proj(attn(LN1(x))) =================
    if constexpr (CW == 2) {
#pragma unroll
      for (int nt = 0; nt < CT; ++nt)
        if (PK_HOOK_IO) PK_ROW_ST(acc[0][nt], ooff[0], nt);
    }
    if constexpr (MODE != 2) layer_norm(xn, T_LN1G, T_LN1B, p.eps1);
    if constexpr (MODE == 1) {                       // a head's proj partial alone: the accumulator starts from zero
#pragma unroll
      for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int nt = 0; nt < CT; ++nt) acc[c][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    } else if constexpr (CW == 2 && MODE == 0) {
      add_vec(xn, T_BP);
    }
    PK_TRACE();   // LN1 done
    // VMEM schedule around a tile boundary (hooks issue one row instruction per fragment group).  With 32 row instructions per tile
    // (NC * CW == 2): the two LAST chunks of a tile carry 16 loads each, the two FIRST chunks of the next one 16 stores each, nothing
    // anywhere else.  Chunk c + 1's pieces were issued at the sync of chunk c - 2, so
    //     VMC(c) = 8 (pieces of chunk c + 2) + the hook instructions issued in [second half of c - 2, first half of c]:
    //     second to last: 8 + 8 = 16;  last: 8 + 16 + 8 = 32;  first: 8 + 8 + 16 + 8 = 40;  second: 40;  third: 8 + 8 + 16 = 32;  fourth: 8 + 8 = 16.
    // With 16 row instructions per tile (NC = CW = 1) only the second to last chunk carries loads and only the first one stores:
    //     second to last 16, last 8 + 16 = 24, first 8 + 8 + 8 = 24, second 8 + 16 = 24, third 8 + 8 = 16, fourth 8.
    // At level 0 the first four chunks of a tile are Q_0, K_0, V_0, P_0 and the last two W2_14, W2_15; at level 1 (four chunks per head
    // matrix, two per W1_j / W2_j) they are the four chunks of Q_0 and the two of W2_31.
    // At level 1 (CW = 2: four chunks per head matrix, two per W1_j / W2_j) the last two chunks (of W2_31) carry the 32 loads in the same
    // way, but the 32 stores of the finished rows are issued in ONE burst in front of the first LayerNorm (which is amortised over four
    // times the MFMA work per row there), and acc <- xn + b_proj follows the LayerNorm directly: acc and xn are never both alive across
    // the heads, so ONE instantiation serves all four (head 0 differs in two wait counts only, a scalar branch):
    //     first chunk: 8 + 8 + 16 + 32 = 64 (-> 63, the field's maximum: a stronger wait);  second: 8 + 8 + 32 = 48;  from the third on: 8.
    constexpr bool IO32 = NC * CW == 2;
    // (all of the above with 8 DMA pieces per wave and chunk; in general DPW + the hook instructions of the window)
    // (the split forms issue no row instruction inside the chunk loops: DPW everywhere)
    constexpr int VMC_E1 = MODE ? DPW : DPW + 8, VMC_E0 = MODE ? DPW : DPW + (IO32 ? 24 : 16);
    constexpr int VMC_T0 = MODE ? DPW : CW == 2 ? 63 : DPW + (IO32 ? 32 : 16), VMC_T1 = MODE ? DPW : CW == 2 ? 48 : DPW + (IO32 ? 32 : 16),
                  VMC_T2 = MODE ? DPW : CW == 2 ? DPW : DPW + (IO32 ? 24 : 8), VMC_T3 = MODE ? DPW : CW == 2 ? DPW : DPW + (IO32 ? 8 : 0);
    // FM: 1 = the tile's first head at compile time (level 0: head 0 is its own instantiation, with the row stores in its hooks),
    // 0 = not the first, 2 = level 1: the wait counts of head 0 chosen at run time (h is uniform: a scalar branch)
#define PK_SYNC_T(TP)                                                                                                       \
  {                                                                                                                         \
    constexpr int vf_ = (TP) == 0 ? VMC_T0 : (TP) == 1 ? VMC_T1 : (TP) == 2 ? VMC_T2 : (TP) == 3 ? VMC_T3 : PK_VMC0;        \
    if constexpr (FM == 0 || vf_ == PK_VMC0) PK_SYNC(PK_VMC0);                                                              \
    else if constexpr (FM == 1) PK_SYNC(vf_);                                                                               \
    else { if (h == 0) PK_SYNC(vf_); else PK_SYNC(PK_VMC0); }                                                               \
  }
#define PK_ST_HOOK_T(TP) { if constexpr (FM == 1 && 16 * (TP) < NC * CT) PK_ST_HOOK(TP) }
    auto head = [&](auto fm_tag, int h) __attribute__((always_inline)) {
      constexpr int FM = decltype(fm_tag)::value;
      const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
      f32x4 t[NC][DT];
      op8 qf[NC][HS], kf[NC][HS];
      f32x4 rb;                                     // relative-position bias of (head h, query slot q, key slots 4 g .. 4 g + 3); -inf = no such pair
      const uint32_t vrb_h = vrb + (uint32_t)h * 1024u;
      // ---------------- q^T = Wq_h a^T  (first head: the previous tile's rows leave in its shadow) ----------------
#pragma unroll
      for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) t[c][dt] = z4;
#define PK_Q_CHUNK(SUB) if constexpr ((SUB) < NQ) PK_CHUNK(PK_SYNC_T(SUB), 0, (void)0, (void)0, PK_MFMA_T(t, af, SUB), PK_ST_HOOK_T(SUB))
      PK_Q_CHUNK(0) PK_Q_CHUNK(1) PK_Q_CHUNK(2) PK_Q_CHUNK(3) PK_Q_CHUNK(4) PK_Q_CHUNK(5) PK_Q_CHUNK(6) PK_Q_CHUNK(7)
#if PD_PAIR_DEBUG
      if (h == 0) { for (int c = 0; c < NC; ++c) dump4(1, c, t[c][0], t[c][1], t[c][2], t[c][3]); }
#endif
#pragma unroll
      for (int c = 0; c < NC; ++c) {
#pragma unroll
        for (int s = 0; s < HS; ++s) qf[c][s] = pk_pack8(t[c][2 * s], t[c][2 * s + 1]);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) t[c][dt] = z4;
      }
      PK_TRACE();   // q done
      // ---------------- k^T = Wk_h a^T ----------------
#define PK_K_CHUNK(SUB)                                                                                                      \
  if constexpr ((SUB) == 0) PK_CHUNK(PK_SYNC_T(NQ + (SUB)), 1, PK_LDS_F4(rb, vrb_h, 0), PK_LANDED(rb), PK_MFMA_T(t, af, SUB), PK_ST_HOOK_T(NQ + (SUB))) \
  else if constexpr ((SUB) < NQ) PK_CHUNK(PK_SYNC_T(NQ + (SUB)), 0, (void)0, (void)0, PK_MFMA_T(t, af, SUB), PK_ST_HOOK_T(NQ + (SUB)))
      PK_K_CHUNK(0) PK_K_CHUNK(1) PK_K_CHUNK(2) PK_K_CHUNK(3) PK_K_CHUNK(4) PK_K_CHUNK(5) PK_K_CHUNK(6) PK_K_CHUNK(7)
      PK_DRAIN();
      PK_TRACE();   // k done
      // ---------------- S^T = K Q^T, softmax over the keys (registers + two row swaps) ----------------
      s16x4 pf[NC];
#pragma unroll
      for (int c = 0; c < NC; ++c) {
#if PD_PAIR_DEBUG
        if (h == 0) dump4(2, c, t[c][0], t[c][1], t[c][2], t[c][3]);
#endif
        f32x4 s4 = z4;
#pragma unroll
        for (int s = 0; s < HS; ++s) {
          kf[c][s] = pk_pack8(t[c][2 * s], t[c][2 * s + 1]);
          s4 = mfma_16x16x32(kf[c][s], qf[c][s], s4);
        }
        float sc[4], mx = -3.0e38f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          sc[r] = __builtin_fmaf(s4[r], p.scale, rb[r]);   // -inf: a padded slot, or a key of the group's other cuboid
          mx = fmaxf(mx, sc[r]);
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
        for (int dt = 0; dt < DT; ++dt) t[c][dt] = z4;
      }
      PK_TRACE();   // softmax done
      // ---------------- v = a Wv_h^T (plain product: lane = feature, 4 consecutive tokens -> the A operand of O^T = V^T P^T) -------
#define PK_V_CHUNK(SUB) if constexpr ((SUB) < NQ) PK_CHUNK(PK_SYNC_T(2 * NQ + (SUB)), 0, (void)0, (void)0, PK_MFMA_V(t, af, SUB), (void)0)
      PK_V_CHUNK(0) PK_V_CHUNK(1) PK_V_CHUNK(2) PK_V_CHUNK(3) PK_V_CHUNK(4) PK_V_CHUNK(5) PK_V_CHUNK(6) PK_V_CHUNK(7)
      PK_DRAIN();
      op8 of[NC][HS];
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        f32x4 o[DT];
#if PD_PAIR_DEBUG
        if (h == 0) dump4(5, c, t[c][0], t[c][1], t[c][2], t[c][3]);
#endif
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[dt] = mfma_16x16x16(pk_pack4(t[c][dt]), pf[c], z4);
#if PD_PAIR_DEBUG
        if (h == 0) dump4(6, c, o[0], o[1], o[2], o[3]);
#endif
#pragma unroll
        for (int s = 0; s < HS; ++s) of[c][s] = pk_pack8(o[2 * s], o[2 * s + 1]);
      }
      if constexpr (FM == 1) add_vec(xn, T_BP);   // the finished rows have left: acc <- x + b_proj, the accumulator of every head's proj
      PK_TRACE();   // v + PV done
      // ---------------- x^T += Wp[:, head h] O_h^T ----------------
#define PK_P_CHUNK(SUB) if constexpr ((SUB) < NQ) PK_CHUNK(PK_SYNC_T(3 * NQ + (SUB)), 0, (void)0, (void)0, PK_MFMA_OUT(of, SUB, NQ1), (void)0)
      PK_P_CHUNK(0) PK_P_CHUNK(1) PK_P_CHUNK(2) PK_P_CHUNK(3) PK_P_CHUNK(4) PK_P_CHUNK(5) PK_P_CHUNK(6) PK_P_CHUNK(7)
    };
    if constexpr (MODE == 1) {
      head(std::integral_constant<int, 0>{}, slice);        // this workgroup's head; FM = 0: no row traffic in its hooks, plain wait counts
    } else if constexpr (MODE == 0 && CW == 1) {
      head(std::integral_constant<int, 1>{}, 0);
#pragma unroll 1
      for (int h = 1; h < HEADS; ++h) head(std::integral_constant<int, 0>{}, h);
    } else if constexpr (MODE == 0) {
#pragma unroll 1
      for (int h = 0; h < HEADS; ++h) head(std::integral_constant<int, 2>{}, h);
    }
    PK_DRAIN();                                     // (a LayerNorm follows)
    PK_TRACE();   // attention done
    PK_STAMP_ATT();
    if constexpr (MODE == 1) {                       // the partial leaves to this head's slab; nothing else to do for the tile
#pragma unroll
      for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int nt = 0; nt < CT; ++nt) PK_ROW_ST(acc[c][nt], roff[c], nt);
      continue;
    }

    // ================= FFN: x += W2 gelu(W1 LN2(x) + b1) + b2 =================
    // (MODE 2: x' -- the partials summed onto x + b_proj at the tile boundary -- sits in acc, as it does here in MODE 0)
    layer_norm(acc, T_LN2G, T_LN2B, p.eps2);
    add_vec(acc, T_B2);
    if constexpr (MODE == 2) {
      // only slice 0's partial carries x' + b_2: a multiplication by 1 or 0 (a conditional assignment of the 128 accumulator registers
      // -- the rows sit in AGPRs here -- cost the kernel 188 spilled registers; a non-finite x' becomes NaN in every slice, as it should)
      const float keep = slice == 0 ? 1.0f : 0.0f;
#pragma unroll
      for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int nt = 0; nt < CT; ++nt) acc[c][nt] = acc[c][nt] * keep;
    }
    PK_TRACE();   // LN2 done
    // Order of the 64-wide hidden slices: W1_0, W1_1, (W2_j, W1_{j+2}) for j = 0..NJ-3, W2_{NJ-2}, W2_{NJ-1} (NW chunks each).  gelu(h_j)
    // has the chunks between W1_j and W2_j to itself, as three software-pipelined stages (polynomial | exp, +1 | rcp, mul) of one value
    // per fragment group: independent short chains beside the MFMAs instead of one 9-deep dependent chain per value (a lone wave hides
    // no VALU latency).
    const uint32_t vb1 = vtab + (uint32_t)(T_B1 * 4) + (MODE == 2 ? (uint32_t)slice * (uint32_t)(GG::HID / NSPL * 4) : 0u);   // (b1 of this slice's hidden units)
    f32x4 hc[NC][4], hn[NC][4], b1n[4];
    float ga[16 * NC], gd[16 * NC];
#define PK_HV(H, v) H[(v) >> 4][((v) >> 2) & 3][(v) & 3]
#define PK_GELU_GROUP(H, VB, NPER, GI)                                                                                    \
  if (!(PD_PAIR_ABLATE & 8)) _Pragma("unroll") for (int u_ = 0; u_ < (NPER); ++u_) {                                      \
    if ((GI) < 16) { const int v_ = (VB) + (GI) * (NPER) + u_; ga[v_] = pk_gelu_arg(PK_HV(H, v_)); }                      \
    if ((GI) >= 1 && (GI) < 17) { const int v_ = (VB) + ((GI) - 1) * (NPER) + u_; gd[v_] = 1.0f + __builtin_amdgcn_exp2f(ga[v_]); } \
    if ((GI) >= 2 && (GI) < 18) { const int v_ = (VB) + ((GI) - 2) * (NPER) + u_; PK_HV(H, v_) = PK_HV(H, v_) * __builtin_amdgcn_rcpf(gd[v_]); } \
  }
#define PK_GELU_TAIL(H, VB, NPER) { PK_GELU_GROUP(H, VB, NPER, 16) PK_GELU_GROUP(H, VB, NPER, 17) }
  // the same three stages for the 16 values of ONE group per wave, one value every SP fragment groups: stage k of value v runs in group
  // SP v + k of a span of chunks (G counts the groups across them).  With two MFMAs per group (one group per wave) nine more VALU
  // instructions in every group make the chunk issue bound; spread over all the chunks between W1_j and W2_j they disappear.
#define PK_GELU_SP(H, SP, G)                                                                                              \
  if (!(PD_PAIR_ABLATE & 8)) _Pragma("unroll") for (int k_ = 0; k_ < 3; ++k_) {                                           \
    const int g_ = (G) - k_;                                                                                              \
    if (g_ >= 0 && g_ % (SP) == 0 && g_ / (SP) < 16) {                                                                    \
      const int v_ = g_ / (SP);                                                                                           \
      if (k_ == 0) ga[v_] = pk_gelu_arg(PK_HV(H, v_));                                                                    \
      if (k_ == 1) gd[v_] = 1.0f + __builtin_amdgcn_exp2f(ga[v_]);                                                        \
      if (k_ == 2) PK_HV(H, v_) = PK_HV(H, v_) * __builtin_amdgcn_rcpf(gd[v_]);                                           \
    }                                                                                                                     \
  }
#define PK_B1_FETCH(ADDR) { PK_LDS_F4(b1n[0], ADDR, 0); PK_LDS_F4(b1n[1], ADDR, 64); PK_LDS_F4(b1n[2], ADDR, 128); PK_LDS_F4(b1n[3], ADDR, 192); }
#define PK_B1_LANDED() { PK_LANDED(b1n[0]); PK_LANDED(b1n[1]); PK_LANDED(b1n[2]); PK_LANDED(b1n[3]); }
#define PK_PACK_H(H)                                                                                                       \
  _Pragma("unroll") for (int c = 0; c < NC; ++c) { hfr[c][0] = pk_pack8(H[c][0], H[c][1]); hfr[c][1] = pk_pack8(H[c][2], H[c][3]); }
    // W1 slice into ACC (its first chunk: the next slice's b1 from BADDR, HOOK0 in the shadow); W2 slice from hfr (HOOK0 beside its first chunk)
#define PK_W1_SLICE(ACC, BADDR, HOOK_A, TAIL_A, HOOK_B, TAIL_B)                                                             \
  PK_CHUNK(PK_SYNC(PK_VMC0), 4, PK_B1_FETCH(BADDR), PK_B1_LANDED(), PK_MFMA_H(ACC, af, 0), HOOK_A)                           \
  TAIL_A;                                                                                                                  \
  if constexpr (NW >= 2) { PK_CHUNK(PK_SYNC(PK_VMC0), 0, (void)0, (void)0, PK_MFMA_H(ACC, af, 1), HOOK_B) TAIL_B; }          \
  if constexpr (NW == 4) {                                                                                                 \
    PK_CHUNK(PK_SYNC(PK_VMC0), 0, (void)0, (void)0, PK_MFMA_H(ACC, af, 2), (void)0)                                          \
    PK_CHUNK(PK_SYNC(PK_VMC0), 0, (void)0, (void)0, PK_MFMA_H(ACC, af, 3), (void)0)                                          \
  }
#define PK_W2_SLICE(HOOK_A, TAIL_A, HOOK_B, TAIL_B)                                                                        \
  PK_CHUNK(PK_SYNC(PK_VMC0), 0, (void)0, (void)0, PK_MFMA_OUT(hfr, 0, NW1), HOOK_A)                                         \
  TAIL_A;                                                                                                                  \
  if constexpr (NW >= 2) { PK_CHUNK(PK_SYNC(PK_VMC0), 0, (void)0, (void)0, PK_MFMA_OUT(hfr, 1, NW1), HOOK_B) TAIL_B; }       \
  if constexpr (NW == 4) {                                                                                                 \
    PK_CHUNK(PK_SYNC(PK_VMC0), 0, (void)0, (void)0, PK_MFMA_OUT(hfr, 2, NW1), (void)0)                                       \
    PK_CHUNK(PK_SYNC(PK_VMC0), 0, (void)0, (void)0, PK_MFMA_OUT(hfr, 3, NW1), (void)0)                                       \
  }
  // gelu of ONE group per wave (NC == 1) spread over a span of chunks: SP = 32 / 16 groups per value when the span is 2 / ... chunks
  static_assert(NW == 1 || NW == 2 || NW == 4, "hidden slices of one, two or four chunks");
  constexpr int SP1 = NW >= 2 ? 2 : 1;             // spans of one slice (W1_1, W2_{NJ-2}): the gelu of a slice is done within its first two chunks
  constexpr int SP2 = 2 * SP1;                     // spans of two slices (W2_j, W1_{j+2})
    // b1 of slice 0: plain wait (the fragment prologue in flight is older and simply lands first)
    PK_B1_FETCH(vb1)
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b1n[0]), "+v"(b1n[1]), "+v"(b1n[2]), "+v"(b1n[3]));
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int ht = 0; ht < 4; ++ht) hc[c][ht] = b1n[ht];
    const uint32_t vb1_1 = vb1 + 256u;
    // ---------------- h_0^T = W1_0 a^T + b1 (in its shadow: b1 of slice 1) ----------------
    PK_W1_SLICE(hc, vb1_1, (void)0, (void)0, (void)0, (void)0)
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int ht = 0; ht < 4; ++ht) hn[c][ht] = b1n[ht];
    PK_TRACE();   // W1_0 done
    // ---------------- h_1 beside the whole of gelu(h_0) (NC values per group); b1 of slice 2 ----------------
    const uint32_t vb1_2 = vb1 + 512u;
    if constexpr (NC == 2) {
      PK_W1_SLICE(hn, vb1_2, PK_GELU_GROUP(hc, 0, NC, gi), PK_GELU_TAIL(hc, 0, NC), (void)0, (void)0)
    } else if constexpr (NW == 1) {
      PK_W1_SLICE(hn, vb1_2, PK_GELU_SP(hc, 1, gi), { PK_GELU_SP(hc, 1, 16) PK_GELU_SP(hc, 1, 17) }, (void)0, (void)0)
    } else {
      PK_W1_SLICE(hn, vb1_2, PK_GELU_SP(hc, 2, gi), (void)0, PK_GELU_SP(hc, 2, 16 + gi), PK_GELU_SP(hc, 2, 32))
    }
    op8 hfr[NC][2];
    PK_PACK_H(hc)
    PK_TRACE();   // W1_1 done
    // invariant: hfr = gelu(h_j) as fragments, hn = h_{j+1} (pre-activation), b1n = b1 of slice j + 2
#pragma unroll 1
    for (int j = 0; j < NJL - 2; ++j) {
#pragma unroll
      for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int ht = 0; ht < 4; ++ht) hc[c][ht] = b1n[ht];
      const uint32_t vb1n = vb1 + (uint32_t)(j + 3 < NJL ? j + 3 : 0) * 256u;
      if constexpr (NC == 2 && PD_PAIR_GELU_BOTH) {
        // x^T += W2[:, slice j] gelu(h_j)^T   beside the first half of gelu(h_{j+1}) (group 0)
        PK_W2_SLICE(PK_GELU_GROUP(hn, 0, 1, gi), PK_GELU_TAIL(hn, 0, 1), (void)0, (void)0)
        // h_{j+2}^T = W1_{j+2} a^T + b1   beside the second half of gelu(h_{j+1}); b1 of slice j + 3
        PK_W1_SLICE(hc, vb1n, PK_GELU_GROUP(hn, 16, 1, gi), PK_GELU_TAIL(hn, 16, 1), (void)0, (void)0)
      } else if constexpr (NC == 2) {
        // x^T += W2[:, slice j] gelu(h_j)^T
        PK_W2_SLICE((void)0, (void)0, (void)0, (void)0)
        // h_{j+2}^T = W1_{j+2} a^T + b1   beside the whole of gelu(h_{j+1}) (NC values per group); b1 of slice j + 3
        PK_W1_SLICE(hc, vb1n, PK_GELU_GROUP(hn, 0, NC, gi), PK_GELU_TAIL(hn, 0, NC), (void)0, (void)0)
      } else if constexpr (!PD_PAIR_GELU_SPREAD) {
        PK_W2_SLICE((void)0, (void)0, (void)0, (void)0)
        PK_W1_SLICE(hc, vb1n, PK_GELU_SP(hn, 1, gi), { PK_GELU_SP(hn, 1, 16) PK_GELU_SP(hn, 1, 17) }, (void)0, (void)0)
      } else {
        // one group per wave: gelu(h_{j+1}) one value every SP2 groups across the chunks of W2_j and W1_{j+2}
        PK_W2_SLICE(PK_GELU_SP(hn, SP2, gi), (void)0, PK_GELU_SP(hn, SP2, 16 + gi), (void)0)
        PK_W1_SLICE(hc, vb1n, PK_GELU_SP(hn, SP2, 16 * NW + gi), (void)0, PK_GELU_SP(hn, SP2, 16 * NW + 16 + gi), (void)0)
        if constexpr (NW == 1) { PK_GELU_SP(hn, SP2, 32) }      // (two chunks: stage 3 of the last value falls behind them)
      }
      PK_PACK_H(hn)
#pragma unroll
      for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int ht = 0; ht < 4; ++ht) hn[c][ht] = hc[c][ht];
    }
    PK_TRACE();   // FFN loop done
    // W2_{NJ-2} beside the whole of gelu(h_{NJ-1}), then W2_{NJ-1}; the next tile's rows arrive beside the tile's last two chunks
    // (the LayerNorm fragments are dead: xn takes their registers)
    if constexpr (NW == 1) {
      if constexpr (NC == 2) {
        PK_CHUNK(PK_SYNC(VMC_E1), 0, (void)0, (void)0, PK_MFMA_OUT(hfr, 0, NW1), { PK_GELU_GROUP(hn, 0, NC, gi) PK_LD_HOOK(1) })
        PK_GELU_TAIL(hn, 0, NC)
      } else {
        PK_CHUNK(PK_SYNC(VMC_E1), 0, (void)0, (void)0, PK_MFMA_OUT(hfr, 0, NW1), { PK_GELU_SP(hn, 1, gi) PK_LD_HOOK(1) })
        PK_GELU_SP(hn, 1, 16) PK_GELU_SP(hn, 1, 17)
      }
      PK_PACK_H(hn)
      PK_CHUNK(PK_SYNC(VMC_E0), 0, (void)0, (void)0, PK_MFMA_OUT(hfr, 0, NW1), PK_LD_HOOK(0))
    } else {
      PK_W2_SLICE(PK_GELU_SP(hn, 2, gi), (void)0, PK_GELU_SP(hn, 2, 16 + gi), PK_GELU_SP(hn, 2, 32))
      PK_PACK_H(hn)
      // the tile's LAST two chunks carry the loads of the next tile's rows (four chunks per slice: two plain ones first)
      if constexpr (NW == 4) {
        PK_CHUNK(PK_SYNC(PK_VMC0), 0, (void)0, (void)0, PK_MFMA_OUT(hfr, 0, NW1), (void)0)
        PK_CHUNK(PK_SYNC(PK_VMC0), 0, (void)0, (void)0, PK_MFMA_OUT(hfr, 1, NW1), (void)0)
      }
      PK_CHUNK(PK_SYNC(VMC_E1), 0, (void)0, (void)0, PK_MFMA_OUT(hfr, NW - 2, NW1), PK_LD_HOOK(1))
      PK_CHUNK(PK_SYNC(VMC_E0), 0, (void)0, (void)0, PK_MFMA_OUT(hfr, NW - 1, NW1), PK_LD_HOOK(0))
    }
    PK_DRAIN();                                     // (the next tile's LayerNorm follows)
    PK_TRACE();   // FFN done
    PK_STAMP_END();
    if constexpr (MODE == 2) {
#pragma unroll
      for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int nt = 0; nt < CT; ++nt) PK_ROW_ST(acc[c][nt], roff[c], nt);
    }
  }
  // ---- the last tile's rows ----
  if constexpr (MODE == 0) {
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int nt = 0; nt < CT; ++nt) PK_ROW_ST(acc[c][nt], roff[c], nt);
  }
  // nothing of this workgroup may still be writing LDS when its allocation is handed to the next one
  asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
#endif
}

template <int NC, int CW, int NWV = 4, int MODE = 0, int WP = 1, int NSPL = 4>
static int launch_pair(const pd_pair_args_k& a, hipStream_t s) {
  using namespace pairk;
  constexpr int LDS_BYTES = G<CW>::LDS_BYTES;
  static bool attr_set_dev[PD_MAX_DEVICES];
  bool& attr_set = attr_set_dev[pd_cur_device()];
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)pair_kernel<NC, CW, NWV, MODE, WP, NSPL>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e != hipSuccess) {
      pd_set_error("pd_attn_ffn_pair: hipFuncSetAttribute(%d) failed: %s", LDS_BYTES, hipGetErrorString(e));
      return PD_ERR_LAUNCH;
    }
    attr_set = true;
  }
  // persistent: every workgroup takes the same number of tiles (the last few one less), one workgroup per CU
  // (split forms: PAIR_NSPL workgroups per tile -- blockIdx.y = the slice)
  const int ncu = pd_num_cus();
  constexpr int WPT = MODE == 2 ? NSPL : MODE == 1 ? PAIR_NSPL : 1;
  const int per_wg = (a.ntiles * WPT + ncu - 1) / ncu;
  const int grid = (a.ntiles + per_wg - 1) / per_wg;
  hipLaunchKernelGGL((pair_kernel<NC, CW, NWV, MODE, WP, NSPL>), dim3((unsigned)grid, (unsigned)WPT), dim3(NWV * 64), LDS_BYTES, s, a);
  PD_CHECK_LAUNCH();
  return PD_OK;
}

// x' = ((((x + b_proj) + A_0) + A_1) + A_2) + A_3 over the four attention partials of the split form, written over A_0: formed ONCE per row
// here instead of by each of the four hidden-quarter workgroups of MODE 2 (whose tile prologue then reads 128 KB instead of 640 KB: at
// 4 trajectories the 208 workgroups pulled 133 MB through HBM / MALL before their first chunk).  The order of the sum is MODE 2's own
// (sum_partials): bit-identical.
__global__ void __launch_bounds__(256) pair_split_xsum_kernel(const float4* __restrict__ x, const float4* __restrict__ bproj, float4* __restrict__ slabs,
                                                              int64_t n4, int c4) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const float4 xv = x[i], b = bproj[i % c4], s0 = slabs[i], s1 = slabs[i + n4], s2 = slabs[i + 2 * n4], s3 = slabs[i + 3 * n4];
    slabs[i] = make_float4(((((xv.x + b.x) + s0.x) + s1.x) + s2.x) + s3.x, ((((xv.y + b.y) + s0.y) + s1.y) + s2.y) + s3.y,
                           ((((xv.z + b.z) + s0.z) + s1.z) + s2.z) + s3.z, ((((xv.w + b.w) + s0.w) + s1.w) + s2.w) + s3.w);
  }
}

// out = ((P_0 + P_1) + P_2) + P_3 over the four FFN partials of the split form (P_0 carries x' + b_2): 16 B per thread, fixed order
__global__ void __launch_bounds__(256) pair_split_sum_kernel(const float4* __restrict__ slabs, float4* __restrict__ out, int64_t n4) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const float4 a = slabs[i], b = slabs[i + n4], c = slabs[i + 2 * n4], d = slabs[i + 3 * n4];
    out[i] = make_float4(((a.x + b.x) + c.x) + d.x, ((a.y + b.y) + c.y) + d.y, ((a.z + b.z) + c.z) + d.z, ((a.w + b.w) + c.w) + d.w);
  }
}

#ifndef PD_PAIR_BIG_FORM
#define PD_PAIR_BIG_FORM 8                       // the form of the 128-row tiles: 2 (four waves x two groups) or 8 (eight waves x one group)
#endif
#if PD_PAIR_DEBUG && !PD_IS_F16
extern "C" float* pd_pair_dbg_buf = nullptr;    // (debugging builds only -- never part of the shipped library: scripts/debug_pair.py)
extern "C" int pd_pair_dbg_stage = 0;
#elif PD_PAIR_DEBUG
extern "C" float* pd_pair_dbg_buf;
extern "C" int pd_pair_dbg_stage;
#endif

#if !PD_IS_F16
extern "C" int pd_attn_ffn_pair_supported(int C, int heads, int hidden, int vol, int act) {
  return ((C == 256 && hidden == 1024) || (C == 512 && hidden == 2048)) && heads == 4 && vol >= 1 && vol <= 16 && act == PD_ACT_GELU;
}

// cuboids of one 16-slot group (the relative-position table of `vecs` is laid out for the same number: packing.pack_pair_vecs)
extern "C" int pd_attn_ffn_pair_cuboids_per_group(int vol) { return vol >= 1 && 2 * vol <= 16 ? 2 : 1; }
extern "C" int pd_f16_attn_ffn_pair(const float*, float*, const void*, const float*, const int32_t*, const int32_t*, int, int, int, int, int, float, float,
                                    float, const pd_call_opts*, pd_stream_t);
#else
extern "C" int pd_attn_ffn_pair_cuboids_per_group(int vol);
#endif

// argument checks + the kernel's argument struct, shared by the one-launch and the split form
static int pair_fill_args(pd_pair_args_k& a, const float* x, float* out, const void* wstream, const float* vecs, const int32_t* tok_index,
                          const int32_t* tok_affine, int B, int ntok, int nc, int vol, int units, float scale, float eps_attn, float eps_ffn,
                          const pd_call_opts* opts) {
  using namespace pairk;
  PD_CHECK_ARG(x && out && wstream && vecs, "pd_attn_ffn_pair: null pointer");
  PD_CHECK_ARG(units == 256 || units == 512, "pd_attn_ffn_pair: units %d (256 or 512)", units);
  PD_CHECK_ARG(B > 0 && ntok > 0 && (int64_t)B * ntok < (1ll << 31), "pd_attn_ffn_pair: bad sizes");
  PD_CHECK_ARG(nc > 0 && vol >= 1 && vol <= 16, "pd_attn_ffn_pair: cuboid volume %d not in 1..16", vol);
  PD_CHECK_ARG(tok_index || (tok_affine && tok_affine[0] > 0), "pd_attn_ffn_pair: neither a token table nor its affine form");
  PD_CHECK_ARG((int64_t)B * ntok * (units * 4) < 0xFFFFF000ll, "pd_attn_ffn_pair: x larger than a 4 GiB buffer descriptor");
  a.x = x; a.out = out; a.wstream = wstream; a.vecs = vecs; a.tok_index = tok_index;
  a.slab_in = nullptr; a.slab_out = nullptr; a.wstream2 = nullptr; a.w2bytes = 0;
  a.scale = scale; a.eps1 = eps_attn; a.eps2 = eps_ffn;
  a.wbytes = (uint32_t)((units == 256 ? G<1>::CH_ALL : G<2>::CH_ALL) * CHUNK) * (uint32_t)((opts && opts->w_fold) ? 2 : 1);
  a.xbytes = (uint32_t)((int64_t)B * ntok * (units * 4));
  a.trace = opts ? opts->trace : nullptr;          // (clock stamps: -DPD_PAIR_TRACE=1 / -DPD_PAIR_DEBUG=1 builds only)
#if PD_PAIR_DEBUG
  a.dbg_buf = pd_pair_dbg_buf;
  a.dbg_stage = pd_pair_dbg_stage;
#else
  a.dbg_buf = nullptr;
  a.dbg_stage = 0;
#endif
  a.B = B; a.ntok = ntok; a.nc = nc; a.vol = vol;
  a.pack = pd_attn_ffn_pair_cuboids_per_group(vol);
  a.aff_on = (tok_affine && tok_affine[0] > 0) ? 1 : 0;
  a.aff_ninner = a.aff_on ? tok_affine[0] : 1;
  a.aff_outer = a.aff_on ? tok_affine[1] : 0;
  a.aff_inner = a.aff_on ? tok_affine[2] : 0;
  a.aff_slot = a.aff_on ? tok_affine[3] : 0;
  a.gps = (nc + a.pack - 1) / a.pack;
  return PD_OK;
}

extern "C" int PD_ENTRY(attn_ffn_pair)(const float* x, float* out, const void* wstream, const float* vecs, const int32_t* tok_index,
                                       const int32_t* tok_affine, int B, int ntok, int nc, int vol, int units, float scale, float eps_attn,
                                       float eps_ffn, const pd_call_opts* opts, pd_stream_t stream) {
  using namespace pairk;
  PD_FORWARD_F16(PD_OPTS_F16(opts), pd_f16_attn_ffn_pair(x, out, wstream, vecs, tok_index, tok_affine, B, ntok, nc, vol, units, scale, eps_attn,
                                                         eps_ffn, opts, stream));
  pd_pair_args_k a;
  const int rc = pair_fill_args(a, x, out, wstream, vecs, tok_index, tok_affine, B, ntok, nc, vol, units, scale, eps_attn, eps_ffn, opts);
  if (rc != PD_OK) return rc;
  const int64_t groups = (int64_t)B * a.gps;
  const bool fold = opts && opts->w_fold;            // W_hi + W_lo chunks in the stream (twice as many): two products per k-step
  if (units == 512) {                               // one group per wave: 64-row tiles
    a.ntiles = (int)((groups + 3) / 4);
    return fold ? launch_pair<1, 2, 4, 0, 2>(a, (hipStream_t)stream) : launch_pair<1, 2>(a, (hipStream_t)stream);
  }
  // 128-row tiles once that leaves no CU idle -- as two groups per wave (four waves, every weight fragment feeds two MFMAs) or as
  // eight waves of one group (two waves per SIMD, 256 registers each: one wave's LayerNorm / softmax / GELU / row traffic runs beside
  // the other's MFMAs; opts->pair_form = 8); below that ONE group per wave and four waves (64-row tiles): twice the workgroups,
  // half the MFMAs per streamed chunk -- the small-batch form
  const int force = opts ? opts->pair_form : 0;      // A/B: 1 / 2 = 16-slot groups per wave (four waves), 8 = eight waves of one group
  const int nc_wave = force ? force : ((groups + 7) / 8 > pd_num_cus() / 2 ? PD_PAIR_BIG_FORM : 1);
  PD_CHECK_ARG(!(fold && nc_wave == 2), "pd_attn_ffn_pair: folded weights (w_fold) run one 16-slot group per wave (pair_form 1 or 8)");
  if (nc_wave == 8) {
    a.ntiles = (int)((groups + 7) / 8);
    return fold ? launch_pair<1, 1, 8, 0, 2>(a, (hipStream_t)stream) : launch_pair<1, 1, 8>(a, (hipStream_t)stream);
  }
  a.ntiles = (int)((groups + 4 * nc_wave - 1) / (4 * nc_wave));
  if (fold) return launch_pair<1, 1, 4, 0, 2>(a, (hipStream_t)stream);
  return nc_wave == 2 ? launch_pair<2, 1>(a, (hipStream_t)stream) : launch_pair<1, 1>(a, (hipStream_t)stream);
}

// ---- the pair as two tile launches + two row sums for small grids (few trajectories per launch): four workgroups per 64-row tile -----------------------
#if !PD_IS_F16
extern "C" int64_t pd_attn_ffn_pair_split_ws_floats(int B, int ntok, int units) { return (int64_t)2 * PAIR_NSPL * B * ntok * units; }
extern "C" int pd_f16_attn_ffn_pair_split(const float*, float*, const void*, const void*, const float*, const int32_t*, const int32_t*, int, int, int, int,
                                          int, float, float, float, float*, int64_t, const pd_call_opts*, pd_stream_t);
#else
extern "C" int64_t pd_attn_ffn_pair_split_ws_floats(int B, int ntok, int units);
#endif

extern "C" int PD_ENTRY(attn_ffn_pair_split)(const float* x, float* out, const void* wstream, const void* wffn_split, const float* vecs,
                                             const int32_t* tok_index, const int32_t* tok_affine, int B, int ntok, int nc, int vol, int units,
                                             float scale, float eps_attn, float eps_ffn, float* ws, int64_t ws_floats, const pd_call_opts* opts,
                                             pd_stream_t stream) {
  using namespace pairk;
  PD_FORWARD_F16(PD_OPTS_F16(opts), pd_f16_attn_ffn_pair_split(x, out, wstream, wffn_split, vecs, tok_index, tok_affine, B, ntok, nc, vol, units,
                                                               scale, eps_attn, eps_ffn, ws, ws_floats, opts, stream));
  pd_pair_args_k a;
  const int rc = pair_fill_args(a, x, out, wstream, vecs, tok_index, tok_affine, B, ntok, nc, vol, units, scale, eps_attn, eps_ffn, opts);
  if (rc != PD_OK) return rc;
  PD_CHECK_ARG(wffn_split && ws, "pd_attn_ffn_pair_split: null pointer");
  // (built for the level-1 blocks: at units 256 a 64-row tile streams 1.5 MB and small grids already have a tile per CU from 4 trajectories on)
  PD_CHECK_ARG(units == 512, "pd_attn_ffn_pair_split: units %d (the split form is built for units 512)", units);
  PD_CHECK_ARG(ws_floats >= pd_attn_ffn_pair_split_ws_floats(B, ntok, units), "pd_attn_ffn_pair_split: workspace of %lld floats, %lld needed",
               (long long)ws_floats, (long long)pd_attn_ffn_pair_split_ws_floats(B, ntok, units));
  PD_CHECK_ARG((int64_t)PAIR_NSPL * a.xbytes < 0xFFFFF000ll, "pd_attn_ffn_pair_split: the four slabs exceed a 4 GiB buffer descriptor");
  PD_CHECK_ARG((((uintptr_t)ws | (uintptr_t)out) & 15) == 0, "pd_attn_ffn_pair_split: ws / out must be 16 B aligned");
  const int64_t groups = (int64_t)B * a.gps, n = (int64_t)B * ntok * units;
  a.ntiles = (int)((groups + 3) / 4);                // one group per wave, four waves: 64-row tiles
  float* slab_a = ws;                                // the four attention partials
  float* slab_f = ws + PAIR_NSPL * n;                // the four FFN partials
  hipStream_t s = (hipStream_t)stream;
  // 1: (tile, head) -> proj partial of that head
  const bool fold = opts && opts->w_fold;
  a.slab_out = slab_a;
  int r = fold ? launch_pair<1, 2, 4, 1, 2>(a, s) : launch_pair<1, 2, 4, 1>(a, s);
  if (r != PD_OK) return r;
  // 2a: x' = x + b_proj + the four partials in order, once per row (over partial 0)
  const int64_t n4 = n / 4;
  const unsigned sum_blocks = (unsigned)std::min<int64_t>((n4 + 255) / 256, 2048);
  PD_CHECK_ARG((((uintptr_t)x | (uintptr_t)vecs) & 15) == 0, "pd_attn_ffn_pair_split: x / vecs must be 16 B aligned");
  hipLaunchKernelGGL(pair_split_xsum_kernel, dim3(sum_blocks), dim3(256), 0, s, (const float4*)x, (const float4*)(vecs + G<2>::T_BP), (float4*)slab_a, n4,
                     G<2>::C / 4);
  PD_CHECK_LAUNCH();
  // 2b: (tile, quarter) -> LayerNorm of x', the quarter's FFN partial (quarter 0 carries x' + b_2)
  a.x = slab_a;
  a.slab_in = nullptr;
  a.slab_out = slab_f;
  a.wstream2 = wffn_split;
  a.w2bytes = (uint32_t)(2 * G<2>::NJ * G<2>::NW * CHUNK) * (fold ? 2u : 1u);
  r = fold ? launch_pair<1, 2, 4, 2, 2>(a, s) : launch_pair<1, 2, 4, 2>(a, s);
  if (r != PD_OK) return r;
  // 3: out = the four FFN partials in order
  hipLaunchKernelGGL(pair_split_sum_kernel, dim3(sum_blocks), dim3(256), 0, s, (const float4*)slab_f, (float4*)out, n4);
  PD_CHECK_LAUNCH();
  return PD_OK;
}

// ---- the FFN alone on the pair kernel's FFN half (MODE 2, NSPL = 1): PositionwiseFFN.forward (cuboid_transformer.py:182-208) for rows whose
//      attention layer the pair kernel cannot take ----------------------------------------------------------------------------------------------
#if !PD_IS_F16
extern "C" int pd_ffn_rows_supported(int C, int hidden, int act) { return ((C == 256 && hidden == 1024) || (C == 512 && hidden == 2048)) && act == PD_ACT_GELU; }
extern "C" int pd_f16_ffn_rows(const float*, float*, const void*, const float*, int64_t, int, float, const pd_call_opts*, pd_stream_t);
#endif

extern "C" int PD_ENTRY(ffn_rows)(const float* x, float* out, const void* wffn, const float* vecs, int64_t rows, int units, float eps,
                                  const pd_call_opts* opts, pd_stream_t stream) {
  using namespace pairk;
  PD_FORWARD_F16(PD_OPTS_F16(opts), pd_f16_ffn_rows(x, out, wffn, vecs, rows, units, eps, opts, stream));
  PD_CHECK_ARG(x && out && wffn && vecs, "pd_ffn_rows: null pointer");
  PD_CHECK_ARG(units == 256 || units == 512, "pd_ffn_rows: units %d (256 or 512)", units);
  PD_CHECK_ARG(rows > 0 && rows < (1ll << 31) && rows * (int64_t)(units * 4) < 0xFFFFF000ll, "pd_ffn_rows: bad row count / x larger than a 4 GiB buffer descriptor");
  PD_CHECK_ARG(!(opts && opts->w_fold), "pd_ffn_rows: no folded-weight form");
  pd_pair_args_k a;
  // rows as 16-slot groups of consecutive rows: group c, slot s -> row 16 c + s (the affine token form with one "sample" holding every row);
  // a last partial group gets out-of-range rows (loads return zero, stores are dropped)
  const int32_t aff[4] = {1, 16, 0, 1};
  const int groups = (int)((rows + 15) / 16);
  const int rc = pair_fill_args(a, x, out, wffn, vecs, nullptr, aff, 1, (int)rows, groups, 16, units, 1.0f, eps, eps, opts);
  if (rc != PD_OK) return rc;
  a.slab_in = nullptr;
  a.slab_out = out;                                  // slice 0 of one: the result itself
  a.wstream2 = wffn;
  a.w2bytes = (uint32_t)(2 * (units == 256 ? G<1>::NJ * G<1>::NW : G<2>::NJ * G<2>::NW) * CHUNK);
  hipStream_t s = (hipStream_t)stream;
  if (units == 512) {
    a.ntiles = (groups + 3) / 4;
    return launch_pair<1, 2, 4, 2, 1, 1>(a, s);
  }
  // 128-row tiles (eight waves, two per SIMD) once they leave no CU idle, else 64-row tiles
  if ((groups + 7) / 8 > pd_num_cus() / 2) {
    a.ntiles = (groups + 7) / 8;
    return launch_pair<1, 1, 8, 2, 1, 1>(a, s);
  }
  a.ntiles = (groups + 3) / 4;
  return launch_pair<1, 1, 4, 2, 1, 1>(a, s);
}

}  // namespace PD_NS
