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
#include "common.h"

#define BLDS16(rsrc, ldsptr, voff, soff) \
  __builtin_amdgcn_raw_ptr_buffer_load_lds((rsrc), (__attribute__((address_space(3))) void*)(ldsptr), 16, (voff), (soff), 0, 0)

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

namespace pairk {
constexpr int C = 256, HEADS = 4, HID = 1024;
constexpr int PF = 4, PFN = PF + 1;                // weight fragments in flight (LDS latency ~ 4 x 32 MFMA clocks)
constexpr int CHUNK = 32768, NSLOT = 4, NFRAG = 32;
constexpr int DMA_PER_WAVE = CHUNK / 4 / 1024;     // 8 x 1 KB per wave per chunk
// fp32 tables in LDS (float offsets) == layout of the `vecs` argument
constexpr int T_LN1G = 0, T_LN1B = 256, T_BP = 512, T_LN2G = 768, T_LN2B = 1024, T_B2 = 1280, T_B1 = 1536, T_RB = 2560, T_FLOATS = 3584;
constexpr int RING_OFF = 16384;
constexpr int LDS_BYTES = RING_OFF + NSLOT * CHUNK;   // 147456
constexpr int CH_ATTN = 16, CH_ALL = 48;           // chunks per tile: 4 heads x (q, k, v, proj); + 16 x (W1_j, W2_j)
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
  uint32_t wbytes;
  unsigned long long* trace;
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
__device__ __forceinline__ bf16x8 pk_pack8(const f32x4& a, const f32x4& b) {
  u32x4 r = {pack_bf16x2(a[0], a[1]), pack_bf16x2(a[2], a[3]), pack_bf16x2(b[0], b[1]), pack_bf16x2(b[2], b[3])};
  return __builtin_bit_cast(bf16x8, r);
}
__device__ __forceinline__ s16x4 pk_pack4(const f32x4& a) {
  u32x2 r = {pack_bf16x2(a[0], a[1]), pack_bf16x2(a[2], a[3])};
  return __builtin_bit_cast(s16x4, r);
}
// GELU as x * sigmoid(x (a + b x^2 + c x^4)), x^2 clamped where the sigmoid is saturated (the quartic term would turn the polynomial
// over at |x| ~ 11).  Coefficients: minimax fit against 0.5 x (1 + erf(x / sqrt 2)) on [-9, 9], max abs error 2.5e-5; pre-multiplied by
// -log2(e) for v_exp_f32.
__device__ __forceinline__ float pk_gelu(float x) {
  constexpr float L2E = -1.4426950408889634f;
  const float x2 = fminf(x * x, 52.0f);
  float pl = fmaf(x2, L2E * -7.03034059e-04f, L2E * 7.40112943e-02f);
  pl = fmaf(x2, pl, L2E * 1.59501577f);
  const float e = __builtin_amdgcn_exp2f(x * pl);
  return x * __builtin_amdgcn_rcpf(1.0f + e);
}

#define PK_WLD(dst, addr, imm) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(imm))
// wait until at most N LDS operations are outstanding; the fragment about to be consumed is tied to the wait (in / out operand), so
// that no MFMA reading it can be moved in front of the wait
#define PK_WAIT_FRAG(frag, N) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(frag) : "n"(N))
#define PK_LDS_F4(dst, addr, imm) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(imm))

// PARTS: bit 0 attention, bit 1 FFN
template <int PARTS>
__global__ void __launch_bounds__(256, 1) pair_kernel(const pd_pair_args_k p) {
#if defined(__HIP_DEVICE_COMPILE__)
  using namespace pairk;
  constexpr bool DO_ATTN = (PARTS & 1) != 0, DO_FFN = (PARTS & 2) != 0;
  constexpr int CH_FIRST = DO_ATTN ? 0 : CH_ATTN, CH_END = DO_FFN ? CH_ALL : CH_ATTN;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane & 15, g = lane >> 4;

  // ---- tables -> LDS (before any DMA is in flight: a compiler-visible LDS store behind a DMA would drain it) ----
  for (int i = tid; i < T_FLOATS; i += 256) ((float*)smem)[i] = p.vecs[i];
  __syncthreads();

  // ---- weight stream: chunk ids CH_FIRST .. CH_END-1 cyclically, chunk number n -> ring slot n & 3 ----
  const auto rW = __builtin_amdgcn_make_buffer_rsrc((void*)p.wstream, 0, p.wbytes, 0x00020000);
  int n_issued = 0, kid = CH_FIRST;
  const uint32_t dma_voff = (uint32_t)lane * 16u;
  auto issue_next = [&]() {
    char* d = smem + RING_OFF + (n_issued & (NSLOT - 1)) * CHUNK + wave * (DMA_PER_WAVE * 1024);
    const uint32_t so = (uint32_t)kid * CHUNK + (uint32_t)wave * (DMA_PER_WAVE * 1024);
#pragma unroll
    for (int i = 0; i < DMA_PER_WAVE; ++i) BLDS16(rW, d + i * 1024, dma_voff, so + i * 1024);
    ++n_issued;
    kid = (kid + 1 == CH_END) ? CH_FIRST : kid + 1;
  };
  issue_next();
  issue_next();
  issue_next();

  const uint32_t vbase = (uint32_t)(uintptr_t)(smem + RING_OFF) + (uint32_t)lane * 16u;   // fragment reads: lane-linear 16 B
  const uint32_t vtab = (uint32_t)(uintptr_t)smem + (uint32_t)g * 16u;                     // fp32 tables: 4 floats at column 4 g
  const uint32_t vrb = (uint32_t)(uintptr_t)smem + (uint32_t)((T_RB + q * 16 + 4 * g) * 4);

  int cc = 0;                                       // chunks consumed by this workgroup
  bf16x8 w[PFN];                                    // fragment pipeline (runs on across chunks, tiles and phases)

  // in the middle of chunk cc: chunk cc+1 has landed for everybody, everybody is past chunk cc-1 -> its slot takes chunk cc+3
  auto mid_sync = [&]() {
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(DMA_PER_WAVE) : "memory");
    issue_next();
  };
  // One chunk = 32 fragments.  extra(): NEXTRA other LDS reads issued at the start (they land before iteration PF).
  // body(i, frag): the MFMAs of fragment i (+ interleaved VALU work).
#define PK_CHUNK(NEXTRA, EXTRA_STMT, BODY_STMT)                                                   \
  {                                                                                               \
    const uint32_t va_ = vbase + (uint32_t)(cc & (NSLOT - 1)) * CHUNK;                            \
    const uint32_t vn_ = vbase + (uint32_t)((cc + 1) & (NSLOT - 1)) * CHUNK;                      \
    _Pragma("unroll") for (int i = 0; i < NFRAG; ++i) {                                           \
      if (i == 0) { EXTRA_STMT; }                                                                 \
      if (i == NFRAG / 2) mid_sync();                                                             \
      if (i + PF < NFRAG) PK_WLD(w[(i + PF) % PFN], va_, (i + PF) * 1024);                        \
      else PK_WLD(w[(i + PF) % PFN], vn_, (i + PF - NFRAG) * 1024);                               \
      PK_WAIT_FRAG(w[i % PFN], PF + (i < PF ? (NEXTRA) : 0));                                     \
      __builtin_amdgcn_sched_barrier(0);                                                          \
      { const bf16x8 wf = w[i % PFN]; BODY_STMT; }                                                \
      __builtin_amdgcn_sched_barrier(0);                                                          \
    }                                                                                             \
    ++cc;                                                                                         \
  }

  bool first = true;
  for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
    // ---- rows of this wave's two cuboids: lane = (slot q, column group g) ----
    const float* xrow[2];
    float* orow[2];
    bool valid[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int64_t gc = (int64_t)tile * 8 + wave * 2 + c;
      int row = -1;
      if (gc < (int64_t)p.B * p.nc && q < p.vol) {
        const int b = (int)(gc / p.nc), cu = (int)(gc - (int64_t)b * p.nc);
        const int tok = p.aff_on ? (cu / p.aff_ninner) * p.aff_outer + (cu % p.aff_ninner) * p.aff_inner + q * p.aff_slot
                                 : p.tok_index[cu * p.vol + q];
        if (tok >= 0 && tok < p.ntok) row = b * p.ntok + tok;
      }
      valid[c] = row >= 0;
      xrow[c] = p.x + (int64_t)(row < 0 ? 0 : row) * C + 4 * g;
      orow[c] = p.out + (int64_t)(row < 0 ? 0 : row) * C + 4 * g;
    }
    // ---- x rows -> acc[c][nt] = x[row][16 nt + 4 g .. +3]: the MFMA C layout of every transposed product below ----
    f32x4 acc[2][16];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int nt = 0; nt < 16; ++nt) {
        acc[c][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (valid[c]) acc[c][nt] = *(const f32x4*)(xrow[c] + nt * 16);
      }
    if (first) {
      // chunks 0..2 landed (the x loads above are younger: the compiler's own wait for them covers the DMA), visible to every wave
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
#pragma unroll
      for (int i = 0; i < PF; ++i) PK_WLD(w[i], vbase, i * 1024);
      first = false;
    }

    bf16x8 af[2][8];                                // LayerNorm output as B-operand fragments: [cuboid][k-step of 32]
    // LayerNorm over the 256 columns of a row (64 in this lane, the rest in lanes q + 16 g'), -> af
    auto layer_norm = [&](int t_gamma, int t_beta, float eps) {
      float mean[2], rstd[2];
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        float s = 0.f;
#pragma unroll
        for (int nt = 0; nt < 16; ++nt) s += (acc[c][nt][0] + acc[c][nt][1]) + (acc[c][nt][2] + acc[c][nt][3]);
        mean[c] = pk_rows4_sum(s) * (1.0f / C);
        float v = 0.f;
#pragma unroll
        for (int nt = 0; nt < 16; ++nt) {
          const float d0 = acc[c][nt][0] - mean[c], d1 = acc[c][nt][1] - mean[c], d2 = acc[c][nt][2] - mean[c], d3 = acc[c][nt][3] - mean[c];
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
        for (int c = 0; c < 2; ++c) {
          f32x4 y[2];
#pragma unroll
          for (int hf = 0; hf < 2; ++hf)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              y[hf][r] = (acc[c][2 * ks + hf][r] - mean[c]) * rstd[c] * gb[ks & 1][hf][r] + gb[ks & 1][2 + hf][r];
          af[c][ks] = pk_pack8(y[0], y[1]);
        }
      }
    };
    // acc[c][nt] += table[16 nt + 4 g .. +3]  (proj / FFN-2 bias: the accumulator starts from residual + bias)
    auto add_vec = [&](int t_off) {
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        f32x4 bv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) PK_LDS_F4(bv[i], vtab, (t_off + 16 * (half * 8 + i)) * 4);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bv[0]), "+v"(bv[1]), "+v"(bv[2]), "+v"(bv[3]), "+v"(bv[4]), "+v"(bv[5]), "+v"(bv[6]), "+v"(bv[7]));
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int c = 0; c < 2; ++c) acc[c][half * 8 + i] += bv[i];
      }
    };

    if constexpr (DO_ATTN) {
      layer_norm(T_LN1G, T_LN1B, p.eps1);
      add_vec(T_BP);
#pragma unroll 1
      for (int h = 0; h < HEADS; ++h) {
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
        f32x4 t[2][4];
        bf16x8 qf[2][2], kf[2][2];
        f32x4 rb;                                   // relative-position bias of (head h, query q, keys 4 g .. 4 g + 3)
        const uint32_t vrb_h = vrb + (uint32_t)h * 1024u;
        // ---------------- q^T = Wq_h a^T ----------------
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) t[c][dt] = z4;
        PK_CHUNK(0, (void)0, {
          t[0][i & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, af[0][i >> 2], t[0][i & 3], 0, 0, 0);
          t[1][i & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, af[1][i >> 2], t[1][i & 3], 0, 0, 0);
        })
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          qf[c][0] = pk_pack8(t[c][0], t[c][1]);
          qf[c][1] = pk_pack8(t[c][2], t[c][3]);
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) t[c][dt] = z4;
        }
        // ---------------- k^T = Wk_h a^T ----------------
        PK_CHUNK(1, PK_LDS_F4(rb, vrb_h, 0), {
          t[0][i & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, af[0][i >> 2], t[0][i & 3], 0, 0, 0);
          t[1][i & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, af[1][i >> 2], t[1][i & 3], 0, 0, 0);
        })
        // ---------------- S^T = K Q^T, softmax over the keys (registers + two row swaps) ----------------
        s16x4 pf[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
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
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) t[c][dt] = z4;
        }
        // ---------------- v = a Wv_h^T (plain product: lane = feature, 4 consecutive tokens -> the A operand of O^T = V^T P^T) -------
        PK_CHUNK(0, (void)0, {
          t[0][i & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0][i >> 2], wf, t[0][i & 3], 0, 0, 0);
          t[1][i & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[1][i >> 2], wf, t[1][i & 3], 0, 0, 0);
        })
        bf16x8 of[2][2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          f32x4 o[4];
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) o[dt] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(pk_pack4(t[c][dt]), pf[c], z4, 0, 0, 0);
          of[c][0] = pk_pack8(o[0], o[1]);
          of[c][1] = pk_pack8(o[2], o[3]);
        }
        // ---------------- x^T += Wp[:, head h] O_h^T ----------------
        PK_CHUNK(0, (void)0, {
          acc[0][i & 15] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, of[0][i >> 4], acc[0][i & 15], 0, 0, 0);
          acc[1][i & 15] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, of[1][i >> 4], acc[1][i & 15], 0, 0, 0);
        })
      }
    }

    if constexpr (DO_FFN) {
      layer_norm(T_LN2G, T_LN2B, p.eps2);
      add_vec(T_B2);
      const uint32_t vb1 = vtab + (uint32_t)(T_B1 * 4);
      f32x4 hc[2][4], hn[2][4], b1n[4];
      // b1 of chunk 0: plain wait (the fragment prologue in flight is older and simply lands first)
#pragma unroll
      for (int ht = 0; ht < 4; ++ht) PK_LDS_F4(b1n[ht], vb1, ht * 64);
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b1n[0]), "+v"(b1n[1]), "+v"(b1n[2]), "+v"(b1n[3]));
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int ht = 0; ht < 4; ++ht) hc[c][ht] = b1n[ht];
      // ---------------- h_0^T = W1_0 a^T + b1 ; the loads in its shadow fetch b1 of chunk 1 ----------------
      PK_CHUNK(4, { PK_LDS_F4(b1n[0], vb1, 256); PK_LDS_F4(b1n[1], vb1, 256 + 64); PK_LDS_F4(b1n[2], vb1, 256 + 128); PK_LDS_F4(b1n[3], vb1, 256 + 192); }, {
        hc[0][i & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, af[0][i >> 2], hc[0][i & 3], 0, 0, 0);
        hc[1][i & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, af[1][i >> 2], hc[1][i & 3], 0, 0, 0);
      })
#pragma unroll 1
      for (int j = 0; j < HID / 64; ++j) {
        bf16x8 hfr[2][2];
        if (j + 1 < HID / 64) {
#pragma unroll
          for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int ht = 0; ht < 4; ++ht) hn[c][ht] = b1n[ht];
          // W1_{j+1} (MFMA) beside gelu(h_j) (VALU), one value per fragment; b1 of chunk j+2 fetched in its shadow
          const uint32_t vb1n = vb1 + (uint32_t)(j + 2 < HID / 64 ? j + 2 : 0) * 256u;
          PK_CHUNK(4, { PK_LDS_F4(b1n[0], vb1n, 0); PK_LDS_F4(b1n[1], vb1n, 64); PK_LDS_F4(b1n[2], vb1n, 128); PK_LDS_F4(b1n[3], vb1n, 192); }, {
            hn[0][i & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, af[0][i >> 2], hn[0][i & 3], 0, 0, 0);
            hn[1][i & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, af[1][i >> 2], hn[1][i & 3], 0, 0, 0);
            hc[i >> 4][(i >> 2) & 3][i & 3] = pk_gelu(hc[i >> 4][(i >> 2) & 3][i & 3]);
          })
        } else {
#pragma unroll
          for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int ht = 0; ht < 4; ++ht)
#pragma unroll
              for (int r = 0; r < 4; ++r) hc[c][ht][r] = pk_gelu(hc[c][ht][r]);
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          hfr[c][0] = pk_pack8(hc[c][0], hc[c][1]);
          hfr[c][1] = pk_pack8(hc[c][2], hc[c][3]);
        }
        // ---------------- x^T += W2[:, chunk j] gelu(h_j)^T ----------------
        PK_CHUNK(0, (void)0, {
          acc[0][i & 15] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, hfr[0][i >> 4], acc[0][i & 15], 0, 0, 0);
          acc[1][i & 15] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, hfr[1][i >> 4], acc[1][i & 15], 0, 0, 0);
        })
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int ht = 0; ht < 4; ++ht) hc[c][ht] = hn[c][ht];
      }
    }

    // ---- the rows go back (same lane -> (row, columns) map as the load) ----
#pragma unroll
    for (int c = 0; c < 2; ++c)
      if (valid[c]) {
#pragma unroll
        for (int nt = 0; nt < 16; ++nt) *(f32x4*)(orow[c] + nt * 16) = acc[c][nt];
      }
  }
  // nothing of this workgroup may still be writing LDS when its allocation is handed to the next one
  asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
#endif
}

template <int PARTS>
static int launch_pair(const pd_pair_args_k& a, hipStream_t s) {
  using namespace pairk;
  static bool attr_set_dev[PD_MAX_DEVICES];
  bool& attr_set = attr_set_dev[pd_cur_device()];
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)pair_kernel<PARTS>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e != hipSuccess) {
      pd_set_error("pd_attn_ffn_pair: hipFuncSetAttribute(%d) failed: %s", LDS_BYTES, hipGetErrorString(e));
      return PD_ERR_LAUNCH;
    }
    attr_set = true;
  }
  // persistent: every workgroup takes the same number of tiles (the last few one less), one workgroup per CU
  const int per_wg = (a.ntiles + 255) / 256;
  const int grid = (a.ntiles + per_wg - 1) / per_wg;
  hipLaunchKernelGGL((pair_kernel<PARTS>), dim3((unsigned)grid), dim3(256), LDS_BYTES, s, a);
  PD_CHECK_LAUNCH();
  return PD_OK;
}

extern "C" unsigned long long* pd_pair_trace = nullptr;

extern "C" int pd_attn_ffn_pair_supported(int C, int heads, int hidden, int vol, int act) {
  return C == 256 && heads == 4 && hidden == 1024 && vol >= 1 && vol <= 16 && act == PD_ACT_GELU;
}

extern "C" int pd_attn_ffn_pair(const float* x, float* out, const void* wstream, const float* vecs, const int32_t* tok_index,
                                const int32_t* tok_affine, int B, int ntok, int nc, int vol, float scale, float eps_attn, float eps_ffn,
                                int parts, pd_stream_t stream) {
  using namespace pairk;
  PD_CHECK_ARG(x && out && wstream && vecs, "pd_attn_ffn_pair: null pointer");
  PD_CHECK_ARG(parts >= 1 && parts <= 3, "pd_attn_ffn_pair: parts must be 1 (attention), 2 (FFN) or 3 (both)");
  PD_CHECK_ARG(B > 0 && ntok > 0 && (int64_t)B * ntok < (1ll << 31), "pd_attn_ffn_pair: bad sizes");
  pd_pair_args_k a;
  a.x = x; a.out = out; a.wstream = wstream; a.vecs = vecs; a.tok_index = tok_index;
  a.scale = scale; a.eps1 = eps_attn; a.eps2 = eps_ffn;
  a.wbytes = (uint32_t)(CH_ALL * CHUNK);
  a.trace = pd_pair_trace;
  if (parts & 1) {
    PD_CHECK_ARG(nc > 0 && vol >= 1 && vol <= 16, "pd_attn_ffn_pair: cuboid volume %d not in 1..16", vol);
    PD_CHECK_ARG(tok_index || (tok_affine && tok_affine[0] > 0), "pd_attn_ffn_pair: neither a token table nor its affine form");
    a.B = B; a.ntok = ntok; a.nc = nc; a.vol = vol;
    a.aff_on = (tok_affine && tok_affine[0] > 0) ? 1 : 0;
    a.aff_ninner = a.aff_on ? tok_affine[0] : 1;
    a.aff_outer = a.aff_on ? tok_affine[1] : 0;
    a.aff_inner = a.aff_on ? tok_affine[2] : 0;
    a.aff_slot = a.aff_on ? tok_affine[3] : 0;
  } else {
    // FFN alone is row-wise: "cuboids" of 16 consecutive rows
    const int64_t M = (int64_t)B * ntok;
    a.B = 1; a.ntok = (int)M; a.nc = (int)((M + 15) / 16); a.vol = 16;
    a.aff_on = 1; a.aff_ninner = a.nc; a.aff_outer = 0; a.aff_inner = 16; a.aff_slot = 1;
  }
  const int64_t cuboids = (int64_t)a.B * a.nc;
  a.ntiles = (int)((cuboids + 7) / 8);
  hipStream_t s = (hipStream_t)stream;
  if (parts == 3) return launch_pair<3>(a, s);
  if (parts == 1) return launch_pair<1>(a, s);
  return launch_pair<2>(a, s);
}
