/*
 * prediff_hip.h -- C ABI of libprediff_hip.so: hand-written HIP (gfx950 / CDNA4) kernels for the
 * PreDiff sampling hot path (Earthformer-UNet denoiser, KL-VAE, DDPM/DDIM step epilogue).
 *
 * The reference (gaozhihan/PreDiff) is pure Python/PyTorch and has no FFI: the boundary it offers is the
 * Python call convention of its nn.Modules.  Each entry point below replaces the ATen op
 * sequence of the cited reference function; the prediff_amd Python modules (same class names / ctor kwargs / state_dict
 * schema as the reference) are the only callers.  All pointers are caller-owned DEVICE pointers, layouts are
 * channels-last, no entry point allocates, every launch goes to the given stream and every function returns
 * 0 on success or a negative pd_status (the Python side raises).  Paths below are relative to
 * /root/reference/src/prediff/.
 */
#ifndef PREDIFF_HIP_H
#define PREDIFF_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* pd_stream_t;   /* hipStream_t */
typedef uint16_t pd_bf16;    /* raw bfloat16 bits */

enum pd_status { PD_OK = 0, PD_ERR_ARG = -1, PD_ERR_UNSUPPORTED = -2, PD_ERR_LAUNCH = -3 };
enum pd_act { PD_ACT_NONE = 0, PD_ACT_GELU = 1, PD_ACT_SILU = 2, PD_ACT_LEAKY = 3, PD_ACT_RELU = 4 };
/* Type of the 16-bit MFMA operands a call reads and writes (`pd_bf16` pointers then carry IEEE half bits): bfloat16 (8-bit mantissa, the
 * throughput engine) or IEEE half (11-bit significand at the same MFMA rate: the TF32-class engine -- the reference runs fp32 with
 * float32_matmul_precision "high", scripts/prediff/sevirlr/prediff_sevirlr_v1.yaml:63).  Weights must be packed in the same type.
 * The hi/lo split (fp32-class) forms exist for bfloat16 only. */
enum pd_operand { PD_OPERAND_BF16 = 0, PD_OPERAND_F16 = 1 };

/* Per-call options of the entry points that read / write 16-bit operands or have an A/B switch: passed on every call, NULL = all
 * defaults (bfloat16, production settings).  Nothing in this library is process-global: two engines in one process (the two-lane mode
 * runs two, each from its own host thread) cannot see each other's settings.  All-zero = defaults. */
typedef struct pd_call_opts {
  int32_t operand;                  /* enum pd_operand */
  int32_t attn_block_table_ids;     /* pd_attn_block_fused_ex: != 0 = load the token table even when an affine form is given (A/B; bit-identical) */
  int32_t ffn_rows128;              /* pd_ffn_fused at units 256: != 0 = the 128-row kernel instead of ffn64_kernel (A/B) */
  int32_t groupnorm_two_launches;   /* pd_groupnorm_silu: != 0 = always the statistics + apply pair of launches (A/B) */
  int32_t pair_form;                /* pd_attn_ffn_pair at units 256: 1 / 2 = groups per wave with four waves (64- / 128-row tiles), 8 = eight
                                       waves of one group (128-row tiles), whatever the grid; 0 = automatic */
  int32_t ffn_debug_flags;          /* profiling ablations of the fused FFN (scripts/bench_ffn.py); 0 in production */
  int32_t attn_block_debug_flags;   /* profiling ablations of the fused attention block (scripts/bench_attn_block.py) */
  int32_t small_grid;               /* != 0: the caller allows kernel choices that depend on the launch size (the engine's small-batch mode; results
                                       then are not bit-identical across batch sizes).  pd_groupnorm_silu: finer channel chunks (twice the workgroups)
                                       at <= 1024 rows per sample when the default chunks would leave more than half of the CUs idle */
  int32_t w_fold;                   /* ABI 4.  pd_attn_ffn_pair / pd_attn_ffn_pair_split: != 0 = the weight stream holds W_hi AND W_lo chunks
                                       (packing.pack_pair_block(fold=True): twice the chunks; precision="fp16x2"), two MFMA products per k-step
                                       against the once-rounded activations.  One 16-slot group per wave only (pair_form 2 is refused) */
  int32_t reserved1;
  unsigned long long* trace;        /* device buffer for per-phase clock stamps of the fused kernels (pd_ffn_fused, pd_attn_block_fused_ex,
                                       pd_attn_ffn_pair in a -DPD_PAIR_TRACE=1 / -DPD_PAIR_DEBUG=1 build), or NULL (production) */
} pd_call_opts;
int pd_sizeof_call_opts(void);

int pd_abi_version(void);
const char* pd_last_error(void);
int pd_sizeof_igemm_args(void);         /* binding self-check: sizeof the argument structs below */
int pd_sizeof_cuboid_attn_args(void);

/* ---------------------------------------------------------------------------------------------------
 * pd_igemm: implicit-GEMM on the MFMA pipes (bf16 x bf16 -> fp32 accumulate).
 *   out[m, n] = act(alpha * sum_{tap, c} A[src(m, tap), c] * W[tap, n, c] + bias[n] + rowvec[m / rows_per_sample, n])
 *               (* mul[m, n]) (+ residual[m % res_period, n])
 * With taps == 1 it is nn.Linear (cuboid_transformer.py:849 qkv, :951 proj, :201-203 ffn_1/ffn_2, :294
 * reduction; cuboid_transformer_unet.py:492 final_proj; taming/attention.py:145-147,180).  With taps == 27 it
 * is nn.Conv3d 3x3x3 pad 1 on (B,T,H,W,C) (models/time_embed.py:92,119); with taps == 9 nn.Conv2d 3x3 incl. the
 * nearest x2 up-sampling fused in the gather (cuboid_transformer.py:373-375, taming/resnet.py:128-141) and the
 * stride-2 / pad (0,1,0,1) down-sampling (taming/resnet.py:183-188).  split != 0 runs the 3-product
 * bf16 hi/lo decomposition (fp32-class accuracy) on {A, A_lo} x {W, W_lo}.
 * ------------------------------------------------------------------------------------------------- */
typedef struct pd_igemm_args {
  const pd_bf16* A;        /* activations, channels-last rows of lda elements */
  const pd_bf16* A_lo;     /* split mode: low part (else NULL) */
  const pd_bf16* W;        /* weights packed [tap][N][ldw] (K contiguous) */
  const pd_bf16* W_lo;
  const float* bias;       /* [N] or NULL */
  const float* rowvec;     /* [n_samples][ld_rowvec] or NULL (timestep-embedding add) */
  const float* residual;   /* fp32 [*, ld_res] or NULL */
  const float* mul;        /* fp32 [M, ld_mul] or NULL (gated FFN) */
  float* out_f32;          /* [M, ld_out] or NULL */
  pd_bf16* out_bf16;       /* [M, ld_outb] or NULL */
  pd_bf16* out_bf16_lo;    /* split mode low part of the bf16 output or NULL */
  int64_t a_batch_stride, w_batch_stride, out_batch_stride, outb_batch_stride, res_batch_stride;  /* grid.z */
  int64_t w_tap_stride;    /* elements between taps of W */
  int32_t nbatch;
  int32_t M, N, Cin, taps; /* Cin: K per tap, multiple of 64 (zero padded) */
  int32_t lda, ldw;
  int32_t B, Ti, Hi, Wi, To, Ho, Wo;      /* conv geometry (input dims before up-sampling) */
  int32_t KT, KH, KW, st, sh, sw, pt, ph, pw, ut, uh, uw;
  int32_t vT, vH, vW;      /* size of the (virtually up-sampled) input the taps are bounds-checked against; 0 = Ti*ut etc. */
  int32_t rows_per_sample, ld_rowvec, ld_res, res_period, ld_mul, act, ld_out, ld_outb, split;
  float alpha;
  int32_t tile;            /* 0 = auto; 1 = 128x128, 2 = 64x64, 3-6 = pipeline variants of 128x128, 7 = 256x256 (8 waves), 8 / 9 = 128x64 / 64x128 */
  int32_t vec_epilogue;    /* set by the library */
  uint32_t a_bytes, w_bytes; /* set by the library: extent of one A / W batch (buffer-descriptor bounds) */
  int32_t debug_flags;     /* profiling ablations only: 1 skip main loop, 2 skip stores, 4 skip activation, 8 keep the dense tap loop of the 256 x 256 Conv3d kernel (0 in production) */
  int32_t ksplit;          /* set by the library: K-slices of a split-K launch (1 = none) */
  float* splitk_ws;        /* caller workspace for split-K partial sums (fp32, splitk_ws_elems elements) or NULL: without it a launch
                              is never split.  Used for small grids (few trajectories per launch): the K loop of a long-K launch is cut
                              into slices that run as separate workgroups, then summed in slice order (deterministic) with the epilogue */
  int64_t splitk_ws_elems;
  int32_t fp8;             /* != 0: A and W hold OCP e4m3 bytes (lda / ldw / strides count elements = bytes; Cin % 128 == 0, lda/ldw % 16 == 0);
                              the tensor scales go in alpha.  Long-K row-wise linear and stride-1 convolution launches only (the 256 x 256
                              kernel, v_mfma_scale_f32_16x16x128_f8f6f4 with unit block scales); no hi/lo `split`, no batch.  With a
                              `splitk_ws` small grids still take the split-K route (fp32 slabs), unless the OUTPUT is e4m3 (below) */
  int32_t out_fp8_log2;    /* k > 0: out_bf16 points to OCP e4m3 BYTES (ld_outb counts bytes) and receives e4m3(v * 2^k), round to nearest
                              even, saturating at +-448 -- the A operand of a following fp8 launch (8-column vector epilogue: N % 8 == 0,
                              no out_bf16_lo; such a launch is never K-split, with or without `splitk_ws`).  0: bf16 output */
  int32_t operand;         /* enum pd_operand: type of A, W and out_bf16 (no `split`, no `fp8` with PD_OPERAND_F16) */
  int32_t disable_256;     /* A/B: != 0 = never hand this launch to the 256 x 256 kernel */
  int32_t min_k_256;       /* A/B: smallest K (taps * Cin) of a launch the automatic choice gives to the 256 x 256 kernel; 0 = default (1024) */
  int32_t splitk_max_tiles;/* A/B: split-K only for grids of at most this many 256 x 256 tiles; 0 = default (128), < 0 = never split */
  int32_t w_fold;          /* ABI 4.  f > 0: TWO weight products per filter tap (precision="fp16x2": W = W_hi + W_lo in the 16-bit operand type,
                              the activations rounded once) -- `taps` = 2 f, W holds the f taps of W_hi followed by the f taps of W_lo, and the
                              K loop walks the activation gather of tap (t mod f) against weight slab t: D = A W_hi^T + A W_lo^T in one fp32
                              accumulator.  f = KT * KH * KW (1 for a row-wise linear layer).  0: one product (taps = KT * KH * KW) */
  int32_t reserved0;
} pd_igemm_args;
int pd_igemm(const pd_igemm_args* a, pd_stream_t stream);

/* nn.LayerNorm(eps, affine) over the last dim C of fp32 rows -> bf16 (hi[, lo]) rows of ld_out elements
 * (pad columns [C, ld_out) are written as zero).  cuboid_transformer.py:813 (attn pre-norm), :197 (FFN pre-norm). */
int pd_layernorm(const float* x, const float* gamma, const float* beta, pd_bf16* out, pd_bf16* out_lo,
                 int64_t rows, int C, int ld_out, float eps, const pd_call_opts* opts, pd_stream_t stream);

/* pd_layernorm with an OCP e4m3 output (the A operand of an fp8 pd_igemm launch): out[row, c] = e4m3(y * fp8_scale), round to nearest
 * even, saturating at +-448; rows of ld_out bytes (pad columns zero). */
int pd_layernorm_fp8(const float* x, const float* gamma, const float* beta, uint8_t* out, int64_t rows, int C, int ld_out, float eps,
                     float fp8_scale, pd_stream_t stream);

/* PatchMerging3D gather + LayerNorm(prod(ds)*C): x (B,T,H,W,C) fp32 -> (B,T/dt,H/dh,W/dw, dt*dh*dw*C) bf16, zero padding at
 * the far edge.  cuboid_transformer.py:274-293. */
int pd_patch_merge_layernorm(const float* x, const float* gamma, const float* beta, pd_bf16* out, pd_bf16* out_lo,
                             int B, int T, int H, int W, int C, int dt, int dh, int dw, int ld_out, float eps,
                             pd_stream_t stream);
/* The same with per-call options and the padding rule of the gather: pad_nearest != 0 = PatchMerging3D(padding_type="nearest") on a shape the down-sampling
 * does not divide (models/utils.py:228-256: the padded grid is the nearest-neighbour resize of the tensor); 0 = zero padding. */
int pd_patch_merge_layernorm_ex(const float* x, const float* gamma, const float* beta, pd_bf16* out, pd_bf16* out_lo,
                             int B, int T, int H, int W, int C, int dt, int dh, int dw, int ld_out, float eps, int pad_nearest,
                             const pd_call_opts* opts, pd_stream_t stream);

/* nn.GroupNorm(G, C, eps) [+ optional (1+scale)*y+shift] [+ SiLU] over channels-last x (B, S, C) fp32 -> bf16 rows of
 * ld_out elements.  `partials` is caller workspace of B*nchunk*G*2 doubles (nchunk from pd_groupnorm_nchunk).
 * models/time_embed.py:89-93,115-120,155-166; taming/resnet.py:457-458,478-485; taming/vae.py:82-84. */
int pd_groupnorm_nchunk(int S, int C);
int pd_groupnorm_silu(const float* x, const float* gamma, const float* beta, const float* ss_scale,
                      const float* ss_shift, int ld_ss, double* partials, pd_bf16* out, pd_bf16* out_lo,
                      int B, int S, int C, int G, int ld_out, float eps, int silu, const pd_call_opts* opts, pd_stream_t stream);

/* The statistics pass of pd_groupnorm_silu alone: stats[b][g] = {mean, rstd} (fp32) of nn.GroupNorm(G, C, eps) over channels-last
 * x (B, S, C); partials as for pd_groupnorm_silu.  For consumers that normalise on the fly (pd_conv2d_gn_silu). */
int pd_groupnorm_stats(const float* x, double* partials, float* stats, int B, int S, int C, int G, float eps, pd_stream_t stream);

/* ResnetBlock2D's  GroupNorm -> SiLU -> Conv2d 3x3 (stride 1, zero padding 1) [+ bias] [+ fp32 residual]  in one launch
 * (taming/resnet.py:454-495 at temb = None): x (N, H, W, Cin) fp32 channels last is normalised with stats (N, G, 2) = {mean, rstd}
 * from pd_groupnorm_stats, passed through SiLU and rounded to bf16 inside the kernel's LDS halo tile; W packed bf16 [9][Cout][Cin]
 * (prediff_amd/packing.py:pack_conv, taps (ky, kx)); out (N, H, W, Cout) fp32, may alias residual.  Supported when
 * pd_conv2d_gn_silu_supported(H, W, Cin, Cout, G): H % 8 == 0, W % 16 == 0, Cin % 64 == 0, Cout % 128 == 0, (Cin / G) % 4 == 0;
 * otherwise use pd_groupnorm_silu + pd_igemm. */
int pd_conv2d_gn_silu_supported(int H, int W, int Cin, int Cout, int G);
int pd_conv2d_gn_silu(const float* x, const float* stats, const float* gamma, const float* beta, const pd_bf16* W, const float* bias,
                      const float* residual, float* out, int N, int H, int W_, int Cin, int Cout, int G, const pd_call_opts* opts,
                      pd_stream_t stream);

/* The VAE's Upsample2D (taming/resnet.py:128-141): nearest x2 up-sampling -> Conv2d 3x3 (stride 1, zero padding 1) [+ bias] in one launch of the
 * same tile kernel: x (N, H / 2, Wd / 2, Cin) fp32 channels last -> out (N, H, Wd, Cout) fp32; H, Wd are the OUTPUT size.  The halo of a pixel
 * tile is staged from the half-resolution source (no normalisation in front of this convolution): no 16-bit cast pass over the input, no gather
 * per filter tap.  Geometry as pd_conv2d_gn_silu_supported(H, Wd, Cin, Cout, 1). */
int pd_conv2d_up2(const float* x, const pd_bf16* W, const float* bias, float* out, int N, int H, int Wd, int Cin, int Cout,
                  const pd_call_opts* opts, pd_stream_t stream);

/* pd_groupnorm_silu with an OCP e4m3 output (the A operand of an fp8 pd_igemm launch): out[b, s, c] = e4m3(y * fp8_scale),
 * round to nearest even, saturating at +-448; rows of C bytes.  C % 4 == 0, C/4 divides 256, 4 | C/G. */
int pd_groupnorm_silu_fp8(const float* x, const float* gamma, const float* beta, const float* ss_scale,
                          const float* ss_shift, int ld_ss, double* partials, uint8_t* out, int B, int S, int C, int G,
                          float eps, int silu, float fp8_scale, pd_stream_t stream);

/* Data gradient of pd_groupnorm_silu on contiguous channels-last rows (ld = C): dx (B, S, C) from x, dy and the forward's
 * partial sums (mean / rstd are re-derived from them); bwd_partials: B * pd_groupnorm_nchunk(S, C) * G * 2 doubles of scratch.
 * gamma / beta get no gradient (frozen guidance network).  C must divide 256.  What autograd derives for
 * models/time_embed.py:89-120,134-169 (GroupNorm32 -> SiLU) inside the knowledge-alignment gradient (alignment.py:60-66). */
int pd_groupnorm_silu_bwd(const float* x, const float* dy, const float* gamma, const float* beta,
                          const double* fwd_partials, double* bwd_partials, float* dx, int B, int S, int C, int G,
                          float eps, int silu, pd_stream_t stream);

/* fp32 rows -> bf16 (hi[, lo]) rows with optional row gather (rows_per_sample_out rows taken from offset row_off inside
 * each rows_per_sample_in block) and zero padded columns.  Used for un-normalised GEMM inputs
 * (cuboid_transformer_unet.py:492 x[:, in_len:], time_embed.py:169 skip_connection input, cuboid_transformer.py:373). */
int pd_cast_rows(const float* x, pd_bf16* out, pd_bf16* out_lo, int64_t n_samples, int rows_per_sample_in, int row_off,
                 int rows_per_sample_out, int C, int ld_in, int ld_out, const pd_call_opts* opts, pd_stream_t stream);

/* Cuboid self-attention core: softmax(scale * q k^T + rel_pos_bias [masked]) v per (sample, cuboid, head).
 * qkv: (B, ntok, 3*C) rows [q | k | v], head h at columns h*hd; tok_index[(nc, vol)] = flat token id or -1 for a padded
 * slot (zero q/k/v that still takes softmax mass); bias (heads, vol, vol) fp32; mask (nc, vol, vol) u8 or NULL.
 * Output o: (B, ntok, C) bf16 (and/or fp32) in natural token order.  MFMA path (vol <= 16, bf16 qkv) or generic fp32 path.
 * cuboid_transformer.py:839-861,947-949,956-962 + 388-467 (reorder) + 470-560 (mask, masked_softmax). */
typedef struct pd_cuboid_attn_args {
  const pd_bf16* qkv_bf16;   /* one of qkv_bf16 / qkv_f32 */
  const float* qkv_f32;
  const int32_t* tok_index;
  const float* bias;
  const uint8_t* mask;
  pd_bf16* out_bf16;
  pd_bf16* out_bf16_lo;
  float* out_f32;
  int32_t B, ntok, C, heads, nc, vol, ld_qkv, ld_out;
  float scale;
  int32_t force_generic;
  int32_t out_fp8_log2;      /* k > 0 (MFMA cores, cuboid volume <= 64 only): out_bf16 points to e4m3 BYTES (ld_out counts bytes) and receives
                                e4m3(o * 2^k), round to nearest even, saturating: the A operand of an fp8 proj launch.  0: bf16 output */
  const int32_t* tok_out;    /* NULL: a slot's result goes to the token it was gathered from (tok_index).  Else [nc][vol]: the token that
                                RECEIVES the slot's result, -1 = nobody -- padding_type "nearest" on a non-divisible shape, where the padded grid
                                is a nearest-neighbour resize of the tokens (several slots read one token) and the un-padding resizes back
                                (models/utils.py:228-270).  Runs on the generic core. */
  int32_t operand;           /* enum pd_operand: type of qkv_bf16 / out_bf16 (MFMA cores; the generic core reads either) */
  int32_t qkv_fp8_log2;      /* k > 0 (ABI 4; MFMA core, cuboid volume <= 64, head_dim % 32 == 0): qkv_bf16 points to OCP e4m3 BYTES holding
                                q, k, v * 2^k (ld_qkv counts bytes) -- what an fp8 pd_igemm launch writes with out_fp8_log2 = k; q k^T and attn v
                                then run on the fp8 MFMA (probabilities as e4m3(P * 256)), fp32 scores / softmax / accumulation
                                (BASELINE.json configs[4]; cuboid_transformer.py:849-861,947-952).  0: 16-bit / fp32 q, k, v */
} pd_cuboid_attn_args;
int pd_cuboid_attention(const pd_cuboid_attn_args* a, pd_stream_t stream);

/* Data gradient of pd_cuboid_attention (fp32): given qkv (B, ntok, 3C) and d_out (B, ntok, C), both in natural token order, writes
 * d_qkv (B, ntok, 3C).  The probabilities are recomputed from qkv + bias; bias table and weights get no gradient (sampling-time
 * guidance only).  Replaces what torch.autograd derives for cuboid_transformer.py:852-861,947-949 when the knowledge-alignment
 * gradient is taken (prediff/knowledge_alignment/alignment.py:60-66 through models.py:459-528).  vol <= 64, head_dim <= 128. */
int pd_cuboid_attention_bwd(const float* qkv, const float* d_out, const int32_t* tok_index, const float* bias,
                            const uint8_t* mask, float* d_qkv, int B, int ntok, int C, int heads, int nc, int vol,
                            int ld_qkv, int ld_dout, int ld_dqkv, float scale, pd_stream_t stream);

/* Row softmax of fp32 scores (rows, n) -> bf16 probabilities (taming/attention.py:176, computed in fp32). */
int pd_softmax_rows(const float* x, pd_bf16* out, pd_bf16* out_lo, int64_t rows, int n, int ld_in, int ld_out,
                    const pd_call_opts* opts, pd_stream_t stream);

/* Denoiser stem: cat([cond, x], T) + observation-indicator channel -> fp32 (B, T_in+T_out, H, W, C+1) rows of ld_out
 * elements.  cuboid_transformer_unet.py:425-428. */
int pd_unet_build_input(const float* x, const float* cond, float* out, int B, int T_in, int T_out, int HW, int C,
                        int ld_out, pd_stream_t stream);

/* Sinusoidal timestep embedding [cos(t f_k) | sin(t f_k)] (models/utils.py:68-88), t int64 (B,) -> fp32 (B, dim).
 * freqs: dim/2 fp32 frequencies exp(-ln(max_period) k / half), computed once on the host exactly as the reference does. */
int pd_timestep_embedding(const int64_t* t, const float* freqs, float* out, int B, int dim, pd_stream_t stream);

/* Small dense layer (one row per workgroup column, M <= 65535 rows): out = act_out(W act_in(x) + b), fp32 throughout (TimeEmbedLayer models/time_embed.py:15-24,
 * emb_layers :105-113).  W (N, K) row-major fp32. */
int pd_linear_small(const float* x, const float* W, const float* b, float* out, int M, int K, int N, int act_in,
                    int act_out, pd_stream_t stream);

/* x[b, s, c] += table[s, c] (PosEmbed with the three tables pre-summed; cuboid_transformer.py:65-90) */
int pd_add_rowtable(float* x, const float* table, int64_t n_samples, int rows_per_sample, int C, pd_stream_t stream);

/* out = a + b (U-Net skip add, cuboid_transformer_unet.py:473-474) */
int pd_add(const float* a, const float* b, float* out, int64_t n, pd_stream_t stream);

/* DDPM ancestral step from the predicted noise (latent_diffusion.py:553-566,592-596,620-631):
 *   z0 = c_recip[t] z - c_recipm1[t] eps ; mean = c1[t] z0 + c2[t] z [- exp(.5 logvar[t]) shift] ;
 *   out = mean + (t != 0) exp(.5 logvar[t]) temperature noise.   coef: 5 fp32 tables of length T:
 *   [sqrt_recip_ac | sqrt_recipm1_ac | post_mean_coef1 | post_mean_coef2 | post_logvar_clipped]. */
int pd_ddpm_step(const float* zt, const float* eps, const float* noise, const float* mean_shift, const int64_t* t,
                 const float* coef, int T, float* out, int B, int64_t per_sample, float temperature, int clip_denoised,
                 pd_stream_t stream);

/* DDIM step (no reference implementation; stable-diffusion lineage of diffusion/utils.py:42-70):
 *   z0 = (z - sqrt(1-a_t) eps)/sqrt(a_t) ; out = sqrt(a_prev) z0 + sqrt(1-a_prev-sigma^2) eps + sigma noise.
 *   coef (B,3) fp32 per sample: [a_t, a_prev, sigma]. */
int pd_ddim_step(const float* zt, const float* eps, const float* noise, const float* coef, float* out, int B,
                 int64_t per_sample, pd_stream_t stream);

/* Layout glue for the frame-wise VAE: fp32 NCHW <-> channels-last NHWC (taming/autoencoder_kl.py:80-113 callers,
 * latent_diffusion.py:361-380,423-432). */
int pd_nchw_to_nhwc(const float* x, float* out, int N, int C, int HW, int ld_out, pd_stream_t stream);
int pd_nhwc_to_nchw(const float* x, float* out, int N, int C, int HW, int ld_in, pd_stream_t stream);

/* PositionwiseFFN.forward, pre-norm (cuboid_transformer.py:182-208) fused into one launch: out = x + W2 act(W1 LN(x) + b1) + b2.
 * x/out fp32 (M, C) (may alias), W1 bf16 (Hd, C), W2 bf16 (C, Hd) as packed for pd_igemm.  The hidden activations stay in LDS.
 * Supported when pd_ffn_fused_supported(C, Hd) (C in {64,128,256}, Hd % 64 == 0); otherwise use pd_layernorm + 2 x pd_igemm. */
int pd_ffn_fused_supported(int C, int Hd);
int pd_ffn_fused(const float* x, float* out, const float* gamma, const float* beta, const pd_bf16* W1, const float* b1,
                 const pd_bf16* W2, const float* b2, int64_t M, int C, int Hd, int act, float eps, const pd_call_opts* opts,
                 pd_stream_t stream);

/* Fused cuboid self-attention block, bf16 engine:  out = x + proj(attention(qkv(LayerNorm(x))))  for head_dim 64 and cuboid volume
 * <= 64 (CuboidSelfAttentionLayer.forward cuboid_transformer.py:812-966 + the residual of :1151), in place allowed (out == x).
 * x/out (B, ntok, C) fp32; Wqkv (3C, C) and Wp (C, C) bf16 row-major; bqkv (3C) / bp (C) fp32 or NULL; tok_index, bias, mask
 * as for pd_cuboid_attention.  Supported when pd_attn_block_fused_supported(C, heads, vol); otherwise use
 * pd_layernorm + pd_igemm + pd_cuboid_attention + pd_igemm. */
int pd_attn_block_fused_supported(int C, int heads, int vol);
int pd_attn_block_fused(const float* x, float* out, const float* gamma, const float* beta, const pd_bf16* Wqkv, const float* bqkv,
                        const pd_bf16* Wp, const float* bp, const int32_t* tok_index, const float* bias, const uint8_t* mask,
                        int B, int ntok, int C, int heads, int nc, int vol, float scale, float eps, pd_stream_t stream);
/* The same with a host-side hint: tok_affine = {n_inner, outer, inner, slot} (HOST pointer, or NULL) states that
 * tok_index[c][s] == (c / n_inner) * outer + (c % n_inner) * inner + s * slot for every cuboid c and slot s < vol (un-shifted, un-padded
 * axial cuboids; n_inner <= 0 = no such form): the kernel then computes the token ids instead of loading the table in front of
 * its row gather (bit-identical results either way). */
int pd_attn_block_fused_ex(const float* x, float* out, const float* gamma, const float* beta, const pd_bf16* Wqkv, const float* bqkv,
                           const pd_bf16* Wp, const float* bp, const int32_t* tok_index, const float* bias, const uint8_t* mask,
                           int B, int ntok, int C, int heads, int nc, int vol, float scale, float eps, const int32_t* tok_affine,
                           const pd_call_opts* opts, pd_stream_t stream);
/* One (CuboidSelfAttentionLayer, PositionwiseFFN) pair of StackCuboidSelfAttentionBlock.forward (cuboid_transformer.py:1147-1156:
 * x = x + attn(x); x = ffn(x) with the FFN's own residual; attention :812-966, FFN :182-208) in ONE launch, bf16 engine, GELU, cuboid
 * volume <= 16, no qkv bias, no attention mask, for units 256 (4 heads of 64, hidden 1024: every level-0 axial layer of the SEVIR-LR
 * denoiser) and units 512 (4 heads of 128, hidden 2048: every level-1 layer); in place allowed.  The rows of a wave (32 x 256 or
 * 16 x 512) stay in its registers from the first LayerNorm to the last residual: x is read once and written once
 * (csrc/pair_block.hip).  A wave works on 16-slot groups: one cuboid each, or pd_attn_ffn_pair_cuboids_per_group(vol) = 2 cuboids of
 * volume <= 8 side by side.
 *   wstream: 48 (units 256) / 192 (units 512) chunks of 32 KB = the four weight matrices as bf16 MFMA fragments in consumption order
 *            (prediff_amd/packing.py: pack_pair_block documents the layout);
 *   vecs:    LN1 gamma, beta, proj bias, LN2 gamma, beta, FFN-2 bias (units floats each), FFN-1 bias (hidden), then the (4, 16, 16) score
 *            table of a group: [head][query slot][key slot] = relative-position bias where both slots belong to the same cuboid, -inf
 *            everywhere else (padded slots, the group's other cuboid)  (pack_pair_vecs): 3584 / 6144 floats;
 *   tok_index [nc][vol] / tok_affine (HOST pointer, 4 ints, or NULL): as for pd_attn_block_fused_ex; one of them is required. */
int pd_attn_ffn_pair_supported(int C, int heads, int hidden, int vol, int act);
int pd_attn_ffn_pair_cuboids_per_group(int vol);
int pd_attn_ffn_pair(const float* x, float* out, const void* wstream, const float* vecs, const int32_t* tok_index,
                     const int32_t* tok_affine, int B, int ntok, int nc, int vol, int units, float scale, float eps_attn,
                     float eps_ffn, const pd_call_opts* opts, pd_stream_t stream);

/* The same pair for SMALL GRIDS (few trajectories per launch), units 512: two tile launches + two row sums that give every 64-row tile to FOUR workgroups,
 * each streaming a quarter of the weights -- (tile, head): LayerNorm-1 + that head's attention + its proj partial; a row sum
 * x' = x + b_proj + the four partials (in head order); (tile, quarter of the hidden units): LayerNorm-2 of x', that quarter's FFN partial;
 * a row sum of the four FFN partials (in order) into out.  x and vecs 16 B aligned.  Deterministic; the fp32 summation order differs from pd_attn_ffn_pair's (partials instead of one
 * running accumulator), so the two agree to fp32 round-off amplified by the 16-bit roundings downstream, not bit for bit.
 *   wffn_split: the FFN chunks of `wstream` in quarter-major order (prediff_amd/packing.py: pack_pair_ffn_split), 128 chunks of 32 KB;
 *   ws: caller workspace of pd_attn_ffn_pair_split_ws_floats(B, ntok, units) floats (2 x 4 partial slabs of x's size), 16 B aligned. */
int64_t pd_attn_ffn_pair_split_ws_floats(int B, int ntok, int units);
int pd_attn_ffn_pair_split(const float* x, float* out, const void* wstream, const void* wffn_split, const float* vecs,
                           const int32_t* tok_index, const int32_t* tok_affine, int B, int ntok, int nc, int vol, int units, float scale,
                           float eps_attn, float eps_ffn, float* ws, int64_t ws_floats, const pd_call_opts* opts, pd_stream_t stream);

/* PositionwiseFFN.forward alone (pre-norm LayerNorm -> Linear -> GELU -> Linear -> + x, cuboid_transformer.py:182-208) on the pair kernel's FFN
 * half (ABI 4): for blocks whose attention layer pd_attn_ffn_pair cannot take (cuboid volume > 16, masks: the full-resolution grid), units 256
 * (hidden 1024) or 512 (hidden 2048), GELU.  x, out: (rows, units) fp32 (may alias); wffn: the FFN chunks in the pair kernel's order
 * (packing.pack_pair_ffn_split(..., nsplit = 1)); vecs: the pair kernel's fp32 tables (packing.pack_pair_vecs; only LayerNorm-2, b1, b2 are read).
 * Rows are independent: any row count (a last partial group of 16 is masked). */
int pd_ffn_rows_supported(int C, int hidden, int act);
int pd_ffn_rows(const float* x, float* out, const void* wffn, const float* vecs, int64_t rows, int units, float eps,
                const pd_call_opts* opts, pd_stream_t stream);

/* SEVIRSkillScore.update (datasets/sevir/evaluation.py:193-239): hits / misses / false alarms of (pred / divisor) vs
 * (target / divisor) at every threshold (>=, NaN in either input counts nowhere), accumulated into counts[thr][t][3]
 * (int64, keep_seq) or counts[thr][3].  Tensors are (outer, T, inner) fp32 in [0,1]; divisor = fp32(1/255). */
int pd_sevir_skill_counts(const float* pred, const float* target, const float* thresholds, int nthr, float divisor,
                          long long* counts, int64_t outer, int T, int64_t inner, int keep_seq, pd_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* PREDIFF_HIP_H */
