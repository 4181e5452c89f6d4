// pd_conv2d_gn_silu: out = Conv2d_3x3(SiLU(GroupNorm(x))) + bias [+ residual] in ONE kernel -- the body of the frame-wise VAE's
// ResnetBlock2D (reference taming/resnet.py:454-495: norm1 -> nonlinearity -> conv1, norm2 -> nonlinearity -> conv2 (+ shortcut);
// Encoder / Decoder taming/vae.py:9-166).  BASELINE.json north_star: "VAE ResBlock conv2d + GroupNorm + SiLU fused with LDS halo
// exchange".
//
// Un-fused, a (GroupNorm, conv) pair is three launches: statistics, apply (read fp32, write the bf16 operand) and an implicit GEMM that
// gathers its A tile from that operand once PER FILTER TAP (nine L2 -> LDS passes over the tile).  Here a workgroup owns an 8 x 16
// pixel tile of one frame and 128 output channels:
//   per 64-channel slice of the input:
//     * the (8+2) x (16+2) HALO of the tile is read ONCE from the fp32 tensor (16 B per lane), normalised with the frame's group
//       statistics, passed through SiLU, rounded to bf16 and written into an LDS tile [180 pixels][64 channels] (out-of-image pixels
//       are the zeros of the convolution's padding) -- the apply pass and the bf16 tensor no longer exist;
//     * all nine taps then read their A fragments from that tile at shifted pixel offsets (row = (py + dy) * 18 + px + dx): the
//       "halo exchange" is an address, not a copy; only the weights of the tap ([128 out][64 in], 16 KB) stream through a 3-slot
//       ring by buffer-descriptor DMA, two tiles ahead;
//   epilogue: accumulators -> per-wave LDS slab -> + bias [+ fp32 residual] -> 16 B row segments of the channels-last fp32 output.
// 256 threads = 4 waves (2 pixel halves x 2 channel halves), wave tile 64 pixels x 64 channels in 16x16x32 MFMAs; 70.5 KB of LDS, two
// workgroups per CU (one's halo staging -- VALU + HBM -- beside the other's MFMAs).
// Numerics are those of the un-fused bf16 path: bf16 activations after GroupNorm -> SiLU, bf16 weights, fp32 accumulation, fp32
// statistics from fp64 partial sums (pd_groupnorm_stats, the same reduction gn_apply uses).
// Geometry: 3 x 3, stride 1, zero padding 1, H % 8 == 0, W % 16 == 0, Cin % 64 == 0, Cout % 128 == 0, (Cin / G) % 4 == 0, G <= 256.
//
// Build note (Makefile: -fno-slp-vectorize for this file).  Under plain -O3 hipcc packs the per-slice scale computation
// sc = rstd * gamma into `v_pk_mul_f32 vD, v[gamma.xy], v[mean:rstd] op_sel:[0,1]` and places it DIRECTLY behind the s_waitcnt of the
// loads that feed it.  With two workgroups on a CU, about 3 % of the workgroups then got 0.0 in the LOW half of that product in lanes
// 48..63 of one wave (the high half and every source register, dumped a few instructions later, were right): whole 16-lane groups of
// the halo tile normalised with scale 0, i.e. diagonal stripes of wrong pixels in random tiles, different on every launch.  Three
// builds with different register allocation and with the statistics coming from LDS or from global memory failed at exactly that
// instruction; one wait state (s_nop 0) in front of it, scalar v_mul_f32 instead, or one workgroup per CU gave 0 wrong tiles in
// 100+ launches of 896 workgroups.  The same three instructions alone (scripts/ubench/hazards) do not reproduce it, so the cause is
// not pinned; without SLP packing the kernel has no packed-f32 VALU at all (which the MFMA loop does not want beside it anyway) and
// the stress in tests/test_hip_kernels.py (1280 workgroups x 16 launches, bit-equal and against torch) passes.  Record:
// profiles/r03_h_conv2d_gn_hazard.md.
#include <algorithm>
#include "common.h"

namespace PD_NS {

#define BLDS16(rsrc, ldsptr, voff, soff) \
  __builtin_amdgcn_raw_ptr_buffer_load_lds((rsrc), (__attribute__((address_space(3))) void*)(ldsptr), 16, (voff), (soff), 0, 0)

// Workgroup barrier as ONE opaque statement: the LDS operations of this wave have completed, and the compiler can move no memory
// access across it.  (__syncthreads() would also drain the weight DMA in flight.)
#define WG_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

struct pd_conv2d_gn_args_k {
  const float* x;          // [N][H][W][Cin] fp32, channels last
  const float* stats;      // [N][G][2]: mean, rstd
  const float* gamma;      // [Cin]
  const float* beta;
  const pd_bf16* W;        // [9 taps (ky, kx)][Cout][Cin] bf16
  const float* bias;       // [Cout] or null
  const float* residual;   // [N][H][W][Cout] fp32 or null (may alias out)
  float* out;              // [N][H][W][Cout] fp32
  int N, H, Wd, Cin, Cout, G;
  uint32_t w_bytes;
};

// UP (pd_conv2d_up2: the VAE's Upsample2D, taming/resnet.py:128-141 -- nearest x2 then Conv2d 3x3 pad 1): the same tile kernel with the halo
// staged from the HALF-resolution fp32 source (virtual pixel (gy, gx) reads source pixel (gy >> 1, gx >> 1); H, Wd are the OUTPUT size) and
// no normalisation / nonlinearity (there is none in front of that convolution): no fp32 -> 16-bit cast pass, no gather per filter tap.
template <bool UP>
__global__ void __launch_bounds__(256, 2) conv2d_gn_kernel(const pd_conv2d_gn_args_k p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int TH = 8, TW = 16, HC = TW + 2, NHALO = (TH + 2) * HC;     // 180 halo pixels
  constexpr int BN = 128, KC = 64;
  constexpr int X_BYTES = NHALO * 128;                                    // 23040
  constexpr int WT = BN * KC * 2;                                         // one tap's weight tile: 16 KB
  constexpr int RS = 3;                                                   // weight-ring slots
  constexpr int NV = (NHALO * 16 + 255) / 256;                            // float4 loads per thread and slice: 12 (the last one partial)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sW = smem;                                  // weight ring, 3 x 16 KB
  char* sX = smem + RS * WT;                        // halo tile [180][64] bf16, 16 B chunk XOR (row >> 1) & 7

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  const int wm = wave >> 1, wn = wave & 1;          // pixel half (tile rows 4 wm .. 4 wm + 3) x channel half
  const int l16 = lane & 15, lg = lane >> 4;

  // block -> (frame n, tile row ty, tile column tx, output-channel tile nt); the channel tiles of one pixel tile are neighbours
  const int NT = p.Cout / BN, TX = p.Wd / TW, TY = p.H / TH;
  int b = blockIdx.x;
  const int nt = b % NT; b /= NT;
  const int tx = b % TX; b /= TX;
  const int ty = b % TY;
  const int n = b / TY;
  const int y0 = ty * TH, x0 = tx * TW, n0 = nt * BN;
  const int nslice = p.Cin / KC, nstep = nslice * 9;

  const auto rW = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, p.w_bytes, 0x00020000);
  // weight DMA: one 256-thread instruction fills [32 out-channels][64 k] (4 KB), lane-linear, source-side swizzle; 4 per tile
  const int drow = tid >> 3, dpos = tid & 7;
  const int dchunk = dpos ^ ((drow >> 1) & 7);       // (rows 32 i + drow share the key: 32 >> 1 = 0 mod 8)
  const uint32_t w_voff = ((uint32_t)drow * (uint32_t)p.Cin + dchunk * 8) * 2u;
  auto issue_w = [&](int s) __attribute__((always_inline)) {       // step s = slice * 9 + tap -> ring slot s % 3
    const int cc = s / 9, tap = s - cc * 9;
    char* d = sW + (s % RS) * WT + wave * 1024;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      BLDS16(rW, d + i * 4096, w_voff, ((tap * p.Cout + n0 + i * 32) * p.Cin + cc * KC) * 2);
  };
#pragma unroll
  for (int s0 = 0; s0 < RS - 1; ++s0) issue_w(s0);

  f32x4 acc[4][4];                                   // [tile row i of the wave][16-channel tile j]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int cpg = p.Cin / p.G;
  const int c4 = tid & 15;                           // this thread's float4 column of every slice (256 % 16 == 0)
  for (int cc = 0; cc < nslice; ++cc) {
    // ---------------- halo of this channel slice: fp32 -> GroupNorm -> SiLU -> bf16 -> LDS ----------------
    const int c = cc * KC + c4 * 4;
    float4 g4 = make_float4(1.f, 1.f, 1.f, 1.f), b4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float2 mr = make_float2(0.f, 1.f);
    if constexpr (!UP) {
      g4 = *(const float4*)(p.gamma + c); b4 = *(const float4*)(p.beta + c);
      mr = *(const float2*)(p.stats + ((int64_t)n * p.G + c / cpg) * 2);     // (mean, rstd) of this thread's group
    }
    float4 v[NV];
    bool inb[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int row = (i * 256 + tid) >> 4;
      const int hy = row / HC, hx = row - hy * HC;
      const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
      inb[i] = row < NHALO && gy >= 0 && gy < p.H && gx >= 0 && gx < p.Wd;
      v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if constexpr (UP) {
        if (inb[i]) v[i] = *(const float4*)(p.x + (((int64_t)n * (p.H >> 1) + (gy >> 1)) * (p.Wd >> 1) + (gx >> 1)) * p.Cin + c);
      } else {
        if (inb[i]) v[i] = *(const float4*)(p.x + (((int64_t)n * p.H + gy) * p.Wd + gx) * p.Cin + c);
      }
    }
    WG_BARRIER();                                    // every wave is done with the previous slice's tile
    const float mean = mr.x;
    float rstd = mr.y;
    // Pinned in SOURCE (not only by the Makefile's -fno-slp-vectorize): one wait state between the s_waitcnt of the statistics load
    // and the first use of rstd, and the four scale products as scalar v_mul_f32 the vectoriser cannot pack -- both forms measured
    // 0 wrong tiles (header note); the Makefile additionally fails the build if any v_pk_*_f32 shows up in this object.
    asm volatile("s_nop 0" : "+v"(rstd));
    auto mul1 = [](float a, float b) __attribute__((always_inline)) {
      float r;
      asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
      return r;
    };
    // this thread's four channels: y = x * sc + sh with sc = rstd gamma, sh = beta - mean rstd gamma; SiLU as y * rcp(1 + exp2(-y log2 e))
    // (v_exp_f32 / v_rcp_f32, 1 ulp each: far below the bf16 rounding that follows) -- the staging is VALU work in front of the MFMAs
    const float sc0 = mul1(rstd, g4.x), sc1 = mul1(rstd, g4.y), sc2 = mul1(rstd, g4.z), sc3 = mul1(rstd, g4.w);
    const float sh0 = b4.x - mean * sc0, sh1 = b4.y - mean * sc1, sh2 = b4.z - mean * sc2, sh3 = b4.w - mean * sc3;
    auto silu = [](float y) __attribute__((always_inline)) {
      return y * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * y));
    };
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int row = (i * 256 + tid) >> 4;
      if (row >= NHALO) continue;
      float y0v, y1v, y2v, y3v;
      if constexpr (UP) {
        y0v = v[i].x; y1v = v[i].y; y2v = v[i].z; y3v = v[i].w;
      } else {
        y0v = silu(fmaf(v[i].x, sc0, sh0)); y1v = silu(fmaf(v[i].y, sc1, sh1));
        y2v = silu(fmaf(v[i].z, sc2, sh2)); y3v = silu(fmaf(v[i].w, sc3, sh3));
      }
      if (!inb[i]) y0v = y1v = y2v = y3v = 0.f;      // zero padding of the convolution input
      *(uint2*)(sX + row * 128 + (((c4 >> 1) ^ ((row >> 1) & 7)) << 4) + ((c4 & 1) << 3)) = make_uint2(pack_op2(y0v, y1v), pack_op2(y2v, y3v));
    }
    // ---------------- nine taps: A fragments from the halo tile at shifted offsets, weights from the ring ----------------
    for (int tap = 0; tap < 9; ++tap) {
      const int s = cc * 9 + tap;
      // tile s has landed (tile s + 1 may stay in flight: 4 DMA instructions); every wave is done with tile s - 1 -> refill its slot
      if (s + 1 < nstep) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      WG_BARRIER();
      if (s + RS - 1 < nstep) issue_w(s + RS - 1);
      const int dy = tap / 3, dx = tap - dy * 3;
      const char* wt = sW + (s % RS) * WT;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        op8 a[4], bw[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = (wm * 4 + i + dy) * HC + l16 + dx;
          a[i] = *(const op8*)(sX + row * 128 + (((ks * 4 + lg) ^ ((row >> 1) & 7)) << 4));
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int nr = wn * 64 + j * 16 + l16;
          bw[j] = *(const op8*)(wt + nr * 128 + (((ks * 4 + lg) ^ ((nr >> 1) & 7)) << 4));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = mfma_16x16x32(a[i], bw[j], acc[i][j]);
      }
    }
  }
  // ---------------- epilogue: lane holds out[pixel (tile row 4 wm + i, column 4 lg + r)][channel n0 + 64 wn + 16 j + l16] ----------------
  WG_BARRIER();                                      // all fragment reads done: the LDS becomes four [64 pixels][64 channels] fp32 slabs
  float* sC = (float*)smem + wave * (64 * 64);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) sC[(i * 16 + 4 * lg + r) * 64 + j * 16 + l16] = acc[i][j][r];
  // (the slab is private to the wave: no workgroup barrier)  16 lanes per pixel (float4 each), 4 pixels per pass
  const int cq = (lane & 15) * 4;
  const int ncol = n0 + wn * 64 + cq;
  float4 bias = make_float4(0.f, 0.f, 0.f, 0.f);
  if (p.bias) bias = *(const float4*)(p.bias + ncol);
#pragma unroll 4
  for (int pass = 0; pass < 16; ++pass) {
    const int pix = pass * 4 + (lane >> 4);          // pixel of the wave's 64: tile row pix >> 4, column pix & 15
    const int gy = y0 + wm * 4 + (pix >> 4), gx = x0 + (pix & 15);
    const int64_t o = (((int64_t)n * p.H + gy) * p.Wd + gx) * p.Cout + ncol;
    const float4 a4 = *(const float4*)(sC + pix * 64 + cq);
    float4 r4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.residual) r4 = *(const float4*)(p.residual + o);
    *(float4*)(p.out + o) = make_float4(a4.x + bias.x + r4.x, a4.y + bias.y + r4.y, a4.z + bias.z + r4.z, a4.w + bias.w + r4.w);
  }
#endif
}

#if !PD_IS_F16
extern "C" int pd_conv2d_gn_silu_supported(int H, int W, int Cin, int Cout, int G) {
  return H > 0 && W > 0 && H % 8 == 0 && W % 16 == 0 && Cin % 64 == 0 && Cout % 128 == 0 && G > 0 && G <= 256 && Cin % G == 0 &&
         (Cin / G) % 4 == 0;
}
extern "C" int pd_f16_conv2d_gn_silu(const float*, const float*, const float*, const float*, const pd_bf16*, const float*, const float*, float*, int, int,
                                     int, int, int, int, const pd_call_opts*, pd_stream_t);
#else
extern "C" int pd_conv2d_gn_silu_supported(int H, int W, int Cin, int Cout, int G);
#endif

template <bool UP>
static int launch_conv2d_gn(const float* x, const float* stats, const float* gamma, const float* beta, const pd_bf16* W, const float* bias,
                            const float* residual, float* out, int N, int H, int Wd, int Cin, int Cout, int G, hipStream_t stream) {
  PD_CHECK_ARG(N > 0 && (int64_t)N * H * Wd * (int64_t)std::max(Cin, Cout) < (1ll << 40), "pd_conv2d_gn_silu: bad N");
  const int64_t wbytes = (int64_t)9 * Cout * Cin * 2;
  PD_CHECK_ARG(wbytes < 0xfffffe00ll, "pd_conv2d_gn_silu: weights larger than a 4 GiB buffer descriptor");
  pd_conv2d_gn_args_k a;
  a.x = x; a.stats = stats; a.gamma = gamma; a.beta = beta; a.W = W; a.bias = bias; a.residual = residual; a.out = out;
  a.N = N; a.H = H; a.Wd = Wd; a.Cin = Cin; a.Cout = Cout; a.G = G;
  a.w_bytes = (uint32_t)wbytes;
  constexpr int lds_main = 180 * 128 + 3 * 16384;
  constexpr int lds_epi = 4 * 64 * 64 * 4;
  constexpr int lds = lds_main > lds_epi ? lds_main : lds_epi;
  static bool attr_set_dev[PD_MAX_DEVICES];
  bool& attr_set = attr_set_dev[pd_cur_device()];
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)conv2d_gn_kernel<UP>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) {
      pd_set_error("pd_conv2d_gn_silu: hipFuncSetAttribute(%d) failed: %s", lds, hipGetErrorString(e));
      return PD_ERR_LAUNCH;
    }
    attr_set = true;
  }
  const int64_t grid = (int64_t)N * (H / 8) * (Wd / 16) * (Cout / 128);
  PD_CHECK_ARG(grid < (1ll << 31), "pd_conv2d_gn_silu: grid too large");
  hipLaunchKernelGGL(conv2d_gn_kernel<UP>, dim3((unsigned)grid), dim3(256), lds, stream, a);
  PD_CHECK_LAUNCH();
  return PD_OK;
}

extern "C" int PD_ENTRY(conv2d_gn_silu)(const float* x, const float* stats, const float* gamma, const float* beta, const pd_bf16* W,
                                        const float* bias, const float* residual, float* out, int N, int H, int Wd, int Cin, int Cout, int G,
                                        const pd_call_opts* opts, pd_stream_t stream) {
  PD_FORWARD_F16(PD_OPTS_F16(opts), pd_f16_conv2d_gn_silu(x, stats, gamma, beta, W, bias, residual, out, N, H, Wd, Cin, Cout, G, opts, stream));
  PD_CHECK_ARG(x && stats && gamma && beta && W && out, "pd_conv2d_gn_silu: null pointer");
  if (!pd_conv2d_gn_silu_supported(H, Wd, Cin, Cout, G)) {
    pd_set_error("pd_conv2d_gn_silu: unsupported geometry H=%d W=%d Cin=%d Cout=%d G=%d (H %% 8, W %% 16, Cin %% 64, Cout %% 128, 4 | Cin/G)",
                 H, Wd, Cin, Cout, G);
    return PD_ERR_UNSUPPORTED;
  }
  return launch_conv2d_gn<false>(x, stats, gamma, beta, W, bias, residual, out, N, H, Wd, Cin, Cout, G, (hipStream_t)stream);
}

#if !PD_IS_F16
extern "C" int pd_f16_conv2d_up2(const float*, const pd_bf16*, const float*, float*, int, int, int, int, int, const pd_call_opts*, pd_stream_t);
#endif
// nearest x2 up-sampling -> Conv2d 3x3 pad 1 [+ bias] (Upsample2D): x (N, H / 2, Wd / 2, Cin) fp32 -> out (N, H, Wd, Cout) fp32; H, Wd = OUTPUT size
extern "C" int PD_ENTRY(conv2d_up2)(const float* x, const pd_bf16* W, const float* bias, float* out, int N, int H, int Wd, int Cin, int Cout,
                                    const pd_call_opts* opts, pd_stream_t stream) {
  PD_FORWARD_F16(PD_OPTS_F16(opts), pd_f16_conv2d_up2(x, W, bias, out, N, H, Wd, Cin, Cout, opts, stream));
  PD_CHECK_ARG(x && W && out, "pd_conv2d_up2: null pointer");
  if (!pd_conv2d_gn_silu_supported(H, Wd, Cin, Cout, 1) || (Cin & 3)) {
    pd_set_error("pd_conv2d_up2: unsupported geometry H=%d W=%d Cin=%d Cout=%d (output H %% 8, W %% 16, Cin %% 64, Cout %% 128)", H, Wd, Cin, Cout);
    return PD_ERR_UNSUPPORTED;
  }
  return launch_conv2d_gn<true>(x, nullptr, nullptr, nullptr, W, bias, nullptr, out, N, H, Wd, Cin, Cout, 1, (hipStream_t)stream);
}

}  // namespace PD_NS
