// SEVIR skill-score counts: hits / misses / false alarms of pred vs target at every threshold in ONE pass over HBM.
// Replaces the per-threshold loop of SEVIRSkillScore.update (datasets/sevir/evaluation.py:233-239: 6 thresholds x
// (_threshold :12-38 + three masked sums :193-211) = ~50 elementwise/reduction launches and ~20 re-reads of the frames).
// Integer work: counts are exact and order independent (int64 atomics), so the result is bit-identical to the reference.
// HBM-bound: 8 B read per pixel, nothing written but (n_thresholds x T x 3) counters.
#include "common.h"

constexpr int SK_MAXTHR = 8;
constexpr int SK_MAXT = 64;        // time steps with their own counters in the small-inner kernel (LDS table)

// (a) T is followed by a large contiguous extent (layouts N T H W C / N T C H W ...): grid (chunks, slabs), a block stays inside
//     one (outer, t) slab of `inner` contiguous elements and walks the slabs with a grid stride.
__global__ void __launch_bounds__(256) sevir_skill_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                          const float* __restrict__ thr, int nthr, float divisor,
                                                          long long* __restrict__ counts, int T, int64_t inner, int keep_seq,
                                                          int64_t nslab) {
  float th[SK_MAXTHR];
#pragma unroll
  for (int k = 0; k < SK_MAXTHR; ++k) th[k] = k < nthr ? thr[k] : 3.0e38f;
  const int lane = threadIdx.x & 63;
  for (int64_t slab = blockIdx.y; slab < nslab; slab += gridDim.y) {
    const int t = keep_seq ? (int)(slab % T) : 0;
    const float* p = pred + slab * inner;
    const float* q = target + slab * inner;
    int h[SK_MAXTHR], ms[SK_MAXTHR], fa[SK_MAXTHR];
#pragma unroll
    for (int k = 0; k < SK_MAXTHR; ++k) h[k] = ms[k] = fa[k] = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < inner; i += (int64_t)gridDim.x * 256) {
      const float pv = p[i] / divisor, tv = q[i] / divisor;      // IEEE division, as data.float() / scale in the reference
      const bool ok = !(isnan(pv) || isnan(tv));
#pragma unroll
      for (int k = 0; k < SK_MAXTHR; ++k) {
        const bool tb = ok && tv >= th[k], pb = ok && pv >= th[k];
        h[k] += tb && pb;
        ms[k] += tb && !pb;
        fa[k] += !tb && pb;
      }
    }
#pragma unroll
    for (int k = 0; k < SK_MAXTHR; ++k) {
      if (k >= nthr) break;
      int a = h[k], b = ms[k], c = fa[k];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); c += __shfl_xor(c, o, 64); }
      if (lane == 0) {
        long long* dst = counts + ((int64_t)k * (keep_seq ? T : 1) + t) * 3;
        if (a) atomicAdd((unsigned long long*)dst, (unsigned long long)a);
        if (b) atomicAdd((unsigned long long*)(dst + 1), (unsigned long long)b);
        if (c) atomicAdd((unsigned long long*)(dst + 2), (unsigned long long)c);
      }
    }
  }
}

// (b) T is the last axis or is followed by a short extent (the reference's default layout "NHWT": inner = 1): neighbouring elements
//     belong to different time steps.  Thread g reads elements g, g + stride, g + 2 stride, ... with stride a multiple of T * inner,
//     so its time step never changes (coalesced reads, private counters); the block folds them through an LDS table [T][thr][3].
__global__ void __launch_bounds__(256) sevir_skill_small_inner_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                                      const float* __restrict__ thr, int nthr, float divisor,
                                                                      long long* __restrict__ counts, int T, int inner, int keep_seq,
                                                                      int64_t total, int64_t stride) {
  __shared__ int tab[SK_MAXT * SK_MAXTHR * 3];
  const int Tk = keep_seq ? T : 1;
  for (int i = threadIdx.x; i < Tk * SK_MAXTHR * 3; i += 256) tab[i] = 0;
  __syncthreads();
  float th[SK_MAXTHR];
#pragma unroll
  for (int k = 0; k < SK_MAXTHR; ++k) th[k] = k < nthr ? thr[k] : 3.0e38f;
  int h[SK_MAXTHR], ms[SK_MAXTHR], fa[SK_MAXTHR];
#pragma unroll
  for (int k = 0; k < SK_MAXTHR; ++k) h[k] = ms[k] = fa[k] = 0;
  const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (g < stride) {
    for (int64_t i = g; i < total; i += stride) {
      const float pv = pred[i] / divisor, tv = target[i] / divisor;
      const bool ok = !(isnan(pv) || isnan(tv));
#pragma unroll
      for (int k = 0; k < SK_MAXTHR; ++k) {
        const bool tb = ok && tv >= th[k], pb = ok && pv >= th[k];
        h[k] += tb && pb;
        ms[k] += tb && !pb;
        fa[k] += !tb && pb;
      }
    }
    const int t = keep_seq ? (int)((g / inner) % T) : 0;
#pragma unroll
    for (int k = 0; k < SK_MAXTHR; ++k) {
      if (k >= nthr) break;
      if (h[k]) atomicAdd(&tab[(t * SK_MAXTHR + k) * 3], h[k]);
      if (ms[k]) atomicAdd(&tab[(t * SK_MAXTHR + k) * 3 + 1], ms[k]);
      if (fa[k]) atomicAdd(&tab[(t * SK_MAXTHR + k) * 3 + 2], fa[k]);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < Tk * nthr * 3; i += 256) {
    const int c = i % 3, k = (i / 3) % nthr, t = i / (3 * nthr);
    const int v = tab[(t * SK_MAXTHR + k) * 3 + c];
    if (v) atomicAdd((unsigned long long*)(counts + ((int64_t)k * Tk + t) * 3 + c), (unsigned long long)v);
  }
}

extern "C" int pd_sevir_skill_counts(const float* pred, const float* target, const float* thresholds, int nthr, float divisor,
                                     long long* counts, int64_t outer, int T, int64_t inner, int keep_seq, pd_stream_t stream) {
  PD_CHECK_ARG(pred && target && thresholds && counts, "pd_sevir_skill_counts: null pointer");
  PD_CHECK_ARG(nthr > 0 && nthr <= SK_MAXTHR && T > 0 && inner > 0 && outer > 0, "pd_sevir_skill_counts: bad sizes");
  const int64_t total = outer * (int64_t)T * inner;
  // per-thread int counters: one thread sees at most total / stride (+1) elements
  PD_CHECK_ARG(total < (1ll << 40), "pd_sevir_skill_counts: more than 2^40 elements in one update");
  if (inner >= 256) {
    const unsigned chunks = (unsigned)min((int64_t)64, (inner + 255) / 256);
    const int64_t nslab = outer * T;
    const unsigned gy = (unsigned)min(nslab, (int64_t)16384);
    hipLaunchKernelGGL(sevir_skill_kernel, dim3(chunks, gy), dim3(256), 0, (hipStream_t)stream, pred, target, thresholds,
                       nthr, divisor, counts, T, inner, keep_seq, nslab);
  } else {
    PD_CHECK_ARG(!keep_seq || T <= SK_MAXT, "pd_sevir_skill_counts: seq_len %d > %d with per-step counters and a short inner extent", T, SK_MAXT);
    const int64_t period = (int64_t)T * inner;                       // < 256 * T
    int64_t blocks = std::max<int64_t>(1, std::min<int64_t>(2048, (total + 255) / 256));
    blocks = std::max<int64_t>(blocks, (period + 255) / 256);
    const int64_t stride = (blocks * 256) / period * period;         // multiple of T * inner: a thread's time step is fixed
    hipLaunchKernelGGL(sevir_skill_small_inner_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, pred, target,
                       thresholds, nthr, divisor, counts, T, (int)inner, keep_seq, total, stride);
  }
  PD_CHECK_LAUNCH();
  return PD_OK;
}
