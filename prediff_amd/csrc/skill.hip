// SEVIR skill-score counts: hits / misses / false alarms of pred vs target at every threshold in ONE pass over HBM.
// Replaces the per-threshold loop of SEVIRSkillScore.update (datasets/sevir/evaluation.py:233-239: 6 thresholds x
// (_threshold :12-38 + three masked sums :193-211) = ~50 elementwise/reduction launches and ~20 re-reads of the frames).
// Integer work: counts are exact and order independent (int64 atomics), so the result is bit-identical to the reference.
// HBM-bound: 8 B read per pixel, nothing written but (n_thresholds x T x 3) counters.
#include "common.h"

constexpr int SK_MAXTHR = 8;

__global__ void __launch_bounds__(256) sevir_skill_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                          const float* __restrict__ thr, int nthr, float divisor,
                                                          long long* __restrict__ counts, int T, int64_t inner, int keep_seq) {
  // grid (chunks, outer*T): a block stays inside one (outer, t) slab of `inner` contiguous elements
  const int64_t slab = blockIdx.y;
  const int t = keep_seq ? (int)(slab % T) : 0;
  const float* p = pred + slab * inner;
  const float* q = target + slab * inner;
  float th[SK_MAXTHR];
#pragma unroll
  for (int k = 0; k < SK_MAXTHR; ++k) th[k] = k < nthr ? thr[k] : 3.0e38f;
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
  const int lane = threadIdx.x & 63;
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

extern "C" int pd_sevir_skill_counts(const float* pred, const float* target, const float* thresholds, int nthr, float divisor,
                                     long long* counts, int64_t outer, int T, int64_t inner, int keep_seq, pd_stream_t stream) {
  PD_CHECK_ARG(pred && target && thresholds && counts, "pd_sevir_skill_counts: null pointer");
  PD_CHECK_ARG(nthr > 0 && nthr <= SK_MAXTHR && T > 0 && inner > 0 && outer > 0 && outer * T < 65536, "pd_sevir_skill_counts: bad sizes");
  const unsigned chunks = (unsigned)min((int64_t)64, (inner + 255) / 256);
  hipLaunchKernelGGL(sevir_skill_kernel, dim3(chunks, (unsigned)(outer * T)), dim3(256), 0, (hipStream_t)stream, pred, target, thresholds,
                     nthr, divisor, counts, T, inner, keep_seq);
  PD_CHECK_LAUNCH();
  return PD_OK;
}
