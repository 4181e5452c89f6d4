// Probe: ds_read_b64 -> s_waitcnt lgkmcnt(0) -> v_pk_mul_f32 (op_sel) immediately, with 12 global dwordx4 loads in flight and two
// workgroups per CU.  conv2d_gn's failing sequence; counts lanes whose product used stale data.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(2))) float f2;

template <int NOPS, int PK>
__global__ void __launch_bounds__(256, 2) k(const float4* x, const float* gam, const float* stats, unsigned* bad, float* sink, int iters, int xlen) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sStat = (float*)(smem + 55808);
  const int tid = threadIdx.x, lane = tid & 63;
  unsigned nb = 0;
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
    const int n = (blockIdx.x + it) & 7;
    __syncthreads();
    if (tid < 32) { sStat[2 * tid] = stats[(n * 32 + tid) * 2]; sStat[2 * tid + 1] = stats[(n * 32 + tid) * 2 + 1]; }
    const int c4 = tid & 15;
    const f2 g = *(const f2*)(gam + c4 * 4);
    float4 v[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) v[i] = x[(((size_t)blockIdx.x * 131 + it * 17 + i * 256 + tid) * 16 + c4) % xlen];
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)(smem + 55808) + (unsigned)(c4 >> 1) * 8u;
    f2 mr, sc;
    if (PK) {
      asm volatile("ds_read_b64 %0, %2\n\t"
                   "s_waitcnt vmcnt(1) lgkmcnt(0)\n\t"
                   ".rept %c4\n\ts_nop 0\n\t.endr\n\t"
                   "v_pk_mul_f32 %1, %3, %0 op_sel:[0,1]\n\t"
                   : "=&v"(mr), "=&v"(sc) : "v"(addr), "v"(g), "n"(NOPS) : "memory");
    } else {
      asm volatile("ds_read_b64 %0, %2\n\t"
                   "s_waitcnt vmcnt(1) lgkmcnt(0)\n\t"
                   ".rept %c4\n\ts_nop 0\n\t.endr\n\t"
                   "v_pk_mul_f32 %1, %3, %0\n\t"
                   : "=&v"(mr), "=&v"(sc) : "v"(addr), "v"(g), "n"(NOPS) : "memory");
    }
    const float rstd = stats[(n * 32 + (c4 >> 1)) * 2 + 1];
    const float mean = stats[(n * 32 + (c4 >> 1)) * 2];
    if (PK ? (sc.x != g.x * rstd || sc.y != g.y * rstd) : (sc.x != g.x * mean || sc.y != g.y * rstd)) ++nb;
#pragma unroll
    for (int i = 0; i < 12; ++i) acc += v[i].x + v[i].w;
  }
  if (nb) { atomicAdd(bad, nb); atomicAdd(bad + 1 + (lane >> 4), nb); }
  if (acc == 12345.678f) sink[0] = acc;
}

static float4* X; static float *G_, *S_, *sink; static unsigned* bad;
template <int NOPS, int PK>
void run(const char* what) {
  hipMemset(bad, 0, 64);
  hipFuncSetAttribute((const void*)k<NOPS, PK>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipLaunchKernelGGL((k<NOPS, PK>), dim3(2048), dim3(256), 65536, 0, X, G_, S_, bad, sink, 32, 1 << 22);
  hipDeviceSynchronize();
  unsigned h[5];
  hipMemcpy(h, bad, 20, hipMemcpyDeviceToHost);
  printf("%s, %d wait states after s_waitcnt: stale products %u (by lane quarter: %u %u %u %u) of %u\n", what, NOPS, h[0], h[1], h[2], h[3], h[4],
         2048u * 256u * 32u);
}

int main() {
  std::vector<float> hx((size_t)4 << 22), hg(64), hs(8 * 32 * 2);
  srand(3);
  for (auto& v : hx) v = (rand() % 2001 - 1000) * 1e-3f;
  for (auto& v : hg) v = 1.f + (rand() % 401 - 200) * 1e-3f;
  for (auto& v : hs) v = 0.5f + (rand() % 1001) * 1e-3f;
  hipMalloc((void**)&X, hx.size() * 4); hipMalloc((void**)&G_, 256); hipMalloc((void**)&S_, hs.size() * 4); hipMalloc((void**)&sink, 64); hipMalloc((void**)&bad, 64);
  hipMemcpy(X, hx.data(), hx.size() * 4, hipMemcpyHostToDevice); hipMemcpy(G_, hg.data(), 256, hipMemcpyHostToDevice);
  hipMemcpy(S_, hs.data(), hs.size() * 4, hipMemcpyHostToDevice);
  for (int rep = 0; rep < 2; ++rep) {
    run<0, 1>("v_pk_mul_f32 op_sel"); run<1, 1>("v_pk_mul_f32 op_sel"); run<2, 1>("v_pk_mul_f32 op_sel"); run<4, 1>("v_pk_mul_f32 op_sel");
    run<8, 1>("v_pk_mul_f32 op_sel");
    run<0, 0>("v_pk_mul_f32 plain"); run<1, 0>("v_pk_mul_f32 plain"); run<4, 0>("v_pk_mul_f32 plain");
  }
  return 0;
}
