// Micro-benchmark (round 2): how fast can ONE workgroup per CU stream an L2-resident weight set into LDS by buffer_load ... lds
// DMA, as a function of the bytes it keeps in flight?  (The fused attention / FFN kernels re-stream 512 KB / 1 MB of weights per
// 128-row tile through a 3 x 32 KB ring; their per-chunk step time is ~2.2k clocks whatever the MFMA content.)
//   hipcc -O3 --offload-arch=gfx950 -o dma_rate dma_rate.hip && ./dma_rate
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

#define BLDS16(rsrc, ldsptr, voff, soff, aux) \
  __builtin_amdgcn_raw_ptr_buffer_load_lds((rsrc), (__attribute__((address_space(3))) void*)(ldsptr), 16, (voff), (soff), 0, (aux))

// 512 threads; chunk = CH KB; ring of NS slots; the workgroup walks `nchunk` chunks of a `wbytes` weight set (wrapping) `reps` times.
// MODE 0: DMA only (wait + barrier per chunk, as the fused kernels do).  MODE 1: + every wave reads the whole landed chunk back
// with ds_read_b128 (the fragment traffic of the GEMMs).
template <int CH, int NS, int MODE, int AUX>
__global__ void __launch_bounds__(512) dma_kernel(const char* w, uint32_t wbytes, int nchunk, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const auto r = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, wbytes, 0x00020000);
  constexpr int CB = CH * 1024;
  constexpr int NI = CB / 8192;                     // DMA instructions per wave per chunk (8 waves x 1 KB each)
  const uint32_t voff = (uint32_t)tid * 16u;
  auto issue = [&](int c) {
    char* d = smem + (c % NS) * CB + wave * 1024;
    const uint32_t base = (uint32_t)(((int64_t)c * CB) % wbytes);
#pragma unroll
    for (int i = 0; i < NI; ++i) BLDS16(r, d + i * 8192, voff, __builtin_amdgcn_readfirstlane(base + i * 8192), AUX);
  };
  float acc = 0.f;
#pragma unroll
  for (int c = 0; c < NS - 1; ++c) issue(c);
  for (int c = 0; c < nchunk; ++c) {
    // chunk c has landed: at most NS - 2 younger chunks stay in flight
    if (c + NS - 1 <= nchunk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * NI) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (MODE == 1) {
      const char* s = smem + (c % NS) * CB;
      for (int i = 0; i < CB / 1024 / 4; ++i) {     // every wave reads a quarter of the chunk (4 KB x ... ) like a 2 x 4 wave grid
        const float4 v = *(const float4*)(s + ((wave & 3) * (CB / 4)) + i * 1024 + (tid & 63) * 16);
        acc += v.x + v.w;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (c + NS - 1 < nchunk) issue(c + NS - 1);
  }
  if (acc == 123.456f) sink[0] = acc;
}

template <int CH, int NS, int MODE, int AUX>
static void run(const char* w, uint32_t wbytes, float* sink, int grid) {
  const int nchunk = 16 * 512 / CH;                 // 16 passes over 512 KB worth of chunks = 8 MB per workgroup
  const int lds = CH * 1024 * NS;
  hipFuncSetAttribute((const void*)dma_kernel<CH, NS, MODE, AUX>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int it = 0; it < 2; ++it) hipLaunchKernelGGL((dma_kernel<CH, NS, MODE, AUX>), dim3(grid), dim3(512), lds, 0, w, wbytes, nchunk, sink);
  hipEventRecord(e0);
  const int reps = 5;
  for (int it = 0; it < reps; ++it) hipLaunchKernelGGL((dma_kernel<CH, NS, MODE, AUX>), dim3(grid), dim3(512), lds, 0, w, wbytes, nchunk, sink);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / reps;
  const double bytes = (double)nchunk * CH * 1024;
  printf("chunk %3d KB ring %d (in flight <= %3d KB) mode %d aux %d grid %4d : %8.1f us  %6.1f GB/s per CU  %6.2f TB/s chip  %6.0f clk/32KB @2.1GHz\n",
         CH, NS, CH * (NS - 1), MODE, AUX, grid, us, bytes / us / 1e3, bytes * grid / us / 1e6, us * 2100.0 / (bytes / 32768.0));
}

int main() {
  const uint32_t wbytes = 512 * 1024;
  char* w; float* sink;
  hipMalloc(&w, wbytes); hipMemset(w, 1, wbytes); hipMalloc(&sink, 4);
  const int G = 256;
  run<32, 2, 0, 0>(w, wbytes, sink, G);
  run<32, 3, 0, 0>(w, wbytes, sink, G);
  run<32, 4, 0, 0>(w, wbytes, sink, G);
  run<16, 3, 0, 0>(w, wbytes, sink, G);
  run<16, 6, 0, 0>(w, wbytes, sink, G);
  run<16, 8, 0, 0>(w, wbytes, sink, G);
  run<8, 12, 0, 0>(w, wbytes, sink, G);
  run<8, 16, 0, 0>(w, wbytes, sink, G);
  run<32, 3, 1, 0>(w, wbytes, sink, G);
  run<32, 4, 1, 0>(w, wbytes, sink, G);
  run<16, 8, 1, 0>(w, wbytes, sink, G);
  run<32, 3, 0, 2>(w, wbytes, sink, G);     // nt
  run<32, 3, 0, 1>(w, wbytes, sink, G);     // sc0
  run<32, 3, 0, 0>(w, wbytes, sink, 32);    // one CU per... fewer workgroups: is it the shared L2 / fabric?
  run<32, 3, 0, 0>(w, wbytes, sink, 8);
  run<32, 4, 0, 0>(w, wbytes, sink, 64);
  return 0;
}
