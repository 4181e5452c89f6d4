// Does v_mfma_f32_16x16x32_bf16 with a destination that PARTIALLY overlaps srcC give the right answer?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int MODE>
__global__ void __launch_bounds__(256) k(const uint4* A, const uint4* B, const float4* C, float4* out, int iters) {
  const int t = threadIdx.x & 63;
  bf16x8 a = ((const bf16x8*)A)[t], b = ((const bf16x8*)B)[t];
  float4 c = C[t];
  float4 r;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
      asm volatile(
          "v_mov_b32 v6, %4\n\tv_mov_b32 v7, %5\n\tv_mov_b32 v8, %6\n\tv_mov_b32 v9, %7\n\t"
          "s_nop 4\n\t"
          "v_mfma_f32_16x16x32_bf16 v[4:7], %8, %9, v[6:9]\n\t"
          "s_nop 15\n\ts_nop 15\n\t"
          "v_mov_b32 %0, v4\n\tv_mov_b32 %1, v5\n\tv_mov_b32 %2, v6\n\tv_mov_b32 %3, v7\n\t"
          : "=&v"(r.x), "=&v"(r.y), "=&v"(r.z), "=&v"(r.w)
          : "v"(c.x), "v"(c.y), "v"(c.z), "v"(c.w), "v"(a), "v"(b)
          : "v4", "v5", "v6", "v7", "v8", "v9");
    } else {
      asm volatile(
          "v_mov_b32 v6, %4\n\tv_mov_b32 v7, %5\n\tv_mov_b32 v8, %6\n\tv_mov_b32 v9, %7\n\t"
          "s_nop 4\n\t"
          "v_mfma_f32_16x16x32_bf16 v[6:9], %8, %9, v[6:9]\n\t"
          "s_nop 15\n\ts_nop 15\n\t"
          "v_mov_b32 %0, v6\n\tv_mov_b32 %1, v7\n\tv_mov_b32 %2, v8\n\tv_mov_b32 %3, v9\n\t"
          : "=&v"(r.x), "=&v"(r.y), "=&v"(r.z), "=&v"(r.w)
          : "v"(c.x), "v"(c.y), "v"(c.z), "v"(c.w), "v"(a), "v"(b)
          : "v4", "v5", "v6", "v7", "v8", "v9");
    }
    out[((size_t)blockIdx.x * iters + it) * 256 + threadIdx.x] = r;
  }
}

int main() {
  const int iters = 8, grid = 4096;
  std::vector<uint32_t> hA(64 * 4), hB(64 * 4);
  std::vector<float> hC(64 * 4);
  srand(1);
  auto bf = [](float f) { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)(u >> 16); };
  for (auto& w : hA) w = bf((rand() % 17 - 8) * 0.125f) | ((uint32_t)bf((rand() % 17 - 8) * 0.125f) << 16);
  for (auto& w : hB) w = bf((rand() % 17 - 8) * 0.125f) | ((uint32_t)bf((rand() % 17 - 8) * 0.125f) << 16);
  for (auto& w : hC) w = (rand() % 33 - 16) * 0.25f;
  uint4 *A, *B; float4 *C, *o0, *o1;
  size_t on = (size_t)grid * iters * 256;
  hipMalloc(&A, 1024); hipMalloc(&B, 1024); hipMalloc(&C, 1024); hipMalloc(&o0, on * 16); hipMalloc(&o1, on * 16);
  hipMemcpy(A, hA.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(B, hB.data(), 1024, hipMemcpyHostToDevice);
  hipMemcpy(C, hC.data(), 1024, hipMemcpyHostToDevice);
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, A, B, C, o0, iters);
    hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, A, B, C, o1, iters);
    hipDeviceSynchronize();
    std::vector<float> h0(on * 4), h1(on * 4);
    hipMemcpy(h0.data(), o0, on * 16, hipMemcpyDeviceToHost); hipMemcpy(h1.data(), o1, on * 16, hipMemcpyDeviceToHost);
    size_t bad = 0, self = 0;
    for (size_t i = 0; i < on * 4; ++i) {
      bad += h0[i] != h1[i];
      self += h1[i] != h1[i % (64 * 4) + ((i / 4 % 256) / 64 == 0 ? 0 : 0)] ? 0 : 0;
    }
    // reference = lane-wise first wave of the plain (dst == srcC) launch
    size_t bad1 = 0;
    for (size_t i = 0; i < on; ++i)
      for (int e = 0; e < 4; ++e) bad1 += h1[i * 4 + e] != h1[(i % 64) * 4 + e];
    printf("rep %d: overlapped-dst vs same-dst mismatches %zu of %zu; same-dst self-inconsistency %zu\n", rep, bad, on * 4, bad1);
  }
  return 0;
}
