// WAR hazard probe (gfx950): v_mfma_f32_16x16x32_bf16 D = v[4:7], C = v[6:9]  (partial overlap chosen by the register allocator
// in conv2d_gn), followed after NOPS wait states by a VALU write of v8 / v9.  How many wait states keep C intact, with 1, 2 and 4
// waves per SIMD issuing MFMAs?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

#define STR2(x) #x
#define STR(x) STR2(x)

template <int NOPS, int PRE>
__global__ void k(const uint4* A, const uint4* B, const float4* C, float4* out, int iters) {
  const int t = threadIdx.x & 63;
  bf16x8 a = ((const bf16x8*)A)[t], b = ((const bf16x8*)B)[t];
  float4 c = C[t];
  float4 r;
  for (int it = 0; it < iters; ++it) {
    asm volatile(
        "v_mov_b32 v6, %4\n\tv_mov_b32 v7, %5\n\tv_mov_b32 v8, %6\n\tv_mov_b32 v9, %7\n\t"
        "v_mov_b32 v10, 0\n\tv_mov_b32 v11, 0\n\tv_mov_b32 v12, 0\n\tv_mov_b32 v13, 0\n\t"
        "s_nop 4\n\t"
        ".rept %c10\n\tv_mfma_f32_16x16x32_bf16 v[10:13], %8, %9, v[10:13]\n\t.endr\n\t"
        "v_mfma_f32_16x16x32_bf16 v[4:7], %8, %9, v[6:9]\n\t"
        ".rept %c11\n\ts_nop 0\n\t.endr\n\t"
        "v_mov_b32 v8, 0x7fc00000\n\tv_mov_b32 v9, 0x7fc00000\n\t"
        "s_nop 15\n\ts_nop 15\n\t"
        "v_mov_b32 %0, v4\n\tv_mov_b32 %1, v5\n\tv_mov_b32 %2, v6\n\tv_mov_b32 %3, v7\n\t"
        : "=&v"(r.x), "=&v"(r.y), "=&v"(r.z), "=&v"(r.w)
        : "v"(c.x), "v"(c.y), "v"(c.z), "v"(c.w), "v"(a), "v"(b), "n"(PRE), "n"(NOPS)
        : "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13");
    out[((size_t)blockIdx.x * iters + it) * blockDim.x + threadIdx.x] = r;
  }
}

static uint4 *A, *B; static float4 *C, *o; static std::vector<float> ref;
template <int NOPS, int PRE>
void run(int block) {
  const int iters = 16, grid = 1024;
  size_t on = (size_t)grid * iters * block;
  hipMemset(o, 0, on * 16);
  hipLaunchKernelGGL((k<NOPS, PRE>), dim3(grid), dim3(block), 0, 0, A, B, C, o, iters);
  hipDeviceSynchronize();
  std::vector<float> h(on * 4);
  hipMemcpy(h.data(), o, on * 16, hipMemcpyDeviceToHost);
  if (ref.empty()) ref.assign(h.begin(), h.begin() + 256);   // first wave of the safest launch (NOPS = 32, 1 wave / SIMD)
  size_t bad = 0;
  for (size_t i = 0; i < on; ++i)
    for (int e = 0; e < 4; ++e) { float x = h[i * 4 + e], y = ref[(i % 64) * 4 + e]; bad += !(x == y); }
  printf("wait states %2d, %d MFMAs in front, %d waves/SIMD: wrong values %zu of %zu\n", NOPS, PRE, block / 256, bad, on * 4);
}

int main() {
  std::vector<uint32_t> hA(64 * 4), hB(64 * 4);
  std::vector<float> hC(64 * 4);
  srand(1);
  auto bf = [](float f) { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)(u >> 16); };
  for (auto& w : hA) w = bf((rand() % 17 - 8) * 0.125f) | ((uint32_t)bf((rand() % 17 - 8) * 0.125f) << 16);
  for (auto& w : hB) w = bf((rand() % 17 - 8) * 0.125f) | ((uint32_t)bf((rand() % 17 - 8) * 0.125f) << 16);
  for (auto& w : hC) w = (rand() % 33 - 16) * 0.25f;
  hipMalloc((void**)&A, 1024); hipMalloc((void**)&B, 1024); hipMalloc((void**)&C, 1024); hipMalloc((void**)&o, (size_t)1024 * 16 * 1024 * 16);
  hipMemcpy(A, hA.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(B, hB.data(), 1024, hipMemcpyHostToDevice);
  hipMemcpy(C, hC.data(), 1024, hipMemcpyHostToDevice);
  run<32, 0>(256);
  for (int block : {256, 512, 1024}) {
    run<0, 0>(block); run<1, 0>(block); run<2, 0>(block); run<3, 0>(block); run<4, 0>(block); run<5, 0>(block); run<6, 0>(block);
    run<8, 0>(block); run<12, 0>(block); run<16, 0>(block);
    run<3, 4>(block); run<5, 4>(block); run<8, 4>(block); run<12, 4>(block); run<16, 4>(block); run<24, 4>(block); run<32, 4>(block);
  }
  return 0;
}
