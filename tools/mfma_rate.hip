// Micro-benchmark: issue rate of the bf16 MFMA shapes on gfx950 (zero operands: cycle-bound, no DVFS give-back).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) short s16x4;

// inline asm keeps hipcc from rotating the accumulators through v_accvgpr moves inside the loop
#define MF(OP, C, A, B) asm volatile(OP " %0, %1, %2, %0" : "+v"(C) : "v"(A), "v"(B))
template <int SHAPE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(seed * (threadIdx.x + j)); b[j] = (__bf16)(seed * (j + 1)); }
  bf16x4 a4 = {a[0], a[1], a[2], a[3]}, b4 = {b[0], b[1], b[2], b[3]};
  f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0;
  f32x16 d0, d1, d2, d3;
  for (int r = 0; r < 16; ++r) d0[r] = d1[r] = d2[r] = d3[r] = 0.f;
  for (int it = 0; it < iters; ++it) {
    if constexpr (SHAPE == 0) {
      MF("v_mfma_f32_16x16x32_bf16", c0, a, b); MF("v_mfma_f32_16x16x32_bf16", c1, a, b);
      MF("v_mfma_f32_16x16x32_bf16", c2, a, b); MF("v_mfma_f32_16x16x32_bf16", c3, a, b);
      MF("v_mfma_f32_16x16x32_bf16", c4, a, b); MF("v_mfma_f32_16x16x32_bf16", c5, a, b);
      MF("v_mfma_f32_16x16x32_bf16", c6, a, b); MF("v_mfma_f32_16x16x32_bf16", c7, a, b);
    } else if constexpr (SHAPE == 1) {
      MF("v_mfma_f32_16x16x16_bf16", c0, a4, b4); MF("v_mfma_f32_16x16x16_bf16", c1, a4, b4);
      MF("v_mfma_f32_16x16x16_bf16", c2, a4, b4); MF("v_mfma_f32_16x16x16_bf16", c3, a4, b4);
      MF("v_mfma_f32_16x16x16_bf16", c4, a4, b4); MF("v_mfma_f32_16x16x16_bf16", c5, a4, b4);
      MF("v_mfma_f32_16x16x16_bf16", c6, a4, b4); MF("v_mfma_f32_16x16x16_bf16", c7, a4, b4);
    } else {
      MF("v_mfma_f32_32x32x16_bf16", d0, a, b); MF("v_mfma_f32_32x32x16_bf16", d1, a, b);
      MF("v_mfma_f32_32x32x16_bf16", d2, a, b); MF("v_mfma_f32_32x32x16_bf16", d3, a, b);
    }
  }
  float s = c0[0] + c1[1] + c2[2] + c3[3] + c4[0] + c5[1] + c6[2] + c7[3] + d0[0] + d1[5] + d2[10] + d3[15];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int SHAPE>
double run(float* out, int iters, float seed, int mfma_per_iter, double flop_per_mfma) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<SHAPE><<<256, 256>>>(out, iters, seed);  // one 4-wave workgroup per CU: one wave per SIMD
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<SHAPE><<<256, 256>>>(out, iters, seed);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double n = (double)iters * mfma_per_iter;
  printf("  %.3f ms, %.2f ns per MFMA per SIMD, %.0f TFLOP/s chip\n", ms, ms * 1e6 / n, n * flop_per_mfma * 1024 / (ms * 1e-3) / 1e12);
  return ms;
}

int main() {
  float* out; hipMalloc(&out, 256 * 256 * 4);
  const int iters = 200000;
  for (float seed : {0.0f, 0.37f}) {
    printf("operands %s\n", seed == 0.f ? "zero" : "non-zero");
    printf(" 16x16x32:"); run<0>(out, iters, seed, 8, 2.0 * 16 * 16 * 32);
    printf(" 16x16x16 (legacy _1k):"); run<1>(out, iters, seed, 8, 2.0 * 16 * 16 * 16);
    printf(" 32x32x16:"); run<2>(out, iters, seed, 4, 2.0 * 32 * 32 * 16);
  }
  return 0;
}
