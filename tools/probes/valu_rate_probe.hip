// issue cost of the VALU instructions the attention softmax is made of, in cycles per wave64 instruction (gfx950)
// build: hipcc --offload-arch=gfx950 -O3 -o valu_rate_probe tools/probes/valu_rate_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(2))) float f32x2;

template <int MODE>
__global__ __launch_bounds__(256) void probe(float* out, int iters, float seed) {
  float a[8];
  f32x2 pa[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = seed + threadIdx.x * 1e-3f + i; pa[i] = f32x2{a[i], a[i] + 0.5f}; }
  const float c = seed * 0.5f;
  const f32x2 pc = {c, c};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if constexpr (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(c));
      if constexpr (MODE == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
      if constexpr (MODE == 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(pa[i]) : "v"(pc));
      if constexpr (MODE == 3) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
      if constexpr (MODE == 4) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
      if constexpr (MODE == 5) asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(a[i]) : "v"(1));
      if constexpr (MODE == 6) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(pa[i]) : "v"(pc));
      if constexpr (MODE == 7) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(pa[i]) : "v"(pc));
      if constexpr (MODE == 8) asm volatile("v_exp_f16 %0, %0" : "+v"(a[i]));
      if constexpr (MODE == 9) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(a[i]) : "v"(c));
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += a[i] + pa[i][0] + pa[i][1];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
  float* out;
  hipMalloc(&out, 2048 * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000, blocks = 1024;  // 4 waves per SIMD
  const char* names[] = {"v_fma_f32", "v_exp_f32", "v_pk_fma_f32", "v_max_f32", "v_cvt_pk_bf16_f32", "v_ldexp_f32", "v_pk_mul_f32",
                         "v_pk_add_f32", "v_exp_f16", "v_lshl_add_u32"};
  for (int mode = 0; mode < 10; ++mode) {
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      switch (mode) {
        case 0: hipLaunchKernelGGL(probe<0>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f); break;
        case 1: hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f); break;
        case 2: hipLaunchKernelGGL(probe<2>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f); break;
        case 3: hipLaunchKernelGGL(probe<3>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f); break;
        case 4: hipLaunchKernelGGL(probe<4>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f); break;
        case 5: hipLaunchKernelGGL(probe<5>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f); break;
        case 6: hipLaunchKernelGGL(probe<6>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f); break;
        case 7: hipLaunchKernelGGL(probe<7>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f); break;
        case 8: hipLaunchKernelGGL(probe<8>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f); break;
        default: hipLaunchKernelGGL(probe<9>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f); break;
      }
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms = 0.f;
      hipEventElapsedTime(&ms, e0, e1);
      best = ms < best ? ms : best;
    }
    const double n_per_simd = (double)blocks * 4 * iters * 8 / (256.0 * 4);  // wave-instructions per SIMD
    printf("%-20s %8.3f ms  %6.2f ns per wave-instruction per SIMD\n", names[mode], best, best * 1e6 / n_per_simd);
  }
  return 0;
}
