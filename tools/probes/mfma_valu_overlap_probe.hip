// does a wave's own VALU work issue under its MFMAs?  One wave per SIMD: a loop of 8 independent 16x16x32 MFMAs, a loop of
// 16 v_exp_f32, and both interleaved in program order (MFMA, exp, exp, MFMA, ...).  overlap works <=> interleaved ~ max.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int MODE>
__global__ __launch_bounds__(256) void probe(float* out, int iters) {
  f32x4 acc[8];
  float e[16];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 16; ++i) e[i] = -1.0f - i * 0.01f - threadIdx.x * 1e-4f;
  bf16x8 a, b;
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x & 3); b[i] = (__bf16)1.0f; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if constexpr (MODE != 1) {
        __builtin_amdgcn_sched_barrier(0);
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (MODE != 0) {
        asm volatile("v_exp_f32 %0, %0" : "+v"(e[2 * i]));
        asm volatile("v_exp_f32 %0, %0" : "+v"(e[2 * i + 1]));
      }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
#pragma unroll
  for (int i = 0; i < 16; ++i) s += e[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
  float* out;
  hipMalloc(&out, 1024 * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 40000;
  const char* names[] = {"8 MFMA", "16 v_exp_f32", "8 x (MFMA, exp, exp)"};
  for (int blocks : {256, 512}) {  // 1 and 2 waves per SIMD
    for (int mode = 0; mode < 3; ++mode) {
      float best = 1e9f;
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(blocks), dim3(256), 0, 0, out, iters);
        else if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(256), 0, 0, out, iters);
        else hipLaunchKernelGGL(probe<2>, dim3(blocks), dim3(256), 0, 0, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
      }
      printf("%d wave(s) per SIMD  %-22s %8.3f ms  %7.1f ns per loop iteration\n", blocks / 256, names[mode], best, best * 1e6 / iters);
    }
  }
  return 0;
}
