// issue rate of the 16-deep against the 32-deep bf16 MFMA on gfx950 (is a 16-deep tail of the head-dim contraction cheaper?)
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_rate_probe tools/probes/mfma_rate_probe.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int MODE>
__global__ __launch_bounds__(256) void probe(float* out, int iters) {
  f32x4 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 a8, b8;
  s16x4 a4, b4;
#pragma unroll
  for (int i = 0; i < 8; ++i) { a8[i] = (__bf16)(float)(threadIdx.x & 3); b8[i] = (__bf16)1.0f; }
#pragma unroll
  for (int i = 0; i < 4; ++i) { a4[i] = (short)0x3f80; b4[i] = (short)0x3f80; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if constexpr (MODE == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, acc[i], 0, 0, 0);
      else acc[i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, acc[i], 0, 0, 0);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
  float* out;
  hipMalloc(&out, 1024 * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000, blocks = 512;
  for (int mode = 0; mode < 2; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(blocks), dim3(256), 0, 0, out, iters);
      else hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(256), 0, 0, out, iters);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms = 0.f;
      hipEventElapsedTime(&ms, e0, e1);
      const double n = (double)blocks * 4 * iters * 8;  // wave-level MFMAs
      const double fl = n * 16 * 16 * (mode == 0 ? 32 : 16) * 2;
      printf("%s: %.3f ms  %.1f TFLOP/s  %.2f ns per MFMA per SIMD\n", mode == 0 ? "16x16x32" : "16x16x16", ms, fl / ms * 1e-9,
             ms * 1e6 / (n / (256.0 * 4)));
    }
  }
  return 0;
}
