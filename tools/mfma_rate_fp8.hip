// Micro-benchmark: issue rate of the OCP fp8 MFMA shapes on gfx950 next to bf16 16x16x32 (zero / random operands).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_rate_fp8.hip -o /tmp/mfma_rate_fp8 && /tmp/mfma_rate_fp8
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int SHAPE>
__global__ __launch_bounds__(256) void k(float* out, int iters, unsigned seed) {
  i32x8 a, b;
  for (int j = 0; j < 8; ++j) {
    unsigned x = seed * (threadIdx.x * 8 + j + 1) * 2654435761u;
    x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
    a[j] = seed ? (int)(x & 0x7f7f7f7fu & 0xf7f7f7f7u) ^ (int)(x & 0x80808080u) : 0;  // no NaN / inf encodings
    x = x * 3266489917u + 1; x ^= x >> 16;
    b[j] = seed ? (int)(x & 0x77777777u) ^ (int)((x >> 3) & 0x80808080u) : 0;
  }
  bf16x8 ab, bb;
  for (int j = 0; j < 8; ++j) { ab[j] = (__bf16)(seed ? (float)((a[j] >> 8) & 255) / 64.f - 2.f : 0.f); bb[j] = (__bf16)(seed ? (float)(b[j] & 255) / 64.f - 2.f : 0.f); }
  f32x4 c[8];
  for (int i = 0; i < 8; ++i) c[i] = f32x4{0, 0, 0, 0};
  f32x16 d[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) d[i][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if constexpr (SHAPE == 0) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, c[i], 0, 0, 0);
      if constexpr (SHAPE == 1) c[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c[i], 0, 0, 0, 0, 0, 0);
      if constexpr (SHAPE == 2) c[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c[i], 1, 0, 0, 0, 0, 0);
      if constexpr (SHAPE == 3) if (i < 4) d[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, d[i], 0, 0, 0, 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += c[i][i & 3];
  for (int i = 0; i < 4; ++i) s += d[i][i * 3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int SHAPE>
void run(const char* name, float* out, int iters, unsigned seed, int per_iter, double flop) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<SHAPE><<<256, 256>>>(out, iters, seed);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  k<SHAPE><<<256, 256>>>(out, iters, seed);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double n = (double)iters * per_iter;
  printf(" %-28s %.3f ms, %.2f ns per MFMA per SIMD, %.0f TFLOP/s chip\n", name, ms, ms * 1e6 / n, n * flop * 1024 / (ms * 1e-3) / 1e12);
}
int main() {
  float* out; (void)hipMalloc(&out, 256 * 256 * 4);
  const int iters = 100000;
  for (unsigned seed : {0u, 7u}) {
    printf("operands %s\n", seed ? "random" : "zero");
    run<0>("bf16 16x16x32", out, iters, seed, 8, 2.0 * 16 * 16 * 32);
    run<1>("f8f6f4 16x16x128 e4m3 x e4m3", out, iters, seed, 8, 2.0 * 16 * 16 * 128);
    run<2>("f8f6f4 16x16x128 e5m2 x e4m3", out, iters, seed, 8, 2.0 * 16 * 16 * 128);
    run<3>("f8f6f4 32x32x64 e4m3 x e4m3", out, iters, seed, 4, 2.0 * 32 * 32 * 64);
  }
  return 0;
}
