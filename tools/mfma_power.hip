// Micro-benchmark for DESIGN.md §7g: does the ORDER in which a wave's 16x16x32 bf16 MFMAs visit its operand fragments
// change the energy per MFMA (socket power / held clock, read with rocm-smi by tools/power_probe.sh)?
// 4 A fragments x 4 B fragments of random data in registers, 16 accumulators, one wave per SIMD, no memory traffic.
//   variant 0: every MFMA reads (a0, b0)                              -- no operand toggling at all
//   variant 1: A-stationary   (a0,b0) (a0,b1) (a0,b2) (a0,b3) (a1,b0) ...   -- one operand changes per MFMA
//   variant 2: diagonal       (a0,b0) (a1,b1) (a2,b2) (a3,b3) (a0,b1) ...   -- both operands change every MFMA
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_power.hip -o /tmp/mfma_power && /tmp/mfma_power <variant> <seconds> [zero]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define MF(C, A, B) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(C) : "v"(A), "v"(B))

__device__ unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int V>
__global__ __launch_bounds__(256) void k(float* out, int iters, int zero) {
  bf16x8 a[4], b[4];
  for (int f = 0; f < 4; ++f)
    for (int j = 0; j < 8; ++j) {
      const unsigned ha = hash((threadIdx.x * 64 + f * 8 + j) * 2 + 1 + blockIdx.x * 77777), hb = hash(ha + 12345);
      a[f][j] = zero ? (__bf16)0.f : (__bf16)(((int)(ha & 0xffff) - 32768) * (1.0f / 32768 / 64));
      b[f][j] = zero ? (__bf16)0.f : (__bf16)(((int)(hb & 0xffff) - 32768) * (1.0f / 32768 / 64));
    }
  f32x4 c[16];
  for (int i = 0; i < 16; ++i) c[i] = f32x4{0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    if constexpr (V == 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) MF(c[i], a[0], b[0]);
    } else if constexpr (V == 1) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) MF(c[4 * i + j], a[i], b[j]);
    } else {
#pragma unroll
      for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int i = 0; i < 4; ++i) MF(c[4 * i + ((i + d) & 3)], a[i], b[(i + d) & 3]);
    }
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += c[i][i & 3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main(int argc, char** argv) {
  const int v = argc > 1 ? atoi(argv[1]) : 1;
  const double secs = argc > 2 ? atof(argv[2]) : 10.0;
  const int zero = argc > 3;
  float* out; hipMalloc(&out, 256 * 256 * 4);
  const int iters = 20000;  // x 16 MFMAs of 16 cycles = 5.1 M cycles per launch (~2.5 ms)
  auto launch = [&] {
    if (v == 0) k<0><<<256, 256>>>(out, iters, zero);
    else if (v == 1) k<1><<<256, 256>>>(out, iters, zero);
    else k<2><<<256, 256>>>(out, iters, zero);
  };
  launch(); hipDeviceSynchronize();
  auto t0 = std::chrono::steady_clock::now();
  long n = 0;
  double el = 0;
  while (el < secs) {
    for (int i = 0; i < 20; ++i) launch();
    hipDeviceSynchronize();
    n += 20;
    el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  }
  const double mf = (double)n * iters * 16;
  printf("variant %d %s: %.3f ms per launch, %.2f ns per MFMA per SIMD, %.0f TFLOP/s chip\n", v, zero ? "zero" : "random",
         el / n * 1e3, el * 1e9 / mf, mf * 2.0 * 16 * 16 * 32 * 1024 / el / 1e12);
  return 0;
}
