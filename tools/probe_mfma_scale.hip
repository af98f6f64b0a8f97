// Probe: v_mfma_scale_f32_16x16x128_f8f6f4 with real E8M0 block scales -- value semantics and issue rate.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_mfma_scale.hip -o /tmp/probe_mfma_scale && /tmp/probe_mfma_scale
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
__global__ void sem(float* out, int sa, int sb) {
  i32x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = 0x38383838; b[j] = 0x38383838; }  // e4m3 1.0 everywhere: plain sum = 128
  if ((threadIdx.x >> 4) == 1) for (int j = 0; j < 8; ++j) a[j] = 0x40404040;  // k-block 1 of A = 2.0
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  f32x4 d = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, sa, 0, sb);
  out[threadIdx.x] = d[0];
}
template <int SCALED>
__global__ __launch_bounds__(256) void rate(float* out, int iters, int sa) {
  i32x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = 0x38383838 + threadIdx.x; b[j] = 0x30303030 + j; }
  f32x4 c[8];
  for (int i = 0; i < 8; ++i) c[i] = f32x4{0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if constexpr (SCALED) c[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c[i], 0, 0, 0, sa, 0, 0x7f7f7f7f);
      else c[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c[i], 0, 0, 0, 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += c[i][i & 3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
  float* out; (void)hipMalloc(&out, 256 * 256 * 4);
  float h[64];
  const int cases[][2] = {{0x7f7f7f7f, 0x7f7f7f7f}, {0x7f7f7f77, 0x7f7f7f7f}, {0x7f7f7f7f, 0x7f7f7f77}, {0x7f7f7f80, 0x7f7f7f81}, {0x7f7f777f, 0x7f7f7f7f}};
  for (auto& cs : cases) {
    sem<<<1, 64>>>(out, cs[0], cs[1]);
    (void)hipMemcpy(h, out, 256, hipMemcpyDeviceToHost);
    printf("scale_a 0x%08x scale_b 0x%08x: D[0][0] = %g (lanes 0, 16, 32, 48: %g %g %g %g)   [unscaled: 32*2 + 96*1 = 160]\n", cs[0], cs[1], h[0], h[0], h[16], h[32], h[48]);
  }
  for (int scaled = 0; scaled < 2; ++scaled) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 100000;
    if (scaled) rate<1><<<256, 256>>>(out, iters, 0x7f7f7f77); else rate<0><<<256, 256>>>(out, iters, 0);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    if (scaled) rate<1><<<256, 256>>>(out, iters, 0x7f7f7f77); else rate<0><<<256, 256>>>(out, iters, 0);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%s: %.3f ms, %.2f ns per MFMA per SIMD\n", scaled ? "v_mfma_scale (real scales)" : "v_mfma (no scales)", ms, ms * 1e6 / (iters * 8.0));
  }
  return 0;
}
