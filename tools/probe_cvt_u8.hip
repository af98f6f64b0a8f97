// Probe: v_cvt_pk_u8_f32 (rounding / saturation / byte select) and the log-domain e4m3 pack built on it.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_cvt_u8.hip -o /tmp/probe_cvt_u8 && /tmp/probe_cvt_u8
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(const float* in, unsigned* out, int n) {
  const int i = threadIdx.x;
  if (i < n) {
    unsigned w = 0xAABBCCDDu;
    w = __builtin_amdgcn_cvt_pk_u8_f32(in[i], 1, w);
    out[i] = w;
  }
}
__global__ void k2(const float* x, float* out, int n) {  // e4m3 byte of 2^x by log-domain rounding, decoded again
  const int i = threadIdx.x;
  if (i < n) {
    unsigned w = 0;
    w = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaf(x[i], 8.0f, 56.0f), 0, w);
    out[i] = __builtin_amdgcn_cvt_f32_fp8((int)w, 0);
  }
}
int main() {
  const float h[16] = {-3.f, -0.4f, 0.f, 0.49f, 0.5f, 0.51f, 1.5f, 2.5f, 2.51f, 126.49f, 126.5f, 200.f, 255.4f, 255.6f, 300.f, 1e9f};
  float *d, *d2; unsigned* o; float* o2;
  (void)hipMalloc(&d, 64); (void)hipMalloc(&o, 64); (void)hipMalloc(&d2, 256); (void)hipMalloc(&o2, 256);
  (void)hipMemcpy(d, h, 64, hipMemcpyHostToDevice);
  k<<<1, 64>>>(d, o, 16);
  unsigned r[16]; (void)hipMemcpy(r, o, 64, hipMemcpyDeviceToHost);
  for (int i = 0; i < 16; ++i) printf("cvt_pk_u8_f32(%g, sel 1, 0xAABBCCDD) = 0x%08x (byte %u)\n", h[i], r[i], (r[i] >> 8) & 255);
  float xs[64], ys[64];
  for (int i = 0; i < 64; ++i) xs[i] = -10.f + i * 0.3f;
  (void)hipMemcpy(d2, xs, 256, hipMemcpyHostToDevice);
  k2<<<1, 64>>>(d2, o2, 64);
  (void)hipMemcpy(ys, o2, 256, hipMemcpyDeviceToHost);
  for (int i = 0; i < 64; ++i) printf("x %6.2f  2^x %12.6f  log-domain e4m3 %12.6f  ratio %.4f\n", xs[i], exp2f(xs[i]), ys[i], ys[i] / exp2f(xs[i]));
  return 0;
}
