// Probe: semantics of ds_read_b64_tr_b8 on gfx950 (which LDS byte lands in which byte of which lane).
//   hipcc --offload-arch=gfx950 -O3 tools/probe_tr8.hip -o /tmp/probe_tr8 && /tmp/probe_tr8
// Every lane supplies an 8-byte-aligned LDS address; the LDS holds its own byte address (low / high byte in two
// passes), so the source address of every returned byte is recovered.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(2))) int i32x2;
__global__ void k(unsigned char* out, int mode, int pass, int stride) {
  __shared__ __attribute__((aligned(16))) unsigned char sm[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) sm[i] = pass ? (unsigned char)(i >> 8) : (unsigned char)i;
  __syncthreads();
  const int lane = threadIdx.x, grp = lane >> 4, i = lane & 15;
  int addr;
  if (mode == 0) addr = lane * 8;                                        // linear
  else if (mode == 1) addr = grp * 2048 + (i >> 1) * stride + 8 * (i & 1);  // lane 2q+p: row q, bytes 8p..8p+7
  else addr = grp * 2048 + (i & 7) * stride + 8 * (i >> 3);                 // lane 8p+q
  i32x2 r = __builtin_amdgcn_ds_read_tr8_b64_v2i32(
      (__attribute__((address_space(3))) i32x2*)((__attribute__((address_space(3))) void*)(sm + addr)));
  ((i32x2*)out)[lane] = r;
}
int main() {
  unsigned char *d, lo[512], hi[512];
  hipMalloc(&d, 512);
  for (int mode = 0; mode < 3; ++mode) {
    const int stride = 64;
    k<<<1, 64>>>(d, mode, 0, stride); hipMemcpy(lo, d, 512, hipMemcpyDeviceToHost);
    k<<<1, 64>>>(d, mode, 1, stride); hipMemcpy(hi, d, 512, hipMemcpyDeviceToHost);
    printf("mode %d (row stride %d): lane -> source byte addresses of its 8 result bytes\n", mode, stride);
    for (int l = 0; l < 64; ++l) {
      printf(" lane %2d:", l);
      for (int j = 0; j < 8; ++j) printf(" %5d", lo[l * 8 + j] | (hi[l * 8 + j] << 8));
      printf("\n");
      if (l == 17 && mode != 0) { l = 31; printf("  ...\n"); }
      if (l == 33 && mode != 0) break;
    }
  }
  return 0;
}
