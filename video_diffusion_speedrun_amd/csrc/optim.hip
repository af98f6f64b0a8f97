// Multi-tensor AdamW on the fp32 master (sharded) parameters, one launch for the whole model:
// torch.optim.AdamW(fused=True) semantics with the reference's betas (0.95, 0.99), eps 1e-8 and
// the per-parameter lr / weight-decay table of DiT.get_mup_setup (train.py:335-344,433;
// model.py:404-465).  HBM-bound: 16 B read + 12 B written per parameter, plus the 2-byte bf16
// shadow copy that feeds the next step's all-gather / compute (model.py:516-518), fused here so
// that no separate cast pass runs.  Also the library-level helpers (version, last error, lane-map
// self test).
#include <string.h>
#include "common.h"
#include "prof.h"
#include "../../include/vds.h"

namespace {

struct AdamT {
  float* p; const float* g; float* m; float* v; bf16_t* pb; long numel; float lr; float wd;
};

__global__ __launch_bounds__(256) void adamw_kernel(const AdamT* desc, const int* chunk_tensor,
                                                    const long* chunk_start, int chunk_elems, float beta1,
                                                    float beta2, float eps, float bc1, float rsqrt_bc2_inv,
                                                    float lr_mult, float grad_scale, const float* sc_dev) {
  if (sc_dev) {  // per-step scalars kept in device memory (whole-step graph replay)
    bc1 = sc_dev[0];
    rsqrt_bc2_inv = sc_dev[1];
    lr_mult = sc_dev[2];
  }
  const AdamT t = desc[chunk_tensor[blockIdx.x]];
  const long s0 = chunk_start[blockIdx.x];
  const long s1 = min(t.numel, s0 + (long)chunk_elems);
  const float lr = t.lr * lr_mult;
  const float decay = 1.0f - lr * t.wd;
  const float step_size = lr / bc1;
#ifndef VDS_ADAMW_U
#define VDS_ADAMW_U 2
#endif
#ifdef VDS_ADAMW_NT
#define VDS_LD(ptr) __builtin_nontemporal_load(ptr)
#define VDS_ST(ptr, val) __builtin_nontemporal_store(val, ptr)
#else
#define VDS_LD(ptr) (*(ptr))
#define VDS_ST(ptr, val) (*(ptr) = (val))
#endif
  constexpr int U = VDS_ADAMW_U;
  const bool vec_ok = (reinterpret_cast<uintptr_t>(t.p + s0) & 15) == 0 && (!t.pb || (reinterpret_cast<uintptr_t>(t.pb + s0) & 7) == 0);
  long i = s0 + threadIdx.x * 4;
  if (vec_ok) {
    // U independent 16-byte vectors per thread and iteration: every load is issued before the first use
    for (; i + (U - 1) * 1024 + 4 <= s1; i += U * 1024) {
      f32x4 p[U], g[U], m[U], v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        p[u] = VDS_LD(reinterpret_cast<const f32x4*>(t.p + i + u * 1024));
        g[u] = VDS_LD(reinterpret_cast<const f32x4*>(t.g + i + u * 1024));
        m[u] = VDS_LD(reinterpret_cast<const f32x4*>(t.m + i + u * 1024));
        v[u] = VDS_LD(reinterpret_cast<const f32x4*>(t.v + i + u * 1024));
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float gg = g[u][e] * grad_scale;
          p[u][e] *= decay;
          m[u][e] = beta1 * m[u][e] + (1.0f - beta1) * gg;
          v[u][e] = beta2 * v[u][e] + (1.0f - beta2) * gg * gg;
          const float denom = sqrtf(v[u][e]) * rsqrt_bc2_inv + eps;
          p[u][e] -= step_size * (m[u][e] / denom);
        }
        VDS_ST(reinterpret_cast<f32x4*>(t.p + i + u * 1024), p[u]);
        VDS_ST(reinterpret_cast<f32x4*>(t.m + i + u * 1024), m[u]);
        VDS_ST(reinterpret_cast<f32x4*>(t.v + i + u * 1024), v[u]);
        if (t.pb) {
          const u32x2 w = {pack_bf2(p[u][0], p[u][1]), pack_bf2(p[u][2], p[u][3])};
          VDS_ST(reinterpret_cast<u32x2*>(t.pb + i + u * 1024), w);
        }
      }
    }
  }
  for (; i < s1; i += 1024) {  // tail of the chunk / unaligned tensors
    for (long j = i; j < min(s1, i + 4); ++j) {
      const float gg = t.g[j] * grad_scale;
      float p = t.p[j] * decay;
      const float m = beta1 * t.m[j] + (1.0f - beta1) * gg;
      const float v = beta2 * t.v[j] + (1.0f - beta2) * gg * gg;
      p -= step_size * (m / (sqrtf(v) * rsqrt_bc2_inv + eps));
      t.p[j] = p; t.m[j] = m; t.v[j] = v;
      if (t.pb) t.pb[j] = f2bf(p);
    }
  }
}

// ------------------------------------------------------------------ lane-map self test ---
// Verifies on the device the MFMA operand/accumulator lane maps, the transposing LDS read,
// the accumulator-as-operand k permutation and the LDS-DMA placement the GEMM / attention
// kernels assume.  out[i] = number of mismatching lanes/elements of test i (0 = pass).
__device__ __forceinline__ float ai(int i, int k) { return (float)((i * 3 + k * 5) % 7 - 3); }
__device__ __forceinline__ float bi(int k, int j) { return (float)((k * 2 + j * 7) % 5 - 2); }

__global__ void selftest_kernel(int* out, const bf16_t* gsrc) {
  __shared__ __attribute__((aligned(16))) bf16_t lds[4096];
  const int lane = threadIdx.x;
  int err;
  // T0: 16x16x32, A[i][k] k=0..31, B[k][j]
  {
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) {
      a[j] = (__bf16)ai(lane & 15, 8 * (lane >> 4) + j);
      b[j] = (__bf16)bi(8 * (lane >> 4) + j, lane & 15);
    }
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    err = 0;
    for (int r = 0; r < 4; ++r) {
      const int row = 4 * (lane >> 4) + r, col = lane & 15;
      float ref = 0;
      for (int k = 0; k < 32; ++k) ref += ai(row, k) * bi(k, col);
      err += (c[r] != ref);
    }
    atomicAdd(out + 0, err);
  }
  // T1: 32x32x16 maps
  f32x16 X;
  {
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) {
      a[j] = (__bf16)ai(lane & 31, 8 * (lane >> 5) + j);
      b[j] = (__bf16)bi(8 * (lane >> 5) + j, lane & 31);
    }
    for (int r = 0; r < 16; ++r) X[r] = 0;
    X = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, X, 0, 0, 0);
    err = 0;
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), col = lane & 31;
      float ref = 0;
      for (int k = 0; k < 16; ++k) ref += ai(row, k) * bi(k, col);
      err += (X[r] != ref);
    }
    atomicAdd(out + 1, err);
  }
  // T2: accumulator as the B operand: Y = C * X, C[i][r] over X's 32 rows r (2 k-steps)
  {
    f32x16 Y;
    for (int r = 0; r < 16; ++r) Y[r] = 0;
    const int h = lane >> 5;
    for (int s = 0; s < 2; ++s) {
      bf16x8 xa, ca;
      for (int j = 0; j < 8; ++j) {
        xa[j] = (__bf16)X[8 * s + j];
        const int krow = 16 * s + 8 * (j >> 2) + 4 * h + (j & 3);
        ca[j] = (__bf16)(float)(((lane & 31) + 2 * krow) % 3 - 1);
      }
      Y = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ca, xa, Y, 0, 0, 0);
    }
    err = 0;
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * h, col = lane & 31;
      float ref = 0;
      for (int kr = 0; kr < 32; ++kr) {
        float x = 0;
        for (int k = 0; k < 16; ++k) x += ai(kr, k) * bi(k, col);
        ref += (float)((row + 2 * kr) % 3 - 1) * x;
      }
      err += (Y[r] != ref);
    }
    atomicAdd(out + 2, err);
  }
  // T3: ds_read_b64_tr_b16: tile [16 rows][64 cols] of 16-bit values row*64+col
  {
    for (int i = lane; i < 1024; i += 64) lds[i] = (bf16_t)i;
    __syncthreads();
    const int g = lane >> 4, i = lane & 15;
    // group g reads the 4x16 block at rows 4g..4g+3, cols 16g..16g+15
    const bf16_t* p = lds + (4 * g + (i >> 2)) * 64 + 16 * g + 4 * (i & 3);
    s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)LDS_PTR(p));
    err = 0;
    for (int qq = 0; qq < 4; ++qq) err += ((unsigned short)t[qq] != (unsigned short)((4 * g + qq) * 64 + 16 * g + i));
    atomicAdd(out + 3, err);
    __syncthreads();
  }
  // T4: LDS-DMA placement (lane-linear 16 B) and SRD out-of-range -> 0
  {
    for (int i = lane; i < 1024; i += 64) lds[i] = 0x7777;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(gsrc, 512 * 2);  // 512 valid elements
    // lanes 0..31 read in range (reversed order), lanes 32..63 out of range
    const unsigned off = lane < 32 ? (unsigned)((31 - lane) * 16) : (unsigned)(1024 + lane * 16);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(lds), 16, off, 0, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    err = 0;
    for (int e = 0; e < 8; ++e) {
      const unsigned short got = lds[lane * 8 + e];
      const unsigned short want = lane < 32 ? gsrc[(31 - lane) * 8 + e] : 0;
      err += (got != want);
    }
    atomicAdd(out + 4, err);
  }
}

__global__ void selftest_fill(bf16_t* g) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < 1024) g[i] = (bf16_t)(i * 3 + 1);
}

thread_local char g_last_error[256] = "";

}  // namespace

extern "C" int vds_version(void) { return 1; }
extern "C" const char* vds_last_error(void) {
  if (vdserr::get()[0]) {  // message of the last failing host-side call (sharding runtime); reported once
    static thread_local char once[512];
    strncpy(once, vdserr::get(), sizeof(once) - 1);
    vdserr::clear();
    return once;
  }
  hipError_t e = hipPeekAtLastError();
  return e == hipSuccess ? "" : hipGetErrorString(e);
}

extern "C" int vds_adamw_multi(const vds_adamw_tensor* desc_dev, const int32_t* chunk_tensor_dev,
                               const int64_t* chunk_start_dev, int32_t n_chunks, int32_t chunk_elems, float beta1,
                               float beta2, float eps, int32_t step, float lr_mult, float grad_scale,
                               vds_stream_t stream) {
  if (!desc_dev || !chunk_tensor_dev || !chunk_start_dev || n_chunks < 0 || step < 1 || chunk_elems < 4) return VDS_ERR_ARG;
  if (n_chunks == 0) return VDS_OK;
  static_assert(sizeof(vds_adamw_tensor) == sizeof(AdamT), "descriptor layout");
  const float bc1 = 1.0f - powf(beta1, (float)step);
  const float bc2 = 1.0f - powf(beta2, (float)step);
  // upper bound of the elements touched (last chunk of a tensor may be short): 16 B read + 14 B written each
  vdsprof::Scope ps(VDS_PROF_ADAMW, (hipStream_t)stream, 0.0, 30.0 * (double)n_chunks * chunk_elems);
  hipLaunchKernelGGL(adamw_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, (const AdamT*)desc_dev,
                     chunk_tensor_dev, (const long*)chunk_start_dev, chunk_elems, beta1, beta2, eps, bc1,
                     1.0f / sqrtf(bc2), lr_mult, grad_scale, (const float*)nullptr);
  return hipGetLastError() == hipSuccess ? VDS_OK : VDS_ERR_LAUNCH;
}

extern "C" int vds_adamw_multi_dev(const vds_adamw_tensor* desc_dev, const int32_t* chunk_tensor_dev,
                                   const int64_t* chunk_start_dev, int32_t n_chunks, int32_t chunk_elems, float beta1,
                                   float beta2, float eps, const float* scalars_dev, float grad_scale,
                                   vds_stream_t stream) {
  if (!desc_dev || !chunk_tensor_dev || !chunk_start_dev || !scalars_dev || n_chunks < 0 || chunk_elems < 4) return VDS_ERR_ARG;
  if (n_chunks == 0) return VDS_OK;
  vdsprof::Scope ps(VDS_PROF_ADAMW, (hipStream_t)stream, 0.0, 30.0 * (double)n_chunks * chunk_elems);
  hipLaunchKernelGGL(adamw_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, (const AdamT*)desc_dev,
                     chunk_tensor_dev, (const long*)chunk_start_dev, chunk_elems, beta1, beta2, eps, 1.0f, 1.0f, 1.0f,
                     grad_scale, scalars_dev);
  return hipGetLastError() == hipSuccess ? VDS_OK : VDS_ERR_LAUNCH;
}

// scratch: >= 2048 + 32 bytes of device memory (first 2048 B: source pattern, then int32[8] results)
extern "C" int vds_selftest_lanemaps(void* scratch_dev, vds_stream_t stream) {
  if (!scratch_dev) return VDS_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  bf16_t* g = (bf16_t*)scratch_dev;
  int* out = (int*)((char*)scratch_dev + 2048);
  if (hipMemsetAsync(out, 0, 32, s) != hipSuccess) return VDS_ERR_LAUNCH;
  hipLaunchKernelGGL(selftest_fill, dim3(4), dim3(256), 0, s, g);
  hipLaunchKernelGGL(selftest_kernel, dim3(1), dim3(64), 0, s, out, (const bf16_t*)g);
  return hipGetLastError() == hipSuccess ? VDS_OK : VDS_ERR_LAUNCH;
}
