// Shared device/host helpers for the gfx950 (MI355X, CDNA4) kernels.
// Written for wave64 / MFMA / LDS-DMA directly: no portability layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned short bf16_t;  // raw bf16 bits in memory
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(2))) int i32x2;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

#define VDS_OK 0
#define VDS_ERR_ARG (-1)
#define VDS_ERR_UNSUPPORTED (-2)
#define VDS_ERR_LAUNCH (-3)

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {
  __bf16 b = (__bf16)f;  // v_cvt_pk_bf16_f32: RNE, NaN-preserving
  return __builtin_bit_cast(bf16_t, b);
}
// pack two floats to two bf16 (lo in bits 0..15)
__device__ __forceinline__ unsigned pack_bf2(float lo, float hi) {
  bf16x2 v;
  v[0] = (__bf16)lo;
  v[1] = (__bf16)hi;
  return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ float bflo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bfhi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }
__device__ __forceinline__ float dsilu_f(float x) {
  float s = 1.0f / (1.0f + __expf(-x));
  return s * (1.0f + x * (1.0f - s));
}
// erf-GELU (nn.GELU() default, model.py:84) and its derivative for the GEMM epilogues.  libm's erff is ~38 VALU
// instructions with both of its branches executed by every wave; next to 256 accumulators per lane that made the
// fc1 / fc2-dgrad epilogues VALU-bound (20k of ~90k cycles per 256^2 tile).  Here the normal CDF comes from the
// Abramowitz-Stegun 7.1.26 form of erf, |error| <= 1.5e-7 in erf (4e-7 absolute in gelu, 3e-7 in gelu' over [-12, 12],
// far below the bf16 rounding of the result): one v_rcp_f32, one v_exp_f32 -- shared with the Gaussian density of the
// derivative -- and a degree-5 Horner chain, branch-free.
//   Phi(x) = x >= 0 ? 1 - h : h,   h = 0.5 (a1 t + .. + a5 t^5) exp(-x^2 / 2),   t = 1 / (1 + p |x| / sqrt 2)
__device__ __forceinline__ void gauss_cdf(float x, float& cdf, float& e_half) {
  const float z = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(z, 0.3275911f, 1.0f));
  float poly = __builtin_fmaf(t, 0.5f * 1.061405429f, 0.5f * -1.453152027f);
  poly = __builtin_fmaf(t, poly, 0.5f * 1.421413741f);
  poly = __builtin_fmaf(t, poly, 0.5f * -0.284496736f);
  poly = __builtin_fmaf(t, poly, 0.5f * 0.254829592f);
  e_half = __builtin_amdgcn_exp2f(x * x * (-0.5f * 1.4426950408889634f));  // exp(-x^2 / 2)
  const float h = poly * t * e_half;
  cdf = x >= 0.f ? 1.0f - h : h;
}
__device__ __forceinline__ float gelu_f(float x) {
  float cdf, e;
  gauss_cdf(x, cdf, e);
  return x * cdf;
}
__device__ __forceinline__ float dgelu_f(float x) {
  // d/dx [x Phi(x)] = Phi(x) + x phi(x)
  float cdf, e;
  gauss_cdf(x, cdf, e);
  return __builtin_fmaf(x * 0.39894228040143268f, e, cdf);
}

// ---- OCP fp8 (e4m3fn = format 0, e5m2 = format 1): saturating casts of 4 / 8 scaled values -----------------
__device__ __forceinline__ float fp8_fmax(int fmt) { return fmt == 0 ? 448.0f : 57344.0f; }
template <int FMT>
__device__ __forceinline__ unsigned fp8_cvt4(float a, float b, float c, float d) {
  int w = 0;
  if constexpr (FMT == 0) {
    w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, w, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
  } else {
    w = __builtin_amdgcn_cvt_pk_bf8_f32(a, b, w, false);
    w = __builtin_amdgcn_cvt_pk_bf8_f32(c, d, w, true);
  }
  return (unsigned)w;
}

// Buffer resource (SRD) for bounds-checked raw buffer loads: out-of-range bytes read as 0.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}

// ---- LDS-DMA issued from inline asm ---------------------------------------------------------
// hipcc's waitcnt pass treats every ds_read_b64_tr_b16 (and any LDS read it cannot disambiguate)
// as possibly aliasing an outstanding `buffer_load ... lds` and drains vmcnt(0) in front of it,
// which serialises the prefetch of the next tile with the current tile's MFMAs.  Issued from
// inline asm the DMA is invisible to that pass; the kernels wait for it themselves with counted
// `s_waitcnt vmcnt(N)` (VDS_WAIT_VM) before the barrier that publishes the tile.
typedef __attribute__((ext_vector_type(4))) int srd_t;
__device__ __forceinline__ srd_t make_srd(const void* p, unsigned bytes) {
  const unsigned long a = (unsigned long)p;
  srd_t r;
  r[0] = __builtin_amdgcn_readfirstlane((int)(a & 0xffffffffu));
  r[1] = __builtin_amdgcn_readfirstlane((int)((a >> 32) & 0xffffu));
  r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
  r[3] = 0x00020000;
  return r;
}
// LDS byte address of a pointer into the dynamic LDS array (wave-uniform callers only)
__device__ __forceinline__ unsigned lds_addr_of(const void* p) {
  return __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)LDS_PTR(p));
}
// 64 lanes x 16 B from buffer offset `voff` (per lane, range-checked: out of range = zeros) to the
// 1 KiB at LDS byte address `lds` (wave-uniform).  M0 is saved and restored inside the statement.
__device__ __forceinline__ void lds_dma16(srd_t srd, unsigned lds, unsigned voff) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "s"(lds), "v"(voff), "s"(srd)
      : "memory");
}
// 64 lanes x 4 B
__device__ __forceinline__ void lds_dma4(srd_t srd, unsigned lds, unsigned voff) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dword %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "s"(lds), "v"(voff), "s"(srd)
      : "memory");
}
#define VDS_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define VDS_WAIT_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

// last error message of the host-side entry points (comm.hip); vds_last_error() reports it
namespace vdserr {
void set(const char* fmt, ...);
const char* get();
void clear();
}  // namespace vdserr

static inline unsigned clamp_u32(size_t v) { return v > 0xffffffffull ? 0xffffffffu : (unsigned)v; }
static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
