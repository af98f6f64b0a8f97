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

// ---- 4x4 transpose of one dword per lane between the four 16-lane rows of a wave (gfx950 v_permlane{32,16}_swap):
// in: row g of the wave holds w[j]; out: row g holds in w[j] what row j held in w[g] (the lane within the row is kept).
// Turns the "4 consecutive columns per lane, 4 lanes per matrix row" layout of a transposed MFMA result into 16
// consecutive columns per lane, i.e. 16-byte stores that cover a matrix row contiguously.
__device__ __forceinline__ void row_transpose4(unsigned (&w)[4]) {
#pragma unroll
  for (int j = 0; j < 2; ++j) {  // rows {2,3} of w[j] <-> rows {0,1} of w[j+2]
    const auto r = __builtin_amdgcn_permlane32_swap(w[j], w[j + 2], false, false);
    w[j] = r[0];
    w[j + 2] = r[1];
  }
#pragma unroll
  for (int j = 0; j < 4; j += 2) {  // odd rows of w[j] <-> even rows of w[j+1]
    const auto r = __builtin_amdgcn_permlane16_swap(w[j], w[j + 1], false, false);
    w[j] = r[0];
    w[j + 1] = r[1];
  }
}

// bf16 store of one 16-row block of a TRANSPOSED MFMA result (O^T, dQ^T, dK^T, dV^T of the attention kernels): lane
// (g = lane >> 4, r = lane & 15) holds columns 16 db + 4 g + 0..3 of row r in v[db][0..3]; `row` = that row's address,
// hd (64 < hd <= 80) columns exist.  wide (wave-uniform; needs 16-byte aligned rows): the four g-lanes of a row
// exchange words so that each owns 16 consecutive columns and stores 2 x 16 bytes (a row's first 128 bytes
// contiguous per instruction pair) instead of 8-byte stores 32 bytes apart -- the store tail of these kernels is
// VMEM-issue-bound.  All 64 lanes must call; `valid` is uniform over the four lanes of a row.
template <int NDB>
__device__ __forceinline__ void store_block_bf16_t(const float (&v)[NDB][4], float mul, bf16_t* row, int hd, int g,
                                                   bool valid, bool wide) {
  static_assert(NDB == 5, "four full 16-column blocks and one partial");
  unsigned lo[NDB], hi[NDB];
#pragma unroll
  for (int db = 0; db < NDB; ++db) {
    lo[db] = pack_bf2(v[db][0] * mul, v[db][1] * mul);
    hi[db] = pack_bf2(v[db][2] * mul, v[db][3] * mul);
  }
  if (wide) {
    unsigned a[4] = {lo[0], lo[1], lo[2], lo[3]}, b[4] = {hi[0], hi[1], hi[2], hi[3]};
    row_transpose4(a);
    row_transpose4(b);
    if (!valid) return;
    *reinterpret_cast<u32x4*>(row + 16 * g) = u32x4{a[0], b[0], a[1], b[1]};
    *reinterpret_cast<u32x4*>(row + 16 * g + 8) = u32x4{a[2], b[2], a[3], b[3]};
    if (64 + 4 * g < hd) *reinterpret_cast<u32x2*>(row + 64 + 4 * g) = u32x2{lo[4], hi[4]};
    return;
  }
  if (!valid) return;
#pragma unroll
  for (int db = 0; db < NDB; ++db)
    if (db * 16 + 4 * g < hd) *reinterpret_cast<u32x2*>(row + db * 16 + 4 * g) = u32x2{lo[db], hi[db]};
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
