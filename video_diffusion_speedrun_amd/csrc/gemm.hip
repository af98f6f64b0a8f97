// bf16 MFMA GEMM for gfx950: C[M,N] = sum_k opA[m,k] opB[k,n], fp32 accumulate.
//
// One kernel template covers the three operand layouts of a Linear layer's forward (NT),
// input-gradient (NN) and weight-gradient (TN) GEMMs plus the fused epilogues of the DiT
// block (bias, erf-GELU, gate*y + residual, GELU', fp32 / split-K atomics).
//
// Structure (MI355X-first, not a warp-tiled CUDA port):
//   * 128x128 output tile per 256-thread workgroup = 4 wave64s in 2x2, each wave 64x64 as
//     4x4 v_mfma_f32_16x16x32_bf16 tiles (64 accumulator VGPRs);
//   * BK = 64; operand tiles go HBM -> LDS with buffer_load_dwordx4 ... lds (LDS-DMA, no VGPR
//     staging); the SRD's num_records makes every out-of-range row read as zero, so ragged M
//     (B*L = 8208k) and the ragged contraction of the weight gradient need no masks;
//   * LDS is written lane-linearly by the DMA, so the bank-conflict swizzle is applied to the
//     per-lane SOURCE address and again on the fragment read (cdna guide rule 21);
//   * k-contiguous operands are read with ds_read_b128; k-major operands (the B of NN, both
//     operands of TN) with ds_read_b64_tr_b16, the gfx950 transposing LDS read;
//   * double-buffered LDS, one barrier per K tile; epilogue staged through LDS in fp32 so
//     that every global store / aux load is a 16-byte row segment;
//   * 1-D grid with an XCD-aware, grouped tile order (8 XCDs, private L2s).
#include "common.h"
#include "prof.h"
#include "config.h"
#include "../../include/vds.h"
#include <cmath>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <type_traits>

#ifndef VDS_GEMM_AUX_NT
#define VDS_GEMM_AUX_NT 1  // aux tiles with the non-temporal hint (NT + NN GEMM classes -1.2 ms per step; 0 = plain loads)
#endif
namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = 16384;             // one operand tile (either format)
constexpr int EPI_LD = 68;                    // floats per staged row (272 B, 16-B aligned)
constexpr int LDS_BYTES = 4 * 64 * EPI_LD * 4;  // 69632 >= 4 tiles (65536)
constexpr int GROUP_M = 8;

struct GemmP {
  int M, N, K;
  const bf16_t* A; long lda;
  const bf16_t* B; long ldb;
  void* C; long ldc;
  void* C2; long ldc2;
  const bf16_t* bias;
  const bf16_t* aux; long ldaux;
  const float* gate; long ldgate;
  int rows_per_batch;
  int row_base;    // GATE_RES: index of this launch's row 0 in the whole matrix (row-peeled launches; gate row = (row + base) / rows_per_batch)
  int split_k;
  int atomic;
  long slab_stride;  // deterministic split-K: split y writes its fp32 tile to C + y * slab_stride floats (0: all splits share C)
  unsigned a_bytes, b_bytes;
  int tiles_m, tiles_n;
  int group_m;
  int joint_xcd;   // split-K launches: XCD-aware order over the joint (split, tile) list
  int narrow;      // 256^2 kernel: the last tile column holds <= 128 columns and runs the 256 x 128 body (round 5)
  const float* sa; const float* sb;  // fp8 path: per-tensor dequantisation factors (device), else unused
  double prof_k;                      // contraction length in elements (profiler flop count)
  // fp8 path, optional: the epilogue also emits its result (gelu output / gelu' product) as fp8 (vds_fp8_out)
  unsigned char* e_q; long e_ldq; unsigned char* e_qt; long e_ldqt;
  const float* e_amax_in; float* e_amax_out; float* e_dq_out; int e_fmt; float* e_colsum;
};

// ---- swizzles -------------------------------------------------------------------------
// k-contiguous tile: [128 rows][64 k] bf16, 128-B rows, 8 chunks of 16 B.
__device__ __forceinline__ int swz_kc(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }
// k-major tile: [64 k][128 cols] bf16, 256-B rows, 8 segments of 32 B.
__device__ __forceinline__ int swz_km(int krow) { return (krow & 3) | (((krow >> 3) & 1) << 2); }

// The k advance is added into the VGPR offset (not soffset) so that the SRD range check
// sees the complete offset: rows past the end of the tensor must read as zero.
// `krem` (k-contiguous operands only): elements of K left in this tile; 16-byte chunks that start
// at or past it are redirected out of range (-> zero) so that K need only be a multiple of 8.
template <bool KMAJOR>
__device__ __forceinline__ void stage_tile(srd_t rsrc, char* lds_tile, const unsigned voff[4],
                                           unsigned koff, int wave, int lane, int krem) {
  const unsigned base = lds_addr_of(lds_tile) + wave * 4096;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    unsigned off = voff[j] + koff;
    if constexpr (!KMAJOR) {
      if (krem < BK) {
        const int row = (wave * 4 + j) * 8 + (lane >> 3);
        if (swz_kc(row, lane & 7) * 8 >= krem) off = 0xfffffff0u;
      }
    }
    lds_dma16(rsrc, base + j * 1024, off);
  }
}

// per-lane source byte offsets (k-tile independent part)
template <bool KMAJOR>
__device__ __forceinline__ void stage_offsets(unsigned voff[4], int wave, int lane, long ld, int origin) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    int q = wave * 4 + j;
    if constexpr (!KMAJOR) {
      int row = q * 8 + (lane >> 3);
      int chunk = swz_kc(row, lane & 7);
      voff[j] = (unsigned)(((long)(origin + row) * ld + chunk * 8) * 2);
    } else {
      int krow = q * 4 + (lane >> 4);
      int pc = lane & 15;
      int chunk = (((pc >> 1) ^ swz_km(krow)) << 1) | (pc & 1);
      voff[j] = (unsigned)(((long)krow * ld + origin + chunk * 8) * 2);
    }
  }
}

// fragment of a k-contiguous tile: 16 rows x 32 k, lane -> row (l&15), k 8*(l>>4)..+7
__device__ __forceinline__ bf16x8 frag_kc(const char* tile, int row0, int ks, int lane) {
  int row = row0 + (lane & 15);
  int chunk = swz_kc(row, ks * 4 + (lane >> 4));
  return *reinterpret_cast<const bf16x8*>(tile + row * 128 + chunk * 16);
}
// fragment of a k-major tile via the transposing read: 16 cols x 32 k
__device__ __forceinline__ bf16x8 frag_km(const char* tile, int col0, int ks, int lane) {
  int g = lane >> 4, i = lane & 15;
  int krow = ks * 32 + 8 * g + (i >> 2);
  int seg = (col0 >> 4) ^ swz_km(krow);
  const char* p = tile + krow * 256 + seg * 32 + (i & 3) * 8;
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)LDS_PTR(p));
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)LDS_PTR(p + 4 * 256));
  s16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, r);
}

// fp8 k-contiguous tile (128-B rows = 128 k): fragment of v_mfma_f32_16x16x128_f8f6f4, lane -> row (l&15),
// k 32*(l>>4)..+31 = the two 16-byte chunks 2g, 2g+1 (adjacent after the swizzle: (2g+1)^x = (2g^x)^1)
typedef __attribute__((ext_vector_type(8))) int i32x8;
__device__ __forceinline__ i32x8 frag_kc8(const char* tile, int row0, int lane) {
  const int row = row0 + (lane & 15);
  const int c0 = swz_kc(row, 2 * (lane >> 4));
  const i32x4 lo = *reinterpret_cast<const i32x4*>(tile + row * 128 + c0 * 16);
  const i32x4 hi = *reinterpret_cast<const i32x4*>(tile + row * 128 + (c0 ^ 1) * 16);
  return i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

// ---- erf-GELU by table (256^2 kernel) ---------------------------------------------------------------------
// The argument of gelu / gelu' in the fc1 / fc2-dgrad epilogues is a bf16 number (the stored pre-activation:
// model.py:84-85 applies nn.GELU to the bf16 output of fc1), so Phi(x) and gelu'(x) = Phi(x) + x phi(x) are functions
// of 16 bits.  A 4096-entry fp32 table -- sign | 16 binades 2^-13 .. 2^3 | 7 mantissa bits, magnitudes outside clamped
// to the nearest entry (|error| <= 1e-4 relative to 0.5 below 2^-13, < 1e-15 above 8) -- holds the correctly rounded
// values; it lives in the 24 KiB of LDS the 256^2 kernel does not use and replaces one v_rcp, one v_exp and ~16 more
// VALU instructions per element by 4 (two elements share the packed 16-bit index arithmetic) + one ds_read_b32: the
// epilogues of these two GEMMs were VALU-bound (47 instructions per element in the fp8 dgrad, 26 of its 46 us per
// tile).
constexpr int LUT_N = 4096, LUT_BYTES = LUT_N * 4, LUT_E0 = 114;  // exponent field of 2^-13
__device__ float g_gelu_lut[2][LUT_N];                             // [0] Phi, [1] gelu'
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) unsigned short u16x2;

// table values of the two bf16 numbers in `w`
// (byte offsets are formed inside the 16-bit halves: magnitude index * 4 <= 8188, sign -> bit 13; the table sits at
// LDS address 0, so a half IS the ds_read address)
__device__ __forceinline__ f32x2 lut_pair(const char* lut, unsigned w) {
  u16x2 m = __builtin_bit_cast(u16x2, w & 0x7fff7fffu);
  m = __builtin_elementwise_sub_sat(m, u16x2{LUT_E0 << 7, LUT_E0 << 7});
  m = __builtin_elementwise_min(m, u16x2{2047, 2047});
  m = m << u16x2{2, 2};
  const unsigned off = __builtin_bit_cast(unsigned, m) | ((w >> 2) & 0x20002000u);
  f32x2 t;
  t[0] = *reinterpret_cast<const float*>(lut + (off & 0xffffu));
  t[1] = *reinterpret_cast<const float*>(lut + (off >> 16));
  return t;
}
__device__ __forceinline__ f32x2 bf2_to_f2(unsigned w) { return f32x2{bflo(w), bfhi(w)}; }
__device__ __forceinline__ unsigned f2_to_bf2(f32x2 v) { return pack_bf2(v[0], v[1]); }

// host: fill the table of the current device on first use (double-precision erfc, rounded once to fp32)
bool ensure_gelu_lut() {
  static std::mutex mu;
  static bool done[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
  std::lock_guard<std::mutex> lk(mu);
  if (done[dev]) return true;
  static float host[2][LUT_N];
  for (int i = 0; i < LUT_N; ++i) {
    const unsigned bits = (unsigned)(((i & 2047) + (LUT_E0 << 7)) | ((i >> 11) << 15)) << 16;
    float xf;
    memcpy(&xf, &bits, 4);
    const double x = xf;
    const double cdf = 0.5 * erfc(-x * 0.70710678118654752440);
    const double pdf = 0.39894228040143267794 * exp(-0.5 * x * x);
    host[0][i] = (float)cdf;
    host[1][i] = (float)(cdf + x * pdf);
  }
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_gelu_lut), host, sizeof(host)) != hipSuccess) return false;
  done[dev] = true;
  return true;
}

// Shared epilogue: a wave's staged 64x64 fp32 sub-tile (stg, EPI_LD floats per row) -> global memory
// with the fused elementwise work.  (row0, col0) = global coordinates of the sub-tile.
__device__ __forceinline__ unsigned cvt4_fp8(int fmt, float a, float b, float c, float d) {
  int w = 0;
  if (fmt == 0) {
    w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, w, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
  } else {
    w = __builtin_amdgcn_cvt_pk_bf8_f32(a, b, w, false);
    w = __builtin_amdgcn_cvt_pk_bf8_f32(c, d, w, true);
  }
  return (unsigned)w;
}

// EMIT (fp8 path, BIAS_GELU / DGELU): the bf16-rounded result is additionally written as fp8, row-major and / or
// transposed, scaled by fmax / *e_amax_in (delayed scaling: the amax of the previous step), its current amax is
// recorded, and (DGELU) its column sums are accumulated -- the consumers' quantisation and bias-gradient passes
// disappear.  The fp8 bytes of the tile are also returned in `ew` (row it*8 + lane/8, columns 8*(lane%8)..+7) for
// the caller's transposed copy.
// The aux operand of a fused epilogue (GATE_RES: the residual stream, DGELU: the saved pre-activation) is requested
// for all 8 row groups of the sub-tile BEFORE the accumulators are staged through LDS, so that the 8 loads are in
// flight together and under the staging instead of one exposed HBM round trip per row group behind the stores.
// WC: columns of the staged sub-tile that exist (64; 32 in the narrow last tile column, whose lanes c8 >= 4 idle)
// NIT: row groups of 8 rows per staged sub-tile (8: 64 rows; 4: the 32-row steps of the compact staging, see gemm_tile)
template <int EPI, int WC = 64, int NIT = 8>
__device__ __forceinline__ void epilogue_prefetch(const GemmP& p, int row0, int col0, int lane, u32x4 (&auxr)[NIT]) {
  if constexpr (EPI == VDS_EPI_GATE_RES || EPI == VDS_EPI_DGELU) {
    const int c8 = lane & 7, rin = lane >> 3;
    const int gcol = col0 + c8 * 8;
    const bool cok = gcol < p.N && (WC == 64 || c8 * 8 < WC);
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const long grow = row0 + it * 8 + rin;
#if VDS_GEMM_AUX_NT  // the aux tile (residual / pre-activation) is read exactly once per launch
      auxr[it] = (grow < p.M && cok) ? __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p.aux + grow * p.ldaux + gcol))
                                     : u32x4{0u, 0u, 0u, 0u};
#else
      auxr[it] = (grow < p.M && cok) ? *reinterpret_cast<const u32x4*>(p.aux + grow * p.ldaux + gcol)
                                     : u32x4{0u, 0u, 0u, 0u};
#endif
    }
  }
}

template <int EPI, bool EMIT = false, bool LUT = false, int WC = 64, int NIT = 8>
__device__ __forceinline__ void epilogue_64x64(const GemmP& p, float* stg, int row0, int col0, int lane,
                                               float (&cs)[8], u32x2 (&ew)[NIT], const u32x4 (&auxr)[NIT],
                                               const char* lut = nullptr) {
  if constexpr (EPI == VDS_EPI_F32) {
    if (p.atomic) {
      // split-K / accumulate: one atomic wave-instruction = 64 consecutive floats of one row (256
      // contiguous bytes, the full-rate shape of global_atomic_add_f32 on gfx950)
      const int acol = col0 + lane;
      if (acol < p.N && lane < WC) {
        float* cbase = reinterpret_cast<float*>(p.C) + (long)row0 * p.ldc + acol;
        const int rmax = min(8 * NIT, p.M - row0);
        for (int row = 0; row < rmax; ++row) atomicAdd(cbase + (long)row * p.ldc, stg[row * EPI_LD + lane]);
      }
      return;
    }
  }
  const int c8 = lane & 7, rin = lane >> 3;
  const int gcol = col0 + c8 * 8;
  const bool col_ok = gcol < p.N && (WC == 64 || c8 * 8 < WC);
  if constexpr (!EMIT) {
    if (!col_ok) return;
  }
  [[maybe_unused]] float e_max = 0.f, e_scale = 1.f, e_fmax = 448.f;
  [[maybe_unused]] const bool emit = EMIT && (p.e_q || p.e_qt);
  if constexpr (EMIT) {
    e_fmax = p.e_fmt == 0 ? 448.0f : 57344.0f;
    const float ain = (emit && p.e_amax_in) ? *p.e_amax_in : 0.f;
    e_scale = ain > 0.f ? e_fmax / ain : 1.0f;
    if (emit && p.e_dq_out && row0 == 0 && col0 == 0 && lane == 0) *p.e_dq_out = ain > 0.f ? ain / e_fmax : 1.0f;
  }
  float bias8[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) bias8[e] = 0.f;
  if constexpr (EPI != VDS_EPI_F32 && EPI != VDS_EPI_DGELU) {
    if (p.bias && col_ok) {
      u32x4 bv = *reinterpret_cast<const u32x4*>(p.bias + gcol);
#pragma unroll
      for (int e = 0; e < 4; ++e) { bias8[2 * e] = bflo(bv[e]); bias8[2 * e + 1] = bfhi(bv[e]); }
    }
  }
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int row = it * 8 + rin;
    const long grow = row0 + row;
    if constexpr (EMIT) ew[it] = u32x2{0u, 0u};
    if (grow >= p.M || !col_ok) continue;
    [[maybe_unused]] u32x4 ev;  // bf16 pairs of the emitted result
    const f32x4 lo = *reinterpret_cast<const f32x4*>(stg + row * EPI_LD + c8 * 8);
    const f32x4 hi = *reinterpret_cast<const f32x4*>(stg + row * EPI_LD + c8 * 8 + 4);
    float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    if constexpr (EPI == VDS_EPI_F32) {
      float* c = reinterpret_cast<float*>(p.C) + (long)blockIdx.y * p.slab_stride + grow * p.ldc + gcol;
      *reinterpret_cast<f32x4*>(c) = lo;
      *reinterpret_cast<f32x4*>(c + 4) = hi;
    } else if constexpr (EPI == VDS_EPI_STORE) {
      u32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = pack_bf2(v[2 * e] + bias8[2 * e], v[2 * e + 1] + bias8[2 * e + 1]);
      *reinterpret_cast<u32x4*>(reinterpret_cast<bf16_t*>(p.C) + grow * p.ldc + gcol) = o;
    } else if constexpr (EPI == VDS_EPI_BIAS_GELU) {
      u32x4 o, o2;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float a = v[2 * e] + bias8[2 * e], b = v[2 * e + 1] + bias8[2 * e + 1];
        o[e] = pack_bf2(a, b);
        if constexpr (LUT) o2[e] = f2_to_bf2(bf2_to_f2(o[e]) * lut_pair(lut, o[e]));  // as epilogue_full
        else o2[e] = pack_bf2(gelu_f(a), gelu_f(b));
      }
      *reinterpret_cast<u32x4*>(reinterpret_cast<bf16_t*>(p.C) + grow * p.ldc + gcol) = o;
      if (p.C2) *reinterpret_cast<u32x4*>(reinterpret_cast<bf16_t*>(p.C2) + grow * p.ldc2 + gcol) = o2;
      if constexpr (EMIT) ev = o2;
    } else if constexpr (EPI == VDS_EPI_GATE_RES) {
      const int b = (int)((grow + p.row_base) / p.rows_per_batch);
      const float* gp = p.gate + (long)b * p.ldgate + gcol;
      const f32x4 g0 = *reinterpret_cast<const f32x4*>(gp);
      const f32x4 g1 = *reinterpret_cast<const f32x4*>(gp + 4);
      const float g[8] = {g0[0], g0[1], g0[2], g0[3], g1[0], g1[1], g1[2], g1[3]};
      const u32x4 xr = auxr[it];
      u32x4 o, o2;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float a = v[2 * e] + bias8[2 * e], bb = v[2 * e + 1] + bias8[2 * e + 1];
        o[e] = pack_bf2(a, bb);
        o2[e] = pack_bf2(bflo(xr[e]) + a * g[2 * e], bfhi(xr[e]) + bb * g[2 * e + 1]);
      }
      if (p.C) *reinterpret_cast<u32x4*>(reinterpret_cast<bf16_t*>(p.C) + grow * p.ldc + gcol) = o;
      *reinterpret_cast<u32x4*>(reinterpret_cast<bf16_t*>(p.C2) + grow * p.ldc2 + gcol) = o2;
    } else if constexpr (EPI == VDS_EPI_DGELU) {
      const u32x4 pr = auxr[it];
      u32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if constexpr (LUT) o[e] = f2_to_bf2(f32x2{v[2 * e], v[2 * e + 1]} * lut_pair(lut, pr[e]));  // as epilogue_full
        else o[e] = pack_bf2(v[2 * e] * dgelu_f(bflo(pr[e])), v[2 * e + 1] * dgelu_f(bfhi(pr[e])));
      }
      if (p.C) *reinterpret_cast<u32x4*>(reinterpret_cast<bf16_t*>(p.C) + grow * p.ldc + gcol) = o;
      if constexpr (EMIT) ev = o;
    }
    if constexpr (EMIT) {
      float f[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) { f[2 * e] = bflo(ev[e]); f[2 * e + 1] = bfhi(ev[e]); }
      if (p.e_colsum)
#pragma unroll
        for (int e = 0; e < 8; ++e) cs[e] += f[e];
      if (emit) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          e_max = fmaxf(e_max, fabsf(f[e]));
          f[e] = fminf(fmaxf(f[e] * e_scale, -e_fmax), e_fmax);
        }
        const u32x2 w = {cvt4_fp8(p.e_fmt, f[0], f[1], f[2], f[3]), cvt4_fp8(p.e_fmt, f[4], f[5], f[6], f[7])};
        ew[it] = w;
        if (p.e_q) *reinterpret_cast<u32x2*>(p.e_q + grow * p.e_ldq + gcol) = w;
      }
    }
  }
  if constexpr (EMIT) {
    if (emit) {
      if (p.e_amax_out) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) e_max = fmaxf(e_max, __shfl_xor(e_max, o));
        unsigned* a = reinterpret_cast<unsigned*>(p.e_amax_out);
        if (lane == 0 && __float_as_uint(e_max) > __atomic_load_n(a, __ATOMIC_RELAXED)) atomicMax(a, __float_as_uint(e_max));
      }
    }
  }
}

// Epilogue of a 64x64 sub-tile that lies completely inside the matrix (and, GATE_RES, inside one sample): no row /
// column predicates, row pointers advanced instead of recomputed, pair-wise (v_pk_*) arithmetic, table GELU.  Which
// outputs exist is a template parameter (C1 / C2: the bf16 results, EQ: fp8 copy + its amax, CS: column sums), so the
// eight rows are ONE basic block: the LDS reads and table look-ups of the next rows are scheduled under the arithmetic
// and the stores of the current one (with a branch per output the rows serialise on their LDS latencies).  Same
// results as epilogue_64x64<.., LUT = true>.  e_max: running |max| of the emitted values (the caller folds it).
// The bf16 results leave with non-temporal stores: each is 0.2-0.9 GB written once and read by a later kernel from HBM
// anyway, and keeping it out of the way of the operands in L2 / MALL measured -4.5 ms (bf16) / -6.5 ms (fp8) per step.
template <int EPI, int EFMT, bool C1, bool C2, bool EQ, bool CS, int WC = 64, int NIT = 8>
__device__ __forceinline__ void epilogue_full(const GemmP& p, const float* stg, const char* lut, int row0, int col0, int lane,
                                              float (&cs)[8], u32x2 (&ew)[NIT], const u32x4 (&auxr)[NIT], float e_scale,
                                              float& e_max) {
  const int c8 = lane & 7, rin = lane >> 3;
  if constexpr (WC < 64) {
    if (c8 * 8 >= WC) {  // narrow last tile column: the wave's sub-tile is 32 columns wide, lanes of chunks 4 .. 7 have no columns
      if constexpr (EQ) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) ew[it] = u32x2{0u, 0u};
      }
      return;
    }
  }
  const int gcol = col0 + c8 * 8;
  const long grow0 = row0 + rin;
  f32x2 bias[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) bias[e] = f32x2{0.f, 0.f};
  if constexpr (EPI == VDS_EPI_STORE || EPI == VDS_EPI_BIAS_GELU || EPI == VDS_EPI_GATE_RES) {
    if (p.bias) {
      const u32x4 bv = *reinterpret_cast<const u32x4*>(p.bias + gcol);
#pragma unroll
      for (int e = 0; e < 4; ++e) bias[e] = bf2_to_f2(bv[e]);
    }
  }
  [[maybe_unused]] f32x2 g[4];
  if constexpr (EPI == VDS_EPI_GATE_RES) {
    const float* gp = p.gate + (long)((row0 + p.row_base) / p.rows_per_batch) * p.ldgate + gcol;
    const f32x4 g0 = *reinterpret_cast<const f32x4*>(gp), g1 = *reinterpret_cast<const f32x4*>(gp + 4);
    g[0] = f32x2{g0[0], g0[1]}; g[1] = f32x2{g0[2], g0[3]}; g[2] = f32x2{g1[0], g1[1]}; g[3] = f32x2{g1[2], g1[3]};
  }
  [[maybe_unused]] bf16_t* c1 = reinterpret_cast<bf16_t*>(p.C) + grow0 * p.ldc + gcol;
  [[maybe_unused]] bf16_t* c2 = reinterpret_cast<bf16_t*>(p.C2) + grow0 * p.ldc2 + gcol;
  [[maybe_unused]] unsigned char* eq = p.e_q + grow0 * p.e_ldq + gcol;
  const long s1 = 8 * p.ldc, s2 = 8 * p.ldc2, sq = 8 * p.e_ldq;
  constexpr float FMAX = EFMT == 0 ? 448.0f : 57344.0f;
  const float* sp = stg + rin * EPI_LD + c8 * 8;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const f32x4 lo = *reinterpret_cast<const f32x4*>(sp + it * 8 * EPI_LD);
    const f32x4 hi = *reinterpret_cast<const f32x4*>(sp + it * 8 * EPI_LD + 4);
    const f32x2 v[4] = {f32x2{lo[0], lo[1]}, f32x2{lo[2], lo[3]}, f32x2{hi[0], hi[1]}, f32x2{hi[2], hi[3]}};
    [[maybe_unused]] u32x4 ev;
    if constexpr (EPI == VDS_EPI_STORE) {
      u32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = f2_to_bf2(v[e] + bias[e]);
      __builtin_nontemporal_store(o, reinterpret_cast<u32x4*>(c1 + it * s1));
    } else if constexpr (EPI == VDS_EPI_BIAS_GELU) {
      u32x4 o, o2;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        o[e] = f2_to_bf2(v[e] + bias[e]);
        o2[e] = f2_to_bf2(bf2_to_f2(o[e]) * lut_pair(lut, o[e]));
      }
      if constexpr (C1) __builtin_nontemporal_store(o, reinterpret_cast<u32x4*>(c1 + it * s1));
      if constexpr (C2) __builtin_nontemporal_store(o2, reinterpret_cast<u32x4*>(c2 + it * s2));
      ev = o2;
    } else if constexpr (EPI == VDS_EPI_GATE_RES) {
      const u32x4 xr = auxr[it];
      u32x4 o, o2;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const f32x2 a = v[e] + bias[e];
        o[e] = f2_to_bf2(a);
        o2[e] = f2_to_bf2(bf2_to_f2(xr[e]) + a * g[e]);
      }
      if constexpr (C1) __builtin_nontemporal_store(o, reinterpret_cast<u32x4*>(c1 + it * s1));
      __builtin_nontemporal_store(o2, reinterpret_cast<u32x4*>(c2 + it * s2));
    } else if constexpr (EPI == VDS_EPI_DGELU) {
      const u32x4 pr = auxr[it];
      u32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = f2_to_bf2(v[e] * lut_pair(lut, pr[e]));
      if constexpr (C1) __builtin_nontemporal_store(o, reinterpret_cast<u32x4*>(c1 + it * s1));
      ev = o;
    }
    if constexpr (EQ || CS) {
      f32x2 f[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) f[e] = bf2_to_f2(ev[e]);
      if constexpr (CS)
#pragma unroll
        for (int e = 0; e < 4; ++e) { cs[2 * e] += f[e][0]; cs[2 * e + 1] += f[e][1]; }
      if constexpr (EQ) {
        float q[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          e_max = fmaxf(fmaxf(e_max, fabsf(f[e][0])), fabsf(f[e][1]));
          const f32x2 sc = f[e] * e_scale;
          q[2 * e] = __builtin_amdgcn_fmed3f(sc[0], -FMAX, FMAX);
          q[2 * e + 1] = __builtin_amdgcn_fmed3f(sc[1], -FMAX, FMAX);
        }
        const u32x2 w = {fp8_cvt4<EFMT>(q[0], q[1], q[2], q[3]), fp8_cvt4<EFMT>(q[4], q[5], q[6], q[7])};
        ew[it] = w;
        *reinterpret_cast<u32x2*>(eq + it * sq) = w;
      }
    }
  }
}

// picks the specialisation for the outputs this launch has; false = not covered (the caller takes epilogue_64x64)
template <int EPI, bool EMIT, int WC = 64, int NIT = 8>
__device__ __forceinline__ bool epilogue_full_dispatch(const GemmP& p, const float* stg, const char* lut, int row0, int col0,
                                                       int lane, float (&cs)[8], u32x2 (&ew)[NIT], const u32x4 (&auxr)[NIT],
                                                       float e_scale, float& e_max) {
  const bool c1 = p.C != nullptr, c2 = p.C2 != nullptr;
  [[maybe_unused]] bool eq = false, eany = false, csum = false;
  if constexpr (EMIT) {
    eq = p.e_q != nullptr;
    eany = p.e_q || p.e_qt;
    csum = p.e_colsum != nullptr;
  }
#define VDS_FULL(E, F, A, B, Q, S) epilogue_full<E, F, A, B, Q, S, WC, NIT>(p, stg, lut, row0, col0, lane, cs, ew, auxr, e_scale, e_max)
  if constexpr (EPI == VDS_EPI_STORE) {
    VDS_FULL(EPI, 0, true, false, false, false);
    return true;
  } else if constexpr (EPI == VDS_EPI_GATE_RES) {
    if (c1) VDS_FULL(EPI, 0, true, true, false, false);
    else VDS_FULL(EPI, 0, false, true, false, false);
    return true;
  } else if constexpr (EPI == VDS_EPI_BIAS_GELU) {
    if (c1 && c2 && !eany) { VDS_FULL(EPI, 0, true, true, false, false); return true; }
    if constexpr (EMIT) {
      if (c1 && !c2 && eq && !csum && p.e_fmt == 0) { VDS_FULL(EPI, 0, true, false, true, false); return true; }
    }
    return false;
  } else if constexpr (EPI == VDS_EPI_DGELU) {
    if (c1 && !eany && !csum) { VDS_FULL(EPI, 0, true, false, false, false); return true; }
    if constexpr (EMIT) {
      if (c1 && !eany && csum) { VDS_FULL(EPI, 0, true, false, false, true); return true; }
      if (!c1 && eq && csum && p.e_fmt == 1) { VDS_FULL(EPI, 1, false, false, true, true); return true; }
      if (!c1 && eq && !csum && p.e_fmt == 1) { VDS_FULL(EPI, 1, false, false, true, false); return true; }
    }
    return false;
  } else {
    return false;
  }
#undef VDS_FULL
}

template <int LAYOUT, int EPI>
__global__ __launch_bounds__(256, 2) void gemm_kernel(GemmP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr bool A_KM = (LAYOUT == VDS_TN);
  constexpr bool B_KM = (LAYOUT != VDS_NT);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  // ---- tile id: XCD-aware (blocks b, b+8, .. share an XCD) + grouped order ------------
  const int nwg = p.tiles_m * p.tiles_n;
  int pid = blockIdx.x;
  {
    int q = nwg >> 3, r = nwg & 7, xcd = pid & 7, idx = pid >> 3;
    pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int group = GROUP_M * p.tiles_n;
  const int first_m = (pid / group) * GROUP_M;
  const int gsz = min(p.tiles_m - first_m, GROUP_M);
  const int tile_m = first_m + (pid % group) % gsz;
  const int tile_n = (pid % group) / gsz;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  // ---- K range (split-K over blockIdx.y, TN only) ---------------------------------------
  const int kt_total = (p.K + BK - 1) / BK;
  int kt_begin = 0, kt_end = kt_total;
  if (p.split_k > 1) {
    int per = (kt_total + p.split_k - 1) / p.split_k;
    kt_begin = blockIdx.y * per;
    kt_end = min(kt_total, kt_begin + per);
    if (kt_begin >= kt_end) return;
  }

  const srd_t ra = make_srd(p.A, p.a_bytes);
  const srd_t rb = make_srd(p.B, p.b_bytes);
  unsigned va[4], vb[4];
  stage_offsets<A_KM>(va, wave, lane, p.lda, m0);
  stage_offsets<B_KM>(vb, wave, lane, p.ldb, n0);
  const unsigned a_step = A_KM ? (unsigned)(BK * p.lda * 2) : BK * 2;
  const unsigned b_step = B_KM ? (unsigned)(BK * p.ldb * 2) : BK * 2;


  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // LDS: [buf0: A | B][buf1: A | B]
  stage_tile<A_KM>(ra, smem, va, kt_begin * a_step, wave, lane, p.K - kt_begin * BK);
  stage_tile<B_KM>(rb, smem + TILE_BYTES, vb, kt_begin * b_step, wave, lane, p.K - kt_begin * BK);
  VDS_WAIT_VM(0);
  __syncthreads();  // tile kt_begin landed (the DMA is invisible to the compiler: waited above)

  int cur = 0;
  for (int kt = kt_begin; kt < kt_end; ++kt) {
    if (kt + 1 < kt_end) {
      char* nxt = smem + (cur ^ 1) * 2 * TILE_BYTES;
      stage_tile<A_KM>(ra, nxt, va, (kt + 1) * a_step, wave, lane, p.K - (kt + 1) * BK);
      stage_tile<B_KM>(rb, nxt + TILE_BYTES, vb, (kt + 1) * b_step, wave, lane, p.K - (kt + 1) * BK);
    }
    const char* ta = smem + cur * 2 * TILE_BYTES;
    const char* tb = ta + TILE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 fa[4], fb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if constexpr (A_KM) fa[i] = frag_km(ta, wm * 64 + i * 16, ks, lane);
        else fa[i] = frag_kc(ta, wm * 64 + i * 16, ks, lane);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if constexpr (B_KM) fb[j] = frag_km(tb, wn * 64 + j * 16, ks, lane);
        else fb[j] = frag_kc(tb, wn * 64 + j * 16, ks, lane);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
    VDS_WAIT_VM(0);
    __syncthreads();  // next tile landed and everyone is done reading `cur`
    cur ^= 1;
  }

  // ---- epilogue: accumulators -> LDS (fp32) -> 16-byte row segments ----------------------
  float* stg = reinterpret_cast<float*>(smem) + wave * 64 * EPI_LD;
  u32x4 auxr[8];
  epilogue_prefetch<EPI>(p, m0 + wm * 64, n0 + wn * 64, lane, auxr);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        stg[(i * 16 + 4 * (lane >> 4) + r) * EPI_LD + j * 16 + (lane & 15)] = acc[i][j][r];
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): own wave's region only, no barrier needed
  __builtin_amdgcn_wave_barrier();

  float cs_unused[8];
  u32x2 ew_unused[8];
  epilogue_64x64<EPI>(p, stg, m0 + wm * 64, n0 + wn * 64, lane, cs_unused, ew_unused, auxr);
}

// =====================================================================================
// 256 x 256 x 64 tile, 8 waves (2 x 4), one workgroup per CU: the large-problem kernel.
//
//   * each wave owns 128 x 64 outputs = 2 x 2 quadrants of 64 x 32 (16 v_mfma_f32_16x16x32_bf16
//     per quadrant and K tile, 128 accumulator VGPRs);
//   * a K tile is staged as FOUR 16-KiB half-tiles (A0, A1, B0, B1): half h of A holds the rows
//     of quadrant-row h of EVERY wave (64-row groups alternate), half h of B the columns of
//     quadrant-column h of every wave (32-column groups alternate), so a K tile is consumed
//     half-tile by half-tile in 4 phases  (A0,B0) -> B1 -> A1 -> (B0 kept in registers)  and each
//     half-tile's LDS slot is recycled for the K tile two ahead as soon as it has been read;
//   * half-tiles arrive by LDS-DMA (2 x buffer_load ... lds per thread and half-tile), one
//     half-tile issued per phase, ~7 phases before its first use; the only waits in the loop
//     are counted (s_waitcnt vmcnt(10) = "all but the 5 youngest half-tiles have landed") and
//     raw s_barrier -- never vmcnt(0), never __syncthreads();
//   * every phase is a LOAD segment (fragment ds_reads, one half-tile issue, waits) and an MFMA
//     segment separated by barriers; waves 4-7 (the SIMD partners of waves 0-3) run one segment
//     behind, so on every SIMD one wave's MFMA cluster overlaps its partner's LDS reads / DMA issue.
// K tiles past the end of the contraction (and the K tail) are fetched as zeros through the SRD
// bounds, which keeps the wait counts uniform to the last iteration.
namespace big {
constexpr int BM = 256;
constexpr int HALF = 16384;                                 // one half-tile
constexpr int LDS_BYTES = 8 * 64 * EPI_LD * 4;               // 139264 >= 2 * BUF (131072)
// Geometry of the output-tile widths: WN = 256, the four-half-tile scheme described above, and WN = 128 (below).
template <int WN> struct Geo;
template <> struct Geo<256> {
  static constexpr int BN = 256, WCOLS = 64, NJ = 4, A0 = 0, A1 = HALF, B0 = 2 * HALF, B1 = 3 * HALF, BUF = 4 * HALF;
};
// WN = 128 (round 5): the body of a LAST tile column that holds at most 128 columns (N = 1152 = 4.5 x 256: the fifth column
// of every DiT-XL N = 1152 linear).  A wave owns 128 x 32 outputs; B is staged as B0 only (the 128 columns, wave column w
// = columns 32 w .. 32 w + 31), a K tile is two phases of 16 MFMAs (A0 B0 | A1, B0 from registers): 6 instead of 8 DMA pieces
// and 32 instead of 64 MFMAs per wave and K tile.  Not a launch geometry: big::gemm_kernel<.., 256> branches into it per
// workgroup (GemmP::narrow), the grid and the tile order stay those of the 256-wide tiling.
template <> struct Geo<128> {
  static constexpr int BN = 128, WCOLS = 32, NJ = 2, A0 = 0, A1 = HALF, B0 = 2 * HALF, B1 = 3 * HALF, BUF = 4 * HALF;
};

// local index (row of a k-contiguous half-tile, or 8-column chunk base of a k-major one) -> offset inside the output
// tile: groups of G consecutive indices, STRIDE apart, starting at OFF (halves alternate in groups of 64 rows / 32 columns)
template <int G, int STRIDE, int OFF>
__device__ __forceinline__ int to_tile(int local) { return (local / G) * STRIDE + OFF + (local % G); }

// ---- k-major fp8 operands (FMT 3: the weight-gradient product dW = dy^T x read straight from the token-major fp8
// copies, no transposed copies) -----------------------------------------------------------------------------------
// A half-tile is [128 tokens][128 bytes]: token row r holds the 128 output rows / columns of this half (64-row groups
// of the two wave rows, or 32-column groups of the four wave columns).  16-byte chunk c of row r is stored at chunk
// c ^ swz8(r) -- the image of csrc/attention_fp8.hip: conflict-free for ds_read_b64_tr_b8, whose 16-lane group reads an
// 8-row x 16-byte block (lane 2q + p supplies row q, bytes 8p..; lane d receives column d, row q in byte q).  A
// fragment = the operand of v_mfma_f32_16x16x128_f8f6f4 for the 16 columns of block db: lane (d = l & 15, g = l >> 4)
// gets tokens 32g .. 32g+31 of column 16 db + d from four such reads.
__device__ __forceinline__ int swz8(int row) { return ((row >> 1) & 3) | ((row >> 3) & 4); }
__device__ __forceinline__ i32x8 frag_km8(const char* tile, int db, int lane) {
  const int m = lane & 15, g = lane >> 4, q = m >> 1, pp = m & 1;
  const int st = ((q >> 1) & 3) | ((g & 1) << 2);  // swz8(32 g + 8 t + q), the same for every t
  const char* p = tile + (32 * g + q) * 128 + ((db ^ st) << 4) + 8 * pp;
  i32x8 r;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    typedef __attribute__((ext_vector_type(2))) int i32x2_t;
    const i32x2_t w = __builtin_amdgcn_ds_read_tr8_b64_v2i32((__attribute__((address_space(3))) i32x2_t*)LDS_PTR(p + t * 8 * 128));
    r[2 * t] = w[0];
    r[2 * t + 1] = w[1];
  }
  return r;
}
// source byte offsets of the two 1-KiB pieces (8 token rows each) this wave stages of a k-major fp8 half-tile;
// ld_bytes = row stride of the [tokens][columns] tensor in bytes
template <int G, int STRIDE, int OFF>
__device__ __forceinline__ void half_offsets_km8(unsigned (&voff)[2], int (&kchunk)[2], int wave, int lane, long ld_bytes,
                                                 int origin) {
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int row = (wave * 2 + j) * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ swz8(row);
    kchunk[j] = 0;
    voff[j] = (unsigned)((long)row * ld_bytes + origin + to_tile<G, STRIDE, OFF>(chunk * 16));
  }
}

// per-lane source byte offsets of the NP = 2 1-KiB pieces this wave stages of one 16-KiB half-tile
template <bool KMAJOR, int G, int STRIDE, int OFF, int NP>
__device__ __forceinline__ void half_offsets(unsigned (&voff)[NP], int (&kchunk)[NP], int wave, int lane, long ld,
                                             int origin) {
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    const int q = wave * NP + j;
    if constexpr (!KMAJOR) {
      const int row = q * 8 + (lane >> 3);
      const int chunk = swz_kc(row, lane & 7);
      kchunk[j] = chunk * 8;
      voff[j] = (unsigned)(((long)(origin + to_tile<G, STRIDE, OFF>(row)) * ld + chunk * 8) * 2);
    } else {
      static_assert(NP == 2, "half-tiles are staged as two 1-KiB pieces per wave");
      const int krow = q * 4 + (lane >> 4);
      const int pc = lane & 15;
      const int chunk = (((pc >> 1) ^ swz_km(krow)) << 1) | (pc & 1);
      kchunk[j] = 0;
      voff[j] = (unsigned)(((long)krow * ld + origin + to_tile<G, STRIDE, OFF>(chunk * 8)) * 2);
    }
  }
}

template <bool KMAJOR, int NP = 2>
__device__ __forceinline__ void issue_half(srd_t rsrc, char* slot, const unsigned (&voff)[NP],
                                           const int (&kchunk)[NP], unsigned koff, int krem, int wave) {
  const unsigned base = lds_addr_of(slot) + wave * (NP * 1024);
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    unsigned off = voff[j] + koff;
    if constexpr (!KMAJOR) {
      if (kchunk[j] >= krem) off = 0xfffffff0u;  // K tail / tiles past the range: zeros
    } else {
      if (krem <= 0) off = 0xfffffff0u;
    }
    lds_dma16(rsrc, base + j * 1024, off);
  }
}

// FMT 0: bf16 operands.  FMT 1 / 2: OCP fp8 operands (A e4m3 / e5m2, B e4m3), NT only: a 128-byte tile row is
// 128 k instead of 64, one v_mfma_f32_16x16x128_f8f6f4 replaces two 16x16x32 bf16 MFMAs in the same 32 cycles --
// bytes staged, LDS reads and the phase schedule are identical, the contraction per K tile doubles.  The
// addressing below counts in 2-byte units (p.K, lda, ldb = bytes / 2).
// One output tile: rows m0 .. m0 + 255, columns n0 .. n0 + BN - 1, K tiles kt_begin .. kt_end - 1.
template <int LAYOUT, int EPI, int FMT, int WN>
__device__ __forceinline__ void gemm_tile(const GemmP& p, char* smem, const int m0, const int n0, const int kt_begin,
                                          const int kt_end) {
  static_assert(FMT == 0 || (FMT != 3 && LAYOUT == VDS_NT) || (FMT == 3 && LAYOUT == VDS_TN),
                "fp8 operands: k-contiguous (FMT 1 / 2, NT) or both k-major (FMT 3, TN: the weight gradient)");
  using G = Geo<WN>;
  constexpr int WCOLS = G::WCOLS, NJ = G::NJ, BUF = G::BUF;
  constexpr int SLOT_A0 = G::A0, SLOT_A1 = G::A1, SLOT_B0 = G::B0, SLOT_B1 = G::B1;
  constexpr int NPB1 = 2;  // 1-KiB pieces per wave of the B1 half-tile
  constexpr bool A_KM = (LAYOUT == VDS_TN);
  constexpr bool B_KM = (LAYOUT != VDS_NT);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const srd_t ra = make_srd(p.A, p.a_bytes);
  const srd_t rb = make_srd(p.B, p.b_bytes);
  constexpr bool USE_LUT = EPI == VDS_EPI_BIAS_GELU || EPI == VDS_EPI_DGELU;
  char* const ring = smem + (USE_LUT ? LUT_BYTES : 0);  // stage ring / epilogue staging area
  if (USE_LUT) {
    // the GELU table (16 KiB) sits at LDS address 0, in front of the ring: two 1-KiB pieces per wave, older than
    // every operand DMA, so the counted waits of the main loop cover them
    const srd_t rl = make_srd(g_gelu_lut[EPI == VDS_EPI_DGELU ? 1 : 0], LUT_BYTES);
    const unsigned lb = lds_addr_of(smem) + wave * 2048;
    lds_dma16(rl, lb, (unsigned)(wave * 2048 + lane * 16));
    lds_dma16(rl, lb + 1024, (unsigned)(wave * 2048 + 1024 + lane * 16));
  }
  unsigned va[2][2], vb0[2], vb1[NPB1];
  int ca[2][2], cb0[2], cb1[NPB1];
  if constexpr (FMT == 3) {  // (p.lda / p.ldb count 2-byte units, origins are byte columns)
    half_offsets_km8<64, 128, 0>(va[0], ca[0], wave, lane, 2 * p.lda, m0);
    half_offsets_km8<64, 128, 64>(va[1], ca[1], wave, lane, 2 * p.lda, m0);
    if constexpr (WN == 128) {
      half_offsets_km8<32, 32, 0>(vb0, cb0, wave, lane, 2 * p.ldb, n0);
    } else {
      half_offsets_km8<32, 64, 0>(vb0, cb0, wave, lane, 2 * p.ldb, n0);
      half_offsets_km8<32, 64, 32>(vb1, cb1, wave, lane, 2 * p.ldb, n0);
    }
  } else {
  half_offsets<A_KM, 64, 128, 0, 2>(va[0], ca[0], wave, lane, p.lda, m0);
  half_offsets<A_KM, 64, 128, 64, 2>(va[1], ca[1], wave, lane, p.lda, m0);
  }
  if constexpr (FMT == 3) {
  } else if constexpr (WN == 128) {
    half_offsets<B_KM, 32, 32, 0, 2>(vb0, cb0, wave, lane, p.ldb, n0);
  } else {
    half_offsets<B_KM, 32, 64, 0, 2>(vb0, cb0, wave, lane, p.ldb, n0);
    half_offsets<B_KM, 32, 64, 32, 2>(vb1, cb1, wave, lane, p.ldb, n0);
  }
  // (FMT 3: a K tile is 128 token rows of lda / ldb BYTES = 2 BK rows in the 2-byte units the addressing counts in)
  const unsigned a_step = A_KM ? (unsigned)((FMT == 3 ? 2 * BK : BK) * p.lda * 2) : BK * 2;
  const unsigned b_step = B_KM ? (unsigned)((FMT == 3 ? 2 * BK : BK) * p.ldb * 2) : BK * 2;

  // issue half-tile `which` (0 A0, 1 B0, 2 B1, 3 A1) of K tile T into its slot of buffer `par`
  // (= (T - kt_begin) & 1, passed as a compile-time constant so that every LDS address is base + immediate)
  auto issue = [&](int T, int which, int par) {
    const int krem = (T < kt_end) ? p.K - T * BK : 0;
    char* buf = ring + par * BUF;
    if (which == 0) issue_half<A_KM>(ra, buf + SLOT_A0, va[0], ca[0], (unsigned)T * a_step, krem, wave);
    else if (which == 3) issue_half<A_KM>(ra, buf + SLOT_A1, va[1], ca[1], (unsigned)T * a_step, krem, wave);
    else if (which == 1) issue_half<B_KM>(rb, buf + SLOT_B0, vb0, cb0, (unsigned)T * b_step, krem, wave);
    else if constexpr (WN != 128) issue_half<B_KM, NPB1>(rb, buf + SLOT_B1, vb1, cb1, (unsigned)T * b_step, krem, wave);
  };

  f32x4 acc[8][NJ];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      // pins the 128 zeroing moves HERE, before the pipeline fill below: hipcc otherwise sinks them behind the first
      // vmcnt wait + barrier, onto the critical path of every tile (the wait hides them for free): fp8 NT + dgrad per
      // block 4.13 -> 4.03 ms, bf16 unchanged.  (Not in the TN kernels: there the pin costs registers -- 85-104 spilled
      // VGPRs, the fp8 weight-gradient GEMMs 2.5x slower.)
      if constexpr (LAYOUT != VDS_TN) asm volatile("" : "+v"(acc[i][j]));
    }

  constexpr int KS = FMT == 0 ? 2 : 1;  // MFMA k-steps per K tile
  using frag_t = std::conditional_t<FMT == 0, bf16x8, i32x8>;
  // The product is issued TRANSPOSED (round 4): the B-tile fragment is the MFMA's A operand and the A-tile fragment its B
  // operand, so that a 16 x 16 accumulator block holds, in lane (c = l & 15, g = l >> 4), the FOUR CONSECUTIVE COLUMNS
  // n = 4 g .. 4 g + 3 of output row m = c (instead of four consecutive rows of one column): the epilogue stages a
  // block with one 16-byte LDS store per lane instead of four 4-byte ones.  Same products, same k order: bit-identical.
  auto mma = [&](const frag_t& a, const frag_t& b, f32x4 c) {
    if constexpr (FMT == 0) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a, c, 0, 0, 0);
    // cbsz: format of the MFMA's A operand (0 e4m3, 1 e5m2) = our B tile (always e4m3), blgp: format of its B operand
    // = our A tile; block scales unused (0 selects the unscaled form)
    else return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(b, a, c, 0, FMT >= 2 ? 1 : 0, 0, 0, 0, 0);
  };
  auto read_a4 = [&](const char* slot, frag_t (&fa)[4][KS]) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        if constexpr (FMT == 3) fa[i][ks] = frag_km8(slot, wr * 4 + i, lane);
        else if constexpr (FMT != 0) fa[i][ks] = frag_kc8(slot, wr * 64 + i * 16, lane);
        else if constexpr (A_KM) fa[i][ks] = frag_km(slot, wr * 64 + i * 16, ks, lane);
        else fa[i][ks] = frag_kc(slot, wr * 64 + i * 16, ks, lane);
      }
  };
  auto read_b2 = [&](const char* slot, frag_t (&fb)[2][KS]) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        if constexpr (FMT == 3) fb[j][ks] = frag_km8(slot, wc * 2 + j, lane);
        else if constexpr (FMT != 0) fb[j][ks] = frag_kc8(slot, wc * 32 + j * 16, lane);
        else if constexpr (B_KM) fb[j][ks] = frag_km(slot, wc * 32 + j * 16, ks, lane);
        else fb[j][ks] = frag_kc(slot, wc * 32 + j * 16, ks, lane);
      }
  };

  if constexpr (WN == 256) {
  // ---- TWO phases of 32 MFMAs per K tile -- quadrant row 0 <- A0, B0, B1; quadrant row 1 <- A1 with
  // both B halves kept in registers -- i.e. 4 barriers per K tile instead of 8 and the same fragment registers.  Issue
  // order per K tile: A0 B0 B1 | A1; A1(T+1) is issued in phase 0 of tile T, A0 B0 B1 of tile T+2 in phase 1; each wait
  // leaves 8 pieces (four half-tiles) in flight and retires a half-tile one phase before it is read; a slot is
  // re-staged one phase after its last read.  Against the four-phase loop of rounds 1-3 (one quadrant per phase; removed in
  // round 6), same-process A/B, B = 12 (profiles/r04/gemm_2phase_vs_4phase.log): qkv forward 0.705 -> 0.668 ms, q_cross
  // 0.252 -> 0.237, fc1 dgrad 0.945 -> 0.908, 8192^3 1.40 -> 1.50 PFLOP/s; results bit-identical.
  issue(kt_begin, 0, 0); issue(kt_begin, 1, 0); issue(kt_begin, 2, 0); issue(kt_begin, 3, 0);
  issue(kt_begin + 1, 0, 1); issue(kt_begin + 1, 1, 1); issue(kt_begin + 1, 2, 1);
  VDS_WAIT_VM(8);
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();  // waves 4-7 run one segment behind
  frag_t fa[4][KS], fb0[2][KS], fb1[2][KS];
  auto quad = [&](int qa, int qb, frag_t (&fb)[2][KS]) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[qa * 4 + i][qb * 2 + j] = mma(fa[i][ks], fb[j][ks], acc[qa * 4 + i][qb * 2 + j]);
  };
  auto k_tile = [&](int T, auto PAR) {
    constexpr int par = decltype(PAR)::value;
    const char* buf = ring + par * BUF;
    // ---- phase 0: quadrant row 0 <- A0, B0, B1 ----
    read_a4(buf + SLOT_A0, fa);
    read_b2(buf + SLOT_B0, fb0);
    read_b2(buf + SLOT_B1, fb1);
    issue(T + 1, 3, par ^ 1);
    VDS_WAIT_LGKM0();
    VDS_WAIT_VM(8);  // A1(T) landed
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_setprio(1);
    quad(0, 0, fb0);
    quad(0, 1, fb1);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_s_barrier();
    // ---- phase 1: quadrant row 1 <- A1 (B0, B1 in registers) ----
    read_a4(buf + SLOT_A1, fa);
    issue(T + 2, 0, par);
    issue(T + 2, 1, par);
    issue(T + 2, 2, par);
    VDS_WAIT_LGKM0();
    VDS_WAIT_VM(8);  // A0, B0, B1 of tile T+1 landed
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_setprio(1);
    quad(1, 1, fb1);
    quad(1, 0, fb0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_s_barrier();
  };
  for (int T = kt_begin; T < kt_end; T += 2) {
    k_tile(T, std::integral_constant<int, 0>{});
    if (T + 1 < kt_end) k_tile(T + 1, std::integral_constant<int, 1>{});
  }
  } else {
  static_assert(WN == 128, "tile widths: 256, and 128 for a narrow last tile column");
  // ---- narrow last tile column: A0 B0 | A1 per K tile (6 pieces per wave), two phases of 16 MFMAs; the waits leave 6
  // pieces (three half-tiles) in flight; retire / re-stage distances as in the two-phase loop above
  issue(kt_begin, 0, 0); issue(kt_begin, 1, 0); issue(kt_begin, 3, 0);
  issue(kt_begin + 1, 0, 1); issue(kt_begin + 1, 1, 1);
  VDS_WAIT_VM(6);
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();  // waves 4-7 run one segment behind
  frag_t fa[4][KS], fb0[2][KS];
  auto quad = [&](int qa) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[qa * 4 + i][j] = mma(fa[i][ks], fb0[j][ks], acc[qa * 4 + i][j]);
  };
  auto k_tile = [&](int T, auto PAR) {
    constexpr int par = decltype(PAR)::value;
    const char* buf = ring + par * BUF;
    read_a4(buf + SLOT_A0, fa);
    read_b2(buf + SLOT_B0, fb0);
    issue(T + 1, 3, par ^ 1);
    VDS_WAIT_LGKM0();
    VDS_WAIT_VM(6);  // A1(T) landed
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_setprio(1);
    quad(0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_s_barrier();
    read_a4(buf + SLOT_A1, fa);
    issue(T + 2, 0, par);
    issue(T + 2, 1, par);
    VDS_WAIT_LGKM0();
    VDS_WAIT_VM(6);  // A0, B0 of tile T+1 landed
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_setprio(1);
    quad(1);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_s_barrier();
  };
  for (int T = kt_begin; T < kt_end; T += 2) {
    k_tile(T, std::integral_constant<int, 0>{});
    if (T + 1 < kt_end) k_tile(T + 1, std::integral_constant<int, 1>{});
  }
  }
  // ---- epilogue: two 64-row quadrant rows per wave through the wave's private staging area ----
  // The aux tiles (residual / pre-activation) of BOTH quadrant rows are requested here, before the drain of the
  // operand pipeline: one exposed HBM round trip per output tile instead of one per quadrant row (with one workgroup
  // per CU nothing else hides it).  The drain waits for everything but these 16 loads (loads return in order).
  constexpr int NIT = 8;                  // row groups of 8 rows per staging step
  constexpr int NSTEP = 16 / NIT;         // steps per wave tile of 128 rows
  constexpr int RSTEP = 8 * NIT;          // rows per step
  u32x4 auxr2[NSTEP][NIT];
#pragma unroll
  for (int st = 0; st < NSTEP; ++st)
    epilogue_prefetch<EPI, G::WCOLS, NIT>(p, m0 + wr * 128 + st * RSTEP, n0 + wc * G::WCOLS, lane, auxr2[st]);
  if (wr == 0) __builtin_amdgcn_s_barrier();  // re-align the two wave groups
  // the zero-fill tail DMAs target LDS the epilogue reuses
  // (a wave whose sub-tile sticks out of the matrix may have issued fewer than 16: it drains everything)
  const bool aux_all = (EPI == VDS_EPI_GATE_RES || EPI == VDS_EPI_DGELU) && m0 + wr * 128 + 128 <= p.M &&
                       n0 + (wc + 1) * G::WCOLS <= p.N;
  if (aux_all) VDS_WAIT_VM(16);
  else VDS_WAIT_VM(0);
  __builtin_amdgcn_s_barrier();

  constexpr int STG_BASE = 0, STG_WAVE = RSTEP * EPI_LD;  // (floats per wave)
  float* stg = reinterpret_cast<float*>(ring + STG_BASE) + wave * STG_WAVE;
  float dq = 1.0f;
  if constexpr (FMT != 0) dq = (p.sa ? *p.sa : 1.0f) * (p.sb ? *p.sb : 1.0f);
  // EMIT also serves the bf16 DGELU GEMM: no fp8 copies there (e_q / e_qt are null), only the column sums of its
  // result -- the fc1 bias gradient -- folded into the epilogue instead of a separate pass over the [tokens, 4D] tensor
  constexpr bool EMIT = (FMT != 0 && (EPI == VDS_EPI_BIAS_GELU || EPI == VDS_EPI_DGELU)) || (FMT == 0 && EPI == VDS_EPI_DGELU);
  float cs[8];  // EMIT: per-lane partial column sums of the emitted result over both quadrant rows
  u32x2 ew[NSTEP][NIT];  // EMIT: the fp8 bytes of every staging step (for the transposed copy)
#pragma unroll
  for (int e = 0; e < 8; ++e) cs[e] = 0.f;
  [[maybe_unused]] float e_scale = 1.0f, e_max = 0.f;
  if constexpr (EMIT) {
    const float fmax = p.e_fmt == 0 ? 448.0f : 57344.0f;
    const float ain = ((p.e_q || p.e_qt) && p.e_amax_in) ? *p.e_amax_in : 0.f;
    e_scale = ain > 0.f ? fmax / ain : 1.0f;
    if ((p.e_q || p.e_qt) && p.e_dq_out && m0 == 0 && n0 == 0 && tid == 0) *p.e_dq_out = ain > 0.f ? ain / fmax : 1.0f;
  }
  const char* lut = smem;
#pragma unroll
  for (int qa = 0; qa < NSTEP; ++qa) {
    const u32x4 (&auxr)[NIT] = auxr2[qa];
    const int row0 = m0 + wr * 128 + qa * RSTEP, col0 = n0 + wc * WCOLS;
#pragma unroll
    for (int i = 0; i < RSTEP / 16; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) {  // lane (c, g): row i * 16 + c, columns j * 16 + 4 g .. + 3 (see `mma`)
        f32x4 v = acc[qa * (RSTEP / 16) + i][j];
        if constexpr (FMT != 0) v *= dq;
        *reinterpret_cast<f32x4*>(stg + (i * 16 + (lane & 15)) * EPI_LD + j * 16 + 4 * (lane >> 4)) = v;
      }
    VDS_WAIT_LGKM0();
    __builtin_amdgcn_wave_barrier();
    // sub-tiles completely inside the matrix (all but the last row / column of tiles) take the lean path
    bool full = row0 + RSTEP <= p.M && col0 + WCOLS <= p.N;
    if constexpr (EPI == VDS_EPI_GATE_RES) full = full && ((row0 + p.row_base) % p.rows_per_batch) + RSTEP <= p.rows_per_batch;
    if constexpr (EPI == VDS_EPI_F32) full = false;
    bool done = false;
    if (full) done = epilogue_full_dispatch<EPI, EMIT, WCOLS, NIT>(p, stg, lut, row0, col0, lane, cs, ew[qa], auxr, e_scale, e_max);
    if (!done) epilogue_64x64<EPI, EMIT, USE_LUT, WCOLS, NIT>(p, stg, row0, col0, lane, cs, ew[qa], auxr, lut);
    __builtin_amdgcn_wave_barrier();  // the staging area is rewritten by the next quadrant row
  }
  if constexpr (EMIT) {
    if ((p.e_q || p.e_qt) && p.e_amax_out) {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) e_max = fmaxf(e_max, __shfl_xor(e_max, o));
      unsigned* a = reinterpret_cast<unsigned*>(p.e_amax_out);
      if (lane == 0 && __float_as_uint(e_max) > __atomic_load_n(a, __ATOMIC_RELAXED)) atomicMax(a, __float_as_uint(e_max));
    }
  }
  if constexpr (EMIT) {
    if (p.e_qt) {
      // transposed fp8 copy of the wave's 128 x 64 outputs: the staging area (fp32 tiles consumed) becomes a
      // [128 m][64 k] byte tile (72-byte rows), read back as 4x4 byte blocks, transposed in registers (v_perm_b32)
      // and stored as 128 contiguous bytes per k row and instruction
      constexpr int TLD = 72;
      unsigned char* bt = reinterpret_cast<unsigned char*>(stg);
      const int c8 = lane & 7, rin = lane >> 3;
#pragma unroll
      for (int qa = 0; qa < NSTEP; ++qa)
#pragma unroll
        for (int it = 0; it < NIT; ++it) *reinterpret_cast<u32x2*>(bt + (qa * RSTEP + it * 8 + rin) * TLD + c8 * 8) = ew[qa][it];
      VDS_WAIT_LGKM0();
      __builtin_amdgcn_wave_barrier();
      const int mq = lane & 31;
      const long m = m0 + wr * 128 + 4 * mq;
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int kq = (lane >> 5) + 2 * it;
        unsigned r[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = *reinterpret_cast<const unsigned*>(bt + (4 * mq + i) * TLD + 4 * kq);
        const unsigned t0 = __builtin_amdgcn_perm(r[1], r[0], 0x05010400u), t1 = __builtin_amdgcn_perm(r[1], r[0], 0x07030602u);
        const unsigned t2 = __builtin_amdgcn_perm(r[3], r[2], 0x05010400u), t3 = __builtin_amdgcn_perm(r[3], r[2], 0x07030602u);
        const unsigned c[4] = {__builtin_amdgcn_perm(t2, t0, 0x05040100u), __builtin_amdgcn_perm(t2, t0, 0x07060302u),
                               __builtin_amdgcn_perm(t3, t1, 0x05040100u), __builtin_amdgcn_perm(t3, t1, 0x07060302u)};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int col = n0 + wc * WCOLS + 4 * kq + j;
          if (col >= p.N || m >= p.M || 4 * kq + j >= WCOLS) continue;
          unsigned char* dst = p.e_qt + (long)col * p.e_ldqt + m;
          if (m + 4 <= p.M) *reinterpret_cast<unsigned*>(dst) = c[j];
          else
            for (int i = 0; i < (int)(p.M - m); ++i) dst[i] = (unsigned char)(c[j] >> (8 * i));
        }
      }
      VDS_WAIT_LGKM0();
      __builtin_amdgcn_wave_barrier();
    }
    if (p.e_colsum) {
      // column sums of the workgroup's 256 rows: lanes sharing a column chunk differ in lane bits 3..5, the two
      // wave rows meet through LDS -> one atomic per column and workgroup
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        cs[e] += __shfl_xor(cs[e], 8);
        cs[e] += __shfl_xor(cs[e], 16);
        cs[e] += __shfl_xor(cs[e], 32);
      }
      if (wr == 1 && lane < 8)
#pragma unroll
        for (int e = 0; e < 8; ++e) stg[lane * 8 + e] = cs[e];
      __syncthreads();
      if (wr == 0 && lane < WCOLS / 8) {
        const float* other = reinterpret_cast<const float*>(ring + STG_BASE) + (wave + 4) * STG_WAVE;
        const int gcol = n0 + wc * WCOLS + lane * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (gcol + e < p.N) atomicAdd(p.e_colsum + gcol + e, cs[e] + other[lane * 8 + e]);
      }
    }
  }
}

template <int LAYOUT, int EPI, int FMT = 0>
__global__ __launch_bounds__(512, 2) void gemm_kernel(GemmP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // ---- tile id: XCD-aware + grouped order (as the 128^2 kernel) ---------------------------
  const int nwg = p.tiles_m * p.tiles_n;
  int pid = blockIdx.x;
  {
    int q = nwg >> 3, r = nwg & 7, xcd = pid & 7, idx = pid >> 3;
    pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int group = p.group_m * p.tiles_n;
  const int first_m = (pid / group) * p.group_m;
  const int gsz = min(p.tiles_m - first_m, p.group_m);
  const int tile_m = first_m + (pid % group) % gsz;
  const int tile_n = (pid % group) / gsz;
  const int m0 = tile_m * BM, n0 = tile_n * 256;

  const int kt_total = (p.K + BK - 1) / BK;
  int kt_begin = 0, kt_end = kt_total;
  if (p.split_k > 1) {
    int per = (kt_total + p.split_k - 1) / p.split_k;
    kt_begin = blockIdx.y * per;
    kt_end = min(kt_total, kt_begin + per);
    if (kt_begin >= kt_end) return;
  }
  if (p.narrow && tile_n == p.tiles_n - 1) {
    gemm_tile<LAYOUT, EPI, FMT, 128>(p, smem, m0, n0, kt_begin, kt_end);
    return;
  }
  gemm_tile<LAYOUT, EPI, FMT, 256>(p, smem, m0, n0, kt_begin, kt_end);
}

template <int LAYOUT, int EPI, int FMT = 0>
int launch(const GemmP& p, hipStream_t s) {
  constexpr bool USE_LUT = EPI == VDS_EPI_BIAS_GELU || EPI == VDS_EPI_DGELU;
  constexpr int LDS_TOTAL = LDS_BYTES + (USE_LUT ? LUT_BYTES : 0);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<LAYOUT, EPI, FMT>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL);
    attr_set = true;
  }
  if constexpr (USE_LUT) {
    if (!ensure_gelu_lut()) return VDS_ERR_LAUNCH;
  }
  dim3 grid(p.tiles_m * p.tiles_n, p.split_k > 1 ? p.split_k : 1, 1);
  const double k = FMT != 0 ? p.prof_k : (double)p.K;
  vdsprof::Scope ps(FMT != 0 ? VDS_PROF_GEMM_FP8 : LAYOUT == VDS_NT ? VDS_PROF_GEMM_NT : LAYOUT == VDS_NN ? VDS_PROF_GEMM_NN
                                                                                                       : VDS_PROF_GEMM_TN,
                    s, 2.0 * p.M * p.N * k, (FMT != 0 ? 1.0 : 2.0) * ((double)p.M * k + (double)p.N * k) + 2.0 * (double)p.M * p.N);
  hipLaunchKernelGGL((gemm_kernel<LAYOUT, EPI, FMT>), grid, dim3(512), LDS_TOTAL, s, p);
  return hipGetLastError() == hipSuccess ? VDS_OK : VDS_ERR_LAUNCH;
}

}  // namespace big

// =====================================================================================
// 256 x 128 x 32 tile, 4 waves (2 x 2), TWO workgroups per CU: the kernel for N = 1152-class problems.
//
// Why a third tiling: at the DiT-XL shapes the 256^2 kernel (one workgroup per CU) leaves two things on the table
// when N = 1152 and / or K = 1152 -- 1152 = 4.5 tiles of 256 (10 % of the MFMA work is padding) and the per-tile
// prologue + epilogue (~4.5 K-tile times, 16-19 % of a K = 1152 launch) is exposed because nothing else runs on the
// CU meanwhile.  Here 1152 = 9 x 128 exactly, and the two co-resident workgroups are not synchronised with each
// other, so one's epilogue / pipeline fill runs under the other's main loop.
//   * each wave owns 128 x 64 outputs (8 x 4 v_mfma_f32_16x16x32_bf16 tiles, 128 accumulator VGPRs) -- the same
//     wave tile, hence the same LDS reads per MFMA, as the 256^2 kernel;
//   * BK = 32: a stage is A[256][32] + B[128][32] = 24 KiB; three stages (72 KiB, so that two workgroups fit in the
//     160 KiB of a CU) give a prefetch distance of two K steps with ONE counted wait (vmcnt(6) = "all but the
//     youngest stage have landed") and one barrier per K step;
//   * k-contiguous stages have 64-byte rows: the bank swizzle is chunk ^ T[(row >> 2) & 3], T = {0, 2, 3, 1}, which
//     makes every ds_read_b128 lane group of a 16-row x 32-k fragment read hit 16 different quad-banks (applied to
//     the per-lane SOURCE address of the LDS-DMA and again on the fragment read, like the 128-byte-row swizzle);
//     k-major stages ([32 k][128] halves) use the 128^2 kernel's format and ds_read_b64_tr_b16.
namespace mid {
constexpr int BM = 256, BN = 128, BK = 32;
constexpr int A_BYTES = 16384, B_BYTES = 8192, STAGE = A_BYTES + B_BYTES;
constexpr int LDS_BYTES = 3 * STAGE;  // 73728 >= the epilogue's 4 x 64 x EPI_LD floats (69632)
static_assert(LDS_BYTES >= 4 * 64 * EPI_LD * 4, "epilogue staging must fit in the stage ring");

__device__ __forceinline__ int swz32(int row) {
  const int h = (row >> 2) & 3;
  return (((h >> 1) ^ h) & 1) << 1 | (h >> 1);
}

// per-lane source byte offsets of the NP 1-KiB pieces this wave stages of one operand stage
//   k-contiguous [ROWS][32 k]: piece q = 16 rows; k-major [32 k][ROWS cols] as ROWS/128 tiles of [32][128]: piece = 4 k rows
template <bool KMAJOR, int ROWS>
__device__ __forceinline__ void offsets(unsigned (&voff)[ROWS / 64], int (&kchunk)[ROWS / 64], int wave, int lane,
                                        long ld, int origin) {
  constexpr int NP = ROWS / 64;
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    const int q = wave * NP + j;
    if constexpr (!KMAJOR) {
      const int row = q * 16 + (lane >> 2);
      const int chunk = (lane & 3) ^ swz32(row);
      kchunk[j] = chunk * 8;
      voff[j] = (unsigned)(((long)(origin + row) * ld + chunk * 8) * 2);
    } else {
      const int t = q >> 3, krow = (q & 7) * 4 + (lane >> 4);  // tile t = columns t*128..
      const int pc = lane & 15;
      const int chunk = (((pc >> 1) ^ swz_km(krow)) << 1) | (pc & 1);
      kchunk[j] = krow;
      voff[j] = (unsigned)(((long)krow * ld + origin + t * 128 + chunk * 8) * 2);
    }
  }
}

template <bool KMAJOR, int NP>
__device__ __forceinline__ void issue(srd_t rsrc, char* slot, const unsigned (&voff)[NP], const int (&kchunk)[NP],
                                      unsigned koff, int krem, int wave) {
  const unsigned base = lds_addr_of(slot) + wave * NP * 1024;
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    unsigned off = voff[j] + koff;
    if (kchunk[j] >= krem) off = 0xfffffff0u;  // K tail / steps past the range: zeros (keeps the wait counts uniform)
    lds_dma16(rsrc, base + j * 1024, off);
  }
}

__device__ __forceinline__ bf16x8 frag_kc32(const char* tile, int row0, int lane) {
  const int row = row0 + (lane & 15);
  const int pc = (lane >> 4) ^ swz32(row);
  return *reinterpret_cast<const bf16x8*>(tile + row * 64 + pc * 16);
}

template <int LAYOUT, int EPI>
__global__ __launch_bounds__(256, 2) void gemm_kernel(GemmP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr bool A_KM = (LAYOUT == VDS_TN);
  constexpr bool B_KM = (LAYOUT != VDS_NT);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  const int nwg = p.tiles_m * p.tiles_n;
  int pid = blockIdx.x, split = blockIdx.y;
  if (p.split_k > 1 && p.joint_xcd) {
    // split-K (weight gradients): the tiles of ONE split stream the same token range, so the (split, tile) list as a
    // whole -- split-major -- is cut into 8 contiguous chunks, one per XCD (workgroups are handed to the XCDs round
    // robin in flattened-id order): an XCD then streams one token range for a compact block of output tiles instead
    // of a slice of every split
    const int total = nwg * p.split_k;
    int L = blockIdx.y * nwg + blockIdx.x;
    const int q = total >> 3, r = total & 7, xcd = L & 7, idx = L >> 3;
    L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    split = L / nwg;
    pid = L - split * nwg;
  } else {
    int q = nwg >> 3, r = nwg & 7, xcd = pid & 7, idx = pid >> 3;
    pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int group = p.group_m * p.tiles_n;
  const int first_m = (pid / group) * p.group_m;
  const int gsz = min(p.tiles_m - first_m, p.group_m);
  const int tile_m = first_m + (pid % group) % gsz;
  const int tile_n = (pid % group) / gsz;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const int kt_total = (p.K + BK - 1) / BK;
  int kt_begin = 0, kt_end = kt_total;
  if (p.split_k > 1) {
    int per = (kt_total + p.split_k - 1) / p.split_k;
    kt_begin = split * per;
    kt_end = min(kt_total, kt_begin + per);
    if (kt_begin >= kt_end) return;
  }

  const srd_t ra = make_srd(p.A, p.a_bytes);
  const srd_t rb = make_srd(p.B, p.b_bytes);
  unsigned va[4], vb[2];
  int ca[4], cb[2];
  offsets<A_KM, 256>(va, ca, wave, lane, p.lda, m0);
  offsets<B_KM, 128>(vb, cb, wave, lane, p.ldb, n0);
  const unsigned a_step = A_KM ? (unsigned)(BK * p.lda * 2) : BK * 2;
  const unsigned b_step = B_KM ? (unsigned)(BK * p.ldb * 2) : BK * 2;

  auto stage_in = [&](int T, int slot) {
    const int krem = (T < kt_end) ? p.K - T * BK : 0;
    char* st = smem + slot * STAGE;
    issue<A_KM, 4>(ra, st, va, ca, (unsigned)T * a_step, krem, wave);
    issue<B_KM, 2>(rb, st + A_BYTES, vb, cb, (unsigned)T * b_step, krem, wave);
  };

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  stage_in(kt_begin, 0);
  stage_in(kt_begin + 1, 1);
  VDS_WAIT_VM(6);  // the first stage has landed (the 6 youngest pieces are the second)
  __builtin_amdgcn_s_barrier();

  auto k_step = [&](int T, auto SLOT) {
    constexpr int slot = decltype(SLOT)::value;
    stage_in(T + 2, (slot + 2) % 3);  // the slot every wave finished reading before the last barrier
    const char* ta = smem + slot * STAGE;
    const char* tb = ta + A_BYTES;
    bf16x8 fa[8], fb[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if constexpr (A_KM) fa[i] = frag_km(ta + wm * B_BYTES, i * 16, 0, lane);  // [32][128] half wm of the 256 rows
      else fa[i] = frag_kc32(ta, wm * 128 + i * 16, lane);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if constexpr (B_KM) fb[j] = frag_km(tb, wn * 64 + j * 16, 0, lane);
      else fb[j] = frag_kc32(tb, wn * 64 + j * 16, lane);
    }
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    VDS_WAIT_VM(6);  // stage T+1 landed (this wave's pieces; the barrier publishes everyone's)
    __builtin_amdgcn_s_barrier();
  };
  for (int T = kt_begin; T < kt_end; T += 3) {
    k_step(T, std::integral_constant<int, 0>{});
    if (T + 1 < kt_end) k_step(T + 1, std::integral_constant<int, 1>{});
    if (T + 2 < kt_end) k_step(T + 2, std::integral_constant<int, 2>{});
  }
  VDS_WAIT_VM(0);  // the zero-fill tail DMAs target LDS the epilogue reuses
  __builtin_amdgcn_s_barrier();

  float* stg = reinterpret_cast<float*>(smem) + wave * 64 * EPI_LD;
  float cs_unused[8];
  u32x2 ew_unused[8];
#pragma unroll
  for (int qa = 0; qa < 2; ++qa) {
    u32x4 auxr[8];
    epilogue_prefetch<EPI>(p, m0 + wm * 128 + qa * 64, n0 + wn * 64, lane, auxr);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          stg[(i * 16 + 4 * (lane >> 4) + r) * EPI_LD + j * 16 + (lane & 15)] = acc[qa * 4 + i][j][r];
    VDS_WAIT_LGKM0();
    __builtin_amdgcn_wave_barrier();
    epilogue_64x64<EPI>(p, stg, m0 + wm * 128 + qa * 64, n0 + wn * 64, lane, cs_unused, ew_unused, auxr);
    __builtin_amdgcn_wave_barrier();  // the staging area is rewritten by the next quadrant row
  }
}

template <int LAYOUT, int EPI>
int launch(const GemmP& p, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<LAYOUT, EPI>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    attr_set = true;
  }
  dim3 grid(p.tiles_m * p.tiles_n, p.split_k > 1 ? p.split_k : 1, 1);
  vdsprof::Scope ps(LAYOUT == VDS_NT ? VDS_PROF_GEMM_NT : LAYOUT == VDS_NN ? VDS_PROF_GEMM_NN : VDS_PROF_GEMM_TN, s,
                    2.0 * p.M * p.N * p.K, 2.0 * ((double)p.M * p.K + (double)p.N * p.K + (double)p.M * p.N));
  hipLaunchKernelGGL((gemm_kernel<LAYOUT, EPI>), grid, dim3(256), LDS_BYTES, s, p);
  return hipGetLastError() == hipSuccess ? VDS_OK : VDS_ERR_LAUNCH;
}
}  // namespace mid

thread_local char g_err[256] = "";

template <int LAYOUT, int EPI>
int launch(const GemmP& p, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<LAYOUT, EPI>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    attr_set = true;
  }
  dim3 grid(p.tiles_m * p.tiles_n, p.split_k > 1 ? p.split_k : 1, 1);
  vdsprof::Scope ps(LAYOUT == VDS_NT ? VDS_PROF_GEMM_NT : LAYOUT == VDS_NN ? VDS_PROF_GEMM_NN : VDS_PROF_GEMM_TN, s,
                    2.0 * p.M * p.N * p.K, 2.0 * ((double)p.M * p.K + (double)p.N * p.K + (double)p.M * p.N));
  hipLaunchKernelGGL((gemm_kernel<LAYOUT, EPI>), grid, dim3(256), LDS_BYTES, s, p);
  return hipGetLastError() == hipSuccess ? VDS_OK : VDS_ERR_LAUNCH;
}


// ---- fixed-order split-K (deterministic mode, round 6) --------------------------------------------------------------------
// vds_set_deterministic(1, workspace, bytes): a split-K weight gradient writes one dense fp32 slab [M, N] per split into the
// caller's workspace with plain stores (GemmP::slab_stride; the epilogue offsets C by blockIdx.y slabs) and this kernel
// adds the slabs to C in split order: C = ((C + s0) + s1) + ..  -- the same words every run, where the fp32 atomics of the
// default mode commit in whatever order the workgroups finish.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* ws, int n_slabs, long slab_stride, float* C, long ldc,
                                                            int M, int N) {
  const long i4 = (long)blockIdx.x * 256 + threadIdx.x;  // one 16-byte piece of a row per thread
  const int n4 = N >> 2;
  if (i4 >= (long)M * n4) return;
  const long row = i4 / n4;
  const int col = (int)(i4 % n4) * 4;
  float* c = C + row * ldc + col;
  f32x4 v = *reinterpret_cast<const f32x4*>(c);
  const float* w = ws + row * N + col;
  for (int sidx = 0; sidx < n_slabs; ++sidx) v += *reinterpret_cast<const f32x4*>(w + (long)sidx * slab_stride);
  *reinterpret_cast<f32x4*>(c) = v;
}

// `tile` of vds_gemm_force_tile / VDS_GEMM_TILE: 0 = cost model | 128 | 256 | 2 (= 256 x 128)
inline int force_tile() { return vdscfg::geti(vdscfg::GEMM_TILE); }

// 256-wide tiling: the last tile column runs the 256 x 128 body when it holds at most 128 columns
inline int narrow_last_column(long N) {
  if (!vdscfg::geti(vdscfg::GEMM_NARROW)) return 0;
  const long rem = N % 256;
  return rem > 0 && rem <= 128;
}

// a split-K launch in deterministic mode: slabs in the workspace + the fixed-order reduction.  `launch_fn` launches the
// GEMM kernel on `p`, `bk` is that kernel's K-tile depth (how it cuts K into splits).  The split count is lowered until
// its slabs fit the workspace (1 = no slabs needed).
template <typename F>
int launch_split_det(GemmP& p, float* C, long ldc, hipStream_t s, int bk, F launch_fn) {
  const int kt_total = cdiv(p.K, bk);
  int split = p.split_k;
  const size_t slab = (size_t)p.M * p.N * 4;
  while (split > 1 && (size_t)split * slab > vdsdet::workspace_bytes()) --split;
  if (split <= 1) {  // one workgroup per output tile: a single (atomic) add per element and call is order-free
    p.split_k = 1;
    return launch_fn(p);
  }
  const int per = cdiv(kt_total, split);
  const int n_eff = cdiv(kt_total, per);  // splits that own K tiles (the others return at once and write nothing)
  float* ws = vdsdet::workspace((size_t)split * p.M * p.N, "vds_gemm (split-K)");
  if (!ws) return VDS_ERR_ARG;
  p.split_k = split;
  p.atomic = 0;
  p.joint_xcd = 0;  // (the slab index is blockIdx.y)
  p.C = ws;
  p.ldc = p.N;
  p.slab_stride = (long)p.M * p.N;
  const int rc = launch_fn(p);
  if (rc != VDS_OK) return rc;
  const long n4 = (long)p.M * (p.N >> 2);
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, (const float*)ws, n_eff,
                     p.slab_stride, C, ldc, p.M, p.N);
  return hipGetLastError() == hipSuccess ? VDS_OK : VDS_ERR_LAUNCH;
}

}  // namespace

extern "C" int vds_gemm_force_tile(int32_t tile) {
  const int prev = force_tile();
  if (tile != 0 && tile != 128 && tile != 256 && tile != 2) return VDS_ERR_ARG;
  vdscfg::g_val[vdscfg::GEMM_TILE] = tile;
  return prev;
}

extern "C" int vds_gemm_bf16(const vds_gemm_args* a, vds_stream_t stream) {
  if (!a || !a->A || !a->B || a->M <= 0 || a->N <= 0 || a->K <= 0) return VDS_ERR_ARG;
  if ((a->N & 7) || (a->lda & 7) || (a->ldb & 7)) return VDS_ERR_ARG;
  GemmP p;
  p.M = a->M; p.N = a->N; p.K = a->K;
  p.A = (const bf16_t*)a->A; p.lda = a->lda;
  p.B = (const bf16_t*)a->B; p.ldb = a->ldb;
  p.C = a->C; p.ldc = a->ldc; p.C2 = a->C2; p.ldc2 = a->ldc2;
  p.bias = (const bf16_t*)a->bias;
  p.aux = (const bf16_t*)a->aux; p.ldaux = a->ldaux;
  p.gate = a->gate; p.ldgate = a->ldgate;
  p.rows_per_batch = a->rows_per_batch > 0 ? a->rows_per_batch : a->M;
  p.row_base = 0;
  p.slab_stride = 0;
  // split_k: > 1 that many K splits (atomic accumulation into a pre-zeroed C); 1 none; <= -2: |split_k| splits and
  // atomic accumulation; 0 / -1 (TN + F32 only): the library picks tiling and split count itself (-1: and always
  // accumulates, so that several calls can sum into one C)
  const bool auto_split = (a->split_k == 0 || a->split_k == -1) && a->layout == VDS_TN && a->epilogue == VDS_EPI_F32;
  p.split_k = a->split_k > 1 ? a->split_k : (a->split_k < -1 ? -a->split_k : 1);
  p.atomic = (a->split_k > 1 || a->split_k < 0) ? 1 : 0;
  p.sa = p.sb = nullptr;
  p.prof_k = a->K;
  p.e_q = p.e_qt = nullptr; p.e_ldq = p.e_ldqt = 0;
  p.e_amax_in = nullptr; p.e_amax_out = p.e_dq_out = p.e_colsum = nullptr; p.e_fmt = 0;
  p.narrow = 0;
  if (a->colsum && !(a->epilogue == VDS_EPI_DGELU && a->C)) return VDS_ERR_ARG;
  p.tiles_m = cdiv(a->M, BM);
  p.tiles_n = cdiv(a->N, BN);
  size_t abytes, bbytes;
  switch (a->layout) {
    case VDS_NT:
      if (a->K & 7) return VDS_ERR_ARG;
      abytes = ((size_t)(a->M - 1) * a->lda + a->K) * 2;
      bbytes = ((size_t)(a->N - 1) * a->ldb + a->K) * 2;
      break;
    case VDS_NN:
      if (a->K & 7) return VDS_ERR_ARG;
      abytes = ((size_t)(a->M - 1) * a->lda + a->K) * 2;
      bbytes = ((size_t)(a->K - 1) * a->ldb + a->N) * 2;
      break;
    case VDS_TN:
      if (a->M & 7) return VDS_ERR_ARG;
      abytes = ((size_t)(a->K - 1) * a->lda + a->M) * 2;
      bbytes = ((size_t)(a->K - 1) * a->ldb + a->N) * 2;
      break;
    default:
      return VDS_ERR_ARG;
  }
  if (abytes >= (1ull << 32) || bbytes >= (1ull << 32)) return VDS_ERR_UNSUPPORTED;
  p.a_bytes = (unsigned)abytes;
  p.b_bytes = (unsigned)bbytes;
  hipStream_t s = (hipStream_t)stream;
  if (p.atomic && !(a->layout == VDS_TN && a->epilogue == VDS_EPI_F32)) return VDS_ERR_ARG;
  if (!a->C && a->epilogue != VDS_EPI_GATE_RES) return VDS_ERR_ARG;
  const bool det = vdscfg::geti(vdscfg::DETERMINISTIC) != 0;
  // deterministic mode: the fc1 bias gradient is a separate fixed-order column sum, not the epilogue's per-tile atomics
  const float* fused_colsum = det ? nullptr : a->colsum;
  // tile choice: 256^2 (one workgroup per CU, deep LDS-DMA pipeline) for problems that fill the chip
  // with such tiles; 128^2 otherwise.  vds_gemm_force_tile / VDS_GEMM_TILE = 128 | 256 | 2 forces one (experiments).
  const int ft = force_tile();
  // model: a 256^2 workgroup (alone on its CU) sustains ~1.24x the rate of two co-resident 128^2
  // workgroups; compare the number of rounds each tiling needs (wave quantisation dominates at
  // these sizes).  Weight gradients (TN, split-K) run on the 256 x 128 or the 128^2 kernel.
  const int tm = cdiv(a->M, 256), tn = cdiv(a->N, 256);
  const long rounds_big = ((long)tm * tn + 255) / 256;
  const long rounds_small = ((long)p.tiles_m * p.tiles_n + 511) / 512;
  bool use_big = a->layout != VDS_TN && a->K >= 256 && (double)rounds_big * (2.0 / 1.24) < (double)rounds_small;
  if (ft == 128) use_big = false;
  if (ft == 256 && a->layout != VDS_TN) use_big = true;
  p.group_m = vdscfg::geti(vdscfg::GEMM_GROUP_M);  // groups of 4 row tiles (1024 rows) measured best at the DiT shapes
  if (p.group_m < 1) p.group_m = 4;
  p.joint_xcd = 0;
  const int group_m_tn = vdscfg::geti(vdscfg::GEMM_GROUP_M_TN) > 0 ? vdscfg::geti(vdscfg::GEMM_GROUP_M_TN) : 0;
  if (a->layout == VDS_TN) {
    // weight gradients: few output tiles, every tile of a split streams the same token range.  Fixed groups of 2 row
    // tiles measured 4-5 % faster than 4 at the DiT-XL shapes (106.4 -> 101.5 ms per step); 0 (default) = sized at the
    // launch so that one XCD's chunk of the joint (split, tile) list is one group (below): 97 ms
    p.group_m = group_m_tn ? group_m_tn : 2;
    p.joint_xcd = vdscfg::geti(vdscfg::GEMM_TN_JOINT);
  }
  // the split-K launch of a weight gradient: atomics (default) or slabs + fixed-order reduction (deterministic mode)
  auto launch_tn = [&](int bk, auto fn) -> int {
    if (det && p.split_k > 1) return launch_split_det(p, (float*)a->C, a->ldc, s, bk, fn);
    return fn(p);
  };
  if (auto_split) {
    // weight gradient dW[M = out features, N = in features] = dy^T x over K = tokens: few output tiles, long K.
    // Candidates: 128^2 tiles (two workgroups per CU) and 256 x 128 tiles (two per CU, ~mid_tn x the per-CU rate:
    // half the transposing LDS reads per MFMA).  Cost of (tiling, s splits) in units of one 128^2 tile x 64-token
    // step: rounds of 512 co-resident workgroups x K steps per split x tile size, plus the fp32-atomic traffic of
    // the partial tiles (64 KB per 128^2 tile at the chip-wide ~1.3 TB/s, ~0.034 units each).
    // (Round 5 measured the 256^2 tiling as a third candidate: 1395 TFLOP/s at 8192^3, -4..-5 % on the fc1 / fc2 weight
    // gradients in a loop of launches, nothing inside the step -- removed in round 6, profiles/r05/gemm_tn_on_256_tiling.log.)
    const double mid_tn = vdscfg::get(vdscfg::GEMM_MID_TN);
    const long kt = cdiv(a->K, 64);
    double best = 1e300;
    int best_tile = 128, best_s = 1;
    for (int cand = 0; cand < 2; ++cand) {
      if ((cand == 0 && ft == 2) || (cand == 1 && ft == 128)) continue;
      const long tiles = cand == 0 ? (long)cdiv(a->M, 128) * cdiv(a->N, 128) : (long)cdiv(a->M, 256) * cdiv(a->N, 128);
      const double unit = cand == 0 ? 1.0 : 2.0 * mid_tn, atom = cand == 0 ? 0.034 : 0.068;
      for (int sp = 1; sp <= 32; ++sp) {
        if (sp > 1 && kt / sp < 8) break;
        const long rounds = (tiles * sp + 511) / 512;
        const double cost = (double)rounds * (double)((kt + sp - 1) / sp) * unit + (sp > 1 ? tiles * sp * atom : 0.0);
        if (cost < best - 1e-9) { best = cost; best_tile = cand == 0 ? 128 : 2; best_s = sp; }
      }
    }
    p.split_k = best_s;
    p.atomic = (best_s > 1 || a->split_k == -1) ? 1 : 0;
    if (best_tile == 2) {
      p.tiles_m = cdiv(a->M, 256);
      p.tiles_n = cdiv(a->N, 128);
      if (p.joint_xcd && !group_m_tn) {
        // one XCD's chunk = (tiles x splits) / 8 consecutive entries of the split-major list: make it a block of
        // whole tile rows of one split (qkv weight gradient: 14 x 9 tiles x 4 splits -> 7 rows x 9 columns per XCD)
        const long chunk = ((long)p.tiles_m * p.tiles_n * p.split_k + 7) / 8;
        long g = (chunk + p.tiles_n / 2) / p.tiles_n;
        p.group_m = (int)(g < 1 ? 1 : (g > p.tiles_m ? p.tiles_m : g));
      }
      return launch_tn(mid::BK, [&](GemmP& q) { return mid::launch<VDS_TN, VDS_EPI_F32>(q, s); });
    }
    return launch_tn(BK, [&](GemmP& q) { return launch<VDS_TN, VDS_EPI_F32>(q, s); });
  }
  // 256 x 128 tiles, two workgroups per CU: no padded half tile when N is a multiple of 128 but not of 256, and the
  // epilogue of one workgroup runs under the main loop of the other.  Measured (tools/bench_gemm_tiles.py, DiT-XL
  // shapes, B = 12): its LDS-DMA issue rate (1.5x the bytes per FLOP of a 256^2 tile, from half as many waves) caps
  // it at ~0.8x the 256^2 kernel on NT / NN problems, so a round of 512 such tiles is priced at 1.3 rounds of 256^2
  // tiles and it only wins where tile quantisation is severe; on weight gradients (TN) it beats the 128^2 kernel
  // by 7-24 % (see auto_split above).  tile = 2 forces it.
  {
    const int tmm = cdiv(a->M, 256), tnm = cdiv(a->N, 128);
    const double cost_mid = (double)(((long)tmm * tnm + 511) / 512) * vdscfg::get(vdscfg::GEMM_MID_FACTOR);
    const double cost_big = (double)rounds_big, cost_small = (double)rounds_small * (1.24 / 2.0);
    bool use_mid = a->K >= 128 && cost_mid < (use_big ? cost_big : cost_small) && a->layout != VDS_TN;
    if (ft == 2) use_mid = true;
    if (a->colsum && a->layout != VDS_NN) use_mid = false;
    if (ft == 128 || ft == 256) use_mid = false;
    if (use_mid) {
      p.tiles_m = tmm;
      p.tiles_n = tnm;
      if (a->colsum) {  // no fused column sums in this tiling: a pass over the result follows the GEMM
        const int rc = mid::launch<VDS_NN, VDS_EPI_DGELU>(p, s);
        return rc != VDS_OK ? rc : vds_colsum_bf16(a->C, a->ldc, a->colsum, a->M, a->N, stream);
      }
#define GOM(L, E) if (a->layout == L && a->epilogue == E) return mid::launch<L, E>(p, s);
      GOM(VDS_NT, VDS_EPI_STORE)
      GOM(VDS_NT, VDS_EPI_BIAS_GELU)
      GOM(VDS_NT, VDS_EPI_GATE_RES)
      GOM(VDS_NN, VDS_EPI_STORE)
      GOM(VDS_NN, VDS_EPI_DGELU)
#undef GOM
      if (a->layout == VDS_TN && a->epilogue == VDS_EPI_F32)
        return launch_tn(mid::BK, [&](GemmP& q) { return mid::launch<VDS_TN, VDS_EPI_F32>(q, s); });
    }
  }
  if (use_big) {
    p.tiles_m = tm;
    p.tiles_n = tn;
    p.narrow = narrow_last_column(a->N);
    p.e_colsum = const_cast<float*>(fused_colsum);  // DGELU only (checked above): column sums of the result in the epilogue
    int rc = VDS_ERR_UNSUPPORTED;
#define GOB(L, E) if (a->layout == L && a->epilogue == E) rc = big::launch<L, E>(p, s);
    GOB(VDS_NT, VDS_EPI_STORE)
    GOB(VDS_NT, VDS_EPI_BIAS_GELU)
    GOB(VDS_NT, VDS_EPI_GATE_RES)
    GOB(VDS_NN, VDS_EPI_STORE)
    GOB(VDS_NN, VDS_EPI_DGELU)
#undef GOB
    if (rc == VDS_OK && a->colsum && !fused_colsum) return vds_colsum_bf16(a->C, a->ldc, a->colsum, a->M, a->N, stream);
    return rc;
  }
  p.tiles_m = cdiv(a->M, BM);
  p.tiles_n = cdiv(a->N, BN);
  if (a->colsum) {  // the smaller tilings have no fused column sums: a pass over the result follows the GEMM
    const int rc = launch<VDS_NN, VDS_EPI_DGELU>(p, s);
    return rc != VDS_OK ? rc : vds_colsum_bf16(a->C, a->ldc, a->colsum, a->M, a->N, stream);
  }
#define GO(L, E) if (a->layout == L && a->epilogue == E) return launch<L, E>(p, s);
  GO(VDS_NT, VDS_EPI_STORE)
  GO(VDS_NT, VDS_EPI_BIAS_GELU)
  GO(VDS_NT, VDS_EPI_GATE_RES)
  GO(VDS_NN, VDS_EPI_STORE)
  GO(VDS_NN, VDS_EPI_DGELU)
#undef GO
  if (a->layout == VDS_TN && a->epilogue == VDS_EPI_F32)
    return launch_tn(BK, [&](GemmP& q) { return launch<VDS_TN, VDS_EPI_F32>(q, s); });
  return VDS_ERR_UNSUPPORTED;
}

// OCP fp8 GEMM, NT only: C[M,N] = (sum_k A[m,k] B[n,k]) * scale_a * scale_b with A, B one byte per element
// (a_fmt / b_fmt: 0 = e4m3fn, 1 = e5m2; B must be e4m3), always on the 256^2 kernel.
extern "C" int vds_gemm_fp8(const vds_gemm_args* a, const float* scale_a, const float* scale_b, int32_t a_fmt,
                            int32_t b_fmt, const vds_fp8_out* emit, vds_stream_t stream) {
  if (!a || !a->A || !a->B || a->M <= 0 || a->N <= 0 || a->K <= 0) return VDS_ERR_ARG;
  if ((a->layout != VDS_NT && a->layout != VDS_TN) || b_fmt != 0 || (a_fmt != 0 && a_fmt != 1)) return VDS_ERR_UNSUPPORTED;
  if ((a->N & 7) || (a->K & 15) || (a->lda & 15) || (a->ldb & 15)) return VDS_ERR_ARG;
  const bool tn = a->layout == VDS_TN;  // C[M,N] = sum_k A[k,m] B[k,n]: both operands [K, .] row-major (token-major)
  if (tn && (a->epilogue != VDS_EPI_F32 || a_fmt != 1 || emit || (a->M & 15))) return VDS_ERR_UNSUPPORTED;
  GemmP p;
  p.M = a->M; p.N = a->N; p.K = a->K / 2;  // the kernel addresses in 2-byte units
  p.A = (const bf16_t*)a->A; p.lda = a->lda / 2;
  p.B = (const bf16_t*)a->B; p.ldb = a->ldb / 2;
  p.C = a->C; p.ldc = a->ldc; p.C2 = a->C2; p.ldc2 = a->ldc2;
  p.bias = (const bf16_t*)a->bias;
  p.aux = (const bf16_t*)a->aux; p.ldaux = a->ldaux;
  p.gate = a->gate; p.ldgate = a->ldgate;
  p.rows_per_batch = a->rows_per_batch > 0 ? a->rows_per_batch : a->M;
  p.row_base = 0;
  p.slab_stride = 0;
  p.split_k = a->split_k > 1 ? a->split_k : (a->split_k < -1 ? -a->split_k : 1);
  p.atomic = (a->split_k > 1 || a->split_k < 0) ? 1 : 0;
  p.sa = scale_a; p.sb = scale_b;
  p.prof_k = a->K;
  p.tiles_m = cdiv(a->M, 256);
  p.tiles_n = cdiv(a->N, 256);
  const size_t abytes = tn ? (size_t)(a->K - 1) * a->lda + a->M : (size_t)(a->M - 1) * a->lda + a->K;
  const size_t bbytes = tn ? (size_t)(a->K - 1) * a->ldb + a->N : (size_t)(a->N - 1) * a->ldb + a->K;
  if (abytes >= (1ull << 32) || bbytes >= (1ull << 32)) return VDS_ERR_UNSUPPORTED;
  p.a_bytes = (unsigned)abytes;
  p.b_bytes = (unsigned)bbytes;
  p.group_m = 4;
  p.joint_xcd = 0;
  p.narrow = narrow_last_column(a->N);
  p.e_q = p.e_qt = nullptr; p.e_ldq = p.e_ldqt = 0;
  p.e_amax_in = nullptr; p.e_amax_out = p.e_dq_out = p.e_colsum = nullptr; p.e_fmt = 0;
  if (emit) {
    if (a->epilogue != VDS_EPI_BIAS_GELU && a->epilogue != VDS_EPI_DGELU) return VDS_ERR_ARG;
    if ((emit->q && (emit->ldq & 7)) || (emit->qt && (emit->ldqt & 3)) || (emit->fmt != 0 && emit->fmt != 1)) return VDS_ERR_ARG;
    if ((emit->q || emit->qt) && !emit->amax_in) return VDS_ERR_ARG;
    p.e_q = (unsigned char*)emit->q; p.e_ldq = emit->ldq;
    p.e_qt = (unsigned char*)emit->qt; p.e_ldqt = emit->ldqt;
    p.e_amax_in = emit->amax_in; p.e_amax_out = emit->amax_out; p.e_dq_out = emit->dq_out;
    p.e_fmt = emit->fmt; p.e_colsum = emit->colsum;
  }
  if (p.atomic && a->epilogue != VDS_EPI_F32) return VDS_ERR_ARG;
  if (!a->C && a->epilogue != VDS_EPI_GATE_RES && !(emit && a->epilogue == VDS_EPI_DGELU)) return VDS_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (tn) {
    if (vdscfg::geti(vdscfg::DETERMINISTIC) && p.split_k > 1)
      return launch_split_det(p, (float*)a->C, a->ldc, s, BK, [&](GemmP& q) { return big::launch<VDS_TN, VDS_EPI_F32, 3>(q, s); });
    return big::launch<VDS_TN, VDS_EPI_F32, 3>(p, s);
  }
  if (vdscfg::geti(vdscfg::DETERMINISTIC) && p.e_colsum) return VDS_ERR_UNSUPPORTED;  // (fused column sums are atomic)
#define GOF(E, F) if (a->epilogue == E && a_fmt == F - 1) return big::launch<VDS_NT, E, F>(p, s);
  GOF(VDS_EPI_STORE, 1)
  GOF(VDS_EPI_BIAS_GELU, 1)
  GOF(VDS_EPI_GATE_RES, 1)
  GOF(VDS_EPI_STORE, 2)
  GOF(VDS_EPI_DGELU, 2)
  GOF(VDS_EPI_F32, 2)
  GOF(VDS_EPI_F32, 1)
#undef GOF
  return VDS_ERR_UNSUPPORTED;
}
