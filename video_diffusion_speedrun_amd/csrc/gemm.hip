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
#include "../../include/vds.h"

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
  int split_k;
  int atomic;
  unsigned a_bytes, b_bytes;
  int tiles_m, tiles_n;
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
__device__ __forceinline__ void stage_tile(__amdgpu_buffer_rsrc_t rsrc, char* lds_tile, const unsigned voff[4],
                                           unsigned koff, int wave, int lane, int krem) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    char* dst = lds_tile + (wave * 4 + j) * 1024;
    unsigned off = voff[j] + koff;
    if constexpr (!KMAJOR) {
      if (krem < BK) {
        const int row = (wave * 4 + j) * 8 + (lane >> 3);
        if (swz_kc(row, lane & 7) * 8 >= krem) off = 0xfffffff0u;
      }
    }
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, LDS_PTR(dst), 16, off, 0, 0, 0);
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

  const __amdgpu_buffer_rsrc_t ra = make_rsrc(p.A, p.a_bytes);
  const __amdgpu_buffer_rsrc_t rb = make_rsrc(p.B, p.b_bytes);
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
  __syncthreads();  // vmcnt(0) + barrier: tile kt_begin landed

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
    __syncthreads();  // next tile landed (vmcnt(0)) and everyone is done reading `cur`
    cur ^= 1;
  }

  // ---- epilogue: accumulators -> LDS (fp32) -> 16-byte row segments ----------------------
  float* stg = reinterpret_cast<float*>(smem) + wave * 64 * EPI_LD;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        stg[(i * 16 + 4 * (lane >> 4) + r) * EPI_LD + j * 16 + (lane & 15)] = acc[i][j][r];
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): own wave's region only, no barrier needed
  __builtin_amdgcn_wave_barrier();

  if constexpr (EPI == VDS_EPI_F32) {
    if (p.atomic) {
      // split-K / accumulate: one atomic wave-instruction = 64 consecutive floats of one row (256
      // contiguous bytes, the full-rate shape of global_atomic_add_f32 on gfx950)
      const int acol = n0 + wn * 64 + lane;
      if (acol < p.N) {
        float* cbase = reinterpret_cast<float*>(p.C) + (long)(m0 + wm * 64) * p.ldc + acol;
        const int rmax = min(64, p.M - (m0 + wm * 64));
        for (int row = 0; row < rmax; ++row) atomicAdd(cbase + (long)row * p.ldc, stg[row * EPI_LD + lane]);
      }
      return;
    }
  }
  const int c8 = lane & 7, rin = lane >> 3;
  const int gcol = n0 + wn * 64 + c8 * 8;
  if (gcol >= p.N) return;
  float bias8[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) bias8[e] = 0.f;
  if constexpr (EPI != VDS_EPI_F32 && EPI != VDS_EPI_DGELU) {
    if (p.bias) {
      u32x4 bv = *reinterpret_cast<const u32x4*>(p.bias + gcol);
#pragma unroll
      for (int e = 0; e < 4; ++e) { bias8[2 * e] = bflo(bv[e]); bias8[2 * e + 1] = bfhi(bv[e]); }
    }
  }
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int row = it * 8 + rin;
    const long grow = m0 + wm * 64 + row;
    if (grow >= p.M) continue;
    const f32x4 lo = *reinterpret_cast<const f32x4*>(stg + row * EPI_LD + c8 * 8);
    const f32x4 hi = *reinterpret_cast<const f32x4*>(stg + row * EPI_LD + c8 * 8 + 4);
    float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    if constexpr (EPI == VDS_EPI_F32) {
      float* c = reinterpret_cast<float*>(p.C) + grow * p.ldc + gcol;
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
        o2[e] = pack_bf2(gelu_f(a), gelu_f(b));
      }
      *reinterpret_cast<u32x4*>(reinterpret_cast<bf16_t*>(p.C) + grow * p.ldc + gcol) = o;
      *reinterpret_cast<u32x4*>(reinterpret_cast<bf16_t*>(p.C2) + grow * p.ldc2 + gcol) = o2;
    } else if constexpr (EPI == VDS_EPI_GATE_RES) {
      const int b = (int)(grow / p.rows_per_batch);
      const float* gp = p.gate + (long)b * p.ldgate + gcol;
      const f32x4 g0 = *reinterpret_cast<const f32x4*>(gp);
      const f32x4 g1 = *reinterpret_cast<const f32x4*>(gp + 4);
      const float g[8] = {g0[0], g0[1], g0[2], g0[3], g1[0], g1[1], g1[2], g1[3]};
      const u32x4 xr = *reinterpret_cast<const u32x4*>(p.aux + grow * p.ldaux + gcol);
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
      const u32x4 pr = *reinterpret_cast<const u32x4*>(p.aux + grow * p.ldaux + gcol);
      u32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        o[e] = pack_bf2(v[2 * e] * dgelu_f(bflo(pr[e])), v[2 * e + 1] * dgelu_f(bfhi(pr[e])));
      *reinterpret_cast<u32x4*>(reinterpret_cast<bf16_t*>(p.C) + grow * p.ldc + gcol) = o;
    }
  }
}

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

}  // namespace

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
  // split_k < 0: |split_k| splits and atomic accumulation into C even for a single split
  p.split_k = a->split_k > 1 ? a->split_k : (a->split_k < -1 ? -a->split_k : 1);
  p.atomic = (a->split_k > 1 || a->split_k < 0) ? 1 : 0;
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
#define GO(L, E) if (a->layout == L && a->epilogue == E) return launch<L, E>(p, s);
  GO(VDS_NT, VDS_EPI_STORE)
  GO(VDS_NT, VDS_EPI_BIAS_GELU)
  GO(VDS_NT, VDS_EPI_GATE_RES)
  GO(VDS_NN, VDS_EPI_STORE)
  GO(VDS_NN, VDS_EPI_DGELU)
  GO(VDS_TN, VDS_EPI_F32)
#undef GO
  return VDS_ERR_UNSUPPORTED;
}
