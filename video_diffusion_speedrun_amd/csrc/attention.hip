// Flash-style full (non-causal) attention for gfx950, forward and backward, bf16 in /
// fp32 softmax, head_dim 32 / 64 / 72 / 96 / 128 (72 = DiT-XL, padded to 80 for the QK contraction and
// to 96 for the 32-wide output blocks).  Replaces F.scaled_dot_product_attention at
// model.py:136 (self, L x L) and model.py:157 (cross, L x 512) and its autograd backward.
//
// All three kernels are built from two MFMA product forms on v_mfma_f32_32x32x16_bf16:
//   F1  X[r, c]   = sum_k A[r,k] B[c,k]      both operands row-major, 16-B fragment reads
//   F2  Y^T[d, c] = sum_r T[r,d] X[r,c]      X = a previous accumulator used in place as the
//                                            B operand (its row index is the contraction),
//                                            T^T read from a row-major LDS tile with the
//                                            transposing read ds_read_b64_tr_b16
// so the softmax statistics always live on the lane that owns the column c and no tile ever
// crosses lanes through LDS:
//   forward   S^T = K Q^T (F1)            O^T  += V^T P^T  (F2)        c = query, lane-local m, l
//   dQ        S^T = K Q^T, dP^T = V dO^T  dQ^T += K^T dS^T (F2)        c = query
//   dK,dV     S = Q K^T,  dP = dO V^T     dV^T += dO^T P, dK^T += Q^T dS (F2)   c = key
// Backward is two kernels (7 products instead of the fused 5) in exchange for no atomics, no
// dS transpose and bitwise-reproducible gradients.
//
// LDS tiles use the 8-row x 32-column sub-tiled, XOR-swizzled image that is conflict-free
// for both the 16-B row reads and the transposing reads (cdna guide T10 image (a)).
// K/V (or Q/dO) tiles go HBM -> LDS by LDS-DMA (buffer_load ... lds issued from inline asm, the
// image swizzle applied to the per-lane source address), one tile ahead, double-buffered, one
// barrier per tile; the forward pass uses a lazy-rescale online softmax per 32-key sub-block.
#include "common.h"
#include <type_traits>
#include "prof.h"
#include "config.h"
#include "../../include/vds.h"
#include <cstdlib>

namespace {

constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;

struct AttnP {
  int B, H, Lq, Lk, hd;
  const bf16_t* q; long q_sb, q_sh, q_sl;
  const bf16_t* k; long k_sb, k_sh, k_sl;
  const bf16_t* v; long v_sb, v_sh, v_sl;
  bf16_t* o; long o_sb, o_sh, o_sl;
  float* lse;
  const bf16_t* d_o; long do_sb, do_sh, do_sl;
  bf16_t* dq; long dq_sb, dq_sh, dq_sl;
  bf16_t* dk; long dk_sb, dk_sh, dk_sl;
  bf16_t* dv; long dv_sb, dv_sh, dv_sl;
  float* delta;
  float scale;
  int n_rt;  // row tiles per (b,h) of the stationary operand
  int tail_last;  // decode_block: the partly filled last tile of every head is scheduled after all full tiles
  int kv_pad_ones;
  // dK/dV kernel, short key sequences (cross-attention: 512 context keys = 4 key tiles per head): the query range is
  // split over q_split workgroups per key tile, each writing an fp32 partial [q_split][B*H][Lk][2][hd] that
  // dkv_reduce_kernel sums into the bf16 outputs (run_bwd picks q_split; 1 = the kernel stores bf16 itself)
  int q_split;
  float* dkv_part;
  // 16-byte row stores in the 16x16x32 kernels' epilogues (store_block_bf16_t): rows 16-byte aligned
  int wide_o, wide_dq, wide_dkv;
};

// ---- LDS image (a): rows x HDP bf16, 8x32 sub-tiles of 512 B -------------------------------
template <int HDP>
__device__ __forceinline__ int img_off(int row, int ch) {
  return (row >> 3) * ((HDP / 32) * 512) + (ch >> 2) * 512 + (row & 7) * 64 + ((((ch & 3) ^ ((row >> 2) & 3))) << 4);
}
// F1 operand (A or B): 32 rows x 16 k; lane -> row (l&31), k = 16*ks + 8*(l>>5) .. +7
template <int HDP>
__device__ __forceinline__ bf16x8 frag_row(const char* tile, int row0, int ks, int lane) {
  return *reinterpret_cast<const bf16x8*>(tile + img_off<HDP>(row0 + (lane & 31), ks * 2 + (lane >> 5)));
}
// F2 A operand = T^T: 32 cols (d0..d0+31) x 16 rows (r0..r0+15) of the row-major tile T,
// k order matching an accumulator used as B: element j of lane-half h is row r0 + 8(j>>2) + 4h + (j&3).
template <int HDP>
__device__ __forceinline__ bf16x8 frag_tr(const char* tile, int r0, int d0, int lane) {
  const int g = lane >> 4, h = g >> 1, i = lane & 15, qq = i >> 2, pp = i & 3;
  const int row = r0 + 4 * h + qq;
  const int ch = (d0 >> 3) + 2 * (g & 1) + (pp >> 1);
  const char* p0 = tile + img_off<HDP>(row, ch) + 8 * (pp & 1);
  const char* p1 = tile + img_off<HDP>(row + 8, ch) + 8 * (pp & 1);
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)LDS_PTR(p0));
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)LDS_PTR(p1));
  s16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, r);
}
// accumulator registers 8s..8s+7 -> bf16 operand fragment of k-step s
__device__ __forceinline__ bf16x8 acc_frag(const f32x16& x, int s) {
  bf16x8 r;
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = (__bf16)x[8 * s + j];
  return r;
}
__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
#ifndef VDS_DELTA_NT
#define VDS_DELTA_NT 1  // delta preprocess: non-temporal loads of O / dO (6.7 -> 6.3 ms per step; 0 = plain loads)
#endif
#ifdef VDS_ATTN_PRIO
#define PRIO_HI() __builtin_amdgcn_s_setprio(1)
#define PRIO_LO() __builtin_amdgcn_s_setprio(0)
#else
#define PRIO_HI()
#define PRIO_LO()
#endif
__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.f;
  return z;
}
// row index inside a 32x32 accumulator: reg -> (reg&3) + 8*(reg>>2) + 4*h
__device__ __forceinline__ int acc_row(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

__device__ __forceinline__ __amdgpu_buffer_rsrc_t slice_rsrc(const bf16_t* base, long sl, int L, int hd) {
  return make_rsrc(base, (unsigned)((((long)(L - 1)) * sl + hd) * 2));
}
__device__ __forceinline__ srd_t slice_srd(const bf16_t* base, long sl, int L, int hd) {
  return make_srd(base, (unsigned)((((long)(L - 1)) * sl + hd) * 2));
}

// Make the compiler retire a prologue load HERE (a use it can see): otherwise its s_waitcnt vmcnt(0)
// for the first use sits inside the main loop, where it also drains the LDS-DMA prefetch of the
// next tile (vmcnt counts every outstanding vector-memory operation, in order).
__device__ __forceinline__ void retire(const bf16x8& f) { asm volatile("" ::"v"(f)); }
__device__ __forceinline__ void retire(float f) { asm volatile("" ::"v"(f)); }

// XCD-aware (b,h,row-tile) decode: all row tiles of a head run on one XCD so that the
// streamed operand (K/V or Q/dO of that head) stays in that XCD's L2.
// Workgroup -> (head, row tile).  blockIdx & 7 is the XCD; each XCD walks its heads one after the other, so that the
// workgroups of a head -- which all stream the same K / V (or Q / dO) rows -- run together and share them in that XCD's
// L2.  tail_last: the sequence length leaves a partly filled last tile (8192 + 16 register tokens = 64 tiles of 128 +
// 1), whose workgroup costs about half a full one (only one of its waves has rows).  Those tail workgroups of all the
// XCD's heads are scheduled after the full ones, so that the partly filled LAST ROUND of the launch is made of cheap
// workgroups: nothing at B = 12 (24 rounds), but at B = 2 the forward launch is 2 rounds + 32 tails instead of 3 rounds.
__device__ __forceinline__ bool decode_block(int n_rt, int tail_last, int BH, int& bh, int& rt) {
  const int pid = blockIdx.x, xcd = pid & 7, idx = pid >> 3;
  if (tail_last && n_rt > 1) {
    const int n_full = n_rt - 1, cut = ((BH + 7) >> 3) * n_full;
    if (idx < cut) {
      bh = (idx / n_full) * 8 + xcd;
      rt = idx % n_full;
    } else {
      bh = (idx - cut) * 8 + xcd;
      rt = n_full;
    }
  } else {
    bh = (idx / n_rt) * 8 + xcd;
    rt = idx % n_rt;
  }
  return bh < BH;
}

// store a transposed accumulator set Y^T[d, c] (c on lanes) as rows Y[c, d] of bf16
template <int NDB>
__device__ __forceinline__ void store_rows(bf16_t* rowp, const f32x16 (&acc)[NDB], float mul, int hd, int h) {
#pragma unroll
  for (int db = 0; db < NDB; ++db)
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      const int d = db * 32 + 8 * rg + 4 * h;
      if (d < hd) {
        u32x2 w;
        w[0] = pack_bf2(acc[db][4 * rg] * mul, acc[db][4 * rg + 1] * mul);
        w[1] = pack_bf2(acc[db][4 * rg + 2] * mul, acc[db][4 * rg + 3] * mul);
        *reinterpret_cast<u32x2*>(rowp + d) = w;
      }
    }
}

// the same accumulator set as fp32 rows (partial sums of the query-split dK/dV kernel)
template <int NDB>
__device__ __forceinline__ void store_rows_f32(float* rowp, const f32x16 (&acc)[NDB], int hd, int h) {
#pragma unroll
  for (int db = 0; db < NDB; ++db)
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      const int d = db * 32 + 8 * rg + 4 * h;
      if (d < hd)
        *reinterpret_cast<f32x4*>(rowp + d) = f32x4{acc[db][4 * rg], acc[db][4 * rg + 1], acc[db][4 * rg + 2], acc[db][4 * rg + 3]};
    }
}
// dk / dv (bf16, strided) = sum over the q_split partials; one thread = 4 consecutive head-dim columns of one
// (b, h, key, dk | dv) row; dk is scaled by the softmax scale here (the bf16 path does it in store_rows)
__global__ __launch_bounds__(256) void dkv_reduce_kernel(AttnP p) {
  const int hq = p.hd >> 2;
  const long n = (long)p.B * p.H * p.Lk * 2 * hq;
  const long gid = (long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= n) return;
  const int c = (int)(gid % hq);
  const int which = (int)((gid / hq) & 1);
  const long row = gid / (2 * hq);  // (b * H + h) * Lk + key
  const int key = (int)(row % p.Lk);
  const int bh = (int)(row / p.Lk), b = bh / p.H, hh = bh % p.H;
  const long stride = (long)p.B * p.H * p.Lk * 2 * p.hd;
  const float* src = p.dkv_part + (row * 2 + which) * p.hd + 4 * c;
  f32x4 a = *reinterpret_cast<const f32x4*>(src);
  for (int sp = 1; sp < p.q_split; ++sp) a += *reinterpret_cast<const f32x4*>(src + sp * stride);
  const float mul = which == 0 ? p.scale : 1.0f;
  bf16_t* dst = which == 0 ? p.dk + b * p.dk_sb + hh * p.dk_sh + (long)key * p.dk_sl
                           : p.dv + b * p.dv_sb + hh * p.dv_sh + (long)key * p.dv_sl;
  const u32x2 w = {pack_bf2(a[0] * mul, a[1] * mul), pack_bf2(a[2] * mul, a[3] * mul)};
  *reinterpret_cast<u32x2*>(dst + 4 * c) = w;
}

// ===================================== forward ==============================================
// ---- lazy-rescale online softmax per 32-key sub-block ---------------------------------------
// The running maximum is only raised (and O, l rescaled) when some row's sub-block maximum exceeds
// it by more than 2^LAZY_THR (wave-uniform, rare after the first tile), so each 32-key sub-block is
// an independent chain  S (MFMA) -> exp2 (VALU) -> P V (MFMA): the S products of sub-block kb+1
// can issue under the VALU work of sub-block kb inside ONE wave, on top of the overlap between
// the waves that share a SIMD.  P <= 2^LAZY_THR keeps bf16's relative precision; sums are fp32.
constexpr float LAZY_THR = 8.0f;
__device__ __forceinline__ float max_with_other_half(float x) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float add_with_other_half(float x) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// ---- explicit software pipeline of the LDS fragment reads ------------------------------------
// All operand fragments of a 32-key sub-block are read from LDS one phase before the MFMAs that
// consume them (K fragments of both sub-blocks at the top of the tile, the V^T fragments of a
// sub-block while the S products / the other sub-block's softmax run), so no MFMA waits on an
// LDS round trip, and each sub-block's VALU softmax runs under the other sub-block's MFMAs.
template <int HDP, int KSQ>
__device__ __forceinline__ void load_kfrags(bf16x8 (&kf)[KSQ], const char* tile, int row0, int lane) {
#pragma unroll
  for (int ks = 0; ks < KSQ; ++ks) kf[ks] = frag_row<HDP>(tile, row0, ks, lane);
}
template <int HDP, int NDB>
__device__ __forceinline__ void load_vfrags(bf16x8 (&vf)[2][NDB], const char* tile, int row0, int lane) {
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
    for (int db = 0; db < NDB; ++db) vf[s2][db] = frag_tr<HDP>(tile, row0 + 16 * s2, db * 32, lane);
}

// one 32-key sub-block of the lazy online softmax: s (scores, fp32) -> P fragments (bf16)
template <int NDB>
__device__ __forceinline__ void lazy_softmax(f32x16& s, float c, float& m, float& l, f32x16 (&o)[NDB],
                                             bf16x8 (&pf)[2]) {
  float mx = fmaxf(s[0], s[1]);
#pragma unroll
  for (int r = 2; r < 16; r += 2) mx = fmaxf(mx, fmaxf(s[r], s[r + 1]));
  mx = max_with_other_half(mx) * c;
  if (__builtin_amdgcn_ballot_w64(mx > m + LAZY_THR) != 0) {  // wave-uniform, rare
    asm volatile("; rescale" ::: "memory");  // keeps this a real branch (no if-conversion)
    const float m_new = fmaxf(m, mx);
    const float alpha = __builtin_amdgcn_exp2f(m - m_new);
    m = m_new;
    l *= alpha;
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
  }
  float ls = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float e = __builtin_amdgcn_exp2f(s[r] * c - m);
    s[r] = e;
    ls += e;
  }
  l += ls;
  pf[0] = acc_frag(s, 0);
  pf[1] = acc_frag(s, 1);
}

// ---- LDS-DMA staging of a ROWS x HDP tile in image (a) ---------------------------------------
// buffer_load ... lds writes 64 lanes x 16 B = 1 KiB contiguously, so the image's swizzle is applied
// to the per-lane SOURCE address (guide rule 21): lane i of piece q fills LDS bytes q*1024 + 16 i,
// i.e. chunk `ch` of row `row` with img_off(row, ch) == that offset.  No staging VGPRs, no
// ds_write traffic (a ds_write_b128 costs 13 LDS-path cycles per KiB, the DMA 4).
// image (b): the same 8 x 32 sub-tiles with the swizzle table T = {0, 2, 3, 1}[(row >> 2) & 3] instead of
// (row >> 2) & 3.  It is the layout of the v_mfma_f32_16x16x32_bf16 kernels below: their fragment reads touch
// 16 rows x 4 chunks (ds_read_b128) resp. 8 rows x 32 B per 32-lane group (ds_read_b64_tr_b16), and T makes both
// conflict-free (the identity table of image (a) is conflict-free for 32 rows x 2 chunks / 4 rows x 64 B).
__device__ __forceinline__ int swz_t(int row) {
  const int h = (row >> 2) & 3;
  return ((((h >> 1) ^ h) & 1) << 1) | (h >> 1);
}
template <int HDP>
__device__ __forceinline__ int imgb_off(int row, int ch) {
  return (row >> 3) * ((HDP / 32) * 512) + (ch >> 2) * 512 + (row & 7) * 64 + (((ch & 3) ^ swz_t(row)) << 4);
}

template <int ROWS, int HDP, int IMG = 0>
struct DmaStage {
  static constexpr int PIECES = ROWS * HDP * 2 / 1024;  // 1-KiB pieces per tile
  static constexpr int PER_WAVE = PIECES / 4;
  unsigned voff[PER_WAVE];  // byte offset of this lane's 16-B chunk relative to the tile's first row
  bool valid[PER_WAVE];
  __device__ __forceinline__ void init(long sl, int hd, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < PER_WAVE; ++i) {
      const int q = wave * PER_WAVE + i;
      const int st = 2 * q + (lane >> 5);
      const int w = lane & 31, r7 = w >> 2, cs = w & 3;
      const int row = (st / (HDP / 32)) * 8 + r7;
      const int ch = (st % (HDP / 32)) * 4 + (cs ^ (IMG == 0 ? ((row >> 2) & 3) : swz_t(row)));
      voff[i] = (unsigned)(((long)row * sl + ch * 8) * 2);
      valid[i] = ch * 8 < hd;
    }
  }
  __device__ __forceinline__ void issue(srd_t rs, char* tile, unsigned row0_bytes, int wave) const {
    const unsigned base = lds_addr_of(tile) + wave * PER_WAVE * 1024;
#pragma unroll
    for (int i = 0; i < PER_WAVE; ++i) {
      const unsigned off = valid[i] ? voff[i] + row0_bytes : 0xfffffff0u;
      lds_dma16(rs, base + i * 1024, off);
    }
  }
};

// ---- forward kernel: lazy softmax + LDS-DMA staging (PIPE: explicit fragment prefetch) ---------
template <int HDP, int HDQ, int WPS, bool PIPE>
__global__ __launch_bounds__(256, WPS) void attn_fwd_kernel(AttnP p) {
  constexpr int KSQ = HDQ / 16, NDB = HDP / 32, TILE = 64 * HDP * 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int bh, qt;
  if (!decode_block(p.n_rt, p.tail_last, p.B * p.H, bh, qt)) return;
  const int b = bh / p.H, hh = bh % p.H;
  const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qrow = qt * 128 + wave * 32 + (lane & 31);

  const __amdgpu_buffer_rsrc_t rq = slice_rsrc(p.q + b * p.q_sb + hh * p.q_sh, p.q_sl, p.Lq, p.hd);
  const srd_t rk = slice_srd(p.k + b * p.k_sb + hh * p.k_sh, p.k_sl, p.Lk, p.hd);
  const srd_t rv = slice_srd(p.v + b * p.v_sb + hh * p.v_sh, p.v_sl, p.Lk, p.hd);

  DmaStage<64, HDP> dk, dv;
  dk.init(p.k_sl, p.hd, wave, lane);
  dv.init(p.v_sl, p.hd, wave, lane);
  const unsigned k_step = (unsigned)(64 * p.k_sl * 2), v_step = (unsigned)(64 * p.v_sl * 2);
  dk.issue(rk, smem, 0, wave);
  dv.issue(rv, smem + TILE, 0, wave);

  bf16x8 qf[KSQ];
#pragma unroll
  for (int ks = 0; ks < KSQ; ++ks) {
    const int e = ks * 16 + 8 * h;
    unsigned off = (unsigned)(((long)qrow * p.q_sl + e) * 2);
    if (e >= p.hd) off = 0xfffffff0u;
    qf[ks] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rq, off, 0, 0));
  }

  f32x16 o[NDB];
#pragma unroll
  for (int i = 0; i < NDB; ++i) o[i] = zero16();
  float m = -1e30f, l = 0.f;
  const float c = p.scale * LOG2E;
  const int nkt = (p.Lk + 63) / 64;
#pragma unroll
  for (int ks = 0; ks < KSQ; ++ks) retire(qf[ks]);
  VDS_WAIT_VM(0);
  __syncthreads();  // tile 0 landed (DMA waited above: the compiler does not see it)

  auto kv_tile = [&](int j, auto PAR) {  // unrolled by two: compile-time LDS buffer parity (see the wide kernel)
    constexpr int par = decltype(PAR)::value;
    if (j + 1 < nkt) {
      char* nk = smem + (par ^ 1) * 2 * TILE;
      dk.issue(rk, nk, (unsigned)(j + 1) * k_step, wave);
      dv.issue(rv, nk + TILE, (unsigned)(j + 1) * v_step, wave);
    }
    const char* kt = smem + par * 2 * TILE;
    const char* vt = kt + TILE;
    const bool ragged = (j == nkt - 1) && (p.Lk & 63);
    if constexpr (PIPE) {
      bf16x8 k0[KSQ], k1[KSQ];
      load_kfrags<HDP, KSQ>(k0, kt, 0, lane);
      load_kfrags<HDP, KSQ>(k1, kt, 32, lane);
      f32x16 s0 = zero16(), s1 = zero16();
#pragma unroll
      for (int ks = 0; ks < KSQ; ++ks) s0 = mfma32(k0[ks], qf[ks], s0);
      bf16x8 v0[2][NDB];
      load_vfrags<HDP, NDB>(v0, vt, 0, lane);
#pragma unroll
      for (int ks = 0; ks < KSQ; ++ks) s1 = mfma32(k1[ks], qf[ks], s1);
      if (ragged) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          if (j * 64 + acc_row(r, h) >= p.Lk) s0[r] = -INFINITY;
          if (j * 64 + 32 + acc_row(r, h) >= p.Lk) s1[r] = -INFINITY;
        }
      }
      bf16x8 pf[2];
      lazy_softmax<NDB>(s0, c, m, l, o, pf);
      bf16x8 v1[2][NDB];
      load_vfrags<HDP, NDB>(v1, vt, 32, lane);
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int db = 0; db < NDB; ++db) o[db] = mfma32(v0[s2][db], pf[s2], o[db]);
      bf16x8 pg[2];
      lazy_softmax<NDB>(s1, c, m, l, o, pg);
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int db = 0; db < NDB; ++db) o[db] = mfma32(v1[s2][db], pg[s2], o[db]);
    } else {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        f32x16 s = zero16();
#pragma unroll
        for (int ks = 0; ks < KSQ; ++ks) s = mfma32(frag_row<HDP>(kt, kb * 32, ks, lane), qf[ks], s);
        if (ragged) {
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (j * 64 + kb * 32 + acc_row(r, h) >= p.Lk) s[r] = -INFINITY;
        }
        bf16x8 pf[2];
        lazy_softmax<NDB>(s, c, m, l, o, pf);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
          for (int db = 0; db < NDB; ++db)
            o[db] = mfma32(frag_tr<HDP>(vt, kb * 32 + 16 * s2, db * 32, lane), pf[s2], o[db]);
      }
    }
    VDS_WAIT_VM(0);
    __syncthreads();  // next tile landed, everyone done reading this one
  };
  for (int j = 0; j < nkt; j += 2) {
    kv_tile(j, std::integral_constant<int, 0>{});
    if (j + 1 < nkt) kv_tile(j + 1, std::integral_constant<int, 1>{});
  }

  const float lt = add_with_other_half(l);
  if (qrow < p.Lq) {
    store_rows<NDB>(p.o + b * p.o_sb + hh * p.o_sh + (long)qrow * p.o_sl, o, 1.0f / lt, p.hd, h);
    if (h == 0) p.lse[((long)b * p.H + hh) * p.Lq + qrow] = (m + __builtin_amdgcn_logf(lt)) * LN2;
  }
}

// ---- forward, wide kernel: 64 queries per wave (two 32-query blocks) --------------------------
// Every K / V^T fragment read from LDS feeds two MFMAs (one per query block).  The kernel is bound by
// VALU issue slots next to the MFMAs (an MFMA gap hides ~24 cycles of VALU issue; the plain online
// softmax needs ~42), so with ONES (k / v rows carry ones columns in their padding, vds_attn_args
// .kv_pad_ones) the softmax sheds its multiply-add and its row-sum adds:
//   * Q is pre-multiplied by scale*log2(e) in registers and carries -m (the running maximum, kept
//     bf16-representable) at column head_dim, where every K row holds 1.0: the QK^T MFMA delivers
//     log2-domain scores MINUS the running maximum, ready for v_exp_f32;
//   * every V row holds 1.0 at columns head_dim and head_dim+4: row head_dim (+4) of the O^T
//     accumulator IS the softmax denominator (rescaled together with O, for free);
//   * the running maximum is only raised when some score exceeds it by 2^LAZY_THR (wave-uniform
//     branch, rare after the first tile); the common path has no cross-lane operation at all.
__device__ __forceinline__ bf16x8 scale_frag(bf16x8 f, float c) {
  bf16x8 r;
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = (__bf16)((float)f[j] * c);
  return r;
}

template <int NDB>
__device__ __forceinline__ void ones_softmax(f32x16& s, float& m, bf16x8& q_ones, f32x16 (&o)[NDB], bf16x8 (&pf)[2],
                                             bool first, int h) {
  float mx = fmaxf(s[0], s[1]);
#pragma unroll
  for (int r = 2; r < 16; r += 2) mx = fmaxf(mx, fmaxf(s[r], s[r + 1]));
  if (first || __builtin_amdgcn_ballot_w64(mx > LAZY_THR) != 0) {  // wave-uniform, rare
    asm volatile("; rescale" ::: "memory");  // keeps this a real branch: hipcc otherwise if-converts it and
                                             // runs the 48 O multiplies on every sub-block with alpha = 1
    const float mxf = max_with_other_half(mx);                      // both lanes of a query agree
    const float m_new = bf2f(f2bf(m + (first ? mxf : fmaxf(mxf, 0.f))));
    const float delta = m_new - m;
    const float alpha = first ? 1.0f : __builtin_amdgcn_exp2f(-delta);
    m = m_new;
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[db][r] *= alpha;  // includes the denominator row
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] -= delta;
    if (h == 1) q_ones[0] = (__bf16)(-m);              // column head_dim of Q' (exact: m is a bf16 value)
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) s[r] = __builtin_amdgcn_exp2f(s[r]);
  pf[0] = acc_frag(s, 0);
  pf[1] = acc_frag(s, 1);
}

template <int HDP, int HDQ, bool ONES>
__global__ __launch_bounds__(256, 2) void attn_fwd_wide_kernel(AttnP p) {
  constexpr int KSQ = HDQ / 16, NDB = HDP / 32, TILE = 64 * HDP * 2;
  static_assert(!ONES || (HDP == 96 && HDQ == 80), "ones columns: head_dim 72 layout");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int bh, qt;
  if (!decode_block(p.n_rt, p.tail_last, p.B * p.H, bh, qt)) return;
  const int b = bh / p.H, hh = bh % p.H;
  const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qrow0 = qt * 256 + wave * 64 + (lane & 31);  // query of block 0; block 1 = + 32
  const int hd_kv = ONES ? p.hd + 8 : p.hd;               // columns of K / V rows that are fetched

  const __amdgpu_buffer_rsrc_t rq = slice_rsrc(p.q + b * p.q_sb + hh * p.q_sh, p.q_sl, p.Lq, p.hd);
  const srd_t rk = slice_srd(p.k + b * p.k_sb + hh * p.k_sh, p.k_sl, p.Lk, hd_kv);
  const srd_t rv = slice_srd(p.v + b * p.v_sb + hh * p.v_sh, p.v_sl, p.Lk, hd_kv);

  DmaStage<64, HDP> dk, dv;
  dk.init(p.k_sl, hd_kv, wave, lane);
  dv.init(p.v_sl, hd_kv, wave, lane);
  const unsigned k_step = (unsigned)(64 * p.k_sl * 2), v_step = (unsigned)(64 * p.v_sl * 2);
  dk.issue(rk, smem, 0, wave);
  dv.issue(rv, smem + TILE, 0, wave);

  const float c = p.scale * LOG2E;
  bf16x8 qf[2][KSQ];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb)
#pragma unroll
    for (int ks = 0; ks < KSQ; ++ks) {
      const int e = ks * 16 + 8 * h;
      unsigned off = (unsigned)(((long)(qrow0 + 32 * qb) * p.q_sl + e) * 2);
      if (e >= p.hd) off = 0xfffffff0u;
      qf[qb][ks] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rq, off, 0, 0));
    }
  if constexpr (ONES) {
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
      for (int ks = 0; ks < KSQ; ++ks) qf[qb][ks] = scale_frag(qf[qb][ks], c);  // also retires the loads
  } else {
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
      for (int ks = 0; ks < KSQ; ++ks) retire(qf[qb][ks]);
  }

  f32x16 o[2][NDB];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb)
#pragma unroll
    for (int i = 0; i < NDB; ++i) o[qb][i] = zero16();
  float m[2], l[2] = {0.f, 0.f};
  m[0] = m[1] = ONES ? 0.f : -1e30f;
  const int nkt = (p.Lk + 63) / 64;
  VDS_WAIT_VM(0);
  __syncthreads();

  // the KV loop is unrolled by two so that the LDS buffer parity is a compile-time constant: every fragment
  // read becomes base register + immediate offset instead of one address add per read
  auto kv_tile = [&](int j, auto PAR) {
    constexpr int par = decltype(PAR)::value;
    if (j + 1 < nkt) {
      char* nk = smem + (par ^ 1) * 2 * TILE;
      dk.issue(rk, nk, (unsigned)(j + 1) * k_step, wave);
      dv.issue(rv, nk + TILE, (unsigned)(j + 1) * v_step, wave);
    }
    const char* kt = smem + par * 2 * TILE;
    const char* vt = kt + TILE;
    const bool ragged = (j == nkt - 1) && (p.Lk & 63);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      bf16x8 kfr[KSQ];
      load_kfrags<HDP, KSQ>(kfr, kt, kb * 32, lane);
      f32x16 s0 = zero16(), s1 = zero16();
#pragma unroll
      for (int ks = 0; ks < KSQ; ++ks) {
        s0 = mfma32(kfr[ks], qf[0][ks], s0);
        s1 = mfma32(kfr[ks], qf[1][ks], s1);
      }
      // keys past Lk: with ONES no mask is needed -- their zero-filled V rows (ones columns included)
      // add nothing to the numerators or to the denominator, whatever exp2 makes of their scores
      if constexpr (!ONES) {
        if (ragged) {
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (j * 64 + kb * 32 + acc_row(r, h) >= p.Lk) { s0[r] = -INFINITY; s1[r] = -INFINITY; }
        }
      }
      bf16x8 vfr[2][NDB];
      load_vfrags<HDP, NDB>(vfr, vt, kb * 32, lane);
      bf16x8 p0[2], p1[2];
      if constexpr (ONES) {
        const bool first = (j == 0) && (kb == 0);
        ones_softmax<NDB>(s0, m[0], qf[0][KSQ - 1], o[0], p0, first, h);
        ones_softmax<NDB>(s1, m[1], qf[1][KSQ - 1], o[1], p1, first, h);
      } else {
        lazy_softmax<NDB>(s0, c, m[0], l[0], o[0], p0);
        lazy_softmax<NDB>(s1, c, m[1], l[1], o[1], p1);
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int db = 0; db < NDB; ++db) {
          o[0][db] = mfma32(vfr[s2][db], p0[s2], o[0][db]);
          o[1][db] = mfma32(vfr[s2][db], p1[s2], o[1][db]);
        }
    }
    VDS_WAIT_VM(0);
    __syncthreads();
  };
  for (int j = 0; j < nkt; j += 2) {
    kv_tile(j, std::integral_constant<int, 0>{});
    if (j + 1 < nkt) kv_tile(j + 1, std::integral_constant<int, 1>{});
  }

#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const int qrow = qrow0 + 32 * qb;
    // ONES: the denominator is row head_dim (lanes 0-31) / head_dim+4 (lanes 32-63) of O^T = register 4 of block 2
    const float lt = ONES ? o[qb][NDB - 1][4] : add_with_other_half(l[qb]);
    if (qrow < p.Lq) {
      store_rows<NDB>(p.o + b * p.o_sb + hh * p.o_sh + (long)qrow * p.o_sl, o[qb], 1.0f / lt, p.hd, h);
      if (h == 0) p.lse[((long)b * p.H + hh) * p.Lq + qrow] = (m[qb] + __builtin_amdgcn_logf(lt)) * LN2;
    }
  }
}

// kv_pad_ones: the (otherwise unused) pad columns hd, hd+1 of the q row receive -lse*log2(e) as a bf16
// (hi, lo) pair; the dK/dV kernel stages Q rows with their pad and holds 1.0 in the matching columns of
// its K fragments, so its QK^T MFMA delivers log2 P directly.
__device__ __forceinline__ void split_bf16(float x, __bf16& hi, __bf16& lo) {
  hi = (__bf16)x;
  lo = (__bf16)(x - (float)hi);
}
__device__ __forceinline__ void annotate_q(const AttnP& p, int b, int hh, int q, float lse2) {
  __bf16 hi, lo;
  split_bf16(-lse2, hi, lo);
  bf16x2 v;
  v[0] = hi;
  v[1] = lo;
  bf16_t* dst = const_cast<bf16_t*>(p.q) + b * p.q_sb + hh * p.q_sh + (long)q * p.q_sl + p.hd;
  *reinterpret_cast<bf16x2*>(dst) = v;
}

// ndelta[b,h,q] = -sum_d dO[q,d] * O[q,d] and lse2 = lse * log2(e) (one wave per row; HBM-bound
// preprocess).  Workspace layout: ndelta[0 .. rows) | lse2[rows .. 2 rows).  The NEGATED delta is
// what the backward kernels load straight into the dP accumulators (dP - delta for free).
__global__ void attn_delta_kernel(AttnP p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long rows = (long)p.B * p.H * p.Lq;
  const long row = (long)blockIdx.x * 4 + wave;
  if (row >= rows) return;
  const int q = row % p.Lq;
  const int bh = row / p.Lq;
  const int b = bh / p.H, hh = bh % p.H;
  const bf16_t* o = p.o + b * p.o_sb + hh * p.o_sh + (long)q * p.o_sl;
  const bf16_t* d = p.d_o + b * p.do_sb + hh * p.do_sh + (long)q * p.do_sl;
  float acc = 0.f;
  for (int e = lane * 2; e < p.hd; e += 128) {
    const unsigned a = *reinterpret_cast<const unsigned*>(o + e);
    const unsigned g = *reinterpret_cast<const unsigned*>(d + e);
    acc += bflo(a) * bflo(g) + bfhi(a) * bfhi(g);
  }
  acc = wave_sum(acc);
  if (lane == 0) {
    const float l2 = p.lse[row] * LOG2E;
    p.delta[row] = -acc;
    p.delta[rows + row] = l2;
    if (p.kv_pad_ones == 1) annotate_q(p, b, hh, q, l2);
  }
}

// token-major fast path of the same preprocess: O and dO are [B*Lq, H*hd] row-major (what the model
// passes), one wave per token reads both rows with 16-byte accesses; each 8-element chunk belongs
// to one head (hd % 8 == 0), the per-chunk partial dots are folded per head through LDS.
__global__ __launch_bounds__(256) void attn_delta_tokmajor_kernel(AttnP p) {
  __shared__ float part[4][192];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long tok = (long)blockIdx.x * 4 + wave;  // over B*Lq
  if (tok >= (long)p.B * p.Lq) return;
  const int b = (int)(tok / p.Lq), q = (int)(tok % p.Lq);
  const int nch = p.H * p.hd / 8, cph = p.hd / 8;
  const bf16_t* o = p.o + b * p.o_sb + (long)q * p.o_sl;
  const bf16_t* d = p.d_o + b * p.do_sb + (long)q * p.do_sl;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int c = lane + 64 * i;
    if (c < nch) {
#if VDS_DELTA_NT  // non-temporal loads of O (saved by the forward pass) and dO: read once
      const u32x4 a = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(o + c * 8));
      const u32x4 g = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(d + c * 8));
#else
      const u32x4 a = *reinterpret_cast<const u32x4*>(o + c * 8);
      const u32x4 g = *reinterpret_cast<const u32x4*>(d + c * 8);
#endif
      float acc = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) acc += bflo(a[e]) * bflo(g[e]) + bfhi(a[e]) * bfhi(g[e]);
      part[wave][c] = acc;
    }
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): this wave's own LDS writes
  __builtin_amdgcn_wave_barrier();
  if (lane < p.H) {
    float acc = 0.f;
    for (int i = 0; i < cph; ++i) acc += part[wave][lane * cph + i];
    const long rows = (long)p.B * p.H * p.Lq;
    const long row = ((long)b * p.H + lane) * p.Lq + q;
    const float l2 = p.lse[row] * LOG2E;
    p.delta[row] = -acc;
    p.delta[rows + row] = l2;
    if (p.kv_pad_ones == 1) annotate_q(p, b, lane, q, l2);
  }
}

// ===================================== dQ ===================================================
// ONES (k / v rows carry ones columns in their padding, vds_attn_args.kv_pad_ones; head_dim 72): the
// kernel is bound by the VALU work between its MFMAs, so the per-query constants move into the MFMAs:
// Q is pre-multiplied by scale*log2(e) in registers and carries -lse*log2(e) as a bf16 (hi, lo) pair at
// columns hd, hd+1 (K holds 1.0 there), dO carries -delta as a (hi, lo) pair at columns hd, hd+4 (V holds
// 1.0 there): S' = log2 P and dP - delta come straight out of the two MFMA chains, leaving
// exp2, one multiply and the bf16 pack per element.  Keys past Lk need no mask in either mode.
template <int HDP, int HDQ, bool ONES>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(AttnP p) {
  constexpr int KSQ = HDQ / 16, NDB = HDP / 32, TILE = 64 * HDP * 2;
  static_assert(!ONES || (HDP == 96 && HDQ == 80), "ones columns: head_dim 72 layout");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int bh, qt;
  if (!decode_block(p.n_rt, p.tail_last, p.B * p.H, bh, qt)) return;
  const int b = bh / p.H, hh = bh % p.H;
  const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qrow = qt * 128 + wave * 32 + (lane & 31);
  const int hd_kv = ONES ? p.hd + 8 : p.hd;

  const __amdgpu_buffer_rsrc_t rq = slice_rsrc(p.q + b * p.q_sb + hh * p.q_sh, p.q_sl, p.Lq, p.hd);
  const __amdgpu_buffer_rsrc_t rdo = slice_rsrc(p.d_o + b * p.do_sb + hh * p.do_sh, p.do_sl, p.Lq, p.hd);
  const srd_t rk = slice_srd(p.k + b * p.k_sb + hh * p.k_sh, p.k_sl, p.Lk, hd_kv);
  const srd_t rv = slice_srd(p.v + b * p.v_sb + hh * p.v_sh, p.v_sl, p.Lk, hd_kv);

  DmaStage<64, HDP> dk, dv;
  dk.init(p.k_sl, hd_kv, wave, lane);
  dv.init(p.v_sl, hd_kv, wave, lane);
  const unsigned k_step = (unsigned)(64 * p.k_sl * 2), v_step = (unsigned)(64 * p.v_sl * 2);
  dk.issue(rk, smem, 0, wave);
  dv.issue(rv, smem + TILE, 0, wave);

  bf16x8 qf[KSQ], dof[KSQ];
#pragma unroll
  for (int ks = 0; ks < KSQ; ++ks) {
    const int e = ks * 16 + 8 * h;
    unsigned off = (unsigned)(((long)qrow * p.q_sl + e) * 2);
    unsigned off2 = (unsigned)(((long)qrow * p.do_sl + e) * 2);
    if (e >= p.hd) { off = 0xfffffff0u; off2 = 0xfffffff0u; }
    qf[ks] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rq, off, 0, 0));
    dof[ks] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rdo, off2, 0, 0));
  }
  const long nrows = (long)p.B * p.H * p.Lq;
  const long srow = ((long)b * p.H + hh) * p.Lq + min(qrow, p.Lq - 1);
  const float ndl = p.delta[srow];  // -delta of this lane's query
  const float lse2 = p.delta[nrows + srow];
  const float c = p.scale * LOG2E;
  if constexpr (ONES) {
#pragma unroll
    for (int ks = 0; ks < KSQ; ++ks) { qf[ks] = scale_frag(qf[ks], c); retire(dof[ks]); }
    if (h == 1) {  // fragment KSQ-1 of lane half 1 holds columns hd .. hd+7
      __bf16 hi, lo;
      split_bf16(-lse2, hi, lo);
      qf[KSQ - 1][0] = hi;
      qf[KSQ - 1][1] = lo;
      split_bf16(ndl, hi, lo);
      dof[KSQ - 1][0] = hi;
      dof[KSQ - 1][4] = lo;
    }
  } else {
#pragma unroll
    for (int ks = 0; ks < KSQ; ++ks) { retire(qf[ks]); retire(dof[ks]); }
    retire(ndl);
    retire(lse2);
  }

  f32x16 dq[NDB];
#pragma unroll
  for (int i = 0; i < NDB; ++i) dq[i] = zero16();
  const int nkt = (p.Lk + 63) / 64;
  VDS_WAIT_VM(0);
  __syncthreads();  // tile 0 landed

  auto kv_tile = [&](int j, auto PAR) {  // unrolled by two: compile-time LDS buffer parity (see the forward kernel)
    constexpr int par = decltype(PAR)::value;
    if (j + 1 < nkt) {
      char* nk = smem + (par ^ 1) * 2 * TILE;
      dk.issue(rk, nk, (unsigned)(j + 1) * k_step, wave);
      dv.issue(rv, nk + TILE, (unsigned)(j + 1) * v_step, wave);
    }
    const char* kt = smem + par * 2 * TILE;
    const char* vt = kt + TILE;
    // keys past Lk need no mask: their K rows are zero-filled, so whatever dS they get multiplies a
    // zero row of K in dQ^T += K^T dS^T
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      f32x16 s = zero16(), dp;
      if constexpr (ONES) {
        dp = zero16();
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) dp[r] = ndl;  // accumulator starts at -delta: dP - delta for free
      }
#pragma unroll
      for (int ks = 0; ks < KSQ; ++ks) {
        s = mfma32(frag_row<HDP>(kt, kb * 32, ks, lane), qf[ks], s);
        dp = mfma32(frag_row<HDP>(vt, kb * 32, ks, lane), dof[ks], dp);
      }
      if constexpr (ONES) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = __builtin_amdgcn_exp2f(s[r]) * dp[r];  // dS^T (unscaled)
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = __builtin_amdgcn_exp2f(s[r] * c - lse2) * dp[r];
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const bf16x8 df = acc_frag(s, s2);
#pragma unroll
        for (int db = 0; db < NDB; ++db)
          dq[db] = mfma32(frag_tr<HDP>(kt, kb * 32 + 16 * s2, db * 32, lane), df, dq[db]);
      }
    }
    VDS_WAIT_VM(0);
    __syncthreads();
  };
  for (int j = 0; j < nkt; j += 2) {
    kv_tile(j, std::integral_constant<int, 0>{});
    if (j + 1 < nkt) kv_tile(j + 1, std::integral_constant<int, 1>{});
  }
  if (qrow < p.Lq)
    store_rows<NDB>(p.dq + b * p.dq_sb + hh * p.dq_sh + (long)qrow * p.dq_sl, dq, p.scale, p.hd, h);
}

// ===================================== dK, dV ===============================================
// Workgroup = 128 keys (4 waves x 32).  Each wave keeps the K and V rows of its 32 keys as MFMA
// B-operand fragments in registers for the whole sweep over 64-row Q/dO tiles; LDS holds only the
// double-buffered Q/dO tiles and their row statistics (lse2, delta), all filled by LDS-DMA.
// Rows past Lq arrive as zeros (SRD bounds): Q = dO = 0 makes their contribution to dV and dK
// vanish whatever P evaluates to.  S and dP have the key on the lane.
// ONES (kv_pad_ones, head_dim 72): Q rows are staged WITH their pad, which the delta preprocess filled
// with -lse*log2(e) (hi, lo); the K fragments are pre-multiplied by scale*log2(e) and hold 1.0 in those
// two columns, so S comes out of the MFMA as log2 P: no multiply-add and no lse reads per element.
template <int HDP, int HDQ, bool ONES>
__global__ __launch_bounds__(256, (HDP > 96 ? 1 : (HDP == 64 ? 3 : 2))) void attn_bwd_dkv_kernel(AttnP p) {
  constexpr int KSQ = HDQ / 16, NDB = HDP / 32, Q_TILE = 64 * HDP * 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // LDS: [buf0: Q | dO][buf1: Q | dO][stats: 2 bufs x (lse2[64], delta[64])]
  char* qbuf = smem;
  char* stats = smem + 4 * Q_TILE;
  int bh, kt_idx, sp = 0;
  if (p.q_split > 1) {  // (head, key tile, query range): the workgroups of a head stay on one XCD like the unsplit order
    const int per = p.n_rt * p.q_split, pid = blockIdx.x, idx = pid >> 3;
    bh = (idx / per) * 8 + (pid & 7);
    if (bh >= p.B * p.H) return;
    kt_idx = (idx % per) % p.n_rt;
    sp = (idx % per) / p.n_rt;
  } else if (!decode_block(p.n_rt, p.tail_last, p.B * p.H, bh, kt_idx)) {
    return;
  }
  const int b = bh / p.H, hh = bh % p.H;
  const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int key0 = kt_idx * 128;
  const int krow = key0 + wave * 32 + (lane & 31);

  const int hd_q = ONES ? p.hd + 8 : p.hd;  // Q row columns that are staged
  const srd_t rq = slice_srd(p.q + b * p.q_sb + hh * p.q_sh, p.q_sl, p.Lq, hd_q);
  const srd_t rdo = slice_srd(p.d_o + b * p.do_sb + hh * p.do_sh, p.do_sl, p.Lq, p.hd);
  const __amdgpu_buffer_rsrc_t rk = slice_rsrc(p.k + b * p.k_sb + hh * p.k_sh, p.k_sl, p.Lk, p.hd);
  const __amdgpu_buffer_rsrc_t rv = slice_rsrc(p.v + b * p.v_sb + hh * p.v_sh, p.v_sl, p.Lk, p.hd);
  const long nrows = (long)p.B * p.H * p.Lq;
  const long srow0 = ((long)b * p.H + hh) * p.Lq;
  // statistics of this head: delta rows then lse2 rows, each bounded to the head's Lq entries
  const srd_t rdl = make_srd(p.delta + srow0, (unsigned)(p.Lq * 4));
  const srd_t rl2 = make_srd(p.delta + nrows + srow0, (unsigned)(p.Lq * 4));
  const float c = p.scale * LOG2E;

  DmaStage<64, HDP> dq_, dd_;
  dq_.init(p.q_sl, hd_q, wave, lane);
  dd_.init(p.do_sl, p.hd, wave, lane);
  const unsigned q_step = (unsigned)(64 * p.q_sl * 2), do_step = (unsigned)(64 * p.do_sl * 2);
  auto issue_tile = [&](int j, int par) {
    char* nb = qbuf + par * 2 * Q_TILE;
    dq_.issue(rq, nb, (unsigned)j * q_step, wave);
    dd_.issue(rdo, nb + Q_TILE, (unsigned)j * do_step, wave);
    if (wave == 0) {  // 64 rows x 4 B each: lse2 then delta
      char* st = stats + par * 512;
      const unsigned sa = lds_addr_of(st);
      lds_dma4(rl2, sa, (unsigned)((j * 64 + lane) * 4));
      lds_dma4(rdl, sa + 256, (unsigned)((j * 64 + lane) * 4));
    }
  };
  // this workgroup's query tiles [j0, j1) (all of them unless the query range is split)
  const int nqt_all = (p.Lq + 63) / 64, jper = (nqt_all + p.q_split - 1) / (p.q_split > 0 ? p.q_split : 1);
  const int j0 = sp * jper, j1 = min(nqt_all, j0 + jper);
  issue_tile(j0, 0);

  // this lane's key row as B-operand fragments (rows past Lk / columns past hd read as zero)
  bf16x8 kf[KSQ], vf[KSQ];
#pragma unroll
  for (int ks = 0; ks < KSQ; ++ks) {
    const int e = ks * 16 + 8 * h;
    unsigned offk = (unsigned)(((long)krow * p.k_sl + e) * 2);
    unsigned offv = (unsigned)(((long)krow * p.v_sl + e) * 2);
    if (e >= p.hd) { offk = 0xfffffff0u; offv = 0xfffffff0u; }
    kf[ks] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rk, offk, 0, 0));
    vf[ks] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rv, offv, 0, 0));
  }
  if constexpr (ONES) {
#pragma unroll
    for (int ks = 0; ks < KSQ; ++ks) kf[ks] = scale_frag(kf[ks], c);
    if (h == 1) {  // columns hd, hd+1 of every key row: 1.0 (they meet -lse2 hi, lo in the Q pad)
      kf[KSQ - 1][0] = (__bf16)1.0f;
      kf[KSQ - 1][1] = (__bf16)1.0f;
    }
  }

  f32x16 dk[NDB], dv[NDB];
#pragma unroll
  for (int i = 0; i < NDB; ++i) { dk[i] = zero16(); dv[i] = zero16(); }
#pragma unroll
  for (int ks = 0; ks < KSQ; ++ks) { retire(kf[ks]); retire(vf[ks]); }
  VDS_WAIT_VM(0);
  __syncthreads();  // tile j0 landed

  auto q_tile = [&](int j, auto PAR) {  // unrolled by two: compile-time LDS buffer parity (see the forward kernel)
    constexpr int par = decltype(PAR)::value;
    if (j + 1 < j1) issue_tile(j + 1, par ^ 1);
    const char* qt = qbuf + par * 2 * Q_TILE;
    const char* dot = qt + Q_TILE;
    const float* stl = reinterpret_cast<const float*>(stats + par * 512);
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      f32x16 s = zero16(), dp;
      // the dP accumulator starts at -delta of its rows (register r <-> row (r&3) + 8 (r>>2) + 4 h:
      // four 16-byte LDS reads land exactly on registers 4 rg .. 4 rg + 3)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const f32x4 d4 = *reinterpret_cast<const f32x4*>(stl + 64 + qb * 32 + 8 * rg + 4 * h);
#pragma unroll
        for (int e = 0; e < 4; ++e) dp[4 * rg + e] = d4[e];
      }
      PRIO_HI();
#pragma unroll
      for (int ks = 0; ks < KSQ; ++ks) {
        s = mfma32(frag_row<HDP>(qt, qb * 32, ks, lane), kf[ks], s);
        dp = mfma32(frag_row<HDP>(dot, qb * 32, ks, lane), vf[ks], dp);
      }
      PRIO_LO();
      f32x16 pm;
      if constexpr (ONES) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float pr = __builtin_amdgcn_exp2f(s[r]);
          pm[r] = pr;
          s[r] = pr * dp[r];  // dS (unscaled)
        }
      } else {
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const f32x4 l4 = *reinterpret_cast<const f32x4*>(stl + qb * 32 + 8 * rg + 4 * h);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int r = 4 * rg + e;
            const float pr = __builtin_amdgcn_exp2f(s[r] * c - l4[e]);
            pm[r] = pr;
            s[r] = pr * dp[r];  // dS (unscaled)
          }
        }
      }
      PRIO_HI();
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const bf16x8 pf = acc_frag(pm, s2);
        const bf16x8 df = acc_frag(s, s2);
#pragma unroll
        for (int db = 0; db < NDB; ++db) {
          dv[db] = mfma32(frag_tr<HDP>(dot, qb * 32 + 16 * s2, db * 32, lane), pf, dv[db]);
          dk[db] = mfma32(frag_tr<HDP>(qt, qb * 32 + 16 * s2, db * 32, lane), df, dk[db]);
        }
      }
      PRIO_LO();
    }
    VDS_WAIT_VM(0);
    __syncthreads();
  };
  for (int j = j0; j < j1; j += 2) {
    q_tile(j, std::integral_constant<int, 0>{});
    if (j + 1 < j1) q_tile(j + 1, std::integral_constant<int, 1>{});
  }
  if (krow < p.Lk) {
    if (p.q_split > 1) {  // fp32 partial of this query range (an empty range stores zeros); dkv_reduce_kernel finishes
      float* part = p.dkv_part + ((((long)sp * p.B * p.H + bh) * p.Lk + krow) * 2) * p.hd;
      store_rows_f32<NDB>(part, dk, p.hd, h);
      store_rows_f32<NDB>(part + p.hd, dv, p.hd, h);
    } else {
      store_rows<NDB>(p.dk + b * p.dk_sb + hh * p.dk_sh + (long)krow * p.dk_sl, dk, p.scale, p.hd, h);
      store_rows<NDB>(p.dv + b * p.dv_sb + hh * p.dv_sh + (long)krow * p.dv_sl, dv, 1.0f, p.hd, h);
    }
  }
}

// ===================================== dK, dV on v_mfma_f32_16x16x32_bf16 ====================
// The attention kernels run power-throttled (the same instruction stream takes 21-25 % less time on all-zero
// operands: gpurun_out/r02e), so what counts is energy per FLOP; MI355X_MICROARCH.md, DVFS item 7:
// the 16x16x32 shape delivers ~1.12-1.15x the FLOP/s of 32x32x16 at equal cycles because the chip holds a higher
// clock on it.  Same algorithm and data flow as attn_bwd_dkv_kernel<96, 80, true> (head_dim 72, ones columns):
//   F1  S[r, c] = sum_k Q[r,k] K[c,k]        A = 16 query rows from LDS (ds_read_b128), B = 16 key rows in registers,
//                                            k in 3 steps of 32 over the 96 padded columns; output: key c on the lane
//                                            (l & 15), query rows 4 (l >> 4) .. + 3
//   F2  dV^T[d, c] += sum_r dO[r,d] P[r,c]   B = the S / dP accumulators of the two 16-row query blocks of a 32-row
//                                            sub-block, used in place (lane (c, g) holds rows 4g..4g+3 of both blocks
//                                            = the 8 contraction indices the MFMA wants from it); A = dO^T read with
//                                            two transposing 4 x 16 reads at rows R + 4g, R + 16 + 4g
// A wave owns 32 keys as two 16-key column blocks; head_dim pads to 96 for the contractions (3 x 32) and to 80 for
// the outputs (5 x 16): 44 MFMAs of 16 cycles per 32 x 32 block = the 704 cycles of the 32x32x16 kernel.
__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
template <int HDP>
__device__ __forceinline__ bf16x8 frag16_row(const char* tile, int row0, int ks, int lane) {
  return *reinterpret_cast<const bf16x8*>(tile + imgb_off<HDP>(row0 + (lane & 15), ks * 4 + (lane >> 4)));
}
// A operand of F2: (T^T)[d0 .. d0+15][32 rows r0 ..]: contraction index 8g + j <-> row r0 + 4g + j (j < 4),
// r0 + 16 + 4g + (j - 4) (j >= 4) -- the order in which lane (c, g) holds two stacked 16-row accumulators
template <int HDP>
__device__ __forceinline__ bf16x8 frag16_tr(const char* tile, int r0, int d0, int lane) {
  const int g = lane >> 4, i = lane & 15, qq = i >> 2, pp = i & 3;
  const int row = r0 + 4 * g + qq;
  const int ch = (d0 >> 3) + (pp >> 1);
  const char* p0 = tile + imgb_off<HDP>(row, ch) + 8 * (pp & 1);
  const char* p1 = tile + imgb_off<HDP>(row + 16, ch) + 8 * (pp & 1);
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)LDS_PTR(p0));
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)LDS_PTR(p1));
  s16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, r);
}
__device__ __forceinline__ bf16x8 pack2(const f32x4& a, const f32x4& b) {
  bf16x8 r;
#pragma unroll
  for (int j = 0; j < 4; ++j) { r[j] = (__bf16)a[j]; r[4 + j] = (__bf16)b[j]; }
  return r;
}

// QPAD = false (round 5, cross-attention: vds_attn_args.kv_pad_ones = 2): the q rows carry no annotated pad (token-major
// views of a linear layer's output), so only head_dim columns of a q row are staged and the S accumulators START from
// -lse2 of their queries, read from the statistics this kernel stages beside -delta anyway.
template <int HDP, bool QPAD = true>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv16_kernel(AttnP p) {
  static_assert(HDP == 96, "head_dim 72 layout (ones columns at 72, 73)");
  constexpr int KS = HDP / 32, NDB = 5, Q_TILE = 64 * HDP * 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* qbuf = smem;
  char* stats = smem + 4 * Q_TILE;
  int bh, kt_idx;
  if (!decode_block(p.n_rt, p.tail_last, p.B * p.H, bh, kt_idx)) return;
  const int b = bh / p.H, hh = bh % p.H;
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int key0 = kt_idx * 128 + wave * 32;

  const int hd_q = QPAD ? p.hd + 8 : p.hd;  // QPAD: Q rows are staged with their pad: -lse*log2(e) as (hi, lo) at columns hd, hd+1
  const srd_t rq = slice_srd(p.q + b * p.q_sb + hh * p.q_sh, p.q_sl, p.Lq, hd_q);
  const srd_t rdo = slice_srd(p.d_o + b * p.do_sb + hh * p.do_sh, p.do_sl, p.Lq, p.hd);
  const __amdgpu_buffer_rsrc_t rk = slice_rsrc(p.k + b * p.k_sb + hh * p.k_sh, p.k_sl, p.Lk, p.hd);
  const __amdgpu_buffer_rsrc_t rv = slice_rsrc(p.v + b * p.v_sb + hh * p.v_sh, p.v_sl, p.Lk, p.hd);
  const long nrows = (long)p.B * p.H * p.Lq;
  const long srow0 = ((long)b * p.H + hh) * p.Lq;
  const srd_t rdl = make_srd(p.delta + srow0, (unsigned)(p.Lq * 4));
  const srd_t rl2 = make_srd(p.delta + nrows + srow0, (unsigned)(p.Lq * 4));
  const float c = p.scale * LOG2E;

  DmaStage<64, HDP, 1> dq_, dd_;
  dq_.init(p.q_sl, hd_q, wave, lane);
  dd_.init(p.do_sl, p.hd, wave, lane);
  const unsigned q_step = (unsigned)(64 * p.q_sl * 2), do_step = (unsigned)(64 * p.do_sl * 2);
  auto issue_tile = [&](int j, int par) {
    char* nb = qbuf + par * 2 * Q_TILE;
    dq_.issue(rq, nb, (unsigned)j * q_step, wave);
    dd_.issue(rdo, nb + Q_TILE, (unsigned)j * do_step, wave);
    if (wave == 0) {
      char* st = stats + par * 512;
      const unsigned sa = lds_addr_of(st);
      lds_dma4(rl2, sa, (unsigned)((j * 64 + lane) * 4));
      lds_dma4(rdl, sa + 256, (unsigned)((j * 64 + lane) * 4));
    }
  };
  issue_tile(0, 0);

  // the two 16-key blocks of this wave as B operands: lane (c = l & 15, g) holds columns 32 ks + 8 g .. + 7 of key c
  bf16x8 kf[2][KS], vf[2][KS];
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) {
    const int krow = key0 + cb * 16 + (lane & 15);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int e = ks * 32 + 8 * g;
      unsigned offk = (unsigned)(((long)krow * p.k_sl + e) * 2);
      unsigned offv = (unsigned)(((long)krow * p.v_sl + e) * 2);
      if (e >= p.hd) { offk = 0xfffffff0u; offv = 0xfffffff0u; }
      kf[cb][ks] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rk, offk, 0, 0));
      vf[cb][ks] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rv, offv, 0, 0));
    }
  }
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) { kf[cb][ks] = scale_frag(kf[cb][ks], c); retire(vf[cb][ks]); }
    if (g == 1) {  // ks = 2, g = 1: columns 72 .. 79; 1.0 at 72, 73 meets -lse2 (hi, lo) in the Q pad
      kf[cb][KS - 1][0] = (__bf16)1.0f;
      kf[cb][KS - 1][1] = (__bf16)1.0f;
    }
  }

  f32x4 dk[NDB][2], dv[NDB][2];
#pragma unroll
  for (int i = 0; i < NDB; ++i)
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) { dk[i][cb] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[i][cb] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  const int nqt = (p.Lq + 63) / 64;
  VDS_WAIT_VM(0);
  __syncthreads();  // tile 0 landed

  // NKB: 16-key blocks of this wave that hold keys (2 except in the head's last workgroup, which owns the 16 register
  // tokens at L = 8208: its other waves only stage and synchronise; see attn_fwd16_kernel)
  auto q_tile = [&](int j, auto PAR, auto NKBT) {
    constexpr int par = decltype(PAR)::value;
    constexpr int NKB = decltype(NKBT)::value;
    if (j + 1 < nqt) issue_tile(j + 1, par ^ 1);
    const char* qt = qbuf + par * 2 * Q_TILE;
    const char* dot = qt + Q_TILE;
    const float* stl = reinterpret_cast<const float*>(stats + par * 512);
    if constexpr (NKB > 0)
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      f32x4 s[2][2], dp[2][2];
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) {
        const f32x4 d4 = *reinterpret_cast<const f32x4*>(stl + 64 + qb * 32 + rb * 16 + 4 * g);  // -delta of rows 4g..
        f32x4 s0 = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (!QPAD) s0 = -*reinterpret_cast<const f32x4*>(stl + qb * 32 + rb * 16 + 4 * g);  // -lse2 of rows 4g..
#pragma unroll
        for (int cb = 0; cb < NKB; ++cb) { s[rb][cb] = s0; dp[rb][cb] = d4; }
      }
      PRIO_HI();
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
          const bf16x8 aq = frag16_row<HDP>(qt, qb * 32 + rb * 16, ks, lane);
          const bf16x8 ad = frag16_row<HDP>(dot, qb * 32 + rb * 16, ks, lane);
#pragma unroll
          for (int cb = 0; cb < NKB; ++cb) {
            s[rb][cb] = mfma16(aq, kf[cb][ks], s[rb][cb]);
            dp[rb][cb] = mfma16(ad, vf[cb][ks], dp[rb][cb]);
          }
        }
      PRIO_LO();
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int cb = 0; cb < NKB; ++cb)
#pragma unroll
          for (int r = 0; r < 4; ++r) s[rb][cb][r] = __builtin_amdgcn_exp2f(s[rb][cb][r]);  // S came out of the MFMA as log2 P
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int cb = 0; cb < NKB; ++cb)
#pragma unroll
          for (int r = 0; r < 4; ++r) dp[rb][cb][r] *= s[rb][cb][r];  // dS (unscaled)
      bf16x8 pf[2], df[2];
#pragma unroll
      for (int cb = 0; cb < NKB; ++cb) {
        pf[cb] = pack2(s[0][cb], s[1][cb]);
        df[cb] = pack2(dp[0][cb], dp[1][cb]);
      }
      PRIO_HI();
#pragma unroll
      for (int db = 0; db < NDB; ++db) {
        const bf16x8 ado = frag16_tr<HDP>(dot, qb * 32, db * 16, lane);
        const bf16x8 aqt = frag16_tr<HDP>(qt, qb * 32, db * 16, lane);
#pragma unroll
        for (int cb = 0; cb < NKB; ++cb) {
          dv[db][cb] = mfma16(ado, pf[cb], dv[db][cb]);
          dk[db][cb] = mfma16(aqt, df[cb], dk[db][cb]);
        }
      }
      PRIO_LO();
    }
    VDS_WAIT_VM(0);
    __syncthreads();
  };
  auto run = [&](auto NKBT) {
    for (int j = 0; j < nqt; j += 2) {
      q_tile(j, std::integral_constant<int, 0>{}, NKBT);
      if (j + 1 < nqt) q_tile(j + 1, std::integral_constant<int, 1>{}, NKBT);
    }
  };
  const int nkb = min(2, max(0, (p.Lk - key0 + 15) >> 4));  // wave-uniform
  if (nkb == 2) run(std::integral_constant<int, 2>{});
  else if (nkb == 1) run(std::integral_constant<int, 1>{});
  else run(std::integral_constant<int, 0>{});
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) {
    const int krow = key0 + cb * 16 + (lane & 15);
    float vk[NDB][4], vv[NDB][4];
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
      for (int e = 0; e < 4; ++e) { vk[db][e] = dk[db][cb][e]; vv[db][e] = dv[db][cb][e]; }
    store_block_bf16_t(vk, p.scale, p.dk + b * p.dk_sb + hh * p.dk_sh + (long)krow * p.dk_sl, p.hd, g, krow < p.Lk, p.wide_dkv);
    store_block_bf16_t(vv, 1.0f, p.dv + b * p.dv_sb + hh * p.dv_sh + (long)krow * p.dv_sl, p.hd, g, krow < p.Lk, p.wide_dkv);
  }
}

// ===================================== forward on v_mfma_f32_16x16x32_bf16 ====================
// attn_fwd_wide_kernel<96, 80, true> (head_dim 72, ones columns, 64 queries per wave) on the 16x16x32 shape, for the
// reason given at attn_bwd_dkv16_kernel (higher held clock at equal MFMA cycles).  S^T = K Q'^T with 16 key rows from
// LDS as A and 16 queries in registers as B (k in 3 steps of 32); O^T += V^T P^T with V^T read by two transposing
// 4 x 16 reads and the two stacked S accumulators of a 32-key sub-block used in place as B.  A wave owns 4 query
// blocks of 16; lane (c = l & 15, g = l >> 4) holds key rows 4g..4g+3 of each 16-key block of query c.  The ones-column
// softmax is unchanged: -m rides in column head_dim of Q', the denominator comes out of the PV MFMA as rows head_dim
// (lanes g = 2) and head_dim + 4 (g = 3) of O^T, the running maximum is raised lazily (wave-uniform branch).
__device__ __forceinline__ float max3f(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }
__device__ __forceinline__ float max_over_lane_groups(float x) {  // max over the 4 lanes l & 15 + 16 g
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  x = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}

template <int HDP>
__global__ __launch_bounds__(256, 2) void attn_fwd16_kernel(AttnP p) {
  static_assert(HDP == 96, "head_dim 72 layout (ones columns at 72, 73 / 72, 76)");
  constexpr int KS = HDP / 32, NDB = 5, TILE = 64 * HDP * 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int bh, qt;
  if (!decode_block(p.n_rt, p.tail_last, p.B * p.H, bh, qt)) return;
  const int b = bh / p.H, hh = bh % p.H;
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qrow0 = qt * 256 + wave * 64 + (lane & 15);  // query of block 0; block cb = + 16 cb
  const int hd_kv = p.hd + 8;

  const __amdgpu_buffer_rsrc_t rq = slice_rsrc(p.q + b * p.q_sb + hh * p.q_sh, p.q_sl, p.Lq, p.hd);
  const srd_t rk = slice_srd(p.k + b * p.k_sb + hh * p.k_sh, p.k_sl, p.Lk, hd_kv);
  const srd_t rv = slice_srd(p.v + b * p.v_sb + hh * p.v_sh, p.v_sl, p.Lk, hd_kv);
  DmaStage<64, HDP, 1> dk, dv;
  dk.init(p.k_sl, hd_kv, wave, lane);
  dv.init(p.v_sl, hd_kv, wave, lane);
  const unsigned k_step = (unsigned)(64 * p.k_sl * 2), v_step = (unsigned)(64 * p.v_sl * 2);
  dk.issue(rk, smem, 0, wave);
  dv.issue(rv, smem + TILE, 0, wave);

  const float c = p.scale * LOG2E;
  bf16x8 qf[4][KS];
#pragma unroll
  for (int cb = 0; cb < 4; ++cb)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int e = ks * 32 + 8 * g;
      unsigned off = (unsigned)(((long)(qrow0 + 16 * cb) * p.q_sl + e) * 2);
      if (e >= p.hd) off = 0xfffffff0u;
      qf[cb][ks] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rq, off, 0, 0));
    }
#pragma unroll
  for (int cb = 0; cb < 4; ++cb)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[cb][ks] = scale_frag(qf[cb][ks], c);  // also retires the loads

  f32x4 o[NDB][4];
#pragma unroll
  for (int db = 0; db < NDB; ++db)
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) o[db][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m[4] = {0.f, 0.f, 0.f, 0.f};
  const int nkt = (p.Lk + 63) / 64;
  VDS_WAIT_VM(0);
  __syncthreads();

  // NCB: query blocks of this wave that hold queries: 4 except in a head's last workgroup (16 of its 256 queries are
  // real at L = 8208), whose other waves only stage and synchronise -- that workgroup then runs at the pace of the one
  // wave that has a block instead of occupying a CU slot for the time of a full one.  Invisible at B = 12 (12.4 rounds
  // of workgroups either way), but at B = 2 the 33rd workgroup of every head opens a third round.
  auto kv_tile = [&](int j, auto PAR, auto NCBT) {
    constexpr int par = decltype(PAR)::value;
    constexpr int NCB = decltype(NCBT)::value;
    if (j + 1 < nkt) {
      char* nk = smem + (par ^ 1) * 2 * TILE;
      dk.issue(rk, nk, (unsigned)(j + 1) * k_step, wave);
      dv.issue(rv, nk + TILE, (unsigned)(j + 1) * v_step, wave);
    }
    const char* kt = smem + par * 2 * TILE;
    const char* vt = kt + TILE;
    if constexpr (NCB > 0)
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      f32x4 s[2][4];
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) s[rb][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
      PRIO_HI();
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
          const bf16x8 ak = frag16_row<HDP>(kt, kb * 32 + rb * 16, ks, lane);
#pragma unroll
          for (int cb = 0; cb < NCB; ++cb) s[rb][cb] = mfma16(ak, qf[cb][ks], s[rb][cb]);
        }
      PRIO_LO();
      // keys past Lk need no mask: their zero-filled V rows (ones columns included) add nothing to the numerators or
      // to the denominator, whatever exp2 makes of their scores
      const bool first = (j == 0) && (kb == 0);
      float mx[4];
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
        // 4 instructions per query block as three-operand maxima (the issue port is ~90 % booked: tools/probes)
        mx[cb] = max3f(max3f(s[0][cb][0], s[0][cb][1], s[0][cb][2]), max3f(s[0][cb][3], s[1][cb][0], s[1][cb][1]),
                       fmaxf(s[1][cb][2], s[1][cb][3]));
      }
      float mxa = mx[0];
      if constexpr (NCB == 4) mxa = fmaxf(max3f(mx[0], mx[1], mx[2]), mx[3]);
      else {
#pragma unroll
        for (int cb = 1; cb < NCB; ++cb) mxa = fmaxf(mxa, mx[cb]);
      }
      if (first || __builtin_amdgcn_ballot_w64(mxa > LAZY_THR) != 0) {  // wave-uniform, rare after the first tile
        asm volatile("; rescale" ::: "memory");  // keeps this a real branch (no if-conversion of the O multiplies)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
          const float mxf = max_over_lane_groups(mx[cb]);  // all four lanes of a query agree
          const float m_new = bf2f(f2bf(m[cb] + (first ? mxf : fmaxf(mxf, 0.f))));
          const float delta = m_new - m[cb];
          const float alpha = first ? 1.0f : __builtin_amdgcn_exp2f(-delta);
          m[cb] = m_new;
#pragma unroll
          for (int db = 0; db < NDB; ++db) o[db][cb] *= alpha;  // includes the denominator rows
#pragma unroll
          for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int r = 0; r < 4; ++r) s[rb][cb][r] -= delta;
          if (g == 1) qf[cb][KS - 1][0] = (__bf16)(-m_new);  // column head_dim of Q' (exact: m is a bf16 value)
        }
      }
      bf16x8 pf[4];
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
          for (int r = 0; r < 4; ++r) s[rb][cb][r] = __builtin_amdgcn_exp2f(s[rb][cb][r]);
        pf[cb] = pack2(s[0][cb], s[1][cb]);
      }
      PRIO_HI();
#pragma unroll
      for (int db = 0; db < NDB; ++db) {
        const bf16x8 av = frag16_tr<HDP>(vt, kb * 32, db * 16, lane);
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) o[db][cb] = mfma16(av, pf[cb], o[db][cb]);
      }
      PRIO_LO();
    }
    VDS_WAIT_VM(0);
    __syncthreads();
  };
  auto run = [&](auto NCBT) {
    for (int j = 0; j < nkt; j += 2) {
      kv_tile(j, std::integral_constant<int, 0>{}, NCBT);
      if (j + 1 < nkt) kv_tile(j + 1, std::integral_constant<int, 1>{}, NCBT);
    }
  };
  const int ncb = min(4, max(0, (p.Lq - (qt * 256 + wave * 64) + 15) >> 4));  // wave-uniform
  if (ncb > 2) run(std::integral_constant<int, 4>{});
  else if (ncb == 2) run(std::integral_constant<int, 2>{});
  else if (ncb == 1) run(std::integral_constant<int, 1>{});
  else run(std::integral_constant<int, 0>{});

#pragma unroll
  for (int cb = 0; cb < 4; ++cb) {
    const int qrow = qrow0 + 16 * cb;
    // the denominator is row head_dim (= 64 + 4*2 + 0: lanes g = 2, register 0 of block 4) of O^T
    const float lt = __shfl(o[NDB - 1][cb][0], (lane & 15) | 32, 64);
    float vo[NDB][4];
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
      for (int e = 0; e < 4; ++e) vo[db][e] = o[db][cb][e];
    store_block_bf16_t(vo, 1.0f / lt, p.o + b * p.o_sb + hh * p.o_sh + (long)qrow * p.o_sl, p.hd, g, qrow < p.Lq, p.wide_o);
    if (qrow < p.Lq && g == 0) p.lse[((long)b * p.H + hh) * p.Lq + qrow] = (m[cb] + __builtin_amdgcn_logf(lt)) * LN2;
  }
}

// ===================================== dQ on v_mfma_f32_16x16x32_bf16 =========================
// attn_bwd_dq_kernel<96, 80, true> on the 16x16x32 shape (see attn_bwd_dkv16_kernel).  S'^T = K Q'^T and
// dP^T = V dO'^T with 16 key rows from LDS as A and 16 queries in registers as B; dQ^T += K^T dS^T with K^T read by
// transposing reads and the stacked dS^T accumulators used in place as B.  Per 32 x 32 block: 34 MFMAs of 16 cycles
// (544) against 16 of 32 (512) -- the contraction pads 72 -> 96 instead of 80 -- in exchange for the higher clock.
// NQ = 2 16-query blocks per wave (three waves per SIMD) reads every K / V fragment for 2 MFMAs: at the full MFMA rate the
// 12 waves of a CU would keep the LDS 100 % busy (136 LDS cycles per wave and 32-key sub-block against 544 MFMA cycles per
// SIMD and wave) -- the kernel is LDS-bound.
// FOLD (round 6): the kernel is its own preprocess.  It holds the dO rows of its queries as fragments anyway; with the O
// rows loaded beside them (once per workgroup: 18 KiB) every lane forms  -delta = -rowsum(dO o O)  of its query (24 products,
// two cross-lane adds), writes -delta and lse * log2(e) to the workspace and the (hi, lo) pair of -lse2 into the pad of its
// Q row -- what attn_delta_*_kernel did in a pass of its own over O and dO (6.5 ms per C3b step).  The host then runs this
// kernel BEFORE the dK/dV kernel, which reads those statistics.
template <int HDP, bool FOLD = false>
__global__ __launch_bounds__(256, 3) void attn_bwd_dq16_kernel(AttnP p) {
  static_assert(HDP == 96, "head_dim 72 layout");
  constexpr int KS = HDP / 32, NDB = 5, TILE = 64 * HDP * 2, NQ = 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int bh, qt;
  if (!decode_block(p.n_rt, p.tail_last, p.B * p.H, bh, qt)) return;
  const int b = bh / p.H, hh = bh % p.H;
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qrow0 = qt * (64 * NQ) + wave * (16 * NQ) + (lane & 15);  // query of column block 0; block cb = + 16 cb
  const int hd_kv = p.hd + 8;

  const __amdgpu_buffer_rsrc_t rq = slice_rsrc(p.q + b * p.q_sb + hh * p.q_sh, p.q_sl, p.Lq, p.hd);
  const __amdgpu_buffer_rsrc_t rdo = slice_rsrc(p.d_o + b * p.do_sb + hh * p.do_sh, p.do_sl, p.Lq, p.hd);
  const srd_t rk = slice_srd(p.k + b * p.k_sb + hh * p.k_sh, p.k_sl, p.Lk, hd_kv);
  const srd_t rv = slice_srd(p.v + b * p.v_sb + hh * p.v_sh, p.v_sl, p.Lk, hd_kv);
  DmaStage<64, HDP, 1> dk, dv;
  dk.init(p.k_sl, hd_kv, wave, lane);
  dv.init(p.v_sl, hd_kv, wave, lane);
  const unsigned k_step = (unsigned)(64 * p.k_sl * 2), v_step = (unsigned)(64 * p.v_sl * 2);
  dk.issue(rk, smem, 0, wave);
  dv.issue(rv, smem + TILE, 0, wave);

  const long nrows = (long)p.B * p.H * p.Lq;
  const float c = p.scale * LOG2E;
  bf16x8 qf[NQ][KS], dof[NQ][KS];
#pragma unroll
  for (int cb = 0; cb < NQ; ++cb) {
    const int qrow = qrow0 + 16 * cb;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int e = ks * 32 + 8 * g;
      unsigned off = (unsigned)(((long)qrow * p.q_sl + e) * 2);
      unsigned off2 = (unsigned)(((long)qrow * p.do_sl + e) * 2);
      if (e >= p.hd) { off = 0xfffffff0u; off2 = 0xfffffff0u; }
      qf[cb][ks] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rq, off, 0, 0));
      dof[cb][ks] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rdo, off2, 0, 0));
    }
    const long srow = ((long)b * p.H + hh) * p.Lq + min(qrow, p.Lq - 1);
    float ndl, lse2;
    if constexpr (FOLD) {
      const __amdgpu_buffer_rsrc_t ro = slice_rsrc(p.o + b * p.o_sb + hh * p.o_sh, p.o_sl, p.Lq, p.hd);
      float acc = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int e = ks * 32 + 8 * g;
        const unsigned off3 = e >= p.hd ? 0xfffffff0u : (unsigned)(((long)qrow * p.o_sl + e) * 2);
        const bf16x8 of = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(ro, off3, 0, 0));
#pragma unroll
        for (int j = 0; j < 8; ++j) acc = __builtin_fmaf((float)of[j], (float)dof[cb][ks][j], acc);
      }
      acc += __shfl_xor(acc, 16);  // the four lanes l, l + 16, l + 32, l + 48 hold the column chunks of one query
      acc += __shfl_xor(acc, 32);
      ndl = -acc;
      lse2 = p.lse[srow] * LOG2E;
      if (g == 0 && qrow < p.Lq) {
        p.delta[srow] = ndl;
        p.delta[nrows + srow] = lse2;
        if (p.kv_pad_ones == 1) annotate_q(p, b, hh, qrow, lse2);
      }
    } else {
      ndl = p.delta[srow];           // -delta of this lane's query
      lse2 = p.delta[nrows + srow];  // lse * log2(e)
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) { qf[cb][ks] = scale_frag(qf[cb][ks], c); retire(dof[cb][ks]); }
    if (g == 1) {  // ks = 2, g = 1: columns 72 .. 79 (K holds 1.0 at 72, 73; V at 72, 76)
      __bf16 hi, lo;
      split_bf16(-lse2, hi, lo);
      qf[cb][KS - 1][0] = hi;
      qf[cb][KS - 1][1] = lo;
      split_bf16(ndl, hi, lo);
      dof[cb][KS - 1][0] = hi;
      dof[cb][KS - 1][4] = lo;
    }
  }

  f32x4 dq[NDB][NQ];
#pragma unroll
  for (int db = 0; db < NDB; ++db)
#pragma unroll
    for (int cb = 0; cb < NQ; ++cb) dq[db][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nkt = (p.Lk + 63) / 64;
  VDS_WAIT_VM(0);
  __syncthreads();  // tile 0 landed

  // NCB: 16-query blocks of this wave that hold queries (see attn_fwd16_kernel)
  auto kv_tile = [&](int j, auto PAR, auto NCBT) {
    constexpr int par = decltype(PAR)::value;
    constexpr int NCB = decltype(NCBT)::value;
    if (j + 1 < nkt) {
      char* nk = smem + (par ^ 1) * 2 * TILE;
      dk.issue(rk, nk, (unsigned)(j + 1) * k_step, wave);
      dv.issue(rv, nk + TILE, (unsigned)(j + 1) * v_step, wave);
    }
    const char* kt = smem + par * 2 * TILE;
    const char* vt = kt + TILE;
    // keys past Lk need no mask: their K rows are zero-filled, so whatever dS they get multiplies a zero row of K
    if constexpr (NCB > 0)
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      f32x4 s[2][NQ], dp[2][NQ];
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) { s[rb][cb] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[rb][cb] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
          const bf16x8 ak = frag16_row<HDP>(kt, kb * 32 + rb * 16, ks, lane);
          const bf16x8 av = frag16_row<HDP>(vt, kb * 32 + rb * 16, ks, lane);
#pragma unroll
          for (int cb = 0; cb < NCB; ++cb) {
            s[rb][cb] = mfma16(ak, qf[cb][ks], s[rb][cb]);
            dp[rb][cb] = mfma16(av, dof[cb][ks], dp[rb][cb]);
          }
        }
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
          for (int r = 0; r < 4; ++r) s[rb][cb][r] = __builtin_amdgcn_exp2f(s[rb][cb][r]);
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
          for (int r = 0; r < 4; ++r) s[rb][cb][r] *= dp[rb][cb][r];  // dS^T (unscaled)
      bf16x8 df[NQ];
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) df[cb] = pack2(s[0][cb], s[1][cb]);
#pragma unroll
      for (int db = 0; db < NDB; ++db) {
        const bf16x8 akt = frag16_tr<HDP>(kt, kb * 32, db * 16, lane);
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) dq[db][cb] = mfma16(akt, df[cb], dq[db][cb]);
      }
    }
    VDS_WAIT_VM(0);
    __syncthreads();
  };
  auto run = [&](auto NCBT) {
    for (int j = 0; j < nkt; j += 2) {
      kv_tile(j, std::integral_constant<int, 0>{}, NCBT);
      if (j + 1 < nkt) kv_tile(j + 1, std::integral_constant<int, 1>{}, NCBT);
    }
  };
  const int ncb = min(NQ, max(0, (p.Lq - (qt * (64 * NQ) + wave * (16 * NQ)) + 15) >> 4));  // wave-uniform
  if (ncb == NQ) run(std::integral_constant<int, NQ>{});
  else if (ncb == 2) run(std::integral_constant<int, 2>{});
  else if (ncb == 1) run(std::integral_constant<int, 1>{});
  else run(std::integral_constant<int, 0>{});
#pragma unroll
  for (int cb = 0; cb < NQ; ++cb) {
    const int qrow = qrow0 + 16 * cb;
    float vq[NDB][4];
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
      for (int e = 0; e < 4; ++e) vq[db][e] = dq[db][cb][e];
    store_block_bf16_t(vq, p.scale, p.dq + b * p.dq_sb + hh * p.dq_sh + (long)qrow * p.dq_sl, p.hd, g, qrow < p.Lq, p.wide_dq);
  }
}

bool rows16(const void* base, long sb, long sh, long sl) {
  return base && ((uintptr_t)base % 16 == 0) && (sb % 8 == 0) && (sh % 8 == 0) && (sl % 8 == 0);
}

AttnP to_p(const vds_attn_args* a) {
  AttnP p;
  p.B = a->B; p.H = a->H; p.Lq = a->Lq; p.Lk = a->Lk; p.hd = a->head_dim;
  p.q = (const bf16_t*)a->q; p.q_sb = a->q_sb; p.q_sh = a->q_sh; p.q_sl = a->q_sl;
  p.k = (const bf16_t*)a->k; p.k_sb = a->k_sb; p.k_sh = a->k_sh; p.k_sl = a->k_sl;
  p.v = (const bf16_t*)a->v; p.v_sb = a->v_sb; p.v_sh = a->v_sh; p.v_sl = a->v_sl;
  p.o = (bf16_t*)a->o; p.o_sb = a->o_sb; p.o_sh = a->o_sh; p.o_sl = a->o_sl;
  p.lse = a->lse;
  p.d_o = (const bf16_t*)a->d_o; p.do_sb = a->do_sb; p.do_sh = a->do_sh; p.do_sl = a->do_sl;
  p.dq = (bf16_t*)a->dq; p.dq_sb = a->dq_sb; p.dq_sh = a->dq_sh; p.dq_sl = a->dq_sl;
  p.dk = (bf16_t*)a->dk; p.dk_sb = a->dk_sb; p.dk_sh = a->dk_sh; p.dk_sl = a->dk_sl;
  p.dv = (bf16_t*)a->dv; p.dv_sb = a->dv_sb; p.dv_sh = a->dv_sh; p.dv_sl = a->dv_sl;
  p.delta = a->delta;
  p.scale = 1.0f / sqrtf((float)a->head_dim);
  p.n_rt = 0;
  p.tail_last = 0;
  p.kv_pad_ones = a->kv_pad_ones;
  p.q_split = 1;
  p.dkv_part = nullptr;
  const bool wide_on = vdscfg::geti(vdscfg::ATTN_WIDE_STORES) != 0;  // A/B knob
  p.wide_o = wide_on && p.hd > 64 && rows16(p.o, p.o_sb, p.o_sh, p.o_sl);
  p.wide_dq = wide_on && p.hd > 64 && rows16(p.dq, p.dq_sb, p.dq_sh, p.dq_sl);
  p.wide_dkv = wide_on && p.hd > 64 && rows16(p.dk, p.dk_sb, p.dk_sh, p.dk_sl) && rows16(p.dv, p.dv_sb, p.dv_sh, p.dv_sl);
  return p;
}

// The kernels are templates on the PADDED head dim (HDP: the Q K^T / dO V^T contraction, HDQ: the width of O / dQ / dK / dV);
// rows are fetched with every 16-byte chunk at or past head_dim redirected out of range (zeros) and only head_dim columns
// are stored, so any head_dim that is a multiple of 8 runs on the next instance (model.py:57 of the reference takes any
// hidden_size // num_heads): <= 32, <= 64, <= 80 (HDP 96, HDQ 80: DiT-XL's 72), <= 96, <= 128.  0 = no instance.
int kernel_instance(int hd) {
  if (hd <= 0 || (hd & 7) || hd > 128) return 0;
  return hd <= 32 ? 32 : hd <= 64 ? 64 : hd <= 80 ? 80 : hd <= 96 ? 96 : 128;
}

bool strides_ok(const vds_attn_args* a, bool bwd) {
  auto ok = [](int64_t s) { return (s & 7) == 0; };
  bool r = ok(a->q_sb) && ok(a->q_sh) && ok(a->q_sl) && ok(a->k_sb) && ok(a->k_sh) && ok(a->k_sl) &&
           ok(a->v_sb) && ok(a->v_sh) && ok(a->v_sl) && (a->o_sb % 4 == 0) && (a->o_sh % 4 == 0) && (a->o_sl % 4 == 0);
  if (bwd)
    r = r && ok(a->do_sb) && ok(a->do_sh) && ok(a->do_sl) && (a->dq_sl % 4 == 0) && (a->dk_sl % 4 == 0) &&
        (a->dv_sl % 4 == 0) && (a->dq_sh % 4 == 0) && (a->dk_sh % 4 == 0) && (a->dv_sh % 4 == 0) &&
        (a->dq_sb % 4 == 0) && (a->dk_sb % 4 == 0) && (a->dv_sb % 4 == 0);
  return r;
}

// which head-dim-72 kernels run on v_mfma_f32_16x16x32_bf16 instead of 32x32x16 (bit 0: dK/dV, bit 1: dQ,
// bit 2: forward).  Knob attn_mfma16 (VDS_ATTN_MFMA16 at load, vds_attn_set_variant afterwards: tests, A/B); default 7 =
// all three: measured +4 % (dK/dV), +3.4 % (forward), +1.4 % (dQ) on random data, same box
constexpr int ATTN_VARIANT_DEFAULT = 7;
int attn_variant() { return vdscfg::geti(vdscfg::ATTN_MFMA16) & 7; }

template <typename K>
void set_lds(K kern, int bytes) {
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

// knob attn_tail_last = 0 keeps the head-major order for ragged lengths too (A/B)
static int tail_last_for(int L, int tile) {
  return (vdscfg::geti(vdscfg::ATTN_TAIL_LAST) && L > tile && (L % tile) != 0) ? 1 : 0;
}

template <int HDP, int HDQ>
int run_fwd(AttnP p, hipStream_t s) {
  constexpr int LDS = 4 * 64 * HDP * 2;
  static bool once = false;
  if (!once) {
    set_lds(attn_fwd_kernel<HDP, HDQ, 2, true>, LDS);
    if constexpr (HDP <= 96) set_lds(attn_fwd_wide_kernel<HDP, HDQ, false>, LDS);
    if constexpr (HDP == 96 && HDQ == 80) set_lds(attn_fwd_wide_kernel<HDP, HDQ, true>, LDS);
    if constexpr (HDP == 96 && HDQ == 80) set_lds(attn_fwd16_kernel<HDP>, LDS);
    once = true;
  }
  const int wide = vdscfg::geti(vdscfg::ATTN_FWD_WIDE);  // 0 / 1 forces (experiments); 2 (default): head_dim 64 / 72, long query sequences
  bool use_wide = false;
  if constexpr (HDP == 96) use_wide = wide == 1 || (wide == 2 && p.Lq >= 2048);
  if constexpr (HDP == 64) use_wide = wide == 1 || (wide == 2 && p.Lq >= 2048);  // +2.7 % at head_dim 64
  p.n_rt = cdiv(p.Lq, use_wide ? 256 : 128);
  p.tail_last = tail_last_for(p.Lq, use_wide ? 256 : 128);
  const int grid = cdiv(p.B * p.H, 8) * 8 * p.n_rt;
  const double fl = 4.0 * p.B * p.H * (double)p.Lq * p.Lk * p.hd;
  const bool ones_fwd = HDP == 96 && HDQ == 80 && use_wide && p.kv_pad_ones && p.hd == 72;
  // (kv_pad_ones == 2: cross-attention on the ones-column kernels -- keeps its own profiler class)
  vdsprof::Scope ps(ones_fwd && p.kv_pad_ones == 1 ? VDS_PROF_ATTN_FWD : VDS_PROF_ATTN_FWD_PLAIN, s, fl,
                    2.0 * p.B * p.H * p.hd * (2.0 * p.Lq + 2.0 * p.Lk));
  if constexpr (HDP == 64) {
    if (use_wide) {
      hipLaunchKernelGGL((attn_fwd_wide_kernel<HDP, HDQ, false>), dim3(grid), dim3(256), LDS, s, p);
      return hipGetLastError() == hipSuccess ? VDS_OK : VDS_ERR_LAUNCH;
    }
  }
  if constexpr (HDP == 96) {
    if (use_wide) {
      bool done = false;
      if constexpr (HDQ == 80) {
        if (p.kv_pad_ones && p.hd == 72) {
          // (round 5 measured a form of this kernel software-pipelined inside the wave: +9.5 % slower, removed in round 6;
          // profiles/r05/attn_fwd_software_pipelined.log)
          if (attn_variant() & 4) hipLaunchKernelGGL((attn_fwd16_kernel<HDP>), dim3(grid), dim3(256), LDS, s, p);
          else hipLaunchKernelGGL((attn_fwd_wide_kernel<HDP, HDQ, true>), dim3(grid), dim3(256), LDS, s, p);
          done = true;
        }
      }
      if (!done) hipLaunchKernelGGL((attn_fwd_wide_kernel<HDP, HDQ, false>), dim3(grid), dim3(256), LDS, s, p);
      return hipGetLastError() == hipSuccess ? VDS_OK : VDS_ERR_LAUNCH;
    }
  }
  hipLaunchKernelGGL((attn_fwd_kernel<HDP, HDQ, 2, true>), dim3(grid), dim3(256), LDS, s, p);
  return hipGetLastError() == hipSuccess ? VDS_OK : VDS_ERR_LAUNCH;
}

// query-range split of the plain dK/dV kernel: target ~3 rounds of the 512 co-resident workgroups, each range at
// least 8 query tiles of 64 (the prologue -- K / V fragments, first tile -- is ~2 tile times)
int dkv_qsplit(int B, int H, int Lq, int Lk) {
  const long wgs = (long)cdiv(B * H, 8) * 8 * cdiv(Lk, 128);
  const int nqt = cdiv(Lq, 64);
  if (wgs >= 3 * 512 || nqt < 16) return 1;
  long sp = (3 * 512 + wgs - 1) / wgs;
  if (sp > nqt / 8) sp = nqt / 8;
  return (int)(sp < 1 ? 1 : sp > 16 ? 16 : sp);
}

template <int HDP, int HDQ>
int run_bwd(AttnP p, hipStream_t s, size_t ws_floats) {
  constexpr int LDS_DQ = 4 * 64 * HDP * 2;
  constexpr int LDS_DKV = 4 * 64 * HDP * 2 + 2 * 128 * 4;
  static bool once = false;
  if (!once) {
    set_lds(attn_bwd_dq_kernel<HDP, HDQ, false>, LDS_DQ);
    if constexpr (HDP == 96 && HDQ == 80) set_lds(attn_bwd_dq_kernel<HDP, HDQ, true>, LDS_DQ);
    set_lds(attn_bwd_dkv_kernel<HDP, HDQ, false>, LDS_DKV);
    if constexpr (HDP == 96 && HDQ == 80) set_lds(attn_bwd_dkv_kernel<HDP, HDQ, true>, LDS_DKV);
    if constexpr (HDP == 96 && HDQ == 80) set_lds(attn_bwd_dkv16_kernel<HDP>, LDS_DKV);
    if constexpr (HDP == 96 && HDQ == 80) set_lds(attn_bwd_dkv16_kernel<HDP, false>, LDS_DKV);
    if constexpr (HDP == 96 && HDQ == 80) set_lds(attn_bwd_dq16_kernel<HDP>, LDS_DQ);
    if constexpr (HDP == 96 && HDQ == 80) set_lds(attn_bwd_dq16_kernel<HDP, true>, LDS_DQ);
    once = true;
  }
  const long rows = (long)p.B * p.H * p.Lq;
  // ALGORITHMIC backward = 2 x forward = 4 products of 2*Lq*Lk*hd (train step = 3 x forward FLOPs, no
  // credit for the recomputed S, nor for the dP the split backward computes twice): dK/dV kernel 2 (dV, dK), dQ
  // kernel 2 (dP, dQ).  The kernels EXECUTE 4 + 3 = 7 products (and head-dim padding on top).
  const double prod = 2.0 * p.B * p.H * (double)p.Lq * p.Lk * p.hd;
  const double qb = 2.0 * p.B * p.H * p.hd * (double)p.Lq, kb = 2.0 * p.B * p.H * p.hd * (double)p.Lk;
  // the 16x16x32 dQ kernel of the ones-column path computes the statistics itself and runs first (attn_bwd_dq16_kernel, FOLD)
  bool fold = false;
  if constexpr (HDP == 96 && HDQ == 80) fold = p.kv_pad_ones && p.hd == 72 && (attn_variant() & 2) && vdscfg::geti(vdscfg::ATTN_DELTA_FOLD);
  auto launch_dq = [&]() {
    p.n_rt = cdiv(p.Lq, 128);
    p.tail_last = tail_last_for(p.Lq, 128);
    const int grid = cdiv(p.B * p.H, 8) * 8 * p.n_rt;
    bool ones = false;
    if constexpr (HDP == 96 && HDQ == 80) ones = p.kv_pad_ones && p.hd == 72;
    vdsprof::Scope ps(ones && p.kv_pad_ones == 1 ? VDS_PROF_ATTN_BWD_DQ : VDS_PROF_ATTN_BWD_DQ_PLAIN, s, 2.0 * prod,
                      (fold ? 4.0 : 3.0) * qb + 2.0 * kb);
    if constexpr (HDP == 96 && HDQ == 80) {
      // (48 queries per wave -- every fragment read feeds 3 MFMAs, two waves per SIMD instead of three -- measured 2-6 %
      // slower in round 5, removed in round 6; profiles/r05/attn_dq_three_blocks_per_wave.log)
      if (fold)
        hipLaunchKernelGGL((attn_bwd_dq16_kernel<HDP, true>), dim3(grid), dim3(256), LDS_DQ, s, p);
      else if (ones && (attn_variant() & 2))
        hipLaunchKernelGGL((attn_bwd_dq16_kernel<HDP>), dim3(grid), dim3(256), LDS_DQ, s, p);
      else if (ones)
        hipLaunchKernelGGL((attn_bwd_dq_kernel<HDP, HDQ, true>), dim3(grid), dim3(256), LDS_DQ, s, p);
    }
    if (!ones) hipLaunchKernelGGL((attn_bwd_dq_kernel<HDP, HDQ, false>), dim3(grid), dim3(256), LDS_DQ, s, p);
  };
  if (fold) launch_dq();
  else {
    vdsprof::Scope ps(VDS_PROF_ATTN_BWD_DELTA, s, 2.0 * rows * p.hd, 2.0 * qb);
    const bool tokmajor = p.o_sh == p.hd && p.do_sh == p.hd && p.o_sl == (long)p.H * p.hd && p.do_sl == p.o_sl &&
                          (p.o_sl & 7) == 0 && (p.o_sb & 7) == 0 && (p.do_sb & 7) == 0 && p.H * p.hd <= 1536 && p.H <= 64;
    if (tokmajor)
      hipLaunchKernelGGL(attn_delta_tokmajor_kernel, dim3((unsigned)(((long)p.B * p.Lq + 3) / 4)), dim3(256), 0, s, p);
    else
      hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, p);
  }
  // (a 64-keys-per-wave, one-wave-per-SIMD variant of this kernel -- half the LDS reads per MFMA -- was
  // measured at 0.55-0.8x: hipcc's single-wave schedule does not overlap the VALU softmax with the MFMAs)
  p.n_rt = cdiv(p.Lk, 128);
  p.tail_last = tail_last_for(p.Lk, 128);
  int grid = cdiv(p.B * p.H, 8) * 8 * p.n_rt;
  {
    bool ones_kv = false;
    // (kv_pad_ones == 2: the q rows carry no annotated pad -- token-major cross-attention queries -- so the dK/dV pass takes
    // the plain kernel, which stages -lse2 itself and masks the K / V pad columns)
    if constexpr (HDP == 96 && HDQ == 80) ones_kv = p.kv_pad_ones == 1 && p.hd == 72;
    vdsprof::Scope ps(ones_kv ? VDS_PROF_ATTN_BWD_DKV : VDS_PROF_ATTN_BWD_DKV_PLAIN, s, 2.0 * prod, 2.0 * qb + 4.0 * kb);
    if constexpr (HDP == 96 && HDQ == 80) {
      // (round 6 measured this kernel as an explicit MFMA / VALU ping-pong -- one 512-thread workgroup per CU, the two wave
      // groups in alternating MFMA and VALU / LDS segments under s_barrier: bit-identical, +5 % slower; the kernel executes 77 %
      // of what a bare MFMA loop reaches at the board's power limit.  profiles/r06/negative_attn_dkv_pingpong.log)
      if (ones_kv && (attn_variant() & 1))
        hipLaunchKernelGGL((attn_bwd_dkv16_kernel<HDP>), dim3(grid), dim3(256), LDS_DKV, s, p);
      else if (ones_kv)
        hipLaunchKernelGGL((attn_bwd_dkv_kernel<HDP, HDQ, true>), dim3(grid), dim3(256), LDS_DKV, s, p);
    }
    bool done16 = false;
    if constexpr (HDP == 96 && HDQ == 80) {
      // cross-attention with padded K / V (kv_pad_ones = 2) and enough workgroups to fill the chip once (B*H*n_rt >= 512;
      // the query-range split below exists in the plain kernel only): the 16x16x32 kernel with S started from -lse2
      const int x16 = vdscfg::geti(vdscfg::CROSS_DKV16);  // 0 = plain kernel, 1 = rule, 2 = also for small grids (tests)
      if (!ones_kv && p.kv_pad_ones == 2 && p.hd == 72 && x16 && ((long)grid >= 512 || x16 == 2) && (attn_variant() & 1)) {
        hipLaunchKernelGGL((attn_bwd_dkv16_kernel<HDP, false>), dim3(grid), dim3(256), LDS_DKV, s, p);
        done16 = true;
      }
    }
    if (!ones_kv && !done16) {
      // few key tiles per head (cross-attention): B*H*n_rt workgroups leave most of the 2 x 256 slots empty (B = 2: 128
      // of 512) or end on a half-filled round (B = 12: 768 = 1.5 rounds).  The query range is then split so that the
      // launch is ~3 rounds of proportionally shorter workgroups; the fp32 partials (q_split x B*H*Lk*2*hd floats, at
      // the end of the caller's workspace) are summed by dkv_reduce_kernel.  Needs the workspace vds_attn_bwd_workspace_
      // bytes asks for (args->ws_floats); knob attn_qsplit = 1 turns it off, = N forces N.
      const int force = vdscfg::geti(vdscfg::ATTN_QSPLIT);
      const int split = force > 0 ? force : dkv_qsplit(p.B, p.H, p.Lq, p.Lk);
      const size_t need = (size_t)2 * rows + (size_t)split * p.B * p.H * p.Lk * 2 * p.hd;
      if (split > 1 && (p.hd & 3) == 0 && ws_floats >= need && (p.dk_sl & 3) == 0) {
        p.q_split = split;
        p.dkv_part = p.delta + 2 * rows;
        p.tail_last = 0;
        const int g2 = cdiv(p.B * p.H, 8) * 8 * p.n_rt * split;
        hipLaunchKernelGGL((attn_bwd_dkv_kernel<HDP, HDQ, false>), dim3(g2), dim3(256), LDS_DKV, s, p);
        const long n = (long)p.B * p.H * p.Lk * 2 * (p.hd >> 2);
        hipLaunchKernelGGL(dkv_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p);
        p.q_split = 1;
        p.dkv_part = nullptr;
      } else {
        hipLaunchKernelGGL((attn_bwd_dkv_kernel<HDP, HDQ, false>), dim3(grid), dim3(256), LDS_DKV, s, p);
      }
    }
  }
  if (!fold) launch_dq();
  return hipGetLastError() == hipSuccess ? VDS_OK : VDS_ERR_LAUNCH;
}

}  // namespace

// K / V of a cross-attention (token-major [B*Lk, ld] rows, head h at columns col0 + h*hd) -> head-major padded rows
// [B,H,Lk,hdp] WITH the ones columns of the head_dim-72 kernels (K: 1.0 at hd, hd+1; V: 1.0 at hd, hd+4; zeros elsewhere
// in the pad), so that the forward and dQ passes of the cross-attention run on the ones-column 16x16x32 kernels
// (vds_attn_args.kv_pad_ones = 2).  One thread per 16-byte chunk of an output row.
__global__ __launch_bounds__(256) void kv_pad_ones_kernel(const bf16_t* kv, long ld, int kcol0, int vcol0, bf16_t* kp,
                                                          bf16_t* vp, int B, int Lk, int H, int hd, int hdp) {
  const int cpr = hdp >> 3;  // chunks per output row
  const long n = (long)B * H * Lk * cpr;
  const long gid = (long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= n) return;
  const int which = blockIdx.y;  // 0 K, 1 V
  const int c = (int)(gid % cpr);
  const long row = gid / cpr;  // (b, h, l)
  const int l = (int)(row % Lk);
  const int h = (int)((row / Lk) % H);
  const int b = (int)(row / ((long)Lk * H));
  u32x4 w = {0u, 0u, 0u, 0u};
  if (c * 8 < hd) {
    w = *reinterpret_cast<const u32x4*>(kv + ((long)b * Lk + l) * ld + (which ? vcol0 : kcol0) + h * hd + c * 8);
  } else if (c * 8 == hd) {
    w[0] = which ? 0x00003f80u : 0x3f803f80u;  // K: 1.0 at hd, hd+1; V: 1.0 at hd ...
    w[2] = which ? 0x00003f80u : 0u;           // ... and hd+4
  }
  *reinterpret_cast<u32x4*>((which ? vp : kp) + row * hdp + c * 8) = w;
}

extern "C" int vds_kv_pad_ones(const void* kv, int64_t ld, int32_t k_col0, int32_t v_col0, void* kp, void* vp, int32_t B,
                               int32_t Lk, int32_t H, int32_t hd, int32_t hdp, vds_stream_t stream) {
  if (!kv || !kp || !vp || B <= 0 || Lk <= 0 || H <= 0 || hd <= 0) return VDS_ERR_ARG;
  if ((hd & 7) || (hdp & 7) || hdp < hd + 8 || (ld & 7) || (k_col0 & 7) || (v_col0 & 7)) return VDS_ERR_ARG;
  const long n = (long)B * H * Lk * (hdp >> 3);
  hipLaunchKernelGGL(kv_pad_ones_kernel, dim3((unsigned)((n + 255) / 256), 2), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)kv, (long)ld, k_col0, v_col0, (bf16_t*)kp, (bf16_t*)vp, B, Lk, H, hd, hdp);
  return hipGetLastError() == hipSuccess ? VDS_OK : VDS_ERR_LAUNCH;
}

extern "C" int vds_attn_set_variant(int32_t mask) {
  const int prev = attn_variant();
  if (mask < -1 || mask > 7) return VDS_ERR_ARG;
  vdscfg::g_val[vdscfg::ATTN_MFMA16] = mask < 0 ? ATTN_VARIANT_DEFAULT : mask;  // -1: back to the default
  return prev;
}

extern "C" int vds_attn_fwd(const vds_attn_args* a, vds_stream_t stream) {
  if (!a || !a->q || !a->k || !a->v || !a->o || !a->lse || a->Lq <= 0 || a->Lk <= 0) return VDS_ERR_ARG;
  if (!strides_ok(a, false)) return VDS_ERR_ARG;
  AttnP p = to_p(a);
  hipStream_t s = (hipStream_t)stream;
  switch (kernel_instance(a->head_dim)) {
    case 32: return run_fwd<32, 32>(p, s);  // (the reference's own smoke test: width 512 / 16 heads, model.py:545-565)
    case 64: return run_fwd<64, 64>(p, s);
    case 80: return run_fwd<96, 80>(p, s);
    case 96: return run_fwd<96, 96>(p, s);
    case 128: return run_fwd<128, 128>(p, s);
    default: return VDS_ERR_UNSUPPORTED;
  }
}

extern "C" size_t vds_attn_bwd_workspace_bytes(const vds_attn_args* a) {
  if (!a || a->B <= 0 || a->H <= 0 || a->Lq <= 0) return 0;
  size_t fl = (size_t)2 * a->B * a->H * a->Lq;
  if (a->Lk > 0 && a->head_dim > 0 && a->kv_pad_ones != 1)  // fp32 partials of the query-split dK/dV kernel (short key sequences)
    fl += (size_t)dkv_qsplit(a->B, a->H, a->Lq, a->Lk) * a->B * a->H * a->Lk * 2 * a->head_dim;
  return fl * sizeof(float);
}

extern "C" int vds_attn_bwd(const vds_attn_args* a, vds_stream_t stream) {
  if (!a || !a->q || !a->k || !a->v || !a->o || !a->lse || !a->d_o || !a->dq || !a->dk || !a->dv || !a->delta)
    return VDS_ERR_ARG;
  if (!strides_ok(a, true)) return VDS_ERR_ARG;
  AttnP p = to_p(a);
  hipStream_t s = (hipStream_t)stream;
  // floats in `delta`: 2*B*H*Lq when the caller does not say (ws_floats = 0: the pre-round-4 contract, no query split)
  const size_t ws = a->ws_floats > 0 ? (size_t)a->ws_floats : (size_t)2 * a->B * a->H * a->Lq;
  switch (kernel_instance(a->head_dim)) {
    case 32: return run_bwd<32, 32>(p, s, ws);
    case 64: return run_bwd<64, 64>(p, s, ws);
    case 80: return run_bwd<96, 80>(p, s, ws);
    case 96: return run_bwd<96, 96>(p, s, ws);
    case 128: return run_bwd<128, 128>(p, s, ws);
    default: return VDS_ERR_UNSUPPORTED;
  }
}
