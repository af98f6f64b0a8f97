// OCP-fp8 flash attention for gfx950 on v_mfma_f32_16x16x128_f8f6f4 (BASELINE config 5: "fp8 weights + activations
// on CDNA4 fp8 MFMA"): forward, dK/dV and dQ of F.scaled_dot_product_attention (model.py:136) with e4m3 Q / K / V / P
// and e5m2 dO / dS, fp32 accumulation, fp32 softmax statistics (running max, LSE, delta) exactly as in the bf16
// kernels of attention.hip.  The reference trains in bf16 only; the recipe is this build's (fp8.py states it).
//
// Why the shapes below: the fp8 MFMA contracts 128 elements in 32 cycles (bf16: 32 elements in 16), so
//   * a QK^T-type product covers the whole head dimension (72 data bytes of a 128-byte row) in ONE instruction
//     (bf16: three 16x16x32, 48 cycles) -- no 72 -> 96 padding pressure, the row pad is free;
//   * a PV-type product (contraction over keys / queries) runs at 2x the bf16 rate when it contracts 128 rows at
//     once, so every streamed tile is 128 rows x 128 B = 16 KiB and P / dS are packed 32 bytes per lane.
// Data flow (same product forms as attention.hip: softmax statistics stay on the lane that owns the column):
//   forward   S^T = K Q^T        A = 16 key rows from LDS (2 x ds_read_b128), B = 16 queries in registers
//             O^T += V^T P^T     A = V^T by ds_read_b64_tr_b8 (8 keys x 16 columns per 16-lane group), B = the eight
//                                16-key S^T accumulators of a tile, exponentiated and packed in place: lane (c, g)
//                                ends up with keys 32g .. 32g+31 of query c = exactly its B-operand bytes
//   dQ        S^T, dP^T = V dO^T;  dQ^T += K^T dS^T   (K^T by transposing reads of the same K tile)
//   dK, dV    S = Q K^T, dP = dO V^T (key on the lane);  dV^T += dO^T P,  dK^T += Q^T dS
// so that a 16-row block i of a tile uses the tile rows 32 (m >> 2) + 4 i + (m & 3) as its MFMA rows m = 0..15:
// accumulator register r of lane (c, g) is then tile row 32 g + 4 i + r, and the eight blocks fill the lane's 32
// contraction bytes in natural order.
// Row constants and scale factors ride in for free.  The producer quantises q with a factor chosen so that
// s_q s_k / sqrt(hd) * log2(e) is a power of two, 2^-E (its top binade is still fully used); every S-type product is
// the block-scaled form of the instruction (v_mfma_scale_f32_16x16x128_f8f6f4, one E8M0 byte per operand: same issue
// rate, tools/probe_mfma_scale.hip) with scale 2^-E, so its accumulator IS the log2-domain exponent, and it starts
// from C = the row constant (8 - lse2 backward; 8 (8 - m) + bias per query forward): no multiply-add per element.
// dP accumulates onto C = -delta with scale 2^-8, which keeps dS = 256 P * dP / 256 below the e5m2 maximum for ANY
// data (|dO_q| <= 2^-4, |V_q| <= 448: |dS_q| <= 2 * 72 * 28 = 4032) -- no clamp.  P is kept as P * 2^8 (e4m3 tops out at 448): forward <= 2^8.8 relative
// to the lazily raised running maximum (and rounded to the e4m3 grid in the log domain: P_BYTE in the forward kernel),
// backward = 256 * the true probability (covers 2^-17 .. 1).
// LDS image: row-major 128-B rows, 16-B chunk c of row r stored at chunk c ^ swz8(r), swz8(r) = ((r >> 1) & 3) |
// ((r >> 3) & 4): conflict-free for the row reads (16 lanes x 16 B over 256 B) and for the transposing reads
// (2 x 8 rows x 16 B per 32-lane half); applied to the per-lane SOURCE address of the LDS-DMA.
#include "common.h"
#include "rope_stage.h"
#include <type_traits>
#include "prof.h"
#include "config.h"
#include "../../include/vds.h"
#include <cstdlib>

namespace {

typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(2))) int i32x2_t;
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;
constexpr int ROWB = 128;            // bytes of an fp8 row (global memory and LDS)
constexpr int TILE = 128 * ROWB;     // one streamed tile: 128 rows
constexpr float P_SHIFT = 8.0f;      // P is held as P * 2^8
constexpr float LOGDOM_BIAS = 56.0f - 0.344f;  // forward: e4m3 byte of 2^x = round(8 x + LOGDOM_BIAS) (attn8_fwd_kernel, P_BYTE)
constexpr float BYTE_LIMIT = 126.4f;  // forward: raise the running maximum when a byte would exceed 0x7E (448; 0x7F is NaN)
constexpr float SEED_HEADROOM = 2.0f; // forward: the first block's maximum + 2 seeds the running maximum (round 4: was 4)
// Backward (round 4): P = 2^x without v_exp_f32 (a quarter-rate instruction; the fp8 backward kernels are bound by the
// VALU work between their MFMAs, profiles/r04_attn_fp8_b6_sq_counters.txt).  The S-type MFMA delivers
// t = (x + 127 - PEXP_C) * 2^23 directly -- its block scale carries the 2^23, the per-query start value of the
// accumulator the (127 - PEXP_C + 8 - lse2) * 2^23 -- and ONE v_cvt_u32_f32 turns t into the bit pattern of the float
// 2^floor(x) * (1 + frac x): 2^x interpolated linearly between powers of two (Schraudolph).  (1 + f) / 2^f lies in
// [1, 1.0615]; PEXP_C = 0.043 centres it: P is within a factor 2^(+-0.043) (+- 3 %) of the exact value, the same
// approximation the forward kernel's log-domain byte makes, and small beside the e4m3 (P) / e5m2 (dS) roundings that
// follow (relative steps 2^-3 / 2^-2).  t < 0 (x < -127) converts to 0; the accumulator's resolution at |t| ~ 2^30 is
// 2^7, i.e. 2^-16 in x.  Building with -DVDS_ATTN8_EXACT_EXP=1 keeps v_exp_f32 (A/B builds; a run-time switch inside the
// loops costs registers: 412 bytes of scratch per lane in the dK/dV kernel).
#ifndef VDS_ATTN8_EXACT_EXP
#define VDS_ATTN8_EXACT_EXP 0
#endif
constexpr bool EXACT_EXP = VDS_ATTN8_EXACT_EXP != 0;
constexpr float PEXP_C = 0.043f;
constexpr float PEXP_ONE = 8388608.0f;  // 2^23

struct Attn8P {
  int B, H, Lq, Lk, hd;
  const unsigned char *q, *k, *v, *d_o;  // [B,H,L,128] fp8 rows: bytes [0,hd) data, V byte hd = 1.0, the rest 0
  bf16_t* o; long o_sb, o_sh, o_sl;
  float* lse;                            // [B,H,Lq]
  bf16_t* dq; long dq_sb, dq_sh, dq_sl;
  bf16_t* dk; long dk_sb, dk_sh, dk_sl;
  bf16_t* dv; long dv_sb, dv_sh, dv_sl;
  const float* stats;                    // backward: [2][B,H,Lq]: -delta / (256 s_do s_v), then 8 - lse log2(e) as (. + 127 - PEXP_C) 2^23
  const float* deq;                      // {s_q, s_k, s_v, s_do, E}: x = x_q * s; s_q s_k log2(e) / sqrt(hd) = 2^-E
  float scale;
  int n_rt;
  int tail_last;  // decode_block: partly filled last tiles after all full ones
  // round 4: the forward output / the query gradient additionally (or, dQ: instead) as fp8, token-major [B*Lq, H*hd]
  // bytes (the layout of the following linear layer's operand), scaled with the previous step's amax (saturating),
  // current amax recorded, dequantisation factor written -- the separate quantisation pass over the bf16 tensor goes
  unsigned char* oq; long oq_ld;    // forward: e4m3 copy of O (row stride oq_ld bytes); null = none
  unsigned char* dqq; long dqq_ld;  // backward: e5m2 copy of dQ; null = none
  const float* e_amax_prev; float* e_amax_cur; float* e_dq_out;
  int wide_dkv;  // dK / dV rows 16-byte aligned: store_block_bf16_t's wide path
  int wide_o, wide_dq;  // same for the bf16 O / dQ rows (store_block_t)
};

__device__ __forceinline__ int swz8(int row) { return ((row >> 1) & 3) | ((row >> 3) & 4); }
// t = (x + 127 - PEXP_C) * 2^23 -> the float 2^x (piecewise linear, see PEXP_C).  A plain conversion (one
// v_cvt_u32_f32, which saturates: t < 0, i.e. x < -127, gives 0.0f), NOT inline asm: t is an MFMA result, and the
// compiler only inserts the wait states between an MFMA and a VALU read of its result for instructions it can see
// (an asm statement here returned garbage at random).
__device__ __forceinline__ float pexp_bits(float t) { return __builtin_bit_cast(float, (unsigned)t); }

// Per-lane LDS byte offsets of the fragment reads, relative to a tile's base: computed once per kernel so that every
// read in the tile loops is `base VGPR + immediate` (tile / buffer / block offsets are compile-time constants).
//   row form  (A operand of a QK^T-type product): MFMA row m of block i = tile row 32 (m >> 2) + 4 i + (m & 3); lane
//     (m, g) takes bytes 32g .. 32g+31 = chunks 2g, 2g+1 at slots (2g + h) ^ swz8(row); swz8(row) = s0 ^ 2 (i & 1)
//     with s0 = ((m >> 1) & 1) | (((m >> 2) & 1) << 2): odd blocks flip bit 5 of the even blocks' byte offset.
//   transposed form (A operand of a PV-type product): (T^T)[16 db + d][k], lane (d = l & 15, g) takes tile rows
//     k = 32g .. 32g+31 of column 16 db + d: four ds_read_b64_tr_b8, each an 8-row x 16-byte block per 16-lane group
//     (lane 2q + p supplies the address of row q, bytes 8p .. 8p+7; lane d receives column d of the 8 rows, row q in
//     byte q -- tools/probe_tr8.hip); swz8(32g + 8t + q) = ((q >> 1) & 3) | ((g & 1) << 2) does not depend on t.
struct Frag8 {
  unsigned rowoff[2][2];  // [block parity][chunk 2g / 2g+1]
  unsigned tr[5];         // column block db, t = 0
  __device__ __forceinline__ void init(int lane) {
    const int m = lane & 15, g = lane >> 4;
    const int s0 = ((m >> 1) & 1) | (((m >> 2) & 1) << 2);
    const unsigned e0 = (unsigned)((32 * (m >> 2) + (m & 3)) * ROWB + (((2 * g) ^ s0) << 4));
    rowoff[0][0] = e0;
    rowoff[0][1] = e0 ^ 16u;
    rowoff[1][0] = e0 ^ 32u;
    rowoff[1][1] = e0 ^ 48u;
    const int q = m >> 1, pp = m & 1;
    const int st = ((q >> 1) & 3) | ((g & 1) << 2);
#pragma unroll
    for (int db = 0; db < 5; ++db) tr[db] = (unsigned)((32 * g + q) * ROWB + ((db ^ st) << 4) + 8 * pp);
  }
  template <int I>
  __device__ __forceinline__ i32x8 row(const char* tile) const {
    const i32x4 lo = *reinterpret_cast<const i32x4*>(tile + rowoff[I & 1][0] + I * 4 * ROWB);
    const i32x4 hi = *reinterpret_cast<const i32x4*>(tile + rowoff[I & 1][1] + I * 4 * ROWB);
    return i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  }
  template <int DB>
  __device__ __forceinline__ i32x8 trans(const char* tile) const {
    i32x8 r;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const char* p = tile + tr[DB] + t * 8 * ROWB;
      const i32x2_t w = __builtin_amdgcn_ds_read_tr8_b64_v2i32((__attribute__((address_space(3))) i32x2_t*)LDS_PTR(p));
      r[2 * t] = w[0];
      r[2 * t + 1] = w[1];
    }
    return r;
  }
};
// cbsz / blgp: format of A / B (0 e4m3, 1 e5m2); the block scales are unused (0 selects the unscaled instruction)
template <int FA, int FB>
__device__ __forceinline__ f32x4 mfma8(const i32x8& a, const i32x8& b, f32x4 c) {
  return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, FA, FB, 0, 0, 0, 0);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (N > 0) {
    static_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}
// block-scaled form: D = C + 2^(sa - 127) 2^(sb - 127) A B (byte 0 of the scale registers, the same for every lane)
template <int FA, int FB>
__device__ __forceinline__ f32x4 mfma8s(const i32x8& a, const i32x8& b, f32x4 c, int sa, int sb) {
  return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, FA, FB, 0, sa, 0, sb);
}
__device__ __forceinline__ f32x4 zero4() { return f32x4{0.f, 0.f, 0.f, 0.f}; }
// (the first convert replaces the low half of `w`, the second the high half: `w` needs no initial value -- the empty
// asm "defines" it without an instruction, which saves the v_mov 0 a zero-initialised temporary costs per dword)
__device__ __forceinline__ int cvt4_e4m3(const f32x4& x) {
  int w;
  asm volatile("" : "=v"(w));
  w = __builtin_amdgcn_cvt_pk_fp8_f32(x[0], x[1], w, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(x[2], x[3], w, true);
  return w;
}
__device__ __forceinline__ int cvt4_e5m2(const f32x4& x) {
  int w;
  asm volatile("" : "=v"(w));
  w = __builtin_amdgcn_cvt_pk_bf8_f32(x[0], x[1], w, false);
  w = __builtin_amdgcn_cvt_pk_bf8_f32(x[2], x[3], w, true);
  return w;
}

// LDS-DMA staging of a 128-row tile (16 pieces of 1 KiB, 4 per wave): lane i of piece pc fills LDS bytes
// pc*1024 + 16 i = slot (i & 7) of row 8 pc + (i >> 3), so it fetches chunk slot ^ swz8(row) of that row.
struct Stage8 {
  unsigned voff[4];
  __device__ __forceinline__ void init(int wave, int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int pc = wave * 4 + i;
      const int row = 8 * pc + (lane >> 3);
      const int ch = (lane & 7) ^ swz8(row);
      voff[i] = (unsigned)(row * ROWB + ch * 16);
    }
  }
  // row0_bytes = first row of the tile * 128; rows past the tensor read as zero (SRD bounds)
  __device__ __forceinline__ void issue(srd_t rs, char* tile, unsigned row0_bytes, int wave) const {
    const unsigned base = lds_addr_of(tile) + wave * 4096;
#pragma unroll
    for (int i = 0; i < 4; ++i) lds_dma16(rs, base + i * 1024, voff[i] + row0_bytes);
  }
};

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
__device__ __forceinline__ float max_over_lane_groups(float x) {  // max over the 4 lanes (l & 15) + 16 g
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  x = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ i32x8 load_row32(__amdgpu_buffer_rsrc_t rs, long row, int g) {
  const unsigned off = (unsigned)(row * ROWB + 32 * g);
  const i32x4 lo = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
  const i32x4 hi = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off + 16, 0, 0));
  return i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
__device__ __forceinline__ void retire(const i32x8& f) { asm volatile("" ::"v"(f)); }
__device__ __forceinline__ void retire(float f) { asm volatile("" ::"v"(f)); }

// ===================================== forward ==============================================
// Workgroup = 256 queries (4 waves x 4 column blocks of 16); K / V tiles of 128 keys, double-buffered, one barrier
// per tile.  Lazy online softmax per tile: the exponent x = score * cs + (8 - m) of every key of the tile is packed
// as its e4m3 byte; when some byte of the wave would exceed 0x7E (wave-uniform, rare after the first tiles) the
// running maxima m are raised to the tile's, O (whose row `hd` is the softmax denominator, from V's ones column) is
// rescaled, and the tile's P is recomputed -- it has not entered O yet.
// Store one 16-row block of a TRANSPOSED result tile (O^T, dQ^T: lane = (g, r) holds, for tile row r = lane & 15, the
// columns 16 db + 4 g + 0..3 in v[db][0..3]) as a bf16 row (`row`, may be null) and / or as fp8 bytes (`row8`, may be
// null; values = the bf16-rounded results * alpha, saturated).  The four g-lanes of a row first exchange their words
// (row_transpose4) so that each lane owns 16 consecutive columns: 16-byte stores, a row's 64 / 128 bytes contiguous
// per instruction, instead of 4- / 8-byte stores 16 / 32 bytes apart.  Returns max |bf16 value| of the valid rows.
// All 64 lanes must call (cross-lane exchange); `valid` is uniform over the four lanes of a row.
template <int HD, int EF, int NDB>
__device__ __forceinline__ float store_block_t(const float (&v)[NDB][4], bf16_t* row, unsigned char* row8, bool emit8,
                                               float alpha, int g, bool valid, bool wide = true) {
  static_assert(NDB == 5 && HD >= 64 && HD <= 80, "four full 16-column blocks and one partial");
  constexpr float FMAX = EF == 0 ? 448.0f : 57344.0f;
  unsigned lo[4], hi[4], q8[4] = {0u, 0u, 0u, 0u}, tlo, thi, t8 = 0u;
  float emax = 0.f;
#pragma unroll
  for (int db = 0; db < NDB; ++db) {
    const unsigned a = pack_bf2(v[db][0], v[db][1]), b = pack_bf2(v[db][2], v[db][3]);
    unsigned w8 = 0u;
    if (emit8) {
      float f[4] = {bflo(a), bfhi(a), bflo(b), bfhi(b)};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (db < 4 || 4 * g + e < HD - 64) emax = fmaxf(emax, fabsf(f[e]));
        f[e] = __builtin_amdgcn_fmed3f(f[e] * alpha, -FMAX, FMAX);
      }
      w8 = fp8_cvt4<EF>(f[0], f[1], f[2], f[3]);
    }
    if (db < 4) { lo[db] = a; hi[db] = b; q8[db] = w8; }
    else { tlo = a; thi = b; t8 = w8; }
  }
  // `wide` (wave-uniform): the bf16 rows are 16-byte aligned (strides in multiples of 8 elements, 16-byte aligned base).
  // Otherwise (8-byte aligned rows: the contract before round 4) the bf16 words leave as 8-byte pieces, untransposed.
  if (row && !wide && valid) {
#pragma unroll
    for (int db = 0; db < 4; ++db) *reinterpret_cast<u32x2*>(row + db * 16 + 4 * g) = u32x2{lo[db], hi[db]};
  }
  row_transpose4(lo);
  row_transpose4(hi);
  if (emit8) row_transpose4(q8);
  if (!valid) return 0.f;
  const bool tail = 4 * g < HD - 64;
  if (row && !wide) {
    if (tail) *reinterpret_cast<u32x2*>(row + 64 + 4 * g) = u32x2{tlo, thi};
  } else if (row) {
    *reinterpret_cast<u32x4*>(row + 16 * g) = u32x4{lo[0], hi[0], lo[1], hi[1]};
    *reinterpret_cast<u32x4*>(row + 16 * g + 8) = u32x4{lo[2], hi[2], lo[3], hi[3]};
    if (tail) *reinterpret_cast<u32x2*>(row + 64 + 4 * g) = u32x2{tlo, thi};
  }
  if (emit8) {
    *reinterpret_cast<u32x4*>(row8 + 16 * g) = u32x4{q8[0], q8[1], q8[2], q8[3]};
    if (tail) *reinterpret_cast<unsigned*>(row8 + 64 + 4 * g) = t8;
  }
  return emax;
}

template <int HD>
__global__ __launch_bounds__(256, 2) void attn8_fwd_kernel(Attn8P p) {
  static_assert(HD == 72, "ones column of V at byte 72: row 72 of O^T = block 4, lanes g = 2, register 0");
  constexpr int NDB = 5;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int bh, qt;
  if (!decode_block(p.n_rt, p.tail_last, p.B * p.H, bh, qt)) return;
  const int b = bh / p.H, hh = bh % p.H;
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qrow0 = qt * 256 + wave * 64 + (lane & 15);
  const long head = (long)b * p.H + hh;

  const __amdgpu_buffer_rsrc_t rq = make_rsrc(p.q + head * p.Lq * ROWB, (unsigned)(p.Lq * ROWB));
  const srd_t rk = make_srd(p.k + head * p.Lk * ROWB, (unsigned)(p.Lk * ROWB));
  const srd_t rv = make_srd(p.v + head * p.Lk * ROWB, (unsigned)(p.Lk * ROWB));
  Stage8 st;
  st.init(wave, lane);
  st.issue(rk, smem, 0, wave);
  st.issue(rv, smem + TILE, 0, wave);

  i32x8 qf[4];
#pragma unroll
  for (int cb = 0; cb < 4; ++cb) qf[cb] = load_row32(rq, qrow0 + 16 * cb, g);
  // S-type products come out as 8 x (log2-domain score): scale 2^(3 - E) (see the file header)
  const int sc_t = __builtin_amdgcn_readfirstlane(127 + 3 - (int)p.deq[4]);
  const float sv = p.deq[2];
  Frag8 fr;
  fr.init(lane);

  f32x4 o[NDB][4];
#pragma unroll
  for (int db = 0; db < NDB; ++db)
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) o[db][cb] = zero4();
  float m[4];  // running maximum (log2 domain), seeded below
  const int nkt = (p.Lk + 127) / 128;
#pragma unroll
  for (int cb = 0; cb < 4; ++cb) retire(qf[cb]);
  retire(sv);
  VDS_WAIT_VM(0);
  __syncthreads();

  // seed the running maxima from MFMA block 0 of the first tile -- tile rows 0-3, 32-35, 64-67, 96-99 (the rows
  // `fr.row<0>` addresses; keys 0-3 are register tokens, the typical attention sinks) -- plus SEED_HEADROOM: the largest
  // P of the first tiles is then ~2^(8 - headroom), and the maximum is raised (and a tile recomputed) only when a later
  // score exceeds the seed by more than headroom + 0.84 in the log2 domain.
  // SUPPORTED DYNAMIC RANGE of a row: P is held as P * 2^8 relative to the running maximum m and leaves the log-domain
  // pack as byte 0 below 2^-6.9, i.e. a key whose score lies more than 14.9 binades (10.3 nat) under m contributes
  // nothing to O or to the LSE, and between 14 and 14.9 binades under m (bytes 1-7, e4m3's subnormals, where the
  // log-domain byte is no longer the value's encoding) it is under-weighted by up to 30 %.  m is at most SEED_HEADROOM
  // = 2 binades above the row's true maximum (when block 0 holds it and nothing ever raises m), so every key within 12
  // binades (8.3 nat, a factor 4096) of the row maximum is represented with e4m3's relative precision; for comparison, e4m3 P relative to the row maximum itself (the usual fp8 flash-attention
  // form) bottoms out at 9 binades.  (Round 3 used 4 binades of headroom -- fewer raises on random data, 10.9 binades
  // of range; with the register tokens as sinks a diffuse 8k-key tail 11-12 binades under them is realistic.)
  // tests/test_attn_fp8_gpu.py::test_attn_fp8_forward_sink_plus_diffuse_tail pins the in-range behaviour at the
  // headline length (sinks 2^9 and 2^12 above an 8k-key tail) and records the error outside it.
  {
    const i32x8 kf = fr.row<0>(smem);
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
      const f32x4 s0 = mfma8s<0, 0>(kf, qf[cb], zero4(), sc_t, 127);
      float t = -INFINITY;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (32 * g + r < p.Lk) t = fmaxf(t, s0[r]);
      m[cb] = max_over_lane_groups(t) * 0.125f + SEED_HEADROOM;
    }
  }

  // P_BYTE: the forward pass is bound by the VALU work between its MFMAs (exp2 alone is a quarter-rate instruction), so
  // P = 2^x is rounded to e4m3 in the LOG domain: an e4m3 byte b >= 8 is the value 2^((b >> 3) - 7) (1 + (b & 7) / 8),
  // i.e. b is a piecewise-linear function of log2(value) that deviates from 8 x + 56 by g = 8 (2^f - 1 - f) in
  // [-0.688, 0] (f = frac x).  byte = round(8 x + 56 - 0.344) -- one v_fma_f32 (shared with the score scaling) and one
  // v_cvt_pk_u8_f32 (round to nearest even, saturating at 0, written straight into its byte) per element instead of
  // fma + v_exp_f32 + convert (and the fma is folded into the block-scaled MFMA) -- is the value 2^x within a factor 2^(+-0.043) (+- 3 %) before the rounding to the byte
  // grid that every e4m3 cast has (+- 0.5 step): 0.35 instead of 0.29 steps rms.  The softmax denominator is the sum of
  // the SAME bytes (V's ones column), so the weights of a row still sum to one exactly.  Below 2^-6 (bytes < 8, e4m3
  // subnormals: 14.8 binades under the row maximum) the byte grid is linear and the value is under-estimated.
  // scores -> packed P of one tile with the current maxima; mt: largest byte value per query block.  Software-pipelined by
  // hand: the S products of block i+1 are issued before the softmax VALU work of block i, and the sched_barrier keeps
  // hipcc from hoisting all 32 independent MFMAs (128 live accumulator registers).  RAGGED (last tile only): keys past
  // Lk -- zero rows -- are kept out of the maximum and of P.
  auto s_phase = [&](const char* kt, i32x8 (&pq)[4], float (&mt)[4], int key_lim, auto RAG) {
    constexpr bool ragged = decltype(RAG)::value;
    // the accumulators start at the query's constant 8 (8 - m) + bias: t = 8 * exponent + 56 - 0.344 leaves the MFMA
    f32x4 nm[4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
      const float c0 = 8.0f * (P_SHIFT - m[cb]) + LOGDOM_BIAS;
      nm[cb] = f32x4{c0, c0, c0, c0};
      mt[cb] = -INFINITY;
    }
    f32x4 xn[4];
    i32x8 kfn;  // K fragment of the block whose products are issued in the next iteration (read one iteration ahead)
    {
      const i32x8 kf = fr.row<0>(kt);
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) xn[cb] = mfma8s<0, 0>(kf, qf[cb], nm[cb], sc_t, 127);
      kfn = fr.row<1>(kt);
    }
    static_for<8>([&](auto I) {
      constexpr int i = decltype(I)::value;
      f32x4 x[4];
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) x[cb] = xn[cb];
      if constexpr (i < 7) {
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) xn[cb] = mfma8s<0, 0>(kfn, qf[cb], nm[cb], sc_t, 127);
      }
      if constexpr (i < 6) kfn = fr.row<i + 2>(kt);
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) {
        // x = the e4m3 byte of 2^exponent before rounding (log-domain rounding: see P_BYTE above)
        if constexpr (ragged) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (4 * i + r >= key_lim) x[cb][r] = -INFINITY;
        }
        mt[cb] = fmaxf(fmaxf(mt[cb], x[cb][0]), x[cb][1]);  // v_max3_f32
        mt[cb] = fmaxf(fmaxf(mt[cb], x[cb][2]), x[cb][3]);
        unsigned w;
        asm volatile("" : "=v"(w));  // (every byte is written below: no initial value, no v_mov)
#pragma unroll
        for (int r = 0; r < 4; ++r) w = __builtin_amdgcn_cvt_pk_u8_f32(x[cb][r], r, w);
        asm volatile("" : "+v"(w));  // pins the pack here: LLVM otherwise sinks it below the (rare) raise-the-maximum
                                     // branch, which keeps all 128 exponents of the tile live
        pq[cb][i] = (int)w;
      }
      __builtin_amdgcn_sched_barrier(0);
    });
  };

  auto kv_tile = [&](int j, auto PAR, auto RAG) {
    constexpr int par = decltype(PAR)::value;
    if (j + 1 < nkt) {
      char* nk = smem + (par ^ 1) * 2 * TILE;
      st.issue(rk, nk, (unsigned)(j + 1) * TILE, wave);
      st.issue(rv, nk + TILE, (unsigned)(j + 1) * TILE, wave);
    }
    const char* kt = smem + par * 2 * TILE;
    const char* vt = kt + TILE;
    const int key_lim = p.Lk - j * 128 - 32 * g;  // keys of this lane's 32-byte range that exist: [0, key_lim)
    i32x8 pq[4];
    float mt[4];
    s_phase(kt, pq, mt, key_lim, RAG);
    const float mxa = fmaxf(fmaxf(mt[0], mt[1]), fmaxf(mt[2], mt[3]));
    if (__builtin_amdgcn_ballot_w64(mxa > BYTE_LIMIT) != 0) {  // wave-uniform, rare after the first tiles
      asm volatile("; raise the running maxima to this tile's, rescale O, recompute the tile's P" ::: "memory");
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) {
        // largest score of the tile, log2 domain: byte value -> exponent -> score
        const float t = (max_over_lane_groups(mt[cb]) - LOGDOM_BIAS) * 0.125f - (P_SHIFT - m[cb]);  // (mt includes 8 (8 - m) + bias)
        if (t > m[cb]) {  // (the same decision in the 4 lanes of a query)
          const float alpha = __builtin_amdgcn_exp2f(m[cb] - t);
          m[cb] = t;
#pragma unroll
          for (int db = 0; db < NDB; ++db) o[db][cb] *= alpha;
        }
      }
      s_phase(kt, pq, mt, key_lim, RAG);  // every exponent of the tile is now <= 8 (byte <= 120)
    }
    // O^T += V^T P^T: the transposed V fragments are double-buffered by hand -- block db+1 is read while the four
    // products of block db run (hipcc otherwise re-uses one register set and exposes an LDS round trip per block)
    i32x8 vfa = fr.trans<0>(vt), vfb;
    static_for<NDB>([&](auto DB) {
      constexpr int db = decltype(DB)::value;
      if constexpr (db + 1 < NDB) {
        if constexpr (db & 1) vfa = fr.trans<db + 1>(vt);
        else vfb = fr.trans<db + 1>(vt);
      }
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) o[db][cb] = mfma8<0, 0>((db & 1) ? vfb : vfa, pq[cb], o[db][cb]);
      __builtin_amdgcn_sched_barrier(0);
    });
    VDS_WAIT_VM(0);
    __syncthreads();
  };
  const bool last_ragged = (p.Lk & 127) != 0;
  const int n_full = last_ragged ? nkt - 1 : nkt;
  int j = 0;
  for (; j + 1 < n_full; j += 2) {
    kv_tile(j, std::integral_constant<int, 0>{}, std::false_type{});
    kv_tile(j + 1, std::integral_constant<int, 1>{}, std::false_type{});
  }
  if (j < n_full) {
    kv_tile(j, std::integral_constant<int, 0>{}, std::false_type{});
    if (last_ragged) kv_tile(j + 1, std::integral_constant<int, 1>{}, std::true_type{});
  } else if (last_ragged) {
    kv_tile(j, std::integral_constant<int, 0>{}, std::true_type{});
  }

  // fp8 emission (p.oq): the bf16-rounded outputs scaled by 448 / (previous amax), saturated, as e4m3 bytes of the
  // token-major [B*Lq, H*hd] operand of the following projection; one filtered atomic max per wave records the amax
  float e_alpha = 1.0f, e_max = 0.f;
  if (p.oq) {
    const float ap = *p.e_amax_prev;
    e_alpha = ap > 0.f ? 448.0f / ap : 1.0f;
    if (blockIdx.x == 0 && tid == 0) *p.e_dq_out = ap > 0.f ? ap / 448.0f : 1.0f;
  }
#pragma unroll
  for (int cb = 0; cb < 4; ++cb) {
    const int qrow = qrow0 + 16 * cb;
    // denominator: row HD of O^T (V's ones column) = block 4, lanes g = 2, register 0
    const float lt = __shfl(o[NDB - 1][cb][0], (lane & 15) | 32, 64);
    const bool valid = qrow < p.Lq;
    const float inv = sv / lt;
    float vals[NDB][4];
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
      for (int e = 0; e < 4; ++e) vals[db][e] = o[db][cb][e] * inv;
    bf16_t* orow = p.o + b * p.o_sb + hh * p.o_sh + (long)qrow * p.o_sl;
    unsigned char* qrow8 = p.oq + ((long)b * p.Lq + qrow) * p.oq_ld + hh * HD;
    e_max = fmaxf(e_max, store_block_t<HD, 0>(vals, orow, qrow8, p.oq != nullptr, e_alpha, g, valid, p.wide_o != 0));
    if (valid && g == 0) p.lse[head * p.Lq + qrow] = (m[cb] - P_SHIFT + __builtin_amdgcn_logf(lt)) * LN2;
  }
  if (p.oq) {
    e_max = wave_max(e_max);
    if (lane == 0 && e_max > *p.e_amax_cur) atomicMax(reinterpret_cast<int*>(p.e_amax_cur), __float_as_int(e_max));
  }
}

// ===================================== dK, dV ===============================================
// Workgroup = 128 keys (4 waves x 2 column blocks of 16), K / V rows of the wave as B operands in registers; Q / dO
// tiles of 128 queries and their row statistics (8 - lse2, -delta') by LDS-DMA, double-buffered.
template <int HD>
__global__ __launch_bounds__(256, 2) void attn8_bwd_dkv_kernel(Attn8P p) {
  static_assert(HD == 72, "5 output blocks of 16");
  constexpr int NDB = 5;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* stats = smem + 4 * TILE;  // [2 bufs][nl[128] | nd[128]] floats
  int bh, kt_idx;
  if (!decode_block(p.n_rt, p.tail_last, p.B * p.H, bh, kt_idx)) return;
  const int b = bh / p.H, hh = bh % p.H;
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int key0 = kt_idx * 128 + wave * 32 + (lane & 15);
  const long head = (long)b * p.H + hh;
  const long nrows = (long)p.B * p.H * p.Lq;

  const srd_t rq = make_srd(p.q + head * p.Lq * ROWB, (unsigned)(p.Lq * ROWB));
  const srd_t rdo = make_srd(p.d_o + head * p.Lq * ROWB, (unsigned)(p.Lq * ROWB));
  const __amdgpu_buffer_rsrc_t rk = make_rsrc(p.k + head * p.Lk * ROWB, (unsigned)(p.Lk * ROWB));
  const __amdgpu_buffer_rsrc_t rv = make_rsrc(p.v + head * p.Lk * ROWB, (unsigned)(p.Lk * ROWB));
  const srd_t rnd = make_srd(p.stats + head * p.Lq, (unsigned)(p.Lq * 4));
  const srd_t rnl = make_srd(p.stats + nrows + head * p.Lq, (unsigned)(p.Lq * 4));

  Stage8 st;
  st.init(wave, lane);
  auto issue_tile = [&](int j, int par) {
    char* nb = smem + par * 2 * TILE;
    st.issue(rq, nb, (unsigned)j * TILE, wave);
    st.issue(rdo, nb + TILE, (unsigned)j * TILE, wave);
    if (wave < 2) {  // 128 rows x 4 B per statistic: wave 0 -> 8 - lse2, wave 1 -> -delta'
      const unsigned sa = lds_addr_of(stats + par * 1024 + wave * 512);
      const srd_t rs = wave == 0 ? rnl : rnd;
      lds_dma4(rs, sa, (unsigned)((j * 128 + lane) * 4));
      lds_dma4(rs, sa + 256, (unsigned)((j * 128 + 64 + lane) * 4));
    }
  };
  issue_tile(0, 0);

  i32x8 kf[2], vf[2];
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) {
    kf[cb] = load_row32(rk, key0 + 16 * cb, g);
    vf[cb] = load_row32(rv, key0 + 16 * cb, g);
  }
  const float s_q = p.deq[0], s_v = p.deq[2], s_do = p.deq[3];
  // S products: log2-domain exponents (x 2^23: the bit pattern of 2^x, see PEXP_C)
  const int sc_s = __builtin_amdgcn_readfirstlane(127 - (int)p.deq[4] + (EXACT_EXP ? 0 : 23));
  constexpr int SC_DP = 127 - 8;                                          // dP products: x 2^-8
  Frag8 fr;
  fr.init(lane);

  f32x4 dk[NDB][2], dv[NDB][2];
#pragma unroll
  for (int db = 0; db < NDB; ++db)
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) { dk[db][cb] = zero4(); dv[db][cb] = zero4(); }
  const int nqt = (p.Lq + 127) / 128;
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) { retire(kf[cb]); retire(vf[cb]); }
  VDS_WAIT_VM(0);
  __syncthreads();

  // NKB: 16-key blocks of this wave that hold keys (2, except in a head's last workgroup: its waves without keys only
  // stage and synchronise; see attn_fwd16_kernel in attention.hip)
  auto q_tile = [&](int j, auto PAR, auto NKBT) {
    constexpr int par = decltype(PAR)::value;
    constexpr int NKB = decltype(NKBT)::value;
    if (j + 1 < nqt) issue_tile(j + 1, par ^ 1);
    const char* qt = smem + par * 2 * TILE;
    const char* dot = qt + TILE;
    const float* stl = reinterpret_cast<const float*>(stats + par * 1024);
    if constexpr (NKB > 0) {
    i32x8 pq[2], dsq[2];
    // software pipeline by hand (as in the forward kernel): the S / dP products of block i+1 are issued before the
    // exp2 / multiply / pack VALU work of block i; accumulator register r <-> query 32 g + 4 i + r of the tile
    f32x4 sn[2], dpn[2];
    i32x8 aqn, adn;   // fragments of the block whose products are issued in the NEXT iteration: read one iteration ahead
    f32x4 nln, ndn;   // of their use (with their rows' constants), so that no MFMA waits on an LDS round trip
    {
      const i32x8 aq = fr.row<0>(qt);
      const i32x8 ad = fr.row<0>(dot);
      const f32x4 nl4 = *reinterpret_cast<const f32x4*>(stl + 32 * g);
      const f32x4 nd4 = *reinterpret_cast<const f32x4*>(stl + 128 + 32 * g);
#pragma unroll
      for (int cb = 0; cb < NKB; ++cb) {
        sn[cb] = mfma8s<0, 0>(aq, kf[cb], nl4, sc_s, 127);     // 8 - lse2 + log2-domain score
        dpn[cb] = mfma8s<1, 0>(ad, vf[cb], nd4, SC_DP, 127);   // (dO V^T - delta) / (256 s_do s_v)
      }
      aqn = fr.row<1>(qt);
      adn = fr.row<1>(dot);
      nln = *reinterpret_cast<const f32x4*>(stl + 32 * g + 4);
      ndn = *reinterpret_cast<const f32x4*>(stl + 128 + 32 * g + 4);
    }
    static_for<8>([&](auto I) {
      constexpr int i = decltype(I)::value;
      f32x4 s[2], dp[2];
#pragma unroll
      for (int cb = 0; cb < NKB; ++cb) { s[cb] = sn[cb]; dp[cb] = dpn[cb]; }
      if constexpr (i < 7) {
#pragma unroll
        for (int cb = 0; cb < NKB; ++cb) {
          sn[cb] = mfma8s<0, 0>(aqn, kf[cb], nln, sc_s, 127);
          dpn[cb] = mfma8s<1, 0>(adn, vf[cb], ndn, SC_DP, 127);
        }
      }
      if constexpr (i < 6) {
        aqn = fr.row<i + 2>(qt);
        adn = fr.row<i + 2>(dot);
        nln = *reinterpret_cast<const f32x4*>(stl + 32 * g + 4 * (i + 2));
        ndn = *reinterpret_cast<const f32x4*>(stl + 128 + 32 * g + 4 * (i + 2));
      }
#pragma unroll
      for (int cb = 0; cb < NKB; ++cb) {
        f32x4 pr;
#pragma unroll
        for (int r = 0; r < 4; ++r) pr[r] = EXACT_EXP ? __builtin_amdgcn_exp2f(s[cb][r]) : pexp_bits(s[cb][r]);  // 256 P
        f32x4 ds;  // P (dP - delta) / (s_do s_v): |.| <= 4032 (file header).  (Scalar multiplies: v_pk_mul_f32 measured slower.)
#pragma unroll
        for (int r = 0; r < 4; ++r) ds[r] = pr[r] * dp[cb][r];
        pq[cb][i] = cvt4_e4m3(pr);
        dsq[cb][i] = cvt4_e5m2(ds);
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    // dV^T += dO^T P, dK^T += Q^T dS: transposed fragments double-buffered by hand (see the forward kernel)
    i32x8 doa = fr.trans<0>(dot), qta = fr.trans<0>(qt), dob, qtb;
    static_for<NDB>([&](auto DB) {
      constexpr int db = decltype(DB)::value;
      if constexpr (db + 1 < NDB) {
        if constexpr (db & 1) { doa = fr.trans<db + 1>(dot); qta = fr.trans<db + 1>(qt); }
        else { dob = fr.trans<db + 1>(dot); qtb = fr.trans<db + 1>(qt); }
      }
#pragma unroll
      for (int cb = 0; cb < NKB; ++cb) {
        dv[db][cb] = mfma8<1, 0>((db & 1) ? dob : doa, pq[cb], dv[db][cb]);
        dk[db][cb] = mfma8<0, 1>((db & 1) ? qtb : qta, dsq[cb], dk[db][cb]);
      }
      __builtin_amdgcn_sched_barrier(0);
    });
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
  const int nkb = min(2, max(0, (p.Lk - (kt_idx * 128 + wave * 32) + 15) >> 4));  // wave-uniform
  if (nkb == 2) run(std::integral_constant<int, 2>{});
  else if (nkb == 1) run(std::integral_constant<int, 1>{});
  else run(std::integral_constant<int, 0>{});
  const float fk = p.scale * s_do * s_v * s_q;  // dS_q = dS / (s_do s_v)
  const float fv = s_do * (1.0f / 256.0f);      // P_q = 256 P
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) {
    const int krow = key0 + 16 * cb;
    float vk[NDB][4], vv[NDB][4];
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
      for (int e = 0; e < 4; ++e) { vk[db][e] = dk[db][cb][e]; vv[db][e] = dv[db][cb][e]; }
    store_block_bf16_t(vk, fk, p.dk + b * p.dk_sb + hh * p.dk_sh + (long)krow * p.dk_sl, HD, g, krow < p.Lk, p.wide_dkv);
    store_block_bf16_t(vv, fv, p.dv + b * p.dv_sb + hh * p.dv_sh + (long)krow * p.dv_sl, HD, g, krow < p.Lk, p.wide_dkv);
  }
}

// ===================================== dQ ===================================================
// Workgroup = 128 queries (4 waves x 2 column blocks of 16), Q / dO rows as B operands in registers, the query's
// 8 - lse2 as the exponent's addend and -delta' as the dP accumulator's start; K / V tiles of 128 keys by LDS-DMA.
// Keys past Lk are zero rows of K, but their dS is NOT harmless: S = 8 - lse2 + 0 there, so P = 2^(8 - lse2), and for a
// query whose lse2 is a few units negative P (dP - delta) leaves the e5m2 range -- the cast does not saturate, inf (or
// NaN) times the zero column of K^T is NaN inside the MFMA and poisons the whole dQ row.  The last, partly filled tile
// therefore runs the RAGGED instantiation, which forces dS = 0 for key index >= Lk before the pack (the forward kernel
// masks the same keys with -inf); full tiles pay nothing.
// NW = 6 (192 queries per workgroup, VDS_ATTN8_DQ_WAVES=6): the kernel needs <= 168 registers, so three waves fit on a
// SIMD with two workgroups of six waves.  Measured SLOWER than NW = 4 (2.7 vs 2.0 ms at B=6, L=8208): kept as an
// experiment only.
template <int HD, int NW>
__global__ __launch_bounds__(64 * NW, (NW == 6 ? 3 : 2)) void attn8_bwd_dq_kernel(Attn8P p) {
  static_assert(HD == 72, "5 output blocks of 16");
  constexpr int NDB = 5;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int bh, qt;
  if (!decode_block(p.n_rt, p.tail_last, p.B * p.H, bh, qt)) return;
  const int b = bh / p.H, hh = bh % p.H;
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qrow0 = qt * (32 * NW) + wave * 32 + (lane & 15);
  const long head = (long)b * p.H + hh;
  const long nrows = (long)p.B * p.H * p.Lq;

  const __amdgpu_buffer_rsrc_t rq = make_rsrc(p.q + head * p.Lq * ROWB, (unsigned)(p.Lq * ROWB));
  const __amdgpu_buffer_rsrc_t rdo = make_rsrc(p.d_o + head * p.Lq * ROWB, (unsigned)(p.Lq * ROWB));
  const srd_t rk = make_srd(p.k + head * p.Lk * ROWB, (unsigned)(p.Lk * ROWB));
  const srd_t rv = make_srd(p.v + head * p.Lk * ROWB, (unsigned)(p.Lk * ROWB));
  Stage8 st;  // the tiles are staged by waves 0-3 (4 pieces of each tile per wave)
  st.init(wave & 3, lane);
  if (wave < 4) {
    st.issue(rk, smem, 0, wave);
    st.issue(rv, smem + TILE, 0, wave);
  }

  i32x8 qf[2], dof[2];
  f32x4 nl4[2], nd4[2];  // the query's constants as accumulator start values
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) {
    const int qrow = qrow0 + 16 * cb;
    qf[cb] = load_row32(rq, qrow, g);
    dof[cb] = load_row32(rdo, qrow, g);
    const long srow = head * p.Lq + min(qrow, p.Lq - 1);
    const float nd = p.stats[srow];
    const float nl = p.stats[nrows + srow];
    nd4[cb] = f32x4{nd, nd, nd, nd};
    nl4[cb] = f32x4{nl, nl, nl, nl};
  }
  const float s_k = p.deq[1], s_v = p.deq[2], s_do = p.deq[3];
  const int sc_s = __builtin_amdgcn_readfirstlane(127 - (int)p.deq[4] + (EXACT_EXP ? 0 : 23));
  constexpr int SC_DP = 127 - 8;
  Frag8 fr;
  fr.init(lane);

  f32x4 dq[NDB][2];
#pragma unroll
  for (int db = 0; db < NDB; ++db)
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) dq[db][cb] = zero4();
  const int nkt = (p.Lk + 127) / 128;
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) { retire(qf[cb]); retire(dof[cb]); retire(nl4[cb][0]); retire(nd4[cb][0]); }
  VDS_WAIT_VM(0);
  __syncthreads();

  // NCB: 16-query blocks of this wave that hold queries (see attn8_bwd_dkv_kernel)
  auto kv_tile = [&](int j, auto PAR, auto NCBT, auto RAG) {
    constexpr int par = decltype(PAR)::value;
    constexpr int NCB = decltype(NCBT)::value;
    constexpr bool ragged = decltype(RAG)::value;
    const int key_lim = p.Lk - j * 128 - 32 * g;  // keys of this lane's 32-row range that exist: [0, key_lim)
    if (j + 1 < nkt && wave < 4) {
      char* nk = smem + (par ^ 1) * 2 * TILE;
      st.issue(rk, nk, (unsigned)(j + 1) * TILE, wave);
      st.issue(rv, nk + TILE, (unsigned)(j + 1) * TILE, wave);
    }
    const char* kt = smem + par * 2 * TILE;
    const char* vt = kt + TILE;
    if constexpr (NCB > 0) {
    i32x8 dsq[2];
    f32x4 sn[2], dpn[2];  // software pipeline by hand, as in the other two kernels
    i32x8 akn, avn;       // fragments read one iteration ahead of the products that consume them
    {
      const i32x8 ak = fr.row<0>(kt);
      const i32x8 av = fr.row<0>(vt);
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
        sn[cb] = mfma8s<0, 0>(ak, qf[cb], nl4[cb], sc_s, 127);
        dpn[cb] = mfma8s<0, 1>(av, dof[cb], nd4[cb], SC_DP, 127);
      }
      akn = fr.row<1>(kt);
      avn = fr.row<1>(vt);
    }
    static_for<8>([&](auto I) {
      constexpr int i = decltype(I)::value;
      f32x4 s[2], dp[2];
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) { s[cb] = sn[cb]; dp[cb] = dpn[cb]; }
      if constexpr (i < 7) {
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
          sn[cb] = mfma8s<0, 0>(akn, qf[cb], nl4[cb], sc_s, 127);
          dpn[cb] = mfma8s<0, 1>(avn, dof[cb], nd4[cb], SC_DP, 127);
        }
      }
      if constexpr (i < 6) {
        akn = fr.row<i + 2>(kt);
        avn = fr.row<i + 2>(vt);
      }
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
        f32x4 ds;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          ds[r] = (EXACT_EXP ? __builtin_amdgcn_exp2f(s[cb][r]) : pexp_bits(s[cb][r])) * dp[cb][r];  // P (dP - delta) / (s_do s_v)
          if constexpr (ragged) {
            if (4 * i + r >= key_lim) ds[r] = 0.f;  // accumulator register r of block i = tile row 32 g + 4 i + r
          }
        }
        dsq[cb][i] = cvt4_e5m2(ds);
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    // dQ^T += K^T dS^T: transposed K fragments double-buffered by hand (see the forward kernel)
    i32x8 kta = fr.trans<0>(kt), ktb;
    static_for<NDB>([&](auto DB) {
      constexpr int db = decltype(DB)::value;
      if constexpr (db + 1 < NDB) {
        if constexpr (db & 1) kta = fr.trans<db + 1>(kt);
        else ktb = fr.trans<db + 1>(kt);
      }
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) dq[db][cb] = mfma8<0, 1>((db & 1) ? ktb : kta, dsq[cb], dq[db][cb]);
      __builtin_amdgcn_sched_barrier(0);
    });
    }
    VDS_WAIT_VM(0);
    __syncthreads();
  };
  const bool last_ragged = (p.Lk & 127) != 0;
  const int n_full = last_ragged ? nkt - 1 : nkt;
  auto run = [&](auto NCBT) {
    int j = 0;
    for (; j + 1 < n_full; j += 2) {
      kv_tile(j, std::integral_constant<int, 0>{}, NCBT, std::false_type{});
      kv_tile(j + 1, std::integral_constant<int, 1>{}, NCBT, std::false_type{});
    }
    if (j < n_full) {
      kv_tile(j, std::integral_constant<int, 0>{}, NCBT, std::false_type{});
      if (last_ragged) kv_tile(j + 1, std::integral_constant<int, 1>{}, NCBT, std::true_type{});
    } else if (last_ragged) {
      kv_tile(j, std::integral_constant<int, 0>{}, NCBT, std::true_type{});
    }
  };
  const int ncb = min(2, max(0, (p.Lq - (qt * (32 * NW) + wave * 32) + 15) >> 4));  // wave-uniform
  if (ncb == 2) run(std::integral_constant<int, 2>{});
  else if (ncb == 1) run(std::integral_constant<int, 1>{});
  else run(std::integral_constant<int, 0>{});
  const float fq = p.scale * s_do * s_v * s_k;
  // fp8 emission (p.dqq): dQ (rounded to bf16 first) as e5m2 bytes of the token-major [B*Lq, H*hd] operand of the
  // q_cross weight- and input-gradient GEMMs; the bf16 store is skipped when the caller passes no dq
  float e_alpha = 1.0f, e_max = 0.f;
  if (p.dqq) {
    const float ap = *p.e_amax_prev;
    e_alpha = ap > 0.f ? 57344.0f / ap : 1.0f;
    if (blockIdx.x == 0 && tid == 0) *p.e_dq_out = ap > 0.f ? ap / 57344.0f : 1.0f;
  }
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) {
    const int qrow = qrow0 + 16 * cb;
    float vals[NDB][4];
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
      for (int e = 0; e < 4; ++e) vals[db][e] = dq[db][cb][e] * fq;
    bf16_t* dqp = p.dq ? p.dq + b * p.dq_sb + hh * p.dq_sh + (long)qrow * p.dq_sl : nullptr;
    unsigned char* q8p = p.dqq + ((long)b * p.Lq + qrow) * p.dqq_ld + hh * HD;
    e_max = fmaxf(e_max, store_block_t<HD, 1>(vals, dqp, q8p, p.dqq != nullptr, e_alpha, g, qrow < p.Lq, p.wide_dq != 0));
  }
  if (p.dqq) {
    e_max = wave_max(e_max);
    if (lane == 0 && e_max > *p.e_amax_cur) atomicMax(reinterpret_cast<int*>(p.e_amax_cur), __float_as_int(e_max));
  }
}

// ===================================== preprocess ===========================================
// delta preprocess of the fp8 backward: one wave per token of the token-major O / dO ([B*Lq, H*hd] bf16, what the
// model passes).  Per head:  stats[0][b,h,q] = -rowsum(dO o O) / (256 s_do s_v),  stats[1][b,h,q] = (8 - lse log2(e) + 127 - 0.043) 2^23 (the exponent's start value: PEXP_C);
// dO leaves as e5m2 rows [B,H,Lq,128] (bytes [hd,128) are never written: the buffer is zeroed once by its owner),
// scaled so that the previous step's amax lands on DO_TARGET; the current amax is recorded (delayed scaling).
constexpr float DO_TARGET = 0.0625f;  // dO_q in [2^-16, 2^-4]: dS_q = 256 P dP' then stays below the e5m2 maximum
__global__ __launch_bounds__(256) void attn8_delta_kernel(const bf16_t* o, long o_sb, long o_sl, const bf16_t* d_o,
                                                          long do_sb, long do_sl, const float* lse, float* stats,
                                                          unsigned char* doq, const float* amax_prev, float* amax_cur,
                                                          float* deq, int B, int H, int Lq, int hd) {
  __shared__ float part[4][192];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long tok = (long)blockIdx.x * 4 + wave;
  const float ap = *amax_prev;
  const float alpha = ap > 0.f ? DO_TARGET / ap : 1.0f;
  const float s_do = 1.0f / alpha;
  if (blockIdx.x == 0 && threadIdx.x == 0) deq[3] = s_do;
  if (tok >= (long)B * Lq) return;
  const int b = (int)(tok / Lq), q = (int)(tok % Lq);
  const int nch = H * hd / 8, cph = hd / 8;
  const bf16_t* orow = o + b * o_sb + (long)q * o_sl;
  const bf16_t* drow = d_o + b * do_sb + (long)q * do_sl;
  float amax = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int c = lane + 64 * i;
    if (c < nch) {
      const u32x4 a = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(orow + c * 8));  // read once
      const u32x4 gq = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(drow + c * 8));
      float acc = 0.f;
      float f[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        f[2 * e] = bflo(gq[e]);
        f[2 * e + 1] = bfhi(gq[e]);
        acc += bflo(a[e]) * f[2 * e] + bfhi(a[e]) * f[2 * e + 1];
      }
      part[wave][c] = acc;
      u32x2 w;
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        int x = 0;
        // saturating: an element that outgrew the previous step's amax is clipped at DO_TARGET -- the bound the
        // backward kernels' dS range rests on (delayed scaling; the cast itself does not saturate)
        const float g0 = __builtin_amdgcn_fmed3f(f[4 * e] * alpha, -DO_TARGET, DO_TARGET);
        const float g1 = __builtin_amdgcn_fmed3f(f[4 * e + 1] * alpha, -DO_TARGET, DO_TARGET);
        const float g2 = __builtin_amdgcn_fmed3f(f[4 * e + 2] * alpha, -DO_TARGET, DO_TARGET);
        const float g3 = __builtin_amdgcn_fmed3f(f[4 * e + 3] * alpha, -DO_TARGET, DO_TARGET);
        x = __builtin_amdgcn_cvt_pk_bf8_f32(g0, g1, x, false);
        x = __builtin_amdgcn_cvt_pk_bf8_f32(g2, g3, x, true);
        w[e] = (unsigned)x;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(f[e]));
      const int hh = c / cph, ci = c % cph;
      *reinterpret_cast<u32x2*>(doq + (((long)b * H + hh) * Lq + q) * ROWB + 8 * ci) = w;
    }
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): this wave's own LDS writes
  __builtin_amdgcn_wave_barrier();
  if (lane < H) {
    float acc = 0.f;
    for (int i = 0; i < cph; ++i) acc += part[wave][lane * cph + i];
    const long rows = (long)B * H * Lq;
    const long row = ((long)b * H + lane) * Lq + q;
    stats[row] = -acc / (s_do * deq[2] * 256.0f);
    const float nl = P_SHIFT - lse[row] * LOG2E;  // the exponent's per-query addend: 8 - lse2 ...
    stats[rows + row] = EXACT_EXP ? nl : (nl + (127.0f - PEXP_C)) * PEXP_ONE;  // ... in the form the S-type MFMA starts from
  }
  amax = wave_max(amax);
  if (lane == 0 && amax > *amax_cur) atomicMax(reinterpret_cast<int*>(amax_cur), __float_as_int(amax));
}

// qkv head split + 3-D RoPE + residual-V mix (model.py:125-134,266-275) with fp8 outputs: q, k, v leave as e4m3 rows
// [B,H,L,128] (bytes [hd,128) zero, V byte hd = 1.0: the softmax denominator's ones column), scaled with the previous
// step's amax (delayed scaling, saturating): k and v so that it lands on 448; q so that s_q s_k log2(e) / sqrt(hd) is
// a power of two 2^-E with its amax in (224, 448] (the attention kernels fold 2^-E into their block-scaled MFMAs).
// The current amax of each tensor is recorded.  The values quantised are the bf16 results of the bf16 kernel
// (vds_qkv_rope_fwd), rounding points included.  v_out (optional): the bf16 v in the padded head-major layout -- block
// 0's v feeds the residual-V mix of the later blocks.
// One thread = one 16-byte output chunk (16 head-dim columns) of one (tensor, token, head); blockIdx.y = tensor.  All
// loads are 16 / 8 bytes wide: the 4-column groups never straddle the rotation halves (hd / 2 is a multiple of 4).
__device__ __forceinline__ void qk_scales(const float* amax_prev, int stride, int hd, float& alpha_q, float& alpha_k, float& E) {
  const float aq = amax_prev[0], ak = amax_prev[stride];
  alpha_k = ak > 0.f ? 448.0f / ak : 1.0f;
  const float cl = LOG2E / sqrtf((float)hd);
  int e = 1;
  if (aq > 0.f) (void)frexpf(448.0f * alpha_k / (cl * aq), &e);  // x = f 2^e, f in [0.5, 1): floor(log2 x) = e - 1
  // the kernels pass 127 - E, 127 + 3 - E and (backward, bit-pattern exp2: round 4) 127 + 23 - E as E8M0 scale bytes
  // (255 = NaN, larger values wrap): E is kept in [-104, 120] (extreme amax products only; alpha_q follows E, so q merely
  // leaves its top binade there instead of wrapping the byte)
  e = e - 1 < -104 ? -103 : e - 1 > 120 ? 121 : e;
  E = (float)(e - 1);
  alpha_q = cl * exp2f(E) / alpha_k;
}
__global__ __launch_bounds__(256) void qkv_rope_fwd_fp8_kernel(const bf16_t* qkv, const float* cosb, const float* sinb,
                                                               const bf16_t* v0, const bf16_t* lamp, unsigned char* q8,
                                                               unsigned char* k8, unsigned char* v8, bf16_t* v_out,
                                                               const float* amax_prev, float* amax_cur, int amax_stride,
                                                               float* deq, int B, int L, int H, int hd, int hdp) {
  const int which = blockIdx.y;
  const long gid = (long)blockIdx.x * 256 + threadIdx.x;
  const long n = (long)B * H * L * 8;
  float alpha;
  {
    float aq, ak, E;
    qk_scales(amax_prev, amax_stride, hd, aq, ak, E);
    const float av = amax_prev[2 * amax_stride];
    alpha = which == 0 ? aq : which == 1 ? ak : (av > 0.f ? 448.0f / av : 1.0f);
    if (gid == 0) {
      deq[which] = 1.0f / alpha;
      if (which == 0) deq[4] = E;
    }
  }
  float amax = 0.f;
  if (gid < n) {
    // thread order (token, head, chunk): a wave reads 8 consecutive heads of one token (contiguous in the qkv row)
    // and writes 8 complete 128-byte head rows
    const int c = (int)(gid & 7);
    const int hh = (int)((gid >> 3) % H);
    const long tok = gid / (8 * H);
    const int l = (int)(tok % L), b = (int)(tok / L);
    const long row = ((long)b * H + hh) * L + l;
    const int half = hd >> 1, D = H * hd;
    const bf16_t* src = qkv + tok * 3 * D + which * D + hh * hd;
    unsigned char* dst = (which == 0 ? q8 : which == 1 ? k8 : v8) + row * ROWB + 16 * c;
    const int d0 = 16 * c;
    float y[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) y[e] = 0.f;
    if (d0 < hd) {
      // own 16 columns (those past hd: the next head's / tensor's first columns, masked below), 4 groups of 4
      u32x2 own[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) own[j] = d0 + 4 * j < hd ? *reinterpret_cast<const u32x2*>(src + d0 + 4 * j) : u32x2{0u, 0u};
      if (which < 2) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int d = d0 + 4 * j;
          if (d < hd) {
            const bool lo = d < half;
            const int jj = lo ? d : d - half;
            const u32x2 pt = *reinterpret_cast<const u32x2*>(src + (lo ? d + half : d - half));
            const f32x4 c4 = *reinterpret_cast<const f32x4*>(cosb + (long)l * half + jj);
            const f32x4 s4 = *reinterpret_cast<const f32x4*>(sinb + (long)l * half + jj);
            const float x[4] = {bflo(own[j][0]), bfhi(own[j][0]), bflo(own[j][1]), bfhi(own[j][1])};
            const float pp[4] = {bflo(pt[0]), bfhi(pt[0]), bflo(pt[1]), bfhi(pt[1])};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float v = lo ? x[e] * c4[e] + pp[e] * s4[e] : x[e] * c4[e] - pp[e] * s4[e];
              y[4 * j + e] = bf2f(f2bf(v));
            }
          }
        }
      } else {
        float lam = 0.f, oml = 0.f;
        if (v0) {
          lam = bf2f(*lamp);
          oml = bf2f(f2bf(1.0f - lam));
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int d = d0 + 4 * j;
          if (d < hd) {
            float x[4] = {bflo(own[j][0]), bfhi(own[j][0]), bflo(own[j][1]), bfhi(own[j][1])};
            if (v0) {
              const u32x2 a = *reinterpret_cast<const u32x2*>(v0 + row * hdp + d);
              const float z[4] = {bflo(a[0]), bfhi(a[0]), bflo(a[1]), bfhi(a[1])};
#pragma unroll
              for (int e = 0; e < 4; ++e) x[e] = bf2f(f2bf(bf2f(f2bf(lam * x[e])) + bf2f(f2bf(oml * z[e]))));
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) y[4 * j + e] = x[e];
          }
        }
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) amax = fmaxf(amax, fabsf(y[e]));
    }
    if (which == 2 && v_out && d0 < hdp) {  // bf16 v with the bf16 kernels' pad (ones at hd, hd+4 when the pad is >= 8 wide)
      bf16_t* vo = v_out + row * hdp + d0;
      const bool ones = (hdp - hd) >= 8;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int d = d0 + 4 * j;
        if (d < hdp) {
          u32x2 w = {pack_bf2(y[4 * j], y[4 * j + 1]), pack_bf2(y[4 * j + 2], y[4 * j + 3])};
          if (d >= hd) w = (ones && (d == hd || d == hd + 4)) ? u32x2{0x3f80u, 0u} : u32x2{0u, 0u};
          *reinterpret_cast<u32x2*>(vo + 4 * j) = w;
        }
      }
    }
    u32x4 w;
#pragma unroll
    for (int e = 0; e < 16; ++e) y[e] = __builtin_amdgcn_fmed3f(y[e] * alpha, -448.0f, 448.0f);  // saturating cast
#pragma unroll
    for (int e = 0; e < 4; ++e) w[e] = fp8_cvt4<0>(y[4 * e], y[4 * e + 1], y[4 * e + 2], y[4 * e + 3]);
    if (which == 2 && hd >= d0 && hd < d0 + 16) {  // ones column: byte hd of every V row = 1.0 (e4m3 0x38)
      const int e = hd - d0;
      w[e >> 2] = (w[e >> 2] & ~(0xffu << (8 * (e & 3)))) | (0x38u << (8 * (e & 3)));
    }
    *reinterpret_cast<u32x4*>(dst) = w;
  }
  amax = wave_max(amax);
  float* ac = amax_cur + which * amax_stride;
  if ((threadIdx.x & 63) == 0 && amax > *ac) atomicMax(reinterpret_cast<int*>(ac), __float_as_int(amax));
}

// Token-tile form of the producer above (rope_stage.h): T consecutive tokens per workgroup, qkv rows copied to LDS by
// LDS-DMA, q / k rotated in place there (bf16, the rounding points of the bf16 kernel), then one lane = one 16-byte
// chunk of a 128-byte fp8 row: every global access is 16 bytes wide and a (tensor, head) leaves the workgroup as a run
// of T complete rows.  Same values, scales and recorded amax as the element-wise kernel.
// NT / ROPE / tid0: the rows hold NT tensors side by side ([tokens, NT * D]), the first of which is tensor tid0 of
// (q, k, v) -- NT = 3, ROPE: the self-attention qkv rows; NT = 1, tid0 = 0 and NT = 2, tid0 = 1 without rotation: the
// q_cross and context_kv outputs of cross-attention (model.py:147-155), which become fp8 rows the same way.
template <int HD, int HDP, int T, int NT = 3, bool ROPE = true>
__global__ __launch_bounds__(256) void qkv_rope_fwd_fp8_tile_kernel(const bf16_t* qkv, const float* cosb, const float* sinb,
                                                                    const bf16_t* v0, const bf16_t* lamp,
                                                                    unsigned char* q8, unsigned char* k8,
                                                                    unsigned char* v8, bf16_t* v_out,
                                                                    const float* amax_prev, float* amax_cur,
                                                                    int amax_stride, float* deq, long ntok, int L, int H,
                                                                    int tid0) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int D = H * HD, row_b = 2 * NT * D;
  const long tok0 = (long)blockIdx.x * T;
  const int nt = (int)min((long)T, ntok - tok0);
  ropestage::issue_rows(qkv + tok0 * NT * D, smem, nt, row_b, wave, lane);
  float alpha[3];
  {
    float E;
    qk_scales(amax_prev, amax_stride, HD, alpha[0], alpha[1], E);
    const float av = amax_prev[2 * amax_stride];
    alpha[2] = av > 0.f ? 448.0f / av : 1.0f;
    if (blockIdx.x == 0 && tid == 0) {
#pragma unroll
      for (int i = 0; i < NT; ++i) deq[tid0 + i] = 1.0f / (tid0 + i == 0 ? alpha[0] : tid0 + i == 1 ? alpha[1] : alpha[2]);
      if (tid0 == 0) deq[4] = E;
    }
  }
  float lam = 0.f, oml = 0.f;
  if (v0) {
    lam = bf2f(*lamp);
    oml = bf2f(f2bf(1.0f - lam));
  }
  VDS_WAIT_VM(0);
  __syncthreads();
  if constexpr (ROPE) {
    ropestage::rotate_rows<HD>(smem, cosb, sinb, tok0, nt, L, H, row_b, tid);
    __syncthreads();
  }
  const ropestage::Div by_nt((unsigned)nt), by_h((unsigned)H);
  const int b0 = (int)(tok0 / L), l0 = (int)(tok0 % L);
  float am[3] = {0.f, 0.f, 0.f};
  const int nch = nt * NT * H * 8;
  for (int u = tid; u < nch; u += 256) {
    const int c = u & 7;
    unsigned t, hh;
    const unsigned r = by_nt.div((unsigned)(u >> 3), t);
    const unsigned tl = by_h.div(r, hh);        // tensor index inside the row
    const unsigned tensor = (unsigned)tid0 + tl;  // ... and among (q, k, v)
    int l = l0 + (int)t, b = b0;
    if (l >= L) { l -= L; ++b; }
    const long row = ((long)b * H + hh) * L + l;
    u32x4 w = {0u, 0u, 0u, 0u};
    if (16 * c < HD) {
      const char* src = smem + t * row_b + ((tl * H + hh) * HD + 16 * c) * 2;
      u32x4 x[2];
      x[0] = *reinterpret_cast<const u32x4*>(src);
      x[1] = (16 * c + 8 < HD) ? *reinterpret_cast<const u32x4*>(src + 16) : u32x4{0u, 0u, 0u, 0u};
      if (tensor == 2 && v0) {
        const bf16_t* z = v0 + row * HDP + 16 * c;
        x[0] = ropestage::mix_v(x[0], *reinterpret_cast<const u32x4*>(z), lam, oml);
        if (16 * c + 8 < HD) x[1] = ropestage::mix_v(x[1], *reinterpret_cast<const u32x4*>(z + 8), lam, oml);
      }
      const float a = tensor == 0 ? alpha[0] : tensor == 1 ? alpha[1] : alpha[2];
      float y[16], m = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        y[2 * e] = bflo(x[0][e]); y[2 * e + 1] = bfhi(x[0][e]);
        y[8 + 2 * e] = bflo(x[1][e]); y[8 + 2 * e + 1] = bfhi(x[1][e]);
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        m = fmaxf(m, fabsf(y[e]));
        y[e] = __builtin_amdgcn_fmed3f(y[e] * a, -448.0f, 448.0f);  // saturating cast
      }
      if (tensor == 0) am[0] = fmaxf(am[0], m);
      else if (tensor == 1) am[1] = fmaxf(am[1], m);
      else am[2] = fmaxf(am[2], m);
#pragma unroll
      for (int e = 0; e < 4; ++e) w[e] = fp8_cvt4<0>(y[4 * e], y[4 * e + 1], y[4 * e + 2], y[4 * e + 3]);
    }
    if (tensor == 2 && c == HD / 16) {  // ones column: byte hd of every V row = 1.0 (e4m3 0x38)
      constexpr int e = HD % 16;
      w[e >> 2] = (w[e >> 2] & ~(0xffu << (8 * (e & 3)))) | (0x38u << (8 * (e & 3)));
    }
    *reinterpret_cast<u32x4*>((tensor == 0 ? q8 : tensor == 1 ? k8 : v8) + row * ROWB + 16 * c) = w;
  }
  if (v_out) {  // bf16 v in the padded head-major layout of the bf16 kernels (ones at hd, hd+4 when the pad is >= 8 wide)
    constexpr int CPR = HDP >> 3, DCH = HD >> 3;
    constexpr bool ONES = (HDP - HD) >= 8;
    const int nv = nt * H * CPR;
    for (int u = tid; u < nv; u += 256) {
      const int c = u % CPR;
      unsigned t;
      const unsigned hh = by_nt.div((unsigned)(u / CPR), t);
      int l = l0 + (int)t, b = b0;
      if (l >= L) { l -= L; ++b; }
      const long dst = (((long)b * H + hh) * L + l) * HDP + 8 * c;
      u32x4 w = {0u, 0u, 0u, 0u};
      if (c < DCH) {
        w = *reinterpret_cast<const u32x4*>(smem + t * row_b + ((2 * H + hh) * HD + 8 * c) * 2);
        if (v0) w = ropestage::mix_v(w, *reinterpret_cast<const u32x4*>(v0 + dst), lam, oml);
      } else if (ONES && c == DCH) {
        w[0] = 0x3f80u; w[2] = 0x3f80u;
      }
      *reinterpret_cast<u32x4*>(v_out + dst) = w;
    }
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    if (i < tid0 || i >= tid0 + NT) continue;
    const float m = wave_max(am[i]);
    float* ac = amax_cur + i * amax_stride;
    if (lane == 0 && m > *ac) atomicMax(reinterpret_cast<int*>(ac), __float_as_int(m));
  }
}

// knob attn_tail_last = 0 keeps the head-major order for ragged lengths too (A/B)
int tail_last_for(int L, int tile) {
  return (vdscfg::geti(vdscfg::ATTN_TAIL_LAST) && L > tile && (L % tile) != 0) ? 1 : 0;
}

template <typename K>
void set_lds(K kern, int bytes) {
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

bool args_ok(const vds_attn_fp8_args* a, bool bwd) {
  if (!a || !a->q || !a->k || !a->v || !a->deq || a->B <= 0 || a->H <= 0 || a->Lq <= 0 || a->Lk <= 0) return false;
  if ((long)a->Lq * ROWB > 0x7fffffffL || (long)a->Lk * ROWB > 0x7fffffffL) return false;
  // fp8 rows (o_q / dq_q): 16-byte stores at byte offset head * head_dim + 16 g of a row -- 8-byte aligned for odd heads
  // (head_dim 72): base and row stride must be multiples of 8 bytes (the hardware's unaligned global stores take the rest).
  // bf16 rows (o / dq): strides in multiples of 4 elements and an 8-byte aligned base (8-byte stores); when they are
  // multiples of 8 elements on a 16-byte aligned base the kernels take the 16-byte row stores (wide_o / wide_dq).
  if (!bwd) {
    if (a->o_q && (!a->e_amax_prev || !a->e_amax_cur || !a->e_dq_out || (a->o_q_ld & 7) || ((uintptr_t)a->o_q & 7))) return false;
    return a->o && a->lse && (a->o_sb % 4 == 0) && (a->o_sh % 4 == 0) && (a->o_sl % 4 == 0) && ((uintptr_t)a->o % 8 == 0);
  }
  if (a->dq_q && (!a->e_amax_prev || !a->e_amax_cur || !a->e_dq_out || (a->dq_q_ld & 7) || ((uintptr_t)a->dq_q & 7))) return false;
  if (a->dq && ((a->dq_sl % 4) || (a->dq_sh % 4) || (a->dq_sb % 4) || ((uintptr_t)a->dq % 8))) return false;
  return a->d_o && a->stats && (a->dq || a->dq_q) && a->dk && a->dv && (a->dk_sl % 4 == 0) &&
         (a->dv_sl % 4 == 0) && (a->dk_sh % 4 == 0) && (a->dv_sh % 4 == 0) && (a->dk_sb % 4 == 0) && (a->dv_sb % 4 == 0);
}

Attn8P to_p(const vds_attn_fp8_args* a) {
  Attn8P p;
  p.B = a->B; p.H = a->H; p.Lq = a->Lq; p.Lk = a->Lk; p.hd = a->head_dim;
  p.q = (const unsigned char*)a->q; p.k = (const unsigned char*)a->k; p.v = (const unsigned char*)a->v;
  p.d_o = (const unsigned char*)a->d_o;
  p.o = (bf16_t*)a->o; p.o_sb = a->o_sb; p.o_sh = a->o_sh; p.o_sl = a->o_sl;
  p.lse = a->lse;
  p.dq = (bf16_t*)a->dq; p.dq_sb = a->dq_sb; p.dq_sh = a->dq_sh; p.dq_sl = a->dq_sl;
  p.dk = (bf16_t*)a->dk; p.dk_sb = a->dk_sb; p.dk_sh = a->dk_sh; p.dk_sl = a->dk_sl;
  p.dv = (bf16_t*)a->dv; p.dv_sb = a->dv_sb; p.dv_sh = a->dv_sh; p.dv_sl = a->dv_sl;
  p.stats = a->stats;
  p.deq = a->deq;
  p.scale = 1.0f / sqrtf((float)a->head_dim);
  p.n_rt = 0;
  p.tail_last = 0;
  p.oq = (unsigned char*)a->o_q; p.oq_ld = a->o_q_ld;
  p.dqq = (unsigned char*)a->dq_q; p.dqq_ld = a->dq_q_ld;
  p.e_amax_prev = a->e_amax_prev; p.e_amax_cur = a->e_amax_cur; p.e_dq_out = a->e_dq_out;
  auto rows16 = [](const void* base, long sb, long sh, long sl) {
    return base && ((uintptr_t)base % 16 == 0) && (sb % 8 == 0) && (sh % 8 == 0) && (sl % 8 == 0);
  };
  p.wide_dkv = rows16(p.dk, p.dk_sb, p.dk_sh, p.dk_sl) && rows16(p.dv, p.dv_sb, p.dv_sh, p.dv_sl);
  p.wide_o = rows16(p.o, p.o_sb, p.o_sh, p.o_sl);
  p.wide_dq = rows16(p.dq, p.dq_sb, p.dq_sh, p.dq_sl);
  return p;
}

}  // namespace

extern "C" int vds_attn_fp8_supported(int32_t head_dim) { return head_dim == 72 ? 1 : 0; }

extern "C" int vds_attn_fp8_fwd(const vds_attn_fp8_args* a, vds_stream_t stream) {
  if (!args_ok(a, false)) return VDS_ERR_ARG;
  if (a->head_dim != 72) return VDS_ERR_UNSUPPORTED;
  constexpr int LDS = 4 * TILE;
  static bool once = false;
  if (!once) { set_lds(attn8_fwd_kernel<72>, LDS); once = true; }
  Attn8P p = to_p(a);
  p.dqq = nullptr;
  p.n_rt = cdiv(p.Lq, 256);
  p.tail_last = tail_last_for(p.Lq, 256);
  const int grid = cdiv(p.B * p.H, 8) * 8 * p.n_rt;
  hipStream_t s = (hipStream_t)stream;
  vdsprof::Scope ps(VDS_PROF_ATTN_FP8_FWD, s, 4.0 * p.B * p.H * (double)p.Lq * p.Lk * p.hd,
                    (double)p.B * p.H * (ROWB * (p.Lq + 2.0 * p.Lk) + 2.0 * p.hd * p.Lq));
  hipLaunchKernelGGL((attn8_fwd_kernel<72>), dim3(grid), dim3(256), LDS, s, p);
  return hipGetLastError() == hipSuccess ? VDS_OK : VDS_ERR_LAUNCH;
}

extern "C" size_t vds_attn_fp8_bwd_workspace_bytes(const vds_attn_fp8_args* a) {
  if (!a || a->B <= 0 || a->H <= 0 || a->Lq <= 0) return 0;
  return (size_t)2 * a->B * a->H * a->Lq * sizeof(float);
}

// dO preprocess: o / d_o token-major bf16 [B*Lq, H*hd] (row strides o_sl / do_sl elements, batch strides o_sb / do_sb)
extern "C" int vds_attn_fp8_delta(const void* o, int64_t o_sb, int64_t o_sl, const void* d_o, int64_t do_sb,
                                  int64_t do_sl, const float* lse, float* stats, void* doq, const float* amax_prev,
                                  float* amax_cur, float* deq, int32_t B, int32_t H, int32_t Lq, int32_t head_dim,
                                  vds_stream_t stream) {
  if (!o || !d_o || !lse || !stats || !doq || !amax_prev || !amax_cur || !deq || B <= 0 || H <= 0 || Lq <= 0)
    return VDS_ERR_ARG;
  if ((head_dim & 7) || (o_sl & 7) || (o_sb & 7) || (do_sl & 7) || (do_sb & 7)) return VDS_ERR_ARG;
  if (H * head_dim > 1536 || H > 64 || head_dim > ROWB) return VDS_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  const double rows = (double)B * H * Lq;
  vdsprof::Scope ps(VDS_PROF_ATTN_BWD_DELTA, s, 2.0 * rows * head_dim, 5.0 * rows * head_dim);
  hipLaunchKernelGGL(attn8_delta_kernel, dim3((unsigned)(((long)B * Lq + 3) / 4)), dim3(256), 0, s, (const bf16_t*)o,
                     (long)o_sb, (long)o_sl, (const bf16_t*)d_o, (long)do_sb, (long)do_sl, lse, stats,
                     (unsigned char*)doq, amax_prev, amax_cur, deq, B, H, Lq, head_dim);
  return hipGetLastError() == hipSuccess ? VDS_OK : VDS_ERR_LAUNCH;
}

extern "C" int vds_attn_fp8_bwd(const vds_attn_fp8_args* a, vds_stream_t stream) {
  if (!args_ok(a, true)) return VDS_ERR_ARG;
  if (a->head_dim != 72) return VDS_ERR_UNSUPPORTED;
  constexpr int LDS_DKV = 4 * TILE + 2 * 1024, LDS_DQ = 4 * TILE;
  static bool once = false;
  if (!once) {
    set_lds(attn8_bwd_dkv_kernel<72>, LDS_DKV);
    set_lds(attn8_bwd_dq_kernel<72, 4>, LDS_DQ);
    set_lds(attn8_bwd_dq_kernel<72, 6>, LDS_DQ);
    once = true;
  }
  Attn8P p = to_p(a);
  p.oq = nullptr;
  hipStream_t s = (hipStream_t)stream;
  const double prod = 2.0 * p.B * p.H * (double)p.Lq * p.Lk * p.hd;  // credit as in attention.hip: 2 + 2 products
  const double bytes = (double)p.B * p.H * ROWB * (2.0 * p.Lq + 2.0 * p.Lk);
  p.n_rt = cdiv(p.Lk, 128);
  p.tail_last = tail_last_for(p.Lk, 128);
  int grid = cdiv(p.B * p.H, 8) * 8 * p.n_rt;
  {
    vdsprof::Scope ps(VDS_PROF_ATTN_FP8_DKV, s, 2.0 * prod, bytes + 4.0 * p.B * p.H * p.hd * (double)p.Lk);
    hipLaunchKernelGGL((attn8_bwd_dkv_kernel<72>), dim3(grid), dim3(256), LDS_DKV, s, p);
  }
  // knob attn8_dq_waves = 4 | 6 (experiments); measured (B=6, L=8208): 2.0 ms with 4 waves, 2.7 ms with 6
  const int dq_waves = vdscfg::geti(vdscfg::ATTN8_DQ_WAVES) == 6 ? 6 : 4;
  p.n_rt = cdiv(p.Lq, 32 * dq_waves);
  p.tail_last = tail_last_for(p.Lq, 32 * dq_waves);
  grid = cdiv(p.B * p.H, 8) * 8 * p.n_rt;
  {
    vdsprof::Scope ps(VDS_PROF_ATTN_FP8_DQ, s, 2.0 * prod, bytes + 2.0 * p.B * p.H * p.hd * (double)p.Lq);
    if (dq_waves == 6) hipLaunchKernelGGL((attn8_bwd_dq_kernel<72, 6>), dim3(grid), dim3(384), LDS_DQ, s, p);
    else hipLaunchKernelGGL((attn8_bwd_dq_kernel<72, 4>), dim3(grid), dim3(256), LDS_DQ, s, p);
  }
  return hipGetLastError() == hipSuccess ? VDS_OK : VDS_ERR_LAUNCH;
}

// amax_prev / amax_cur: the q, k, v entries at element stride `amax_stride` (fp8.AmaxHistory rows are [prev, cur] pairs)
extern "C" int vds_qkv_rope_fwd_fp8(const void* qkv, const float* cosb, const float* sinb, const void* v0,
                                    const void* lam, void* q8, void* k8, void* v8, void* v_out, const float* amax_prev,
                                    float* amax_cur, int32_t amax_stride, float* deq, int32_t B, int32_t L, int32_t H,
                                    int32_t hd, int32_t hdp, vds_stream_t stream) {
  if (!qkv || !cosb || !sinb || !q8 || !k8 || !v8 || !amax_prev || !amax_cur || !deq || (hd & 7) || hd >= ROWB ||
      hdp < hd)
    return VDS_ERR_ARG;
  if (v0 && !lam) return VDS_ERR_ARG;
  if ((hd & 7) || (hdp & 3) || ((H * hd) & 7)) return VDS_ERR_ARG;  // 8-byte groups of 4 columns; hd / 2 % 4 == 0
  const long n = (long)B * H * L * 8;
  hipStream_t s = (hipStream_t)stream;
  vdsprof::Scope ps(VDS_PROF_QKV_ROPE_FWD, s, 0.0, (double)B * L * H * (6.0 * hd + 3.0 * ROWB + (v0 ? 2.0 * hd : 0.0)));
  // token-tile kernel (rope_stage.h) for the model's head size; knob rope_tile = 0 keeps the element-wise kernel (A/B)
  const int tile = vdscfg::geti(vdscfg::ROPE_TILE);
  if (tile > 0 && hd == 72 && hdp == 96 && H <= 256 && L >= 8 &&
      ropestage::lds_bytes(tile == 2 ? 2 : tile == 8 ? 8 : 4, H * 72) <= 160 * 1024) {  // else: the element-wise kernel
    const long ntok = (long)B * L;
#define ROPE8_TILE(T)                                                                                               \
  do {                                                                                                              \
    static bool attr = false;                                                                                       \
    if (!attr) {                                                                                                    \
      set_lds(qkv_rope_fwd_fp8_tile_kernel<72, 96, T>, 160 * 1024);                                                 \
      attr = true;                                                                                                  \
    }                                                                                                               \
    hipLaunchKernelGGL((qkv_rope_fwd_fp8_tile_kernel<72, 96, T>), dim3((unsigned)((ntok + T - 1) / T)), dim3(256),  \
                       ropestage::lds_bytes(T, H * 72), s, (const bf16_t*)qkv, cosb, sinb, (const bf16_t*)v0,       \
                       (const bf16_t*)lam, (unsigned char*)q8, (unsigned char*)k8, (unsigned char*)v8,              \
                       (bf16_t*)v_out, amax_prev, amax_cur, amax_stride, deq, ntok, L, H, 0);                       \
    return hipGetLastError() == hipSuccess ? VDS_OK : VDS_ERR_LAUNCH;                                               \
  } while (0)
    if (tile == 2) ROPE8_TILE(2);
    else if (tile == 8) ROPE8_TILE(8);
    else ROPE8_TILE(4);
#undef ROPE8_TILE
  }
  hipLaunchKernelGGL(qkv_rope_fwd_fp8_kernel, dim3((unsigned)((n + 255) / 256), 3), dim3(256), 0, s, (const bf16_t*)qkv,
                     cosb, sinb, (const bf16_t*)v0, (const bf16_t*)lam, (unsigned char*)q8, (unsigned char*)k8,
                     (unsigned char*)v8, (bf16_t*)v_out, amax_prev, amax_cur, amax_stride, deq, B, L, H, hd, hdp);
  return hipGetLastError() == hipSuccess ? VDS_OK : VDS_ERR_LAUNCH;
}

// cross-attention operands as fp8 rows (no rotation, no residual-V): q [B*Lq, H*hd] (the q_cross output, model.py:147)
// -> q8 [B,H,Lq,128]; kv [B*Lk, 2*H*hd] (the context_kv output, model.py:149-155: k columns, then v columns) -> k8, v8
// [B,H,Lk,128].  Scales, amax slots and deq entries as vds_qkv_rope_fwd_fp8 (q's factor tied to k's).  Both inputs
// contiguous; head_dim 72.
extern "C" int vds_cross_qkv_fp8(const void* q, const void* kv, void* q8, void* k8, void* v8, const float* amax_prev,
                                 float* amax_cur, int32_t amax_stride, float* deq, int32_t B, int32_t Lq, int32_t Lk,
                                 int32_t H, int32_t hd, vds_stream_t stream) {
  if (!q || !kv || !q8 || !k8 || !v8 || !amax_prev || !amax_cur || !deq || B < 1 || Lq < 1 || Lk < 1 || H < 1) return VDS_ERR_ARG;
  constexpr int T = 4;
  if (hd != 72 || H > 256 || Lq < T || Lk < T) return VDS_ERR_UNSUPPORTED;  // (a tile of T tokens spans at most two samples)
  if ((T * 4 * H * 72 + 1023) / 1024 * 1024 > 160 * 1024) return VDS_ERR_UNSUPPORTED;  // the kv rows of a tile must fit the LDS
  hipStream_t s = (hipStream_t)stream;
  static bool attr = false;
  if (!attr) {
    set_lds(qkv_rope_fwd_fp8_tile_kernel<72, 96, T, 1, false>, 160 * 1024);
    set_lds(qkv_rope_fwd_fp8_tile_kernel<72, 96, T, 2, false>, 160 * 1024);
    attr = true;
  }
  vdsprof::Scope ps(VDS_PROF_QKV_ROPE_FWD, s, 0.0, (double)B * H * ((double)Lq + 2.0 * Lk) * (2.0 * hd + ROWB));
  const int D = H * 72;
  const long nq = (long)B * Lq, nk = (long)B * Lk;
  hipLaunchKernelGGL((qkv_rope_fwd_fp8_tile_kernel<72, 96, T, 1, false>), dim3((unsigned)((nq + T - 1) / T)), dim3(256),
                     (T * 2 * D + 1023) / 1024 * 1024, s, (const bf16_t*)q, (const float*)nullptr, (const float*)nullptr,
                     (const bf16_t*)nullptr, (const bf16_t*)nullptr, (unsigned char*)q8, (unsigned char*)k8,
                     (unsigned char*)v8, (bf16_t*)nullptr, amax_prev, amax_cur, amax_stride, deq, nq, Lq, H, 0);
  hipLaunchKernelGGL((qkv_rope_fwd_fp8_tile_kernel<72, 96, T, 2, false>), dim3((unsigned)((nk + T - 1) / T)), dim3(256),
                     (T * 4 * D + 1023) / 1024 * 1024, s, (const bf16_t*)kv, (const float*)nullptr, (const float*)nullptr,
                     (const bf16_t*)nullptr, (const bf16_t*)nullptr, (unsigned char*)q8, (unsigned char*)k8,
                     (unsigned char*)v8, (bf16_t*)nullptr, amax_prev, amax_cur, amax_stride, deq, nk, Lk, H, 1);
  return hipGetLastError() == hipSuccess ? VDS_OK : VDS_ERR_LAUNCH;
}
