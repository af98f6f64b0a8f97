// HBM-bound glue of the DiT train step, written for wave64 + 16-byte accesses:
// fused RMSNorm+modulate (fwd/bwd), gate backward, column sums, qkv split + 3-D RoPE +
// residual-V (fwd/bwd), the B-row "small" linears (time embed / adaLN), sinusoid,
// patchify / unpatchify, register tokens, noising, flow-matching loss, casts.
// Reference call sites are cited per kernel (file:line of the reference repo).
#include <cstdlib>
#include "common.h"
#include "rope_stage.h"
#include "prof.h"
#include "config.h"
#include "../../include/vds.h"

namespace {
// streaming accesses of the row kernels below (each activation-sized tensor is read / written once per launch), with
// the non-temporal hint where bit WHICH of VDS_EW_NT is set: 1 = rmsnorm_mod_bwd loads, 2 = its store, 4 = gate_bwd,
// 8 = rmsnorm_mod_fwd, 16 = qkv_rope_bwd_tok loads, 32 = gate_bwd's saved y (same-box A/B: DESIGN.md Appendix A)
#ifndef VDS_EW_NT
#define VDS_EW_NT 35
#endif
template <int WHICH>
__device__ __forceinline__ u32x4 ld_stream(const bf16_t* p) {
  if constexpr ((VDS_EW_NT & WHICH) != 0) return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
  else return *reinterpret_cast<const u32x4*>(p);
}
template <int WHICH>
__device__ __forceinline__ void st_stream(bf16_t* p, u32x4 v) {
  if constexpr ((VDS_EW_NT & WHICH) != 0) __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p));
  else *reinterpret_cast<u32x4*>(p) = v;
}

__device__ __forceinline__ void unpack8(const u32x4& u, float (&f)[8]) {
#pragma unroll
  for (int e = 0; e < 4; ++e) { f[2 * e] = bflo(u[e]); f[2 * e + 1] = bfhi(u[e]); }
}
__device__ __forceinline__ u32x4 pack8(const float (&f)[8]) {
  u32x4 u;
#pragma unroll
  for (int e = 0; e < 4; ++e) u[e] = pack_bf2(f[2 * e], f[2 * e + 1]);
  return u;
}
__device__ __forceinline__ void load8f(const float* p, float (&f)[8]) {
  const f32x4 a = *reinterpret_cast<const f32x4*>(p);
  const f32x4 b = *reinterpret_cast<const f32x4*>(p + 4);
  f[0] = a[0]; f[1] = a[1]; f[2] = a[2]; f[3] = a[3]; f[4] = b[0]; f[5] = b[1]; f[6] = b[2]; f[7] = b[3];
}

// ---- fp8 emission (BASELINE config 5; the recipe is stated in fp8.py / quant.hip) --------------------------------
// The three producers of the fp8 GEMMs' activation / gradient operands -- RMSNorm+modulate (qkv / fc1 input), gate
// backward (fc2 output gradient) and the qkv/RoPE backward (qkv output gradient) -- can write their result directly
// as fp8 (QF = 0: e4m3, 1: e5m2) instead of bf16: the value is rounded to bf16 first, then scaled by fmax / *amax_in
// (the previous step's amax: delayed scaling) and cast with saturation, i.e. bit-identical to vds_quant_fp8 of the
// bf16 result; amax_in / fmax goes to *dq_out.
// max |x| of the bf16 values is recorded PER WAVE with a plain store: amax_part[w] = maximum over the rows wave w
// handled (w < B*L); the tensor's amax is the maximum over the array, which the caller takes.  These kernels run up to
// ~10^5 one-row waves per launch, and every shared-word scheme measured worse on RMSNorm+modulate (101 us per launch
// without recording): one atomic max filtered by a read of the running value at the end of the wave 128 us (an exposed
// round trip per wave), filtered by a read at wave start 230 us (the first ~10^4 waves all see 0 and their atomics
// serialise), 64 slots with the read issued early 116-119 us (a device-coherent read crosses the XCDs' private L2s
// and, loads returning in order, holds up the modulation loads behind it).
struct QOut {
  unsigned char* q;       // [rows, ldq] fp8, row-major (the transposed copy is vds_transpose_fp8's job)
  long ldq;
  const float* amax_in;
  float* amax_part;       // f32 [B*L] or null
  float* dq_out;          // or null
};
template <int QF>
struct QState {
  float scale, fmax, mxf;
  float* part;
  __device__ __forceinline__ void init(const QOut& qo, bool writer, long wave_id) {
    fmax = fp8_fmax(QF);
    const float am = *qo.amax_in;
    part = qo.amax_part ? qo.amax_part + wave_id : nullptr;
    scale = am > 0.f ? fmax / am : 1.0f;
    mxf = 0.f;
    if (writer && qo.dq_out) *qo.dq_out = am > 0.f ? am / fmax : 1.0f;
  }
  // 8 results -> 8 fp8 bytes
  __device__ __forceinline__ u32x2 cvt(const float (&o)[8]) {
    const u32x4 pk = pack8(o);
    float v[8];
    unpack8(pk, v);
    // (these kernels sit close to VALU-bound with the conversion added: v_max3 with |.| modifiers and v_med3 keep it short)
#pragma unroll
    for (int e = 0; e < 4; ++e) mxf = __builtin_fmaxf(mxf, __builtin_fmaxf(__builtin_fabsf(v[2 * e]), __builtin_fabsf(v[2 * e + 1])));
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = __builtin_amdgcn_fmed3f(v[e] * scale, -fmax, fmax);
    return u32x2{fp8_cvt4<QF>(v[0], v[1], v[2], v[3]), fp8_cvt4<QF>(v[4], v[5], v[6], v[7])};
  }
  __device__ __forceinline__ void finish() {  // once per wave, at its end
    if (!part) return;
    const float m = wave_max(mxf);
    if ((threadIdx.x & 63) == 0) *part = m;
  }
};

// ------------------------------------------------------------------ RMSNorm + modulate ---
// model.py:34-41 (RMSNorm, fp32 statistics, eps 1e-6) fused with model.py:123/144/164/389
// (norm_x*(1+scale)+shift).  One wave per token row, the row stays in registers; each lane
// owns 16-byte chunks lane, lane+64, ...  NC = chunks per lane (D <= 512*NC).
template <int NC, int QF = -1>
__global__ __launch_bounds__(256) void rmsnorm_mod_fwd_kernel(const bf16_t* x, long ldx, const bf16_t* w,
                                                              const float* mod, long ldmod, int shift_col,
                                                              int scale_col, bf16_t* y, long ldy, float* rstd,
                                                              int B, int L, int D, float eps, QOut qo) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long row = (long)blockIdx.x * 4 + wave;
  if (row >= (long)B * L) return;
  QState<(QF < 0 ? 0 : QF)> qs;
  if constexpr (QF >= 0) qs.init(qo, blockIdx.x == 0 && threadIdx.x == 0, row);
  const int b = (int)(row / L);
  const int nch = D >> 3;
  u32x4 raw[NC];  // the row stays packed between the two passes (the fp8 form needs the registers: 8 waves / SIMD)
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int c = lane + 64 * i;
    raw[i] = u32x4{0u, 0u, 0u, 0u};
    if (c < nch) {
      raw[i] = ld_stream<8>(x + row * ldx + c * 8);
      float v[8];
      unpack8(raw[i], v);
#pragma unroll
      for (int e = 0; e < 8; ++e) ss += v[e] * v[e];
    }
  }
  ss = wave_sum(ss);
  const float r = rsqrtf(ss / (float)D + eps);
  if (lane == 0) rstd[row] = r;
  const float* mrow = mod + (long)b * ldmod;
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int c = lane + 64 * i;
    if (c < nch) {
      float sh[8], sc[8], o[8], wv[8], v[8];
      load8f(mrow + shift_col + c * 8, sh);
      load8f(mrow + scale_col + c * 8, sc);
      if (w) unpack8(*reinterpret_cast<const u32x4*>(w + c * 8), wv);
      unpack8(raw[i], v);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float xn = v[e] * r;
        if (w) xn *= wv[e];
        o[e] = xn * (1.0f + sc[e]) + sh[e];
      }
      if constexpr (QF >= 0) *reinterpret_cast<u32x2*>(qo.q + row * qo.ldq + c * 8) = qs.cvt(o);
      else st_stream<8>(y + row * ldy + c * 8, pack8(o));
    }
  }
  if constexpr (QF >= 0) qs.finish();
}

// Four rows per wave (round 5): D = 1152 is 144 chunks of 16 bytes = 2.25 wave-instructions per row in the mapping above,
// i.e. the third load / store of every row runs with 16 of 64 lanes (measured, tools/bench_rmsnorm_width.py: 4.67 TB/s at
// D = 1152 against 5.39 at D = 1024 and 5.31 at D = 1536).  Here each 16-lane group of a wave owns one of four
// consecutive rows and a lane the chunks lane16 + 16 i, i < NCQ = D / 128: every instruction has all 64 lanes at work
// (9 instead of 12 instructions per stream for four rows), the row statistics reduce over 16 lanes.  Same arithmetic
// per element; the sum of squares is taken in a different order (16 partial sums of 9 x 8 terms instead of 64 of 3 x 8).
// DiT-S / B / XL widths: D = 384, 768, 1152 (NCQ = 3, 6, 9).
template <int NCQ, int QF = -1>
__global__ __launch_bounds__(256) void rmsnorm_mod_fwd_q4_kernel(const bf16_t* x, long ldx, const bf16_t* w,
                                                                 const float* mod, long ldmod, int shift_col,
                                                                 int scale_col, bf16_t* y, long ldy, float* rstd,
                                                                 int B, int L, float eps, QOut qo) {
  constexpr int D = NCQ * 128;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l16 = lane & 15;
  const long row0 = ((long)blockIdx.x * 4 + wave) * 4;  // first of this wave's four rows
  const long rows = (long)B * L;
  if (row0 >= rows) return;
  QState<(QF < 0 ? 0 : QF)> qs;
  if constexpr (QF >= 0) qs.init(qo, blockIdx.x == 0 && threadIdx.x == 0, row0);
  const long row = row0 + (lane >> 4);
  const bool live = row < rows;
  const long rowc = live ? row : rows - 1;  // (idle groups of the last wave compute on a valid row and store nothing)
  const int b = (int)(rowc / L);
  u32x4 raw[NCQ];
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < NCQ; ++i) {
    raw[i] = ld_stream<8>(x + rowc * ldx + (l16 + 16 * i) * 8);
    float v[8];
    unpack8(raw[i], v);
#pragma unroll
    for (int e = 0; e < 8; ++e) ss += v[e] * v[e];
  }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);  // over the 16 lanes of the row
  const float r = rsqrtf(ss / (float)D + eps);
  if (l16 == 0 && live) rstd[row] = r;
  const float* mrow = mod + (long)b * ldmod;
#pragma unroll
  for (int i = 0; i < NCQ; ++i) {
    const int c = l16 + 16 * i;
    float sh[8], sc[8], o[8], wv[8], v[8];
    load8f(mrow + shift_col + c * 8, sh);
    load8f(mrow + scale_col + c * 8, sc);
    if (w) unpack8(*reinterpret_cast<const u32x4*>(w + c * 8), wv);
    unpack8(raw[i], v);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float xn = v[e] * r;
      if (w) xn *= wv[e];
      o[e] = xn * (1.0f + sc[e]) + sh[e];
    }
    if constexpr (QF >= 0) {
      const u32x2 q8 = qs.cvt(o);  // (every lane converts: the wave's amax covers its live rows only, see below)
      if (live) *reinterpret_cast<u32x2*>(qo.q + row * qo.ldq + c * 8) = q8;
    } else {
      if (live) st_stream<8>(y + row * ldy + c * 8, pack8(o));
    }
  }
  if constexpr (QF >= 0) {
    if (!live) qs.mxf = 0.f;  // an idle group repeated the last row: nothing of its own to record
    qs.finish();
  }
}

// (Measured negative, round 2: requesting the next row's dy / x / dres one row ahead in registers -- 36 more VGPRs --
// dropped the kernel below its 3 waves per SIMD and tripled its time; occupancy, not an explicit prefetch, is what
// keeps the loads in flight here.)
// Backward.  grid = (row-groups, B); each wave walks rows of ONE sample so that the
// d(shift), d(scale) sums over L stay in registers; one atomicAdd per column per block.
// dy and x stay PACKED (bf16) between the two passes over a row and the norm-weight terms are
// compiled out when there is no weight (the reference's default), which keeps the kernel at
// >= 4 waves per SIMD -- it is HBM-bound and needs the loads in flight.
template <int NC, bool HAS_W>
// (round 6: at D = 1152 (NC = 3) three waves per SIMD spilled 20 registers into the row loop; two waves, unspilled, run
// 3-4 % faster: 185 -> 180 us with a residual gradient, 150 -> 144 us without, tools/bench_row_kernels.py.  Requesting
// the next row's streams before working on the current one needs 72 more registers and spills again: measured, not kept.)
__global__ __launch_bounds__(256, ((NC <= 2 && !HAS_W) ? 3 : 2)) void rmsnorm_mod_bwd_kernel(const bf16_t* dy, long lddy, const bf16_t* x, long ldx,
                                                              const bf16_t* w, const float* mod, long ldmod,
                                                              int shift_col, int scale_col, const float* rstd,
                                                              const bf16_t* dres, long lddres, bf16_t* dx, long lddx,
                                                              float* dmod, float* dw, int B, int L, int D,
                                                              int rows_per_block, float* det) {
  __shared__ float red[4][64 * NC * 8 + 8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.y;
  const int nch = D >> 3;
  const int l0 = blockIdx.x * rows_per_block;
  const int l1 = min(L, l0 + rows_per_block);
  float sc1[NC][8], a_shift[NC][8], a_scale[NC][8];
  float wv[HAS_W ? NC : 1][8], a_w[HAS_W ? NC : 1][8];
  const float* mrow = mod + (long)b * ldmod;
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int c = lane + 64 * i;
#pragma unroll
    for (int e = 0; e < 8; ++e) { a_shift[i][e] = 0.f; a_scale[i][e] = 0.f; sc1[i][e] = 0.f; }
    if constexpr (HAS_W) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { a_w[i][e] = 0.f; wv[i][e] = 1.f; }
    }
    if (c < nch) {
      load8f(mrow + scale_col + c * 8, sc1[i]);
#pragma unroll
      for (int e = 0; e < 8; ++e) sc1[i][e] += 1.0f;
      if constexpr (HAS_W) unpack8(*reinterpret_cast<const u32x4*>(w + c * 8), wv[i]);
    }
  }
  for (int l = l0 + wave; l < l1; l += 4) {
    const long row = (long)b * L + l;
    const float r = rstd[row];
    u32x4 pdy[NC], px[NC], pres[NC];
    float dot = 0.f;
    // all three streams of the row are requested together: the residual gradient is only needed after the row
    // reduction, and loading it there would expose a second HBM round trip per row
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const int c = lane + 64 * i;
      if (c < nch) {
        pdy[i] = ld_stream<1>(dy + row * lddy + c * 8);
        px[i] = ld_stream<1>(x + row * ldx + c * 8);
        pres[i] = dres ? ld_stream<1>(dres + row * lddres + c * 8) : u32x4{0u, 0u, 0u, 0u};
      } else {
        pdy[i] = u32x4{0u, 0u, 0u, 0u};
        px[i] = u32x4{0u, 0u, 0u, 0u};
        pres[i] = u32x4{0u, 0u, 0u, 0u};
      }
    }
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      float dyv[8], xv[8];
      unpack8(pdy[i], dyv);
      unpack8(px[i], xv);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float xhat = xv[e] * r;
        a_shift[i][e] += dyv[e];
        float t = dyv[e] * sc1[i][e];  // d/d(xhat*w)
        if constexpr (HAS_W) {
          a_scale[i][e] += dyv[e] * xhat * wv[i][e];
          a_w[i][e] += t * xhat;
          t *= wv[i][e];                // d/d xhat
        } else {
          a_scale[i][e] += dyv[e] * xhat;
        }
        dot += t * xhat;
      }
    }
    dot = wave_sum(dot) / (float)D;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const int c = lane + 64 * i;
      if (c < nch) {
        float o[8], dyv[8], xv[8];
        unpack8(pres[i], o);  // zeros without a residual gradient
        unpack8(pdy[i], dyv);
        unpack8(px[i], xv);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float g = dyv[e] * sc1[i][e];
          if constexpr (HAS_W) g *= wv[i][e];
          o[e] += r * (g - xv[e] * r * dot);
        }
        st_stream<2>(dx + row * lddx + c * 8, pack8(o));
      }
    }
  }
  // block reduction of the column sums, one pass each through LDS
  float* drow = dmod + (long)b * ldmod;
  for (int pass = 0; pass < (HAS_W ? 3 : 2); ++pass) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NC; ++i)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float v = pass == 0 ? a_shift[i][e] : a_scale[i][e];
        if constexpr (HAS_W) { if (pass == 2) v = a_w[i][e]; }
        red[wave][(i * 64 + lane) * 8 + e] = v;
      }
    __syncthreads();
    for (int idx = threadIdx.x; idx < NC * 64 * 8; idx += 256) {
      const int i = idx / 512, rem = idx % 512, ln = rem / 8, e = rem % 8;
      const int col = (ln + 64 * i) * 8 + e;
      if (col < D) {
        const float sm = red[0][idx] + red[1][idx] + red[2][idx] + red[3][idx];
        // deterministic mode: the workgroup's partial goes to det[pass][b][blockIdx.x][D]; det_finish_kernel sums in order
        if (det) det[(((long)pass * gridDim.y + b) * gridDim.x + blockIdx.x) * D + col] = sm;
        else if (pass == 0) atomicAdd(drow + shift_col + col, sm);
        else if (pass == 1) atomicAdd(drow + scale_col + col, sm);
        else atomicAdd(dw + col, sm);
      }
    }
  }
}

// ------------------------------------------------------------------------ gate backward ---
// x_new = x + y*gate (model.py:139,160,165): dy = dx_new*gate, dgate = sum_l dx_new*y,
// dbias = sum_{b,l} dy.  Same sample-per-blockIdx.y structure as above.
template <int NC, int QF = -1>
__global__ __launch_bounds__(256) void gate_bwd_kernel(const bf16_t* dxn, long lddxn, const bf16_t* y, long ldy,
                                                       const float* mod, long ldmod, int gate_col, bf16_t* dy,
                                                       long lddy, float* dmod, float* dbias, int B, int L, int D,
                                                       int rows_per_block, QOut qo, float* det) {
  __shared__ float red[4][64 * NC * 8 + 8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.y;
  const int nch = D >> 3;
  const int l0 = blockIdx.x * rows_per_block;
  const int l1 = min(L, l0 + rows_per_block);
  float gt[NC][8], a_g[NC][8], a_b[NC][8];
  QState<(QF < 0 ? 0 : QF)> qs;
  if constexpr (QF >= 0)
    qs.init(qo, blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0, ((long)blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave);
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int c = lane + 64 * i;
#pragma unroll
    for (int e = 0; e < 8; ++e) { a_g[i][e] = 0.f; a_b[i][e] = 0.f; gt[i][e] = 0.f; }
    if (c < nch) load8f(mod + (long)b * ldmod + gate_col + c * 8, gt[i]);
  }
  for (int l = l0 + wave; l < l1; l += 4) {
    const long row = (long)b * L + l;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const int c = lane + 64 * i;
      if (c < nch) {
        float d[8], yv[8], o[8];
        unpack8(ld_stream<4>(dxn + row * lddxn + c * 8), d);
        unpack8(ld_stream<32>(y + row * ldy + c * 8), yv);  // (the saved forward result: bit 32)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          a_g[i][e] += d[e] * yv[e];
          o[e] = d[e] * gt[i][e];
          a_b[i][e] += o[e];
        }
        if constexpr (QF >= 0) *reinterpret_cast<u32x2*>(qo.q + row * qo.ldq + c * 8) = qs.cvt(o);
        else st_stream<4>(dy + row * lddy + c * 8, pack8(o));
      }
    }
  }
  if constexpr (QF >= 0) qs.finish();
  for (int pass = 0; pass < 2; ++pass) {
    if (pass == 1 && !dbias) break;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NC; ++i)
#pragma unroll
      for (int e = 0; e < 8; ++e) red[wave][(i * 64 + lane) * 8 + e] = pass == 0 ? a_g[i][e] : a_b[i][e];
    __syncthreads();
    for (int idx = threadIdx.x; idx < NC * 64 * 8; idx += 256) {
      const int i = idx / 512, rem = idx % 512, ln = rem / 8, e = rem % 8;
      const int col = (ln + 64 * i) * 8 + e;
      if (col < D) {
        const float s = red[0][idx] + red[1][idx] + red[2][idx] + red[3][idx];
        if (det) det[(((long)pass * gridDim.y + b) * gridDim.x + blockIdx.x) * D + col] = s;  // (as rmsnorm_mod_bwd_kernel)
        else if (pass == 0) atomicAdd(dmod + (long)b * ldmod + gate_col + col, s);
        else atomicAdd(dbias + col, s);
      }
    }
  }
}

// column sums of a bf16 matrix (bias gradients): block = 64 column-chunks x 4 row lanes
// rps > 0: only rows r with (r % rps) >= roff are summed (token rows of the [B*L] buffer, register rows skipped)
__global__ __launch_bounds__(256) void colsum_kernel(const bf16_t* x, long ldx, float* out, int M, int N,
                                                     int rows_per_block, int rps, int roff, float* det) {
  __shared__ float red[4][64 * 8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  const int r0 = blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
  float a[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) a[e] = 0.f;
  if (c * 8 < N)
    for (int r = r0 + wave; r < r1; r += 4) {
      if (rps > 0 && (r % rps) < roff) continue;
      float v[8];
      unpack8(*reinterpret_cast<const u32x4*>(x + (long)r * ldx + c * 8), v);
#pragma unroll
      for (int e = 0; e < 8; ++e) a[e] += v[e];
    }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[wave][lane * 8 + e] = a[e];
  __syncthreads();
  for (int idx = threadIdx.x; idx < 512; idx += 256) {
    const int col = blockIdx.x * 512 + idx;
    if (col >= N) continue;
    const float sm = red[0][idx] + red[1][idx] + red[2][idx] + red[3][idx];
    if (det) det[(long)blockIdx.y * N + col] = sm;
    else atomicAdd(out + col, sm);
  }
}

// ---- deterministic mode (vds_set_deterministic): the second stage of every column / scalar sum -------------------------
// target[g * tstride + c] += sum over p < n_parts of part[(g * n_parts + p) * width + c], p in index order (8 loads in
// flight, the adds strictly sequential): the same words every run.
__global__ __launch_bounds__(256) void det_finish_kernel(const float* part, int n_parts, int width, float* target, long tstride,
                                                         int groups) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)groups * width) return;
  const int g = (int)(i / width), c = (int)(i % width);
  const float* p = part + (long)g * n_parts * width + c;
  float v = target[(long)g * tstride + c];
  int k = 0;
  for (; k + 8 <= n_parts; k += 8) {
    float t[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) t[j] = p[(long)(k + j) * width];
#pragma unroll
    for (int j = 0; j < 8; ++j) v += t[j];
  }
  for (; k < n_parts; ++k) v += p[(long)k * width];
  target[(long)g * tstride + c] = v;
}
// a scalar with many partials (the lambda gradient: one per workgroup of the RoPE backward): thread t sums partials t,
// t + 256, .. in order, then a fixed butterfly over the lanes and a fixed sum over the four waves
__global__ __launch_bounds__(256) void det_finish_scalar_kernel(const float* part, int n_parts, float* target) {
  __shared__ float red[4];
  float v = 0.f;
  for (int k = threadIdx.x; k < n_parts; k += 256) v += part[k];
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) target[0] += (red[0] + red[1]) + (red[2] + red[3]);
}

// ------------------------------------------------- qkv split + RoPE + residual-V (fwd) ---
// model.py:126-134,266-275.  One thread = 4 elements d..d+3 of the first half of one head of
// one token and their rotation partners d+hd/2.. (hd/2 is a multiple of 4 for 64/72/128).
__global__ __launch_bounds__(256) void qkv_rope_fwd_kernel(const bf16_t* qkv, const float* cosb, const float* sinb,
                                                           const bf16_t* v0, const bf16_t* lamp, bf16_t* q,
                                                           bf16_t* k, bf16_t* v, int B, int L, int H, int hd,
                                                           int hdp) {
  const int upt = H * (hd >> 3);
  const long gid = (long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (long)B * L * upt) return;
  const long tok = gid / upt;
  const int u = (int)(gid % upt);
  const int hq = hd >> 3, hh = u / hq, i = u % hq, half = hd >> 1;
  const int b = (int)(tok / L), l = (int)(tok % L);
  const int D = H * hd;
  const bf16_t* src = qkv + tok * 3 * D + hh * hd + 4 * i;
  const long dst = (((long)b * H + hh) * L + l) * hdp + 4 * i;
  const f32x4 c4 = *reinterpret_cast<const f32x4*>(cosb + (long)l * half + 4 * i);
  const f32x4 s4 = *reinterpret_cast<const f32x4*>(sinb + (long)l * half + 4 * i);
#pragma unroll
  for (int which = 0; which < 2; ++which) {
    const u32x2 lo = *reinterpret_cast<const u32x2*>(src + which * D);
    const u32x2 hi = *reinterpret_cast<const u32x2*>(src + which * D + half);
    const float x1[4] = {bflo(lo[0]), bfhi(lo[0]), bflo(lo[1]), bfhi(lo[1])};
    const float x2[4] = {bflo(hi[0]), bfhi(hi[0]), bflo(hi[1]), bfhi(hi[1])};
    float y1[4], y2[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      y1[e] = x1[e] * c4[e] + x2[e] * s4[e];
      y2[e] = x2[e] * c4[e] - x1[e] * s4[e];
    }
    bf16_t* o = (which == 0 ? q : k) + dst;
    u32x2 w1 = {pack_bf2(y1[0], y1[1]), pack_bf2(y1[2], y1[3])};
    u32x2 w2 = {pack_bf2(y2[0], y2[1]), pack_bf2(y2[2], y2[3])};
    *reinterpret_cast<u32x2*>(o) = w1;
    *reinterpret_cast<u32x2*>(o + half) = w2;
  }
  {
    const u32x2 lo = *reinterpret_cast<const u32x2*>(src + 2 * D);
    const u32x2 hi = *reinterpret_cast<const u32x2*>(src + 2 * D + half);
    u32x2 w1 = lo, w2 = hi;
    if (v0) {
      const float lam = bf2f(*lamp);
      const u32x2 a = *reinterpret_cast<const u32x2*>(v0 + dst);
      const u32x2 c = *reinterpret_cast<const u32x2*>(v0 + dst + half);
      // reference (bf16 tensors): lam*v rounds, (1-lam) rounds, (1-lam)*v0 rounds, sum rounds
      const float oml = bf2f(f2bf(1.0f - lam));
      auto mix = [&](unsigned vr, unsigned v0r) {
        const float a0 = bf2f(f2bf(lam * bflo(vr))) + bf2f(f2bf(oml * bflo(v0r)));
        const float a1 = bf2f(f2bf(lam * bfhi(vr))) + bf2f(f2bf(oml * bfhi(v0r)));
        return pack_bf2(a0, a1);
      };
      w1[0] = mix(lo[0], a[0]); w1[1] = mix(lo[1], a[1]);
      w2[0] = mix(hi[0], c[0]); w2[1] = mix(hi[1], c[1]);
    }
    *reinterpret_cast<u32x2*>(v + dst) = w1;
    *reinterpret_cast<u32x2*>(v + dst + half) = w2;
  }
  // pad columns hd..hdp of this (token, head): zero, except K[hd] = K[hd+1] = V[hd] = V[hd+4] = 1.0 when
  // there are >= 8 pad columns -- the attention kernels (kv_pad_ones) use the K ones columns to add a
  // per-query constant (-max forward, -lse as a bf16 hi/lo pair backward) to the scores inside the QK^T
  // MFMA, and the V ones columns to get the softmax row sums out of the PV MFMA (forward) and to
  // subtract delta inside the dO V^T MFMA (backward).  Consumers without the flag ignore the pad.
  const int npad = (hdp - hd) >> 2;
  if (i < npad) {
    const long pd = (((long)b * H + hh) * L + l) * hdp + hd + 4 * i;
    const u32x2 z = {0u, 0u};
    const u32x2 one = {0x3f80u, 0u}, one2 = {0x3f803f80u, 0u};
    const bool ones = (hdp - hd) >= 8;
    *reinterpret_cast<u32x2*>(q + pd) = z;
    *reinterpret_cast<u32x2*>(k + pd) = (ones && i == 0) ? one2 : z;
    *reinterpret_cast<u32x2*>(v + pd) = (ones && i < 2) ? one : z;
  }
}

// Token-tile form of the kernel above (rope_stage.h): T consecutive tokens per workgroup, qkv rows copied to LDS by
// LDS-DMA, q / k rotated in place there, every global access 16 bytes wide and every (tensor, head) written as a run of T
// complete head rows.  Same arithmetic and rounding points.
template <int HD, int HDP, int T>
__global__ __launch_bounds__(256) void qkv_rope_fwd_tile_kernel(const bf16_t* qkv, const float* cosb, const float* sinb,
                                                                const bf16_t* v0, const bf16_t* lamp, bf16_t* q,
                                                                bf16_t* k, bf16_t* v, long ntok, int L, int H) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int D = H * HD, row_b = 6 * D;
  const long tok0 = (long)blockIdx.x * T;
  const int nt = (int)min((long)T, ntok - tok0);
  ropestage::issue_rows(qkv + tok0 * 3 * D, smem, nt, row_b, wave, lane);
  float lam = 0.f, oml = 0.f;
  if (v0) {
    lam = bf2f(*lamp);
    oml = bf2f(f2bf(1.0f - lam));
  }
  VDS_WAIT_VM(0);
  __syncthreads();
  ropestage::rotate_rows<HD>(smem, cosb, sinb, tok0, nt, L, H, row_b, tid);
  __syncthreads();
  constexpr int CPR = HDP >> 3, DCH = HD >> 3;
  constexpr bool ONES = (HDP - HD) >= 8;
  const ropestage::Div by_nt((unsigned)nt), by_h((unsigned)H);
  const int b0 = (int)(tok0 / L), l0 = (int)(tok0 % L);
  const int nch = nt * 3 * H * CPR;
  for (int u = tid; u < nch; u += 256) {
    const int c = u % CPR;
    unsigned t, hh;
    const unsigned r = by_nt.div((unsigned)(u / CPR), t);
    const unsigned tensor = by_h.div(r, hh);
    int l = l0 + (int)t, b = b0;
    if (l >= L) { l -= L; ++b; }
    const long dst = (((long)b * H + hh) * L + l) * HDP + 8 * c;
    u32x4 w = {0u, 0u, 0u, 0u};
    if (c < DCH) {
      w = *reinterpret_cast<const u32x4*>(smem + t * row_b + ((tensor * H + hh) * HD + 8 * c) * 2);
      if (tensor == 2 && v0) w = ropestage::mix_v(w, *reinterpret_cast<const u32x4*>(v0 + dst), lam, oml);
    } else if (ONES && c == DCH) {
      // K[hd] = K[hd+1] = 1.0, V[hd] = V[hd+4] = 1.0 (see the pad comment of the kernel above)
      if (tensor == 1) w[0] = 0x3f803f80u;
      if (tensor == 2) { w[0] = 0x3f80u; w[2] = 0x3f80u; }
    }
    *reinterpret_cast<u32x4*>((tensor == 0 ? q : tensor == 1 ? k : v) + dst) = w;
  }
}

// stand-alone apply_rotary_emb (model.py:266-275): y1 = x1 c + x2 s, y2 = x2 c - x1 s on the two halves of every
// head row, fp32 math, bf16 out; inverse = the transposed rotation (its backward).  One thread = 4 + 4 elements.
__global__ __launch_bounds__(256) void rope_apply_kernel(const bf16_t* x, long x_sb, long x_sh, long x_sl,
                                                         const float* cosb, const float* sinb, bf16_t* y, long y_sb,
                                                         long y_sh, long y_sl, int B, int H, int L, int hd,
                                                         int inverse) {
  const int hq = hd >> 3, half = hd >> 1;
  const long gid = (long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (long)B * H * L * hq) return;
  const int i = (int)(gid % hq);
  const long row = gid / hq;
  const int l = (int)(row % L), h = (int)((row / L) % H), b = (int)(row / ((long)L * H));
  const bf16_t* src = x + b * x_sb + h * x_sh + l * x_sl + 4 * i;
  bf16_t* dst = y + b * y_sb + h * y_sh + l * y_sl + 4 * i;
  const f32x4 c4 = *reinterpret_cast<const f32x4*>(cosb + (long)l * half + 4 * i);
  f32x4 s4 = *reinterpret_cast<const f32x4*>(sinb + (long)l * half + 4 * i);
  if (inverse) s4 = -s4;
  const u32x2 lo = *reinterpret_cast<const u32x2*>(src);
  const u32x2 hi = *reinterpret_cast<const u32x2*>(src + half);
  const float x1[4] = {bflo(lo[0]), bfhi(lo[0]), bflo(lo[1]), bfhi(lo[1])};
  const float x2[4] = {bflo(hi[0]), bfhi(hi[0]), bflo(hi[1]), bfhi(hi[1])};
  float y1[4], y2[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    y1[e] = x1[e] * c4[e] + x2[e] * s4[e];
    y2[e] = x2[e] * c4[e] - x1[e] * s4[e];
  }
  const u32x2 w1 = {pack_bf2(y1[0], y1[1]), pack_bf2(y1[2], y1[3])};
  const u32x2 w2 = {pack_bf2(y2[0], y2[1]), pack_bf2(y2[2], y2[3])};
  *reinterpret_cast<u32x2*>(dst) = w1;
  *reinterpret_cast<u32x2*>(dst + half) = w2;
}

// backward of the above
__global__ __launch_bounds__(256) void qkv_rope_bwd_kernel(const bf16_t* dq, const bf16_t* dk, const bf16_t* dv,
                                                           const float* cosb, const float* sinb,
                                                           const bf16_t* qkv_raw, const bf16_t* v0,
                                                           const bf16_t* lamp, float* dv0_acc, float* dlam,
                                                           bf16_t* dqkv, int mix, int add_dv0, int B, int L, int H,
                                                           int hd, int hdp, float* det) {
  __shared__ float red[4];
  const int upt = H * (hd >> 3);
  const long gid = (long)blockIdx.x * 256 + threadIdx.x;
  float dl = 0.f;
  if (gid < (long)B * L * upt) {
    const long tok = gid / upt;
    const int u = (int)(gid % upt);
    const int hq = hd >> 3, hh = u / hq, i = u % hq, half = hd >> 1;
    const int b = (int)(tok / L), l = (int)(tok % L);
    const int D = H * hd;
    bf16_t* dstp = dqkv + tok * 3 * D + hh * hd + 4 * i;
    const long src = (((long)b * H + hh) * L + l) * hdp + 4 * i;
    const f32x4 c4 = *reinterpret_cast<const f32x4*>(cosb + (long)l * half + 4 * i);
    const f32x4 s4 = *reinterpret_cast<const f32x4*>(sinb + (long)l * half + 4 * i);
#pragma unroll
    for (int which = 0; which < 2; ++which) {
      const bf16_t* g = (which == 0 ? dq : dk) + src;
      const u32x2 lo = *reinterpret_cast<const u32x2*>(g);
      const u32x2 hi = *reinterpret_cast<const u32x2*>(g + half);
      const float g1[4] = {bflo(lo[0]), bfhi(lo[0]), bflo(lo[1]), bfhi(lo[1])};
      const float g2[4] = {bflo(hi[0]), bfhi(hi[0]), bflo(hi[1]), bfhi(hi[1])};
      float d1[4], d2[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        d1[e] = g1[e] * c4[e] - g2[e] * s4[e];
        d2[e] = g1[e] * s4[e] + g2[e] * c4[e];
      }
      u32x2 w1 = {pack_bf2(d1[0], d1[1]), pack_bf2(d1[2], d1[3])};
      u32x2 w2 = {pack_bf2(d2[0], d2[1]), pack_bf2(d2[2], d2[3])};
      *reinterpret_cast<u32x2*>(dstp + which * D) = w1;
      *reinterpret_cast<u32x2*>(dstp + which * D + half) = w2;
    }
    {
      const u32x2 lo = *reinterpret_cast<const u32x2*>(dv + src);
      const u32x2 hi = *reinterpret_cast<const u32x2*>(dv + src + half);
      float g[8] = {bflo(lo[0]), bfhi(lo[0]), bflo(lo[1]), bfhi(lo[1]), bflo(hi[0]), bfhi(hi[0]), bflo(hi[1]), bfhi(hi[1])};
      float* acc1 = dv0_acc + src;
      float* acc2 = dv0_acc + src + half;
      if (mix) {
        const float lam = bf2f(*lamp);
        const float oml = 1.0f - lam;
        const u32x2 r1 = *reinterpret_cast<const u32x2*>(qkv_raw + tok * 3 * D + 2 * D + hh * hd + 4 * i);
        const u32x2 r2 = *reinterpret_cast<const u32x2*>(qkv_raw + tok * 3 * D + 2 * D + hh * hd + 4 * i + half);
        const u32x2 a1 = *reinterpret_cast<const u32x2*>(v0 + src);
        const u32x2 a2 = *reinterpret_cast<const u32x2*>(v0 + src + half);
        const float vr[8] = {bflo(r1[0]), bfhi(r1[0]), bflo(r1[1]), bfhi(r1[1]), bflo(r2[0]), bfhi(r2[0]), bflo(r2[1]), bfhi(r2[1])};
        const float vz[8] = {bflo(a1[0]), bfhi(a1[0]), bflo(a1[1]), bfhi(a1[1]), bflo(a2[0]), bfhi(a2[0]), bflo(a2[1]), bfhi(a2[1])};
        if (mix == 1) {  // (mix == 2: the (1 - lambda) dv terms of all blocks are summed later, vds_dv0_reduce)
          f32x4 o1 = *reinterpret_cast<f32x4*>(acc1), o2 = *reinterpret_cast<f32x4*>(acc2);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            if (e < 4) o1[e] += oml * g[e]; else o2[e - 4] += oml * g[e];
          }
          *reinterpret_cast<f32x4*>(acc1) = o1;
          *reinterpret_cast<f32x4*>(acc2) = o2;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          dl += g[e] * (vr[e] - vz[e]);
          g[e] *= lam;
        }
      } else if (add_dv0) {
        const f32x4 o1 = *reinterpret_cast<const f32x4*>(acc1), o2 = *reinterpret_cast<const f32x4*>(acc2);
#pragma unroll
        for (int e = 0; e < 4; ++e) { g[e] += o1[e]; g[e + 4] += o2[e]; }
      }
      u32x2 w1 = {pack_bf2(g[0], g[1]), pack_bf2(g[2], g[3])};
      u32x2 w2 = {pack_bf2(g[4], g[5]), pack_bf2(g[6], g[7])};
      *reinterpret_cast<u32x2*>(dstp + 2 * D) = w1;
      *reinterpret_cast<u32x2*>(dstp + 2 * D + half) = w2;
    }
  }
  if (mix) {
    dl = wave_sum(dl);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = dl;
    __syncthreads();
    if (threadIdx.x == 0) {
      if (det) det[blockIdx.x] = red[0] + red[1] + red[2] + red[3];  // deterministic mode: det_finish_scalar_kernel sums in order
      else atomicAdd(dlam, red[0] + red[1] + red[2] + red[3]);
    }
  }
}

// ------------------------------ backward of qkv split + RoPE + residual-V, wave-per-token form -------
// The kernel above moves 8 bytes per lane and access (the rotation partner of element d is d + hd/2, and
// hd/2 = 36 is not a multiple of 8 for head_dim 72); 8-byte global accesses run at 0.54-0.70x the rate of
// 16-byte ones (MI355X_MICROARCH.md).  Here one wave owns one token: every global access is 16 bytes per lane --
// the token's [3D] output row as 3D/8 consecutive chunks, the head-major gradient rows as (head, chunk) items --
// and the rotation partners meet in a wave-private LDS copy of the token's dq and dk rows (5 KB per wave).
// Item = chunk c = h * hd/8 + j of a section; lane l handles items l, l + 64, ... (NI of them, D <= 512 * NI).
// Measured on the DiT-XL step (B = 12, same box): 18.8 ms per step against 22.2 ms.  The same restructuring of
// the FORWARD kernel was slower (14.3 against 13.4 ms: its scattered 144-byte head-row stores gain nothing from
// 16-byte lanes and it pays the LDS exchange) and is not in the tree.
template <int NI, int QF = -1>
__global__ __launch_bounds__(256) void qkv_rope_bwd_tok_kernel(const bf16_t* dq, const bf16_t* dk, const bf16_t* dv,
                                                               const float* cosb, const float* sinb,
                                                               const bf16_t* qkv_raw, const bf16_t* v0,
                                                               const bf16_t* lamp, float* dv0_acc, float* dlam,
                                                               bf16_t* dqkv, int mix, int add_dv0, int B, int L,
                                                               int H, int hd, int hdp, QOut qo, float* det) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ float red[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long tok = (long)blockIdx.x * 4 + wave;
  const bool live = tok < (long)B * L;
  float dl = 0.f;
  QState<(QF < 0 ? 0 : QF)> qs;
  if constexpr (QF >= 0) qs.init(qo, blockIdx.x == 0 && threadIdx.x == 0, tok);
  if (live) {
    const int b = (int)(tok / L), l = (int)(tok % L);
    const int D = H * hd, nch = D >> 3, cph = hd >> 3, half = hd >> 1;
    const int wstride = 2 * D * 2 + 2 * half * 4;
    char* wq = smem + wave * wstride;
    char* wk = wq + D * 2;
    float* wc = reinterpret_cast<float*>(wk + D * 2);
    float* wsn = wc + half;
    bf16_t* dst = dqkv + tok * 3 * D;
    u32x4 rq[NI], rk[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int c = lane + 64 * i;
      if (c < nch) {
        const int hh = c / cph, j = c % cph;
        const long so = (((long)b * H + hh) * L + l) * hdp + 8 * j;
        rq[i] = ld_stream<16>(dq + so);
        rk[i] = ld_stream<16>(dk + so);
        *reinterpret_cast<u32x4*>(wq + c * 16) = rq[i];
        *reinterpret_cast<u32x4*>(wk + c * 16) = rk[i];
      }
    }
    if (lane < half / 4) {
      *reinterpret_cast<f32x4*>(wc + 4 * lane) = *reinterpret_cast<const f32x4*>(cosb + (long)l * half + 4 * lane);
      *reinterpret_cast<f32x4*>(wsn + 4 * lane) = *reinterpret_cast<const f32x4*>(sinb + (long)l * half + 4 * lane);
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
      const float lam = mix ? bf2f(*lamp) : 0.f, oml = 1.0f - lam;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int c = lane + 64 * i;
      if (c >= nch) continue;
      const int hh = c / cph, j = c % cph;
      const long so = (((long)b * H + hh) * L + l) * hdp + 8 * j;
      float gq[8], gk[8], oq[8], ok_[8];
      unpack8(rq[i], gq);
      unpack8(rk[i], gk);
#pragma unroll
      for (int g4 = 0; g4 < 2; ++g4) {
        const int d0 = 8 * j + 4 * g4;
        const bool first = d0 < half;
        const int pd = first ? d0 + half : d0 - half, ci = first ? d0 : d0 - half;
        const u32x2 pq = *reinterpret_cast<const u32x2*>(wq + (hh * hd + pd) * 2);
        const u32x2 pk = *reinterpret_cast<const u32x2*>(wk + (hh * hd + pd) * 2);
        const f32x4 c4 = *reinterpret_cast<const f32x4*>(wc + ci);
        const f32x4 s4 = *reinterpret_cast<const f32x4*>(wsn + ci);
        const float pqf[4] = {bflo(pq[0]), bfhi(pq[0]), bflo(pq[1]), bfhi(pq[1])};
        const float pkf[4] = {bflo(pk[0]), bfhi(pk[0]), bflo(pk[1]), bfhi(pk[1])};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          // transposed rotation: d1 = g1 c - g2 s ; d2 = g1 s + g2 c
          const float sg = first ? -s4[e] : s4[e];
          oq[4 * g4 + e] = gq[4 * g4 + e] * c4[e] + pqf[e] * sg;
          ok_[4 * g4 + e] = gk[4 * g4 + e] * c4[e] + pkf[e] * sg;
        }
      }
      unsigned char* qdst = nullptr;
      if constexpr (QF >= 0) {
        qdst = qo.q + tok * qo.ldq;
        *reinterpret_cast<u32x2*>(qdst + c * 8) = qs.cvt(oq);
        *reinterpret_cast<u32x2*>(qdst + D + c * 8) = qs.cvt(ok_);
      } else {
        *reinterpret_cast<u32x4*>(dst + c * 8) = pack8(oq);
        *reinterpret_cast<u32x4*>(dst + D + c * 8) = pack8(ok_);
      }
      float g[8];
      unpack8(ld_stream<16>(dv + so), g);
      if (mix) {
        float vr[8], vz[8];
        unpack8(*reinterpret_cast<const u32x4*>(qkv_raw + tok * 3 * D + 2 * D + c * 8), vr);
        unpack8(*reinterpret_cast<const u32x4*>(v0 + so), vz);
        if (mix == 1) {  // (mix == 2: the (1 - lambda) dv terms of all blocks are summed later, vds_dv0_reduce)
          float a[8];
          load8f(dv0_acc + so, a);
#pragma unroll
          for (int e = 0; e < 8; ++e) a[e] += oml * g[e];
          *reinterpret_cast<f32x4*>(dv0_acc + so) = f32x4{a[0], a[1], a[2], a[3]};
          *reinterpret_cast<f32x4*>(dv0_acc + so + 4) = f32x4{a[4], a[5], a[6], a[7]};
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          dl += g[e] * (vr[e] - vz[e]);
          g[e] *= lam;
        }
      } else if (add_dv0) {
        float a[8];
        load8f(dv0_acc + so, a);
#pragma unroll
        for (int e = 0; e < 8; ++e) g[e] += a[e];
      }
      if constexpr (QF >= 0) *reinterpret_cast<u32x2*>(qdst + 2 * D + c * 8) = qs.cvt(g);
      else *reinterpret_cast<u32x4*>(dst + 2 * D + c * 8) = pack8(g);
    }
  }
  if constexpr (QF >= 0)
    if (live) qs.finish();  // (wave-uniform; an idle tail wave must not overwrite entry 0)
  if (mix) {
    dl = wave_sum(dl);
    if (lane == 0) red[wave] = dl;
    __syncthreads();
    if (threadIdx.x == 0) {
      if (det) det[blockIdx.x] = red[0] + red[1] + red[2] + red[3];  // deterministic mode: det_finish_scalar_kernel sums in order
      else atomicAdd(dlam, red[0] + red[1] + red[2] + red[3]);
    }
  }
}

// ------------------------------------------------------------------ small-M linears ------
// y[b,n] = sum_k act(x[b,k]) W[n,k] + bias[n]; one wave per 4 output columns (the x chunk a lane loads is
// reused for 4 rows of W), M <= 16 rows.
// (time_embed model.py:318-322, adaLN_modulation model.py:89-91, final_modulation 339-341)
// Batched form (blockIdx.z = index into device pointer tables): the same linear layer of nb weight sets applied to
// ONE shared input x -- the adaLN modulation of every DiT block (model.py:89-94,107) in a single launch.
struct SLBatch {
  const void* const* W;       // [nb] bf16 [N,K]; null = not batched
  const void* const* bias;    // [nb] bf16 [N] or null
  float* const* dW;           // [nb] f32 [N,K]
  float* const* dbias;        // [nb] f32 [N]
  long y_stride;              // elements between the [M,N] outputs / output gradients of consecutive sets
};

template <int MB>
__global__ __launch_bounds__(256) void small_linear_fwd_kernel(const float* x, const bf16_t* W, const bf16_t* bias,
                                                               float* y, int M, int N, int K, int act_in, SLBatch bt) {
  if (bt.W) {
    W = (const bf16_t*)bt.W[blockIdx.z];
    bias = bt.bias ? (const bf16_t*)bt.bias[blockIdx.z] : nullptr;
    y += (long)blockIdx.z * bt.y_stride;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n0 = (blockIdx.x * 4 + wave) * 4;
  if (n0 >= N) return;
  float acc[4][MB];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int b = 0; b < MB; ++b) acc[i][b] = 0.f;
  for (int c = lane; c < (K >> 3); c += 64) {
    float wv[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
      unpack8(*reinterpret_cast<const u32x4*>(W + (long)min(n0 + i, N - 1) * K + c * 8), wv[i]);
#pragma unroll
    for (int b = 0; b < MB; ++b)
      if (b < M) {
        float xv[8];
        load8f(x + (long)b * K + c * 8, xv);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float xe = act_in ? silu_f(xv[e]) : xv[e];
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[i][b] += xe * wv[i][e];
        }
      }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (n0 + i >= N) break;
    const float bs = bias ? bf2f(bias[n0 + i]) : 0.f;
#pragma unroll
    for (int b = 0; b < MB; ++b)
      if (b < M) {
        const float s = wave_sum(acc[i][b]);
        if (lane == 0) y[(long)b * N + n0 + i] = s + bs;
      }
  }
}

// dW[n,k] = sum_b dy[b,n] act(x[b,k]);  dbias[n] = sum_b dy[b,n]   (thread per (4 rows n, k-chunk): the
// activated x chunk is built once and reused for the 4 rows)
__global__ __launch_bounds__(256) void small_linear_dw_kernel(const float* dy, const float* x, float* dW,
                                                              float* dbias, int M, int N, int K, int act_in, SLBatch bt) {
  if (bt.dW) {
    dW = bt.dW[blockIdx.z];
    dbias = bt.dbias ? bt.dbias[blockIdx.z] : nullptr;
    dy += (long)blockIdx.z * bt.y_stride;
  }
  const int kc = K >> 3;
  const long gid = (long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (long)((N + 3) / 4) * kc) return;
  const int n0 = (int)(gid / kc) * 4, c = (int)(gid % kc);
  float acc[4][8], sb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    sb[i] = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[i][e] = 0.f;
  }
  for (int b = 0; b < M; ++b) {
    float xv[8];
    load8f(x + (long)b * K + c * 8, xv);
    if (act_in)
#pragma unroll
      for (int e = 0; e < 8; ++e) xv[e] = silu_f(xv[e]);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float g = dy[(long)b * N + min(n0 + i, N - 1)];
      sb[i] += g;
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[i][e] += g * xv[e];
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (n0 + i >= N) break;
    float* o = dW + (long)(n0 + i) * K + c * 8;
    *reinterpret_cast<f32x4*>(o) = f32x4{acc[i][0], acc[i][1], acc[i][2], acc[i][3]};
    *reinterpret_cast<f32x4*>(o + 4) = f32x4{acc[i][4], acc[i][5], acc[i][6], acc[i][7]};
    if (c == 0 && dbias) dbias[n0 + i] = sb[i];
  }
}

// dx[b,k] += act'(x[b,k]) * sum_n dy[b,n] W[n,k].
// block = (64-column range of W: 128-byte row segments) x (range of rows n); thread = (row lane tid/8, 8-column
// chunk tid%8).  Every thread streams one 16-byte piece of W per row step (32 rows per step, unrolled), the 32 row
// lanes are summed with lane exchanges + 4 KB of LDS, and only the n-ranges meet in atomics (dx accumulates the
// modulation gradients of every block anyway).  W is read exactly once at full HBM request parallelism.
// dW / dbias, batched form (round 4): tile = 64 rows of dW x 128 columns per workgroup; the tile's slices of x (SiLU
// applied once) and dy sit in LDS, a thread owns 4 rows x 8 columns and reads them with 16-byte LDS accesses instead of
// 60 loads through L1 per 128 bytes written; dW leaves with non-temporal stores (read next by the optimizer).
template <int MB>
__global__ __launch_bounds__(256) void small_linear_dw_tile_kernel(const float* dy, const float* x, int M, int N, int K,
                                                                   int act_in, SLBatch bt) {
  __shared__ __attribute__((aligned(16))) float xs[MB][128];
  __shared__ __attribute__((aligned(16))) float dys[MB][64];
  float* dW = bt.dW[blockIdx.z];
  float* dbias = bt.dbias ? bt.dbias[blockIdx.z] : nullptr;
  dy += (long)blockIdx.z * bt.y_stride;
  const int tid = threadIdx.x;
  const int k0 = blockIdx.x * 128, n0 = blockIdx.y * 64;
  for (int i = tid; i < MB * 128; i += 256) {
    const int b = i >> 7, k = k0 + (i & 127);
    float v = (b < M && k < K) ? x[(long)b * K + k] : 0.f;
    xs[b][i & 127] = act_in ? silu_f(v) : v;
  }
  for (int i = tid; i < MB * 64; i += 256) {
    const int b = i >> 6, n = n0 + (i & 63);
    dys[b][i & 63] = (b < M && n < N) ? dy[(long)b * N + n] : 0.f;
  }
  __syncthreads();
  const int c = tid & 15, ng = tid >> 4;  // 8 columns k0 + 8c .., 4 rows n0 + 4 ng ..
  float acc[4][8], sb[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[i][e] = 0.f;
#pragma unroll
  for (int b = 0; b < MB; ++b) {
    const f32x4 x0 = *reinterpret_cast<const f32x4*>(&xs[b][8 * c]), x1 = *reinterpret_cast<const f32x4*>(&xs[b][8 * c + 4]);
    const f32x4 g = *reinterpret_cast<const f32x4*>(&dys[b][4 * ng]);
    const float xv[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      sb[i] += g[i];
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[i][e] += g[i] * xv[e];
    }
  }
  const int k = k0 + 8 * c;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int n = n0 + 4 * ng + i;
    if (n >= N || k >= K) continue;
    float* o = dW + (long)n * K + k;
    __builtin_nontemporal_store(f32x4{acc[i][0], acc[i][1], acc[i][2], acc[i][3]}, reinterpret_cast<f32x4*>(o));
    __builtin_nontemporal_store(f32x4{acc[i][4], acc[i][5], acc[i][6], acc[i][7]}, reinterpret_cast<f32x4*>(o + 4));
    if (blockIdx.x == 0 && c == 0 && dbias) dbias[n] = sb[i];
  }
}

// dx.  round 4: the workgroup's slice of dy is staged in LDS TRANSPOSED ([row n][16 samples]: three ds_read_b128 per W
// chunk instead of 12 scalar loads through L1) and W is read with the non-temporal hint (once per pass).
constexpr int SL_DX_ROWS = 384;  // rows of W per workgroup at most (the host picks the row split accordingly)
template <int MB>
__global__ __launch_bounds__(256) void small_linear_dx_kernel(const float* dy, const float* x, const bf16_t* W,
                                                              float* dx, int M, int N, int K, int act_in,
                                                              int rows_per_block, SLBatch bt, float* det) {
  __shared__ float red[4][8][MB][8];
  __shared__ __attribute__((aligned(16))) float dyt[SL_DX_ROWS][MB];
  if (bt.W) {
    W = (const bf16_t*)bt.W[blockIdx.z];
    dy += (long)blockIdx.z * bt.y_stride;
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = tid & 7, r = tid >> 3;
  const int kc = K >> 3, cg = blockIdx.x * 8 + c;
  const bool live = cg < kc;
  const int n_lo = blockIdx.y * rows_per_block, n_hi = min(N, n_lo + rows_per_block);
  float acc[MB][8];
#pragma unroll
  for (int b = 0; b < MB; ++b)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[b][e] = 0.f;
  const bf16_t* wp = W + (long)cg * 8;
  for (int n0 = n_lo; n0 < n_hi; n0 += SL_DX_ROWS) {  // the workgroup's rows in LDS-sized slices
    const int n1 = min(n_hi, n0 + SL_DX_ROWS);
    __syncthreads();  // (the previous slice has been consumed)
#pragma unroll
    for (int b = 0; b < MB; ++b)
      for (int j = tid; j < n1 - n0; j += 256) dyt[j][b] = b < M ? dy[(long)b * N + n0 + j] : 0.f;
    __syncthreads();
#pragma unroll 4
    for (int n = n0 + r; n < n1; n += 32) {
      float wv[8];
      u32x4 raw = u32x4{0, 0, 0, 0};
      if (live) raw = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wp + (long)n * K));
      unpack8(raw, wv);
      float g[MB];
#pragma unroll
      for (int q = 0; q < MB / 4; ++q) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(&dyt[n - n0][4 * q]);
        g[4 * q] = t[0]; g[4 * q + 1] = t[1]; g[4 * q + 2] = t[2]; g[4 * q + 3] = t[3];
      }
#pragma unroll
      for (int b = 0; b < MB; ++b)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[b][e] += g[b] * wv[e];
    }
  }
  // sum the 8 row lanes of this wave (lanes differing in bits 3..5), then the 4 waves through LDS
#pragma unroll
  for (int b = 0; b < MB; ++b)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float v = acc[b][e];
      v += __shfl_xor(v, 8);
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 32);
      if (lane < 8) red[wave][lane][b][e] = v;
    }
  __syncthreads();
  for (int i = tid; i < 8 * MB * 8; i += 256) {
    const int e = i & 7, b = (i >> 3) % MB, cc = i / (8 * MB);
    const int k = (blockIdx.x * 8 + cc) * 8 + e;
    if (b >= M || k >= K) continue;
    float v = red[0][cc][b][e] + red[1][cc][b][e] + red[2][cc][b][e] + red[3][cc][b][e];
    if (act_in) v *= dsilu_f(x[(long)b * K + k]);
    if (det) det[(((long)blockIdx.z * gridDim.y + blockIdx.y) * M + b) * K + k] = v;  // deterministic mode: [z][y][M][K] partials
    else atomicAdd(dx + (long)b * K + k, v);
  }
}

// model.py:12-22 (+ the .to(bf16) of model.py:374-376)
__global__ void timestep_embedding_kernel(const float* t, float* out, int B, int D) {
  const int half = D >> 1;
  const int gid = blockIdx.x * 256 + threadIdx.x;
  if (gid >= B * half) return;
  const int b = gid / half, i = gid % half;
  const float f = expf(-9.210340371976184f * (float)i / (float)half);
  const float a = t[b] * f;
  out[(long)b * D + i] = bf2f(f2bf(cosf(a)));
  out[(long)b * D + half + i] = bf2f(f2bf(sinf(a)));
}

// --------------------------------------------------------------- patchify / unpatchify ---
// model.py:182-186: token (h w t), feature (c dt dh dw). thread = one (token, c, dt, dh) -> p elements
// token n of sample b is row b * rps + roff + n of `out` (rps = n tokens, roff = 0: the dense [B*N, P] matrix;
// rps = 16 + N, roff = 16: rows of the [B*L] token buffer, register rows left to the caller)
__global__ void patchify_kernel(const bf16_t* x, bf16_t* out, int B, int C, int T, int H, int W, int pt, int p,
                                int rps, int roff) {
  const int t = T / pt, h = H / p, w = W / p;
  const int fpt = C * pt * p;  // feature groups of p elements per token
  const long gid = (long)blockIdx.x * 256 + threadIdx.x;
  const long ntok = (long)B * h * w * t;
  if (gid >= ntok * fpt) return;
  const long tok = gid / fpt;
  int f = (int)(gid % fpt);
  const int dh = f % p; f /= p;
  const int dt = f % pt; f /= pt;
  const int c = f;
  long r = tok;
  const int ti = (int)(r % t); r /= t;
  const int wi = (int)(r % w); r /= w;
  const int hi = (int)(r % h); r /= h;
  const int b = (int)r;
  const bf16_t* src = x + ((((long)b * C + c) * T + ti * pt + dt) * H + hi * p + dh) * W + wi * p;
  const long orow = (long)b * rps + roff + (tok - (long)b * h * w * t);
  bf16_t* dst = out + orow * (fpt * p) + ((c * pt + dt) * p + dh) * p;
  for (int dw = 0; dw < p; ++dw) dst[dw] = src[dw];
}

// model.py:392-401: y[b,(h w t),(p1 p2 p3 c)] <-> out[b,c,(t p3),(h p1),(w p2)]; thread = one out element
template <bool BWD>
__global__ void unpatchify_kernel(const bf16_t* in, bf16_t* outp, int B, int C, int T, int H, int W, int pt, int p,
                                  int rps, int roff) {
  const int t = T / pt, h = H / p, w = W / p;
  const long gid = (long)blockIdx.x * 256 + threadIdx.x;
  const long total = (long)B * C * (t * pt) * (h * p) * (w * p);
  if (gid >= total) return;
  long r = gid;
  const int x = (int)(r % (w * p)); r /= (w * p);
  const int yy = (int)(r % (h * p)); r /= (h * p);
  const int tt = (int)(r % (t * pt)); r /= (t * pt);
  const int c = (int)(r % C); r /= C;
  const int b = (int)r;
  const int wi = x / p, p2 = x % p, hi = yy / p, p1 = yy % p, ti = tt / pt, p3 = tt % pt;
  const long tok = (long)b * rps + roff + ((long)hi * w + wi) * t + ti;  // row of the token matrix (see patchify_kernel)
  const long tokidx = tok * ((long)p * p * pt * C) + (((p1 * p + p2) * pt + p3) * C + c);
  // image index over the full [B,C,T,H,W] tensor (T,H,W may exceed t*pt.. when not divisible)
  const long img = ((((long)b * C + c) * T + tt) * H + yy) * W + x;
  if (BWD) outp[tokidx] = in[img];
  else outp[img] = in[tokidx];
}

// model.py:362: x[b, 0:R, :] = register_tokens
__global__ void fill_registers_kernel(const bf16_t* reg, bf16_t* x, long bs, int B, int R, int D) {
  const long gid = (long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (long)B * R * D) return;
  const int b = (int)(gid / ((long)R * D));
  const long rd = gid % ((long)R * D);
  x[(long)b * bs + rd] = reg[rd];
}
__global__ void registers_bwd_kernel(const bf16_t* dx, long bs, float* dreg, int B, int R, int D) {
  const long gid = (long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (long)R * D) return;
  float s = 0.f;
  for (int b = 0; b < B; ++b) s += bf2f(dx[(long)b * bs + gid]);
  dreg[gid] += s;
}

// ---------------------------------------------------------------- noising + loss ---------
// train.py:115-117 in bf16 with the reference's rounding points: (1-t) rounds, each product
// rounds, the sum rounds.
__global__ void noise_kernel(const bf16_t* x, const bf16_t* n, const float* t, bf16_t* zt, bf16_t* v, int B,
                             long per) {
  const long gid = (long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (long)B * per) return;
  const int b = (int)(gid / per);
  const float tb = t[b];
  const float omt = bf2f(f2bf(1.0f - tb));
  const float xv = bf2f(x[gid]), nv = bf2f(n[gid]);
  const float a = bf2f(f2bf(xv * omt)), c = bf2f(f2bf(nv * tb));
  zt[gid] = f2bf(a + c);
  v[gid] = f2bf(xv - nv);
}

// train.py:121-125: fp32 MSE per sample -> batch mean; also emits d(loss)/d(out) in bf16.
// FIXED-ORDER two-stage reduction (the reference's `.pow(2).mean(dim=(1,2,3,4)).mean()` is one): stage 1 leaves one partial
// sum of squares per workgroup (strided loop -> wave butterfly -> 4 adds, all in an order that depends on `per` only),
// stage 2 (one workgroup) sums a sample's partials in a fixed lane / butterfly order, divides by `per`, then sums the
// samples in index order and divides by B.  No atomics: the scalar is bit-identical run to run and the loss of a batch
// of identical samples equals the single-sample loss to the bit.
__global__ __launch_bounds__(256) void flow_loss_partial_kernel(const bf16_t* v, const bf16_t* out, float* partials,
                                                                bf16_t* dout, float gscale, int B, long per,
                                                                int blocks_per_sample) {
  __shared__ float red[4];
  const int b = blockIdx.y;
  const long base = (long)b * per;
  const float k = 2.0f * gscale / ((float)B * (float)per);
  float s = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < per; i += (long)blocks_per_sample * 256) {
    const float d = bf2f(out[base + i]) - bf2f(v[base + i]);
    s += d * d;
    if (dout) dout[base + i] = f2bf(d * k);
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partials[(long)b * blocks_per_sample + blockIdx.x] = ((red[0] + red[1]) + (red[2] + red[3]));
}

__global__ __launch_bounds__(256) void flow_loss_final_kernel(const float* partials, float* loss, float* per_sample,
                                                              int B, long per, int blocks_per_sample) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int b = wave; b < B; b += 4) {  // one wave per sample
    const float* p = partials + (long)b * blocks_per_sample;
    float s = 0.f;
    for (int j = lane; j < blocks_per_sample; j += 64) s += p[j];
    s = wave_sum(s);
    if (lane == 0) per_sample[b] = s / (float)per;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float tot = 0.f;
    for (int b = 0; b < B; ++b) tot += per_sample[b];
    loss[0] = tot / (float)B;
  }
}

// d(loss)/d(out) for an upstream gradient held in device memory; 8 elements per thread, 16-byte accesses
__global__ __launch_bounds__(256) void flow_loss_bwd_kernel(const bf16_t* v, const bf16_t* out, const float* gloss,
                                                            bf16_t* dout, float inv_n, long n8) {
  const float k = 2.0f * gloss[0] * inv_n;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
    float a[8], b[8];
    unpack8(*reinterpret_cast<const u32x4*>(out + i * 8), a);
    unpack8(*reinterpret_cast<const u32x4*>(v + i * 8), b);
    u32x4 w;
#pragma unroll
    for (int e = 0; e < 4; ++e) w[e] = pack_bf2((a[2 * e] - b[2 * e]) * k, (a[2 * e + 1] - b[2 * e + 1]) * k);
    *reinterpret_cast<u32x4*>(dout + i * 8) = w;
  }
}

// Euler + classifier-free-guidance update of the sampler (sampling/sample.py:139-146), with the
// reference's bf16 rounding points: out = bf16(u + bf16(cfg * bf16(c - u))) when guided, acc(f32) += dt * out,
// latents = bf16(acc).  8 elements per thread, 16-byte accesses.
__global__ void cfg_euler_kernel(const bf16_t* cond, const bf16_t* uncond, float* acc, bf16_t* lat, float cfg,
                                 float dt, long n8) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n8) return;
  float c[8], a[8], o[8];
  unpack8(*reinterpret_cast<const u32x4*>(cond + i * 8), c);
  if (uncond) {
    float u[8];
    unpack8(*reinterpret_cast<const u32x4*>(uncond + i * 8), u);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float d = bf2f(f2bf(c[e] - u[e]));
      const float m = bf2f(f2bf(cfg * d));
      c[e] = bf2f(f2bf(u[e] + m));
    }
  }
  load8f(acc + i * 8, a);
#pragma unroll
  for (int e = 0; e < 8; ++e) { a[e] += dt * c[e]; o[e] = a[e]; }
  *reinterpret_cast<f32x4*>(acc + i * 8) = f32x4{a[0], a[1], a[2], a[3]};
  *reinterpret_cast<f32x4*>(acc + i * 8 + 4) = f32x4{a[4], a[5], a[6], a[7]};
  *reinterpret_cast<u32x4*>(lat + i * 8) = pack8(o);
}

__global__ void cast_f32_bf16_kernel(const float* s, bf16_t* d, long n) {
  const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 8;
  if (i + 8 <= n) {
    float f[8];
    load8f(s + i, f);
    *reinterpret_cast<u32x4*>(d + i) = pack8(f);
  } else {
    for (long j = i; j < n; ++j) d[j] = f2bf(s[j]);
  }
}
__global__ void cast_bf16_f32_kernel(const bf16_t* s, float* d, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) d[i] = bf2f(s[i]);
}

// model.py:219-263: rows of cos/sin for this call.  Row 16+i is the table entry of position
// unravel(i,(t,h,w)) + start (row-major (t h w) flattening); rows < n_reg are
// cos=1, sin=0.  tab_* are the per-axis tables [128, n] the reference's buffer factorises into.
__global__ void rope_rows_kernel(const float* tab_t_cos, const float* tab_t_sin, const float* tab_s_cos,
                                 const float* tab_s_sin, int nt, int ns, int t, int h, int w, int st, int sh, int sw,
                                 int n_reg, float* cosb, float* sinb, const int* start_dev) {
  if (start_dev) {  // offsets kept in device memory (whole-step graph replay: nothing per-step in the arguments)
    st = min(max(start_dev[0], 0), 128 - t);
    sh = min(max(start_dev[1], 0), 128 - h);
    sw = min(max(start_dev[2], 0), 128 - w);
  }
  const int half = nt + 2 * ns;
  const long gid = (long)blockIdx.x * 256 + threadIdx.x;
  const long rows = n_reg + (long)t * h * w;
  if (gid >= rows * half) return;
  const int row = (int)(gid / half), f = (int)(gid % half);
  float c = 1.f, s = 0.f;
  if (row >= n_reg) {
    int i = row - n_reg;
    const int wi = i % w; i /= w;
    const int hi = i % h; i /= h;
    const int ti = i;
    if (f < nt) { c = tab_t_cos[(st + ti) * nt + f]; s = tab_t_sin[(st + ti) * nt + f]; }
    else if (f < nt + ns) { c = tab_s_cos[(sh + hi) * ns + f - nt]; s = tab_s_sin[(sh + hi) * ns + f - nt]; }
    else { c = tab_s_cos[(sw + wi) * ns + f - nt - ns]; s = tab_s_sin[(sw + wi) * ns + f - nt - ns]; }
  }
  cosb[gid] = c;
  sinb[gid] = s;
}

inline int ok() { return hipGetLastError() == hipSuccess ? VDS_OK : VDS_ERR_LAUNCH; }
// deterministic mode: second stage of a column sum (see det_finish_kernel)
inline void det_finish(const float* part, int n_parts, int width, float* target, long tstride, int groups, hipStream_t s) {
  const long n = (long)groups * width;
  hipLaunchKernelGGL(det_finish_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, part, n_parts, width, target,
                     tstride, groups);
}
// rows of one sample per workgroup: ~768 workgroups in total (3 per CU, enough waves to stream HBM)
// while every workgroup still folds >= 8 rows into its column sums before the atomics
inline int rows_per_block_for(int L, int B) {
  const int min_rows = vdscfg::geti(vdscfg::EW_MIN_ROWS) < 1 ? 1 : vdscfg::geti(vdscfg::EW_MIN_ROWS);
  const int target = vdscfg::geti(vdscfg::EW_WGS) > 0 ? vdscfg::geti(vdscfg::EW_WGS) : 768;
  const long total = (long)L * B;
  int rpb = (int)((total + target - 1) / target);
  // (round 5) mid-sized problems (C3b at B = 2: 16 416 rows -> 22 per workgroup): the column-sum epilogue -- two LDS passes
  // and D atomics per workgroup -- is worth 32 rows; measured 4.66 -> 4.40 (rmsnorm_mod_bwd) and 3.18 -> 2.89 ms (gate_bwd)
  // per step.  Small problems (C1: 1088 rows) keep 8 rows per workgroup: they need the workgroups more.
  const int floor_rows = (min_rows == 8 && total >= 8192) ? 32 : min_rows;
  return rpb < floor_rows ? floor_rows : rpb;
}

}  // namespace

#define NC_DISPATCH(D, CALL)                       \
  do {                                             \
    const int nc_ = ((D) / 8 + 63) / 64;           \
    if (nc_ == 1) { CALL(1); }                     \
    else if (nc_ == 2) { CALL(2); }                \
    else if (nc_ == 3) { CALL(3); }                \
    else if (nc_ == 4) { CALL(4); }                \
    else return VDS_ERR_UNSUPPORTED;               \
  } while (0)

// widths whose 16-byte chunk count is a multiple of 16 but not of 64 (D = 384, 768, 1152): the four-rows-per-wave form
// of the RMSNorm kernels; knob rmsnorm_q4 = 0 keeps the one-row-per-wave form (A/B)
static bool rmsnorm_q4(int D) {
  if (D != 384 && D != 768 && D != 1152) return false;
  return vdscfg::geti(vdscfg::RMSNORM_Q4) != 0;
}

extern "C" int vds_rmsnorm_mod_fwd(const void* x, int64_t ldx, const void* w, const float* mod, int64_t ldmod,
                                   int32_t shift_col, int32_t scale_col, void* y, int64_t ldy, float* rstd,
                                   int32_t B, int32_t L, int32_t D, float eps, vds_stream_t stream) {
  if (!x || !mod || !y || !rstd || (D & 7) || (ldx & 7) || (ldy & 7) || (shift_col & 3) || (scale_col & 3) || (ldmod & 3))
    return VDS_ERR_ARG;
  const long rows = (long)B * L;
  hipStream_t s = (hipStream_t)stream;
  vdsprof::Scope ps(VDS_PROF_RMSNORM_FWD, s, 0.0, 4.0 * rows * D + 4.0 * rows);
  if (rmsnorm_q4(D)) {  // DiT-S / B / XL widths: four rows per wave, every lane of every instruction at work
#define CALLQ(NCQ)                                                                                             \
  hipLaunchKernelGGL((rmsnorm_mod_fwd_q4_kernel<NCQ>), dim3((unsigned)((rows + 15) / 16)), dim3(256), 0, s,     \
                     (const bf16_t*)x, (long)ldx, (const bf16_t*)w, mod, (long)ldmod, shift_col, scale_col,   \
                     (bf16_t*)y, (long)ldy, rstd, B, L, eps, QOut{})
    if (D == 384) CALLQ(3);
    else if (D == 768) CALLQ(6);
    else CALLQ(9);
#undef CALLQ
    return ok();
  }
#define CALL(NC)                                                                                              \
  hipLaunchKernelGGL((rmsnorm_mod_fwd_kernel<NC>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s,          \
                     (const bf16_t*)x, (long)ldx, (const bf16_t*)w, mod, (long)ldmod, shift_col, scale_col, \
                     (bf16_t*)y, (long)ldy, rstd, B, L, D, eps, QOut{})
  NC_DISPATCH(D, CALL);
#undef CALL
  return ok();
}

static bool qout_ok(const void* q, int64_t ldq, int32_t fmt, const float* amax_in) {
  return q && amax_in && !(ldq & 7) && (fmt == 0 || fmt == 1);
}

extern "C" int vds_rmsnorm_mod_fwd_fp8(const void* x, int64_t ldx, const void* w, const float* mod, int64_t ldmod,
                                       int32_t shift_col, int32_t scale_col, void* q, int64_t ldq, int32_t fmt,
                                       const float* amax_in, float* amax_part, float* dq_out, float* rstd, int32_t B,
                                       int32_t L, int32_t D, float eps, vds_stream_t stream) {
  if (!x || !mod || !rstd || (D & 7) || (ldx & 7) || (shift_col & 3) || (scale_col & 3) || (ldmod & 3) ||
      !qout_ok(q, ldq, fmt, amax_in))
    return VDS_ERR_ARG;
  const long rows = (long)B * L;
  hipStream_t s = (hipStream_t)stream;
  const QOut qo{(unsigned char*)q, (long)ldq, amax_in, amax_part, dq_out};
  vdsprof::Scope ps(VDS_PROF_RMSNORM_FWD, s, 0.0, 3.0 * rows * D + 4.0 * rows);
  if (rmsnorm_q4(D)) {
#define CALLQ(NCQ)                                                                                               \
  do {                                                                                                           \
    if (fmt == 0)                                                                                                \
      hipLaunchKernelGGL((rmsnorm_mod_fwd_q4_kernel<NCQ, 0>), dim3((unsigned)((rows + 15) / 16)), dim3(256), 0, s, \
                         (const bf16_t*)x, (long)ldx, (const bf16_t*)w, mod, (long)ldmod, shift_col, scale_col,   \
                         (bf16_t*)nullptr, 0L, rstd, B, L, eps, qo);                                             \
    else                                                                                                         \
      hipLaunchKernelGGL((rmsnorm_mod_fwd_q4_kernel<NCQ, 1>), dim3((unsigned)((rows + 15) / 16)), dim3(256), 0, s, \
                         (const bf16_t*)x, (long)ldx, (const bf16_t*)w, mod, (long)ldmod, shift_col, scale_col,   \
                         (bf16_t*)nullptr, 0L, rstd, B, L, eps, qo);                                             \
  } while (0)
    if (D == 384) CALLQ(3);
    else if (D == 768) CALLQ(6);
    else CALLQ(9);
#undef CALLQ
    return ok();
  }
#define CALL(NC)                                                                                               \
  do {                                                                                                         \
    if (fmt == 0)                                                                                              \
      hipLaunchKernelGGL((rmsnorm_mod_fwd_kernel<NC, 0>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s,    \
                         (const bf16_t*)x, (long)ldx, (const bf16_t*)w, mod, (long)ldmod, shift_col, scale_col, \
                         (bf16_t*)nullptr, 0L, rstd, B, L, D, eps, qo);                                        \
    else                                                                                                       \
      hipLaunchKernelGGL((rmsnorm_mod_fwd_kernel<NC, 1>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s,    \
                         (const bf16_t*)x, (long)ldx, (const bf16_t*)w, mod, (long)ldmod, shift_col, scale_col, \
                         (bf16_t*)nullptr, 0L, rstd, B, L, D, eps, qo);                                        \
  } while (0)
  NC_DISPATCH(D, CALL);
#undef CALL
  return ok();
}

extern "C" int vds_rmsnorm_mod_bwd(const void* dy, int64_t lddy, const void* x, int64_t ldx, const void* w,
                                   const float* mod, int64_t ldmod, int32_t shift_col, int32_t scale_col,
                                   const float* rstd, const void* dres, int64_t lddres, void* dx, int64_t lddx,
                                   float* dmod, float* dw, int32_t B, int32_t L, int32_t D, vds_stream_t stream) {
  if (!dy || !x || !mod || !rstd || !dx || !dmod || (D & 7) || (lddy & 7) || (ldx & 7) || (lddx & 7)) return VDS_ERR_ARG;
  if (w && !dw) return VDS_ERR_ARG;
  const int rpb = rows_per_block_for(L, B);
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((L + rpb - 1) / rpb, B);
  float* det = nullptr;
  if (vdsdet::on() && !(det = vdsdet::workspace((size_t)(w ? 3 : 2) * B * grid.x * D, "vds_rmsnorm_mod_bwd"))) return VDS_ERR_ARG;
  vdsprof::Scope ps(VDS_PROF_RMSNORM_BWD, s, 0.0, (dres ? 8.0 : 6.0) * B * L * D + 4.0 * B * L);
#define CALL(NC)                                                                                                  \
  do {                                                                                                            \
    if (w)                                                                                                        \
      hipLaunchKernelGGL((rmsnorm_mod_bwd_kernel<NC, true>), grid, dim3(256), 0, s, (const bf16_t*)dy, (long)lddy, \
                         (const bf16_t*)x, (long)ldx, (const bf16_t*)w, mod, (long)ldmod, shift_col, scale_col,   \
                         rstd, (const bf16_t*)dres, (long)lddres, (bf16_t*)dx, (long)lddx, dmod, dw, B, L, D, rpb, det); \
    else                                                                                                          \
      hipLaunchKernelGGL((rmsnorm_mod_bwd_kernel<NC, false>), grid, dim3(256), 0, s, (const bf16_t*)dy, (long)lddy, \
                         (const bf16_t*)x, (long)ldx, (const bf16_t*)nullptr, mod, (long)ldmod, shift_col,        \
                         scale_col, rstd, (const bf16_t*)dres, (long)lddres, (bf16_t*)dx, (long)lddx, dmod,       \
                         (float*)nullptr, B, L, D, rpb, det);                                                     \
  } while (0)
  NC_DISPATCH(D, CALL);
#undef CALL
  if (det) {
    const long plane = (long)B * grid.x * D;
    det_finish(det, (int)grid.x, D, dmod + shift_col, ldmod, B, s);
    det_finish(det + plane, (int)grid.x, D, dmod + scale_col, ldmod, B, s);
    if (w) det_finish(det + 2 * plane, (int)(B * grid.x), D, dw, 0, 1, s);
  }
  return ok();
}

extern "C" int vds_gate_bwd(const void* dxn, int64_t lddxn, const void* y, int64_t ldy, const float* mod,
                            int64_t ldmod, int32_t gate_col, void* dy, int64_t lddy, float* dmod, float* dbias,
                            int32_t B, int32_t L, int32_t D, vds_stream_t stream) {
  if (!dxn || !y || !mod || !dy || !dmod || (D & 7)) return VDS_ERR_ARG;
  const int rpb = rows_per_block_for(L, B);
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((L + rpb - 1) / rpb, B);
  float* det = nullptr;
  if (vdsdet::on() && !(det = vdsdet::workspace((size_t)2 * B * grid.x * D, "vds_gate_bwd"))) return VDS_ERR_ARG;
  vdsprof::Scope ps(VDS_PROF_GATE_BWD, s, 0.0, 6.0 * B * L * D);
#define CALL(NC)                                                                                             \
  hipLaunchKernelGGL((gate_bwd_kernel<NC>), grid, dim3(256), 0, s, (const bf16_t*)dxn, (long)lddxn,           \
                     (const bf16_t*)y, (long)ldy, mod, (long)ldmod, gate_col, (bf16_t*)dy, (long)lddy, dmod, \
                     dbias, B, L, D, rpb, QOut{}, det)
  NC_DISPATCH(D, CALL);
#undef CALL
  if (det) {
    det_finish(det, (int)grid.x, D, dmod + gate_col, ldmod, B, s);
    if (dbias) det_finish(det + (long)B * grid.x * D, (int)(B * grid.x), D, dbias, 0, 1, s);
  }
  return ok();
}

extern "C" int vds_gate_bwd_fp8(const void* dxn, int64_t lddxn, const void* y, int64_t ldy, const float* mod,
                                int64_t ldmod, int32_t gate_col, void* q, int64_t ldq, int32_t fmt,
                                const float* amax_in, float* amax_part, float* dq_out, float* dmod, float* dbias,
                                int32_t B, int32_t L, int32_t D, vds_stream_t stream) {
  if (!dxn || !y || !mod || !dmod || (D & 7) || !qout_ok(q, ldq, fmt, amax_in)) return VDS_ERR_ARG;
  const int rpb = rows_per_block_for(L, B);
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((L + rpb - 1) / rpb, B);
  const QOut qo{(unsigned char*)q, (long)ldq, amax_in, amax_part, dq_out};
  float* det = nullptr;
  if (vdsdet::on() && !(det = vdsdet::workspace((size_t)2 * B * grid.x * D, "vds_gate_bwd_fp8"))) return VDS_ERR_ARG;
  vdsprof::Scope ps(VDS_PROF_GATE_BWD, s, 0.0, 5.0 * B * L * D);
#define CALL(NC)                                                                                                   \
  do {                                                                                                             \
    if (fmt == 0)                                                                                                  \
      hipLaunchKernelGGL((gate_bwd_kernel<NC, 0>), grid, dim3(256), 0, s, (const bf16_t*)dxn, (long)lddxn,          \
                         (const bf16_t*)y, (long)ldy, mod, (long)ldmod, gate_col, (bf16_t*)nullptr, 0L, dmod, dbias, \
                         B, L, D, rpb, qo, det);                                                                   \
    else                                                                                                           \
      hipLaunchKernelGGL((gate_bwd_kernel<NC, 1>), grid, dim3(256), 0, s, (const bf16_t*)dxn, (long)lddxn,          \
                         (const bf16_t*)y, (long)ldy, mod, (long)ldmod, gate_col, (bf16_t*)nullptr, 0L, dmod, dbias, \
                         B, L, D, rpb, qo, det);                                                                   \
  } while (0)
  NC_DISPATCH(D, CALL);
#undef CALL
  if (det) {
    det_finish(det, (int)grid.x, D, dmod + gate_col, ldmod, B, s);
    if (dbias) det_finish(det + (long)B * grid.x * D, (int)(B * grid.x), D, dbias, 0, 1, s);
  }
  return ok();
}

extern "C" int vds_colsum_bf16_rows(const void* x, int64_t ldx, float* out, int32_t M, int32_t N,
                                    int32_t rows_per_sample, int32_t row_offset, vds_stream_t stream) {
  if (!x || !out || (N & 7) || (ldx & 7) || rows_per_sample < 0 || row_offset < 0) return VDS_ERR_ARG;
  const int rpb = M > 8192 ? (M + 127) / 128 : 64;
  dim3 grid((N / 8 + 63) / 64, (M + rpb - 1) / rpb);
  float* det = nullptr;
  if (vdsdet::on() && !(det = vdsdet::workspace((size_t)grid.y * N, "vds_colsum_bf16"))) return VDS_ERR_ARG;
  hipLaunchKernelGGL(colsum_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (long)ldx, out, M, N, rpb,
                     rows_per_sample, row_offset, det);
  if (det) det_finish(det, (int)grid.y, N, out, 0, 1, (hipStream_t)stream);
  return ok();
}
extern "C" int vds_colsum_bf16(const void* x, int64_t ldx, float* out, int32_t M, int32_t N, vds_stream_t stream) {
  return vds_colsum_bf16_rows(x, ldx, out, M, N, 0, 0, stream);
}

// the wave-per-token kernels need 16-byte aligned head rows (hdp % 8 == 0), hd/2 a multiple of 4 and D <= 2048;
// knob rope_tok = 0 keeps the 8-byte-access backward kernel (A/B)
static bool rope_tok_form(int H, int hd, int hdp) {
  const int on = vdscfg::geti(vdscfg::ROPE_TOK);
  return on && (hdp & 7) == 0 && (hd & 7) == 0 && H * hd <= 2048;
}

extern "C" int vds_qkv_rope_fwd(const void* qkv, const float* cosb, const float* sinb, const void* v0,
                                const void* lam, void* q, void* k, void* v, int32_t B, int32_t L, int32_t H,
                                int32_t hd, int32_t hdp, vds_stream_t stream) {
  if (!qkv || !cosb || !sinb || !q || !k || !v || (hd & 7) || hdp < hd || (hdp & 3)) return VDS_ERR_ARG;
  if (v0 && !lam) return VDS_ERR_ARG;
  if (((hdp - hd) >> 2) > (hd >> 3)) return VDS_ERR_UNSUPPORTED;
  const long n = (long)B * L * H * (hd >> 3);
  vdsprof::Scope ps(VDS_PROF_QKV_ROPE_FWD, (hipStream_t)stream, 0.0, (v0 ? 14.0 : 12.0) * B * L * H * hd);
  // token-tile kernel (16-byte accesses, LDS staging) for the head sizes the model builds; knob rope_tile = 0 keeps the
  // element-wise kernel (A/B), 2 | 4 | 8 picks the tokens per workgroup
  const int tile = vdscfg::geti(vdscfg::ROPE_TILE);
  if (tile > 0 && H * hd <= 2048 && H <= 256 && L >= 8) {
    const long ntok = (long)B * L;
#define ROPE_TILE(HD, HDP, T)                                                                                      \
  do {                                                                                                             \
    const int lds = ropestage::lds_bytes(T, H * HD);                                                               \
    static bool attr = false;                                                                                      \
    if (!attr) {                                                                                                   \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&qkv_rope_fwd_tile_kernel<HD, HDP, T>),              \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                           \
      attr = true;                                                                                                 \
    }                                                                                                              \
    hipLaunchKernelGGL((qkv_rope_fwd_tile_kernel<HD, HDP, T>), dim3((unsigned)((ntok + T - 1) / T)), dim3(256), lds, \
                       (hipStream_t)stream, (const bf16_t*)qkv, cosb, sinb, (const bf16_t*)v0, (const bf16_t*)lam,  \
                       (bf16_t*)q, (bf16_t*)k, (bf16_t*)v, ntok, L, H);                                             \
    return ok();                                                                                                   \
  } while (0)
#define ROPE_TILE_T(HD, HDP)                       \
  do {                                             \
    if (tile == 2) ROPE_TILE(HD, HDP, 2);          \
    else if (tile == 8) ROPE_TILE(HD, HDP, 8);     \
    else ROPE_TILE(HD, HDP, 4);                    \
  } while (0)
    if (hd == 72 && hdp == 96) ROPE_TILE_T(72, 96);
    if (hd == 64 && hdp == 64) ROPE_TILE_T(64, 64);
    if (hd == 128 && hdp == 128) ROPE_TILE_T(128, 128);
    if (hd == 96 && hdp == 96) ROPE_TILE_T(96, 96);
    if (hd == 32 && hdp == 32) ROPE_TILE_T(32, 32);
#undef ROPE_TILE_T
#undef ROPE_TILE
  }
  hipLaunchKernelGGL(qkv_rope_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)qkv, cosb, sinb, (const bf16_t*)v0, (const bf16_t*)lam, (bf16_t*)q, (bf16_t*)k,
                     (bf16_t*)v, B, L, H, hd, hdp);
  return ok();
}

// ---- residual-V: the gradient that reaches v_0 from the blocks that mixed it in (model.py:129-130), summed once -------
// dv0[b,h,l,:hd] = sum_i (1 - lambda_i) dv_i[b,h,l,:hd] over the n mixed blocks (fp32 accumulation in registers, one write)
// instead of an fp32 read-modify-write of dv0 in every block's RoPE backward (vds_qkv_rope_bwd with mix = 2 leaves it out):
// 27 x 0.9 GB of accumulator traffic per DiT-XL step at B = 12 become one 6 GB pass over the blocks' bf16 dv tensors.
struct Dv0Args { const bf16_t* dv[32]; const bf16_t* lam[32]; int n; };
__global__ __launch_bounds__(256) void dv0_reduce_kernel(Dv0Args a, float* out, long rows, int hd, int hdp, int accumulate) {
  const int cpr = hd >> 3;
  const long gid = (long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= rows * cpr) return;
  const long so = (gid / cpr) * hdp + (gid % cpr) * 8;
  float acc[8];
  if (accumulate) load8f(out + so, acc);
  else {
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  }
  for (int i0 = 0; i0 < a.n; i0 += 4) {  // four independent 16-byte loads in flight
    u32x4 r[4];
    float w[4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (i0 + u < a.n) {
        r[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(a.dv[i0 + u] + so));  // read once
        w[u] = 1.0f - bf2f(*a.lam[i0 + u]);
      }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (i0 + u < a.n) {
        float g[8];
        unpack8(r[u], g);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += w[u] * g[e];
      }
  }
  *reinterpret_cast<f32x4*>(out + so) = f32x4{acc[0], acc[1], acc[2], acc[3]};
  *reinterpret_cast<f32x4*>(out + so + 4) = f32x4{acc[4], acc[5], acc[6], acc[7]};
}

// dv / lam: host arrays of n device pointers (dv_i: bf16 [B,H,L,hdp] contiguous; lam_i: one bf16); out: f32 [B,H,L,hdp]
// (columns >= hd are not touched).  accumulate != 0: added to what out holds.  Any n (32 tensors per launch).
extern "C" int vds_dv0_reduce(const void* const* dv, const void* const* lam, int32_t n, float* out, int32_t accumulate,
                              int32_t B, int32_t H, int32_t L, int32_t hd, int32_t hdp, vds_stream_t stream) {
  if (!dv || !lam || !out || n < 0 || B <= 0 || H <= 0 || L <= 0 || (hd & 7) || (hdp & 7) || hdp < hd) return VDS_ERR_ARG;
  if (((uintptr_t)out & 15) != 0) return VDS_ERR_ARG;
  const long rows = (long)B * H * L;
  const long items = rows * (hd >> 3);
  for (int i0 = 0; i0 < n; i0 += 32) {
    Dv0Args a;
    a.n = n - i0 < 32 ? n - i0 : 32;
    for (int i = 0; i < 32; ++i) {
      a.dv[i] = i < a.n ? (const bf16_t*)dv[i0 + i] : nullptr;
      a.lam[i] = i < a.n ? (const bf16_t*)lam[i0 + i] : nullptr;
      if (i < a.n && (!a.dv[i] || !a.lam[i] || ((uintptr_t)a.dv[i] & 15) != 0)) return VDS_ERR_ARG;
    }
    vdsprof::Scope ps(VDS_PROF_QKV_ROPE_BWD, (hipStream_t)stream, 0.0, (2.0 * a.n + 4.0 + ((accumulate || i0) ? 4.0 : 0.0)) * rows * hd);
    hipLaunchKernelGGL(dv0_reduce_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, out,
                       rows, hd, hdp, (accumulate || i0) ? 1 : 0);
  }
  return ok();
}

extern "C" int vds_qkv_rope_bwd(const void* dq, const void* dk, const void* dv, const float* cosb,
                                const float* sinb, const void* qkv_raw, const void* v0, const void* lam,
                                float* dv0_acc, float* dlam, void* dqkv, int32_t mix, int32_t add_dv0, int32_t B,
                                int32_t L, int32_t H, int32_t hd, int32_t hdp, vds_stream_t stream) {
  if (!dq || !dk || !dv || !cosb || !sinb || !dqkv || (hd & 7)) return VDS_ERR_ARG;
  if (mix && (!qkv_raw || !v0 || !lam || !dlam || (mix == 1 && !dv0_acc) || mix < 0 || mix > 2)) return VDS_ERR_ARG;
  if (add_dv0 && !dv0_acc) return VDS_ERR_ARG;
  const long n = (long)B * L * H * (hd >> 3);
  vdsprof::Scope ps(VDS_PROF_QKV_ROPE_BWD, (hipStream_t)stream, 0.0, (mix ? 24.0 : 12.0) * B * L * H * hd);
  const bool tok = rope_tok_form(H, hd, hdp);
  const unsigned nblk = tok ? (unsigned)(((long)B * L + 3) / 4) : (unsigned)((n + 255) / 256);
  float* det = nullptr;  // deterministic mode: the lambda gradient as per-workgroup partials + one fixed-order sum
  if (mix && vdsdet::on() && !(det = vdsdet::workspace(nblk, "vds_qkv_rope_bwd"))) return VDS_ERR_ARG;
  if (tok) {
    const int D = H * hd, lds = 4 * (4 * D + 4 * hd);
    const dim3 grid(nblk);
#define ROPE_BWD(NI)                                                                                                \
  hipLaunchKernelGGL(qkv_rope_bwd_tok_kernel<NI>, grid, dim3(256), lds, (hipStream_t)stream, (const bf16_t*)dq,    \
                     (const bf16_t*)dk, (const bf16_t*)dv, cosb, sinb, (const bf16_t*)qkv_raw, (const bf16_t*)v0,  \
                     (const bf16_t*)lam, dv0_acc, dlam, (bf16_t*)dqkv, mix, add_dv0, B, L, H, hd, hdp, QOut{}, det)
    if (D <= 512) ROPE_BWD(1);
    else if (D <= 1024) ROPE_BWD(2);
    else if (D <= 1536) ROPE_BWD(3);
    else ROPE_BWD(4);
#undef ROPE_BWD
  } else {
    hipLaunchKernelGGL(qkv_rope_bwd_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)dq, (const bf16_t*)dk, (const bf16_t*)dv, cosb, sinb, (const bf16_t*)qkv_raw,
                       (const bf16_t*)v0, (const bf16_t*)lam, dv0_acc, dlam, (bf16_t*)dqkv, mix, add_dv0, B, L, H, hd, hdp, det);
  }
  if (det) hipLaunchKernelGGL(det_finish_scalar_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)det, (int)nblk, dlam);
  return ok();
}

// as vds_qkv_rope_bwd, the [B*L, 3D] result written as fp8 (wave-per-token kernel only)
extern "C" int vds_qkv_rope_bwd_fp8(const void* dq, const void* dk, const void* dv, const float* cosb,
                                    const float* sinb, const void* qkv_raw, const void* v0, const void* lam,
                                    float* dv0_acc, float* dlam, void* q, int64_t ldq, int32_t fmt,
                                    const float* amax_in, float* amax_part, float* dq_out, int32_t mix,
                                    int32_t add_dv0, int32_t B, int32_t L, int32_t H, int32_t hd, int32_t hdp,
                                    vds_stream_t stream) {
  if (!dq || !dk || !dv || !cosb || !sinb || (hd & 7) || !qout_ok(q, ldq, fmt, amax_in)) return VDS_ERR_ARG;
  if (mix && (!qkv_raw || !v0 || !lam || !dlam || (mix == 1 && !dv0_acc) || mix < 0 || mix > 2)) return VDS_ERR_ARG;
  if (add_dv0 && !dv0_acc) return VDS_ERR_ARG;
  if ((hdp & 7) || H * hd > 2048 || ldq < 3 * H * hd) return VDS_ERR_UNSUPPORTED;
  vdsprof::Scope ps(VDS_PROF_QKV_ROPE_BWD, (hipStream_t)stream, 0.0, (mix ? 21.0 : 9.0) * B * L * H * hd);
  const int D = H * hd, lds = 4 * (4 * D + 4 * hd);
  const dim3 grid((unsigned)(((long)B * L + 3) / 4));
  const QOut qo{(unsigned char*)q, (long)ldq, amax_in, amax_part, dq_out};
  float* det = nullptr;
  if (mix && vdsdet::on() && !(det = vdsdet::workspace(grid.x, "vds_qkv_rope_bwd_fp8"))) return VDS_ERR_ARG;
#define ROPE_BWD_Q(NI, F)                                                                                             \
  hipLaunchKernelGGL((qkv_rope_bwd_tok_kernel<NI, F>), grid, dim3(256), lds, (hipStream_t)stream, (const bf16_t*)dq, \
                     (const bf16_t*)dk, (const bf16_t*)dv, cosb, sinb, (const bf16_t*)qkv_raw, (const bf16_t*)v0,    \
                     (const bf16_t*)lam, dv0_acc, dlam, (bf16_t*)nullptr, mix, add_dv0, B, L, H, hd, hdp, qo, det)
#define ROPE_BWD_F(F)              \
  do {                             \
    if (D <= 512) ROPE_BWD_Q(1, F); \
    else if (D <= 1024) ROPE_BWD_Q(2, F); \
    else if (D <= 1536) ROPE_BWD_Q(3, F); \
    else ROPE_BWD_Q(4, F);          \
  } while (0)
  if (fmt == 0) ROPE_BWD_F(0);
  else ROPE_BWD_F(1);
#undef ROPE_BWD_F
#undef ROPE_BWD_Q
  if (det) hipLaunchKernelGGL(det_finish_scalar_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)det, (int)grid.x, dlam);
  return ok();
}

// ---- round 4: the batched adaLN linears as streaming kernels -----------------------------------------------------
// The 28 x [9D, D] modulation weights (669 MB at DiT-XL) are read once per pass against a 12-row input: the kernels
// above re-read x (forward: 6 bytes of x through L1 per byte of W, SiLU recomputed per wave) resp. dy (12 scalar loads
// per 16 bytes of W) and ran at 0.7 / 0.34 TB/s.  Here the small operand lives in LDS and W streams past it once.
//
// forward on v_mfma_f32_16x16x32_bf16: y^T[n, m] = sum_k W[n, k] x[m, k].  A = 16 rows of W straight from global memory
// (lane (r = l & 15, g = l >> 4) loads the 16 bytes k0 + 8g .. + 7 of row n0 + r: its MFMA fragment), B = x^T from LDS as
// bf16 (hi, mid, lo) terms -- x = hi + mid + lo to 2^-25, three MFMAs per K step -- so the fp32 input keeps its precision.  One wave =
// 16 output columns; the accumulator lane (c, g) holds y[m = c][n0 + 4g .. + 3]: one 16-byte store.
typedef __attribute__((ext_vector_type(8))) __bf16 sl_bf16x8;
constexpr int SL_FWD_WAVES = 8;   // 16 output columns per wave: 128 per workgroup
constexpr int SL_FWD_TERMS = 3;   // x = hi + mid + lo in bf16: 2^-25, i.e. fp32 (two terms: 2^-17, which the sharded-vs-
                                  // unsharded gradient comparison of tests/test_model_gpu.py sees at the 1e-4 level)
__global__ __launch_bounds__(64 * SL_FWD_WAVES) void small_linear_fwd_mfma_kernel(const float* x, const bf16_t* W,
                                                                               const bf16_t* bias, float* y, int M, int N,
                                                                               int K, int act_in, SLBatch bt) {
  extern __shared__ __attribute__((aligned(16))) char sl_smem[];
  if (bt.W) {  // batched: set blockIdx.z of the pointer tables (else: the one weight matrix passed directly)
    W = (const bf16_t*)bt.W[blockIdx.z];
    bias = bt.bias ? (const bf16_t*)bt.bias[blockIdx.z] : nullptr;
    y += (long)blockIdx.z * bt.y_stride;
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ldx = K * 2 + 16;  // bytes per staged row (+16: rows 4 banks apart, conflict-free ds_read_b128)
  const int term = 16 * ldx;   // bytes per term
  for (int i = tid; i < 16 * (K >> 2); i += 64 * SL_FWD_WAVES) {  // 4 consecutive k per thread
    const int m = i / (K >> 2), k4 = (i % (K >> 2)) * 4;
    f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
    if (m < M) v = *reinterpret_cast<const f32x4*>(x + (long)m * K + k4);
    float rem[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) rem[e] = act_in ? silu_f(v[e]) : v[e];
#pragma unroll
    for (int t = 0; t < SL_FWD_TERMS; ++t) {
      float p[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        p[e] = bf2f(f2bf(rem[e]));
        rem[e] -= p[e];  // exact
      }
      *reinterpret_cast<u32x2*>(sl_smem + t * term + m * ldx + k4 * 2) = u32x2{pack_bf2(p[0], p[1]), pack_bf2(p[2], p[3])};
    }
  }
  __syncthreads();
  const int n0 = (blockIdx.x * SL_FWD_WAVES + wave) * 16;
  if (n0 >= N) return;
  const int r = lane & 15, g = lane >> 4;
  const bf16_t* wrow = W + (long)min(n0 + r, N - 1) * K + 8 * g;
  const char* bx = sl_smem + r * ldx + 16 * g;
  f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nks = K >> 5;
  constexpr int U = 6;  // K steps whose W loads are in flight together
  auto step = [&](u32x4 raw, int ks) {
    const sl_bf16x8 av = __builtin_bit_cast(sl_bf16x8, raw);
#pragma unroll
    for (int t = SL_FWD_TERMS - 1; t >= 0; --t)  // smallest term first
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, *reinterpret_cast<const sl_bf16x8*>(bx + t * term + ks * 64), acc, 0, 0, 0);
  };
  int ks = 0;
  for (; ks + U <= nks; ks += U) {
    u32x4 a[U];
#pragma unroll
    for (int u = 0; u < U; ++u) a[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wrow + (ks + u) * 32));
#pragma unroll
    for (int u = 0; u < U; ++u) step(a[u], ks + u);
  }
  for (; ks < nks; ++ks) step(*reinterpret_cast<const u32x4*>(wrow + ks * 32), ks);
  // accumulator: column j = lane & 15 = m, rows 4 g + e = output columns n0 + 4 g + e
  const int m = r, nn = n0 + 4 * g;
  if (m < M && nn < N) {
    f32x4 o = acc;
    if (bias) {
      const u32x2 bv = *reinterpret_cast<const u32x2*>(bias + nn);
      o[0] += bflo(bv[0]); o[1] += bfhi(bv[0]); o[2] += bflo(bv[1]); o[3] += bfhi(bv[1]);
    }
    *reinterpret_cast<f32x4*>(y + (long)m * N + nn) = o;
  }
}

// launches the MFMA forward when the shape suits it (the per-block form of the sharding runtime and the batched form of
// the unsharded model then run the SAME per-row arithmetic: their results are bit-identical, which the sharded-vs-
// unsharded comparisons of tests/test_model_gpu.py rely on).  VDS_ADALN_MFMA=0: the row kernel (A/B).
static bool small_linear_fwd_mfma(const float* x, const bf16_t* W, const bf16_t* bias, float* y, int M, int N, int K,
                                  int act_in, const SLBatch& bt, int nb, hipStream_t s) {
  const int mfma_on = vdscfg::geti(vdscfg::ADALN_MFMA);
  const size_t lds = (size_t)SL_FWD_TERMS * 16 * (K * 2 + 16);
  if (!mfma_on || M > 16 || (K & 31) || (N & 15) || lds > 160 * 1024 || N < 1024) return false;  // (small N: the row kernel's finer grid)
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&small_linear_fwd_mfma_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  hipLaunchKernelGGL(small_linear_fwd_mfma_kernel, dim3((N / 16 + SL_FWD_WAVES - 1) / SL_FWD_WAVES, 1, nb),
                     dim3(64 * SL_FWD_WAVES), lds, s, x, W, bias, y, M, N, K, act_in, bt);
  return true;
}

extern "C" int vds_small_linear_fwd(const float* x, const void* W, const void* bias, float* y, int32_t M,
                                    int32_t N, int32_t K, int32_t act_in, vds_stream_t stream) {
  if (!x || !W || !y || M < 1 || (K & 7)) return VDS_ERR_ARG;
  const dim3 grid((N + 15) / 16);
  // the kernels keep <= 16 rows in registers: larger batches (the reference trains with up to 64 samples per rank,
  // train.py:150) run as row chunks -- rows are independent
  for (int r0 = 0; r0 < M; r0 += 16) {
    const int m = min(16, M - r0);
    const float* xs = x + (long)r0 * K;
    float* ys = y + (long)r0 * N;
    if (small_linear_fwd_mfma(xs, (const bf16_t*)W, (const bf16_t*)bias, ys, m, N, K, act_in, SLBatch{}, 1, (hipStream_t)stream))
      continue;
    if (m <= 4)
      hipLaunchKernelGGL(small_linear_fwd_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, xs, (const bf16_t*)W,
                         (const bf16_t*)bias, ys, m, N, K, act_in, SLBatch{});
    else if (m <= 8)
      hipLaunchKernelGGL(small_linear_fwd_kernel<8>, grid, dim3(256), 0, (hipStream_t)stream, xs, (const bf16_t*)W,
                         (const bf16_t*)bias, ys, m, N, K, act_in, SLBatch{});
    else
      hipLaunchKernelGGL(small_linear_fwd_kernel<16>, grid, dim3(256), 0, (hipStream_t)stream, xs, (const bf16_t*)W,
                         (const bf16_t*)bias, ys, m, N, K, act_in, SLBatch{});
  }
  return ok();
}
extern "C" int vds_small_linear_bwd(const float* dy, const float* x, const void* W, float* dW, float* dbias,
                                    float* dx, int32_t M, int32_t N, int32_t K, int32_t act_in, vds_stream_t stream) {
  if (!dy || !x || M < 1 || (K & 7)) return VDS_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (dW) {
    const long n = (long)((N + 3) / 4) * (K >> 3);
    hipLaunchKernelGGL(small_linear_dw_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, dy, x, dW, dbias, M, N, K, act_in, SLBatch{});
  }
  if (dx) {
    if (!W) return VDS_ERR_ARG;
    const int gx = ((K >> 3) + 7) / 8;
    int ny = max(1, min((N + 31) / 32, (512 + gx - 1) / gx));
    const int rows = ((N + ny - 1) / ny + 31) / 32 * 32;
    ny = (N + rows - 1) / rows;
    const dim3 grid(gx, ny);
    float* det = nullptr;
    if (vdsdet::on() && !(det = vdsdet::workspace((size_t)ny * 16 * K, "vds_small_linear_bwd"))) return VDS_ERR_ARG;
    for (int r0 = 0; r0 < M; r0 += 16) {  // <= 16 rows per launch (rows are independent; dx accumulates)
      const int m = min(16, M - r0);
      const float* dys = dy + (long)r0 * N;
      const float* xs = x + (long)r0 * K;
      float* dxs = dx + (long)r0 * K;
      if (m <= 4)
        hipLaunchKernelGGL(small_linear_dx_kernel<4>, grid, dim3(256), 0, s, dys, xs, (const bf16_t*)W, dxs, m, N, K, act_in, rows, SLBatch{}, det);
      else if (m <= 8)
        hipLaunchKernelGGL(small_linear_dx_kernel<8>, grid, dim3(256), 0, s, dys, xs, (const bf16_t*)W, dxs, m, N, K, act_in, rows, SLBatch{}, det);
      else
        hipLaunchKernelGGL(small_linear_dx_kernel<16>, grid, dim3(256), 0, s, dys, xs, (const bf16_t*)W, dxs, m, N, K, act_in, rows, SLBatch{}, det);
      if (det) det_finish(det, ny, m * K, dxs, 0, 1, s);
    }
  }
  return ok();
}

// the batched forms: nb weight sets (device pointer tables) applied to one shared input of M <= 16 rows
extern "C" int vds_small_linear_fwd_batched(const float* x, const void* const* W_ptrs, const void* const* bias_ptrs,
                                            float* y, int64_t y_stride, int32_t nb, int32_t M, int32_t N, int32_t K,
                                            int32_t act_in, vds_stream_t stream) {
  if (!x || !W_ptrs || !y || nb < 1 || M < 1 || M > 16 || (K & 7)) return VDS_ERR_ARG;
  const dim3 grid((N + 15) / 16, 1, nb);
  const SLBatch bt{W_ptrs, bias_ptrs, nullptr, nullptr, (long)y_stride};
  hipStream_t s = (hipStream_t)stream;
  if (small_linear_fwd_mfma(x, nullptr, nullptr, y, M, N, K, act_in, bt, nb, s)) return ok();
  if (M <= 4) hipLaunchKernelGGL(small_linear_fwd_kernel<4>, grid, dim3(256), 0, s, x, (const bf16_t*)nullptr, (const bf16_t*)nullptr, y, M, N, K, act_in, bt);
  else if (M <= 8) hipLaunchKernelGGL(small_linear_fwd_kernel<8>, grid, dim3(256), 0, s, x, (const bf16_t*)nullptr, (const bf16_t*)nullptr, y, M, N, K, act_in, bt);
  else hipLaunchKernelGGL(small_linear_fwd_kernel<16>, grid, dim3(256), 0, s, x, (const bf16_t*)nullptr, (const bf16_t*)nullptr, y, M, N, K, act_in, bt);
  return ok();
}

extern "C" int vds_small_linear_bwd_batched(const float* dy, int64_t dy_stride, const float* x, const void* const* W_ptrs,
                                            float* const* dW_ptrs, float* const* dbias_ptrs, float* dx, int32_t nb,
                                            int32_t M, int32_t N, int32_t K, int32_t act_in, vds_stream_t stream) {
  if (!dy || !x || nb < 1 || M < 1 || M > 16 || (K & 7)) return VDS_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (dW_ptrs) {
    const SLBatch bt{nullptr, nullptr, dW_ptrs, dbias_ptrs, (long)dy_stride};
    const dim3 tg((K + 127) / 128, (N + 63) / 64, nb);
    if (M <= 4) hipLaunchKernelGGL(small_linear_dw_tile_kernel<4>, tg, dim3(256), 0, s, dy, x, M, N, K, act_in, bt);
    else if (M <= 8) hipLaunchKernelGGL(small_linear_dw_tile_kernel<8>, tg, dim3(256), 0, s, dy, x, M, N, K, act_in, bt);
    else hipLaunchKernelGGL(small_linear_dw_tile_kernel<16>, tg, dim3(256), 0, s, dy, x, M, N, K, act_in, bt);
  }
  if (dx) {
    if (!W_ptrs) return VDS_ERR_ARG;
    const int gx = ((K >> 3) + 7) / 8;
    // row split: enough workgroups over all sets for ~8 per CU, each with several LDS slices of rows (the epilogue --
    // a 3-level reduction and 8 x M x 8 atomics -- is paid once per workgroup)
    int ny = max(1, min((N + 31) / 32, (2048 + gx * nb - 1) / (gx * nb)));
    const int rows = ((N + ny - 1) / ny + 31) / 32 * 32;
    ny = (N + rows - 1) / rows;
    const dim3 grid(gx, ny, nb);
    const SLBatch bt{W_ptrs, nullptr, nullptr, nullptr, (long)dy_stride};
    float* det = nullptr;
    if (vdsdet::on() && !(det = vdsdet::workspace((size_t)nb * ny * M * K, "vds_small_linear_bwd_batched"))) return VDS_ERR_ARG;
    if (M <= 4) hipLaunchKernelGGL(small_linear_dx_kernel<4>, grid, dim3(256), 0, s, dy, x, (const bf16_t*)nullptr, dx, M, N, K, act_in, rows, bt, det);
    else if (M <= 8) hipLaunchKernelGGL(small_linear_dx_kernel<8>, grid, dim3(256), 0, s, dy, x, (const bf16_t*)nullptr, dx, M, N, K, act_in, rows, bt, det);
    else hipLaunchKernelGGL(small_linear_dx_kernel<16>, grid, dim3(256), 0, s, dy, x, (const bf16_t*)nullptr, dx, M, N, K, act_in, rows, bt, det);
    if (det) det_finish(det, nb * ny, M * K, dx, 0, 1, s);
  }
  return ok();
}

extern "C" int vds_timestep_embedding(const float* t, float* out, int32_t B, int32_t D, vds_stream_t stream) {
  if (!t || !out || (D & 1)) return VDS_ERR_ARG;
  hipLaunchKernelGGL(timestep_embedding_kernel, dim3((B * (D / 2) + 255) / 256), dim3(256), 0, (hipStream_t)stream, t, out, B, D);
  return ok();
}

extern "C" int vds_patchify_rows(const void* latent, void* patches, int32_t B, int32_t C, int32_t T, int32_t H,
                                 int32_t W, int32_t pt, int32_t p, int32_t rows_per_sample, int32_t row_offset,
                                 vds_stream_t stream) {
  if (!latent || !patches || pt < 1 || p < 1) return VDS_ERR_ARG;
  const int ntok = (H / p) * (W / p) * (T / pt);
  if (rows_per_sample == 0) rows_per_sample = ntok;
  if (row_offset < 0 || rows_per_sample < row_offset + ntok) return VDS_ERR_ARG;
  const long n = (long)B * ntok * C * pt * p;
  hipLaunchKernelGGL(patchify_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)latent, (bf16_t*)patches, B, C, T, H, W, pt, p, rows_per_sample, row_offset);
  return ok();
}
extern "C" int vds_patchify(const void* latent, void* patches, int32_t B, int32_t C, int32_t T, int32_t H,
                            int32_t W, int32_t pt, int32_t p, vds_stream_t stream) {
  return vds_patchify_rows(latent, patches, B, C, T, H, W, pt, p, 0, 0, stream);
}

extern "C" int vds_unpatchify_rows(const void* y, void* out, int32_t B, int32_t C, int32_t T, int32_t H, int32_t W,
                                   int32_t pt, int32_t p, int32_t rows_per_sample, int32_t row_offset, int32_t bwd,
                                   vds_stream_t stream) {
  if (!y || !out || (T % pt) || (H % p) || (W % p)) return VDS_ERR_ARG;
  const int ntok = (H / p) * (W / p) * (T / pt);
  if (rows_per_sample == 0) rows_per_sample = ntok;
  if (row_offset < 0 || rows_per_sample < row_offset + ntok) return VDS_ERR_ARG;
  const long n = (long)B * C * T * H * W;
  const dim3 grid((unsigned)((n + 255) / 256));
  if (bwd)  // y = d(out) [B,C,T,H,W] in, out = d(token rows) written
    hipLaunchKernelGGL((unpatchify_kernel<true>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)y,
                       (bf16_t*)out, B, C, T, H, W, pt, p, rows_per_sample, row_offset);
  else
    hipLaunchKernelGGL((unpatchify_kernel<false>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)y,
                       (bf16_t*)out, B, C, T, H, W, pt, p, rows_per_sample, row_offset);
  return ok();
}
extern "C" int vds_unpatchify(const void* y, void* out, int32_t B, int32_t C, int32_t T, int32_t H, int32_t W,
                              int32_t pt, int32_t p, vds_stream_t stream) {
  return vds_unpatchify_rows(y, out, B, C, T, H, W, pt, p, 0, 0, 0, stream);
}

extern "C" int vds_unpatchify_bwd(const void* dout, void* dy, int32_t B, int32_t C, int32_t T, int32_t H,
                                  int32_t W, int32_t pt, int32_t p, vds_stream_t stream) {
  return vds_unpatchify_rows(dout, dy, B, C, T, H, W, pt, p, 0, 0, 1, stream);
}

extern "C" int vds_fill_registers(const void* reg, void* x, int64_t batch_stride, int32_t B, int32_t R, int32_t D,
                                  vds_stream_t stream) {
  if (!reg || !x) return VDS_ERR_ARG;
  const long n = (long)B * R * D;
  hipLaunchKernelGGL(fill_registers_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)reg, (bf16_t*)x, (long)batch_stride, B, R, D);
  return ok();
}

extern "C" int vds_registers_bwd(const void* dx, int64_t batch_stride, float* dreg, int32_t B, int32_t R, int32_t D,
                                 vds_stream_t stream) {
  if (!dx || !dreg) return VDS_ERR_ARG;
  const long n = (long)R * D;
  hipLaunchKernelGGL(registers_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)dx, (long)batch_stride, dreg, B, R, D);
  return ok();
}

extern "C" int vds_noise_latents(const void* x, const void* noise, const float* t, void* z_t, void* v, int32_t B,
                                 int64_t per_sample, vds_stream_t stream) {
  if (!x || !noise || !t || !z_t || !v) return VDS_ERR_ARG;
  const long n = (long)B * per_sample;
  hipLaunchKernelGGL(noise_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)x, (const bf16_t*)noise, t, (bf16_t*)z_t, (bf16_t*)v, B, (long)per_sample);
  return ok();
}

static int flow_loss_bps(long per_sample_n) {
  long bps = (per_sample_n + 256 * 16 - 1) / (256 * 16);
  return bps < 1 ? 1 : bps > 256 ? 256 : (int)bps;
}

extern "C" int64_t vds_flow_loss_workspace_floats(int32_t B, int64_t per_sample_n) {
  if (B <= 0 || per_sample_n <= 0) return 0;
  return (int64_t)B * flow_loss_bps(per_sample_n);
}

extern "C" int vds_flow_loss(const void* v, const void* out, float* loss_out, float* per_sample, void* dout,
                             float gscale, int32_t B, int64_t per_sample_n, float* workspace, vds_stream_t stream) {
  if (!v || !out || !loss_out || !per_sample || !workspace || B <= 0 || per_sample_n <= 0) return VDS_ERR_ARG;
  const int bps = flow_loss_bps(per_sample_n);
  hipLaunchKernelGGL(flow_loss_partial_kernel, dim3(bps, B), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)v,
                     (const bf16_t*)out, workspace, (bf16_t*)dout, gscale, B, (long)per_sample_n, bps);
  hipLaunchKernelGGL(flow_loss_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)workspace,
                     loss_out, per_sample, B, (long)per_sample_n, bps);
  return ok();
}

extern "C" int vds_rope_apply(const void* x, int64_t x_sb, int64_t x_sh, int64_t x_sl, const float* cos,
                              const float* sin, void* y, int64_t y_sb, int64_t y_sh, int64_t y_sl, int32_t B,
                              int32_t H, int32_t L, int32_t hd, int32_t inverse, vds_stream_t stream) {
  if (!x || !cos || !sin || !y || B <= 0 || H <= 0 || L <= 0 || hd <= 0 || (hd & 7)) return VDS_ERR_ARG;
  if ((x_sb | x_sh | x_sl | y_sb | y_sh | y_sl) & 3) return VDS_ERR_ARG;  // 8-byte accesses
  const long n = (long)B * H * L * (hd >> 3);
  hipLaunchKernelGGL(rope_apply_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)x, (long)x_sb, (long)x_sh, (long)x_sl, cos, sin, (bf16_t*)y, (long)y_sb,
                     (long)y_sh, (long)y_sl, B, H, L, hd, inverse);
  return ok();
}

extern "C" int vds_flow_loss_bwd(const void* v, const void* out, const float* gloss_dev, void* dout, int32_t B,
                                 int64_t per_sample_n, vds_stream_t stream) {
  const long n = (long)B * per_sample_n;
  if (!v || !out || !gloss_dev || !dout || n <= 0 || (n & 7)) return VDS_ERR_ARG;
  long blocks = (n / 8 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(flow_loss_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)v, (const bf16_t*)out, gloss_dev, (bf16_t*)dout, 1.0f / (float)n, n / 8);
  return ok();
}

extern "C" int vds_cfg_euler_step(const void* cond, const void* uncond, float* acc, void* latents, float cfg_scale,
                                  float dt, int64_t n, vds_stream_t stream) {
  if (!cond || !acc || !latents || n <= 0 || (n & 7)) return VDS_ERR_ARG;
  const long n8 = n / 8;
  hipLaunchKernelGGL(cfg_euler_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)cond, (const bf16_t*)uncond, acc, (bf16_t*)latents, cfg_scale, dt, n8);
  return ok();
}

extern "C" int vds_cast_f32_bf16(const float* src, void* dst, int64_t n, vds_stream_t stream) {
  if (!src || !dst) return VDS_ERR_ARG;
  if (n == 0) return VDS_OK;
  const long thr = (n + 7) / 8;
  hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3((unsigned)((thr + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, (long)n);
  return ok();
}

extern "C" int vds_cast_bf16_f32(const void* src, float* dst, int64_t n, vds_stream_t stream) {
  if (!src || !dst) return VDS_ERR_ARG;
  if (n == 0) return VDS_OK;
  hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, dst, (long)n);
  return ok();
}

extern "C" int vds_rope_rows(const float* tab_t_cos, const float* tab_t_sin, const float* tab_s_cos,
                             const float* tab_s_sin, int32_t nt, int32_t ns, int32_t t, int32_t h, int32_t w,
                             int32_t st, int32_t sh, int32_t sw, int32_t n_reg, float* cosb, float* sinb,
                             vds_stream_t stream) {
  if (!tab_t_cos || !tab_t_sin || !tab_s_cos || !tab_s_sin || !cosb || !sinb) return VDS_ERR_ARG;
  if (st < 0 || sh < 0 || sw < 0 || st + t > 128 || sh + h > 128 || sw + w > 128) return VDS_ERR_ARG;
  const long n = ((long)n_reg + (long)t * h * w) * (nt + 2 * ns);
  hipLaunchKernelGGL(rope_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, tab_t_cos,
                     tab_t_sin, tab_s_cos, tab_s_sin, nt, ns, t, h, w, st, sh, sw, n_reg, cosb, sinb, (const int*)nullptr);
  return ok();
}

extern "C" int vds_rope_rows_dev(const float* tab_t_cos, const float* tab_t_sin, const float* tab_s_cos,
                                 const float* tab_s_sin, int32_t nt, int32_t ns, int32_t t, int32_t h, int32_t w,
                                 const int32_t* start_dev, int32_t n_reg, float* cosb, float* sinb,
                                 vds_stream_t stream) {
  if (!tab_t_cos || !tab_t_sin || !tab_s_cos || !tab_s_sin || !cosb || !sinb || !start_dev) return VDS_ERR_ARG;
  if (t > 128 || h > 128 || w > 128) return VDS_ERR_ARG;
  const long n = ((long)n_reg + (long)t * h * w) * (nt + 2 * ns);
  hipLaunchKernelGGL(rope_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, tab_t_cos,
                     tab_t_sin, tab_s_cos, tab_s_sin, nt, ns, t, h, w, 0, 0, 0, n_reg, cosb, sinb, (const int*)start_dev);
  return ok();
}
