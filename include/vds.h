/* vds.h -- C ABI of the MI355X-native video-DiT train-step kernels (libvds_hip.so).
 *
 * The reference (fal-ai-community/video-diffusion-speedrun) has no FFI layer: its operator
 * surface is the set of torch call sites inside model.py / train.py.
 * Each entry point below replaces one (group of) call site(s), cited as file:line of the
 * reference.  Conventions:
 *   - plain pointers + sizes only, no torch types; all pointers are DEVICE pointers;
 *   - the caller owns/allocates every buffer (inputs, outputs, workspace); kernels never
 *     allocate, free or synchronise;
 *   - every call is asynchronous on the hipStream_t passed last (void* here so that the
 *     header is usable from plain C / ctypes / cgo without the HIP headers);
 *   - return 0 on success, negative VDS_ERR_* on bad arguments / unsupported shapes /
 *     launch failure (the Python host turns that into RuntimeError, matching the
 *     reference's exception convention);
 *   - bf16 tensors are raw uint16 bit patterns; "f32" are IEEE floats;
 *   - thread-compatible: one caller thread per process (one process per GPU).
 */
#ifndef VDS_H
#define VDS_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VDS_OK 0
#define VDS_ERR_ARG (-1)
#define VDS_ERR_UNSUPPORTED (-2)
#define VDS_ERR_LAUNCH (-3)

typedef void* vds_stream_t; /* hipStream_t */

int vds_version(void);
const char* vds_last_error(void);

/* ------------------------------------------------------------------ GEMM (MFMA) ------
 * C[M,N] = sum_k opA[m,k] * opB[k,n], bf16 operands, fp32 accumulate.
 *   layout VDS_NT: A[M,K] (lda), B[N,K] (ldb)   nn.Linear forward  y = x W^T
 *                  (model.py:125,138,147,150,159,84,86,184,390,90,319,321)
 *   layout VDS_NN: A[M,K] (lda), B[K,N] (ldb)   its input gradient  dx = dy W
 *   layout VDS_TN: A[K,M] (lda), B[K,N] (ldb)   its weight gradient dW = dy^T x  (fp32 out)
 * epilogues fuse the elementwise work the reference runs as separate kernels:
 *   VDS_EPI_STORE      C = bf16(acc + bias[n])
 *   VDS_EPI_BIAS_GELU  C = bf16(pre), C2 = bf16(gelu_erf(pre)), pre = acc + bias[n]  (model.py:84-85)
 *   VDS_EPI_GATE_RES   C = bf16(y), C2 = bf16(aux[m,n] + y*gate[m/rows_per_batch, n]),
 *                      y = acc + bias[n]                         (model.py:138-139,159-160,165)
 *   VDS_EPI_DGELU      C = bf16(acc * gelu'(aux[m,n]))           (backward of model.py:85)
 *   VDS_EPI_F32        C(f32) = acc; a K split (vds_gemm_args.split_k) runs over blockIdx.y and accumulates
 *                      atomically into the pre-zeroed C; the accumulate forms let several calls sum into one C
 * Requirements: K % 8 == 0 for NT / NN (any K for TN); N % 8 == 0 (TN: M % 8 == 0 too);
 * ld* % 8 == 0; every tensor < 4 GiB. */
enum { VDS_NT = 0, VDS_NN = 1, VDS_TN = 2 };
enum { VDS_EPI_STORE = 0, VDS_EPI_BIAS_GELU = 1, VDS_EPI_GATE_RES = 2, VDS_EPI_DGELU = 3, VDS_EPI_F32 = 4 };

typedef struct vds_gemm_args {
  int32_t layout, epilogue;
  int32_t M, N, K;
  const void* A; int64_t lda;
  const void* B; int64_t ldb;
  void* C; int64_t ldc;
  void* C2; int64_t ldc2;
  const void* bias;            /* bf16 [N] or NULL */
  const void* aux; int64_t ldaux; /* bf16 [M,N]: residual (GATE_RES) or pre-activation (DGELU) */
  const float* gate; int64_t ldgate; /* f32 [batch, ldgate], column n */
  int32_t rows_per_batch;
  int32_t split_k;             /* TN + F32 only: 0 = the library picks tiling and K split (atomics iff it splits; C
                                  pre-zeroed), -1 = the same but always accumulating atomically into C, 1 = no
                                  split, > 1 = that many splits + atomics, <= -2 = |split_k| splits, accumulate */
  float* colsum;               /* VDS_EPI_DGELU only, may be NULL: colsum[n] += sum_m C[m,n] (f32, atomically; the bias
                                  gradient of the layer below, model.py:84) -- fused into the epilogue of the 256^2
                                  kernel, a pass over C after the GEMM for the smaller tilings.  Needs C != NULL. */
} vds_gemm_args;

int vds_gemm_bf16(const vds_gemm_args* args, vds_stream_t stream);
/* Tests / experiments: pin the tiling vds_gemm_bf16 picks (0 = by its cost model (default), 128 = 128x128 tiles,
 * 256 = 256x256, 2 = 256x128 with two workgroups per CU); returns the previous setting, VDS_ERR_ARG for other
 * values.  Knob "gemm_tile" (VDS_GEMM_TILE sets the initial value).  Results are identical across tilings up
 * to fp32 summation order. */
int vds_gemm_force_tile(int32_t tile);
/* ------------------------------------------------------------ process-wide settings -----
 * Every tuning knob of the library lives in ONE table (csrc/config.h lists them with their defaults) that is filled from
 * the environment once, when the library is loaded; no entry point reads the environment afterwards.  vds_knob_set
 * changes an entry by name (e.g. "gemm_narrow", "cross_dkv16", "attn_wide_stores": same-process A/B measurements and
 * tests); VDS_ERR_ARG for an unknown name.  vds_knob_get returns the value, NaN for an unknown name.  Besides the
 * communicator this table is the library's only mutable global state. */
int vds_knob_set(const char* name, double value);
double vds_knob_get(const char* name);

/* Deterministic mode (default off).  The default backward accumulates with fp32 atomics in several places -- the split-K
 * slices of the weight-gradient GEMMs, the per-workgroup column sums of the modulation / bias / RMSNorm-weight gradients,
 * the lambda gradients -- so two runs differ in the last bits (autograd's reductions behind the reference's
 * train.py:431-433 have a fixed order).  on = 1: every one of them becomes a fixed-order reduction: partial results go to
 * `workspace` (caller-allocated, 16-byte aligned, `workspace_bytes` long; kept until the mode is switched off) with plain
 * stores and one pass sums them in index order.  Two backward passes over the same inputs then give bit-identical
 * gradients.  A split-K GEMM uses as many splits as fit the workspace (M*N*4 bytes each); row kernels need
 * 3 * D * 4 bytes per workgroup (<= 2048 workgroups).  All deterministic launches must be on one stream.  Returns the
 * previous mode, VDS_ERR_ARG for a bad argument. */
int vds_set_deterministic(int32_t on, void* workspace, size_t workspace_bytes);

/* The same GEMM with OCP fp8 operands (BASELINE config 5; no reference counterpart -- the reference trains in
 * bf16): layout VDS_NT: A[M,K] and B[N,K] one byte per element (a_fmt / b_fmt: 0 = e4m3fn, 1 = e5m2; B must
 * be e4m3fn), K, lda, ldb multiples of 16.  C = epilogue((sum_k A B) * *scale_a * *scale_b) with per-tensor
 * dequantisation factors read from device memory (NULL = 1).  Runs v_mfma_f32_16x16x128_f8f6f4 (2x the bf16
 * MFMA rate).  Input gradients are NT products with the transposed WEIGHT copy written by vds_quant_fp8.
 * layout VDS_TN (round 4; epilogue VDS_EPI_F32, a_fmt = 1 only): C[M,N] = sum_k A[k,m] B[k,n] with A [K,M] e5m2 and
 * B [K,N] e4m3, both row-major, K, M, lda, ldb multiples of 16 -- the weight gradient dW = dy^T x contracted straight
 * from the token-major fp8 copies (both operands k-major: ds_read_b64_tr_b8 fragments), split_k as vds_gemm_bf16.
 *
 * emit (may be NULL; VDS_EPI_BIAS_GELU / VDS_EPI_DGELU only): the epilogue also writes its result -- gelu(pre) resp.
 * acc * gelu'(aux), rounded to bf16 first -- as fp8 for the next GEMMs, so that no separate quantisation pass reads
 * it back: q[M,N] row-major and / or qt[N,M] transposed, scaled by fmax / *amax_in (delayed scaling: the caller
 * passes the amax recorded by the previous step), *dq_out = *amax_in / fmax, *amax_out = max(*amax_out, max |result|)
 * of this launch, colsum[n] += sum_m result[m,n] (bias gradient).  With emit, C2 (BIAS_GELU) / C (DGELU) may be NULL. */
typedef struct vds_fp8_out {
  void* q; int64_t ldq;
  void* qt; int64_t ldqt;
  const float* amax_in;
  float* amax_out;
  float* dq_out;
  int32_t fmt;      /* 0 = e4m3fn, 1 = e5m2 */
  float* colsum;    /* f32 [N] or NULL */
} vds_fp8_out;

int vds_gemm_fp8(const vds_gemm_args* args, const float* scale_a, const float* scale_b, int32_t a_fmt,
                 int32_t b_fmt, const vds_fp8_out* emit, vds_stream_t stream);

/* amax[0] = max(amax[0], max |x|) over a bf16 matrix x[M,K] (row stride ldx elements); the caller zeroes amax. */
int vds_absmax(const void* x, int64_t ldx, int32_t M, int32_t K, float* amax, vds_stream_t stream);

/* Per-tensor fp8 quantisation of a bf16 matrix: q[m,k] = sat(x[m,k] * fmax / *amax) (fmt 0: e4m3fn, fmax 448;
 * fmt 1: e5m2, fmax 57344), written row-major to q[M,K] (ldq, may be NULL) and / or transposed to qt[K,M] (ldt,
 * may be NULL); *dq_out = *amax / fmax is the factor vds_gemm_fp8 multiplies back in.  *amax == 0 -> scale 1.
 * amax_out (may be NULL): *amax_out = max(*amax_out, max |x|) -- with delayed scaling `amax` is the value a previous
 * step recorded and this call records the next one, so that no separate vds_absmax pass reads x. */
int vds_quant_fp8(const void* x, int64_t ldx, int32_t M, int32_t K, int32_t fmt, const float* amax, void* q,
                  int64_t ldq, void* qt, int64_t ldt, float* dq_out, float* amax_out, vds_stream_t stream);

/* qt[k,m] = q[m,k] for a one-byte-per-element matrix q[M,K] (row stride ldq; K, ldq multiples of 16, ldt of 4): the
 * k-contiguous copy of an fp8 operand whose row-major copy came straight out of its producer (the *_fp8 entry points
 * below).  1 B read + 1 B written per element against 2 + 1 + 1 of a vds_quant_fp8 pass over the bf16 tensor. */
int vds_transpose_fp8(const void* q, int64_t ldq, int32_t M, int32_t K, void* qt, int64_t ldt, vds_stream_t stream);

/* fp8-emitting forms of three producers of the fp8 GEMMs' operands (config 5): the bf16 result of the plain entry
 * point is not written; instead each value, rounded to bf16 first, is scaled by fmax / *amax_in (delayed scaling:
 * the amax the previous step recorded; 0 -> scale 1) and cast with saturation to q[rows, ldq] (fmt 0 = e4m3fn,
 * 1 = e5m2; ldq a multiple of 8) -- bit-identical to vds_quant_fp8 of the plain result.  *dq_out = *amax_in / fmax
 * (may be NULL).  max |result| is recorded per wave with plain stores (amax_part, f32 [B*L], may be NULL): entry w
 * receives the maximum over the rows wave w handled -- one entry per row / token for the RMSNorm and RoPE kernels, the
 * first 4 * (number of workgroups) entries for gate backward -- and the tensor's amax is the maximum over the array
 * (the caller zeroes it before the launch and reduces it afterwards; no atomics: ~10^5 short waves per launch).  All other arguments as in
 * vds_rmsnorm_mod_fwd / vds_gate_bwd / vds_qkv_rope_bwd (the latter needs hdp % 8 == 0 and H*hd <= 2048, else
 * VDS_ERR_UNSUPPORTED). */
int vds_rmsnorm_mod_fwd_fp8(const void* x, int64_t ldx, const void* w, const float* mod, int64_t ldmod,
                            int32_t shift_col, int32_t scale_col, void* q, int64_t ldq, int32_t fmt,
                            const float* amax_in, float* amax_part, float* dq_out, float* rstd, int32_t B, int32_t L,
                            int32_t D, float eps, vds_stream_t stream);
int vds_gate_bwd_fp8(const void* dxn, int64_t lddxn, const void* y, int64_t ldy, const float* mod, int64_t ldmod,
                     int32_t gate_col, void* q, int64_t ldq, int32_t fmt, const float* amax_in, float* amax_part,
                     float* dq_out, float* dmod, float* dbias, int32_t B, int32_t L, int32_t D, vds_stream_t stream);
int vds_qkv_rope_bwd_fp8(const void* dq, const void* dk, const void* dv, const float* cosb, const float* sinb,
                         const void* qkv_raw, const void* v0, const void* lam, float* dv0_acc, float* dlam, void* q,
                         int64_t ldq, int32_t fmt, const float* amax_in, float* amax_part, float* dq_out, int32_t mix,
                         int32_t add_dv0, int32_t B, int32_t L, int32_t H, int32_t hd, int32_t hdp,
                         vds_stream_t stream);

/* --------------------------------------------------------------- attention (MFMA) ----
 * F.scaled_dot_product_attention(q,k,v) full/non-causal (model.py:136,157), flash style.
 * q/k/v are addressed as base + b*stride_b + h*stride_h + l*stride_l (elements), rows of
 * head_dim contiguous bf16; O/dO the same.  lse [B,H,Lq] f32 = log-sum-exp of the scaled
 * scores (natural log).  head_dim in {64, 72, 128}. */
typedef struct vds_attn_args {
  int32_t B, H, Lq, Lk, head_dim;
  const void* q; int64_t q_sb, q_sh, q_sl;
  const void* k; int64_t k_sb, k_sh, k_sl;
  const void* v; int64_t v_sb, v_sh, v_sl;
  void* o; int64_t o_sb, o_sh, o_sl;
  float* lse;
  /* backward only */
  const void* d_o; int64_t do_sb, do_sh, do_sl;
  void* dq; int64_t dq_sb, dq_sh, dq_sl;
  void* dk; int64_t dk_sb, dk_sh, dk_sl;
  void* dv; int64_t dv_sb, dv_sh, dv_sl;
  float* delta;                /* workspace [2,B,H,Lq] f32: -rowsum(dO*O), then lse*log2(e) */
  /* nonzero = every k row carries 1.0 at columns head_dim and head_dim+1 and every v row 1.0 at columns
   * head_dim and head_dim+4 (zeros in the rest of [head_dim, head_dim+8)), as vds_qkv_rope_fwd writes
   * them for hdp >= hd+8; the kernels then fold the per-query constants of the softmax (max, lse, delta)
   * into their MFMAs instead of spending VALU instructions on them.  With the flag the q rows must be
   * padded to head_dim+8 columns as well: vds_attn_bwd OVERWRITES their columns head_dim, head_dim+1
   * (scratch: -lse*log2(e) as a bf16 hi/lo pair for the dK/dV kernel).
   * 2 (round 5; cross-attention, model.py:157): k / v carry the ones columns (vds_kv_pad_ones) but the q rows have NO pad
   * (token-major views of a linear layer's output): forward and dQ run on the ones-column kernels -- they keep q / dO in
   * registers and set the pad columns there --, dK/dV on the plain kernel; nothing is written into q. */
  int32_t kv_pad_ones;
  /* backward only: number of floats `delta` points to.  0 = exactly 2*B*H*Lq (the statistics).  With the size
   * vds_attn_bwd_workspace_bytes returns, the dK/dV kernel may split the query range over several workgroups per key
   * tile when the key sequence is short (cross-attention, model.py:157: 512 context keys = 4 key tiles per head) and
   * keep its fp32 partial sums behind the statistics. */
  int64_t ws_floats;
} vds_attn_args;

int vds_attn_fwd(const vds_attn_args* args, vds_stream_t stream);
int vds_attn_bwd(const vds_attn_args* args, vds_stream_t stream);
/* K / V of a cross-attention, token-major bf16 rows [B*Lk, ld] with head h of K at columns k_col0 + h*hd and of V at
 * v_col0 + h*hd (the context_kv output, model.py:150-156), copied into head-major padded rows kp, vp [B,H,Lk,hdp]
 * (hdp >= hd + 8, multiples of 8) with the ones columns of vds_attn_args.kv_pad_ones and zeros in the rest of the pad. */
int vds_kv_pad_ones(const void* kv, int64_t ld, int32_t k_col0, int32_t v_col0, void* kp, void* vp, int32_t B, int32_t Lk,
                    int32_t H, int32_t hd, int32_t hdp, vds_stream_t stream);
/* Tests / experiments: which of the head_dim-72 (ones-column) kernels run on v_mfma_f32_16x16x32_bf16 instead of
 * v_mfma_f32_32x32x16_bf16 -- bit 0: dK/dV, bit 1: dQ, bit 2: forward; -1 = back to the default (7); knob "attn_mfma16".
 * Returns the previous mask.  Results agree up to fp32 summation order. */
int vds_attn_set_variant(int32_t mask);
/* bytes of the caller-allocated `delta` workspace vds_attn_bwd wants for these B, H, Lq, Lk, head_dim, kv_pad_ones:
 * 2*B*H*Lq floats of statistics + the partial sums of a query-split dK/dV launch (see ws_floats) */
size_t vds_attn_bwd_workspace_bytes(const vds_attn_args* args);

/* ------------------------------------------------------ normalisation / modulation ---
 * y = bf16( rmsnorm(x)[*w] * (1 + scale[b]) + shift[b] ), rstd saved (model.py:34-41,123,144,164,389).
 * x,y [B*L, D] bf16; mod f32 [B, ldmod] with shift at column shift_col, scale at scale_col;
 * w bf16 [D] or NULL; rstd f32 [B*L]. */
int vds_rmsnorm_mod_fwd(const void* x, int64_t ldx, const void* w, const float* mod, int64_t ldmod,
                        int32_t shift_col, int32_t scale_col, void* y, int64_t ldy, float* rstd,
                        int32_t B, int32_t L, int32_t D, float eps, vds_stream_t stream);
/* dx = bf16( dres + d/dx ), dmod[b, shift_col..] += sum_l dy, dmod[b, scale_col..] += sum_l dy*xhat*w,
 * dw (f32 [D], atomically +=) if w != NULL.  dres may be NULL. */
int vds_rmsnorm_mod_bwd(const void* dy, int64_t lddy, const void* x, int64_t ldx, const void* w,
                        const float* mod, int64_t ldmod, int32_t shift_col, int32_t scale_col,
                        const float* rstd, const void* dres, int64_t lddres, void* dx, int64_t lddx,
                        float* dmod, float* dw, int32_t B, int32_t L, int32_t D, vds_stream_t stream);

/* gate backward of  x_new = x + y*gate  (model.py:139,160,165):
 *   dy = bf16(dx_new * gate[b]),  dmod[b, gate_col + n] += sum_l dx_new*y,
 *   dbias[n] (f32, atomically +=) += sum_{b,l} dy   if dbias != NULL. */
int vds_gate_bwd(const void* dxn, int64_t lddxn, const void* y, int64_t ldy, const float* mod,
                 int64_t ldmod, int32_t gate_col, void* dy, int64_t lddy, float* dmod, float* dbias,
                 int32_t B, int32_t L, int32_t D, vds_stream_t stream);

/* column sums: out[n] (f32, atomically +=) += sum_m x[m,n]  (bias gradients). */
int vds_colsum_bf16(const void* x, int64_t ldx, float* out, int32_t M, int32_t N, vds_stream_t stream);
/* the same over the token rows only of a [B * rows_per_sample] buffer: rows r with r % rows_per_sample < row_offset
 * (the register rows) are skipped; rows_per_sample = 0: every row */
int vds_colsum_bf16_rows(const void* x, int64_t ldx, float* out, int32_t M, int32_t N, int32_t rows_per_sample,
                         int32_t row_offset, vds_stream_t stream);

/* ----------------------------------------------------- qkv split / RoPE / residual-V --
 * forward of model.py:126-134: qkv [B,L,3D] token-major (k h d) ->
 *   q,k [B,H,L,hdp] rotated (fp32 math, half-split, cos/sin [L, hd/2] f32; model.py:266-275)
 *   v   [B,H,L,hdp] = lam*v_raw + (1-lam)*v0  (model.py:129-130) when v0 != NULL else v_raw.
 * hdp >= hd is the padded row length (pad columns are written as zero, except that with hdp >= hd+8
 * k gets 1.0 at columns hd, hd+1 and v 1.0 at columns hd and hd+4: see vds_attn_args.kv_pad_ones). lam: bf16 device scalar
 * (the bf16-cast lambda_param, as under the reference's bf16 param policy). */
int vds_qkv_rope_fwd(const void* qkv, const float* cosb, const float* sinb, const void* v0,
                     const void* lam, void* q, void* k, void* v, int32_t B, int32_t L, int32_t H,
                     int32_t hd, int32_t hdp, vds_stream_t stream);
/* backward: dq,dk,dv [B,H,L,hdp] -> dqkv [B,L,3D] (un-rotated, dv*lam);
 *   dv0_acc (f32 [B,H,L,hdp]) += (1-lam)*dv  when mix == 1;  dlam (f32 scalar) += sum dv*(v_raw - v0) when mix != 0;
 *   mix == 2 (round 5): as 1 without the dv0_acc update (dv0_acc may be NULL) -- the caller sums the blocks' terms in
 *   one pass with vds_dv0_reduce before block 0's call;
 *   when add_dv0 != 0 (block 0): dv_total = dv + dv0_acc. */
int vds_qkv_rope_bwd(const void* dq, const void* dk, const void* dv, const float* cosb,
                     const float* sinb, const void* qkv_raw, const void* v0, const void* lam,
                     float* dv0_acc, float* dlam, void* dqkv, int32_t mix, int32_t add_dv0, int32_t B,
                     int32_t L, int32_t H, int32_t hd, int32_t hdp, vds_stream_t stream);
/* residual-V (/root/reference/model.py:129-130, backward): out[b,h,l,:hd] (f32 [B,H,L,hdp]; + its old value when
 * accumulate != 0) = sum_i (1 - lam_i) dv_i[b,h,l,:hd] over n blocks.  dv / lam: HOST arrays of n DEVICE pointers (dv_i bf16
 * [B,H,L,hdp] contiguous, 16-byte aligned; lam_i one bf16).  Columns >= hd of out are not touched. */
int vds_dv0_reduce(const void* const* dv, const void* const* lam, int32_t n, float* out, int32_t accumulate, int32_t B,
                   int32_t H, int32_t L, int32_t hd, int32_t hdp, vds_stream_t stream);

/* apply_rotary_emb (model.py:266-275) on its own: y = rotate(x) for x, y [B,H,L,hd] bf16 addressed as
 * base + b*sb + h*sh + l*sl (elements, multiples of 4; rows of hd contiguous), cos / sin f32 [L, hd/2];
 * halves are (x1, x2) = (x[:hd/2], x[hd/2:]): y1 = x1 cos + x2 sin, y2 = x2 cos - x1 sin, fp32 math.
 * inverse != 0 applies the transposed rotation (the backward of the forward).  hd % 8 == 0. */
int vds_rope_apply(const void* x, int64_t x_sb, int64_t x_sh, int64_t x_sl, const float* cos, const float* sin,
                   void* y, int64_t y_sb, int64_t y_sh, int64_t y_sl, int32_t B, int32_t H, int32_t L, int32_t hd,
                   int32_t inverse, vds_stream_t stream);

/* cos/sin rows [n_reg + t*h*w, nt + 2*ns] f32 of ThreeDimRotary.forward (model.py:219-263) for
 * the offsets (st,sh,sw): gathered from the per-axis tables tab_t_* [128, nt], tab_s_* [128, ns]
 * (the factors of the reference's freqs_hwt buffers); register rows are cos=1, sin=0. */
int vds_rope_rows(const float* tab_t_cos, const float* tab_t_sin, const float* tab_s_cos,
                  const float* tab_s_sin, int32_t nt, int32_t ns, int32_t t, int32_t h, int32_t w,
                  int32_t st, int32_t sh, int32_t sw, int32_t n_reg, float* cosb, float* sinb,
                  vds_stream_t stream);

/* the same with the offsets (st, sh, sw) read from device memory (int32[3], clamped to the table): nothing
 * per-step remains in the arguments, so a captured whole-step graph can be replayed */
int vds_rope_rows_dev(const float* tab_t_cos, const float* tab_t_sin, const float* tab_s_cos,
                      const float* tab_s_sin, int32_t nt, int32_t ns, int32_t t, int32_t h, int32_t w,
                      const int32_t* start_dev, int32_t n_reg, float* cosb, float* sinb, vds_stream_t stream);

/* ------------------------------------------------------------- small-M linears (B rows) --
 * y[b, n] = act_out( sum_k act_in(x[b,k]) * W[n,k] + bias[n] ), M = B rows (the per-GPU batch; any M >= 1, run
 * in chunks of <= 16 rows that stay in registers): time_embed / adaLN_modulation / final_modulation
 * (model.py:90,318-322,339-341).  x f32 [M,K]; W bf16 [N,K]; bias bf16 [N]; y f32 [M,N].  act: 0 none, 1 SiLU.
 * K % 8 == 0. */
int vds_small_linear_fwd(const float* x, const void* W, const void* bias, float* y, int32_t M,
                         int32_t N, int32_t K, int32_t act_in, vds_stream_t stream);
/* dW[n,k] (f32) = sum_b dy[b,n]*act_in(x[b,k]);  dbias[n] (f32) = sum_b dy[b,n]  (overwritten);
 * dx[b,k] (f32) += act_in'(x[b,k]) * sum_n dy[b,n] W[n,k]   (atomically accumulated: the
 * conditioning vector c collects a gradient from every block).  dW/dbias or dx may be NULL. */
int vds_small_linear_bwd(const float* dy, const float* x, const void* W, float* dW, float* dbias,
                         float* dx, int32_t M, int32_t N, int32_t K, int32_t act_in,
                         vds_stream_t stream);
/* The same for nb weight sets sharing ONE input x (M <= 16 rows): the adaLN modulation linears of every DiT block
 * (model.py:89-94,107) in one launch instead of one per block.  W_ptrs / bias_ptrs / dW_ptrs / dbias_ptrs: device
 * arrays of nb device pointers (bias_ptrs / dbias_ptrs may be NULL); set i writes y + i * y_stride ([M,N] f32) and
 * reads its output gradient at dy + i * dy_stride; dx (shared, f32 [M,K]) accumulates over all sets. */
int vds_small_linear_fwd_batched(const float* x, const void* const* W_ptrs, const void* const* bias_ptrs, float* y,
                                 int64_t y_stride, int32_t nb, int32_t M, int32_t N, int32_t K, int32_t act_in,
                                 vds_stream_t stream);
int vds_small_linear_bwd_batched(const float* dy, int64_t dy_stride, const float* x, const void* const* W_ptrs,
                                 float* const* dW_ptrs, float* const* dbias_ptrs, float* dx, int32_t nb, int32_t M,
                                 int32_t N, int32_t K, int32_t act_in, vds_stream_t stream);
/* sinusoid [cos | sin](t * f_i), t unscaled (model.py:12-22), rounded through bf16 like
 * the reference's .to(x.dtype): out f32 [B, D]. */
int vds_timestep_embedding(const float* t, float* out, int32_t B, int32_t D, vds_stream_t stream);

/* ------------------------------------------------------------ patchify / unpatchify ---
 * latent [B,C,T,H,W] bf16 -> patches [B*N, C*pt*p*p] bf16, token order (h w t)
 * (Conv3d k=stride + rearrange, model.py:182-186); the GEMM with patch_proj follows. */
int vds_patchify(const void* latent, void* patches, int32_t B, int32_t C, int32_t T, int32_t H,
                 int32_t W, int32_t pt, int32_t p, vds_stream_t stream);
/* tokens y [B*N, p*p*pt*C] bf16 (p1 p2 p3 c) -> out [B,C,T,H,W] bf16 (model.py:392-401) */
int vds_unpatchify(const void* y, void* out, int32_t B, int32_t C, int32_t T, int32_t H, int32_t W,
                   int32_t pt, int32_t p, vds_stream_t stream);
/* gradient wrt tokens: dy[B*N, P] bf16 from dout [B,C,T,H,W] bf16 (inverse permutation) */
int vds_unpatchify_bwd(const void* dout, void* dy, int32_t B, int32_t C, int32_t T, int32_t H,
                       int32_t W, int32_t pt, int32_t p, vds_stream_t stream);
/* The same three with the tokens addressed as rows of the model's [B * (R + N)] token buffer: token n of sample b
 * is row b * rows_per_sample + row_offset + n (rows_per_sample = R + N, row_offset = R = 16 register tokens,
 * model.py:362,386), so that patch embedding and final layer run as ONE GEMM over all B * L rows instead of one
 * per sample; the R rows in front of every sample are not touched.  rows_per_sample = 0 means N (dense).
 * vds_unpatchify_rows: bwd = 0 tokens -> image, bwd != 0 image gradient -> token-row gradient. */
int vds_patchify_rows(const void* latent, void* patches, int32_t B, int32_t C, int32_t T, int32_t H, int32_t W,
                      int32_t pt, int32_t p, int32_t rows_per_sample, int32_t row_offset, vds_stream_t stream);
int vds_unpatchify_rows(const void* y, void* out, int32_t B, int32_t C, int32_t T, int32_t H, int32_t W,
                        int32_t pt, int32_t p, int32_t rows_per_sample, int32_t row_offset, int32_t bwd,
                        vds_stream_t stream);
/* x[b, 0:R] = reg[0:R] ; used to prepend the 16 register tokens (model.py:362) */
int vds_fill_registers(const void* reg, void* x, int64_t batch_stride, int32_t B, int32_t R,
                       int32_t D, vds_stream_t stream);
/* dreg[r,d] (f32) += sum_b dx[b, r, d] */
int vds_registers_bwd(const void* dx, int64_t batch_stride, float* dreg, int32_t B, int32_t R,
                      int32_t D, vds_stream_t stream);

/* ------------------------------------------------------------ noising + loss (train.py) --
 * z_t = x(1-t) + n t ; v = x - n  in bf16 (train.py:115-117); t f32 [B] already bf16-rounded. */
int vds_noise_latents(const void* x, const void* noise, const float* t, void* z_t, void* v,
                      int32_t B, int64_t per_sample, vds_stream_t stream);
/* loss = mean_b mean_chw (v - out)^2 in f32 (train.py:121-125), as a FIXED-ORDER two-stage reduction (per-workgroup
 * partial sums -> one pass that sums a sample's partials, divides by per_sample_n, then sums the samples in index order
 * and divides by B): loss_out[0] and per_sample[b] are WRITTEN (no pre-zeroing), bit-identical run to run, and a batch of
 * identical samples gives the single-sample loss to the bit.  workspace: vds_flow_loss_workspace_floats(B, per_sample_n)
 * floats, caller-allocated, contents irrelevant.  dout (may be NULL) = bf16( 2 (out - v) * gscale/(B*per_sample_n) ). */
int64_t vds_flow_loss_workspace_floats(int32_t B, int64_t per_sample_n);
int vds_flow_loss(const void* v, const void* out, float* loss_out, float* per_sample, void* dout,
                  float gscale, int32_t B, int64_t per_sample_n, float* workspace, vds_stream_t stream);
/* its backward for an arbitrary upstream gradient (autograd of train.py:121-125 under `(loss * s).backward()`,
 * loss scaling, micro-batch accumulation): dout = bf16( 2 (out - v) * *gloss_dev / (B*per_sample) ), the upstream
 * scalar read from device memory (no host synchronisation; graph-capturable).  B*per_sample % 8 == 0. */
int vds_flow_loss_bwd(const void* v, const void* out, const float* gloss_dev, void* dout, int32_t B,
                      int64_t per_sample_n, vds_stream_t stream);

/* ------------------------------------------------------------ sampler --
 * One Euler step of sampling/sample.py:139-146:  out = uncond + cfg_scale*(cond - uncond) in bf16
 * (uncond may be NULL: out = cond), acc (f32) += dt * out, latents = bf16(acc).  n % 8 == 0. */
int vds_cfg_euler_step(const void* cond, const void* uncond, float* acc, void* latents, float cfg_scale,
                       float dt, int64_t n, vds_stream_t stream);

/* -------------------------------------------------------------------- optimizer ------
 * Multi-tensor AdamW on fp32 master shards (torch.optim.AdamW(fused=True) semantics,
 * train.py:340-344,433) + bf16 shadow copy for the next all-gather.
 * desc: one vds_adamw_tensor per parameter (device array). */
typedef struct vds_adamw_tensor {
  float* p; const float* g; float* m; float* v; void* p_bf16; int64_t numel; float lr; float wd;
} vds_adamw_tensor;
/* work list: chunk i updates elements [chunk_start[i], chunk_start[i]+chunk_elems) of tensor
 * chunk_tensor[i]; lr_mult is the LR-schedule multiplier of this step (train.py:349-364,434),
 * step counts from 1, grad_scale multiplies every gradient (1.0 normally). */
int vds_adamw_multi(const vds_adamw_tensor* desc_dev, const int32_t* chunk_tensor_dev,
                    const int64_t* chunk_start_dev, int32_t n_chunks, int32_t chunk_elems,
                    float beta1, float beta2, float eps, int32_t step, float lr_mult,
                    float grad_scale, vds_stream_t stream);

/* the same with the per-step scalars in device memory: scalars_dev = { 1 - beta1^step, 1/sqrt(1 - beta2^step),
 * lr_mult } (graph replay: the host refreshes them with a copy before each replay) */
int vds_adamw_multi_dev(const vds_adamw_tensor* desc_dev, const int32_t* chunk_tensor_dev,
                        const int64_t* chunk_start_dev, int32_t n_chunks, int32_t chunk_elems, float beta1,
                        float beta2, float eps, const float* scalars_dev, float grad_scale, vds_stream_t stream);

/* ------------------------------------------- parameter / gradient sharding (model.py:512-542) ------
 * What the reference gets from FSDP2 `fully_shard` (apply_fsdp, model.py:512-542; mesh model.py:475-498): a bf16
 * all-gather of a shard group's parameters before use and an fp32 reduce-scatter-AVERAGE of its gradients after its
 * backward (MixedPrecisionPolicy(bf16, fp32), model.py:516-519), here straight on RCCL over xGMI, one process per
 * GPU, ONE collective per flat group buffer.  RCCL is bound at run time (dlopen of librccl.so.1; VDS_RCCL_PATH
 * overrides), so the library loads without it and these calls return VDS_ERR_UNSUPPORTED when it is missing.
 *
 *   rank 0:  vds_comm_unique_id(id, 128)  -> ship the 128 bytes to every rank (any host channel)
 *   all:     vds_comm_init(rank, world, id, 128)     on the calling thread's current HIP device
 *   per step and group:  vds_all_gather_bf16 / vds_reduce_scatter_f32_avg on the caller's communication stream
 *   end:     vds_comm_destroy()
 *
 * One communicator per process (the only global mutable state of the library).  All collectives are asynchronous
 * on `stream`; every rank must issue the same sequence.  shard_elems = elements per rank; the full buffer is
 * world * shard_elems, rank r's part at offset r * shard_elems (in-place allowed: shard == full + rank*shard_elems).
 * VDS_COMM_SCHEDULE=allpairs (read by vds_comm_init) switches the reduce-scatter from RCCL's algorithm to an explicit
 * all-pairs exchange over the point-to-point xGMI mesh (one ncclSend/ncclRecv per peer + a local averaging kernel
 * with a fixed summation order); it needs vds_reduce_scatter_workspace_bytes(shard_elems) bytes of caller memory. */
#define VDS_COMM_ID_BYTES 128
int vds_comm_available(void); /* VDS_OK when RCCL could be bound (local, no communication): every rank checks this and
                                 the ranks agree on it BEFORE any of them enters the collective vds_comm_init */
int vds_comm_unique_id(void* out, size_t bytes);
int vds_comm_init(int32_t rank, int32_t world, const void* unique_id, size_t bytes);
int vds_comm_info(int32_t* rank, int32_t* world, int32_t* rccl_version, int32_t* allpairs); /* NULLs allowed */
int vds_comm_destroy(void);
int vds_all_gather_bf16(const void* shard, void* full, int64_t shard_elems, vds_stream_t stream);
int vds_all_gather_f32(const float* shard, float* full, int64_t shard_elems, vds_stream_t stream); /* fp32 masters: checkpoints */
size_t vds_reduce_scatter_workspace_bytes(int64_t shard_elems); /* 0 unless the all-pairs schedule is on */
int vds_reduce_scatter_f32_avg(const float* full, float* shard, int64_t shard_elems, void* workspace,
                               size_t ws_bytes, vds_stream_t stream);
/* the local half of the all-pairs schedule: out[i] = (sum over ranks r of chunk_r[i]) / world in rank order, where
 * chunk_rank = own[n] and the other world-1 chunks lie in staged[(world-1) * n] in increasing rank order.  n % 4 == 0. */
int vds_average_chunks_f32(const float* own, const float* staged, float* out, int64_t n, int32_t world,
                           int32_t rank, vds_stream_t stream);
/* utils.py:11-15 avg_scalar_across_ranks (loss logging): buf[i] = mean over ranks, in place */
int vds_all_reduce_f32_avg(float* buf, int64_t n, vds_stream_t stream);

/* f32 -> bf16 cast (FSDP param_dtype cast before all-gather, model.py:516-518) */
int vds_cast_f32_bf16(const float* src, void* dst, int64_t n, vds_stream_t stream);
/* bf16 -> f32 */
int vds_cast_bf16_f32(const void* src, float* dst, int64_t n, vds_stream_t stream);

/* ------------------------------------------------ fp8 attention (BASELINE config 5) -------
 * F.scaled_dot_product_attention (model.py:136) and its backward on v_mfma_f32_16x16x128_f8f6f4: Q / K / V / P in
 * OCP e4m3, dO / dS in e5m2, fp32 accumulation and softmax statistics (csrc/attention_fp8.hip; recipe: fp8.py).  The
 * reference trains in bf16 only.  q, k, v, d_o: fp8 rows [B,H,L,128] (contiguous; bytes [0, head_dim) data, byte
 * head_dim of every V row = 1.0 (0x38), all other pad bytes 0) as written by vds_qkv_rope_fwd_fp8 / vds_attn_fp8_delta;
 * deq: device float[8] = {s_q, s_k, s_v, s_do, E, -, -, -}: dequantisation factors (x = x_q * s) and the exponent E with
 * s_q s_k log2(e) / sqrt(head_dim) = 2^-E exactly (the kernels' block-scaled MFMAs rely on it), written by the same two
 * producers.
 * o: bf16, last dim contiguous; lse f32 [B,H,Lq]; dq, dk, dv: bf16 strided like attention's.  Alignment contract of the
 * bf16 outputs (o, dq, dk, dv): strides multiples of 4 elements and 8-byte aligned bases (else VDS_ERR_ARG); when the
 * strides are multiples of 8 elements on a 16-byte aligned base the rows leave with 16-byte stores (faster; same bits).
 * The fp8 outputs o_q / dq_q (below): base and row stride multiples of 8 bytes (else VDS_ERR_ARG).
 * stats: the f32 [2,B,H,Lq] workspace vds_attn_fp8_delta filled (vds_attn_fp8_bwd_workspace_bytes).  head_dim 72. */
typedef struct vds_attn_fp8_args {
  int32_t B, H, Lq, Lk, head_dim;
  const void* q; const void* k; const void* v;
  void* o; int64_t o_sb, o_sh, o_sl;
  float* lse;
  const void* d_o;
  void* dq; int64_t dq_sb, dq_sh, dq_sl;
  void* dk; int64_t dk_sb, dk_sh, dk_sl;
  void* dv; int64_t dv_sb, dv_sh, dv_sl;
  const float* stats;
  const float* deq;
  /* round 4, optional (NULL = off): fp8 copies written by the kernels' epilogues in the layout of the following
   * linear layer's operand -- token-major [B*Lq, H*head_dim] bytes, row stride *_ld bytes -- so that no quantisation
   * pass reads the bf16 tensor back.  vds_attn_fp8_fwd: o_q = O as e4m3; vds_attn_fp8_bwd: dq_q = dQ as e5m2 (dq may
   * then be NULL: no bf16 dQ is written).  Values are the bf16-rounded results scaled by fmax / *e_amax_prev (delayed
   * scaling, saturating; 0 = unscaled), *e_dq_out = the dequantisation factor, *e_amax_cur = max(*e_amax_cur, max |x|). */
  void* o_q; int64_t o_q_ld;
  void* dq_q; int64_t dq_q_ld;
  const float* e_amax_prev; float* e_amax_cur; float* e_dq_out;
} vds_attn_fp8_args;
int vds_attn_fp8_supported(int32_t head_dim); /* 1 / 0 */
int vds_attn_fp8_fwd(const vds_attn_fp8_args* a, vds_stream_t stream);
size_t vds_attn_fp8_bwd_workspace_bytes(const vds_attn_fp8_args* a);
/* backward preprocess over the token-major bf16 O / dO ([B*Lq, H*head_dim], row strides o_sl / do_sl, batch strides
 * o_sb / do_sb, in elements): stats (see above), dO as e5m2 rows into doq ([B,H,Lq,128], pad bytes left untouched: zero
 * the buffer once), scaled by the previous step's amax (*amax_prev; 0 = unscaled) with the current one accumulated
 * into *amax_cur; writes deq[3] and reads deq[2]. */
int vds_attn_fp8_delta(const void* o, int64_t o_sb, int64_t o_sl, const void* d_o, int64_t do_sb, int64_t do_sl,
                       const float* lse, float* stats, void* doq, const float* amax_prev, float* amax_cur, float* deq,
                       int32_t B, int32_t H, int32_t Lq, int32_t head_dim, vds_stream_t stream);
int vds_attn_fp8_bwd(const vds_attn_fp8_args* a, vds_stream_t stream);
/* vds_qkv_rope_fwd with fp8 outputs (the values quantised are that kernel's bf16 results): q8 / k8 / v8 e4m3 rows
 * [B,H,L,128]; v_out (may be NULL): additionally the bf16 v in the padded head-major layout [B,H,L,hdp]; amax_prev /
 * amax_cur: the q, k, v amax entries at element stride amax_stride (delayed scaling: scale = 448 / previous amax);
 * writes deq[0..2] and deq[4] (q's factor is tied to k's: see deq above). */
int vds_qkv_rope_fwd_fp8(const void* qkv, const float* cosb, const float* sinb, const void* v0, const void* lam,
                         void* q8, void* k8, void* v8, void* v_out, const float* amax_prev, float* amax_cur,
                         int32_t amax_stride, float* deq, int32_t B, int32_t L, int32_t H, int32_t hd, int32_t hdp,
                         vds_stream_t stream);
/* The operands of cross-attention (model.py:146-157) as fp8 rows for vds_attn_fp8_*: q = the q_cross output
 * [B*Lq, H*hd] bf16 (contiguous) -> q8 [B,H,Lq,128]; kv = the context_kv output [B*Lk, 2*H*hd] (k columns, then v
 * columns) -> k8, v8 [B,H,Lk,128] with V's ones byte.  No rotation, no residual-V.  amax_prev / amax_cur / deq as in
 * vds_qkv_rope_fwd_fp8 (deq[0..2], deq[4]).  head_dim 72. */
int vds_cross_qkv_fp8(const void* q, const void* kv, void* q8, void* k8, void* v8, const float* amax_prev,
                      float* amax_cur, int32_t amax_stride, float* deq, int32_t B, int32_t Lq, int32_t Lk, int32_t H,
                      int32_t hd, vds_stream_t stream);

/* ------------------------------------------------------------------ live profiling ----
 * Per-kernel-class timing with HIP events recorded on the launch stream around each launch of
 * an enabled class (bench.py's roofline leg).  Off by default (no events, no overhead).
 * vds_prof_collect synchronises the recorded events, sums them per class and resets.
 * flops / bytes are the ALGORITHMIC work of the recorded launches (per kernel). */
enum { VDS_PROF_GEMM_NT = 0, VDS_PROF_GEMM_NN, VDS_PROF_GEMM_TN, VDS_PROF_ATTN_FWD, VDS_PROF_ATTN_BWD_DELTA,
       VDS_PROF_ATTN_BWD_DKV, VDS_PROF_ATTN_BWD_DQ, VDS_PROF_RMSNORM_FWD, VDS_PROF_RMSNORM_BWD, VDS_PROF_ADAMW,
       VDS_PROF_QKV_ROPE_FWD, VDS_PROF_QKV_ROPE_BWD, VDS_PROF_GATE_BWD,
       /* the attention kernel instances without the ones-column contract (cross-attention, hd 64/128) */
       VDS_PROF_ATTN_FWD_PLAIN, VDS_PROF_ATTN_BWD_DKV_PLAIN, VDS_PROF_ATTN_BWD_DQ_PLAIN, VDS_PROF_GEMM_FP8,
       VDS_PROF_ATTN_FP8_FWD, VDS_PROF_ATTN_FP8_DKV, VDS_PROF_ATTN_FP8_DQ,
       VDS_PROF_FP8_QUANT, /* vds_quant_fp8 / vds_transpose_fp8 / vds_absmax: the fp8 path's separate operand passes */
       VDS_PROF_NCLASS };
typedef struct vds_prof_stat { int64_t launches; double ms; double flops; double bytes; } vds_prof_stat;
int vds_prof_enable(uint32_t class_mask);
int vds_prof_collect(vds_prof_stat* out /* [VDS_PROF_NCLASS] */);
const char* vds_prof_class_name(int cls);

/* hardware self-test of the MFMA / LDS-transpose / LDS-DMA lane maps the kernels rely on.
 * scratch_dev: >= 2080 bytes of device memory; int32[8] mismatch counts are left at byte
 * offset 2048 (all zero = every map as assumed). */
int vds_selftest_lanemaps(void* scratch_dev, vds_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* VDS_H */
