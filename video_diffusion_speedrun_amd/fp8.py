"""fp8 linears of the DiT block (BASELINE config 5: "fp8 weights + activations on CDNA4 fp8 MFMA").

The reference trains in bf16 only (model.py:516-518), so the recipe is this build's own and is stated here:

  * every linear layer of a block -- qkv, mlp.0 (fc1), mlp.2 (fc2), and (round 3, `all_linears`) attn_proj, q_cross,
    context_kv, cross_proj -- runs on `vds_gemm_fp8` (v_mfma_f32_16x16x128_f8f6f4, 2x the bf16 MFMA rate) in forward,
    input gradient and weight gradient; the self-attention and (`cross_attention`, L x 512 context keys) the
    cross-attention products run on the same instruction (csrc/attention_fp8.hip, `attention`: e4m3 Q / K / V / P,
    e5m2 dO / dS, fp32 softmax statistics; P is rounded to the e4m3 grid in the log domain in the forward pass, see
    P_BYTE there; operands as 128-byte fp8 rows from vds_qkv_rope_fwd_fp8 / vds_cross_qkv_fp8); norms, modulation,
    residuals, loss and the optimizer stay as in the bf16 path;
  * OCP e4m3fn for activations and weights, e5m2 for gradients, fp32 accumulation, bf16 / fp32 outputs;
  * per-tensor scaling with saturating casts.  Weights were scaled by their current amax until round 3; since round 4
    they use the amax their previous quantisation pass recorded, like everything else.  Activations and
    gradients use delayed scaling in training: the scale comes from the amax the previous step recorded
    (`AmaxHistory`, one device table, no host synchronisation), and the pass that quantises a tensor records its
    current amax for the next step -- one read of the tensor instead of two.  With a history no operand is
    quantised by a separate pass over a bf16 tensor: gelu(fc1) and the fc2 input gradient ([tokens, 4D]) leave
    their GEMM's epilogue as fp8 (`emit`), and the RMSNorm+modulate outputs, the fc2 output gradient (gate
    backward) and the qkv output gradient (RoPE backward) leave their producer kernels as fp8 (`ops.*_fp8`; the
    k-contiguous copies are 1-byte transposes, `Q.from_rowmajor`).  The first step (no history) and no-grad
    forwards compute each tensor's own amax first;
  * every operand is quantised once per use site into a row-major copy and, where the backward pass contracts
    over its other index, a transposed copy, so that all three products of a linear layer are NT GEMMs:
        y  = x_q  W_q^T          dx = dy_q (W_q^T)^T         dW = dy_q^T^T ... = (dy^T)_q (x^T)_q^T
  * master weights, gradients and optimizer state remain fp32; the bf16 compute copy is the quantiser's input.

Parity: tests/test_fp8_gpu.py (kernels: exact against torch's float8 dtypes) and
tests/test_model_gpu.py::test_fp8_step_close_to_oracle (whole step against the fp32 oracle, tolerance stated there).
"""
from __future__ import annotations

import os

import torch

from . import ops
from ._lib import EPI_BIAS_GELU, EPI_DGELU, EPI_F32, EPI_GATE_RES, EPI_STORE

bf16, f32 = torch.bfloat16, torch.float32
E4M3, E5M2 = ops.FP8_E4M3, ops.FP8_E5M2
# rows of the delayed-scaling table per DiT block: 0 gelu(fc1), 1 d(fc2 input), 2 xn1 (qkv input), 3 xn3 (fc1 input),
# 4 d(mlp output), 5 d(qkv output); fp8 attention: 6 q, 7 k, 8 v (post-RoPE / lambda-mix), 9 d(attention output)
# the four 1152^2-class linears (`enable_fp8(all_linears=True)`): 10 attention output (attn_proj input), 11 cross-attention
# output (cross_proj input), 12 xn2 (q_cross input), 13 d(q_cross output), 14 d(context_kv output), 15 d(attn_proj
# output), 16 d(cross_proj output); one more row after the blocks' rows: the text context (context_kv input, shared)
# fp8 cross-attention (same kernels, Lk = context length): 17 q_cross output, 18 / 19 the k / v halves of the context_kv
# output, 20 d(cross-attention output)
# Round 4: the weight-gradient GEMMs contract the token-major (row-major) fp8 copies directly -- both operands k-major,
# fragments by ds_read_b64_tr_b8 (vds_gemm_fp8 with layout VDS_TN) -- so NO activation or gradient needs a transposed
# copy any more: the 1-byte transposes, the transposed halves of the quantiser passes and of the GEMM epilogues'
# emission are gone.  VDS_FP8_TN=0 restores the NT products of transposed copies (A/B).  Weights keep both copies
# (forward: W, input gradient: W^T; 1.3-5 M elements each).
TN = os.environ.get("VDS_FP8_TN", "1") != "0"
# round 4: 21 .. 27 the seven weights of a block (qkv, attn_proj, q_cross, context_kv, cross_proj, mlp.0, mlp.2): their scale is
# the amax of the previous step's quantisation pass too (a weight moves by ~lr per step and the cast saturates), which
# removes the 196 absmax launches (+ their zero fills) a DiT-XL step spent on reading every weight twice
ROWS = 28
ROW_W = 21
ROW_Q, ROW_DO = 6, 9
ROW_ATTN, ROW_CATT, ROW_XN2, ROW_DQC, ROW_DCKV, ROW_DY_AP, ROW_DY_CP = 10, 11, 12, 13, 14, 15, 16
ROW_QC, ROW_DOC = 17, 20


class Q:
    """a quantised matrix: row-major copy `q` [M,K], transposed copy `t` [K,M] (either may be None), factor `s`"""
    __slots__ = ("q", "t", "s", "rows", "cols")

    def __init__(self, x: torch.Tensor = None, fmt: int = 0, rowmajor: bool = True, transposed: bool = False,
                 hist: "AmaxHistory" = None, row: int = 0, weight: bool = False, remeasure: bool = False):
        """hist / row: this tensor's slot in the delayed-scaling table.  With a history the single quantisation pass
        scales by the previous step's amax and records the current one; without (first step, weights, inference)
        the tensor's own amax is computed first (and recorded, so that the next step has a history).
        transposed: the tensor is also a weight-gradient operand; with TN (default) that needs the row-major copy
        only, except for weights (`weight=True`), whose transposed copy feeds the input-gradient GEMM.
        remeasure: ignore the history for this tensor this time (its own amax, two passes) but still record: weights
        whose bf16 compute copy had to be re-cast, i.e. that were written by something else than the fused optimizer
        since the last step (load_state_dict, an EMA swap, a foreign optimizer) -- their recorded amax describes the old
        values and a larger tensor would be clipped at it."""
        if x is None:
            return
        if TN and transposed and not weight:
            rowmajor, transposed = True, False
        self.rows, self.cols = x.shape
        if hist is not None and hist.ready and not remeasure:
            self.q, self.t, self.s = ops.quant_fp8(x, fmt, hist.prev(row), rowmajor, transposed, amax_out=hist.cur(row))
            return
        amax = ops.absmax(x)
        if hist is not None:
            torch.maximum(hist.cur(row), amax, out=hist.cur(row))
        self.q, self.t, self.s = ops.quant_fp8(x, fmt, amax, rowmajor, transposed)

    @classmethod
    def empty(cls, M: int, K: int, fmt: int, rowmajor: bool, transposed: bool, device):
        """uninitialised buffers for a GEMM epilogue to fill (`emit_args`)"""
        if TN and transposed:
            rowmajor, transposed = True, False
        o = cls()
        o.rows, o.cols = M, K
        o.q = torch.empty(M, K, dtype=ops.fp8_dtypes[fmt], device=device) if rowmajor else None
        o.t = torch.empty(K, M, dtype=ops.fp8_dtypes[fmt], device=device) if transposed else None
        o.s = torch.empty(1, dtype=f32, device=device)
        return o

    @classmethod
    def from_rowmajor(cls, q: torch.Tensor, s: torch.Tensor, transposed: bool):
        """wrap the row-major fp8 copy a producer kernel emitted (`ops.*_fp8`); the transposed copy, where the
        backward pass needs one, is a 1-byte transpose of it (no second pass over a bf16 tensor)"""
        o = cls()
        o.rows, o.cols = q.shape
        o.q, o.s = q, s
        o.t = ops.transpose_fp8(q) if (transposed and not TN) else None
        return o

    def emit_args(self, fmt: int, amax_in, amax_out, colsum=None):
        return dict(q=self.q, qt=self.t, amax_in=amax_in, amax_out=amax_out, dq_out=self.s, fmt=fmt, colsum=colsum)


class AmaxHistory:
    """Delayed-scaling state of the quantised tensors: one (previous, current) amax pair per tensor in a single
    device table; `roll()` at the start of a training forward makes the last step's amax the scale source of
    this step.  No host synchronisation anywhere.

    `ready` (quantise with the previous step's amax) turns on only after a COMPLETE step -- forward and backward --
    has been recorded: the gradient rows are written in backward only, so a grad-enabled forward that is never
    followed by its backward (a validation loss outside no_grad, an exception) must not arm delayed scaling with
    amax 0 for them.  Once on, a row that recorded nothing in some later step keeps its older scale."""

    def __init__(self, n: int, device):
        self.tab = torch.zeros(n, 2, dtype=f32, device=device)
        # per-wave partial maxima of the producer kernels that emit fp8 themselves (`ops.*_fp8`: one f32 per token
        # row and tensor, plain stores; the current amax of such a tensor is the maximum over its row of this table),
        # allocated by the first training forward that knows the token count
        self.part_tab = None
        self.ready = False
        self._fwd_seen = False   # a training forward recorded its rows since the last roll
        self._bwd_seen = False   # ... and its backward completed (host flag set by DiT._backward_impl)

    def ensure_part(self, rows: int):
        """allocate the partial-maxima table for launches over `rows` token rows.  Called by every training forward
        BEFORE `roll()`, whether or not delayed scaling is armed yet: `roll()` must always contain the fold of the
        table (a HIP-graph capture of the first armed step would otherwise freeze a roll without it, and the rows the
        producer kernels feed would keep their first scale for ever)."""
        if self.part_tab is None or self.part_tab.shape[1] < rows:
            assert self.part_tab is None or not bool(self.part_tab.any()), "token count grew inside a step"
            self.part_tab = torch.zeros(self.tab.shape[0], rows, dtype=f32, device=self.tab.device)

    def roll(self):
        if self._fwd_seen:
            cur = self.tab[:, 1]
            if self.part_tab is not None:
                torch.maximum(cur, self.part_tab.amax(dim=1), out=cur)
                self.part_tab.zero_()
            self.tab[:, 0].copy_(torch.where(cur > 0, cur, self.tab[:, 0]))
            cur.zero_()
            if self._bwd_seen:
                self.ready = True
        self._fwd_seen, self._bwd_seen = True, False

    def reset(self):
        """Forget every recorded amax and disarm delayed scaling: the next training step quantises each tensor by its
        OWN amax again (two passes), and delayed scaling re-arms after that step's backward.  Called when parameter
        memory is replaced wholesale (DiT.load_state_dict / invalidate_compute_copy): since round 4 the weights are
        scaled by the previous step's amax too, and a checkpoint whose weights are larger than the ones the history was
        recorded on would otherwise be saturated at the old amax for one optimizer step, silently."""
        self.tab[:, 1].zero_()  # (the previous-step column is kept: a captured HIP graph may still read it)
        if self.part_tab is not None:
            self.part_tab.zero_()
        self.ready = False
        self._fwd_seen = self._bwd_seen = False

    def scratch(self, n: int):
        """n zeroed floats for a producer's amax record that nobody reads (no-grad forwards)"""
        if getattr(self, "_scratch", None) is None or self._scratch.numel() < n:
            self._scratch = torch.zeros(max(n, 8), dtype=f32, device=self.tab.device)
        return self._scratch[:n]

    def backward_done(self):
        self._bwd_seen = True

    def prev(self, i: int):
        return self.tab[i, 0:1]

    def cur(self, i: int):
        return self.tab[i, 1:2]

    def part(self, i: int, rows: int):
        """the partial-maxima row of tensor i for a launch over `rows` token rows"""
        if self.part_tab is None or self.part_tab.shape[1] < rows:
            self.ensure_part(rows)
        return self.part_tab[i]


def supported(M: int, N: int, K: int) -> bool:
    """alignment the fp8 GEMM needs for all three products of a [M,K] x [N,K] linear layer"""
    return M % 16 == 0 and N % 16 == 0 and K % 16 == 0


def fwd(xq: Q, wq: Q, out, bias=None):
    M, K, N = xq.rows, xq.cols, wq.rows
    ops.gemm_fp8(EPI_STORE, M, N, K, xq.q, wq.q, xq.s, wq.s, E4M3, out, out.stride(0), bias=bias)


def fwd_gelu(xq: Q, wq: Q, bias):
    M, K, N = xq.rows, xq.cols, wq.rows
    pre = torch.empty(M, N, dtype=bf16, device=xq.q.device)
    act = torch.empty(M, N, dtype=bf16, device=xq.q.device)
    ops.gemm_fp8(EPI_BIAS_GELU, M, N, K, xq.q, wq.q, xq.s, wq.s, E4M3, pre, N, act, N, bias=bias)
    return pre, act


def fwd_gelu_emit(xq: Q, wq: Q, bias, amax_in, amax_out, transposed: bool):
    """pre (bf16) and gelu(pre) directly as fp8 (row-major [+ transposed]); no bf16 activation is written"""
    M, K, N = xq.rows, xq.cols, wq.rows
    pre = torch.empty(M, N, dtype=bf16, device=xq.q.device)
    act = Q.empty(M, N, E4M3, True, transposed, xq.q.device)
    ops.gemm_fp8(EPI_BIAS_GELU, M, N, K, xq.q, wq.q, xq.s, wq.s, E4M3, pre, N, None, 0, bias=bias,
                 emit=act.emit_args(E4M3, amax_in, amax_out))
    return pre, act


def fwd_gate_res(xq: Q, wq: Q, bias, mod, gate_col: int, res, rows_per_batch: int):
    M, K, N = xq.rows, xq.cols, wq.rows
    y = torch.empty(M, N, dtype=bf16, device=xq.q.device)
    xn = torch.empty(M, N, dtype=bf16, device=xq.q.device)
    ops.gemm_fp8(EPI_GATE_RES, M, N, K, xq.q, wq.q, xq.s, wq.s, E4M3, y, N, xn, N, bias=bias, aux=res,
                 ldaux=res.stride(0), gate=mod[:, gate_col:], ldgate=mod.stride(0), rows_per_batch=rows_per_batch)
    return y, xn


def dgrad(dyq: Q, wq: Q, pre=None):
    """dx [M,K] = dy [M,N] W [N,K] as dy_q . (W^T)_q^T; with `pre`: times gelu'(pre)"""
    M, N, K = dyq.rows, dyq.cols, wq.cols
    dx = torch.empty(M, K, dtype=bf16, device=dyq.q.device)
    if pre is None:
        ops.gemm_fp8(EPI_STORE, M, K, N, dyq.q, wq.t, dyq.s, wq.s, E5M2, dx, K)
    else:
        ops.gemm_fp8(EPI_DGELU, M, K, N, dyq.q, wq.t, dyq.s, wq.s, E5M2, dx, K, aux=pre, ldaux=pre.stride(0))
    return dx


def dgrad_gelu_emit(dyq: Q, wq: Q, pre, amax_in, amax_out, colsum):
    """dh = (dy W) * gelu'(pre) directly as e5m2 (row-major + transposed) with its column sums (the bias gradient)
    accumulated into `colsum`; no bf16 dh is written"""
    M, N, K = dyq.rows, dyq.cols, wq.cols
    dh = Q.empty(M, K, E5M2, True, True, dyq.q.device)
    ops.gemm_fp8(EPI_DGELU, M, K, N, dyq.q, wq.t, dyq.s, wq.s, E5M2, None, 0, aux=pre, ldaux=pre.stride(0),
                 emit=dh.emit_args(E5M2, amax_in, amax_out, colsum))
    return dh


def wgrad(dyq: Q, xq: Q, dW: torch.Tensor, n_cu: int = 256):
    """dW [N,K] (f32, accumulated atomically: pre-zeroed by the model) = dy^T x as (dy^T)_q . (x^T)_q^T"""
    M, N, K = dyq.rows, dyq.cols, xq.cols
    tiles = ((N + 255) // 256) * ((K + 255) // 256)
    split = ops._wgrad_split(tiles, (M + 127) // 128, n_cu)
    if TN and dyq.q is not None and xq.q is not None and N % 16 == 0:
        ops.gemm_fp8(EPI_F32, N, K, M, dyq.q, xq.q, dyq.s, xq.s, E5M2, dW, K, split_k=-split, tn=True)
    else:
        ops.gemm_fp8(EPI_F32, N, K, M, dyq.t, xq.t, dyq.s, xq.s, E5M2, dW, K, split_k=-split)
