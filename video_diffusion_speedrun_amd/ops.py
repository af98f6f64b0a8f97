"""Tensor-level wrappers over the C ABI (include/vds.h).  torch is used for device memory and
the current HIP stream only; every computation below is a hand-written gfx950 kernel.

All tensors must live on the GPU; bf16 activations are torch.bfloat16, statistics / modulation
/ gradients of parameters are torch.float32.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import (EPI_BIAS_GELU, EPI_DGELU, EPI_F32, EPI_GATE_RES, EPI_STORE, VDS_NN, VDS_NT, VDS_TN,
                   AttnArgs, GemmArgs, check)

bf16 = torch.bfloat16
f32 = torch.float32


# private torch entry point (resolved once; a torch version without it falls back to the public, slower form)
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream() -> int:
    """raw handle of torch's current HIP stream.  (`torch.cuda.current_stream().cuda_stream` builds a Stream object per
    call: 2.65 us against 0.3 us for this form -- a third of the host cost of a small launch, and launch-bound
    workloads (C1: ~1600 launches of a few microseconds per step) are host-bound; tools/bench_host_overhead.py)"""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    if t is None:
        return None
    assert t.is_cuda, "vds ops need GPU tensors (no CPU fallback)"
    return t.data_ptr()


def _rows(t: torch.Tensor) -> Tuple[int, int]:
    """(ld, cols) of a 2-D row-major view whose last dim is contiguous."""
    assert t.dim() == 2 and t.stride(1) == 1, (t.shape, t.stride())
    return t.stride(0), t.shape[1]


# ------------------------------------------------------------------------------- GEMM ----
def gemm(layout: int, epi: int, M: int, N: int, K: int, A, lda, B, ldb, Cp=None, ldc=0, C2=None, ldc2=0,
         bias=None, aux=None, ldaux=0, gate=None, ldgate=0, rows_per_batch=0, split_k=1, colsum=None):
    a = GemmArgs(layout, epi, M, N, K, _p(A), lda, _p(B), ldb, _p(Cp), ldc, _p(C2), ldc2, _p(bias), _p(aux),
                 ldaux, _p(gate), ldgate, rows_per_batch, split_k, _p(colsum))
    check(_lib.load().vds_gemm_bf16(C.byref(a), _stream()), f"vds_gemm_bf16(layout={layout},epi={epi},M={M},N={N},K={K})")


def gemm_force_tile(tile: int) -> int:
    """tests / experiments: pin the GEMM tiling (0 auto, 128, 256, 2 = 256x128); returns the previous setting"""
    prev = _lib.load().vds_gemm_force_tile(int(tile))
    if prev < 0:
        raise ValueError(f"vds_gemm_force_tile({tile}): not a tiling")
    return prev


def knob_set(name: str, value: float) -> float:
    """set one entry of the library's knob table (csrc/config.h); returns the previous value"""
    lib = _lib.load()
    prev = lib.vds_knob_get(name.encode())
    check(lib.vds_knob_set(name.encode(), float(value)), f"vds_knob_set({name})")
    return prev


def knob_get(name: str) -> float:
    v = _lib.load().vds_knob_get(name.encode())
    if v != v:
        raise KeyError(name)
    return v


_det_ws = [None]


def is_deterministic() -> bool:
    return _det_ws[0] is not None


def set_deterministic(on: bool, workspace_bytes: int = 1 << 30, device="cuda") -> bool:
    """fixed-order reductions in every backward kernel (include/vds.h: vds_set_deterministic); the workspace for the
    partial results is a torch allocation this module keeps alive while the mode is on.  Returns the previous mode."""
    lib = _lib.load()
    if on:
        ws = torch.empty(int(workspace_bytes), dtype=torch.uint8, device=device)
        prev = lib.vds_set_deterministic(1, ws.data_ptr(), ws.numel())
        _det_ws[0] = ws
    else:
        torch.cuda.synchronize()
        prev = lib.vds_set_deterministic(0, None, 0)
        _det_ws[0] = None
    if prev < 0:
        raise ValueError("vds_set_deterministic")
    return bool(prev)


def linear_fwd(x: torch.Tensor, W: torch.Tensor, bias: Optional[torch.Tensor] = None,
               out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """y = x W^T + b.  x [M,K] bf16, W [N,K] bf16 -> y [M,N] bf16."""
    M, K = x.shape
    N = W.shape[0]
    y = out if out is not None else torch.empty(M, N, dtype=bf16, device=x.device)
    gemm(VDS_NT, EPI_STORE, M, N, K, x, x.stride(0), W, W.stride(0), y, y.stride(0), bias=bias)
    return y


def linear_fwd_gelu(x, W, bias):
    """(pre, act) = (x W^T + b, gelu_erf(pre))   (model.py:84-85)."""
    M, K = x.shape
    N = W.shape[0]
    pre = torch.empty(M, N, dtype=bf16, device=x.device)
    act = torch.empty(M, N, dtype=bf16, device=x.device)
    gemm(VDS_NT, EPI_BIAS_GELU, M, N, K, x, x.stride(0), W, W.stride(0), pre, N, act, N, bias=bias)
    return pre, act


def linear_fwd_gate_res(x, W, bias, mod, gate_col: int, res, rows_per_batch: int):
    """y = x W^T + b ; x_new = res + y * gate[b]  (model.py:138-139,159-160,165).
    mod: f32 [B, 9D] modulation table, gate at column gate_col."""
    M, K = x.shape
    N = W.shape[0]
    y = torch.empty(M, N, dtype=bf16, device=x.device)
    xn = torch.empty(M, N, dtype=bf16, device=x.device)
    gate = mod[:, gate_col:]
    gemm(VDS_NT, EPI_GATE_RES, M, N, K, x, x.stride(0), W, W.stride(0), y, N, xn, N, bias=bias, aux=res,
         ldaux=res.stride(0), gate=gate, ldgate=mod.stride(0), rows_per_batch=rows_per_batch)
    return y, xn


def linear_dgrad(dy, W, pre: Optional[torch.Tensor] = None, colsum: Optional[torch.Tensor] = None):
    """dx = dy W  (dy [M,N], W [N,K] -> [M,K]); with `pre`: dx *= gelu'(pre) (fused), and then optionally
    colsum[k] += sum_m dx[m,k] (f32; the bias gradient of the layer whose pre-activation `pre` is)."""
    M, N = dy.shape
    K = W.shape[1]
    dx = torch.empty(M, K, dtype=bf16, device=dy.device)
    if pre is None:
        assert colsum is None
        gemm(VDS_NN, EPI_STORE, M, K, N, dy, dy.stride(0), W, W.stride(0), dx, K)
    else:
        gemm(VDS_NN, EPI_DGELU, M, K, N, dy, dy.stride(0), W, W.stride(0), dx, K, aux=pre, ldaux=pre.stride(0),
             colsum=colsum)
    return dx


# --------------------------------------------------------------------------- fp8 GEMM ----
FP8_E4M3, FP8_E5M2 = 0, 1
fp8_dtypes = (torch.float8_e4m3fn, torch.float8_e5m2)


def absmax(x: torch.Tensor, amax: Optional[torch.Tensor] = None) -> torch.Tensor:
    """amax[0] = max |x| of a bf16 matrix (f32 device scalar; accumulates into `amax` when given)."""
    ld, K = _rows(x)
    if amax is None:
        amax = torch.zeros(1, dtype=f32, device=x.device)
    check(_lib.load().vds_absmax(_p(x), ld, x.shape[0], K, _p(amax), _stream()), "vds_absmax")
    return amax


def quant_fp8(x: torch.Tensor, fmt: int, amax: torch.Tensor, rowmajor: bool = True, transposed: bool = False,
              amax_out: Optional[torch.Tensor] = None):
    """bf16 [M,K] -> (q [M,K] fp8 | None, qt [K,M] fp8 | None, dq f32[1]) with per-tensor scale fmax / amax;
    amax_out (f32[1]) additionally accumulates max |x| (delayed scaling: the next step's `amax`)."""
    ld, K = _rows(x)
    M = x.shape[0]
    assert (M % 4 == 0 or not transposed) and K % 8 == 0
    q = torch.empty(M, K, dtype=fp8_dtypes[fmt], device=x.device) if rowmajor else None
    qt = torch.empty(K, M, dtype=fp8_dtypes[fmt], device=x.device) if transposed else None
    dq = torch.empty(1, dtype=f32, device=x.device)
    check(_lib.load().vds_quant_fp8(_p(x), ld, M, K, fmt, _p(amax), _p(q), K, _p(qt), M, _p(dq), _p(amax_out),
                                    _stream()), "vds_quant_fp8")
    return q, qt, dq


def gemm_fp8(epi: int, M: int, N: int, K: int, A, B, sa, sb, a_fmt: int, Cp=None, ldc=0, C2=None, ldc2=0, bias=None,
             aux=None, ldaux=0, gate=None, ldgate=0, rows_per_batch=0, split_k=1, emit=None, tn: bool = False):
    """C = epilogue((A[M,K] . B[N,K]^T) * sa * sb), A / B contiguous fp8 (A e4m3 or e5m2, B e4m3).
    tn: C[M,N] = sum_k A[k,m] B[k,n] with A [K,M] (e5m2) and B [K,N] (e4m3) row-major -- the weight gradient read
    straight from the token-major fp8 copies (EPI_F32 only).
    emit = dict(q=, qt=, amax_in=, amax_out=, dq_out=, fmt=, colsum=): fp8 copies of the epilogue result (vds_fp8_out)."""
    if tn:
        assert A.is_contiguous() and B.is_contiguous() and A.shape == (K, M) and B.shape == (K, N) and emit is None
        a = GemmArgs(VDS_TN, epi, M, N, K, _p(A), M, _p(B), N, _p(Cp), ldc, _p(C2), ldc2, _p(bias), _p(aux), ldaux,
                     _p(gate), ldgate, rows_per_batch, split_k, None)
        check(_lib.load().vds_gemm_fp8(C.byref(a), _p(sa), _p(sb), a_fmt, 0, None, _stream()),
              f"vds_gemm_fp8(tn,M={M},N={N},K={K})")
        return
    assert A.is_contiguous() and B.is_contiguous() and A.shape == (M, K) and B.shape == (N, K)
    a = GemmArgs(VDS_NT, epi, M, N, K, _p(A), K, _p(B), K, _p(Cp), ldc, _p(C2), ldc2, _p(bias), _p(aux), ldaux,
                 _p(gate), ldgate, rows_per_batch, split_k, None)
    e = None
    if emit is not None:
        q, qt = emit.get("q"), emit.get("qt")
        assert (q is None or q.shape == (M, N)) and (qt is None or qt.shape == (N, M))
        e = C.byref(_lib.Fp8Out(_p(q), N, _p(qt), M, _p(emit.get("amax_in")), _p(emit.get("amax_out")),
                                _p(emit.get("dq_out")), int(emit.get("fmt", 0)), _p(emit.get("colsum"))))
    check(_lib.load().vds_gemm_fp8(C.byref(a), _p(sa), _p(sb), a_fmt, 0, e, _stream()),
          f"vds_gemm_fp8(epi={epi},M={M},N={N},K={K})")


def _wgrad_split(tiles: int, kt: int, slots: int) -> int:
    """split-K factor of a weight-gradient GEMM: minimise (rounds of `slots` co-resident workgroups) x
    (K tiles per workgroup) plus the fp32-atomic traffic of the partial tiles (64 KB each at the
    chip-wide ~1.3 TB/s atomic rate, priced in units of one K-tile step ~1.5 us)."""
    best, best_cost = 1, None
    for s in range(1, 17):
        if s > 1 and kt // s < 8:
            break
        rounds = -(-(tiles * s) // slots)
        cost = rounds * -(-kt // s) + (tiles * s * 0.034 if s > 1 else 0.0)
        if best_cost is None or cost < best_cost - 1e-9:
            best, best_cost = s, cost
    return best


def linear_wgrad(dy, x, dW: torch.Tensor, accumulate: bool = False, split_k: Optional[int] = None):
    """dW[N,K] (f32) = dy^T x   (dy [M,N], x [M,K]).  The library picks tiling and split-K itself (split_k=None):
    a split over the tokens adds atomically into dW, which must then be pre-zeroed (the model zeroes its flat gradient
    buffers once per step); accumulate=True forces the atomic path so several calls sum into dW."""
    M, N = dy.shape
    K = x.shape[1]
    assert dW.dtype == f32 and dW.numel() == N * K and dW.is_contiguous()
    if split_k is None:
        split = -1 if accumulate else 0
    else:
        split = -max(2, split_k) if accumulate else split_k
    gemm(VDS_TN, EPI_F32, N, K, M, dy, dy.stride(0), x, x.stride(0), dW, K, split_k=split)


# -------------------------------------------------------------------------- attention ----
def _st(t):
    assert t.stride(3) == 1
    return t.stride(0), t.stride(1), t.stride(2)


def attn_fwd(q, k, v, o, lse, kv_pad_ones: bool = False):
    """q [B,H,Lq,hd], k/v [B,H,Lk,hd], o [B,H,Lq,hd] strided views (last dim contiguous); lse f32 [B,H,Lq].
    kv_pad_ones: k / v are views into qkv_rope_fwd outputs whose pad carries the ones columns."""
    B, H, Lq, hd = q.shape
    Lk = k.shape[2]
    a = AttnArgs()
    a.B, a.H, a.Lq, a.Lk, a.head_dim = B, H, Lq, Lk, hd
    a.q, (a.q_sb, a.q_sh, a.q_sl) = _p(q), _st(q)
    a.k, (a.k_sb, a.k_sh, a.k_sl) = _p(k), _st(k)
    a.v, (a.v_sb, a.v_sh, a.v_sl) = _p(v), _st(v)
    a.o, (a.o_sb, a.o_sh, a.o_sl) = _p(o), _st(o)
    a.lse = _p(lse)
    a.kv_pad_ones = int(kv_pad_ones)  # (True = 1: self-attention layout; 2: ones in k / v only, cross-attention)
    check(_lib.load().vds_attn_fwd(C.byref(a), _stream()), f"vds_attn_fwd(B={B},H={H},Lq={Lq},Lk={Lk},hd={hd})")


def kv_pad_ones(kv: torch.Tensor, B: int, Lk: int, H: int, hd: int, hdp: int, k_col0: int, v_col0: int):
    """kv [B*Lk, ld] token-major bf16 (the context_kv output) -> (k, v) [B,H,Lk,hdp] with the ones columns of the
    head_dim-72 attention kernels in the pad (vds_kv_pad_ones)"""
    kp = torch.empty(B, H, Lk, hdp, dtype=bf16, device=kv.device)
    vp = torch.empty(B, H, Lk, hdp, dtype=bf16, device=kv.device)
    check(_lib.load().vds_kv_pad_ones(_p(kv), kv.stride(0), k_col0, v_col0, _p(kp), _p(vp), B, Lk, H, hd, hdp, _stream()),
          "vds_kv_pad_ones")
    return kp, vp


def attn_bwd_workspace_floats(B, H, Lq, Lk, hd, kv_pad_ones: bool = False) -> int:
    """floats of workspace vds_attn_bwd wants: the [2,B,H,Lq] statistics + the fp32 partials of a query-split dK/dV launch"""
    a = AttnArgs()
    a.B, a.H, a.Lq, a.Lk, a.head_dim = B, H, Lq, Lk, hd
    a.kv_pad_ones = int(kv_pad_ones)
    return _lib.load().vds_attn_bwd_workspace_bytes(C.byref(a)) // 4


def attn_bwd(q, k, v, o, lse, do, dq, dk, dv, delta=None, kv_pad_ones: bool = False):
    """delta: f32 workspace of 2*B*H*Lq elements (rowsum(dO*O), then lse*log2 e); allocated here when None
    (size from vds_attn_bwd_workspace_bytes)."""
    B, H, Lq, hd = q.shape
    assert lse.is_contiguous()
    Lk = k.shape[2]
    a = AttnArgs()
    a.B, a.H, a.Lq, a.Lk, a.head_dim = B, H, Lq, Lk, hd
    a.kv_pad_ones = int(kv_pad_ones)
    if delta is None:
        delta = torch.empty(_lib.load().vds_attn_bwd_workspace_bytes(C.byref(a)) // 4, dtype=f32, device=q.device)
    assert delta.numel() >= 2 * B * H * Lq and delta.is_contiguous(), "attn_bwd: delta workspace is [2,B,H,Lq] f32"
    a.ws_floats = delta.numel()
    a.q, (a.q_sb, a.q_sh, a.q_sl) = _p(q), _st(q)
    a.k, (a.k_sb, a.k_sh, a.k_sl) = _p(k), _st(k)
    a.v, (a.v_sb, a.v_sh, a.v_sl) = _p(v), _st(v)
    a.o, (a.o_sb, a.o_sh, a.o_sl) = _p(o), _st(o)
    a.lse = _p(lse)
    a.d_o, (a.do_sb, a.do_sh, a.do_sl) = _p(do), _st(do)
    a.dq, (a.dq_sb, a.dq_sh, a.dq_sl) = _p(dq), _st(dq)
    a.dk, (a.dk_sb, a.dk_sh, a.dk_sl) = _p(dk), _st(dk)
    a.dv, (a.dv_sb, a.dv_sh, a.dv_sl) = _p(dv), _st(dv)
    a.delta = _p(delta)
    a.kv_pad_ones = int(kv_pad_ones)
    check(_lib.load().vds_attn_bwd(C.byref(a), _stream()), f"vds_attn_bwd(B={B},H={H},Lq={Lq},Lk={Lk},hd={hd})")


# ---------------------------------------------------------------------- fp8 attention ----
FP8_ROW = 128  # bytes of an fp8 head row ([B,H,L,128]: head_dim data bytes + pad)


def attn_fp8_supported(hd: int) -> bool:
    return bool(_lib.load().vds_attn_fp8_supported(int(hd)))


def qkv_rope_fwd_fp8(qkv, cos, sin, v0, lam, B, L, H, hd, hdp, amax_prev, amax_cur, amax_stride, deq, want_v=False):
    """qkv_rope_fwd with e4m3 outputs: -> (q8, k8, v8 [B,H,L,128] float8_e4m3fn, v bf16 [B,H,L,hdp] or None).
    amax_prev / amax_cur: f32 views whose elements 0, stride, 2*stride are the q, k, v amax of the previous / this
    step; deq: f32[8], entries 0..2 receive the dequantisation factors, entry 4 the exponent E that ties q's factor
    to k's (s_q s_k log2(e) / sqrt(hd) = 2^-E: include/vds.h)."""
    dev = qkv.device
    q8 = torch.empty(B, H, L, FP8_ROW, dtype=torch.float8_e4m3fn, device=dev)
    k8 = torch.empty_like(q8)
    v8 = torch.empty_like(q8)
    v = torch.empty(B, H, L, hdp, dtype=bf16, device=dev) if want_v else None
    check(_lib.load().vds_qkv_rope_fwd_fp8(_p(qkv), _p(cos), _p(sin), _p(v0), _p(lam), _p(q8), _p(k8), _p(v8), _p(v),
                                           _p(amax_prev), _p(amax_cur), amax_stride, _p(deq), B, L, H, hd, hdp,
                                           _stream()), "vds_qkv_rope_fwd_fp8")
    return q8, k8, v8, v


def cross_qkv_fp8(qc, ckv, B, Lq, Lk, H, hd, amax_prev, amax_cur, amax_stride, deq):
    """the cross-attention operands as fp8 rows: qc [B*Lq, H*hd] (q_cross output), ckv [B*Lk, 2*H*hd] (context_kv output:
    k columns, then v columns) -> (q8 [B,H,Lq,128], k8, v8 [B,H,Lk,128]) e4m3; amax / deq as qkv_rope_fwd_fp8"""
    dev = qc.device
    assert qc.is_contiguous() and ckv.is_contiguous() and qc.shape == (B * Lq, H * hd) and ckv.shape == (B * Lk, 2 * H * hd)
    q8 = torch.empty(B, H, Lq, FP8_ROW, dtype=torch.float8_e4m3fn, device=dev)
    k8 = torch.empty(B, H, Lk, FP8_ROW, dtype=torch.float8_e4m3fn, device=dev)
    v8 = torch.empty_like(k8)
    check(_lib.load().vds_cross_qkv_fp8(_p(qc), _p(ckv), _p(q8), _p(k8), _p(v8), _p(amax_prev), _p(amax_cur), amax_stride,
                                        _p(deq), B, Lq, Lk, H, hd, _stream()), "vds_cross_qkv_fp8")
    return q8, k8, v8


def _attn8_args(q8, k8, v8, deq, hd):
    B, H, Lq, row = q8.shape
    assert row == FP8_ROW and q8.is_contiguous() and k8.is_contiguous() and v8.is_contiguous()
    assert deq.dtype == f32 and deq.numel() >= 5 and deq.is_contiguous()
    a = _lib.Attn8Args()
    a.B, a.H, a.Lq, a.Lk, a.head_dim = B, H, Lq, k8.shape[2], hd
    a.q, a.k, a.v, a.deq = _p(q8), _p(k8), _p(v8), _p(deq)
    return a


def attn_fp8_fwd(q8, k8, v8, deq, o, lse, hd, emit=None):
    """fp8 self-attention forward: q8/k8/v8 from qkv_rope_fwd_fp8, o [B,H,Lq,hd] bf16 strided view, lse f32 [B,H,Lq].
    emit = (amax_prev f32[1], amax_cur f32[1]): also returns (o as e4m3 [B*Lq, H*hd], dequantisation factor f32[1]),
    written by the kernel's epilogue with the previous step's amax (the attn_proj / cross_proj operand)."""
    a = _attn8_args(q8, k8, v8, deq, hd)
    a.o, (a.o_sb, a.o_sh, a.o_sl) = _p(o), _st(o)
    a.lse = _p(lse)
    out = None
    if emit is not None:
        oq = torch.empty(a.B * a.Lq, a.H * hd, dtype=fp8_dtypes[0], device=q8.device)
        sc = torch.empty(1, dtype=f32, device=q8.device)
        a.o_q, a.o_q_ld = _p(oq), oq.stride(0)
        a.e_amax_prev, a.e_amax_cur, a.e_dq_out = _p(emit[0]), _p(emit[1]), _p(sc)
        out = (oq, sc)
    check(_lib.load().vds_attn_fp8_fwd(C.byref(a), _stream()), f"vds_attn_fp8_fwd(B={a.B},H={a.H},Lq={a.Lq},Lk={a.Lk})")
    return out


def attn_fp8_delta(o2d, do2d, lse, doq, amax_prev, amax_cur, deq, B, H, L, hd):
    """backward preprocess over the token-major bf16 o / dO [B*L, H*hd]: -> stats f32 [2,B,H,L]; fills doq (e5m2
    [B,H,L,128], pad bytes untouched: allocate it zeroed once), deq[3]; records dO's amax"""
    assert o2d.stride(1) == 1 and do2d.stride(1) == 1 and doq.is_contiguous() and doq.shape == (B, H, L, FP8_ROW)
    stats = torch.empty(2, B, H, L, dtype=f32, device=o2d.device)
    check(_lib.load().vds_attn_fp8_delta(_p(o2d), L * o2d.stride(0), o2d.stride(0), _p(do2d), L * do2d.stride(0),
                                         do2d.stride(0), _p(lse), _p(stats), _p(doq), _p(amax_prev), _p(amax_cur),
                                         _p(deq), B, H, L, hd, _stream()), "vds_attn_fp8_delta")
    return stats


def attn_fp8_bwd(q8, k8, v8, doq, stats, deq, dq, dk, dv, hd, emit_dq=None):
    """dq / dk / dv: bf16 strided [B,H,L,hd] views.  emit_dq = (amax_prev, amax_cur): dQ additionally -- or, with
    dq=None, only -- as e5m2 [B*Lq, H*hd] from the kernel's epilogue; returns (dq_q, dequantisation factor)."""
    a = _attn8_args(q8, k8, v8, deq, hd)
    a.d_o, a.stats = _p(doq), _p(stats)
    out = None
    if emit_dq is not None:
        dqq = torch.empty(a.B * a.Lq, a.H * hd, dtype=fp8_dtypes[1], device=q8.device)
        sc = torch.empty(1, dtype=f32, device=q8.device)
        a.dq_q, a.dq_q_ld = _p(dqq), dqq.stride(0)
        a.e_amax_prev, a.e_amax_cur, a.e_dq_out = _p(emit_dq[0]), _p(emit_dq[1]), _p(sc)
        out = (dqq, sc)
    if dq is not None:
        a.dq, (a.dq_sb, a.dq_sh, a.dq_sl) = _p(dq), _st(dq)
    a.dk, (a.dk_sb, a.dk_sh, a.dk_sl) = _p(dk), _st(dk)
    a.dv, (a.dv_sb, a.dv_sh, a.dv_sl) = _p(dv), _st(dv)
    check(_lib.load().vds_attn_fp8_bwd(C.byref(a), _stream()), f"vds_attn_fp8_bwd(B={a.B},H={a.H},Lq={a.Lq},Lk={a.Lk})")
    return out


def attn_fp8_qk_factors(amax_q: float, amax_k: float, hd: int):
    """(alpha_q, alpha_k, E) the fp8 RoPE producer derives from the two amax values: k onto 448, q so that
    log2(e) / sqrt(hd) / (alpha_q alpha_k) = 2^-E with its amax in (224, 448].  For tests and tools that build fp8
    operands by hand; the product path gets them from vds_qkv_rope_fwd_fp8."""
    import math
    alpha_k = 448.0 / amax_k
    cl = math.log2(math.e) / math.sqrt(hd)
    E = math.floor(math.log2(448.0 * alpha_k / (cl * amax_q)))
    return cl * 2.0 ** E / alpha_k, alpha_k, float(E)


def attn_set_variant(mask: int) -> int:
    """tests / A-B: which head_dim-72 attention kernels use the 16x16x32 MFMA shape (bit 0 dK/dV, 1 dQ, 2 forward;
    -1 = default); returns the previous mask"""
    return _lib.load().vds_attn_set_variant(int(mask))


def heads_view(t: torch.Tensor, B: int, L: int, H: int, hd: int, offset: int = 0) -> torch.Tensor:
    """[B*L, ld] token-major buffer -> [B,H,L,hd] view of columns offset + h*hd + d."""
    ld = t.stride(0)
    return t.as_strided((B, H, L, hd), (L * ld, hd, ld, 1), t.storage_offset() + offset)


# ---------------------------------------------------------- norm / modulation / gates ----
def rmsnorm_mod_fwd(x, w, mod, shift_col, scale_col, B, L, eps=1e-6):
    D = x.shape[1]
    y = torch.empty(B * L, D, dtype=bf16, device=x.device)
    rstd = torch.empty(B * L, dtype=f32, device=x.device)
    check(_lib.load().vds_rmsnorm_mod_fwd(_p(x), x.stride(0), _p(w), _p(mod), mod.stride(0), shift_col, scale_col,
                                          _p(y), D, _p(rstd), B, L, D, eps, _stream()), "vds_rmsnorm_mod_fwd")
    return y, rstd


def _fp8_out(rows, cols, fmt, device):
    return torch.empty(rows, cols, dtype=fp8_dtypes[fmt], device=device), torch.empty(1, dtype=f32, device=device)


def transpose_fp8(q: torch.Tensor) -> torch.Tensor:
    """fp8 [M,K] (contiguous, K % 16 == 0) -> [K,M]"""
    M, K = q.shape
    assert q.is_contiguous() and K % 16 == 0 and M % 4 == 0
    qt = torch.empty(K, M, dtype=q.dtype, device=q.device)
    check(_lib.load().vds_transpose_fp8(_p(q), K, M, K, _p(qt), M, _stream()), "vds_transpose_fp8")
    return qt


def _part_ok(t, rows):
    assert t is None or (t.dtype == f32 and t.is_contiguous() and t.numel() >= rows)


def rmsnorm_mod_fwd_fp8(x, w, mod, shift_col, scale_col, B, L, fmt, amax_in, amax_part, eps=1e-6):
    """rmsnorm_mod_fwd whose result leaves the kernel as fp8 (row-major) -> (q, dq f32[1], rstd);
    amax_part: f32 [>= B*L], zeroed by the caller: per-wave maxima of |result|; its maximum is the tensor's amax (or None)"""
    _part_ok(amax_part, B * L)
    D = x.shape[1]
    q, dq = _fp8_out(B * L, D, fmt, x.device)
    rstd = torch.empty(B * L, dtype=f32, device=x.device)
    check(_lib.load().vds_rmsnorm_mod_fwd_fp8(_p(x), x.stride(0), _p(w), _p(mod), mod.stride(0), shift_col, scale_col,
                                              _p(q), D, fmt, _p(amax_in), _p(amax_part), _p(dq), _p(rstd), B, L, D,
                                              eps, _stream()), "vds_rmsnorm_mod_fwd_fp8")
    return q, dq, rstd


def gate_bwd_fp8(dxn, y, mod, gate_col, dmod, dbias, B, L, fmt, amax_in, amax_part):
    """gate_bwd whose dy leaves the kernel as fp8 (row-major) -> (q, dq f32[1])"""
    _part_ok(amax_part, B * L)
    D = y.shape[1]
    q, dq = _fp8_out(B * L, D, fmt, y.device)
    check(_lib.load().vds_gate_bwd_fp8(_p(dxn), dxn.stride(0), _p(y), y.stride(0), _p(mod), mod.stride(0), gate_col,
                                       _p(q), D, fmt, _p(amax_in), _p(amax_part), _p(dq), _p(dmod), _p(dbias), B, L, D,
                                       _stream()), "vds_gate_bwd_fp8")
    return q, dq


def qkv_rope_bwd_fp8(dq, dk, dv, cos, sin, qkv_raw, v0, lam, dv0_acc, dlam, mix, add_dv0, B, L, H, hd, hdp, fmt,
                     amax_in, amax_part):
    """qkv_rope_bwd whose [B*L, 3D] result leaves the kernel as fp8 (row-major) -> (q, dq f32[1])"""
    _part_ok(amax_part, B * L)
    q, s = _fp8_out(B * L, 3 * H * hd, fmt, dq.device)
    check(_lib.load().vds_qkv_rope_bwd_fp8(_p(dq), _p(dk), _p(dv), _p(cos), _p(sin), _p(qkv_raw), _p(v0), _p(lam),
                                           _p(dv0_acc), _p(dlam), _p(q), 3 * H * hd, fmt, _p(amax_in), _p(amax_part),
                                           _p(s), int(mix), int(add_dv0), B, L, H, hd, hdp, _stream()),
          "vds_qkv_rope_bwd_fp8")
    return q, s


def rmsnorm_mod_bwd(dy, x, w, mod, shift_col, scale_col, rstd, dres, dmod, dw, B, L):
    D = x.shape[1]
    dx = torch.empty(B * L, D, dtype=bf16, device=x.device)
    check(_lib.load().vds_rmsnorm_mod_bwd(_p(dy), dy.stride(0), _p(x), x.stride(0), _p(w), _p(mod), mod.stride(0),
                                          shift_col, scale_col, _p(rstd), _p(dres),
                                          dres.stride(0) if dres is not None else 0, _p(dx), D, _p(dmod), _p(dw),
                                          B, L, D, _stream()), "vds_rmsnorm_mod_bwd")
    return dx


def gate_bwd(dxn, y, mod, gate_col, dmod, dbias, B, L):
    D = y.shape[1]
    dy = torch.empty(B * L, D, dtype=bf16, device=y.device)
    check(_lib.load().vds_gate_bwd(_p(dxn), dxn.stride(0), _p(y), y.stride(0), _p(mod), mod.stride(0), gate_col,
                                   _p(dy), D, _p(dmod), _p(dbias), B, L, D, _stream()), "vds_gate_bwd")
    return dy


def colsum(x, out, rows_per_sample: int = 0, row_offset: int = 0):
    """out[n] += sum_m x[m, n]; with rows_per_sample: only rows m with m % rows_per_sample >= row_offset (token rows)"""
    M, N = x.shape
    check(_lib.load().vds_colsum_bf16_rows(_p(x), x.stride(0), _p(out), M, N, rows_per_sample, row_offset, _stream()),
          "vds_colsum_bf16_rows")


# ----------------------------------------------------------------- qkv / rope / res-V ----
def qkv_rope_fwd(qkv, cos, sin, v0, lam, B, L, H, hd, hdp):
    q = torch.empty(B, H, L, hdp, dtype=bf16, device=qkv.device)
    k = torch.empty_like(q)
    v = torch.empty_like(q)
    check(_lib.load().vds_qkv_rope_fwd(_p(qkv), _p(cos), _p(sin), _p(v0), _p(lam), _p(q), _p(k), _p(v), B, L, H, hd,
                                       hdp, _stream()), "vds_qkv_rope_fwd")
    return q, k, v


def qkv_rope_bwd(dq, dk, dv, cos, sin, qkv_raw, v0, lam, dv0_acc, dlam, mix, add_dv0, B, L, H, hd, hdp):
    dqkv = torch.empty(B * L, 3 * H * hd, dtype=bf16, device=dq.device)
    check(_lib.load().vds_qkv_rope_bwd(_p(dq), _p(dk), _p(dv), _p(cos), _p(sin), _p(qkv_raw), _p(v0), _p(lam),
                                       _p(dv0_acc), _p(dlam), _p(dqkv), int(mix), int(add_dv0), B, L, H, hd, hdp,
                                       _stream()), "vds_qkv_rope_bwd")
    return dqkv


def dv0_reduce(dvs, lams, out, B, H, L, hd, hdp, accumulate: bool = False):
    """out[B,H,L,hdp] f32 (+)= sum_i (1 - lam_i) dvs[i][..., :hd]: the residual-V gradient of v_0 from all mixed blocks in one
    pass (vds_dv0_reduce; the blocks' RoPE backward then runs with mix = 2)"""
    import ctypes
    n = len(dvs)
    assert n == len(lams) and out.dtype == f32 and out.is_contiguous()
    for t in dvs:
        assert t.dtype == bf16 and t.is_contiguous() and tuple(t.shape) == (B, H, L, hdp)
    pa = (ctypes.c_void_p * max(n, 1))(*[t.data_ptr() for t in dvs])
    la = (ctypes.c_void_p * max(n, 1))(*[t.data_ptr() for t in lams])
    check(_lib.load().vds_dv0_reduce(ctypes.cast(pa, ctypes.c_void_p), ctypes.cast(la, ctypes.c_void_p), n, _p(out),
                                     int(accumulate), B, H, L, hd, hdp, _stream()), "vds_dv0_reduce")


def rope_apply(x, cos, sin, inverse: bool = False):
    """apply_rotary_emb (model.py:266-275): x [B,H,L,hd] bf16 (last dim contiguous), cos / sin f32 [L, hd/2]"""
    B, H, L, hd = x.shape
    assert x.dtype == bf16 and x.stride(3) == 1 and cos.dtype == f32 and cos.shape == (L, hd // 2) == sin.shape
    assert cos.is_contiguous() and sin.is_contiguous()
    y = torch.empty(B, H, L, hd, dtype=bf16, device=x.device)
    check(_lib.load().vds_rope_apply(_p(x), x.stride(0), x.stride(1), x.stride(2), _p(cos), _p(sin), _p(y),
                                     y.stride(0), y.stride(1), y.stride(2), B, H, L, hd, int(inverse), _stream()),
          "vds_rope_apply")
    return y


def rope_rows(tabs, thw, start, n_reg, device):
    """tabs = (t_cos, t_sin, s_cos, s_sin) device tables; returns cos, sin [n_reg + t*h*w, hd/2] f32."""
    t, h, w = thw
    nt, ns = tabs[0].shape[1], tabs[2].shape[1]
    rows = n_reg + t * h * w
    cos = torch.empty(rows, nt + 2 * ns, dtype=f32, device=device)
    sin = torch.empty_like(cos)
    if isinstance(start, torch.Tensor):  # offsets in device memory (int32[3]): whole-step graph replay
        assert start.is_cuda and start.dtype == torch.int32 and start.numel() == 3
        check(_lib.load().vds_rope_rows_dev(_p(tabs[0]), _p(tabs[1]), _p(tabs[2]), _p(tabs[3]), nt, ns, t, h, w,
                                            _p(start), n_reg, _p(cos), _p(sin), _stream()), "vds_rope_rows_dev")
        return cos, sin
    check(_lib.load().vds_rope_rows(_p(tabs[0]), _p(tabs[1]), _p(tabs[2]), _p(tabs[3]), nt, ns, t, h, w, start[0],
                                    start[1], start[2], n_reg, _p(cos), _p(sin), _stream()), "vds_rope_rows")
    return cos, sin


# ------------------------------------------------------------------- small linears ----
def small_linear_fwd(x, W, bias, act_in: int):
    M, K = x.shape
    N = W.shape[0]
    assert x.dtype == f32 and x.is_contiguous() and W.is_contiguous()
    y = torch.empty(M, N, dtype=f32, device=x.device)
    check(_lib.load().vds_small_linear_fwd(_p(x), _p(W), _p(bias), _p(y), M, N, K, act_in, _stream()),
          "vds_small_linear_fwd")
    return y


def small_linear_bwd(dy, x, W, dW, dbias, dx, act_in: int):
    M, N = dy.shape
    K = x.shape[1]
    assert dy.is_contiguous() and x.is_contiguous()
    check(_lib.load().vds_small_linear_bwd(_p(dy), _p(x), _p(W), _p(dW), _p(dbias), _p(dx), M, N, K, act_in,
                                           _stream()), "vds_small_linear_bwd")


def ptr_table(tensors) -> torch.Tensor:
    """device array of the tensors' device pointers (int64), for the batched entry points"""
    return torch.tensor([t.data_ptr() for t in tensors], dtype=torch.int64, device=tensors[0].device)


def small_linear_fwd_batched(x, W_tab, bias_tab, nb: int, N: int, act_in: int):
    """y[i] = act(x) W_i^T + b_i for the nb weight sets of the pointer tables; x f32 [M <= 16, K] -> f32 [nb, M, N]"""
    M, K = x.shape
    assert x.dtype == f32 and x.is_contiguous() and M <= 16
    y = torch.empty(nb, M, N, dtype=f32, device=x.device)
    check(_lib.load().vds_small_linear_fwd_batched(_p(x), _p(W_tab), _p(bias_tab), _p(y), M * N, nb, M, N, K, act_in,
                                                   _stream()), "vds_small_linear_fwd_batched")
    return y


def small_linear_bwd_batched(dy, x, W_tab, dW_tab, dbias_tab, dx, act_in: int):
    """dy f32 [nb, M, N]; writes every dW_i / dbias_i, accumulates dx (f32 [M, K]) over the sets"""
    nb, M, N = dy.shape
    K = x.shape[1]
    assert dy.is_contiguous() and x.is_contiguous() and M <= 16
    check(_lib.load().vds_small_linear_bwd_batched(_p(dy), M * N, _p(x), _p(W_tab), _p(dW_tab), _p(dbias_tab), _p(dx),
                                                   nb, M, N, K, act_in, _stream()), "vds_small_linear_bwd_batched")


def timestep_embedding(t, D):
    B = t.shape[0]
    out = torch.empty(B, D, dtype=f32, device=t.device)
    check(_lib.load().vds_timestep_embedding(_p(t), _p(out), B, D, _stream()), "vds_timestep_embedding")
    return out


# ------------------------------------------------------------- patches / registers ----
def patchify(x, pt, p, lead_rows: int = 0):
    """[B,C,T,H,W] -> patch rows; lead_rows = R: rows of a [B * (R + N)] token buffer (R zero rows per sample first)"""
    B, Cc, T, H, W = x.shape
    n = (T // pt) * (H // p) * (W // p)
    alloc = torch.zeros if lead_rows else torch.empty
    out = alloc(B * (lead_rows + n), Cc * pt * p * p, dtype=bf16, device=x.device)
    check(_lib.load().vds_patchify_rows(_p(x), _p(out), B, Cc, T, H, W, pt, p, lead_rows + n, lead_rows, _stream()),
          "vds_patchify_rows")
    return out


def unpatchify(y, B, Cc, T, H, W, pt, p, lead_rows: int = 0):
    n = (T // pt) * (H // p) * (W // p)
    out = torch.empty(B, Cc, T, H, W, dtype=bf16, device=y.device)
    check(_lib.load().vds_unpatchify_rows(_p(y), _p(out), B, Cc, T, H, W, pt, p, lead_rows + n, lead_rows, 0, _stream()),
          "vds_unpatchify_rows")
    return out


def unpatchify_bwd(dout, pt, p, lead_rows: int = 0):
    B, Cc, T, H, W = dout.shape
    n = (T // pt) * (H // p) * (W // p)
    alloc = torch.zeros if lead_rows else torch.empty
    dy = alloc(B * (lead_rows + n), Cc * pt * p * p, dtype=bf16, device=dout.device)
    check(_lib.load().vds_unpatchify_rows(_p(dout), _p(dy), B, Cc, T, H, W, pt, p, lead_rows + n, lead_rows, 1, _stream()),
          "vds_unpatchify_rows(bwd)")
    return dy


def fill_registers(reg, x, batch_stride, B, R, D):
    check(_lib.load().vds_fill_registers(_p(reg), _p(x), batch_stride, B, R, D, _stream()), "vds_fill_registers")


def registers_bwd(dx, batch_stride, dreg, B, R, D):
    check(_lib.load().vds_registers_bwd(_p(dx), batch_stride, _p(dreg), B, R, D, _stream()), "vds_registers_bwd")


# ---------------------------------------------------------------- noising / loss ----
def noise_latents(x, noise, t):
    B = x.shape[0]
    per = x.numel() // B
    zt = torch.empty_like(x)
    v = torch.empty_like(x)
    check(_lib.load().vds_noise_latents(_p(x), _p(noise), _p(t), _p(zt), _p(v), B, per, _stream()),
          "vds_noise_latents")
    return zt, v


def flow_loss(v, out, want_grad: bool, gscale: float = 1.0):
    B = v.shape[0]
    per = v.numel() // B
    lib = _lib.load()
    # [loss | per-sample | per-workgroup partials]: every word is written by the kernels (fixed-order two-stage reduction)
    acc = torch.empty(1 + B + lib.vds_flow_loss_workspace_floats(B, per), dtype=f32, device=v.device)
    dout = torch.empty_like(out) if want_grad else None
    check(lib.vds_flow_loss(_p(v), _p(out), _p(acc), _p(acc[1:]), _p(dout), gscale, B, per, _p(acc[1 + B:]), _stream()),
          "vds_flow_loss")
    return acc[0], acc[1:1 + B], dout


def flow_loss_bwd(v, out, gloss):
    """d loss / d out = 2 (out - v) gloss / numel, bf16; gloss: f32 device scalar (the upstream gradient)"""
    B = v.shape[0]
    assert gloss.is_cuda and gloss.dtype == f32 and gloss.numel() == 1 and out.is_contiguous() and v.is_contiguous()
    dout = torch.empty_like(out)
    check(_lib.load().vds_flow_loss_bwd(_p(v), _p(out), _p(gloss), _p(dout), B, v.numel() // B, _stream()),
          "vds_flow_loss_bwd")
    return dout


def cfg_euler_step(cond, uncond, acc, latents, cfg_scale: float, dt: float):
    """acc (f32) += dt * (uncond + cfg*(cond-uncond)) [bf16 math like the reference]; latents = bf16(acc)."""
    n = cond.numel()
    assert acc.dtype == f32 and acc.numel() == n and latents.numel() == n and cond.is_contiguous()
    check(_lib.load().vds_cfg_euler_step(_p(cond), _p(uncond), _p(acc), _p(latents), float(cfg_scale), float(dt), n,
                                         _stream()), "vds_cfg_euler_step")


def cast_f32_bf16(src, dst):
    check(_lib.load().vds_cast_f32_bf16(_p(src), _p(dst), src.numel(), _stream()), "vds_cast_f32_bf16")


def cast_bf16_f32(src, dst):
    check(_lib.load().vds_cast_bf16_f32(_p(src), _p(dst), src.numel(), _stream()), "vds_cast_bf16_f32")


def selftest_lanemaps(device="cuda"):
    scratch = torch.zeros(2080 // 4 + 8, dtype=torch.int32, device=device)
    check(_lib.load().vds_selftest_lanemaps(_p(scratch), _stream()), "vds_selftest_lanemaps")
    torch.cuda.synchronize()
    return scratch[512:520].tolist()


# ------------------------------------------------------------------ live profiling ----
def prof_enable(mask: int = 0xFFFFFFFF):
    """record HIP event pairs (on the launch stream) around every launch of the enabled kernel classes"""
    check(_lib.load().vds_prof_enable(mask & 0xFFFFFFFF), "vds_prof_enable")


def prof_collect():
    """-> {class name: dict(launches, ms, flops, bytes)} of the launches since prof_enable; resets."""
    lib = _lib.load()
    arr = (_lib.ProfStat * _lib.PROF_NCLASS)()
    check(lib.vds_prof_collect(C.cast(arr, C.c_void_p)), "vds_prof_collect")
    out = {}
    for i in range(_lib.PROF_NCLASS):
        if arr[i].launches:
            out[lib.vds_prof_class_name(i).decode()] = dict(launches=arr[i].launches, ms=arr[i].ms, flops=arr[i].flops,
                                                            bytes=arr[i].bytes)
    return out
