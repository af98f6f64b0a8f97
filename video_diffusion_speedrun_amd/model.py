"""Drop-in host API of the reference's `model.py` (DiT / DiTBlock / PatchEmbed / RMSNorm /
ThreeDimRotary / timestep_embedding / get_mup_setup) with the compute re-built as hand-written
gfx950 kernels behind the C ABI of include/vds.h.

Same constructor arguments, parameter names (state-dict keys) and call signatures as the
reference (model.py:279-292,358-402,404-465), so `train.py`-style code keeps working:

    dit = DiT(in_channels=16, patch_size=2, depth=28, num_heads=16, hidden_size=1152,
              cross_attn_input_size=4096, residual_v=True, train_bias_and_rms=False).to("cuda")
    out = dit(z_t, caption_encoded, t)          # [B,C,T,H,W] bf16
    loss.backward()                              # hand-written backward, fills p.grad (fp32)

What is different underneath (by design):
  * the modules below hold parameters only; `DiT.forward` runs ONE autograd node whose forward
    and backward are explicit sequences of HIP kernel launches (no torch compute ops);
  * parameters live in flat fp32 groups (params.py); kernels read a bf16 copy (the reference's
    bf16 param policy, model.py:516-518) and write fp32 gradients straight into the flat
    gradient buffer that `p.grad` aliases;
  * RoPE offsets can be pinned (`rope_start=`) -- by default they are drawn from the global CPU
    RNG with the reference's exact call sequence (model.py:223-226), so equal seeds give equal
    offsets.
There is no CPU fallback: a CPU tensor or a missing libvds_hip.so raises.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch
from torch import nn

import os

from . import fp8 as F8
from . import ops
from .params import FlatGroup

_NO_EMIT = os.environ.get("VDS_FP8_NO_EMIT") == "1"  # experiments: quantise every fp8 operand in a separate pass
_NO_PRODUCER_EMIT = os.environ.get("VDS_FP8_PRODUCER_EMIT") == "0"  # experiments: only the GEMM epilogues emit fp8
_NO_ATTN_EMIT = os.environ.get("VDS_FP8_ATTN_EMIT") == "0"  # experiments: attention results quantised in a separate pass
_CROSS_ONES = os.environ.get("VDS_CROSS_ONES", "1") != "0"  # head_dim 72: cross-attention forward / dQ on the ones-column kernels
# residual-V: d v_0 = sum over the mixed blocks of (1 - lambda_i) dv_i, summed in ONE pass before block 0's RoPE backward
# (ops.dv0_reduce) instead of an fp32 read-modify-write of the accumulator in every block (0: the per-block form)
_DV0_DEFER = os.environ.get("VDS_DV0_DEFER", "1") != "0"
_DV0_CHUNK = max(1, int(os.environ.get("VDS_DV0_CHUNK", "9")))  # dv tensors kept alive between two reductions


bf16, f32 = torch.bfloat16, torch.float32
N_REG = 16  # register tokens (model.py:316,362,386)
# head_dim -> row length of the head-major q / k / v buffers: head_dim itself, except 72 -> 96, which leaves room for the
# ones columns of the DiT-XL kernels.  The reference accepts any head_dim = hidden_size // num_heads that its RoPE table
# can be built for (a multiple of 8: model.py:192-209 reshapes arange(0, hd/2, 4) to hd/8 columns); so does this model up
# to 128 (round 5): the attention kernels are templates on a padded head dim (32 / 64 / 80+96 / 96 / 128) that fetch a
# row's chunks past head_dim as zeros, so e.g. 48 runs on the 64 instance and 112 on the 128 one (csrc/attention.hip,
# kernel_instance); head dims above 128 raise in DiT.__init__.
HDP_OF = {hd: hd for hd in range(8, 129, 8)}
HDP_OF[72] = 96


def timestep_embedding(t: torch.Tensor, dim: int, max_period: int = 10000) -> torch.Tensor:
    """[cos | sin](t * f_i), t unscaled (model.py:12-22).  HIP kernel; returns f32 [B, dim]
    holding bf16-rounded values (the reference casts the embedding to the model dtype)."""
    assert max_period == 10000
    return ops.timestep_embedding(t.to(f32).contiguous(), dim)


# ------------------------------------------------------------------ parameter holders ----
class _Linear(nn.Module):
    """nn.Linear-compatible parameter holder (weight [out,in], optional bias), reference init."""

    def __init__(self, in_features, out_features, bias=True):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        self.bias = nn.Parameter(torch.empty(out_features)) if bias else None
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if bias:
            bound = 1 / math.sqrt(in_features)
            nn.init.uniform_(self.bias, -bound, bound)


class _Conv3dParams(nn.Module):
    """Conv3d(kernel=stride) parameter holder: weight [D, C, pt, p, p], bias [D] (model.py:173-178)."""

    def __init__(self, in_channels, out_channels, kernel):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, *kernel))
        self.bias = nn.Parameter(torch.empty(out_channels))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        fan_in = in_channels * kernel[0] * kernel[1] * kernel[2]
        nn.init.uniform_(self.bias, -1 / math.sqrt(fan_in), 1 / math.sqrt(fan_in))


class _Act(nn.Module):
    """parameter-free placeholder keeping the reference's nn.Sequential indices (SiLU / GELU)."""

    def __init__(self, kind):
        super().__init__()
        self.kind = kind


def _bf16_copy(p: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    """bf16 compute copy of a stand-alone module's fp32 parameter (cast kernel, not a torch op)"""
    if p is None:
        return None
    if p.dtype == bf16:
        return p.detach().contiguous()
    out = torch.empty(p.shape, dtype=bf16, device=p.device)
    ops.cast_f32_bf16(p.detach().to(f32).contiguous(), out)
    return out


def _need_gpu(x, what):
    if not x.is_cuda:
        raise RuntimeError(f"video_diffusion_speedrun_amd.{what} runs on the GPU only (no CPU fallback)")


class _RMSNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, eps):
        D = x.shape[-1]
        x2 = x.reshape(-1, D).to(bf16).contiguous()
        rows = x2.shape[0]
        mod = torch.zeros(1, 2 * D, dtype=f32, device=x.device)  # shift = scale = 0: the bare norm of model.py:34-41
        w = _bf16_copy(weight)
        y, rstd = ops.rmsnorm_mod_fwd(x2, w, mod, 0, D, 1, rows, eps)
        ctx.save_for_backward(x2, w, mod, rstd)
        ctx.meta = (x.shape, x.dtype, weight.dtype if weight is not None else None)
        return y.view(x.shape).to(x.dtype)

    @staticmethod
    def backward(ctx, dy):
        x2, w, mod, rstd = ctx.saved_tensors
        shape, xdt, wdt = ctx.meta
        D = shape[-1]
        dmod = torch.zeros(1, 2 * D, dtype=f32, device=dy.device)
        dw = torch.zeros(D, dtype=f32, device=dy.device) if w is not None else None
        dx = ops.rmsnorm_mod_bwd(dy.reshape(-1, D).to(bf16).contiguous(), x2, w, mod, 0, D, rstd, None, dmod, dw, 1,
                                 x2.shape[0])
        return dx.view(shape).to(xdt), (dw.to(wdt) if dw is not None else None), None


class RMSNorm(nn.Module):
    """model.py:25-41: y = (x.float() * rsqrt(mean(x^2) + eps) [* weight]).to(x.dtype), over the last dim.
    Inside DiT the norm runs fused with the adaLN modulation (`vds_rmsnorm_mod_fwd` with the block's shift / scale);
    called on its own it is the same kernel with zero shift / scale, differentiable through `vds_rmsnorm_mod_bwd`."""

    def __init__(self, dim, eps=1e-6, trainable=False):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(dim)) if trainable else None

    def forward(self, x):
        _need_gpu(x, "RMSNorm")
        return _RMSNormFn.apply(x, self.weight, self.eps)


class _RotaryFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, cos, sin):
        hd = x.shape[-1]
        cos2 = cos.reshape(-1, hd // 2).to(f32).contiguous()
        sin2 = sin.reshape(-1, hd // 2).to(f32).contiguous()
        ctx.save_for_backward(cos2, sin2)
        ctx.xdt = x.dtype
        xb = x if x.dtype == bf16 and x.stride(-1) == 1 else x.to(bf16).contiguous()
        return ops.rope_apply(xb, cos2, sin2).to(x.dtype)

    @staticmethod
    def backward(ctx, dy):
        cos2, sin2 = ctx.saved_tensors
        dyb = dy if dy.dtype == bf16 and dy.stride(-1) == 1 else dy.to(bf16).contiguous()
        return ops.rope_apply(dyb, cos2, sin2, inverse=True).to(ctx.xdt), None, None


def apply_rotary_emb(x, cos, sin):
    """model.py:266-275: rotate the two halves of every head row, x [B,H,L,hd], cos / sin broadcastable
    [1,1,L,hd/2]; fp32 math, result in x.dtype.  (Inside DiTBlock the rotation is fused with the qkv head split.)"""
    _need_gpu(x, "apply_rotary_emb")
    if x.dim() != 4 or cos.numel() != x.shape[2] * (x.shape[3] // 2) or sin.numel() != cos.numel():
        raise ValueError(f"apply_rotary_emb: x {tuple(x.shape)} needs cos / sin of L x hd/2 = "
                         f"{x.shape[2]} x {x.shape[3] // 2} elements, got {tuple(cos.shape)}")
    return _RotaryFn.apply(x, cos, sin)


class _PatchEmbedFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, pt, p):
        if x.requires_grad:
            raise NotImplementedError("PatchEmbed: the gradient w.r.t. the input latents is not implemented (the "
                                      "train step never needs it: the latents are data)")
        B = x.shape[0]
        D = weight.shape[0]
        patches = ops.patchify(x.to(bf16).contiguous(), pt, p)           # [B*N, C*pt*p*p], tokens in (h w t) order
        y = ops.linear_fwd(patches, _bf16_copy(weight).view(D, -1), _bf16_copy(bias))
        ctx.save_for_backward(patches)
        ctx.meta = (weight.shape, weight.dtype, bias.dtype)
        return y.view(B, -1, D)

    @staticmethod
    def backward(ctx, dy):
        (patches,) = ctx.saved_tensors
        wshape, wdt, bdt = ctx.meta
        D = wshape[0]
        dy2 = dy.reshape(-1, D).to(bf16).contiguous()
        dW = torch.zeros(D, patches.shape[1], dtype=f32, device=dy.device)
        ops.linear_wgrad(dy2, patches, dW)
        db = torch.zeros(D, dtype=f32, device=dy.device)
        ops.colsum(dy2, db)
        return None, dW.view(wshape).to(wdt), db.to(bdt), None, None


class PatchEmbed(nn.Module):
    """model.py:170-186: Conv3d(kernel = stride = (pt, p, p)) + bias, tokens ordered (h w t): [B,C,T,H,W] ->
    [B, N, D] bf16.  A conv whose kernel equals its stride is a GEMM over gathered patches: `vds_patchify` +
    `vds_gemm_bf16`; backward = weight-gradient GEMM + column sums."""

    def __init__(self, patch_size=16, in_channels=3, embed_dim=768, time_patch_size=16):
        super().__init__()
        self.patch_proj = _Conv3dParams(in_channels, embed_dim, (time_patch_size, patch_size, patch_size))
        self.patch_size, self.time_patch_size = patch_size, time_patch_size

    def forward(self, x):
        _need_gpu(x, "PatchEmbed")
        return _PatchEmbedFn.apply(x, self.patch_proj.weight, self.patch_proj.bias, self.time_patch_size,
                                   self.patch_size)


class ThreeDimRotary(nn.Module):
    """model.py:189-263.  Keeps the per-axis factors of the reference's [128,128,128,d] cos/sin
    buffers (37 KB instead of 2 x 302 MB at head_dim 72); rows are gathered on the device."""

    def __init__(self, dim, base=100, h=128, w=128, t=128):
        super().__init__()
        assert h == 128 and w == 128 and t == 128
        self.dim, self.h, self.w, self.t = dim, h, w, t
        inv_s = 1.0 / (base ** (torch.arange(0, dim, 4).float() / dim))
        inv_t = 1.0 / (base ** (torch.arange(0, dim, 2).float() / dim))
        pos = torch.arange(128).float()
        ft, fs = torch.outer(pos, inv_t), torch.outer(pos, inv_s)
        self.register_buffer("tab_t_cos", ft.cos().contiguous(), persistent=False)
        self.register_buffer("tab_t_sin", ft.sin().contiguous(), persistent=False)
        self.register_buffer("tab_s_cos", fs.cos().contiguous(), persistent=False)
        self.register_buffer("tab_s_sin", fs.sin().contiguous(), persistent=False)

    def draw_start(self, thw) -> Tuple[int, int, int]:
        """(start_t, start_h, start_w); drawn h, w, t from the global CPU RNG (model.py:224-226)."""
        this_t, this_h, this_w = thw
        start_h = torch.randint(0, self.h - this_h + 1, (1,)).item()
        start_w = torch.randint(0, self.w - this_w + 1, (1,)).item()
        start_t = torch.randint(0, self.t - this_t + 1, (1,)).item()
        return int(start_t), int(start_h), int(start_w)

    def rows(self, thw, start, n_reg):
        tabs = (self.tab_t_cos, self.tab_t_sin, self.tab_s_cos, self.tab_s_sin)
        return ops.rope_rows(tabs, thw, start, n_reg, self.tab_t_cos.device)

    def forward(self, x, time_height_width=None, extend_with_register_tokens=0, start=None):
        if start is None:
            start = self.draw_start(time_height_width)
        cos, sin = self.rows(time_height_width, start, extend_with_register_tokens)
        return cos[None, None], sin[None, None]


class DiTBlock(nn.Module):
    """model.py:44-167.  `forward(x, context, c, v_0=None, rope=None) -> (x, v)` has the reference's signature and
    runs the block's HIP kernel sequence (`_fwd`) under one autograd node whose backward is `_bwd`; DiT.forward
    drives the same two methods for all blocks inside its own whole-model node."""

    def __init__(self, hidden_size, cross_attn_input_size, num_heads, mlp_ratio=4.0, qkv_bias=True,
                 residual_v=False):
        super().__init__()
        self.hidden_size, self.num_heads = hidden_size, num_heads
        self.head_dim = hidden_size // num_heads
        self.residual_v = residual_v
        self.norm1 = RMSNorm(hidden_size, trainable=qkv_bias)
        self.qkv = _Linear(hidden_size, hidden_size * 3, bias=qkv_bias)
        self.attn_proj = _Linear(hidden_size, hidden_size, bias=False)
        if residual_v:
            self.lambda_param = nn.Parameter(torch.tensor(0.5).reshape(1))
        if cross_attn_input_size is not None:
            self.norm2 = RMSNorm(hidden_size, trainable=qkv_bias)
            self.q_cross = _Linear(hidden_size, hidden_size, bias=qkv_bias)
            self.context_kv = _Linear(cross_attn_input_size, hidden_size * 2, bias=qkv_bias)
            self.cross_proj = _Linear(hidden_size, hidden_size, bias=False)
        else:
            self.norm2 = self.q_cross = self.context_kv = self.cross_proj = None
        self.norm3 = RMSNorm(hidden_size, trainable=qkv_bias)
        mlp_hidden = int(hidden_size * mlp_ratio)
        self.mlp = nn.Sequential(_Linear(hidden_size, mlp_hidden), _Act("gelu"), _Linear(mlp_hidden, hidden_size))
        self.adaLN_modulation = nn.Sequential(_Act("silu"), _Linear(hidden_size, 9 * hidden_size, bias=True))
        self.adaLN_modulation[-1].weight.data.zero_()
        self.adaLN_modulation[-1].bias.data.zero_()

    # ---- stand-alone call (reference signature, model.py:96-167) ---------------------------------------
    _own_group: Optional[FlatGroup] = None

    def _group_and_prefix(self, device):
        """the FlatGroup that holds this block's parameters + the name prefix they carry in it: the enclosing
        DiT's block group when there is a current one, else a group of its own (built on first use)"""
        named = list(self.named_parameters())
        g = getattr(named[0][1], "_vds_group", None)
        if g is not None and g.master is not None and g.is_current() and g.device == device:
            if g.world > 1:
                raise RuntimeError("DiTBlock.forward on a block of a sharded DiT: call the model (the sharding "
                                   "runtime gathers / reduces per block inside DiT.forward)")
            full0 = named[0][1]._vds_name
            pre = full0[:len(full0) - len(named[0][0])]
            if all(getattr(p, "_vds_group", None) is g and p._vds_name == pre + n for n, p in named):
                return g, pre
        g = FlatGroup("block", named)
        g.materialize(device)
        self._own_group = g
        return g, ""

    def forward(self, x, context, c, v_0=None, rope=None):
        """x [B,L,D], context [B,Lc,Cc] (None without cross-attention), c [B,D] conditioning, v_0 [B,H,L,hd] (block
        0's v; None in block 0), rope = (cos, sin) each [1,1,L,hd/2] or None -> (x [B,L,D], v [B,H,L,hd]) in bf16.
        Gradients flow to x, c, v_0 and the block's parameters (fp32 .grad); `context` is data (the frozen text
        encoder's output, train.py:70-87) and gets none."""
        _need_gpu(x, "DiTBlock")
        params = [p for p in self.parameters()]
        save = torch.is_grad_enabled()  # (always off inside an autograd.Function's forward: decided here)
        return _BlockFunction.apply(self, save, x, context, c, v_0, rope[0] if rope is not None else None,
                                    rope[1] if rope is not None else None, *params)

    # ---- the kernel sequence of one block (used by DiT's whole-model autograd node and by `forward`) ----
    def _fwd(self, G, pre, X, ctx2d, cvec, v0, cos, sin, B, L, Lc, save, fp8=None, mod=None):
        """X [B*L, D] bf16 token buffer -> (X_out, v [B,H,L,hdp], saved).  G: the FlatGroup holding this block's
        parameters under the name prefix `pre`; cvec f32 [B, D]; v0: block 0's v in the padded head-major layout
        or None; cos / sin f32 [L, hd/2]; fp8 = (AmaxHistory | None, block index) when the fp8 linears are on."""
        D, H, hd = self.hidden_size, self.num_heads, self.head_dim
        hdp = HDP_OF[hd]
        use_fp8 = fp8 is not None
        fp8_hist, i, fp8_attn, fp8_lin, q_ctx = fp8 if use_fp8 else (None, 0, 0, False, None)
        W = lambda n: G.w(pre + n)
        Wo = lambda n: G.w(pre + n) if G.has(pre + n) else None
        dev = X.device
        R0 = F8.ROWS * i
        moved = bool(getattr(G, "recast", True))  # fp8: this group's weights changed outside the fused optimizer
        if mod is None:  # (DiT.forward computes the modulation of all blocks in one batched launch and passes it in)
            mod = ops.small_linear_fwd(cvec, W("adaLN_modulation.1.weight"), W("adaLN_modulation.1.bias"), 1)
        # --- self attention (model.py:122-139)
        f8 = use_fp8 and F8.supported(B * L, 3 * D, D)
        if use_fp8 and not f8:
            raise ValueError(f"fp8 linears need B*L ({B * L}) and hidden size ({D}) to be multiples of 16")
        hist = fp8_hist if (f8 and save) else None
        emit = hist is not None and hist.ready and not _NO_EMIT  # delayed scaling: producers write fp8 themselves
        pemit = emit and not _NO_PRODUCER_EMIT
        xn1 = xn3 = None
        if pemit:
            fq, fs, rstd1 = ops.rmsnorm_mod_fwd_fp8(X, Wo("norm1.weight"), mod, 0, D, B, L, F8.E4M3,
                                                  hist.prev(F8.ROWS * i + 2), hist.part(F8.ROWS * i + 2, B * L))
            q_xn1 = F8.Q.from_rowmajor(fq, fs, save)
        else:
            xn1, rstd1 = ops.rmsnorm_mod_fwd(X, Wo("norm1.weight"), mod, 0, D, B, L)
        if f8:
            if not pemit:
                q_xn1 = F8.Q(xn1, F8.E4M3, True, save, hist, F8.ROWS * i + 2)
            q_wqkv = F8.Q(W("qkv.weight"), F8.E4M3, True, save, hist, R0 + F8.ROW_W + 0, weight=True, remeasure=moved)
            qkv = torch.empty(B * L, 3 * D, dtype=bf16, device=dev)
            F8.fwd(q_xn1, q_wqkv, qkv, Wo("qkv.bias"))
        else:
            qkv = ops.linear_fwd(xn1, W("qkv.weight"), Wo("qkv.bias"))
        mix = self.residual_v and v0 is not None
        attn = torch.empty(B * L, D, dtype=bf16, device=dev)
        lse1 = torch.empty(B, H, L, dtype=f32, device=dev)
        # fp8 attention (csrc/attention_fp8.hip): delayed scaling only -- q, k, v leave the RoPE kernel as e4m3 scaled by
        # the previous step's amax; until a complete step has been recorded the bf16 kernels run and the amax is taken
        a8_on = use_fp8 and fp8_attn and ops.attn_fp8_supported(hd)
        a8 = a8_on and fp8_hist.ready
        q = k = q8 = k8 = v8 = deq = e_attn = None
        if a8:
            r0 = F8.ROWS * i + F8.ROW_Q
            deq = torch.empty(8, dtype=f32, device=dev)  # {s_q, s_k, s_v, s_do, E}: include/vds.h, vds_attn_fp8_args
            cur = fp8_hist.tab[r0:r0 + 3, 1] if save else fp8_hist.scratch(5)[0::2]  # (no-grad forwards record nothing; the producers write elements 0, 2, 4 at amax_stride 2)
            q8, k8, v8, v = ops.qkv_rope_fwd_fp8(qkv, cos, sin, v0 if mix else None, W("lambda_param") if mix else None,
                                                 B, L, H, hd, hdp, fp8_hist.tab[r0:r0 + 3, 0], cur, 2, deq,
                                                 want_v=self.residual_v and v0 is None)
            # (the attn_proj operand leaves the attention epilogue as e4m3: no quantisation pass over `attn`)
            e_attn = ops.attn_fp8_fwd(q8, k8, v8, deq, ops.heads_view(attn, B, L, H, hd), lse1, hd,
                                      emit=(hist.prev(R0 + F8.ROW_ATTN), hist.cur(R0 + F8.ROW_ATTN))
                                      if (pemit and fp8_lin and not _NO_ATTN_EMIT) else None)
        else:
            q, k, v = ops.qkv_rope_fwd(qkv, cos, sin, v0 if mix else None, W("lambda_param") if mix else None, B, L, H,
                                       hd, hdp)
            ops.attn_fwd(q[..., :hd], k[..., :hd], v[..., :hd], ops.heads_view(attn, B, L, H, hd), lse1,
                         kv_pad_ones=(hdp - hd) >= 8)
            if a8_on and save:  # first steps: record the amax the next step quantises with
                for j, t in enumerate((q, k, v)):
                    ops.absmax(t.view(B * H * L, hdp)[:, :hd], fp8_hist.cur(F8.ROWS * i + F8.ROW_Q + j))
        f8l = f8 and fp8_lin  # the 1152^2-class linears in fp8 too
        q_attn = q_wap = q_xn2 = q_wqc = q_wkv = q_catt = q_wcp = None
        if f8l:
            q_attn = (F8.Q.from_rowmajor(*e_attn, save) if e_attn is not None else
                      F8.Q(attn, F8.E4M3, True, save, hist, R0 + F8.ROW_ATTN))
            q_wap = F8.Q(W("attn_proj.weight"), F8.E4M3, True, save, hist, R0 + F8.ROW_W + 1, weight=True, remeasure=moved)
            y_sa, X1 = F8.fwd_gate_res(q_attn, q_wap, None, mod, 2 * D, X, L)
        else:
            y_sa, X1 = ops.linear_fwd_gate_res(attn, W("attn_proj.weight"), None, mod, 2 * D, X, L)
        # --- cross attention (model.py:142-160)
        has_cross = G.has(pre + "q_cross.weight")
        if has_cross:
            f8c = f8l and q_ctx is not None and F8.supported(B * Lc, 2 * D, ctx2d.shape[1])
            xn2 = None
            if f8c and pemit:
                fq, fs, rstd2 = ops.rmsnorm_mod_fwd_fp8(X1, Wo("norm2.weight"), mod, 3 * D, 4 * D, B, L, F8.E4M3,
                                                      hist.prev(R0 + F8.ROW_XN2), hist.part(R0 + F8.ROW_XN2, B * L))
                q_xn2 = F8.Q.from_rowmajor(fq, fs, save)
            else:
                xn2, rstd2 = ops.rmsnorm_mod_fwd(X1, Wo("norm2.weight"), mod, 3 * D, 4 * D, B, L)
            if f8c:
                if q_xn2 is None:
                    q_xn2 = F8.Q(xn2, F8.E4M3, True, save, hist, R0 + F8.ROW_XN2)
                q_wqc = F8.Q(W("q_cross.weight"), F8.E4M3, True, save, hist, R0 + F8.ROW_W + 2, weight=True, remeasure=moved)
                qc = torch.empty(B * L, D, dtype=bf16, device=dev)
                F8.fwd(q_xn2, q_wqc, qc, Wo("q_cross.bias"))
                q_wkv = F8.Q(W("context_kv.weight"), F8.E4M3, True, save, hist, R0 + F8.ROW_W + 3, weight=True, remeasure=moved)
                ckv = torch.empty(B * Lc, 2 * D, dtype=bf16, device=dev)
                F8.fwd(q_ctx, q_wkv, ckv, Wo("context_kv.bias"))
            else:
                qc = ops.linear_fwd(xn2, W("q_cross.weight"), Wo("q_cross.bias"))
                ckv = ops.linear_fwd(ctx2d, W("context_kv.weight"), Wo("context_kv.bias"))
            catt = torch.empty(B * L, D, dtype=bf16, device=dev)
            lse2 = torch.empty(B, H, L, dtype=f32, device=dev)
            # fp8 cross-attention: the same kernels as the self-attention (Lk = context length), operands quantised by
            # vds_cross_qkv_fp8 with the previous step's amax; the bf16 kernels record the amax until a step is complete
            qc8 = kc8 = vc8 = deqc = e_catt = None
            c8_on = a8_on and fp8_attn >= 2 and Lc >= 4  # (vds_cross_qkv_fp8 works on tiles of 4 tokens)
            c8 = c8_on and fp8_hist.ready
            if c8:
                rc = R0 + F8.ROW_QC
                deqc = torch.empty(8, dtype=f32, device=dev)
                cur = fp8_hist.tab[rc:rc + 3, 1] if save else fp8_hist.scratch(5)[0::2]
                qc8, kc8, vc8 = ops.cross_qkv_fp8(qc, ckv, B, L, Lc, H, hd, fp8_hist.tab[rc:rc + 3, 0], cur, 2, deqc)
                e_catt = ops.attn_fp8_fwd(qc8, kc8, vc8, deqc, ops.heads_view(catt, B, L, H, hd), lse2, hd,
                                          emit=(hist.prev(R0 + F8.ROW_CATT), hist.cur(R0 + F8.ROW_CATT))
                                          if (pemit and f8c and not _NO_ATTN_EMIT) else None)
            else:
                if _CROSS_ONES and hd == 72:
                    # round 5: forward (and dQ) of the cross-attention on the ones-column 16x16x32 kernels of the
                    # self-attention: K / V get head-major padded copies with the ones columns (512 context rows: a 38 MB
                    # copy), the queries stay token-major (the kernels keep them in registers and set their pad columns there)
                    kp, vp = ops.kv_pad_ones(ckv, B, Lc, H, hd, hdp, 0, D)
                    ops.attn_fwd(ops.heads_view(qc, B, L, H, hd), kp[..., :hd], vp[..., :hd],
                                 ops.heads_view(catt, B, L, H, hd), lse2, kv_pad_ones=2)
                    del kp, vp  # (the backward pass makes them again from ckv: 12 us, nothing extra saved)
                else:
                    ops.attn_fwd(ops.heads_view(qc, B, L, H, hd), ops.heads_view(ckv, B, Lc, H, hd, 0),
                                 ops.heads_view(ckv, B, Lc, H, hd, D), ops.heads_view(catt, B, L, H, hd), lse2)
                if c8_on and save:
                    for j, t in enumerate((qc, ckv[:, :D], ckv[:, D:])):
                        ops.absmax(t, fp8_hist.cur(R0 + F8.ROW_QC + j))
            if f8c:
                q_catt = (F8.Q.from_rowmajor(*e_catt, save) if e_catt is not None else
                          F8.Q(catt, F8.E4M3, True, save, hist, R0 + F8.ROW_CATT))
                q_wcp = F8.Q(W("cross_proj.weight"), F8.E4M3, True, save, hist, R0 + F8.ROW_W + 4, weight=True, remeasure=moved)
                y_ca, X2 = F8.fwd_gate_res(q_catt, q_wcp, None, mod, 5 * D, X1, L)
            else:
                y_ca, X2 = ops.linear_fwd_gate_res(catt, W("cross_proj.weight"), None, mod, 5 * D, X1, L)
        else:
            X2 = X1
            f8c = False
        # --- MLP (model.py:163-165)
        if pemit:
            fq, fs, rstd3 = ops.rmsnorm_mod_fwd_fp8(X2, Wo("norm3.weight"), mod, 6 * D, 7 * D, B, L, F8.E4M3,
                                                  hist.prev(F8.ROWS * i + 3), hist.part(F8.ROWS * i + 3, B * L))
            q_xn3 = F8.Q.from_rowmajor(fq, fs, save)
        else:
            xn3, rstd3 = ops.rmsnorm_mod_fwd(X2, Wo("norm3.weight"), mod, 6 * D, 7 * D, B, L)
        if f8:
            if not pemit:
                q_xn3 = F8.Q(xn3, F8.E4M3, True, save, hist, F8.ROWS * i + 3)
            q_w1 = F8.Q(W("mlp.0.weight"), F8.E4M3, True, save, hist, R0 + F8.ROW_W + 5, weight=True, remeasure=moved)
            if emit:  # gelu(fc1) leaves the GEMM as fp8
                hpre, q_hact = F8.fwd_gelu_emit(q_xn3, q_w1, W("mlp.0.bias"), hist.prev(F8.ROWS * i), hist.cur(F8.ROWS * i), save)
                hact = None
            else:
                hpre, hact = F8.fwd_gelu(q_xn3, q_w1, W("mlp.0.bias"))
                q_hact = F8.Q(hact, F8.E4M3, True, save, hist, F8.ROWS * i)
            q_w2 = F8.Q(W("mlp.2.weight"), F8.E4M3, True, save, hist, R0 + F8.ROW_W + 6, weight=True, remeasure=moved)
            y_mlp, X3 = F8.fwd_gate_res(q_hact, q_w2, W("mlp.2.bias"), mod, 8 * D, X2, L)
            if hist is not None:
                G.recast = False  # every weight of the block has recorded its current amax
        else:
            hpre, hact = ops.linear_fwd_gelu(xn3, W("mlp.0.weight"), W("mlp.0.bias"))
            y_mlp, X3 = ops.linear_fwd_gate_res(hact, W("mlp.2.weight"), W("mlp.2.bias"), mod, 8 * D, X2, L)
        bs = None
        if save:
            bs = _Saved()
            bs.mod, bs.X, bs.X1, bs.X2 = mod, X, X1, X2
            bs.xn1, bs.rstd1, bs.qkv, bs.q, bs.k, bs.v, bs.attn, bs.lse1, bs.y_sa = xn1, rstd1, qkv, q, k, v, attn, lse1, y_sa
            bs.a8, bs.a8_on, bs.q8, bs.k8, bs.v8, bs.deq = a8, a8_on, q8, k8, v8, deq
            if a8:
                bs.v = None  # (block 0's bf16 v lives on as v_0; the backward contracts the fp8 copies)
            bs.mix, bs.has_cross = mix, has_cross
            if has_cross:
                bs.xn2, bs.rstd2, bs.qc, bs.ckv, bs.catt, bs.lse2, bs.y_ca = xn2, rstd2, qc, ckv, catt, lse2, y_ca
                bs.c8, bs.c8_on, bs.qc8, bs.kc8, bs.vc8, bs.deqc = c8, c8_on, qc8, kc8, vc8, deqc
                if c8:
                    bs.qc = bs.ckv = None  # the backward contracts the fp8 rows
            bs.xn3, bs.rstd3, bs.hpre, bs.hact, bs.y_mlp = xn3, rstd3, hpre, hact, y_mlp
            bs.f8, bs.f8l, bs.f8c = f8, f8l, f8c
            if f8:  # the backward contracts the fp8 copies: the bf16 GEMM inputs need not be kept
                bs.q_xn1, bs.q_wqkv, bs.q_xn3, bs.q_w1, bs.q_hact, bs.q_w2 = q_xn1, q_wqkv, q_xn3, q_w1, q_hact, q_w2
                bs.xn1 = bs.xn3 = bs.hact = None
            if f8l:
                bs.q_attn, bs.q_wap = q_attn, q_wap
                if not F8.TN:
                    q_attn.q = None  # (the weight gradient contracts the transposed copy only)
            if f8c:
                bs.q_xn2, bs.q_wqc, bs.q_wkv, bs.q_catt, bs.q_wcp = q_xn2, q_wqc, q_wkv, q_catt, q_wcp
                if not F8.TN:
                    q_catt.q = q_xn2.q = None
                bs.xn2 = None
        return X3, v, bs


    def _bwd(self, G, pre, bs, dX, sv, dc, dv0, B, L, Lc, first, fp8=None, dmod=None):
        """backward of `_fwd`: dX [B*L, D] bf16 -> d(input tokens); parameter gradients go to G's fp32 gradient
        buffer; dc (f32 [B, D]) and dv0 (f32 [B,H,L,hdp]) accumulate the conditioning / residual-V gradients.
        sv carries cos, sin, v0, ctx2d, cvec.  first: this is the block whose v was handed out as v_0."""
        D, H, hd = self.hidden_size, self.num_heads, self.head_dim
        hdp = HDP_OF[hd]
        fp8_hist, i, _, _, q_ctx = fp8 if fp8 is not None else (None, 0, False, False, None)
        R0 = F8.ROWS * i
        W = lambda n: G.w(pre + n)
        Wo = lambda n: G.w(pre + n) if G.has(pre + n) else None
        Gr = lambda n: G.g(pre + n)
        Go = lambda n: G.g(pre + n) if G.has(pre + n) else None
        dev = dX.device
        mod = bs.mod
        batched_adaln = dmod is not None  # DiT.backward: the adaLN weight gradients of all blocks in one launch at the end
        # (round 5 measured the weight-gradient GEMMs on a second stream: +3.3 % slower -- two MFMA kernels with different LDS
        # footprints fragment the CUs; removed in round 6, profiles/r05/wgrad_side_stream_ab.log)
        wgrad, wgrad8 = ops.linear_wgrad, F8.wgrad
        if dmod is None:
            dmod = torch.zeros(B, 9 * D, dtype=f32, device=dev)
        # --- MLP
        hist = fp8_hist if bs.f8 else None
        # (deterministic mode: the fc2 input gradient keeps its separate, fixed-order bias-gradient pass -- the GEMM epilogue's
        # fused column sums are per-tile atomics -- so the producers do not emit fp8 in backward)
        emit = hist is not None and hist.ready and not _NO_EMIT and not ops.is_deterministic()
        pemit = emit and not _NO_PRODUCER_EMIT
        if pemit:  # the fc2 output gradient leaves gate_bwd as e5m2
            fq, fs = ops.gate_bwd_fp8(dX, bs.y_mlp, mod, 8 * D, dmod, Gr("mlp.2.bias"), B, L, F8.E5M2,
                                    hist.prev(F8.ROWS * i + 4), hist.part(F8.ROWS * i + 4, B * L))
            q_dy = F8.Q.from_rowmajor(fq, fs, True)
            dy = None
        else:
            dy = ops.gate_bwd(dX, bs.y_mlp, mod, 8 * D, dmod, Gr("mlp.2.bias"), B, L)
        if bs.f8:
            if not pemit:
                q_dy = F8.Q(dy, F8.E5M2, True, True, hist, F8.ROWS * i + 4)
            wgrad8(q_dy, bs.q_hact, Gr("mlp.2.weight"))
            if emit:  # the fc2 input gradient leaves the GEMM as e5m2 (+ transposed, + bias gradient)
                dh = None
                q_dh = F8.dgrad_gelu_emit(q_dy, bs.q_w2, bs.hpre, hist.prev(F8.ROWS * i + 1), hist.cur(F8.ROWS * i + 1),
                                          Gr("mlp.0.bias"))
            else:
                dh = F8.dgrad(q_dy, bs.q_w2, pre=bs.hpre)
                q_dh = F8.Q(dh, F8.E5M2, True, True, hist, F8.ROWS * i + 1)
                ops.colsum(dh, Gr("mlp.0.bias"))
            wgrad8(q_dh, bs.q_xn3, Gr("mlp.0.weight"))
            dxn = F8.dgrad(q_dh, bs.q_w1)
            del q_dy, q_dh
        else:
            wgrad(dy, bs.hact, Gr("mlp.2.weight"))
            dh = ops.linear_dgrad(dy, W("mlp.2.weight"), pre=bs.hpre, colsum=Gr("mlp.0.bias"))  # + fc1 bias gradient
            wgrad(dh, bs.xn3, Gr("mlp.0.weight"))
            dxn = ops.linear_dgrad(dh, W("mlp.0.weight"))
        del dh
        dX2 = ops.rmsnorm_mod_bwd(dxn, bs.X2, Wo("norm3.weight"), mod, 6 * D, 7 * D, bs.rstd3, dX, dmod,
                                  Go("norm3.weight"), B, L)
        # --- cross attention
        def gate_bwd_q(dXo, y, col, row):
            """the gradient w.r.t. a gated projection's output as an fp8 operand (e5m2, row-major + transposed)"""
            if pemit:
                fq, fs = ops.gate_bwd_fp8(dXo, y, mod, col, dmod, None, B, L, F8.E5M2, hist.prev(row), hist.part(row, B * L))
                return F8.Q.from_rowmajor(fq, fs, True)
            return F8.Q(ops.gate_bwd(dXo, y, mod, col, dmod, None, B, L), F8.E5M2, True, True, hist, row)

        if bs.has_cross:
            if bs.f8c:
                q_dy = gate_bwd_q(dX2, bs.y_ca, 5 * D, R0 + F8.ROW_DY_CP)
                wgrad8(q_dy, bs.q_catt, Gr("cross_proj.weight"))
                dcatt = F8.dgrad(q_dy, bs.q_wcp)
                del q_dy
            else:
                dy = ops.gate_bwd(dX2, bs.y_ca, mod, 5 * D, dmod, None, B, L)
                wgrad(dy, bs.catt, Gr("cross_proj.weight"))
                dcatt = ops.linear_dgrad(dy, W("cross_proj.weight"))
            # fp8 cross-attention + fp8 q_cross: dQ leaves the attention kernel as e5m2 (and as bf16 only for a bias gradient)
            emit_dqc = bs.c8 and bs.f8c and pemit and not _NO_ATTN_EMIT
            need_dqc = not emit_dqc or G.has(pre + "q_cross.bias")
            dqc = torch.empty(B * L, D, dtype=bf16, device=dev) if need_dqc else None
            dckv = torch.empty(B * Lc, 2 * D, dtype=bf16, device=dev)
            e_dqc = None
            if bs.c8:
                rd = R0 + F8.ROW_DOC
                if getattr(sv, "doq", None) is None:  # one e5m2 dO buffer per backward pass; its pad bytes stay zero
                    sv.doq = torch.zeros(B, H, L, ops.FP8_ROW, dtype=torch.float8_e5m2, device=dev)
                stats = ops.attn_fp8_delta(bs.catt, dcatt, bs.lse2, sv.doq, fp8_hist.prev(rd), fp8_hist.cur(rd), bs.deqc,
                                           B, H, L, hd)
                e_dqc = ops.attn_fp8_bwd(bs.qc8, bs.kc8, bs.vc8, sv.doq, stats, bs.deqc,
                                         ops.heads_view(dqc, B, L, H, hd) if need_dqc else None,
                                         ops.heads_view(dckv, B, Lc, H, hd, 0), ops.heads_view(dckv, B, Lc, H, hd, D), hd,
                                         emit_dq=(hist.prev(R0 + F8.ROW_DQC), hist.cur(R0 + F8.ROW_DQC))
                                         if emit_dqc else None)
            else:
                # (workspace sized by the library: statistics + the fp32 partials of its query-split dK/dV launch)
                if _CROSS_ONES and hd == 72:
                    kp, vp = ops.kv_pad_ones(bs.ckv, B, Lc, H, hd, hdp, 0, D)
                    kvw, vvw, ones = kp[..., :hd], vp[..., :hd], 2
                else:
                    kvw, vvw, ones = ops.heads_view(bs.ckv, B, Lc, H, hd, 0), ops.heads_view(bs.ckv, B, Lc, H, hd, D), 0
                ops.attn_bwd(ops.heads_view(bs.qc, B, L, H, hd), kvw, vvw, ops.heads_view(bs.catt, B, L, H, hd), bs.lse2,
                             ops.heads_view(dcatt, B, L, H, hd), ops.heads_view(dqc, B, L, H, hd),
                             ops.heads_view(dckv, B, Lc, H, hd, 0), ops.heads_view(dckv, B, Lc, H, hd, D), None,
                             kv_pad_ones=ones)
                if bs.c8_on:
                    ops.absmax(dcatt, fp8_hist.cur(R0 + F8.ROW_DOC))
            if G.has(pre + "context_kv.bias"):
                ops.colsum(dckv, Gr("context_kv.bias"))
            if G.has(pre + "q_cross.bias"):
                ops.colsum(dqc, Gr("q_cross.bias"))
            if bs.f8c:
                q_dckv = F8.Q(dckv, F8.E5M2, F8.TN, not F8.TN, hist, R0 + F8.ROW_DCKV)  # (a weight-gradient operand only)
                wgrad8(q_dckv, q_ctx, Gr("context_kv.weight"))
                q_dqc = (F8.Q.from_rowmajor(*e_dqc, True) if e_dqc is not None else
                         F8.Q(dqc, F8.E5M2, True, True, hist, R0 + F8.ROW_DQC))
                wgrad8(q_dqc, bs.q_xn2, Gr("q_cross.weight"))
                dxn = F8.dgrad(q_dqc, bs.q_wqc)
                del q_dckv, q_dqc
            else:
                wgrad(dckv, sv.ctx2d, Gr("context_kv.weight"))
                wgrad(dqc, bs.xn2, Gr("q_cross.weight"))
                dxn = ops.linear_dgrad(dqc, W("q_cross.weight"))
            dX1 = ops.rmsnorm_mod_bwd(dxn, bs.X1, Wo("norm2.weight"), mod, 3 * D, 4 * D, bs.rstd2, dX2, dmod,
                                      Go("norm2.weight"), B, L)
        else:
            dX1 = dX2
        # --- self attention
        if bs.f8l:
            q_dy = gate_bwd_q(dX1, bs.y_sa, 2 * D, R0 + F8.ROW_DY_AP)
            wgrad8(q_dy, bs.q_attn, Gr("attn_proj.weight"))
            dattn = F8.dgrad(q_dy, bs.q_wap)
            del q_dy
        else:
            dy = ops.gate_bwd(dX1, bs.y_sa, mod, 2 * D, dmod, None, B, L)
            wgrad(dy, bs.attn, Gr("attn_proj.weight"))
            dattn = ops.linear_dgrad(dy, W("attn_proj.weight"))
        dq = torch.empty(B, H, L, hdp, dtype=bf16, device=dev)
        dk = torch.empty_like(dq)
        dv = torch.empty_like(dq)
        if bs.a8:
            r9 = F8.ROWS * i + F8.ROW_DO
            if getattr(sv, "doq", None) is None:  # one e5m2 dO buffer per backward pass; its pad bytes stay zero
                sv.doq = torch.zeros(B, H, L, ops.FP8_ROW, dtype=torch.float8_e5m2, device=dev)
            stats = ops.attn_fp8_delta(bs.attn, dattn, bs.lse1, sv.doq, fp8_hist.prev(r9), fp8_hist.cur(r9), bs.deq,
                                       B, H, L, hd)
            ops.attn_fp8_bwd(bs.q8, bs.k8, bs.v8, sv.doq, stats, bs.deq, dq[..., :hd], dk[..., :hd], dv[..., :hd], hd)
        else:
            delta = torch.empty(2, B, H, L, dtype=f32, device=dev)
            ops.attn_bwd(bs.q[..., :hd], bs.k[..., :hd], bs.v[..., :hd], ops.heads_view(bs.attn, B, L, H, hd), bs.lse1,
                         ops.heads_view(dattn, B, L, H, hd), dq[..., :hd], dk[..., :hd], dv[..., :hd], delta,
                         kv_pad_ones=(hdp - hd) >= 8)
            if bs.a8_on:
                ops.absmax(dattn, fp8_hist.cur(F8.ROWS * i + F8.ROW_DO))
        defer = getattr(sv, "dv_terms", None)  # DiT.backward: [(dv, lambda)] of the mixed blocks, summed at block 0
        mix_mode = (2 if defer is not None else 1) if bs.mix else 0
        if mix_mode == 2:
            defer.append((dv, W("lambda_param")))  # (keeps dv alive until the next reduction)
        # summed every _DV0_CHUNK blocks (and at block 0): at most that many dv tensors stay alive (DiT-XL, B = 12: 9 x 303 MB
        # instead of 27 x 303 MB = 8 GB), for one more read-modify-write of the fp32 accumulator per chunk
        if dv0 is not None and defer and (first or len(defer) >= _DV0_CHUNK):
            ops.dv0_reduce([t for t, _ in defer], [l for _, l in defer], dv0, B, H, L, hd, hdp,
                           accumulate=getattr(sv, "dv0_started", False))
            sv.dv0_started = True
            defer.clear()
        rope_args = (dq, dk, dv, sv.cos, sv.sin, bs.qkv if bs.mix else None, sv.v0 if bs.mix else None,
                     W("lambda_param") if bs.mix else None, dv0 if bs.mix else (dv0 if first else None),
                     Gr("lambda_param") if bs.mix else None, mix_mode, first and dv0 is not None, B, L, H, hd, hdp)
        # the qkv output gradient leaves the RoPE backward as e5m2 (a qkv bias gradient needs the bf16 tensor)
        emit_dqkv = pemit and not G.has(pre + "qkv.bias") and hdp % 8 == 0 and D <= 2048
        if emit_dqkv:
            fq, fs = ops.qkv_rope_bwd_fp8(*rope_args, F8.E5M2, hist.prev(F8.ROWS * i + 5), hist.part(F8.ROWS * i + 5, B * L))
            q_dqkv = F8.Q.from_rowmajor(fq, fs, True)
        else:
            dqkv = ops.qkv_rope_bwd(*rope_args)
        if G.has(pre + "qkv.bias"):
            ops.colsum(dqkv, Gr("qkv.bias"))
        if bs.f8:
            if not emit_dqkv:
                q_dqkv = F8.Q(dqkv, F8.E5M2, True, True, fp8_hist, F8.ROWS * i + 5)
            wgrad8(q_dqkv, bs.q_xn1, Gr("qkv.weight"))
            dxn = F8.dgrad(q_dqkv, bs.q_wqkv)
            del q_dqkv
        else:
            wgrad(dqkv, bs.xn1, Gr("qkv.weight"))
            dxn = ops.linear_dgrad(dqkv, W("qkv.weight"))
        dX0 = ops.rmsnorm_mod_bwd(dxn, bs.X, Wo("norm1.weight"), mod, 0, D, bs.rstd1, dX1, dmod, Go("norm1.weight"),
                                  B, L)
        # --- adaLN modulation (model.py:89-94,107)
        if not batched_adaln:
            ops.small_linear_bwd(dmod, sv.cvec, W("adaLN_modulation.1.weight"), Gr("adaLN_modulation.1.weight"),
                                 Gr("adaLN_modulation.1.bias"), dc, 1)
        return dX0


class _Saved:
    """activations kept for the backward pass"""
    pass


def _offsets(groups):
    off = 0
    for g in groups:
        yield off
        off += g.padded


def grad_arena(groups, device):
    """one fp32 allocation for the full-size gradient buffers of all flat groups (4.5 GB for DiT-XL), 1 KiB-aligned
    slices (a group's padded size is a multiple of 256 elements)"""
    return torch.zeros(sum(g.padded for g in groups), dtype=f32, device=device)


class DiT(nn.Module):
    def __init__(self, in_channels=4, patch_size=2, time_patch_size=2, hidden_size=1152, depth=28, num_heads=16,
                 mlp_ratio=4.0, cross_attn_input_size=128, residual_v=False, train_bias_and_rms=True,
                 use_rope=True):
        super().__init__()
        self.in_channels = self.out_channels = in_channels
        self.patch_size, self.time_patch_size = patch_size, time_patch_size
        self.hidden_size, self.num_heads, self.depth, self.mlp_ratio = hidden_size, num_heads, depth, mlp_ratio
        self.use_rope = use_rope
        self.residual_v = residual_v
        self.cross_attn_input_size = cross_attn_input_size
        self.head_dim = hidden_size // num_heads
        if self.head_dim not in HDP_OF:
            raise ValueError(f"head_dim {self.head_dim} = hidden_size / num_heads has no attention kernel instance "
                             "(multiples of 8 up to 128: DiT-S/B = 64, DiT-XL = 72, the reference's sweep = 128, its "
                             "smoke test = 32; the reference's own RoPE table needs a multiple of 8 too, model.py:192-209); "
                             "the kernels are templates on the padded head dim, see csrc/attention.hip")
        self.patch_embed = PatchEmbed(patch_size, in_channels, hidden_size, time_patch_size)
        if use_rope:
            self.rope = ThreeDimRotary(hidden_size // (2 * num_heads), h=128, w=128, t=128)
        else:
            # model.py:312-314: without RoPE the reference registers a learned table it never reads -- its forward
            # calls self.rope unconditionally (model.py:364) and raises AttributeError.  The parameter is
            # kept so that such a model's state dict (and train.py:287's constant class "positional_embedding") round
            # trips with strict loading; forward refuses like the reference does.
            self.positional_embedding = nn.Parameter(torch.zeros(1, 2048, hidden_size))
        self.register_tokens = nn.Parameter(torch.randn(1, N_REG, hidden_size))
        self.time_embed = nn.Sequential(_Linear(hidden_size, 4 * hidden_size), _Act("silu"),
                                        _Linear(4 * hidden_size, hidden_size))
        self.blocks = nn.ModuleList([
            DiTBlock(hidden_size=hidden_size, num_heads=num_heads, mlp_ratio=mlp_ratio,
                     cross_attn_input_size=cross_attn_input_size, residual_v=residual_v,
                     qkv_bias=train_bias_and_rms) for _ in range(depth)])
        self.final_modulation = nn.Sequential(_Act("silu"), _Linear(hidden_size, 2 * hidden_size, bias=True))
        self.final_norm = RMSNorm(hidden_size, trainable=train_bias_and_rms)
        self.final_proj = _Linear(hidden_size, patch_size * patch_size * time_patch_size * self.out_channels)
        nn.init.zeros_(self.final_modulation[-1].weight)
        nn.init.zeros_(self.final_modulation[-1].bias)
        nn.init.zeros_(self.final_proj.weight)
        nn.init.zeros_(self.final_proj.bias)
        self.paramstatus = {}
        for n, p in self.named_parameters():
            self.paramstatus[n] = {"shape": p.shape, "requires_grad": p.requires_grad}
        # flat groups (built lazily on the device): root + one per block, FSDP units of model.py:523-541
        self._groups: Optional[List[FlatGroup]] = None
        self._world, self._rank, self._pg = 1, 0, None
        self._fsdp = None  # set by fsdp.apply_fsdp
        self.fp8 = False   # enable_fp8(): qkv / mlp GEMMs on the fp8 MFMA path (fp8.py; BASELINE config 5)
        self.fp8_attn = 0  # ... and the attention products (1: self-attention, 2: cross-attention too)
        self.fp8_lin = False   # ... and the four remaining linears of a block

    def enable_fp8(self, on: bool = True, attention: bool = True, all_linears: bool = True,
                   cross_attention: bool = True):
        """Run the qkv and MLP linears of every block in OCP fp8 (e4m3 activations / weights, e5m2 gradients,
        per-tensor scaling) and, with `attention`, the attention products on the fp8 MFMA as well (head_dim 72:
        e4m3 Q / K / V / P, e5m2 dO / dS; fp8.py states the recipe).  The reference has no such mode.
        `cross_attention=False` keeps the cross-attention products (L x 512 context keys) on the bf16 kernels: in fp8
        they are 1.1 % of the step faster at the C3b shape and leave the 200-step loss curve where it was, at a
        cross-attention output 4.6 % instead of 0.3 % from the fp32 oracle."""
        self.fp8 = bool(on)
        self.fp8_attn = (2 if cross_attention else 1) if (on and attention) else 0  # 0 none, 1 self, 2 self + cross
        self.fp8_lin = bool(on and all_linears)  # attn_proj, q_cross, context_kv, cross_proj too (qkv / MLP always)
        self._fp8_hist = None  # fp8.AmaxHistory: 6 rows per block (gelu(fc1), d fc2-in, xn1, xn3, d mlp-out, d qkv)
        return self

    # --------------------------------------------------------------------- parameters ----
    def _group_members(self):
        root, blocks = [], [[] for _ in range(self.depth)]
        for n, p in self.named_parameters():
            if n.startswith("blocks."):
                i = int(n.split(".")[1])
                blocks[i].append((n, p))
            else:
                root.append((n, p))
        return root, blocks

    def _ensure_groups(self, device):
        if self._groups is not None and all(g.is_current() and g.device == device for g in self._groups):
            return
        full_values = None
        if self._groups is not None and self._world > 1:
            raise RuntimeError("parameters of a sharded DiT were replaced; re-apply apply_fsdp")
        root, blocks = self._group_members()
        groups = [FlatGroup("root", root, self._world, self._rank, self._pg)]
        groups += [FlatGroup(f"blocks.{i}", m, self._world, self._rank, self._pg) for i, m in enumerate(blocks)]
        self._grad_arena = grad_arena(groups, device)
        off = 0
        for g in groups:
            g.materialize(device, full_values, gfull=self._grad_arena[off:off + g.padded])
            off += g.padded
        self._groups = groups

    def _apply(self, fn, *a, **k):  # .to()/.cuda() replace tensors: flat views are rebuilt lazily
        r = super()._apply(fn, *a, **k)
        if self._world == 1:
            self._groups = None
        return r

    def _adaln_tables(self):
        """device pointer tables (weights, biases, weight gradients, bias gradients) of the blocks' adaLN linears for
        the batched launches; rebuilt when the flat buffers move"""
        key = tuple(g.full.data_ptr() for g in self._groups[1:]) + tuple(g.gfull.data_ptr() for g in self._groups[1:])
        if getattr(self, "_adaln_key", None) != key:
            W = [self.block_group(i).w(f"blocks.{i}.adaLN_modulation.1.weight") for i in range(self.depth)]
            b = [self.block_group(i).w(f"blocks.{i}.adaLN_modulation.1.bias") for i in range(self.depth)]
            gW = [self.block_group(i).g(f"blocks.{i}.adaLN_modulation.1.weight") for i in range(self.depth)]
            gb = [self.block_group(i).g(f"blocks.{i}.adaLN_modulation.1.bias") for i in range(self.depth)]
            self._adaln_tabs = tuple(ops.ptr_table(t) for t in (W, b, gW, gb))
            self._adaln_key = key
        return self._adaln_tabs

    @property
    def root_group(self) -> FlatGroup:
        return self._groups[0]

    def block_group(self, i) -> FlatGroup:
        return self._groups[1 + i]

    def zero_grad(self, set_to_none: bool = True):
        super().zero_grad(set_to_none)

    def _zero_grad_buffers(self):
        """clear the full-size fp32 gradient buffers the weight-gradient kernels accumulate into: ONE fill when the
        groups' buffers are slices of the model's arena (the normal case), else one per group"""
        arena = getattr(self, "_grad_arena", None)
        if arena is not None and all(g.gfull.data_ptr() == arena.data_ptr() + 4 * o for g, o in
                                     zip(self._groups, _offsets(self._groups))):
            arena.zero_()
        else:
            for g in self._groups:
                g.gfull.zero_()

    # ------------------------------------------------------------------------ forward ----
    def forward(self, x, context, timesteps, rope_start: Optional[Tuple[int, int, int]] = None):
        """x [B,C,T,H,W], context [B,Lc,Cc], timesteps [B] -> [B,C,T,H,W] bf16 (model.py:358-402)."""
        if not x.is_cuda:
            raise RuntimeError("video_diffusion_speedrun_amd.DiT runs on the GPU only (no CPU fallback)")
        if not self.use_rope:
            raise AttributeError("'DiT' object has no attribute 'rope': a DiT built with use_rope=False cannot run -- the "
                                 "reference's forward calls self.rope unconditionally (model.py:364) and fails the same "
                                 "way; the model only exists to hold / convert such a checkpoint")
        b, c, t, h, w = x.shape
        thw = (t // self.time_patch_size, h // self.patch_size, w // self.patch_size)
        if rope_start is None:
            rope_start = self.rope.draw_start(thw)
        if not isinstance(rope_start, torch.Tensor):  # a device int32[3] tensor is read by the kernel (graph.py)
            rope_start = tuple(rope_start)
        self._ensure_groups(x.device)
        params = [p for p in self.parameters() if p.requires_grad]
        need_grad = torch.is_grad_enabled() and len(params) > 0
        if need_grad:
            return _DiTFunction.apply(self, x, context, timesteps, rope_start, *params)
        out, _ = self._forward_impl(x, context, timesteps, rope_start, save=False)
        return out

    def _gather_all(self):
        """bf16 compute copies of every group.  world == 1: one cast kernel per group."""
        if self._fsdp is not None:
            return  # the sharding runtime gathers group by group, overlapped with compute
        for g in self._groups:
            g.gather(ops.cast_f32_bf16, None)

    def _forward_impl(self, x, context, timesteps, rope_start, save: bool):
        D, H, hd = self.hidden_size, self.num_heads, self.head_dim
        pt, p = self.time_patch_size, self.patch_size
        B, C, T, Hh, Ww = x.shape
        t, h, w = T // pt, Hh // p, Ww // p
        N = t * h * w
        L = N + N_REG
        dev = x.device
        fs = self._fsdp
        self._gather_all()
        if fs is not None:
            fs.pre_forward_root()
        R = self.root_group
        x = x.to(bf16).contiguous()
        context = context.to(bf16).contiguous()
        Lc, Cc = context.shape[1], context.shape[2]
        ctx2d = context.view(B * Lc, Cc)
        sv = _Saved() if save else None
        if self.fp8:
            if getattr(self, "_fp8_hist", None) is None or self._fp8_hist.tab.device != dev:
                self._fp8_hist = F8.AmaxHistory(F8.ROWS * self.depth + 1, dev)
            if save:
                self._fp8_hist.ensure_part(B * (t * h * w + N_REG))  # before roll(): see AmaxHistory.ensure_part
                self._fp8_hist.roll()
            # the text context is the same operand for every block's context_kv: quantised once per step
            q_ctx = None
            if self.fp8_lin and self.cross_attn_input_size is not None and F8.supported(B * Lc, 2 * D, Cc):
                q_ctx = F8.Q(ctx2d, F8.E4M3, True, save, self._fp8_hist if save else None, F8.ROWS * self.depth)
            if save:
                sv.q_ctx = q_ctx

        # patch embed + register tokens -> token buffer X [B*L, D]   (model.py:360-362).  ONE GEMM over all B*L rows:
        # the patches are laid out as rows of the token buffer (16 zero rows in front of every sample, whose outputs --
        # the bias -- are then overwritten by the register tokens)
        patches = ops.patchify(x, pt, p, lead_rows=N_REG)
        P = patches.shape[1]
        Wpe = R.w("patch_embed.patch_proj.weight").view(D, P)
        X = ops.linear_fwd(patches, Wpe, R.w("patch_embed.patch_proj.bias"))
        ops.fill_registers(R.w("register_tokens").view(N_REG, D), X, L * D, B, N_REG, D)
        cos, sin = self.rope.rows((t, h, w), rope_start, N_REG)

        # timestep conditioning (model.py:374-377), kept in fp32
        temb = ops.timestep_embedding(timesteps.to(f32).contiguous(), D)
        h1 = ops.small_linear_fwd(temb, R.w("time_embed.0.weight"), R.w("time_embed.0.bias"), 0)
        cvec = ops.small_linear_fwd(h1, R.w("time_embed.2.weight"), R.w("time_embed.2.bias"), 1)
        if save:
            sv.patches, sv.temb, sv.h1, sv.cvec, sv.cos, sv.sin = patches, temb, h1, cvec, cos, sin
            sv.dims = (B, C, T, Hh, Ww, t, h, w, N, L, Lc, Cc)
            sv.ctx2d = ctx2d
            sv.blocks = []

        # adaLN modulation of every block in ONE launch (K4 of SURVEY 2.3): all blocks read the same conditioning vector.
        # Not under sharding -- it would need every group's gathered weights before block 0 -- nor beyond 16 samples.
        mods = None
        if fs is None and B <= 16 and self.depth > 1 and os.environ.get("VDS_ADALN_BATCH", "1") != "0":
            wt, bt, _, _ = self._adaln_tables()
            mods = ops.small_linear_fwd_batched(cvec, wt, bt, self.depth, 9 * D, 1)
        v0 = None
        for i in range(self.depth):
            if fs is not None:
                fs.pre_forward_block(i)
            X, v, bs = self.blocks[i]._fwd(self.block_group(i), f"blocks.{i}.", X, ctx2d, cvec, v0, cos, sin, B, L, Lc,
                                           save, (self._fp8_hist, i, self.fp8_attn, self.fp8_lin, q_ctx) if self.fp8 else None,
                                           mods[i] if mods is not None else None)
            if v0 is None:
                v0 = v
            if save:
                sv.blocks.append(bs)
            if fs is not None:
                fs.post_forward_block(i)
        if save:
            sv.v0 = v0
            sv.x_last = X
            sv.batched_adaln = mods is not None

        # final layer (model.py:386-401)
        fmod = ops.small_linear_fwd(cvec, R.w("final_modulation.1.weight"), R.w("final_modulation.1.bias"), 1)
        wfn = R.w("final_norm.weight") if R.has("final_norm.weight") else None
        # norm, modulation and projection run over all B*L rows (the 16 register rows of a sample are 0.2 % of them and
        # are simply not read back: the reference slices them off first, model.py:386); unpatchify picks the token rows
        xnf, rstdf = ops.rmsnorm_mod_fwd(X, wfn, fmod, 0, D, B, L)
        yf = ops.linear_fwd(xnf, R.w("final_proj.weight"), R.w("final_proj.bias"))
        out = ops.unpatchify(yf, B, C, T, Hh, Ww, pt, p, lead_rows=N_REG)
        if save:
            sv.fmod, sv.xnf, sv.rstdf = fmod, xnf, rstdf
        if fs is not None:
            fs.post_forward_root()
        return out, sv

    # ----------------------------------------------------------------------- backward ----
    def _backward_impl(self, sv: _Saved, dout: torch.Tensor):
        D, H, hd = self.hidden_size, self.num_heads, self.head_dim
        pt, p = self.time_patch_size, self.patch_size
        B, C, T, Hh, Ww, t, h, w, N, L, Lc, Cc = sv.dims
        dev = dout.device
        fs = self._fsdp
        R = self.root_group
        # gradients already held in p.grad (a second backward before zero_grad: micro-batch accumulation, like
        # autograd's accumulate-into-.grad in the reference loop) are set aside and added back at the end
        held = [g.hold_grads() for g in self._groups]  # per parameter: only slices whose .grad is live are kept
        self._zero_grad_buffers()
        if fs is not None:
            fs.pre_backward_root()
        Pd = p * p * pt * C
        dout = dout.to(bf16).contiguous()
        # final layer
        dyf = ops.unpatchify_bwd(dout, pt, p, lead_rows=N_REG)  # [B*L, P], register rows zero
        ops.linear_wgrad(dyf, sv.xnf, R.g("final_proj.weight"))
        ops.colsum(dyf, R.g("final_proj.bias"))
        dxnf = ops.linear_dgrad(dyf, R.w("final_proj.weight"))
        # the small zero-initialised accumulators of this pass come out of ONE fill: d(final modulation) [B, 2D],
        # d(conditioning vector) [B, D], d(time-embed hidden) [B, 4D] and the adaLN modulation gradients of all blocks
        nmods = self.depth * B * 9 * D if sv.batched_adaln else 0
        zs = torch.zeros(B * 7 * D + nmods, dtype=f32, device=dev)
        dfmod, dc, dh1 = zs[:B * 2 * D].view(B, 2 * D), zs[B * 2 * D:B * 3 * D].view(B, D), zs[B * 3 * D:B * 7 * D].view(B, 4 * D)
        dmods = zs[B * 7 * D:].view(self.depth, B, 9 * D) if sv.batched_adaln else None
        wfn = R.w("final_norm.weight") if R.has("final_norm.weight") else None
        dwfn = R.g("final_norm.weight") if wfn is not None else None
        # register rows: dxnf = 0 there, so they get a zero gradient from the final layer (they were sliced off)
        dX = ops.rmsnorm_mod_bwd(dxnf, sv.x_last, wfn, sv.fmod, 0, D, sv.rstdf, None, dfmod, dwfn, B, L)
        ops.small_linear_bwd(dfmod, sv.cvec, R.w("final_modulation.1.weight"), R.g("final_modulation.1.weight"),
                             R.g("final_modulation.1.bias"), dc, 1)
        hdp = HDP_OF[hd]
        dv0 = torch.zeros(B, H, L, hdp, dtype=f32, device=dev) if (self.residual_v and self.depth > 1) else None
        sv.dv_terms = [] if (_DV0_DEFER and dv0 is not None) else None
        for i in reversed(range(self.depth)):
            if fs is not None:
                fs.pre_backward_block(i)
            dX = self.blocks[i]._bwd(self.block_group(i), f"blocks.{i}.", sv.blocks[i], dX, sv, dc, dv0, B, L, Lc,
                                     i == 0, (self._fp8_hist, i, self.fp8_attn, self.fp8_lin, sv.q_ctx) if self.fp8 else None,
                                     dmods[i] if dmods is not None else None)
            sv.blocks[i] = None
            if fs is not None:
                fs.post_backward_block(i)
        if dmods is not None:  # adaLN weight / bias gradients of all blocks and their fan-in to dc: one launch each
            wt, _, gwt, gbt = self._adaln_tables()
            ops.small_linear_bwd_batched(dmods, sv.cvec, wt, gwt, gbt, dc, 1)
        # registers + patch embed (model.py:360-362)
        ops.registers_bwd(dX, L * D, R.g("register_tokens").view(N_REG, D), B, N_REG, D)
        # patch embedding: weight gradient over all B*L rows (the register rows of `patches` are zero), bias gradient
        # over the token rows only
        ops.linear_wgrad(dX, sv.patches, R.g("patch_embed.patch_proj.weight").view(D, Pd))
        ops.colsum(dX, R.g("patch_embed.patch_proj.bias"), rows_per_sample=L, row_offset=N_REG)
        # time embed MLP
        ops.small_linear_bwd(dc, sv.h1, R.w("time_embed.2.weight"), R.g("time_embed.2.weight"),
                             R.g("time_embed.2.bias"), dh1, 1)
        ops.small_linear_bwd(dh1, sv.temb, None, R.g("time_embed.0.weight"), R.g("time_embed.0.bias"), None, 0)
        if fs is not None:
            fs.post_backward_root()
        else:
            for g in self._groups:
                g.publish_grads()
        for g, h in zip(self._groups, held):
            g.add_held(h)
        if self.fp8 and getattr(self, "_fp8_hist", None) is not None:
            self._fp8_hist.backward_done()

    # ------------------------------------------------------------------- muP table ----
    def get_mup_setup(self, learning_rate, weight_decay, constant_param_classes):
        """muP-style per-parameter (lr, weight decay) table; same contract as the reference
        (model.py:404-465): returns (optimizer param groups -- one per distinct (lr, wd) pair, in first-use
        order --, {name: {"lr", "wd", "shape"}}).  The cascade, later stages overriding earlier ones:
          1. vectors / scalars (name has bias | norm | lambda): lr/100, no decay; matrices: lr * 32/fan_in and
             wd * fan_in/1024, fan_in = last dim of the FULL tensor shape;
          2. names containing one of `constant_param_classes`: lr/100, no decay;
          3. names containing "time", then "modulation": lr/10 (decay untouched)."""
        table, buckets = {}, {}
        for name, p in self.named_parameters():
            name = name.replace("_fsdp_wrapped_module.", "")  # the prefix the reference strips (model.py:417)
            info = self.paramstatus[name]
            if not info["requires_grad"]:
                continue
            fan_in = info["shape"][-1]
            if "bias" in name or "norm" in name or "lambda" in name:
                lr, wd = learning_rate * 0.01, 0.0
            else:
                lr, wd = learning_rate * (32 / fan_in), weight_decay * fan_in / 1024
            if any(c in name for c in constant_param_classes):
                lr, wd = learning_rate * 0.01, 0.0
            for tag in ("time", "modulation"):
                if tag in name:
                    lr = learning_rate * 0.1
            bucket = buckets.setdefault((lr, wd), {"params": [], "weight_decay": wd, "lr": lr})
            bucket["params"].append(p)
            table[name] = {"lr": lr, "wd": wd, "shape": info["shape"]}
        return list(buckets.values()), table

    # --------------------------------------------------------------- state / copies ----
    def invalidate_compute_copy(self):
        """Force the next forward to re-cast the fp32 masters into the bf16 compute copies.  In-place writes
        through the nn.Parameters are detected by themselves (FlatGroup.mark_shadow_fresh); call this after
        writing parameter memory some other way (`p.data.xxx_()`, raw pointers, custom kernels)."""
        for g in self._groups or []:
            g.invalidate_shadow()
        # fp8: the weights' delayed-scaling history describes the OLD values (a larger checkpoint would be clipped at the
        # old amax for a step): re-measure everything on the next step
        if getattr(self, "_fp8_hist", None) is not None:
            self._fp8_hist.reset()

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        r = super().load_state_dict(state_dict, strict=strict, assign=assign)
        self.invalidate_compute_copy()
        return r

    def full_state_dict(self) -> Dict[str, torch.Tensor]:
        """fp32 full tensors keyed like the reference state dict (one all-gather per group when sharded)."""
        if self._groups is None:
            return {k: v.detach().clone() for k, v in self.state_dict().items()}
        out = {}
        for g in self._groups:
            out.update(g.full_tensors())
        return out


class _BlockFunction(torch.autograd.Function):
    """DiTBlock.forward as one autograd node (see DiTBlock.forward for the contract)."""

    @staticmethod
    def forward(ctx, block, need, x, context, c, v_0, cos, sin, *params):
        B, L, D = x.shape
        H, hd = block.num_heads, block.head_dim
        hdp = HDP_OF[hd]
        dev = x.device
        if (context is None) != (block.q_cross is None):
            raise ValueError("DiTBlock: `context` must be given exactly when the block has cross-attention")
        G, pre = block._group_and_prefix(dev)
        G.gather(ops.cast_f32_bf16, None)
        X = x.reshape(B * L, D).to(bf16).contiguous()
        if context is not None:
            Lc = context.shape[1]
            ctx2d = context.to(bf16).contiguous().view(B * Lc, context.shape[2])
        else:
            Lc, ctx2d = 0, None
        cvec = c.to(f32).contiguous()
        if cos is None:  # rope=None: no rotation (model.py:132-134) = the identity rotation
            cos2 = torch.ones(L, hd // 2, dtype=f32, device=dev)
            sin2 = torch.zeros(L, hd // 2, dtype=f32, device=dev)
        else:
            cos2 = cos.reshape(L, hd // 2).to(f32).contiguous()
            sin2 = sin.reshape(L, hd // 2).to(f32).contiguous()
        v0p = None
        if v_0 is not None and block.residual_v:
            if tuple(v_0.shape) != (B, H, L, hd):
                raise ValueError(f"DiTBlock: v_0 must be [B,H,L,hd] = {(B, H, L, hd)}, got {tuple(v_0.shape)}")
            if v_0.dtype == bf16 and v_0.stride() == (H * L * hdp, L * hdp, hdp, 1):
                v0p = v_0.as_strided((B, H, L, hdp), (H * L * hdp, L * hdp, hdp, 1))  # a `v` this class returned
            else:
                v0p = torch.zeros(B, H, L, hdp, dtype=bf16, device=dev)
                v0p[..., :hd].copy_(v_0)
        Xo, v, bs = block._fwd(G, pre, X, ctx2d, cvec, v0p, cos2, sin2, B, L, Lc, need)
        if need:
            sv = _Saved()
            sv.cos, sv.sin, sv.v0, sv.ctx2d, sv.cvec = cos2, sin2, v0p, ctx2d, cvec
            ctx.state = (block, G, pre, bs, sv, (B, L, Lc, D, H, hd, hdp), x.dtype, c.dtype,
                         v_0.dtype if v_0 is not None else None)
        ctx.set_materialize_grads(False)
        return Xo.view(B, L, D), v[..., :hd]

    @staticmethod
    def backward(ctx, dx, dv_out):
        block, G, pre, bs, sv, (B, L, Lc, D, H, hd, hdp), xdt, cdt, v0dt = ctx.state
        ctx.state = None
        dev = sv.cvec.device
        held = G.hold_grads()  # accumulate like autograd (see DiT._backward_impl)
        G.gfull.zero_()
        dX = (dx.reshape(B * L, D).to(bf16).contiguous() if dx is not None
              else torch.zeros(B * L, D, dtype=bf16, device=dev))
        dc = torch.zeros(B, D, dtype=f32, device=dev)
        dv0 = None
        if bs.mix:
            if dv_out is not None:
                raise NotImplementedError("DiTBlock: a gradient through the returned (lambda-mixed) v of a block "
                                          "that was given v_0 is not implemented; the model only consumes block 0's v")
            dv0 = torch.zeros(B, H, L, hdp, dtype=f32, device=dev)
        elif dv_out is not None:
            dv0 = torch.zeros(B, H, L, hdp, dtype=f32, device=dev)
            dv0[..., :hd].copy_(dv_out)
        dX0 = block._bwd(G, pre, bs, dX, sv, dc, dv0, B, L, Lc, first=(not bs.mix and dv0 is not None))
        G.publish_grads()
        G.add_held(held)
        gv0 = dv0[..., :hd].to(v0dt) if (bs.mix and v0dt is not None) else None
        n_params = len(list(block.parameters()))
        return (None, None, dX0.view(B, L, D).to(xdt), None, dc.to(cdt), gv0, None, None) + (None,) * n_params


class _DiTFunction(torch.autograd.Function):
    """One autograd node for the whole model: forward/backward are explicit kernel sequences;
    parameter gradients are written into the flat fp32 gradient buffers that p.grad aliases."""

    @staticmethod
    def forward(ctx, model, x, context, timesteps, rope_start, *params):
        out, sv = model._forward_impl(x, context, timesteps, rope_start, save=True)
        ctx.model, ctx.sv = model, sv
        return out

    @staticmethod
    def backward(ctx, dout):
        model, sv = ctx.model, ctx.sv
        ctx.sv = None
        model._backward_impl(sv, dout)
        return (None,) * (5 + len([p for p in model.parameters() if p.requires_grad]))
