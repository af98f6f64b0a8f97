"""CPU oracle for the video-DiT train-step hot path.  *** TEST INFRASTRUCTURE ONLY ***

This file is a from-scratch CPU restatement (plain torch ops on CPU tensors, explicit
formulas, no nn.Module) of the algorithm in the reference repository
`fal-ai-community/video-diffusion-speedrun` (model.py / train.py).  It is the parity
checker for the hand-written HIP path in `video_diffusion_speedrun_amd/`:

  * only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may
    import it; the product package never does (it fails loudly without its HIP library);
  * it never reads /root/reference (which does not exist on the GPU box).  Its parity with
    the reference is pinned by `tests/golden/*.pt`, produced by `oracle/make_golden.py`
    (which DOES import the reference, in the build container only) and checked by
    `tests/test_oracle_golden.py`.

Every function cites the reference file:line it restates.  Parameters are passed as a flat
dict keyed exactly like the reference `DiT.state_dict()`, so reference
checkpoints, the golden fixtures and the HIP model all exchange weights without renaming.

Precision: every op runs in the dtype of its inputs (fp32 oracle = "truth"; bf16 oracle =
the reference's rounding points under its bf16 FSDP policy), except where the reference
itself forces fp32 (RMSNorm statistics, RoPE, sinusoid, loss).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import torch

Tensor = torch.Tensor
N_REG = 16  # register tokens, model.py:316,362,386


@dataclass
class DiTConfig:
    """Constructor arguments of the reference `DiT` (model.py:279-292)."""

    in_channels: int = 4
    patch_size: int = 2
    time_patch_size: int = 2
    hidden_size: int = 1152
    depth: int = 28
    num_heads: int = 16
    mlp_ratio: float = 4.0
    cross_attn_input_size: Optional[int] = 128
    residual_v: bool = False
    train_bias_and_rms: bool = True

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_heads

    @property
    def patch_dim(self) -> int:
        return self.patch_size * self.patch_size * self.time_patch_size * self.in_channels


# --------------------------------------------------------------------------------------
# elementary ops
# --------------------------------------------------------------------------------------
def timestep_embedding(t: Tensor, dim: int, max_period: float = 10000.0) -> Tensor:
    """[cos(t f_i) | sin(t f_i)], f_i = exp(-ln(max_period) i / half); t is NOT scaled by
    1000 (model.py:12-22).  fp32 result."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32) / half)
    args = t.reshape(-1, 1).float() * freqs.reshape(1, -1)
    return torch.cat([args.cos(), args.sin()], dim=-1)


def silu(x: Tensor) -> Tensor:
    """x * sigmoid(x) (nn.SiLU, model.py:90,320,340)."""
    return x * torch.sigmoid(x)


def gelu_erf(x: Tensor) -> Tensor:
    """Exact erf GELU, nn.GELU() default (model.py:85)."""
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def linear(x: Tensor, w: Tensor, b: Optional[Tensor] = None) -> Tensor:
    """y = x W^T + b, W stored [out, in] (nn.Linear)."""
    y = x @ w.t()
    if b is not None:
        y = y + b
    return y


def rms_norm(x: Tensor, weight: Optional[Tensor] = None, eps: float = 1e-6) -> Tensor:
    """x.float() * rsqrt(mean(x^2) + eps) [* weight] cast back to x.dtype (model.py:34-41)."""
    xf = x.float()
    inv = torch.rsqrt((xf * xf).mean(dim=-1, keepdim=True) + eps)
    y = xf * inv
    if weight is not None:
        y = y * weight
    return y.to(x.dtype)


def modulate(xn: Tensor, shift: Tensor, scale: Tensor) -> Tensor:
    """norm_x * (1 + scale) + shift with [B,D] modulation broadcast over tokens
    (model.py:123,144,164,389)."""
    return xn * (1 + scale[:, None, :]) + shift[:, None, :]


def patchify(x: Tensor, pt: int, p: int) -> Tensor:
    """[B,C,T,H,W] -> [B, N, C*pt*p*p]; token order (h w t) with t fastest (model.py:185),
    feature order (c, dt, dh, dw) = Conv3d weight layout (model.py:173-178)."""
    B, C, T, H, W = x.shape
    t, h, w = T // pt, H // p, W // p
    x = x[:, :, : t * pt, : h * p, : w * p]  # Conv3d floors odd extents
    x = x.reshape(B, C, t, pt, h, p, w, p)
    x = x.permute(0, 4, 6, 2, 1, 3, 5, 7)  # b h w t c dt dh dw
    return x.reshape(B, h * w * t, C * pt * p * p)


def patch_embed(x: Tensor, w: Tensor, b: Tensor, pt: int, p: int) -> Tensor:
    """Conv3d(kernel=stride=(pt,p,p)) + rearrange == patchify then GEMM (model.py:182-186)."""
    return linear(patchify(x, pt, p), w.reshape(w.shape[0], -1), b)


def unpatchify(y: Tensor, C: int, t: int, h: int, w: int, pt: int, p: int) -> Tensor:
    """'b (h w t) (p1 p2 p3 c) -> b c (t p3) (h p1) (w p2)' (model.py:392-401):
    feature order (dh, dw, dt, c) with c fastest."""
    B = y.shape[0]
    y = y.reshape(B, h, w, t, p, p, pt, C)  # b h w t p1 p2 p3 c
    y = y.permute(0, 7, 3, 6, 1, 4, 2, 5)  # b c t p3 h p1 w p2
    return y.reshape(B, C, t * pt, h * p, w * p)


# --------------------------------------------------------------------------------------
# 3-D RoPE (model.py:189-275)
# --------------------------------------------------------------------------------------
def rope_freq_tables(rot_dim: int, base: float = 100.0) -> Tuple[Tensor, Tensor]:
    """inv_freq_time (rot_dim/2 entries, step 2) and inv_freq_space (rot_dim/4, step 4)
    for `ThreeDimRotary(dim=rot_dim)`, rot_dim = hidden/(2*heads) = head_dim/2
    (model.py:192-193,310-312)."""
    inv_t = 1.0 / (base ** (torch.arange(0, rot_dim, 2).float() / rot_dim))
    inv_s = 1.0 / (base ** (torch.arange(0, rot_dim, 4).float() / rot_dim))
    return inv_t, inv_s


def rope_cos_sin(head_dim: int, thw: Tuple[int, int, int], start_thw: Tuple[int, int, int],
                 n_register: int = N_REG) -> Tuple[Tensor, Tensor]:
    """cos/sin rows [n_register + t*h*w, head_dim/2] fp32.

    Row i (after the register rows, which are cos=1/sin=0, model.py:243-261) is the table
    entry of position (ti,hi,wi) = unravel(i,(t,h,w)) + start, i.e. the [t,h,w,:] slice is
    flattened ROW-MAJOR (t h w) (model.py:239-240) although tokens are ordered (h w t):
    SURVEY Q1, reproduced on purpose.  Feature layout [t: d/2 | h: d/4 | w: d/4] of
    d = head_dim/2 (model.py:214).  The angle is position*inv_freq in fp32 exactly as
    torch.outer(arange, inv_freq) builds it (model.py:198-210)."""
    t, h, w = thw
    st, sh, sw = start_thw
    rot = head_dim // 2
    inv_t, inv_s = rope_freq_tables(rot)
    pt = torch.arange(st, st + t, dtype=torch.float32)
    ph = torch.arange(sh, sh + h, dtype=torch.float32)
    pw = torch.arange(sw, sw + w, dtype=torch.float32)
    ft = torch.outer(pt, inv_t).reshape(t, 1, 1, -1).expand(t, h, w, -1)
    fh = torch.outer(ph, inv_s).reshape(1, h, 1, -1).expand(t, h, w, -1)
    fw = torch.outer(pw, inv_s).reshape(1, 1, w, -1).expand(t, h, w, -1)
    ang = torch.cat([ft, fh, fw], dim=3).reshape(t * h * w, -1)
    cos, sin = ang.cos(), ang.sin()
    if n_register > 0:
        cos = torch.cat([torch.ones(n_register, cos.shape[1]), cos], 0)
        sin = torch.cat([torch.zeros(n_register, sin.shape[1]), sin], 0)
    return cos, sin


def draw_rope_offsets(thw: Tuple[int, int, int], table: int = 128) -> Tuple[int, int, int]:
    """The reference draws (start_h, start_w, start_t) IN THAT ORDER from the global CPU
    RNG on every forward (model.py:223-226).  Returns (start_t,start_h,start_w)."""
    t, h, w = thw
    sh = int(torch.randint(0, table - h + 1, (1,)).item())
    sw = int(torch.randint(0, table - w + 1, (1,)).item())
    st = int(torch.randint(0, table - t + 1, (1,)).item())
    return st, sh, sw


def apply_rotary(x: Tensor, cos: Tensor, sin: Tensor) -> Tensor:
    """Half-split rotation in fp32: y1 = x1 c + x2 s, y2 = -x1 s + x2 c (model.py:266-275).
    x [B,H,L,hd]; cos/sin [L,hd/2]."""
    xf = x.float()
    d = xf.shape[-1] // 2
    x1, x2 = xf[..., :d], xf[..., d:]
    y1 = x1 * cos + x2 * sin
    y2 = x2 * cos - x1 * sin
    return torch.cat([y1, y2], dim=-1).to(x.dtype)


def attention(q: Tensor, k: Tensor, v: Tensor) -> Tensor:
    """softmax(q k^T / sqrt(hd)) v, full, no mask (F.scaled_dot_product_attention,
    model.py:136,157).  Scores/softmax in fp32 (what flash kernels do), output in q.dtype."""
    if q.shape[:-2].numel() * q.shape[-2] * k.shape[-2] > CHUNKED_ATTENTION_ABOVE:
        return _ChunkedAttention.apply(q, k, v)  # same math, scores never materialised for all queries at once
    scale = 1.0 / math.sqrt(q.shape[-1])
    s = (q.float() @ k.float().transpose(-1, -2)) * scale
    p = torch.softmax(s, dim=-1)
    return (p @ v.float()).to(q.dtype)


CHUNKED_ATTENTION_ABOVE = 1 << 29  # score elements (2 GiB of fp32): BASELINE config 4 (33 808 tokens) needs 73 GB otherwise


def attention_chunked(q: Tensor, k: Tensor, v: Tensor, do: Optional[Tensor] = None, chunk: int = 1024):
    """The same softmax(q k^T / sqrt(hd)) v (model.py:136,157) evaluated one block of `chunk` query rows at
    a time in fp32, with the closed-form backward of SDPA (dV = P^T dO, dP = dO V^T,
    dS = P o (dP - rowsum(dO o O)), dQ = dS K / sqrt(hd), dK = dS^T Q / sqrt(hd)).
    Returns (o, lse) or (o, lse, dq, dk, dv), all fp32.  Pinned against `attention` + autograd by
    tests/test_oracle_golden.py."""
    qf, kf, vf = q.float(), k.float(), v.float()
    scale = 1.0 / math.sqrt(q.shape[-1])
    Lq = q.shape[-2]
    o = torch.empty_like(qf)
    lse = torch.empty(q.shape[:-1], dtype=torch.float32)
    if do is not None:
        dof = do.float()
        dq, dk, dv = torch.empty_like(qf), torch.zeros_like(kf), torch.zeros_like(vf)
    for r0 in range(0, Lq, chunk):
        r1 = min(Lq, r0 + chunk)
        s = (qf[..., r0:r1, :] @ kf.transpose(-1, -2)) * scale
        l = torch.logsumexp(s, dim=-1)
        p = torch.exp(s - l[..., None])
        oc = p @ vf
        o[..., r0:r1, :], lse[..., r0:r1] = oc, l
        if do is not None:
            doc = dof[..., r0:r1, :]
            dv += p.transpose(-1, -2) @ doc
            ds = p * (doc @ vf.transpose(-1, -2) - (doc * oc).sum(-1, keepdim=True))
            dq[..., r0:r1, :] = (ds @ kf) * scale
            dk += (ds.transpose(-1, -2) @ qf[..., r0:r1, :]) * scale
    return (o, lse) if do is None else (o, lse, dq, dk, dv)


class _ChunkedAttention(torch.autograd.Function):
    """`attention` for sequences whose score matrix does not fit in host memory (autograd wrapper of
    attention_chunked; the backward recomputes the probabilities chunk by chunk)."""

    @staticmethod
    def forward(ctx, q, k, v):
        o, _ = attention_chunked(q, k, v)
        ctx.save_for_backward(q, k, v)
        return o.to(q.dtype)

    @staticmethod
    def backward(ctx, do):
        q, k, v = ctx.saved_tensors
        _, _, dq, dk, dv = attention_chunked(q, k, v, do)
        return dq.to(q.dtype), dk.to(k.dtype), dv.to(v.dtype)


def split_heads(x: Tensor, n: int, H: int) -> Tuple[Tensor, ...]:
    """'b l (k h d) -> k b h l d' (model.py:126,149-154)."""
    B, L, _ = x.shape
    x = x.reshape(B, L, n, H, -1).permute(2, 0, 3, 1, 4)
    return tuple(x[i] for i in range(n))


def merge_heads(x: Tensor) -> Tensor:
    """'b h l d -> b l (h d)' (model.py:137,158)."""
    B, H, L, d = x.shape
    return x.permute(0, 2, 1, 3).reshape(B, L, H * d)


# --------------------------------------------------------------------------------------
# DiT block / model
# --------------------------------------------------------------------------------------
def block_forward(P: Dict[str, Tensor], pre: str, cfg: DiTConfig, x: Tensor,
                  context: Optional[Tensor], c: Tensor, v_0: Optional[Tensor],
                  cos: Tensor, sin: Tensor, cap: Optional[dict] = None):
    """DiTBlock.forward (model.py:96-167).  Returns (x, v)."""
    H = cfg.num_heads
    g = lambda n: P.get(pre + n)
    mod = linear(silu(c), g("adaLN_modulation.1.weight"), g("adaLN_modulation.1.bias"))
    (shift_sa, scale_sa, gate_sa, shift_ca, scale_ca, gate_ca,
     shift_mlp, scale_mlp, gate_mlp) = mod.chunk(9, dim=1)

    # self attention (model.py:122-139)
    xn = modulate(rms_norm(x, g("norm1.weight")), shift_sa, scale_sa)
    q, k, v = split_heads(linear(xn, g("qkv.weight"), g("qkv.bias")), 3, H)
    if cfg.residual_v and v_0 is not None:
        lam = g("lambda_param")
        v = lam * v + (1 - lam) * v_0
    qr = apply_rotary(q, cos, sin)
    kr = apply_rotary(k, cos, sin)
    att = merge_heads(attention(qr, kr, v))
    y_sa = linear(att, g("attn_proj.weight"))
    x1 = x + y_sa * gate_sa[:, None, :]
    if cap is not None:
        cap.update({pre + "mod": mod, pre + "xn1": xn, pre + "q_rope": qr, pre + "k_rope": kr,
                    pre + "v": v, pre + "attn": att, pre + "y_sa": y_sa, pre + "x_sa": x1})

    # cross attention (model.py:142-160)
    x2 = x1
    if g("q_cross.weight") is not None:
        xn2 = modulate(rms_norm(x1, g("norm2.weight")), shift_ca, scale_ca)
        (qc,) = split_heads(linear(xn2, g("q_cross.weight"), g("q_cross.bias")), 1, H)
        kc, vc = split_heads(linear(context, g("context_kv.weight"), g("context_kv.bias")), 2, H)
        catt = merge_heads(attention(qc, kc, vc))
        y_ca = linear(catt, g("cross_proj.weight"))
        x2 = x1 + y_ca * gate_ca[:, None, :]
        if cap is not None:
            cap.update({pre + "cattn": catt, pre + "y_ca": y_ca, pre + "x_ca": x2})

    # MLP (model.py:163-165)
    xn3 = modulate(rms_norm(x2, g("norm3.weight")), shift_mlp, scale_mlp)
    hmid = gelu_erf(linear(xn3, g("mlp.0.weight"), g("mlp.0.bias")))
    y_mlp = linear(hmid, g("mlp.2.weight"), g("mlp.2.bias"))
    x3 = x2 + y_mlp * gate_mlp[:, None, :]
    if cap is not None:
        cap.update({pre + "y_mlp": y_mlp, pre + "x_out": x3})
    return x3, v


def dit_forward(P: Dict[str, Tensor], cfg: DiTConfig, x: Tensor, context: Tensor,
                timesteps: Tensor, rope_start: Tuple[int, int, int],
                cap: Optional[dict] = None) -> Tensor:
    """DiT.forward (model.py:358-402) with the RoPE offsets (start_t,start_h,start_w) given
    explicitly (the reference draws them, see `draw_rope_offsets`)."""
    B, C, T, Hh, Ww = x.shape
    pt, p = cfg.time_patch_size, cfg.patch_size
    t, h, w = T // pt, Hh // p, Ww // p
    tok = patch_embed(x, P["patch_embed.patch_proj.weight"], P["patch_embed.patch_proj.bias"], pt, p)
    tok = torch.cat([P["register_tokens"].expand(B, -1, -1).to(tok.dtype), tok], dim=1)
    cos, sin = rope_cos_sin(cfg.head_dim, (t, h, w), rope_start)
    temb = timestep_embedding(timesteps, cfg.hidden_size).to(tok.dtype)
    c = linear(silu(linear(temb, P["time_embed.0.weight"], P["time_embed.0.bias"])),
               P["time_embed.2.weight"], P["time_embed.2.bias"])
    if cap is not None:
        cap.update({"tokens": tok, "t_emb": c, "rope_cos": cos, "rope_sin": sin})
    v_0 = None
    for i in range(cfg.depth):
        tok, v = block_forward(P, f"blocks.{i}.", cfg, tok, context, c, v_0, cos, sin, cap)
        if v_0 is None:
            v_0 = v
    tok = tok[:, N_REG:, :]
    fmod = linear(silu(c), P["final_modulation.1.weight"], P["final_modulation.1.bias"])
    fshift, fscale = fmod.chunk(2, dim=1)
    tok = modulate(rms_norm(tok, P.get("final_norm.weight")), fshift, fscale)
    y = linear(tok, P["final_proj.weight"], P["final_proj.bias"])
    if cap is not None:
        cap.update({"final_tokens": y})
    return unpatchify(y, C, t, h, w, pt, p)


# --------------------------------------------------------------------------------------
# train-step harness (train.py:51-145)
# --------------------------------------------------------------------------------------
def time_shift(z: Tensor, alpha: float = 8.0) -> Tensor:
    """t = sigmoid(z); t <- alpha t / (1 + (alpha-1) t) (train.py:93-96), in z.dtype."""
    t = torch.sigmoid(z)
    return t * alpha / (1 + (alpha - 1) * t)


def noise_latents(x: Tensor, noise: Tensor, t: Tensor) -> Tuple[Tensor, Tensor]:
    """z_t = x (1-t) + noise t ; v = x - noise (train.py:115-117), dtype of x (bf16)."""
    tr = t.reshape(-1, 1, 1, 1, 1)
    return x * (1 - tr) + noise * tr, x - noise


def flow_loss(v: Tensor, out: Tensor) -> Tuple[Tensor, Tensor]:
    """per-sample mean((v.float()-out.float())^2) over (C,T,H,W), then batch mean
    (train.py:121-125).  Returns (loss, per_sample)."""
    per = (v.float() - out.float()).pow(2).mean(dim=(1, 2, 3, 4))
    return per.mean(), per


def train_forward(P, cfg: DiTConfig, latent: Tensor, context: Tensor, z: Tensor, noise: Tensor,
                  rope_start, compute_dtype=torch.bfloat16, cap: Optional[dict] = None):
    """train.py::forward with the random draws (z, noise) and the encoded caption passed in.
    latent/context/z/noise are cast to `compute_dtype` as the reference does
    (train.py:73,84,90-92,103-105)."""
    x = latent.to(compute_dtype)
    ctx = context.to(compute_dtype)
    t = time_shift(z.to(compute_dtype))
    z_t, v = noise_latents(x, noise.to(compute_dtype), t)
    out = dit_forward(P, cfg, z_t, ctx, t, rope_start, cap)
    loss, per = flow_loss(v, out)
    if cap is not None:
        cap.update({"t": t, "z_t": z_t, "v_objective": v, "output": out, "loss_per_sample": per})
    return loss


# --------------------------------------------------------------------------------------
# Euler + classifier-free-guidance sampler (sampling/sample.py:77-159)
# --------------------------------------------------------------------------------------
def sample_euler_cfg(P, cfg: DiTConfig, latents: Tensor, prompt_embeds: Tensor, negative_embeds: Tensor,
                     inference_steps: int, cfg_scale: float, rope_starts, dtype=torch.bfloat16,
                     alpha: float = 8.0) -> Tensor:
    """The sampling loop of generate_image: for i = steps..1 the shifted times t, t_next
    (sample.py:126-136), model output for the prompt and -- when cfg_scale > 1 -- for the
    negative embeddings, `uncond + cfg * (cond - uncond)` in the model dtype (sample.py:139-142),
    Euler update of the fp32 accumulator by dt = t - t_next (sample.py:145-146).
    `rope_starts`: the RoPE offsets of every model call, in call order (cond, uncond per step).
    Returns the fp32 accumulator [1,C,T,H,W]."""
    Pd = {k: v.to(dtype) for k, v in P.items()}
    lat = latents.to(dtype)
    acc = lat.to(torch.float32)
    call = 0
    for i in range(inference_steps, 0, -1):
        t = i / inference_steps
        t_next = (i - 1) / inference_steps
        t = t * alpha / (1 + (alpha - 1) * t)
        t_next = t_next * alpha / (1 + (alpha - 1) * t_next)
        dt = t - t_next
        tt = torch.tensor([t] * lat.shape[0]).to(dtype)
        out = dit_forward(Pd, cfg, lat, prompt_embeds.to(dtype), tt, tuple(rope_starts[call]))
        call += 1
        if cfg_scale > 1:
            un = dit_forward(Pd, cfg, lat, negative_embeds.to(dtype), tt, tuple(rope_starts[call]))
            call += 1
            out = un + cfg_scale * (out - un)
        acc = acc + dt * out.to(torch.float32)
        lat = acc.to(dtype)
    return acc


# --------------------------------------------------------------------------------------
# parameters, muP table, AdamW, LR schedule
# --------------------------------------------------------------------------------------
def param_shapes(cfg: DiTConfig) -> Dict[str, Tuple[int, ...]]:
    """Names and shapes of `DiT.named_parameters()` in registration order
    (model.py:305-350; SURVEY §8(b) listing)."""
    D, pt, p, C = cfg.hidden_size, cfg.time_patch_size, cfg.patch_size, cfg.in_channels
    Hm = int(D * cfg.mlp_ratio)
    S: Dict[str, Tuple[int, ...]] = {}
    S["register_tokens"] = (1, N_REG, D)
    S["patch_embed.patch_proj.weight"] = (D, C, pt, p, p)
    S["patch_embed.patch_proj.bias"] = (D,)
    S["time_embed.0.weight"] = (4 * D, D)
    S["time_embed.0.bias"] = (4 * D,)
    S["time_embed.2.weight"] = (D, 4 * D)
    S["time_embed.2.bias"] = (D,)
    for i in range(cfg.depth):
        b = f"blocks.{i}."
        if cfg.residual_v:
            S[b + "lambda_param"] = (1,)
        if cfg.train_bias_and_rms:
            S[b + "norm1.weight"] = (D,)
        S[b + "qkv.weight"] = (3 * D, D)
        if cfg.train_bias_and_rms:
            S[b + "qkv.bias"] = (3 * D,)
        S[b + "attn_proj.weight"] = (D, D)
        if cfg.cross_attn_input_size is not None:
            if cfg.train_bias_and_rms:
                S[b + "norm2.weight"] = (D,)
            S[b + "q_cross.weight"] = (D, D)
            if cfg.train_bias_and_rms:
                S[b + "q_cross.bias"] = (D,)
            S[b + "context_kv.weight"] = (2 * D, cfg.cross_attn_input_size)
            if cfg.train_bias_and_rms:
                S[b + "context_kv.bias"] = (2 * D,)
            S[b + "cross_proj.weight"] = (D, D)
        if cfg.train_bias_and_rms:
            S[b + "norm3.weight"] = (D,)
        S[b + "mlp.0.weight"] = (Hm, D)
        S[b + "mlp.0.bias"] = (Hm,)
        S[b + "mlp.2.weight"] = (D, Hm)
        S[b + "mlp.2.bias"] = (D,)
        S[b + "adaLN_modulation.1.weight"] = (9 * D, D)
        S[b + "adaLN_modulation.1.bias"] = (9 * D,)
    S["final_modulation.1.weight"] = (2 * D, D)
    S["final_modulation.1.bias"] = (2 * D,)
    if cfg.train_bias_and_rms:
        S["final_norm.weight"] = (D,)
    S["final_proj.weight"] = (cfg.patch_dim, D)
    S["final_proj.bias"] = (cfg.patch_dim,)
    return S


def init_params(cfg: DiTConfig, seed: int = 0, randomize_zero_init: bool = True,
                init_std_factor: float = 0.1, dtype=torch.float32) -> Dict[str, Tensor]:
    """Synthetic weights of the reference architecture: nn.Linear-style uniform(+-1/sqrt(fan_in)),
    all 2-D params x init_std_factor (train.py:247-251), registers ~N(0,1), lambda 0.5
    (model.py:66,316).  The reference zero-inits adaLN / final_modulation / final_proj
    (model.py:93-94,347-350), which makes every layer's gradient but final_proj's vanish, so
    benchmarks and fixtures re-draw them N(0,0.02) (`randomize_zero_init`)."""
    gen = torch.Generator().manual_seed(seed)
    P: Dict[str, Tensor] = {}
    shapes = param_shapes(cfg)
    for n, shp in shapes.items():
        if n == "register_tokens":
            w = torch.randn(shp, generator=gen)
        elif n.endswith("lambda_param"):
            w = torch.full(shp, 0.5)
        elif "norm" in n:
            w = torch.ones(shp)
        else:
            fan_in = 1
            if len(shp) > 1:
                for s in shp[1:]:
                    fan_in *= s
            else:  # bias: fan_in of its weight (same key with .weight)
                wshape = shapes[n[: -len("bias")] + "weight"]
                for s in wshape[1:]:
                    fan_in *= s
            bound = 1.0 / math.sqrt(fan_in)
            w = (torch.rand(shp, generator=gen) * 2 - 1) * bound
            if len(shp) == 2:
                w = w * init_std_factor
            zero_init = ("adaLN_modulation" in n) or ("final_modulation" in n) or ("final_proj" in n)
            if zero_init:
                w = torch.randn(shp, generator=gen) * 0.02 if randomize_zero_init else torch.zeros(shp)
        P[n] = w.to(dtype)
    return P


def mup_settings(shapes: Dict[str, Tuple[int, ...]], learning_rate: float, weight_decay: float,
                 constant_param_classes) -> Dict[str, Dict[str, float]]:
    """name -> {lr, wd} per the rule cascade of DiT.get_mup_setup (model.py:404-465):
    bias|norm|lambda -> lr*0.01, wd 0; else lr*32/last_dim, wd*last_dim/1024; constant
    classes -> lr*0.01, wd 0; 'time' -> lr*0.1; 'modulation' -> lr*0.1 (later rules win)."""
    out = {}
    for n, shp in shapes.items():
        special = None
        for key in ("bias", "norm", "lambda"):
            if key in n:
                special = key
                break
        if special is not None:
            lr, wd = learning_rate * 0.01, 0.0
        else:
            last = shp[-1]
            lr, wd = learning_rate * (32 / last), weight_decay * last / 1024
        if any(cls in n for cls in constant_param_classes):
            lr, wd = learning_rate * 0.01, 0.0
        if "time" in n:
            lr = learning_rate * 0.1
        if "modulation" in n:
            lr = learning_rate * 0.1
        out[n] = {"lr": lr, "wd": wd}
    return out


def adamw_step(p: Tensor, g: Tensor, m: Tensor, v: Tensor, step: int, lr: float, wd: float,
               beta1: float = 0.95, beta2: float = 0.99, eps: float = 1e-8) -> None:
    """torch.optim.AdamW single-tensor update, in place, `step` counted from 1
    (train.py:340-344: betas (0.95,0.99), eps 1e-8, decoupled weight decay)."""
    p.mul_(1 - lr * wd)
    m.mul_(beta1).add_(g, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)


def lr_lambda(step: int, kind: str, warmup: int, total: int) -> float:
    """HF get_{cosine,linear}_schedule_with_warmup multipliers (train.py:349-364);
    'constant' = linear with total 1e10."""
    if kind == "constant":
        kind, total = "linear", 10_000_000_000
    if step < warmup:
        return step / max(1, warmup)
    if kind == "linear":
        return max(0.0, (total - step) / max(1, total - warmup))
    prog = (step - warmup) / max(1, total - warmup)
    return max(0.0, 0.5 * (1.0 + math.cos(math.pi * 0.5 * 2.0 * prog)))


def train_step_flops(cfg: DiTConfig, N: int, Lc: int = 512) -> float:
    """Algorithmic FLOPs of one train step per sample = 3 x forward."""
    D, L = cfg.hidden_size, N + N_REG
    Cc = cfg.cross_attn_input_size or 0
    per_block = 28 * L * D * D + 4 * L * L * D + 4 * L * Lc * D + 4 * Lc * Cc * D + 18 * D * D
    once = 4 * N * cfg.patch_dim * D + 20 * D * D
    return 3.0 * (cfg.depth * per_block + once)
