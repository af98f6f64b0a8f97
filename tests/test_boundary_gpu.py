"""The block-level boundary of SURVEY.md 8(b): `DiTBlock.forward(x, context, c, v_0, rope) -> (x, v)`,
`RMSNorm.forward`, `PatchEmbed.forward` and `apply_rotary_emb` called with the reference's signatures
(model.py:96-167, 34-41, 182-186, 266-275), running on the HIP kernels with autograd through the hand-written
backward kernels -- against the reference's per-block captures (tests/golden/g1_*.pt: `patch_tokens`, `t_emb`,
`blocks.{i}.x_out`) and against the fp32 oracle for the gradients."""
import os

import pytest
import torch

from oracle import dit_oracle as O

pytestmark = pytest.mark.gpu
bf16, f32 = torch.bfloat16, torch.float32


@pytest.fixture(scope="module")
def M():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from video_diffusion_speedrun_amd import model
    return model


def rel(a, b):
    a, b = a.float().cpu().flatten(), b.float().cpu().flatten()
    return ((a - b).norm() / (b.norm() + 1e-20)).item()


def load_g1(golden_dir, name):
    fx = torch.load(os.path.join(golden_dir, name), weights_only=False)
    cfg = O.DiTConfig(**fx["cfg"])
    P = O.init_params(cfg, **fx["param_init"])
    P.update({k: v.clone() for k, v in fx["param_tweaks"].items()})
    return fx, cfg, P


def make_block(M, cfg, P, i):
    blk = M.DiTBlock(hidden_size=cfg.hidden_size, cross_attn_input_size=cfg.cross_attn_input_size,
                     num_heads=cfg.num_heads, mlp_ratio=cfg.mlp_ratio, qkv_bias=cfg.train_bias_and_rms,
                     residual_v=cfg.residual_v)
    pre = f"blocks.{i}."
    blk.load_state_dict({k[len(pre):]: v for k, v in P.items() if k.startswith(pre)}, strict=True)
    return blk.cuda()


@pytest.mark.parametrize("name", ["g1_tiny_hd64.pt", "g1_tiny_hd72.pt"])
def test_block_forward_matches_reference_captures(M, golden_dir, name):
    """two stand-alone DiTBlocks chained exactly like model.py:379-384 on the reference's own inputs to block 0"""
    fx, cfg, P = load_g1(golden_dir, name)
    inter = fx["fp32"]["inter"]
    B, C, T, H, W = fx["x"].shape
    thw = (T // cfg.time_patch_size, H // cfg.patch_size, W // cfg.patch_size)
    rope = M.ThreeDimRotary(cfg.hidden_size // (2 * cfg.num_heads)).cuda()
    cos, sin = rope(None, time_height_width=thw, extend_with_register_tokens=16, start=tuple(fx["rope_start"]))
    assert cos.shape == (1, 1, 16 + thw[0] * thw[1] * thw[2], cfg.head_dim // 2)
    x0 = torch.cat([P["register_tokens"].repeat(B, 1, 1), inter["patch_tokens"]], 1).to(bf16).cuda()
    c = inter["t_emb"].to(bf16).cuda()
    ctx = fx["context"].to(bf16).cuda()
    b0, b1 = make_block(M, cfg, P, 0), make_block(M, cfg, P, 1)
    with torch.no_grad():
        x1, v = b0(x0, ctx, c, v_0=None, rope=(cos, sin))
        assert x1.shape == x0.shape and v.shape == (B, cfg.num_heads, x0.shape[1], cfg.head_dim)
        e_ref = rel(fx["bf16"]["inter"]["blocks.0.x_out"], inter["blocks.0.x_out"]) if "inter" in fx["bf16"] else 0.0
        assert rel(x1, inter["blocks.0.x_out"]) <= max(2.5 * e_ref, 1.5e-2)
        x2, v1 = b1(inter["blocks.0.x_out"].to(bf16).cuda(), ctx, c, v_0=v, rope=(cos, sin))
        assert rel(x2, inter["blocks.1.x_out"]) <= max(2.5 * e_ref, 1.5e-2)
        # a plain [B,H,L,hd] copy of v_0 (not the padded buffer the block handed out) gives the same result
        x2b, _ = b1(inter["blocks.0.x_out"].to(bf16).cuda(), ctx, c, v_0=v.clone().contiguous(), rope=(cos, sin))
        assert torch.equal(x2, x2b)


def test_block_backward_matches_oracle(M):
    """gradients w.r.t. x, c, v_0 and every parameter of a block that mixes v_0 (model.py:129-130), and of a
    block 0 whose returned v is consumed downstream, against the fp32 oracle's autograd"""
    cfg = O.DiTConfig(in_channels=16, hidden_size=144, depth=2, num_heads=2, cross_attn_input_size=64,
                      residual_v=True, train_bias_and_rms=True)
    P = O.init_params(cfg, seed=5, randomize_zero_init=True, init_std_factor=1.0)
    g = torch.Generator().manual_seed(6)
    B, L, Lc, D, H, hd = 2, 16 + 64, 24, 144, 2, 72
    x = torch.randn(B, L, D, generator=g).to(bf16)
    ctx = torch.randn(B, Lc, 64, generator=g).to(bf16)
    c = torch.randn(B, D, generator=g).to(bf16)
    gx = torch.randn(B, L, D, generator=g).to(bf16)
    cos, sin = O.rope_cos_sin(hd, (4, 4, 4), (3, 5, 7), 16)
    # oracle: block 0 then block 1 (which mixes block 0's v); loss = <x2, gx>
    leaf = {k: w.clone().requires_grad_(True) for k, w in P.items() if k.startswith("blocks.")}
    xo, co = x.float().requires_grad_(True), c.float().requires_grad_(True)
    y1, v0 = O.block_forward(leaf, "blocks.0.", cfg, xo, ctx.float(), co, None, cos, sin)
    y2, _ = O.block_forward(leaf, "blocks.1.", cfg, y1, ctx.float(), co, v0, cos, sin)
    (y2 * gx.float()).sum().backward()
    # HIP: the same chain through the reference-signature calls
    b0, b1 = make_block(M, cfg, P, 0), make_block(M, cfg, P, 1)
    xh, ch = x.cuda().requires_grad_(True), c.cuda().requires_grad_(True)
    rope = (cos[None, None].cuda(), sin[None, None].cuda())
    h1, hv = b0(xh, ctx.cuda(), ch, v_0=None, rope=rope)
    h2, _ = b1(h1, ctx.cuda(), ch, v_0=hv, rope=rope)
    assert rel(h2, y2.detach()) <= 1.5e-2
    (h2.float() * gx.cuda().float()).sum().backward()
    assert rel(xh.grad, xo.grad) <= 3e-2 and rel(ch.grad, co.grad) <= 3e-2, (rel(xh.grad, xo.grad), rel(ch.grad, co.grad))
    bad = []
    for i, blk in enumerate((b0, b1)):
        for n, p in blk.named_parameters():
            ref = leaf[f"blocks.{i}.{n}"].grad
            if ref is None or n == "lambda_param":
                continue
            e = rel(p.grad, ref)
            if e > 3e-2:
                bad.append((i, n, e))
    assert not bad, bad
    lam = b1.lambda_param.grad.item()
    assert abs(lam - leaf["blocks.1.lambda_param"].grad.item()) <= 0.1 * abs(lam) + 1e-3


def test_rmsnorm_patchembed_rotary_are_callable(M, golden_dir):
    fx, cfg, P = load_g1(golden_dir, "g1_tiny_hd72.pt")
    g = torch.Generator().manual_seed(7)
    # RMSNorm (model.py:34-41), with and without weight, forward + backward vs the oracle
    for trainable in (False, True):
        norm = M.RMSNorm(144, trainable=trainable).cuda()
        if trainable:
            with torch.no_grad():
                norm.weight.copy_(torch.rand(144, generator=g) + 0.5)
        x = torch.randn(3, 50, 144, generator=g).to(bf16)
        gy = torch.randn(3, 50, 144, generator=g).to(bf16)
        xo = x.float().requires_grad_(True)
        wo = norm.weight.detach().cpu().clone().requires_grad_(True) if trainable else None
        yo = O.rms_norm(xo, wo)
        (yo * gy.float()).sum().backward()
        xh = x.cuda().requires_grad_(True)
        yh = norm(xh)
        assert yh.dtype == bf16 and rel(yh, yo.detach()) <= 5e-3
        (yh.float() * gy.cuda().float()).sum().backward()
        assert rel(xh.grad, xo.grad) <= 1e-2
        if trainable:
            assert norm.weight.grad.dtype == f32 and rel(norm.weight.grad, wo.grad) <= 1e-2
    # PatchEmbed (model.py:170-186) vs the reference's captured patch tokens
    pe = M.PatchEmbed(cfg.patch_size, cfg.in_channels, cfg.hidden_size, cfg.time_patch_size)
    pe.load_state_dict({"patch_proj.weight": P["patch_embed.patch_proj.weight"],
                        "patch_proj.bias": P["patch_embed.patch_proj.bias"]})
    pe = pe.cuda()
    tok = pe(fx["x"].to(bf16).cuda())
    assert rel(tok, fx["fp32"]["inter"]["patch_tokens"]) <= 1e-2
    gt = torch.randn(tok.shape, generator=g).to(bf16)
    (tok.float() * gt.cuda().float()).sum().backward()
    wo = P["patch_embed.patch_proj.weight"].clone().requires_grad_(True)
    bo = P["patch_embed.patch_proj.bias"].clone().requires_grad_(True)
    to = O.patch_embed(fx["x"].to(bf16).float(), wo, bo, cfg.time_patch_size, cfg.patch_size)
    (to * gt.float()).sum().backward()
    assert rel(pe.patch_proj.weight.grad, wo.grad) <= 1e-2 and rel(pe.patch_proj.bias.grad, bo.grad) <= 1e-2
    # apply_rotary_emb (model.py:266-275)
    q = torch.randn(2, 2, 80, 72, generator=g).to(bf16)
    gq = torch.randn(2, 2, 80, 72, generator=g).to(bf16)
    cos, sin = O.rope_cos_sin(72, (4, 4, 4), (1, 2, 3), 16)
    qo = q.float().requires_grad_(True)
    ro = O.apply_rotary(qo, cos, sin)
    (ro * gq.float()).sum().backward()
    qh = q.cuda().requires_grad_(True)
    rh = M.apply_rotary_emb(qh, cos[None, None].cuda(), sin[None, None].cuda())
    assert rel(rh, ro.detach()) <= 5e-3
    (rh.float() * gq.cuda().float()).sum().backward()
    assert rel(qh.grad, qo.grad) <= 5e-3
    with pytest.raises(RuntimeError):
        M.apply_rotary_emb(q, cos[None, None], sin[None, None])   # CPU tensors: no fallback
