"""CPU: the oracle restatement reproduces the reference's golden vectors.

The fixtures were produced by oracle/make_golden.py from the imported reference; nothing
here reads /root/reference."""
import os

import pytest
import torch

from oracle import dit_oracle as O

torch.set_num_threads(min(8, os.cpu_count() or 1))


def rel(a, b):
    return ((a.float() - b.float()).norm() / (b.float().norm() + 1e-30)).item()


def check_digest(t, d, tol):
    f = t.detach().float().flatten()
    assert f.numel() == d["numel"]
    assert rel(f[:: d["step"]], d["sample"]) < tol
    assert abs(f.norm().item() - d["norm"]) <= tol * max(d["norm"], 1e-12) * 4


def load_g1(golden_dir, name):
    fx = torch.load(os.path.join(golden_dir, name), weights_only=False)
    cfg = O.DiTConfig(**fx["cfg"])
    P = O.init_params(cfg, **fx["param_init"])
    P.update({k: v.clone() for k, v in fx["param_tweaks"].items()})
    for k, d in fx["param_digest"].items():
        check_digest(P[k], d, 1e-6)  # the seeded init regenerates the exact weights
    return fx, cfg, P


@pytest.mark.parametrize("name", ["g1_tiny_hd64.pt", "g1_tiny_hd72.pt"])
def test_g1_fp32_forward_backward(golden_dir, name):
    fx, cfg, P = load_g1(golden_dir, name)
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    cap = {}
    out = O.dit_forward(Pg, cfg, fx["x"], fx["context"], fx["t"], fx["rope_start"], cap)
    ref = fx["fp32"]
    assert rel(out, ref["out"]) < 2e-5
    for i in range(cfg.depth):
        assert rel(cap[f"blocks.{i}.x_out"], ref["inter"][f"blocks.{i}.x_out"]) < 2e-5
        assert rel(cap[f"blocks.{i}.mod"], ref["inter"][f"blocks.{i}.mod"]) < 2e-5
    assert rel(cap["t_emb"], ref["inter"]["t_emb"]) < 2e-5
    assert rel(cap["final_tokens"], ref["inter"]["final_tokens"]) < 2e-5
    (out * fx["dout"]).sum().backward()
    for k, g in ref["grads"].items():
        if isinstance(g, dict):
            check_digest(Pg[k].grad, g, 3e-4)
        else:
            assert rel(Pg[k].grad, g) < 3e-4, k
    assert Pg["blocks.0.lambda_param"].grad is None  # unused in block 0, as in the reference


@pytest.mark.parametrize("name", ["g1_tiny_hd64.pt", "g1_tiny_hd72.pt"])
def test_g1_bf16_rounding_points(golden_dir, name):
    """bf16 oracle vs the reference run with bf16 parameters: same rounding points, so the
    two agree far tighter than bf16-vs-fp32 does."""
    fx, cfg, P = load_g1(golden_dir, name)
    Pb = {k: v.to(torch.bfloat16) for k, v in P.items()}
    out = O.dit_forward(Pb, cfg, fx["x"].bfloat16(), fx["context"].bfloat16(), fx["t"].bfloat16(),
                        fx["rope_start"])
    e_ref = rel(out, fx["bf16"]["out"])
    e_fp32 = rel(fx["bf16"]["out"], fx["fp32"]["out"])
    assert e_ref < 1.5e-2, e_ref
    assert e_fp32 < 5e-2, e_fp32


def test_g2_dit_s_config1(golden_dir):
    fx = torch.load(os.path.join(golden_dir, "g2_dit_s_c1.pt"), weights_only=False)
    cfg = O.DiTConfig(**fx["cfg"])
    P = O.init_params(cfg, seed=fx["param_seed"], randomize_zero_init=True, init_std_factor=0.1)
    g = torch.Generator().manual_seed(fx["input_seed"])
    x = torch.randn(4, 16, 8, 16, 16, generator=g)
    ctx = torch.randn(4, 512, 4096, generator=g)
    t = O.time_shift(torch.randn(4, generator=g))
    v = torch.randn(4, 16, 8, 16, 16, generator=g)
    check_digest(x, fx["x_digest"], 1e-6)
    check_digest(ctx, fx["ctx_digest"], 1e-6)
    assert torch.allclose(t, fx["t"])
    Pg = {k: w.clone().requires_grad_(True) for k, w in P.items()}
    out = O.dit_forward(Pg, cfg, x, ctx, t, fx["rope_start"])
    assert rel(out, fx["out"]) < 5e-5
    loss, _ = O.flow_loss(v, out)
    assert abs(loss.item() - fx["loss"]) / fx["loss"] < 1e-5
    loss.backward()
    for k, d in fx["grad_digest"].items():
        check_digest(Pg[k].grad, d, 1e-3)
    for k, gfull in fx["grad_full"].items():
        assert rel(Pg[k].grad, gfull) < 1e-3, k


def test_g3_harness(golden_dir):
    fx = torch.load(os.path.join(golden_dir, "g3_harness.pt"), weights_only=False)
    cfg = O.DiTConfig(**fx["cfg"])
    P = O.init_params(cfg, seed=fx["param_seed"], randomize_zero_init=True, init_std_factor=1.0)
    Pb = {k: v.to(torch.bfloat16) for k, v in P.items()}
    # the generator draws of train.py:90-92,103-105
    gen = torch.Generator().manual_seed(fx["gen_seed"])
    z = torch.randn(fx["latent"].shape[0], dtype=torch.bfloat16, generator=gen)
    noise = torch.randn(fx["latent"].shape, dtype=torch.bfloat16, generator=gen)
    assert torch.equal(z, fx["z"]) and torch.equal(noise, fx["noise"])
    # the global-RNG draws: caption-drop mask then the three RoPE offsets (train.py:86, model.py:224-226)
    torch.manual_seed(fx["global_seed"])
    mask = torch.rand(fx["latent"].shape[0]) < 0.01
    assert torch.equal(mask, fx["zero_mask"])
    assert O.draw_rope_offsets((2, 4, 4)) == tuple(fx["rope_start"])
    cap = {}
    loss = O.train_forward(Pb, cfg, fx["latent"], fx["context"], z, noise, fx["rope_start"], cap=cap)
    assert torch.equal(cap["t"], fx["t"])
    assert torch.equal(cap["z_t"], fx["z_t"])
    assert rel(cap["output"], fx["out"]) < 3e-2
    assert abs(loss.item() - fx["loss"]) / fx["loss"] < 1e-2


def test_g4_mup_tables_and_adamw(golden_dir):
    fx = torch.load(os.path.join(golden_dir, "g4_optim.pt"), weights_only=False)
    consts = ["patch_proj", "context_kv", "positional_embedding"]
    for tag, kw in (("dit_s", dict(hidden_size=384, depth=12, num_heads=6)),
                    ("dit_xl", dict(hidden_size=1152, depth=28, num_heads=16))):
        cfg = O.DiTConfig(in_channels=16, cross_attn_input_size=4096, residual_v=True,
                          train_bias_and_rms=False, **kw)
        shapes = O.param_shapes(cfg)
        mine = O.mup_settings(shapes, 1e-4, 0.1, consts)
        ref = fx[tag]["settings"]
        assert list(mine.keys()) == list(ref.keys())  # same names, same registration order
        for k in ref:
            assert tuple(shapes[k]) == tuple(ref[k]["shape"])
            assert mine[k]["lr"] == ref[k]["lr"] and mine[k]["wd"] == ref[k]["wd"], k
        assert len({(v["lr"], v["wd"]) for v in mine.values()}) == fx[tag]["n_groups"]
    a = fx["adamw"]
    cfg = O.DiTConfig(**a["cfg"])
    mine = O.mup_settings(O.param_shapes(cfg), a["lr"], a["wd"], consts)
    for k, p0 in a["p0"].items():
        p, m, v = p0.clone(), torch.zeros_like(p0), torch.zeros_like(p0)
        for s in range(2):
            mult = O.lr_lambda(s, "cosine", 20, 1000)
            O.adamw_step(p, a["grads"][s][k], m, v, s + 1, mine[k]["lr"] * mult, mine[k]["wd"])
        assert rel(p, a["p2"][k]) < 1e-6, k
    sc = fx["sched"]
    for s, (c, l) in enumerate(zip(sc["cosine"], sc["linear"])):
        assert abs(O.lr_lambda(s, "cosine", sc["warmup"], sc["total"]) - c) < 1e-12
        assert abs(O.lr_lambda(s, "linear", sc["warmup"], sc["total"]) - l) < 1e-12


def test_patchify_roundtrip_and_token_order():
    """token order (h w t), feature order (c dt dh dw) in, (dh dw dt c) out."""
    x = torch.arange(2 * 3 * 4 * 4 * 6, dtype=torch.float32).reshape(2, 3, 4, 4, 6)
    tok = O.patchify(x, 2, 2)
    assert tok.shape == (2, 2 * 3 * 2, 3 * 8)
    # token 1 is (h=0,w=0,t=1): its first feature is x[b,0,2,0,0]
    assert tok[0, 1, 0] == x[0, 0, 2, 0, 0]
    # token index of (h=1,w=2,t=0) = (1*3+2)*2
    assert tok[1, (1 * 3 + 2) * 2, 0] == x[1, 0, 0, 2, 4]
    y = torch.randn(2, 12, 2 * 2 * 2 * 3)
    img = O.unpatchify(y, 3, 2, 2, 3, 2, 2)
    assert img.shape == (2, 3, 4, 4, 6)
    # feature (p1=1,p2=0,p3=1,c=2) of token (h=1,w=2,t=1) lands at [c=2, t*2+1, h*2+1, w*2+0]
    f = ((1 * 2 + 0) * 2 + 1) * 3 + 2
    assert img[0, 2, 3, 3, 4] == y[0, (1 * 3 + 2) * 2 + 1, f]


def test_g5_sampler_euler_cfg(golden_dir):
    """sampling/sample.py::generate_image (run unmodified when the fixture was made) vs the oracle loop"""
    fx = torch.load(os.path.join(golden_dir, "g5_sampler.pt"), weights_only=False)
    cfg = O.DiTConfig(**fx["cfg"])
    P = O.init_params(cfg, seed=fx["param_seed"], randomize_zero_init=True, init_std_factor=1.0)
    ctx = fx["context"]
    r = fx["fp32"]
    acc = O.sample_euler_cfg(P, cfg, r["latents0"], ctx, torch.zeros_like(ctx), fx["steps"], fx["cfg_scale"],
                             r["rope_starts"], dtype=torch.float32)
    assert rel(acc.squeeze(0), r["out"]) < 1e-5
    assert len(r["rope_starts"]) == 2 * fx["steps"]  # cond + uncond call per step
    # cfg_scale <= 1: a single model call per step
    acc1 = O.sample_euler_cfg(P, cfg, r["latents0"], ctx, torch.zeros_like(ctx), 2, 1.0, r["rope_starts"],
                              dtype=torch.float32)
    assert torch.isfinite(acc1).all()


def test_chunked_attention_equals_plain_attention(monkeypatch):
    """the long-sequence path of the oracle (query-chunked SDPA with its closed-form backward, used above
    2^29 score elements: BASELINE config 4) equals the plain softmax(QK^T)V + autograd path"""
    g = torch.Generator().manual_seed(0)
    q, k, v = (torch.randn(2, 3, n, 72, generator=g) for n in (301, 217, 217))
    do = torch.randn(2, 3, 301, 72, generator=g)
    qa, ka, va = (t.clone().requires_grad_(True) for t in (q, k, v))
    o_ref = O.attention(qa, ka, va)
    o_ref.backward(do)
    o, lse, dq, dk, dv = O.attention_chunked(q, k, v, do, chunk=64)
    s = (q @ k.transpose(-1, -2)) / 72 ** 0.5
    assert rel(o, o_ref.detach()) < 1e-6 and rel(lse, torch.logsumexp(s, -1)) < 1e-6
    assert rel(dq, qa.grad) < 1e-5 and rel(dk, ka.grad) < 1e-5 and rel(dv, va.grad) < 1e-5
    # and `attention` itself switches to it (autograd wrapper) above the threshold
    monkeypatch.setattr(O, "CHUNKED_ATTENTION_ABOVE", 1000)
    qb, kb, vb = (t.clone().requires_grad_(True) for t in (q, k, v))
    o2 = O.attention(qb, kb, vb)
    o2.backward(do)
    assert rel(o2.detach(), o_ref.detach()) < 1e-6 and rel(qb.grad, qa.grad) < 1e-5
    assert rel(kb.grad, ka.grad) < 1e-5 and rel(vb.grad, va.grad) < 1e-5
