"""GPU parity at the workloads BASELINE.json names, at their real sizes (SURVEY.md §8 shape table):

    C2  DiT-B/2   D=768, 12 heads of 64, latent [16,16,32,32] pt=1  -> 4096+16 tokens
    C3b DiT-XL/2  D=1152, 16 heads of 72, latent [16,16,64,64] pt=2 -> 8192+16 tokens   (bf16 and, = C5, fp8 GEMMs)
    C4  DiT-XL/2  latent [16,33,64,64] pt=1                         -> 33792+16 tokens

Each test runs ONE block of the model (+ patch embed, conditioning, final layer) at the full token
count and context [512,4096] against the fp32 CPU oracle -- output, loss and every parameter
gradient -- plus the attention kernels alone at the C4 length against the query-chunked fp32
oracle, and a deeper DiT-XL stack at a shorter sequence.  Tolerances: see test_model_gpu.py; the
measured figures are appended to gpurun_out/parity_report.jsonl.
"""
import math
import os

import pytest
import torch

from oracle import dit_oracle as O

pytestmark = pytest.mark.gpu
bf16, f32 = torch.bfloat16, torch.float32

# bf16 path, per parameter tensor.  Measured on the MI355X (gpurun_out/parity_report.jsonl, round 2): worst
# rel-L2 8.6e-3 (blocks.5.qkv.weight at depth 6), worst cosine 0.99997 -> bounds at ~2x the measured error
GRAD_COS, GRAD_REL = 0.9995, 2e-2
# fp8 linears + fp8 self-attention (e4m3 / e5m2 operands, P and dS included): test_model_gpu.py::test_fp8_step_close_to_oracle,
# test_attn_fp8_gpu.py::test_fp8_attention_step_close_to_oracle
FP8_GRAD_COS, FP8_GRAD_REL = 0.995, 0.11  # round 4: measured worst at the headline shape 0.9975 / 7.8e-2 (was 0.98 / 0.2)


@pytest.fixture(scope="module")
def vds():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from video_diffusion_speedrun_amd import model, ops, optim, train
    return dict(model=model, ops=ops, optim=optim, train=train)


def rel(a, b):
    a, b = a.float().cpu().flatten(), b.float().cpu().flatten()
    return ((a - b).norm() / (b.norm() + 1e-20)).item()


def cosine(a, b):
    a, b = a.float().cpu().flatten(), b.float().cpu().flatten()
    return (a @ b / (a.norm() * b.norm() + 1e-30)).item()


def build(vds, cfg, P, fp8=False):
    m = vds["model"].DiT(in_channels=cfg.in_channels, patch_size=cfg.patch_size, time_patch_size=cfg.time_patch_size,
                         hidden_size=cfg.hidden_size, depth=cfg.depth, num_heads=cfg.num_heads,
                         mlp_ratio=cfg.mlp_ratio, cross_attn_input_size=cfg.cross_attn_input_size,
                         residual_v=cfg.residual_v, train_bias_and_rms=cfg.train_bias_and_rms)
    m.load_state_dict(P, strict=True)
    m = m.to("cuda")
    return m.enable_fp8() if fp8 else m


def oracle_step(cfg, P, x, ctx, t, v, start):
    """fp32 oracle forward + loss + backward on the bf16-rounded inputs; returns detached results"""
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    Pg = {k: w.clone().requires_grad_(True) for k, w in P.items()}
    cap = {}
    o_ref = O.dit_forward(Pg, cfg, x.float(), ctx.float(), t.float(), start, cap)
    mixed = {i: cap[f"blocks.{i}.v"] for i in range(1, cfg.depth)} if cfg.residual_v else {}
    for vm in mixed.values():
        vm.retain_grad()
    l_ref, _ = O.flow_loss(v, o_ref)
    l_ref.backward()
    grads = {k: w.grad for k, w in Pg.items() if w.grad is not None}
    # d loss / d lambda_i = sum(dv_i * (v_raw_i - v_0)) is a scalar made of B*L*D signed products that largely
    # cancel; its natural error scale is the sum of the |products|, not the cancelled sum:
    # v_raw - v_0 = (v_mixed - v_0) / lambda  (model.py:129-130)
    v0 = cap["blocks.0.v"].detach()
    for i, vm in mixed.items():
        lam = P[f"blocks.{i}.lambda_param"].item()
        grads[f"blocks.{i}.lambda_param.l1"] = (vm.grad * (vm.detach() - v0) / lam).abs().sum()
    # sub-layer outputs of the LAST block (the model output is blind to in-block errors -- with N(0, 0.02)
    # gates a block contributes ~2 % of the residual stream -- so these are compared directly)
    pre = f"blocks.{cfg.depth - 1}."
    subs = {n: cap[pre + n].detach() for n in SUBLAYERS if pre + n in cap}
    return o_ref.detach(), l_ref.item(), grads, subs


SUBLAYERS = ("q_rope", "k_rope", "attn", "y_sa", "cattn", "y_ca", "y_mlp")
# rel-L2 of a bf16 sub-layer output against the fp32 oracle (measured worst: see parity_report.jsonl, "sublayers")
SUB_REL, FP8_SUB_REL = 1.5e-2, 1.2e-1  # fp8: measured <= 6.0e-2 (cross_proj output; cross-attention itself 4.6e-2)


def sublayer_errors(m, out, subs, B, L):
    """{name: rel-L2} of the last block's sub-layer outputs, read from the activations the forward saved for its
    backward (`out.grad_fn.sv`); call BEFORE backward (which frees them)"""
    bs = out.grad_fn.sv.blocks[-1]
    H, hd, D = m.num_heads, m.head_dim, m.hidden_size
    got = {"attn": bs.attn.view(B, L, D), "y_sa": bs.y_sa.view(B, L, D), "y_mlp": bs.y_mlp.view(B, L, D)}
    if bs.q is not None:  # (the fp8 attention path keeps e4m3 copies instead)
        got["q_rope"], got["k_rope"] = bs.q[..., :hd], bs.k[..., :hd]
    if bs.has_cross:
        got["cattn"], got["y_ca"] = bs.catt.view(B, L, D), bs.y_ca.view(B, L, D)
    return {n: rel(got[n], subs[n]) for n in got if n in subs}


def check_step(vds, m, x, ctx, t, v, start, ref, cos_min, rel_max, out_tol=2.5e-2, sub_tol=SUB_REL, lam_tol=None,
               log=None):
    """log = (parity_log, name): the measured figures are recorded BEFORE the assertions (a failing run keeps them)"""
    o_ref, l_ref, g_ref, subs = ref
    out = m(x.cuda(), ctx.cuda(), t.cuda(), rope_start=start)
    e_out = rel(out, o_ref)
    L = (x.shape[2] // m.time_patch_size) * (x.shape[3] // m.patch_size) * (x.shape[4] // m.patch_size) + 16
    sub_err = sublayer_errors(m, out, subs, x.shape[0], L)
    loss, _ = vds["train"].flow_loss(out, v.cuda())
    e_loss = abs(loss.item() - l_ref) / abs(l_ref)
    loss.backward()
    rows, lam = [], []
    for k, p in m.named_parameters():
        if k not in g_ref or float(g_ref[k].abs().max()) == 0:
            continue
        if k.endswith("lambda_param"):  # (name, got, reference, |got - ref| / sum of |terms|)
            got, want = p.grad.item(), g_ref[k].item()
            lam.append((k, got, want, abs(got - want) / g_ref[k + ".l1"].item()))
            continue
        rows.append((k, cosine(p.grad, g_ref[k]), rel(p.grad, g_ref[k])))
    worst_cos = min(rows, key=lambda r: r[1])
    worst_rel = max(rows, key=lambda r: r[2])
    per_block = {}  # worst gradient rel-L2 per block (error growth over depth)
    for k, c, e in rows:
        if k.startswith("blocks."):
            b = int(k.split(".")[1])
            per_block[b] = max(per_block.get(b, 0.0), e)
    figures = dict(out_rel=e_out, loss_rel=e_loss, worst_cos=worst_cos, worst_rel=worst_rel, lambda_param=lam,
                   sublayers=sub_err, worst_rel_per_block=[round(per_block[b], 5) for b in sorted(per_block)])
    if log is not None:
        log[0](log[1], **figures)
    assert e_out <= out_tol, figures
    assert sub_err and all(e <= sub_tol for e in sub_err.values()), sub_err
    assert e_loss <= 1e-2, figures
    bad = [r for r in rows if not (r[1] >= cos_min and r[2] <= rel_max)]
    assert not bad, bad
    assert all(r[3] <= (lam_tol or LAMBDA_ERR) for r in lam), [r for r in lam if r[3] > (lam_tol or LAMBDA_ERR)]
    return figures


def make_inputs(lat_shape, Lc, Cc, seed, tval):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(*lat_shape, generator=g).to(bf16)
    ctx = torch.randn(lat_shape[0], Lc, Cc, generator=g).to(bf16)
    t = torch.tensor([tval] * lat_shape[0]).to(bf16)
    v = torch.randn(*lat_shape, generator=g).to(bf16)
    return x, ctx, t, v


# --------------------------------------------------------------------------------- C2 ----
@pytest.mark.timeout(600)
def test_c2_dit_b_block_vs_oracle(vds, parity_log):
    """BASELINE configs[1]: DiT-B width (768 = 12 heads of 64), latent [1,16,16,32,32] with time-patch 1 ->
    4096+16 tokens, context [512,4096]; one block + embed / final layers vs the fp32 oracle."""
    cfg = O.DiTConfig(in_channels=16, patch_size=2, time_patch_size=1, hidden_size=768, depth=1, num_heads=12,
                      cross_attn_input_size=4096, residual_v=True, train_bias_and_rms=False)
    P = O.init_params(cfg, seed=81, randomize_zero_init=True, init_std_factor=0.1)
    x, ctx, t, v = make_inputs((1, 16, 16, 32, 32), 512, 4096, 82, 0.45)
    start = (11, 5, 60)
    ref = oracle_step(cfg, P, x, ctx, t, v, start)
    fig = check_step(vds, build(vds, cfg, P), x, ctx, t, v, start, ref, GRAD_COS, GRAD_REL)
    parity_log("c2_dit_b_block", **fig)


@pytest.mark.timeout(600)
def test_c2_dit_b_two_blocks_residual_v(vds, parity_log):
    """same shape, two blocks: block 1 mixes its V with block 0's (residual-V, lambda_param gradient)"""
    cfg = O.DiTConfig(in_channels=16, patch_size=2, time_patch_size=1, hidden_size=768, depth=2, num_heads=12,
                      cross_attn_input_size=4096, residual_v=True, train_bias_and_rms=True)
    P = O.init_params(cfg, seed=83, randomize_zero_init=True, init_std_factor=0.1)
    x, ctx, t, v = make_inputs((1, 16, 16, 32, 32), 512, 4096, 84, 0.7)
    start = (0, 96, 3)
    ref = oracle_step(cfg, P, x, ctx, t, v, start)
    fig = check_step(vds, build(vds, cfg, P), x, ctx, t, v, start, ref, GRAD_COS, GRAD_REL)
    parity_log("c2_dit_b_two_blocks", **fig)
    assert fig["lambda_param"]


# blocks.{i>0}.lambda_param gradient: |got - ref| <= LAMBDA_ERR * sum_j |dv_j (v_raw_j - v_0_j)| -- every one of the
# B*L*D bf16 products carries ~2^-9 relative error; checked per block (the HIP kernel accumulates them in fp32)
LAMBDA_ERR = 1e-4  # measured worst 4.1e-5 (gpurun_out/parity_report.jsonl, DiT-XL depth 6)


# -------------------------------------------------------------------------------- C3a ----
@pytest.mark.timeout(900)
def test_c3a_literal_shape_block_vs_oracle(vds, parity_log):
    """BASELINE configs[2] literally: DiT-XL width (16 heads of 72) on latent [1,16,17,32,32] -- an ODD frame count,
    which only round-trips with `time_patch_size=1` (model.py:283; with pt=2 the Conv3d floors T to 16 and the
    unpatchify of model.py:392-401 no longer matches the latent) -> 17*16*16 = 4352 video tokens + 16 registers,
    context [512,4096].  Two blocks (block 1 mixes block 0's V) + embed / final layers against the fp32 oracle:
    output, loss, every gradient, the last block's sub-layer outputs."""
    cfg = O.DiTConfig(in_channels=16, patch_size=2, time_patch_size=1, hidden_size=1152, depth=2, num_heads=16,
                      cross_attn_input_size=4096, residual_v=True, train_bias_and_rms=False)
    P = O.init_params(cfg, seed=61, randomize_zero_init=True, init_std_factor=0.1)
    x, ctx, t, v = make_inputs((1, 16, 17, 32, 32), 512, 4096, 62, 0.5)
    start = (100, 3, 57)  # 17 frames from t = 100: the RoPE time axis is used up to row 116 of its 128
    ref = oracle_step(cfg, P, x, ctx, t, v, start)
    m = build(vds, cfg, P)
    fig = check_step(vds, m, x, ctx, t, v, start, ref, GRAD_COS, GRAD_REL, log=(parity_log, "c3a_literal_shape"))
    assert len(fig["lambda_param"]) == 1
    out = m(x.cuda(), ctx.cuda(), t.cuda(), rope_start=start)
    assert tuple(out.shape) == (1, 16, 17, 32, 32)


# ---------------------------------------------------------------------------- C3b / C5 ----
@pytest.fixture(scope="module")
def headline():
    cfg = O.DiTConfig(in_channels=16, patch_size=2, time_patch_size=2, hidden_size=1152, depth=1, num_heads=16,
                      cross_attn_input_size=4096, residual_v=True, train_bias_and_rms=False)
    P = O.init_params(cfg, seed=71, randomize_zero_init=True, init_std_factor=0.1)
    x, ctx, t, v = make_inputs((1, 16, 16, 64, 64), 512, 4096, 72, 0.55)
    start = (9, 21, 33)
    return cfg, P, (x, ctx, t, v), start, oracle_step(cfg, P, x, ctx, t, v, start)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("fp8", [False, True], ids=["c3b_bf16", "c5_fp8"])
def test_headline_shape_block_vs_oracle(vds, headline, parity_log, fp8):
    """ONE DiT-XL block (+ embed / final layers) at the headline shape -- latent [1,16,16,64,64], 8192+16
    tokens, 16 heads of 72, context [512,4096] -- against the fp32 CPU oracle: output, loss and every
    gradient; bf16 (C3b) and with DiT.enable_fp8() (BASELINE configs[4]: fp8 qkv / MLP GEMMs and fp8 self-attention
    products) under the fp8 bounds.  (The oracle needs ~14 GB of host memory and some tens of seconds; it runs once for both.)"""
    cfg, P, (x, ctx, t, v), start, ref = headline
    m = build(vds, cfg, P, fp8=fp8)
    if fp8:
        # delayed scaling: one pass records the amax history (its attention still runs on the bf16 kernels), the
        # checked pass then quantises q / k / v / dO with it and runs the fp8 attention kernels
        out = m(x.cuda(), ctx.cuda(), t.cuda(), rope_start=start)
        loss, _ = vds["train"].flow_loss(out, v.cuda())
        loss.backward()
        m.zero_grad()
        del out, loss
        vds["ops"].prof_enable()
    fig = check_step(vds, m, x, ctx, t, v, start, ref, FP8_GRAD_COS if fp8 else GRAD_COS,
                     FP8_GRAD_REL if fp8 else GRAD_REL, sub_tol=FP8_SUB_REL if fp8 else SUB_REL)
    if fp8:
        stats = vds["ops"].prof_collect()
        vds["ops"].prof_enable(0)
        assert stats["gemm_fp8"]["launches"] == 20  # the 7 linears of the block x (fwd, dgrad, wgrad); context_kv has no dgrad
        # ... and so did the three attention kernels of the self- AND the cross-attention; no bf16 attention launch is left
        assert [stats[k]["launches"] for k in ("attn_fp8_fwd", "attn_fp8_dkv", "attn_fp8_dq")] == [2, 2, 2]
        assert not any(k in stats for k in ("attn_fwd", "attn_bwd_dkv", "attn_bwd_dq", "attn_fwd_plain",
                                            "attn_bwd_dkv_plain", "attn_bwd_dq_plain"))
    parity_log("headline_block_" + ("c5_fp8" if fp8 else "c3b_bf16"), **fig)


@pytest.mark.timeout(900)
def test_dit_xl_depth6_vs_oracle(vds, parity_log):
    """DiT-XL width (16 heads of 72) at depth 6 on 1024+16 tokens, B=2: fan-in of the residual-V gradient and of
    the conditioning gradient over several blocks, every gradient vs the fp32 oracle"""
    cfg = O.DiTConfig(in_channels=16, patch_size=2, time_patch_size=2, hidden_size=1152, depth=6, num_heads=16,
                      cross_attn_input_size=4096, residual_v=True, train_bias_and_rms=False)
    P = O.init_params(cfg, seed=91, randomize_zero_init=True, init_std_factor=0.1)
    x, ctx, t, v = make_inputs((2, 16, 8, 32, 32), 64, 4096, 92, 0.3)
    t = torch.tensor([0.3, 0.85]).to(bf16)
    start = (50, 2, 77)
    ref = oracle_step(cfg, P, x, ctx, t, v, start)
    fig = check_step(vds, build(vds, cfg, P), x, ctx, t, v, start, ref, GRAD_COS, GRAD_REL)
    parity_log("dit_xl_depth6", **fig)
    assert len(fig["lambda_param"]) == 5


# ----------------------------------------------------------------- full depth (28 blocks) ----
# Error grows with depth (the residual-V and conditioning gradients fan in over all blocks, model.py:379-384): the
# full DiT-XL depth is checked at a short sequence in the default suite and at the headline sequence on request.
GRAD_COS_28, GRAD_REL_28 = 0.999, 4e-2  # measured: parity_report.jsonl "dit_xl_depth28_*" (bounds ~2x the worst block)
LAMBDA_ERR_28 = 4e-4  # lambda_param gradients at depth 28 (see LAMBDA_ERR; the upstream dv carries 28 blocks of bf16 error)


def _depth28(vds, parity_log, lat, Lc, name, seed):
    cfg = O.DiTConfig(in_channels=16, patch_size=2, time_patch_size=2, hidden_size=1152, depth=28, num_heads=16,
                      cross_attn_input_size=4096, residual_v=True, train_bias_and_rms=False)
    P = O.init_params(cfg, seed=seed, randomize_zero_init=True, init_std_factor=0.1)
    x, ctx, t, v = make_inputs(lat, Lc, 4096, seed + 1, 0.4)
    start = (7, 40, 19)
    ref = oracle_step(cfg, P, x, ctx, t, v, start)
    fig = check_step(vds, build(vds, cfg, P), x, ctx, t, v, start, ref, GRAD_COS_28, GRAD_REL_28,
                     lam_tol=LAMBDA_ERR_28, log=(parity_log, name))
    assert len(fig["lambda_param"]) == 27 and len(fig["worst_rel_per_block"]) == 28
    return fig


@pytest.mark.timeout(1500)
def test_dit_xl_depth28_short_sequence_vs_oracle(vds, parity_log):
    """the full-depth DiT-XL (28 blocks, 16 heads of 72) on 1024+16 tokens, B=1, context [64,4096]: output, loss,
    every gradient (worst per block recorded) and the last block's sub-layer outputs vs the fp32 oracle"""
    _depth28(vds, parity_log, (1, 16, 8, 32, 32), 64, "dit_xl_depth28_1040_tokens", 121)


@pytest.mark.slow
@pytest.mark.timeout(3600)
def test_dit_xl_depth28_headline_sequence_vs_oracle(vds, parity_log):
    """opt-in (-m "gpu and slow"; ~10 min of CPU oracle, ~30 GB of host memory): the whole C3b model -- 28 blocks at
    8192+16 tokens, context [512,4096], B=1 -- against the fp32 oracle.  The log of the last run on the MI355X box is
    kept under profiles/."""
    _depth28(vds, parity_log, (1, 16, 16, 64, 64), 512, "dit_xl_depth28_8208_tokens", 131)


# --------------------------------------------------------------------------------- C4 ----
C4_L = 33 * 32 * 32 + 16  # 33 808 tokens


@pytest.mark.timeout(900)
def test_c4_attention_kernels_vs_chunked_oracle(vds, parity_log):
    """self-attention forward, dQ and dK/dV kernels at the C4 length (33 808 queries x 33 808 keys, head_dim
    72 with the ones-column padding the model uses), 2 heads, every row against the query-chunked fp32 oracle"""
    ops = vds["ops"]
    B, H, hd, hdp, L = 1, 2, 72, 96, C4_L
    g = torch.Generator().manual_seed(101)
    q, k, v = (torch.randn(B, H, L, hd, generator=g).to(bf16) for _ in range(3))
    q[:, :, 777] *= 6.0                      # a few peaked rows: exercises the lazy rescale far into the key loop
    k[:, :, L - 100] = 4.0 * q[:, :, 12345]
    do = torch.randn(B, L, H * hd, generator=g).to(bf16)
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    o_ref, lse_ref, dq_ref, dk_ref, dv_ref = O.attention_chunked(
        q, k, v, do.reshape(B, L, H, hd).permute(0, 2, 1, 3), chunk=1024)

    def pad(t_, ones):
        out = torch.zeros(*t_.shape[:-1], hdp, dtype=bf16)
        out[..., :hd] = t_
        for c in ones:
            out[..., c] = 1
        return out.cuda()
    qd, kd, vd = pad(q, []), pad(k, [hd, hd + 1]), pad(v, [hd, hd + 4])
    o = torch.zeros(B * L, H * hd, dtype=bf16, device="cuda")
    lse = torch.zeros(B, H, L, dtype=f32, device="cuda")
    ov = ops.heads_view(o, B, L, H, hd)
    ops.attn_fwd(qd[..., :hd], kd[..., :hd], vd[..., :hd], ov, lse, kv_pad_ones=True)
    dq, dk, dv = torch.zeros_like(qd), torch.zeros_like(kd), torch.zeros_like(vd)
    delta = torch.zeros(2, B, H, L, dtype=f32, device="cuda")
    ops.attn_bwd(qd[..., :hd], kd[..., :hd], vd[..., :hd], ov, lse,
                 ops.heads_view(do.reshape(B * L, H * hd).cuda(), B, L, H, hd),
                 dq[..., :hd], dk[..., :hd], dv[..., :hd], delta, kv_pad_ones=True)
    fig = dict(o=rel(o.view(B, L, H, hd).permute(0, 2, 1, 3), o_ref), lse=rel(lse, lse_ref),
               dq=rel(dq[..., :hd], dq_ref), dk=rel(dk[..., :hd], dk_ref), dv=rel(dv[..., :hd], dv_ref))
    parity_log("c4_attention", **fig)
    assert fig["o"] <= 1e-2 and fig["lse"] <= 2e-3, fig
    assert fig["dq"] <= 2e-2 and fig["dk"] <= 2e-2 and fig["dv"] <= 2e-2, fig  # measured 5e-3 / 1.1e-2 / 1.1e-2
    # per-row worst case too: no single query / key row may be off (a wrong tile would hide in the norm)
    row_err = ((o.view(B, L, H, hd).permute(0, 2, 1, 3).float().cpu() - o_ref).norm(dim=-1) /
               (o_ref.norm(dim=-1) + 1e-3)).max().item()
    assert row_err <= 0.1, row_err
    assert dq[..., hd:].abs().max().item() == 0


@pytest.mark.timeout(1500)
def test_c4_dit_xl_block_vs_oracle(vds, parity_log):
    """BASELINE configs[3]: ONE DiT-XL block at latent [1,16,33,64,64], time-patch 1 -> 33 792+16 tokens, context
    [512,4096], against the fp32 oracle (its attention runs query-chunked: oracle.attention_chunked); then the
    size-independent properties at this length: a duplicated sample gives bit-identical outputs."""
    cfg = O.DiTConfig(in_channels=16, patch_size=2, time_patch_size=1, hidden_size=1152, depth=1, num_heads=16,
                      cross_attn_input_size=4096, residual_v=True, train_bias_and_rms=False)
    P = O.init_params(cfg, seed=111, randomize_zero_init=True, init_std_factor=0.1)
    x, ctx, t, v = make_inputs((1, 16, 33, 64, 64), 512, 4096, 112, 0.62)
    start = (40, 17, 64)
    ref = oracle_step(cfg, P, x, ctx, t, v, start)
    m = build(vds, cfg, P)
    fig = check_step(vds, m, x, ctx, t, v, start, ref, GRAD_COS, GRAD_REL)
    parity_log("c4_dit_xl_block", **fig)
    with torch.no_grad():
        o2 = m(x.repeat(2, 1, 1, 1, 1).cuda(), ctx.repeat(2, 1, 1).cuda(), t.repeat(2).cuda(), rope_start=start)
    assert torch.equal(o2[0], o2[1])
