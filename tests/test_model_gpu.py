"""GPU parity of the whole HIP train step (DiT forward, backward, loss, harness, AdamW) against
the committed golden fixtures of the reference and against the CPU oracle on seeded inputs.

Tolerances (bf16 compute, fp32 accumulation; the reference's own bf16 run differs from its fp32
run by `e_ref` = a few 1e-3..1e-2 at these sizes):
    outputs  : rel L2 error vs the fp32 reference <= max(2.5 * e_ref, 1.5e-2)
    loss     : relative error <= 1e-2
    grads    : cosine >= GRAD_COS and rel L2 error <= GRAD_REL per parameter tensor -- about 2x the worst values
               measured on the MI355X (recorded per run in gpurun_out/parity_report.jsonl by `_worst_grad_report`)
"""
import os

import pytest
import torch

from oracle import dit_oracle as O

pytestmark = pytest.mark.gpu
bf16, f32 = torch.bfloat16, torch.float32
GRAD_COS, GRAD_REL = 0.999, 3e-2
_WORST = {"cos": (1.0, None), "rel": (0.0, None)}


def _track(name, c, e):
    if c < _WORST["cos"][0]:
        _WORST["cos"] = (c, name)
    if e > _WORST["rel"][0]:
        _WORST["rel"] = (e, name)


@pytest.fixture(scope="module", autouse=True)
def _worst_grad_report(parity_log):
    yield
    parity_log("test_model_gpu.worst_gradient", worst_cos=_WORST["cos"], worst_rel=_WORST["rel"])
    for name, (wl, wp) in _MARGINS.items():
        parity_log("test_model_gpu.run_to_run." + name, worst_loss_rel=wl, worst_param_rel=wp)


@pytest.fixture(scope="module")
def vds():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import video_diffusion_speedrun_amd as pkg
    from video_diffusion_speedrun_amd import model, ops, optim, train
    return dict(pkg=pkg, model=model, ops=ops, optim=optim, train=train)


def rel(a, b):
    a, b = a.float().cpu().flatten(), b.float().cpu().flatten()
    return ((a - b).norm() / (b.norm() + 1e-20)).item()


_MARGINS = {}


def _note(name, loss_pairs, p1, p0):
    """worst relative loss / parameter difference of a run-to-run comparison, kept for the parity report"""
    wl = max((abs(a - b) / abs(a) for a, b in loss_pairs), default=0.0)
    wp = max((rel(p1[k], p0[k]) for k in p0), default=0.0)
    cur = _MARGINS.get(name, (0.0, 0.0))
    _MARGINS[name] = (max(cur[0], wl), max(cur[1], wp))
    return wl, wp


def cosine(a, b):
    a, b = a.float().cpu().flatten(), b.float().cpu().flatten()
    return (a @ b / (a.norm() * b.norm() + 1e-30)).item()


def build(vds, cfg: O.DiTConfig, P):
    m = vds["model"].DiT(in_channels=cfg.in_channels, patch_size=cfg.patch_size,
                         time_patch_size=cfg.time_patch_size, hidden_size=cfg.hidden_size, depth=cfg.depth,
                         num_heads=cfg.num_heads, mlp_ratio=cfg.mlp_ratio,
                         cross_attn_input_size=cfg.cross_attn_input_size, residual_v=cfg.residual_v,
                         train_bias_and_rms=cfg.train_bias_and_rms)
    missing, unexpected = m.load_state_dict(P, strict=True)
    # same names, same registration order as the reference / oracle table
    assert [n for n, _ in m.named_parameters()] == list(O.param_shapes(cfg).keys())
    return m.to("cuda")


def load_g1(golden_dir, name):
    fx = torch.load(os.path.join(golden_dir, name), weights_only=False)
    cfg = O.DiTConfig(**fx["cfg"])
    P = O.init_params(cfg, **fx["param_init"])
    P.update({k: v.clone() for k, v in fx["param_tweaks"].items()})
    return fx, cfg, P


LAMBDA_ERR = 5e-4  # |got - reference| / sum of |terms|; measured <= 2.5e-4 here (small models), <= 1.3e-4 on the workload tests


def lambda_term_sums(cfg, P, x, ctx, t, start, upstream):
    """sum_j |dv_j (v_raw_j - v_0_j)| per mixed block, from the fp32 oracle on the fixture's inputs: d loss / d lambda_i
    is a scalar made of B*L*D signed bf16 products that largely cancel, so its error is judged against the sum of
    the |products| (v_raw - v_0 = (v_mixed - v_0) / lambda, model.py:129-130).  `upstream(out)` -> scalar to backward."""
    Pg = {k: w.clone().requires_grad_(True) for k, w in P.items()}
    cap = {}
    out = O.dit_forward(Pg, cfg, x.float(), ctx.float(), t.float(), start, cap)
    mixed = {i: cap[f"blocks.{i}.v"] for i in range(1, cfg.depth)}
    for vm in mixed.values():
        vm.retain_grad()
    upstream(out).backward()
    v0 = cap["blocks.0.v"].detach()
    return {f"blocks.{i}.lambda_param": (vm.grad * (vm.detach() - v0) / P[f"blocks.{i}.lambda_param"].item()).abs().sum().item()
            for i, vm in mixed.items()}


def check_lambda(named, ref_grads, l1):
    """every lambda_param gradient against the REFERENCE's value (fixture), error normalised by the block's term sum"""
    rows = []
    for k in ref_grads:
        if not k.endswith("lambda_param") or k not in l1:
            continue
        ref = ref_grads[k]["sample"] if isinstance(ref_grads[k], dict) else ref_grads[k]
        got, want = named[k].grad.float().cpu().reshape(()).item(), ref.float().reshape(()).item()
        rows.append((k, got, want, abs(got - want) / l1[k]))
    assert rows and all(r[3] <= LAMBDA_ERR for r in rows), rows


def check_grad(name, got, ref, report):
    if isinstance(ref, dict):
        g = got.float().cpu().flatten()[:: ref["step"]]
        r = ref["sample"]
    else:
        g, r = got, ref
    c, e = cosine(g, r), rel(g, r)
    report.append((name, c, e))
    _track(name, c, e)
    return c >= GRAD_COS and e <= GRAD_REL


@pytest.mark.parametrize("name", ["g1_tiny_hd64.pt", "g1_tiny_hd72.pt"])
def test_g1_golden_forward_backward(vds, golden_dir, name):
    fx, cfg, P = load_g1(golden_dir, name)
    m = build(vds, cfg, P)
    x, ctx, t = fx["x"].cuda().to(bf16), fx["context"].cuda().to(bf16), fx["t"].cuda().to(bf16)
    out = m(x, ctx, t, rope_start=tuple(fx["rope_start"]))
    ref32, ref16 = fx["fp32"]["out"], fx["bf16"]["out"]
    e_ref = rel(ref16, ref32)
    e = rel(out, ref32)
    assert e <= max(2.5 * e_ref, 1.5e-2), (e, e_ref)
    (out.float() * fx["dout"].cuda()).sum().backward()
    report, bad = [], []
    named = dict(m.named_parameters())
    for k, g in fx["fp32"]["grads"].items():
        p = named[k]
        assert p.grad is not None and p.grad.dtype == f32, k
        if k.endswith("lambda_param"):
            continue
        if not check_grad(k, p.grad, g, report):
            bad.append(report[-1])
    assert not bad, f"gradient mismatches (name, cosine, rel): {bad}"
    l1 = lambda_term_sums(cfg, P, fx["x"].to(bf16), fx["context"].to(bf16), fx["t"].to(bf16), tuple(fx["rope_start"]),
                          lambda o: (o * fx["dout"]).sum())
    check_lambda(named, fx["fp32"]["grads"], l1)


def test_g2_dit_s_config1(vds, golden_dir):
    """BASELINE config 1 shape (DiT-S/2, 4 x [16,8,16,16] latents, 512 x 4096 context)."""
    fx = torch.load(os.path.join(golden_dir, "g2_dit_s_c1.pt"), weights_only=False)
    cfg = O.DiTConfig(**fx["cfg"])
    P = O.init_params(cfg, seed=fx["param_seed"], randomize_zero_init=True, init_std_factor=0.1)
    g = torch.Generator().manual_seed(fx["input_seed"])
    x = torch.randn(4, 16, 8, 16, 16, generator=g)
    ctx = torch.randn(4, 512, 4096, generator=g)
    t = O.time_shift(torch.randn(4, generator=g))
    v = torch.randn(4, 16, 8, 16, 16, generator=g)
    m = build(vds, cfg, P)
    out = m(x.cuda().to(bf16), ctx.cuda().to(bf16), t.cuda().to(bf16), rope_start=tuple(fx["rope_start"]))
    assert rel(out, fx["out"]) <= 2.5e-2
    loss, _ = vds["train"].flow_loss(out, v.cuda().to(bf16))
    assert abs(loss.item() - fx["loss"]) / fx["loss"] <= 1e-2
    loss.backward()
    report, bad = [], []
    named = dict(m.named_parameters())
    for k, d in fx["grad_digest"].items():
        if k.endswith("lambda_param"):
            continue
        if not check_grad(k, named[k].grad, d, report):
            bad.append(report[-1])
    assert not bad, f"gradient mismatches (name, cosine, rel): {bad}"
    l1 = lambda_term_sums(cfg, P, x.to(bf16), ctx.to(bf16), t.to(bf16), tuple(fx["rope_start"]),
                          lambda o: O.flow_loss(v.to(bf16), o)[0])
    check_lambda(named, fx["grad_digest"], l1)


def test_g3_harness(vds, golden_dir):
    """train.py::forward: same generator draws, time shift, noising, loss."""
    fx = torch.load(os.path.join(golden_dir, "g3_harness.pt"), weights_only=False)
    cfg = O.DiTConfig(**fx["cfg"])
    P = O.init_params(cfg, seed=fx["param_seed"], randomize_zero_init=True, init_std_factor=1.0)
    m = build(vds, cfg, P)
    ops = vds["ops"]
    t = O.time_shift(fx["z"])
    assert torch.equal(t, fx["t"])
    zt, v = ops.noise_latents(fx["latent"].to(bf16).cuda(), fx["noise"].cuda(), t.float().cuda())
    assert torch.equal(zt.cpu(), fx["z_t"])
    out = m(zt, fx["context"].cuda(), t.cuda(), rope_start=tuple(fx["rope_start"]))
    assert rel(out, fx["out"]) <= 3e-2
    loss, _ = vds["train"].flow_loss(out, v)
    assert abs(loss.item() - fx["loss"]) / fx["loss"] <= 1e-2
    # the harness entry point itself, with a seeded device generator
    gen = torch.Generator(device="cuda").manual_seed(5)
    batch = {"latent": fx["latent"], "context": fx["context"], "prompt": [""] * 3}
    total, diff = vds["train"].forward(m, batch, None, None, "cuda", 0, False, generator=gen)
    assert torch.isfinite(total) and total.item() > 0
    total.backward()
    assert all(p.grad is not None for p in m.parameters())


@pytest.mark.parametrize("hd,H,train_bias", [(64, 4, False), (72, 2, True), (128, 2, False)])
def test_oracle_parity_random(vds, hd, H, train_bias):
    """a mid-size model vs the fp32 CPU oracle on seeded inputs, ragged token count"""
    cfg = O.DiTConfig(in_channels=16, patch_size=2, time_patch_size=1, hidden_size=hd * H, depth=3, num_heads=H,
                      cross_attn_input_size=256, residual_v=True, train_bias_and_rms=train_bias)
    P = O.init_params(cfg, seed=11, randomize_zero_init=True, init_std_factor=0.5)
    g = torch.Generator().manual_seed(12)
    B = 2
    x = torch.randn(B, 16, 3, 12, 10, generator=g).to(bf16)
    ctx = torch.randn(B, 77, 256, generator=g).to(bf16)
    t = O.time_shift(torch.randn(B, generator=g)).to(bf16)
    v = torch.randn(B, 16, 3, 12, 10, generator=g).to(bf16)
    start = (5, 9, 17)
    Pg = {k: w.clone().requires_grad_(True) for k, w in P.items()}
    o_ref = O.dit_forward(Pg, cfg, x.float(), ctx.float(), t.float(), start)
    l_ref, _ = O.flow_loss(v, o_ref)
    l_ref.backward()
    Pb = {k: w.to(bf16) for k, w in P.items()}
    o_bf = O.dit_forward(Pb, cfg, x, ctx, t, start)
    e_ref = rel(o_bf, o_ref)
    m = build(vds, cfg, P)
    out = m(x.cuda(), ctx.cuda(), t.cuda(), rope_start=start)
    e = rel(out, o_ref)
    assert e <= max(2.5 * e_ref, 1.5e-2), (e, e_ref)
    loss, _ = vds["train"].flow_loss(out, v.cuda())
    assert abs(loss.item() - l_ref.item()) / l_ref.item() <= 1e-2
    loss.backward()
    bad = []
    for k, p in m.named_parameters():
        if Pg[k].grad is None or k.endswith("lambda_param"):
            continue
        c, e = cosine(p.grad, Pg[k].grad), rel(p.grad, Pg[k].grad)
        _track(k, c, e)
        if not (c >= GRAD_COS and e <= GRAD_REL):
            bad.append((k, c, e))
    assert not bad, bad
    l1 = lambda_term_sums(cfg, P, x, ctx, t, start, lambda o: O.flow_loss(v, o)[0])
    check_lambda(dict(m.named_parameters()), {k: w.grad for k, w in Pg.items() if w.grad is not None}, l1)


def test_muadamw_matches_reference(vds, golden_dir):
    """two AdamW steps on fixed gradients vs torch.optim.AdamW values recorded from the reference setup"""
    fx = torch.load(os.path.join(golden_dir, "g4_optim.pt"), weights_only=False)["adamw"]
    cfg = O.DiTConfig(**fx["cfg"])
    consts = ["patch_proj", "context_kv", "positional_embedding"]
    table = O.mup_settings(O.param_shapes(cfg), fx["lr"], fx["wd"], consts)
    ps = {k: torch.nn.Parameter(v.clone().cuda()) for k, v in fx["p0"].items()}
    groups = [{"params": [p], "lr": table[k]["lr"], "weight_decay": table[k]["wd"]} for k, p in ps.items()]
    opt = vds["optim"].MuAdamW(groups, betas=(0.95, 0.99))
    for s in range(2):
        mult = O.lr_lambda(s, "cosine", 20, 1000)
        for gi, (k, p) in enumerate(ps.items()):
            p.grad = fx["grads"][s][k].clone().cuda()
            opt.param_groups[gi]["lr"] = table[k]["lr"] * mult
        opt.step()
    for k, p in ps.items():
        assert rel(p.data, fx["p2"][k]) < 2e-6, k


def test_train_steps_decrease_loss_and_mup_groups(vds):
    """end-to-end: get_mup_setup -> MuAdamW -> 3 steps on a fixed batch lower the loss; bf16 shadow
    written by the optimizer equals a fresh cast of the fp32 master."""
    cfg = O.DiTConfig(in_channels=16, hidden_size=128, depth=2, num_heads=2, cross_attn_input_size=64,
                      residual_v=True, train_bias_and_rms=False)
    P = O.init_params(cfg, seed=21, randomize_zero_init=True, init_std_factor=1.0)
    m = build(vds, cfg, P)
    groups, settings = m.get_mup_setup(3e-3, 0.1, ["patch_proj", "context_kv", "positional_embedding"])
    ref = O.mup_settings(O.param_shapes(cfg), 3e-3, 0.1, ["patch_proj", "context_kv", "positional_embedding"])
    assert {k: (v["lr"], v["wd"]) for k, v in settings.items()} == {k: (v["lr"], v["wd"]) for k, v in ref.items()}
    opt = vds["optim"].MuAdamW(groups, betas=(0.95, 0.99))
    g = torch.Generator().manual_seed(3)
    batch = {"latent": torch.randn(2, 16, 4, 8, 8, generator=g), "context": torch.randn(2, 16, 64, generator=g),
             "prompt": ["", ""]}
    losses = []
    for s in range(4):
        gen = torch.Generator(device="cuda").manual_seed(100)
        torch.manual_seed(0)
        loss = vds["train"].train_step(m, opt, None, batch, "cuda", generator=gen, rope_start=(1, 2, 3))
        losses.append(loss.item())
    assert losses[-1] < losses[0], losses
    for grp in m._groups:
        assert torch.equal(grp.shadow.cpu(), grp.master.to(bf16).cpu())


def test_full_size_properties_dit_xl_block(vds):
    """BASELINE headline shape (DiT-XL width, 8192+16 tokens) on a 2-block model: duplicated samples
    give identical outputs, and the batch-mean gradient of two identical samples equals the
    single-sample gradient; zero-initialised final_proj gives output == 0 exactly."""
    cfg = O.DiTConfig(in_channels=16, patch_size=2, time_patch_size=2, hidden_size=1152, depth=2, num_heads=16,
                      cross_attn_input_size=4096, residual_v=True, train_bias_and_rms=False)
    P = O.init_params(cfg, seed=5, randomize_zero_init=True, init_std_factor=0.1)
    m = build(vds, cfg, P)
    g = torch.Generator().manual_seed(6)
    x1 = torch.randn(1, 16, 16, 64, 64, generator=g).to(bf16).cuda()
    c1 = torch.randn(1, 512, 4096, generator=g).to(bf16).cuda()
    t1 = torch.tensor([0.6]).to(bf16).cuda()
    v1 = torch.randn(1, 16, 16, 64, 64, generator=g).to(bf16).cuda()
    start = (3, 4, 5)
    o1 = m(x1, c1, t1, rope_start=start)
    l1, _ = vds["train"].flow_loss(o1, v1)
    l1.backward()
    g1 = {k: p.grad.clone() for k, p in m.named_parameters()}
    m.zero_grad()  # a second backward without it accumulates into .grad, like autograd
    o2 = m(x1.repeat(2, 1, 1, 1, 1), c1.repeat(2, 1, 1), t1.repeat(2), rope_start=start)
    assert torch.equal(o2[0], o2[1]) and torch.equal(o2[0], o1[0])
    l2, _ = vds["train"].flow_loss(o2, v1.repeat(2, 1, 1, 1, 1))
    # the loss is a fixed-order two-stage reduction (train.py:121-125): per-sample means of identical samples are
    # identical words and (p + p) / 2 == p, so the batch loss equals the single-sample loss TO THE BIT, run to run
    assert l2.item() == l1.item(), (l2.item(), l1.item())
    l1_again, _ = vds["train"].flow_loss(o1, v1)
    assert l1_again.item() == l1.item()
    l2.backward()
    for k, p in m.named_parameters():
        if g1[k].abs().max() > 0:
            assert cosine(p.grad, g1[k]) > 0.999 and rel(p.grad, g1[k]) < 2e-2, k
    with torch.no_grad():
        m.final_proj.weight.zero_()
        m.final_proj.bias.zero_()
        o0 = m(x1, c1, t1, rope_start=start)
    assert o0.abs().max().item() == 0


@pytest.mark.parametrize("shape", ["headline_width", "small_with_weights_and_biases"])
def test_deterministic_mode_gives_bit_identical_gradients(vds, shape, request):
    """vds_set_deterministic(1): every fp32-atomic accumulation of the backward pass (split-K weight gradients, modulation /
    bias / RMSNorm-weight column sums, lambda gradients, the adaLN fan-in) becomes a fixed-order reduction, like autograd's
    reductions behind the reference's train.py:431-433 -- two backward passes over the same inputs give the SAME BITS in
    every gradient, at DiT-XL width with 8192+16 tokens (real split counts) and on a small model with norm weights and
    biases; the values agree with the default mode to fp32 rounding."""
    ops = vds["ops"]
    if shape == "headline_width":
        cfg = O.DiTConfig(in_channels=16, patch_size=2, time_patch_size=2, hidden_size=1152, depth=2, num_heads=16,
                          cross_attn_input_size=4096, residual_v=True, train_bias_and_rms=False)
        P = O.init_params(cfg, seed=5, randomize_zero_init=True, init_std_factor=0.1)
        lat, Lc, B = (16, 16, 64, 64), 512, 2
    else:
        cfg = O.DiTConfig(in_channels=16, hidden_size=144, depth=3, num_heads=2, cross_attn_input_size=64,
                          residual_v=True, train_bias_and_rms=True)
        P = O.init_params(cfg, seed=7, randomize_zero_init=True, init_std_factor=1.0)
        lat, Lc, B = (16, 8, 16, 16), 24, 3
    m = build(vds, cfg, P)
    g = torch.Generator().manual_seed(6)
    x = torch.randn(B, *lat, generator=g).to(bf16).cuda()
    c = torch.randn(B, Lc, cfg.cross_attn_input_size, generator=g).to(bf16).cuda()
    t = torch.rand(B, generator=g).to(bf16).cuda()
    v = torch.randn(B, *lat, generator=g).to(bf16).cuda()

    def grads():
        m.zero_grad()
        out = m(x, c, t, rope_start=(3, 4, 5))
        loss, _ = vds["train"].flow_loss(out, v)
        loss.backward()
        torch.cuda.synchronize()
        return loss.item(), {k: p.grad.clone() for k, p in m.named_parameters()}

    l_atomic, g_atomic = grads()
    ops.set_deterministic(True, 1 << 30)
    request.addfinalizer(lambda: ops.set_deterministic(False))
    l1, g1 = grads()
    l2, g2 = grads()
    assert l1 == l2 == l_atomic
    for k in g1:
        assert torch.equal(g1[k], g2[k]), k
        if float(g_atomic[k].abs().max()) > 0:  # (lambda gradients: sums of cancelling terms, the atomic order shows)
            assert rel(g1[k], g_atomic[k]) <= (5e-3 if k.endswith("lambda_param") else 2e-5), (k, rel(g1[k], g_atomic[k]))


@pytest.mark.parametrize("det", [False, True], ids=["atomics", "deterministic"])
def test_shard_runtime_on_one_gpu_matches_unsharded(vds, det, monkeypatch, request):
    """the stream / event / RCCL choreography of fsdp.ShardRuntime, forced on at world_size 1
    (1-rank nccl group): three optimizer steps give the same losses and parameters as the
    unsharded model (same kernels, same order; the collectives are exact copies at W=1; the only
    difference allowed is the summation order of the fp32 atomic accumulations) -- and in deterministic mode
    (fixed-order reductions; the unsharded model on the per-block adaLN form the sharded one uses, so that both run the
    same kernel sequence) THE SAME BITS: resident copies, the reshard_after_forward ring and no sharding at all are then
    indistinguishable in every loss and every parameter."""
    import torch.distributed as dist
    from video_diffusion_speedrun_amd.fsdp import apply_fsdp
    if det:
        monkeypatch.setenv("VDS_ADALN_BATCH", "0")
        vds["ops"].set_deterministic(True, 256 << 20)
        request.addfinalizer(lambda: vds["ops"].set_deterministic(False))
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        cfg = O.DiTConfig(in_channels=16, hidden_size=128, depth=3, num_heads=2, cross_attn_input_size=64,
                          residual_v=True, train_bias_and_rms=False)
        P = O.init_params(cfg, seed=31, randomize_zero_init=True, init_std_factor=1.0)
        g = torch.Generator().manual_seed(3)
        batch = {"latent": torch.randn(2, 16, 4, 8, 8, generator=g), "context": torch.randn(2, 16, 64, generator=g),
                 "prompt": ["", ""]}
        results = []
        for sharded in (False, True, "reshard_after_forward"):
            m = build(vds, cfg, P)
            if sharded:
                m = apply_fsdp(m, torch.bfloat16, torch.float32, force_runtime=True, reshard_after_forward=sharded != True)
                assert m._fsdp is not None
            groups, _ = m.get_mup_setup(3e-3, 0.1, ["patch_proj", "context_kv", "positional_embedding"])
            opt = vds["optim"].MuAdamW(groups, betas=(0.95, 0.99))
            losses = []
            for s in range(3):
                gen = torch.Generator(device="cuda").manual_seed(100 + s)
                torch.manual_seed(0)
                loss = vds["train"].train_step(m, opt, None, batch, "cuda", generator=gen, rope_start=(1, 2, 3))
                losses.append(loss.item())
            torch.cuda.synchronize()
            if sharded:  # (the memory-bounded mode gathers every block but the last a second time, in backward)
                extra = cfg.depth - 1 if sharded != True else 0
                assert m._fsdp.n_all_gather == 3 * (1 + cfg.depth + extra) and m._fsdp.n_reduce_scatter == 3 * (1 + cfg.depth)
            results.append((losses, {k: v.clone() for k, v in m.full_state_dict().items()}))
        (l0, p0) = results[0]
        # fp32 atomic accumulation order differs from run to run
        for l1, p1 in results[1:]:
            if det:
                assert l1 == l0, (l0, l1)
                for k in p0:
                    assert torch.equal(p1[k], p0[k]), k
                continue
            _note("shard_runtime_w1", zip(l0, l1), p1, p0)
            # (measured over nine runs on three boxes: losses equal, parameters <= 8.5e-8: profiles/r06/parity_report.jsonl)
            assert all(abs(a - b) <= 1e-6 * abs(a) for a, b in zip(l0, l1)), (l0, l1)
            for k in p0:
                assert rel(p1[k], p0[k]) <= 2e-6, (k, rel(p1[k], p0[k]))
        from video_diffusion_speedrun_amd import comm
        assert comm.info()["active"] and comm.info()["world"] == 1  # the collectives went through vds_comm_*
    finally:
        from video_diffusion_speedrun_amd import comm
        comm.destroy()
        dist.destroy_process_group()


def test_g5_sampler_matches_reference(vds, golden_dir):
    """Euler + CFG sampler (sampling/sample.py::generate_image) on the HIP forward path vs the
    reference's latents; also the batched cond/uncond forward vs two separate forwards."""
    from video_diffusion_speedrun_amd.sampling import generate_latents, shifted_times
    fx = torch.load(os.path.join(golden_dir, "g5_sampler.pt"), weights_only=False)
    cfg = O.DiTConfig(**fx["cfg"])
    P = O.init_params(cfg, seed=fx["param_seed"], randomize_zero_init=True, init_std_factor=1.0)
    m = build(vds, cfg, P)
    ctx = fx["context"]
    r16, r32 = fx["bf16"], fx["fp32"]
    acc = generate_latents(m, ctx, None, fx["steps"], fx["cfg_scale"], latents=r16["latents0"],
                           rope_starts=r16["rope_starts"])
    assert acc.dtype == f32 and tuple(acc.shape) == (1, 16, 16, 8, 8)
    e_ref = rel(r16["out"], r32["out"])          # the reference's own bf16-vs-fp32 gap
    e = rel(acc.squeeze(0), r32["out"])
    assert e <= max(2.5 * e_ref, 1.5e-2), (e, e_ref)
    # batched guidance: same offsets for both calls of a step -> ONE B=2 forward per step
    shared = [s for s in r16["rope_starts"][0::2] for _ in (0, 1)]
    a2 = generate_latents(m, ctx, None, fx["steps"], fx["cfg_scale"], latents=r16["latents0"], rope_starts=shared)
    ref2 = O.sample_euler_cfg(P, cfg, r16["latents0"], ctx, torch.zeros_like(ctx), fx["steps"], fx["cfg_scale"], shared,
                              dtype=torch.float32)
    assert rel(a2, ref2) <= 1.5e-2
    # default call path: random offsets, seeded noise, reference latent shape for 64x64
    a3 = generate_latents(m, ctx, inference_steps=2, cfg_scale=6.0, height=64, width=64, seed=1)
    assert torch.isfinite(a3).all() and tuple(a3.shape) == (1, 16, 16, 8, 8)
    t, tn = shifted_times(2, 2)
    assert abs(t - 1.0) < 1e-12 and 0 < tn < 1


@pytest.mark.parametrize("det", [False, True], ids=["atomics", "deterministic"])
def test_checkpoint_resume_is_exact(vds, tmp_path, det, request):
    """save after 2 steps (weights + AdamW shard + step), resume in a fresh model / optimizer: the
    third step's loss and the final weights equal those of the uninterrupted run -- up to the fp32 atomic
    summation order by default, TO THE BIT in deterministic mode (vds_set_deterministic: fixed-order reductions)"""
    from video_diffusion_speedrun_amd import checkpoint as ck
    if det:
        vds["ops"].set_deterministic(True, 256 << 20)
        request.addfinalizer(lambda: vds["ops"].set_deterministic(False))
    cfg = O.DiTConfig(in_channels=16, hidden_size=128, depth=2, num_heads=2, cross_attn_input_size=64,
                      residual_v=True, train_bias_and_rms=False)
    P = O.init_params(cfg, seed=41, randomize_zero_init=True, init_std_factor=1.0)
    g = torch.Generator().manual_seed(3)
    batch = {"latent": torch.randn(2, 16, 4, 8, 8, generator=g), "context": torch.randn(2, 16, 64, generator=g),
             "prompt": ["", ""]}
    consts = ["patch_proj", "context_kv", "positional_embedding"]

    def fresh(Pinit):
        m = build(vds, cfg, Pinit)
        groups, _ = m.get_mup_setup(3e-3, 0.1, consts)
        return m, vds["optim"].MuAdamW(groups, betas=(0.95, 0.99))

    def step(m, opt, s):
        gen = torch.Generator(device="cuda").manual_seed(100 + s)
        torch.manual_seed(0)
        return vds["train"].train_step(m, opt, None, batch, "cuda", generator=gen, rope_start=(1, 2, 3)).item()

    m, opt = fresh(P)
    for s in range(2):
        step(m, opt, s)
    ck.save_checkpoint(str(tmp_path / "c"), m, opt, step=2)
    l3 = step(m, opt, 2)
    ref = m.full_state_dict()
    P2 = O.init_params(cfg, seed=42, randomize_zero_init=True, init_std_factor=1.0)  # different weights
    m2, opt2 = fresh(P2)
    step(m2, opt2, 0)  # materialise the flat groups and the optimizer state before loading into them
    assert ck.load_checkpoint(str(tmp_path / "c"), m2, opt2) == 2
    l3b = step(m2, opt2, 2)
    got = m2.full_state_dict()
    if det:
        assert l3 == l3b, (l3, l3b)
        for k in ref:
            assert torch.equal(got[k], ref[k]), k
        return
    _note("checkpoint_resume", [(l3, l3b)], got, ref)
    assert abs(l3 - l3b) <= 1e-6 * abs(l3), (l3, l3b)
    for k in ref:  # the restored state is bit-exact; the step after it differs by the fp32 atomic summation order only
        assert rel(got[k], ref[k]) <= 5e-7, k  # (measured 5e-9 .. 6e-9)


def test_device_prefetcher_feeds_the_train_step(vds):
    """data path: serialized rows -> DataLoader -> pinned staging + side-stream H2D ->
    the HIP train step; batches arrive in order, in bf16, on the GPU, bit-equal to the rows"""
    from video_diffusion_speedrun_amd import data as D
    g = torch.Generator().manual_seed(0)
    lat = [torch.randn(16, 4, 8, 8, generator=g).to(bf16) for _ in range(6)]
    rows = [{"serialized_latent": D.serialize_tensor(t), "caption": f"c{i}"} for i, t in enumerate(lat)]
    dl = D.create_dataloader("train", 2, 0, False, dataset=D.LatentDataset(rows=rows))
    cfg = O.DiTConfig(in_channels=16, hidden_size=128, depth=1, num_heads=2, cross_attn_input_size=64,
                      residual_v=True, train_bias_and_rms=False)
    m = build(vds, cfg, O.init_params(cfg, seed=51, randomize_zero_init=True, init_std_factor=1.0))
    groups, _ = m.get_mup_setup(1e-3, 0.1, [])
    opt = vds["optim"].MuAdamW(groups, betas=(0.95, 0.99))
    seen = 0
    for i, batch in enumerate(D.DevicePrefetcher(dl, "cuda")):
        assert batch["latent"].is_cuda and batch["latent"].dtype == bf16 and batch["prompt"] == [f"c{2*i}", f"c{2*i+1}"]
        assert torch.equal(batch["latent"].cpu(), torch.stack(lat[2 * i:2 * i + 2]))
        batch["context"] = torch.randn(2, 8, 64, generator=g)
        loss = vds["train"].train_step(m, opt, None, batch, "cuda")
        assert torch.isfinite(loss)
        seen += 1
    assert seen == 3


def test_oracle_parity_long_sequence_hd72(vds):
    """2048+16 tokens at head_dim 72: the sequence length at which the model switches to the wide
    forward attention kernel with the ones-column contract -- forward, loss and gradients vs the oracle,
    then a few optimizer steps must lower the loss"""
    cfg = O.DiTConfig(in_channels=16, patch_size=2, time_patch_size=2, hidden_size=144, depth=2, num_heads=2,
                      cross_attn_input_size=64, residual_v=True, train_bias_and_rms=False)
    P = O.init_params(cfg, seed=61, randomize_zero_init=True, init_std_factor=1.0)
    g = torch.Generator().manual_seed(62)
    B = 1
    x = torch.randn(B, 16, 16, 32, 32, generator=g).to(bf16)
    ctx = torch.randn(B, 40, 64, generator=g).to(bf16)
    t = O.time_shift(torch.randn(B, generator=g)).to(bf16)
    v = torch.randn(B, 16, 16, 32, 32, generator=g).to(bf16)
    start = (7, 3, 11)
    Pg = {k: w.clone().requires_grad_(True) for k, w in P.items()}
    o_ref = O.dit_forward(Pg, cfg, x.float(), ctx.float(), t.float(), start)
    l_ref, _ = O.flow_loss(v, o_ref)
    l_ref.backward()
    o_bf = O.dit_forward({k: w.to(bf16) for k, w in P.items()}, cfg, x, ctx, t, start)
    e_ref = rel(o_bf, o_ref)
    m = build(vds, cfg, P)
    out = m(x.cuda(), ctx.cuda(), t.cuda(), rope_start=start)
    e = rel(out, o_ref)
    assert e <= max(2.5 * e_ref, 1.5e-2), (e, e_ref)
    loss, _ = vds["train"].flow_loss(out, v.cuda())
    assert abs(loss.item() - l_ref.item()) / l_ref.item() <= 1e-2
    loss.backward()
    bad = []
    for k, p in m.named_parameters():
        if Pg[k].grad is None or k.endswith("lambda_param"):
            continue
        c, e = cosine(p.grad, Pg[k].grad), rel(p.grad, Pg[k].grad)
        _track(k, c, e)
        if not (c >= GRAD_COS and e <= GRAD_REL):
            bad.append((k, c, e))
    assert not bad, bad
    groups, _ = m.get_mup_setup(3e-3, 0.1, ["patch_proj", "context_kv", "positional_embedding"])
    opt = vds["optim"].MuAdamW(groups, betas=(0.95, 0.99))
    batch = {"latent": x.float(), "context": ctx, "prompt": [""]}
    losses = []
    for s in range(4):
        gen = torch.Generator(device="cuda").manual_seed(7)
        torch.manual_seed(0)
        losses.append(vds["train"].train_step(m, opt, None, batch, "cuda", generator=gen, rope_start=start).item())
    assert all(map(lambda z: z == z, losses)) and losses[-1] < losses[0], losses


def test_graph_replay_matches_eager_steps(vds):
    """graph.GraphedTrainStep (whole-step HIP-graph replay): two eager steps, capture, four
    replays give the same losses and parameters as six eager steps from the same seeds -- same kernels
    in the same order; the RoPE offsets, AdamW scalars and LR multiplier reach the replays through
    device memory, the z / noise / caption-dropout draws through torch's graph-safe device generator.
    Allowed difference: summation order of fp32 atomics and the last-ulp of the host-side powf (AdamW's
    m / sqrt(v) turns a 1e-7 gradient difference into up to ~1e-5 of a small parameter tensor per step)."""
    from video_diffusion_speedrun_amd.graph import GraphedTrainStep
    cfg = O.DiTConfig(in_channels=16, hidden_size=128, depth=3, num_heads=2, cross_attn_input_size=64,
                      residual_v=True, train_bias_and_rms=False)
    P = O.init_params(cfg, seed=41, randomize_zero_init=True, init_std_factor=1.0)
    g = torch.Generator().manual_seed(5)
    batches = [{"latent": torch.randn(2, 16, 4, 8, 8, generator=g).cuda(), "context": torch.randn(2, 16, 64, generator=g).cuda(),
                "prompt": ["", ""]} for _ in range(6)]
    results = []
    for graphed in (False, True):
        m = build(vds, cfg, P)
        groups, _ = m.get_mup_setup(3e-3, 0.1, ["patch_proj", "context_kv", "positional_embedding"])
        opt = vds["optim"].MuAdamW(groups, betas=(0.95, 0.99))
        sched = vds["train"].get_schedule(opt, "cosine", 3, 50)
        torch.manual_seed(77)  # CPU RNG (RoPE offsets) and the default device generator (z, noise, dropout)
        torch.cuda.manual_seed(77)
        losses = []
        if graphed:
            gs = GraphedTrainStep(m, opt, sched, "cuda", eager_steps=2)
            for b in batches:
                losses.append(gs.step(b).item())
            assert gs.n_replays == 4 and opt._step == 6
        else:
            for b in batches:
                losses.append(vds["train"].train_step(m, opt, sched, b, "cuda").item())
        torch.cuda.synchronize()
        results.append((losses, {k: v.clone() for k, v in m.full_state_dict().items()}))
    (l0, p0), (l1, p1) = results
    assert len(set(l0)) == 6  # different batches / draws every step
    _note("graph_replay", zip(l0, l1), p1, p0)
    assert all(abs(a - b) <= 1e-4 * abs(a) for a, b in zip(l0, l1)), (l0, l1)
    for k in p0:
        assert rel(p1[k], p0[k]) <= 2e-4, (k, rel(p1[k], p0[k]))


def test_graph_replay_with_the_sharding_runtime(vds):
    """SURVEY 8 f-4 at W > 1 (the reference compiles the FSDP-wrapped model, train.py:323-329): the whole step INCLUDING
    the sharding runtime's communication stream, per-group events and RCCL all-gathers / reduce-scatters is captured
    and replayed.  One GPU, runtime forced on over a 1-rank RCCL group: 2 eager steps + 4 replays give the losses and
    parameters of 6 eager steps of the unsharded model."""
    import torch.distributed as dist
    from video_diffusion_speedrun_amd.fsdp import apply_fsdp
    from video_diffusion_speedrun_amd.graph import GraphedTrainStep
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    cfg = O.DiTConfig(in_channels=16, hidden_size=128, depth=3, num_heads=2, cross_attn_input_size=64,
                      residual_v=True, train_bias_and_rms=False)
    P = O.init_params(cfg, seed=41, randomize_zero_init=True, init_std_factor=1.0)
    g = torch.Generator().manual_seed(5)
    batches = [{"latent": torch.randn(2, 16, 4, 8, 8, generator=g).cuda(), "context": torch.randn(2, 16, 64, generator=g).cuda(),
                "prompt": ["", ""]} for _ in range(6)]
    results = []
    for graphed in (False, True):
        m = build(vds, cfg, P)
        if graphed:
            m = apply_fsdp(m, torch.bfloat16, torch.float32, force_runtime=True)
            assert m._fsdp is not None
        groups, _ = m.get_mup_setup(3e-3, 0.1, ["patch_proj", "context_kv", "positional_embedding"])
        opt = vds["optim"].MuAdamW(groups, betas=(0.95, 0.99))
        sched = vds["train"].get_schedule(opt, "cosine", 3, 50)
        torch.manual_seed(77)
        torch.cuda.manual_seed(77)
        losses = []
        if graphed:
            gs = GraphedTrainStep(m, opt, sched, "cuda", eager_steps=2)
            for b in batches:
                losses.append(gs.step(b).item())
            assert gs.n_replays == 4 and opt._step == 6
            assert m._fsdp.n_all_gather == 6 * (1 + cfg.depth) == m._fsdp.n_reduce_scatter
        else:
            for b in batches:
                losses.append(vds["train"].train_step(m, opt, sched, b, "cuda").item())
        torch.cuda.synchronize()
        results.append((losses, {k: v.clone() for k, v in m.full_state_dict().items()}))
    (l0, p0), (l1, p1) = results
    _note("graph_replay_shard_runtime", zip(l0, l1), p1, p0)
    assert all(abs(a - b) <= 1e-4 * abs(a) for a, b in zip(l0, l1)), (l0, l1)
    for k in p0:
        assert rel(p1[k], p0[k]) <= 2e-4, (k, rel(p1[k], p0[k]))


@pytest.mark.parametrize("reshard", [False, True], ids=["resident", "reshard_after_forward"])
def test_graph_replay_with_two_emulated_ranks(vds, monkeypatch, reshard, parity_log):
    """VERDICT r5 item 8 / missing 3 (the reference compiles the FSDP-wrapped model, train.py:323-329): whole-step capture and
    replay of a model sharded over MORE than one rank.  Two replicas of one model in this process, each rank 0 / 1 of a
    world of 2 with the real shard layout, communication stream, per-group events, separate gathered and reduced buffers
    and sharded optimizer -- resident copies or the reshard_after_forward ring -- and each behind its own
    GraphedTrainStep; only the two collectives are an in-process exchange (RCCL refuses two ranks on one device).  Both
    ranks see the same batches and noise, so the average of their gradients is either one's and the all-gather can take the
    other rank's shard as it stood at the start of the step: 2 eager steps + 3 replays must give the losses and parameters
    of 5 eager steps of the unsharded model."""
    from video_diffusion_speedrun_amd import params as PM
    from video_diffusion_speedrun_amd.fsdp import ReshardRuntime, apply_fsdp
    from video_diffusion_speedrun_amd.graph import GraphedTrainStep
    W = 2
    cfg = O.DiTConfig(in_channels=16, hidden_size=128, depth=4, num_heads=2, cross_attn_input_size=64,
                      residual_v=True, train_bias_and_rms=False)
    P = O.init_params(cfg, seed=43, randomize_zero_init=True, init_std_factor=1.0)
    g = torch.Generator().manual_seed(7)
    batches = [{"latent": torch.randn(2, 16, 4, 8, 8, generator=g).cuda(), "context": torch.randn(2, 16, 64, generator=g).cuda(),
                "prompt": ["", ""]} for _ in range(5)]
    reps, snaps = [], {}

    def find(buf, attr):
        for r, m in enumerate(reps):
            for gi, grp in enumerate(m._groups):
                t = getattr(grp, attr)
                if t is not None and t.data_ptr() == buf.data_ptr():
                    return r, gi
        raise AssertionError("collective on an unknown buffer")

    def fake_all_gather(out, inp, group=None):
        r, gi = find(inp, "shadow")
        if (r, gi) not in snaps:
            snaps[(r, gi)] = torch.empty_like(inp)
        snaps[(r, gi)].copy_(inp)  # this rank's shard as it stands at the start of its step
        parts = out.view(W, -1)
        for r2, m2 in enumerate(reps):  # ranks run one after the other: an earlier rank has already stepped
            parts[r2].copy_(snaps[(r2, gi)] if r2 <= r else m2._groups[gi].master.to(bf16))

    def fake_reduce_scatter(out, inp, group=None):
        r, _ = find(inp, "gfull")
        out.copy_(inp.view(W, -1)[r])  # the average over ranks with identical gradients

    monkeypatch.setattr(PM, "all_gather_flat", fake_all_gather)
    monkeypatch.setattr(PM, "reduce_scatter_avg", fake_reduce_scatter)
    consts = ["patch_proj", "context_kv", "positional_embedding"]
    ref = build(vds, cfg, P)
    opt_ref = vds["optim"].MuAdamW(ref.get_mup_setup(3e-3, 0.1, consts)[0], betas=(0.95, 0.99))
    sched_ref = vds["train"].get_schedule(opt_ref, "cosine", 3, 50)
    steps = []
    for r in range(W):
        m = apply_fsdp(build(vds, cfg, P), torch.bfloat16, torch.float32, process_group=object(), world_rank=(W, r),
                       reshard_after_forward=reshard)
        assert isinstance(m._fsdp, ReshardRuntime) == reshard and m._fsdp.world == W
        reps.append(m)
    for m in reps:
        opt = vds["optim"].MuAdamW(m.get_mup_setup(3e-3, 0.1, consts)[0], betas=(0.95, 0.99))
        steps.append(GraphedTrainStep(m, opt, vds["train"].get_schedule(opt, "cosine", 3, 50), "cuda", eager_steps=2))
    l_ref, l_rep = [], [[] for _ in range(W)]
    for k, b in enumerate(batches):
        torch.manual_seed(500 + k)
        torch.cuda.manual_seed(500 + k)
        l_ref.append(vds["train"].train_step(ref, opt_ref, sched_ref, b, "cuda").item())
        for r in range(W):
            torch.manual_seed(500 + k)       # RoPE offsets (global CPU RNG) ...
            torch.cuda.manual_seed(500 + k)  # ... and z / noise draws (device generator; a replay takes the current state)
            l_rep[r].append(steps[r].step(b).item())
    torch.cuda.synchronize()
    assert all(s_.n_replays == 3 for s_ in steps)
    n_ag = (1 + cfg.depth + (cfg.depth - 1 if reshard else 0)) * len(batches)
    assert all(m._fsdp.n_all_gather == n_ag and m._fsdp.n_reduce_scatter == (1 + cfg.depth) * len(batches) for m in reps)
    for r in range(W):
        assert all(abs(a - b_) <= 1e-6 * abs(a) for a, b_ in zip(l_ref, l_rep[r])), (r, l_ref, l_rep[r])
    want = ref.full_state_dict()
    worst = 0.0
    for gi, grp0 in enumerate(reps[0]._groups):
        flat = torch.cat([m._groups[gi].master for m in reps])
        for n in grp0.names:
            o0 = grp0.offsets[n]
            got = flat[o0:o0 + want[n].numel()].view(want[n].shape)
            worst = max(worst, rel(got, want[n]))
            assert rel(got, want[n]) <= 1e-6, (n, rel(got, want[n]))  # (measured 2e-8: profiles/r06/parity_report.jsonl)
    parity_log("graph_replay_two_emulated_ranks", reshard=reshard, worst_param_rel=worst,
               worst_loss_rel=max(abs(a - b_) / abs(a) for r in range(W) for a, b_ in zip(l_ref, l_rep[r])))


def test_graph_replay_with_fp8_keeps_rolling_the_amax_history(vds):
    """under replay every row of the delayed-scaling table must keep receiving its current amax (the roll
    with the fold of the producers' partial maxima is part of the captured step)"""
    from video_diffusion_speedrun_amd.graph import GraphedTrainStep
    cfg = O.DiTConfig(in_channels=16, hidden_size=144, depth=2, num_heads=2, cross_attn_input_size=64,
                      residual_v=True, train_bias_and_rms=False)
    m = build(vds, cfg, O.init_params(cfg, seed=43, randomize_zero_init=True, init_std_factor=1.0)).enable_fp8()
    groups, _ = m.get_mup_setup(3e-2, 0.1, ["patch_proj", "context_kv", "positional_embedding"])
    opt = vds["optim"].MuAdamW(groups, betas=(0.95, 0.99))
    gs = GraphedTrainStep(m, opt, None, "cuda", eager_steps=1)
    g = torch.Generator().manual_seed(6)
    seen = []
    for s in range(6):
        b = {"latent": (torch.randn(2, 16, 4, 8, 8, generator=g) * (1.0 + s)).cuda(),
             "context": torch.randn(2, 16, 64, generator=g).cuda()}
        loss = gs.step(b)
        assert torch.isfinite(loss)
        torch.cuda.synchronize()
        seen.append(m._fp8_hist.tab[:, 0].clone())
    assert gs.n_replays >= 3  # (fp8 delays the capture until the history is armed)
    used = seen[-1] > 0
    assert int(used.sum()) >= 10 * cfg.depth
    # the inputs grow from step to step: every used row's scale source must have moved during the replays (a bf16
    # maximum can repeat by chance between two given steps: any change over the replayed steps counts).  Exempt: the
    # weight rows (round 4: weights use delayed scaling too) of parameters the muP table barely trains -- context_kv is
    # a constant class (lr x 0.01, train.py:287), its bf16 amax legitimately stands still -- they must only be recorded.
    from video_diffusion_speedrun_amd import fp8 as F8
    moved = (seen[-1] != seen[-3]) | (seen[-2] != seen[-4]) | (seen[-1] != seen[-2]) | ~used
    for i in range(cfg.depth):
        r = F8.ROWS * i + F8.ROW_W + 3  # context_kv.weight
        assert float(seen[-1][r]) > 0
        moved[r] = True
    assert bool(moved.all()), torch.nonzero(~moved).flatten().tolist()


def test_graph_replay_rejects_other_shapes(vds):
    from video_diffusion_speedrun_amd.graph import GraphedTrainStep
    cfg = O.DiTConfig(in_channels=16, hidden_size=128, depth=1, num_heads=2, cross_attn_input_size=64,
                      residual_v=True, train_bias_and_rms=False)
    m = build(vds, cfg, O.init_params(cfg, seed=42, randomize_zero_init=True, init_std_factor=1.0))
    groups, _ = m.get_mup_setup(3e-3, 0.1, ["patch_proj", "context_kv", "positional_embedding"])
    gs = GraphedTrainStep(m, vds["optim"].MuAdamW(groups, betas=(0.95, 0.99)), None, "cuda", eager_steps=1)
    gs.step({"latent": torch.randn(2, 16, 4, 8, 8).cuda(), "context": torch.randn(2, 16, 64).cuda()})
    with pytest.raises(ValueError):
        gs.step({"latent": torch.randn(2, 16, 4, 8, 16).cuda(), "context": torch.randn(2, 16, 64).cuda()})


@pytest.mark.parametrize("D,H", [(128, 4), (512, 16), (192, 2), (96, 2), (160, 2), (224, 2), (80, 2), (208, 2)],
                         ids=["hd32", "hd32_reference_smoke_width", "hd96", "hd48_on_64", "hd80_on_96x80", "hd112_on_128",
                              "hd40_on_64", "hd104_on_128"])
def test_head_dim_32_and_96_vs_oracle(vds, D, H):
    """head_dim 32 (the reference's own smoke test builds width 512 with its default 16 heads, model.py:545-565),
    head_dim 96 (no padding at all) and -- round 5 -- head dims that have no kernel instance of their own and run on the
    next one with their rows' tails fetched as zeros (48 / 40 -> 64, 80 -> 96 x 80, 112 / 104 -> 128; the reference takes
    any hidden_size // num_heads, model.py:57): forward, loss and every gradient against the fp32 oracle"""
    cfg = O.DiTConfig(in_channels=16, hidden_size=D, depth=2, num_heads=H, cross_attn_input_size=64,
                      residual_v=True, train_bias_and_rms=True)
    P = O.init_params(cfg, seed=41, randomize_zero_init=True, init_std_factor=1.0)
    g = torch.Generator().manual_seed(42)
    lat = (2, 16, 4, 16, 16)
    x = torch.randn(*lat, generator=g).to(bf16)
    ctx = torch.randn(lat[0], 24, 64, generator=g).to(bf16)
    t = torch.tensor([0.25, 0.7]).to(bf16)
    v = torch.randn(*lat, generator=g).to(bf16)
    start = (3, 1, 2)
    Pg = {k: w.clone().requires_grad_(True) for k, w in P.items()}
    o_ref = O.dit_forward(Pg, cfg, x.float(), ctx.float(), t.float(), start)
    l_ref, _ = O.flow_loss(v, o_ref)
    l_ref.backward()
    m = build(vds, cfg, P)
    out = m(x.cuda(), ctx.cuda(), t.cuda(), rope_start=start)
    loss, _ = vds["train"].flow_loss(out, v.cuda())
    loss.backward()
    assert rel(out, o_ref) <= 2.5e-2, rel(out, o_ref)
    assert abs(loss.item() - l_ref.item()) / l_ref.item() <= 1e-2
    bad = []
    for k, p in m.named_parameters():
        if Pg[k].grad is None or k.endswith("lambda_param") or float(Pg[k].grad.abs().max()) == 0:
            continue
        c, e = cosine(p.grad, Pg[k].grad), rel(p.grad, Pg[k].grad)
        _track(f"hd{D // H}:" + k, c, e)
        if not (c >= GRAD_COS and e <= GRAD_REL):
            bad.append((k, c, e))
    assert not bad, bad
    with pytest.raises(ValueError, match="no attention kernel instance"):
        vds["model"].DiT(in_channels=16, hidden_size=272, depth=1, num_heads=2)  # head_dim 136 > 128: stated, not silent
    with pytest.raises(ValueError, match="no attention kernel instance"):
        vds["model"].DiT(in_channels=16, hidden_size=200, depth=1, num_heads=4)  # head_dim 50: no RoPE table either


@pytest.mark.parametrize("D,H,lat", [(144, 2, (2, 16, 4, 8, 8)), (256, 4, (2, 16, 4, 16, 16))])
def test_fp8_step_close_to_oracle(vds, D, H, lat):
    """BASELINE config 5 (no reference counterpart): DiT.enable_fp8() runs the qkv / MLP GEMMs of every block in
    OCP fp8 (fp8.py).  Against the fp32 CPU oracle of the same step the stated tolerance is: output within 2.5e-2
    relative (the bf16 path's own bound), loss within 1e-2, every parameter gradient cosine >= 0.99 and relative
    error <= 0.15 (measured: cosine >= 0.996, <= 0.09 -- e4m3 / e5m2 carry 3 / 2 mantissa bits); gradients of
    layers that stay in bf16 keep the bf16 bound of 6e-2."""
    cfg = O.DiTConfig(in_channels=16, hidden_size=D, depth=3, num_heads=H, cross_attn_input_size=64,
                      residual_v=True, train_bias_and_rms=False)
    P = O.init_params(cfg, seed=51, randomize_zero_init=True, init_std_factor=1.0)
    g = torch.Generator().manual_seed(52)
    x = torch.randn(*lat, generator=g).to(bf16)
    ctx = torch.randn(lat[0], 16, 64, generator=g).to(bf16)
    t = torch.tensor([0.3, 0.8]).to(bf16)
    v = torch.randn(*lat, generator=g).to(bf16)
    start = (1, 2, 3)
    Pg = {k: w.clone().requires_grad_(True) for k, w in P.items()}
    o_ref = O.dit_forward(Pg, cfg, x.float(), ctx.float(), t.float(), start)
    l_ref, _ = O.flow_loss(v, o_ref)
    l_ref.backward()
    m = build(vds, cfg, P).enable_fp8(attention=False)  # (fp8 attention: tests/test_attn_fp8_gpu.py)
    from video_diffusion_speedrun_amd import ops
    ops.prof_enable()
    out = m(x.cuda(), ctx.cuda(), t.cuda(), rope_start=start)
    loss, _ = vds["train"].flow_loss(out, v.cuda())
    loss.backward()
    stats = ops.prof_collect()
    ops.prof_enable(0)
    # all 7 linears of a block ran in fp8: forward, input gradient, weight gradient (context_kv has no input gradient:
    # the text context is data)
    assert stats["gemm_fp8"]["launches"] == 20 * cfg.depth
    assert rel(out, o_ref) <= 2.5e-2, rel(out, o_ref)
    assert abs(loss.item() - l_ref.item()) / l_ref.item() <= 1e-2
    bad = []
    for k, p in m.named_parameters():
        if Pg[k].grad is None or k.endswith("lambda_param") or float(Pg[k].grad.abs().max()) == 0:
            continue
        c, e = cosine(p.grad, Pg[k].grad), rel(p.grad, Pg[k].grad)
        if not (c >= 0.99 and e <= 0.15):
            bad.append((k, c, e))
    assert not bad, bad
    # second pass over the same inputs: gelu(fc1) and the fc2 input gradient now leave their GEMMs as fp8, scaled
    # by the amax the first pass recorded (delayed scaling with an exact history) -> the same gradients
    g1 = {k: p.grad.clone() for k, p in m.named_parameters()}
    m.zero_grad()
    out2 = m(x.cuda(), ctx.cuda(), t.cuda(), rope_start=start)
    assert m._fp8_hist.ready
    loss2, _ = vds["train"].flow_loss(out2, v.cuda())
    loss2.backward()
    assert rel(out2, out) <= 1e-6
    for k, p in m.named_parameters():
        if float(g1[k].abs().max()) > 0:  # (a lambda gradient is a sum of cancelling terms accumulated with fp32 atomics)
            assert rel(p.grad, g1[k]) <= (5e-3 if k.endswith("lambda_param") else 1e-4), (k, rel(p.grad, g1[k]))
    with pytest.raises(ValueError):  # 4*(2*4*4 + 16) is fine, but an odd token count is not: clear error, no fallback
        m(torch.randn(1, 16, 2, 6, 6).cuda(), ctx[:1].cuda(), t[:1].cuda(), rope_start=start)


def _emulate_ranks(vds, monkeypatch, cfg, P, W):
    """W replicas of one model in this process with the real shard layout, stream / event choreography, HIP kernels
    and sharded optimizer; only the two collectives are replaced by an in-process exchange (RCCL refuses two ranks on
    one device).  Returns (replicas, pending reduce-scatter inputs per group)."""
    from video_diffusion_speedrun_amd import params as PM
    from video_diffusion_speedrun_amd.fsdp import apply_fsdp
    reps, tokens, pending = [], [object() for _ in range(W)], {}

    def find(buf, attr):
        for r, m in enumerate(reps):
            for gi, grp in enumerate(m._groups):
                if getattr(grp, attr).data_ptr() == buf.data_ptr():
                    return r, gi
        raise AssertionError("collective on an unknown buffer")

    def fake_all_gather(out, inp, group=None):
        _, gi = find(inp, "shadow")
        parts = out.view(W, -1)
        for r2, m2 in enumerate(reps):  # what rank r2 contributes: the bf16 cast of its fp32 master shard
            parts[r2].copy_(m2._groups[gi].master.to(bf16))

    def fake_reduce_scatter(out, inp, group=None):
        r, gi = find(inp, "gfull")
        pending.setdefault(gi, {})[r] = inp.clone()
        if len(pending[gi]) == W:
            avg = sum(pending[gi].values()) / W
            for r2, m2 in enumerate(reps):
                m2._groups[gi].gshard.copy_(avg.view(W, -1)[r2])

    monkeypatch.setattr(PM, "all_gather_flat", fake_all_gather)
    monkeypatch.setattr(PM, "reduce_scatter_avg", fake_reduce_scatter)
    for r in range(W):
        m = build(vds, cfg, P)
        reps.append(apply_fsdp(m, torch.bfloat16, torch.float32, process_group=tokens[r], world_rank=(W, r)))
        assert m._fsdp is not None and m._groups[1].world == W and m._groups[1].rank == r
    return reps, pending


@pytest.mark.parametrize("W,prefetch", [(2, 0), (4, 1)], ids=["w2_all_upfront", "w4_window1"])
def test_two_emulated_ranks_on_one_gpu(vds, monkeypatch, W, prefetch):
    """The sharded train step with world_size 2 / 4 on ONE GPU (see _emulate_ranks).  Each rank takes its own
    micro-batch; after one step the concatenated parameter shards must equal the unsharded model stepped on the
    concatenated batch (gradient = average over ranks, model.py:516-519).  prefetch: the all-gather window of
    fsdp.ShardRuntime (0 = every group up-front, 1 = one group ahead like FSDP2)."""
    cfg = O.DiTConfig(in_channels=16, hidden_size=128, depth=3, num_heads=2, cross_attn_input_size=64,
                      residual_v=True, train_bias_and_rms=False)
    P = O.init_params(cfg, seed=61, randomize_zero_init=True, init_std_factor=1.0)
    g = torch.Generator().manual_seed(62)
    x = torch.randn(4, 16, 4, 8, 8, generator=g).to(bf16).cuda()
    ctx = torch.randn(4, 16, 64, generator=g).to(bf16).cuda()
    t = torch.tensor([0.2, 0.5, 0.7, 0.9]).to(bf16).cuda()
    v = torch.randn(4, 16, 4, 8, 8, generator=g).to(bf16).cuda()
    start = (1, 2, 3)

    def step(m, sl):
        groups, _ = m.get_mup_setup(3e-3, 0.1, ["patch_proj", "context_kv", "positional_embedding"])
        opt = vds["optim"].MuAdamW(groups, betas=(0.95, 0.99))
        out = m(x[sl], ctx[sl], t[sl], rope_start=start)
        loss, _ = vds["train"].flow_loss(out, v[sl])
        loss.backward()
        return opt, loss

    ref = build(vds, cfg, P)
    opt, loss_ref = step(ref, slice(0, 4))
    want_g = [g.gfull.clone() for g in ref._groups]  # the full-batch gradient of every flat group (world 1: gshard = gfull)
    opt.step()
    want = ref.full_state_dict()

    reps, pending = _emulate_ranks(vds, monkeypatch, cfg, P, W)
    per = 4 // W
    opts, losses = [], []
    for r in range(W):
        reps[r]._fsdp.prefetch = prefetch
        o, l = step(reps[r], slice(per * r, per * (r + 1)))
        opts.append(o)
        losses.append(l.item())
    torch.cuda.synchronize()  # rank 0's gradient shards were completed while rank 1 ran its backward
    assert all(len(d) == W for d in pending.values()) and len(pending) == 1 + cfg.depth
    assert abs(sum(losses) / W - loss_ref.item()) <= 1e-5 * abs(loss_ref.item())
    # the reduce-scattered gradient shards = the full-batch gradient (fp32 sums in another order: 1e-5) ...
    for gi in range(len(want_g)):
        gs = torch.cat([m._groups[gi].gshard for m in reps])
        assert rel(gs[:want_g[gi].numel()], want_g[gi]) <= 1e-5, (gi, rel(gs[:want_g[gi].numel()], want_g[gi]))
    for o in opts:
        o.step()
    torch.cuda.synchronize()
    # ... and the stepped parameters agree.  (2e-4: the FIRST AdamW step is g / (|g| + eps), which turns the 1e-7 order
    # differences of the gradients into visible ones wherever |g| is near eps; 1e-4 failed once in five full-suite runs
    # at 1.06e-4 on one tensor.  The gradient check above is the tight one.)
    for gi, grp0 in enumerate(reps[0]._groups):
        flat = torch.cat([m._groups[gi].master for m in reps])
        for n in grp0.names:
            o0 = grp0.offsets[n]
            got = flat[o0:o0 + want[n].numel()].view(want[n].shape)
            assert rel(got, want[n]) <= 2e-4, (n, rel(got, want[n]))
    # every parameter of a sharded replica is this rank's 1-D piece, and the pieces tile the tensor
    others = [dict(m.named_parameters()) for m in reps[1:]]
    for n, p0 in reps[0].named_parameters():
        assert p0.dim() == 1 and p0.numel() + sum(o[n].numel() for o in others) == want[n].numel()
    for i in range(1, cfg.depth):  # a residual-V lambda (1 element) lives on exactly one rank
        assert sum(dict(m.named_parameters())[f"blocks.{i}.lambda_param"].numel() for m in reps) == 1


def test_sharded_hip_step_matches_the_reference_fsdp_run(vds, monkeypatch, golden_dir):
    """SURVEY 8(c) G5 on the HIP path: the inputs of the REFERENCE's own 2-rank `apply_fsdp` train step
    (tests/golden/g6_fsdp.pt, oracle/make_golden_fsdp.py) through two emulated ranks of this build: per-rank losses,
    the reduce-scattered gradient and the parameters after the sharded MuAdamW step vs what the reference produced
    (both sides compute in bf16: gradients within 3e-2, the AdamW update direction cosine >= 0.98)."""
    fx = torch.load(os.path.join(golden_dir, "g6_fsdp.pt"), weights_only=False)
    cfg = O.DiTConfig(**fx["cfg"])
    P = O.init_params(cfg, seed=fx["param_seed"], randomize_zero_init=True, init_std_factor=1.0)
    W = fx["world"]
    reps, pending = _emulate_ranks(vds, monkeypatch, cfg, P, W)
    b, opts = fx["batch"], []
    for r, m in enumerate(reps):
        sl = slice(2 * r, 2 * r + 2)
        groups, _ = m.get_mup_setup(fx["lr"], fx["wd"], fx["consts"])
        opts.append(vds["optim"].MuAdamW(groups, betas=(0.95, 0.99)))
        tt = O.time_shift(b["z"][sl].to(bf16))
        zt, v = vds["ops"].noise_latents(b["latent"][sl].to(bf16).cuda(), b["noise"][sl].to(bf16).cuda(),
                                         tt.float().cuda())
        out = m(zt, b["context"][sl].to(bf16).cuda(), tt.cuda(), rope_start=tuple(fx["rope_start"]))
        loss, _ = vds["train"].flow_loss(out, v)
        loss.backward()
        assert abs(loss.item() - fx["losses"][r]) <= 1e-2 * fx["losses"][r]
    torch.cuda.synchronize()
    for o in opts:
        o.step()
    torch.cuda.synchronize()
    for gi, grp0 in enumerate(reps[0]._groups):
        gflat = torch.cat([m._groups[gi].gshard for m in reps])
        pflat = torch.cat([m._groups[gi].master for m in reps])
        for n in grp0.names:
            if n not in fx["reduced_grads"] or n.endswith("lambda_param"):
                continue
            o0, want_g = grp0.offsets[n], fx["reduced_grads"][n]
            got_g = gflat[o0:o0 + want_g.numel()].view(want_g.shape)
            assert rel(got_g, want_g) <= 3e-2 and cosine(got_g, want_g) >= 0.999, (n, rel(got_g, want_g))
            want_p = fx["params_after_step"][n]
            got_p = pflat[o0:o0 + want_p.numel()].view(want_p.shape)
            assert rel(got_p, want_p) <= 1e-3, (n, rel(got_p, want_p))
            if want_p.numel() >= 1024:   # direction of the first AdamW update (~ lr * sign(g)): sensitive to small g
                assert cosine(got_p.cpu() - P[n], want_p - P[n]) >= 0.98, n


def test_reference_written_checkpoint_loads_on_the_gpu(vds, tmp_path, golden_dir):
    """the DCP directory the reference wrote (tests/golden/g6_dcp: `get_model_state_dict` + `dcp.save` of its 2-rank
    FSDP model, train.py:553,581-584) loaded into a live GPU model -- flat groups already materialised -- gives the
    same forward as a model built directly from the reference's parameters"""
    import shutil
    from video_diffusion_speedrun_amd import checkpoint as ck
    fx = torch.load(os.path.join(golden_dir, "g6_fsdp.pt"), weights_only=False)
    cfg = O.DiTConfig(**fx["cfg"])
    d = str(tmp_path / "ref_ckpt")
    shutil.copytree(os.path.join(golden_dir, "g6_dcp"), d)
    m = build(vds, cfg, O.init_params(cfg, seed=1234, randomize_zero_init=True, init_std_factor=1.0))
    g = torch.Generator().manual_seed(8)
    x = torch.randn(2, 16, 4, 8, 8, generator=g).to(bf16).cuda()
    ctx = torch.randn(2, 6, 64, generator=g).to(bf16).cuda()
    t = torch.tensor([0.25, 0.75]).cuda()
    with torch.no_grad():
        m(x, ctx, t, rope_start=(1, 2, 3))                       # materialises the flat groups and bf16 copies
        assert ck.load_checkpoint(d, m) == 0
        got = m(x, ctx, t, rope_start=(1, 2, 3))
        want = build(vds, cfg, fx["params_after_step"])(x, ctx, t, rope_start=(1, 2, 3))
    assert torch.equal(got, want)
    sd = m.full_state_dict()
    assert all(torch.equal(sd[n].cpu(), w) for n, w in fx["params_after_step"].items())


def test_parameter_writes_after_a_step_reach_the_next_forward(vds):
    """MuAdamW writes the bf16 compute copy itself and the next forward skips its cast; an in-place
    `load_state_dict` / `p.mul_()` between the step and the forward must still be seen (stale shadow)"""
    cfg = O.DiTConfig(in_channels=16, hidden_size=128, depth=2, num_heads=2, cross_attn_input_size=64,
                      residual_v=True, train_bias_and_rms=False)
    P = O.init_params(cfg, seed=21, randomize_zero_init=True, init_std_factor=1.0)
    P2 = O.init_params(cfg, seed=22, randomize_zero_init=True, init_std_factor=1.0)
    m = build(vds, cfg, P)
    groups, _ = m.get_mup_setup(3e-3, 0.1, ["patch_proj", "context_kv", "positional_embedding"])
    opt = vds["optim"].MuAdamW(groups, betas=(0.95, 0.99))
    g = torch.Generator().manual_seed(3)
    batch = {"latent": torch.randn(2, 16, 4, 8, 8, generator=g), "context": torch.randn(2, 16, 64, generator=g),
             "prompt": ["", ""]}
    vds["train"].train_step(m, opt, None, batch, "cuda", rope_start=(1, 2, 3))
    assert all(grp.shadow_fresh for grp in m._groups)
    m.load_state_dict(P2)                      # non-assign: copies into the flat fp32 masters
    x, ctx, t = batch["latent"].cuda().to(bf16), batch["context"].cuda().to(bf16), torch.tensor([0.3, 0.6]).cuda()
    with torch.no_grad():
        got = m(x, ctx, t, rope_start=(1, 2, 3))
        want = build(vds, cfg, P2)(x, ctx, t, rope_start=(1, 2, 3))
    assert torch.equal(got, want)
    with torch.no_grad():
        m.final_proj.weight.mul_(0.0)
        m.final_proj.bias.mul_(0.0)
        assert m(x, ctx, t, rope_start=(1, 2, 3)).abs().max().item() == 0


def test_upstream_gradient_and_accumulation_like_autograd(vds):
    """`(loss * s).backward()` scales every gradient by s, and a second backward before zero_grad adds into
    p.grad -- the autograd semantics the reference's loop relies on (_FlowLoss ignored both)"""
    cfg = O.DiTConfig(in_channels=16, hidden_size=128, depth=2, num_heads=2, cross_attn_input_size=64,
                      residual_v=True, train_bias_and_rms=False)
    m = build(vds, cfg, O.init_params(cfg, seed=23, randomize_zero_init=True, init_std_factor=1.0))
    g = torch.Generator().manual_seed(4)
    x = torch.randn(4, 16, 4, 8, 8, generator=g).to(bf16).cuda()
    ctx = torch.randn(4, 16, 64, generator=g).to(bf16).cuda()
    t = torch.tensor([0.2, 0.4, 0.6, 0.8]).cuda()
    v = torch.randn(4, 16, 4, 8, 8, generator=g).to(bf16).cuda()

    def grads(sl, scale=1.0, zero=True):
        if zero:
            m.zero_grad()
        out = m(x[sl], ctx[sl], t[sl], rope_start=(1, 2, 3))
        loss, _ = vds["train"].flow_loss(out, v[sl])
        (loss * scale).backward()
        return {k: p.grad.clone() for k, p in m.named_parameters()}

    full = grads(slice(0, 4))
    half = grads(slice(0, 4), scale=0.5)
    for k in full:
        if float(full[k].abs().max()) > 0:
            assert rel(half[k], 0.5 * full[k]) <= 2e-3, k   # dout is re-rounded to bf16 after scaling
    grads(slice(0, 2), scale=0.5)
    acc = grads(slice(2, 4), scale=0.5, zero=False)          # two micro-batches of 2 == one batch of 4
    for k in full:
        if float(full[k].abs().max()) > 0 and not k.endswith("lambda_param"):
            assert rel(acc[k], full[k]) <= 1e-2, (k, rel(acc[k], full[k]))


def test_per_gpu_batch_above_16(vds):
    """the reference's default is 64 samples per rank (train.py:150): B = 18 through the whole step vs the oracle"""
    cfg = O.DiTConfig(in_channels=16, hidden_size=128, depth=2, num_heads=2, cross_attn_input_size=64,
                      residual_v=True, train_bias_and_rms=False)
    P = O.init_params(cfg, seed=27, randomize_zero_init=True, init_std_factor=1.0)
    g = torch.Generator().manual_seed(28)
    B = 18
    x = torch.randn(B, 16, 2, 8, 8, generator=g).to(bf16)
    ctx = torch.randn(B, 8, 64, generator=g).to(bf16)
    t = torch.rand(B, generator=g).to(bf16)
    v = torch.randn(B, 16, 2, 8, 8, generator=g).to(bf16)
    Pg = {k: w.clone().requires_grad_(True) for k, w in P.items()}
    o_ref = O.dit_forward(Pg, cfg, x.float(), ctx.float(), t.float(), (1, 2, 3))
    l_ref, _ = O.flow_loss(v, o_ref)
    l_ref.backward()
    m = build(vds, cfg, P)
    out = m(x.cuda(), ctx.cuda(), t.cuda(), rope_start=(1, 2, 3))
    assert rel(out, o_ref) <= 2.5e-2
    loss, _ = vds["train"].flow_loss(out, v.cuda())
    loss.backward()
    for k, p in m.named_parameters():
        if Pg[k].grad is not None and not k.endswith("lambda_param") and float(Pg[k].grad.abs().max()) > 0:
            assert cosine(p.grad, Pg[k].grad) >= GRAD_COS and rel(p.grad, Pg[k].grad) <= GRAD_REL, k


def test_fp8_weight_history_follows_weights_replaced_in_process(vds, monkeypatch):
    """the seven block weights are quantised with the amax of the PREVIOUS step.  Weights replaced in
    process must not be clipped at the old amax.  (a) load_state_dict of a checkpoint with 10x larger block weights into
    an armed model: its next step equals the step of a fresh model given the same weights (which measures every amax
    itself).  (b) an in-place write through the parameters (an EMA swap): the next TRAINING step measures the weights'
    own amax again (7 absmax passes per block), also when an evaluation forward ran in between, and the step after it
    is back on the history."""
    from video_diffusion_speedrun_amd import ops
    cfg = O.DiTConfig(in_channels=16, hidden_size=144, depth=2, num_heads=2, cross_attn_input_size=64,
                      residual_v=True, train_bias_and_rms=False)
    P = O.init_params(cfg, seed=61, randomize_zero_init=True, init_std_factor=1.0)
    big = {k: (w * 10.0 if (k.startswith("blocks.") and k.endswith("weight") and w.dim() == 2) else w.clone())
           for k, w in P.items()}
    g = torch.Generator().manual_seed(62)
    lat = (2, 16, 4, 8, 8)
    x = torch.randn(*lat, generator=g).to(bf16).cuda()
    ctx = torch.randn(2, 16, 64, generator=g).to(bf16).cuda()
    t = torch.tensor([0.3, 0.8]).to(bf16).cuda()
    v = torch.randn(*lat, generator=g).to(bf16).cuda()
    start = (1, 2, 3)
    calls = [0]
    real_absmax = ops.absmax

    def counting_absmax(*a, **k):
        calls[0] += 1
        return real_absmax(*a, **k)

    monkeypatch.setattr(ops, "absmax", counting_absmax)

    def step(m):
        m.zero_grad()
        calls[0] = 0
        out = m(x, ctx, t, rope_start=start)
        loss, _ = vds["train"].flow_loss(out, v)
        loss.backward()
        return out.detach().clone(), {k: p.grad.clone() for k, p in m.named_parameters()}, calls[0]

    fresh = build(vds, cfg, big).enable_fp8(attention=False)
    o_ref, g_ref, _ = step(fresh)
    # (a)
    m = build(vds, cfg, P).enable_fp8(attention=False)
    step(m)
    step(m)
    assert m._fp8_hist.ready  # delayed scaling armed on the small weights
    m.load_state_dict(big, strict=True)
    assert not m._fp8_hist.ready
    o, gr, _ = step(m)
    assert rel(o, o_ref) <= 1e-6, rel(o, o_ref)
    for k in g_ref:
        if float(g_ref[k].abs().max()) > 0:
            assert rel(gr[k], g_ref[k]) <= (5e-3 if k.endswith("lambda_param") else 1e-4), (k, rel(gr[k], g_ref[k]))
    # (b)
    m = build(vds, cfg, P).enable_fp8(attention=False)
    opt = vds["optim"].MuAdamW(m.get_mup_setup(1e-4, 0.0, ["patch_proj", "context_kv", "positional_embedding"])[0])
    for _ in range(3):
        _, _, n_steady = step(m)
        opt.step()
    assert m._fp8_hist.ready
    _, _, n_steady = step(m)  # weights written by the fused optimizer only: the history is trusted
    opt.step()
    with torch.no_grad():
        for k, p in m.named_parameters():
            p.mul_(1.25)
        m(x, ctx, t, rope_start=start)  # an evaluation forward in between must not consume the mark
    _, _, n_moved = step(m)
    assert n_moved == n_steady + 7 * cfg.depth, (n_steady, n_moved)
    opt.step()
    _, _, n_after = step(m)
    assert n_after == n_steady, (n_steady, n_after)


