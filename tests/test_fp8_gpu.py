"""fp8 path (BASELINE config 5) kernel tests: quantiser and fp8 GEMM through the C ABI.
The reference has no fp8 path; the checker here is torch's own float8 dtypes on the CPU (OCP e4m3fn / e5m2,
round-to-nearest-even) and exact small-integer products."""
import pytest
import torch

pytestmark = pytest.mark.gpu
bf16, f32 = torch.bfloat16, torch.float32


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from video_diffusion_speedrun_amd import ops
    return ops


def gen(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(bf16)


@pytest.mark.parametrize("fmt", [0, 1])
@pytest.mark.parametrize("M,K", [(300, 136), (128, 128), (516, 1152), (4, 8)])
def test_quant_matches_torch_float8(ops, fmt, M, K):
    x = gen(M, K, seed=M + K, scale=3.0)
    x[0, 0] = 0.0
    amax = ops.absmax(x.cuda())
    assert amax.item() == x.float().abs().max().item()
    q, qt, dq = ops.quant_fp8(x.cuda(), fmt, amax, rowmajor=True, transposed=True)
    fmax = 448.0 if fmt == 0 else 57344.0
    scale = torch.tensor(fmax, dtype=f32) / amax.cpu()
    want = (x.float() * scale).clamp(-fmax, fmax).to(ops.fp8_dtypes[fmt])
    assert torch.equal(q.cpu().view(torch.uint8), want.view(torch.uint8))
    assert torch.equal(qt.cpu().view(torch.uint8), want.view(torch.uint8).t().contiguous())
    assert abs(dq.item() - amax.item() / fmax) <= 1e-7 * dq.item()
    # delayed scaling: quantise with a given (older) amax while recording the current one in the same pass
    rec = torch.zeros(1, device="cuda")
    old = amax * 0.75
    q3, _, _ = ops.quant_fp8(x.cuda(), fmt, old, amax_out=rec)
    want3 = (x.float() * (torch.tensor(fmax, dtype=f32) / old.cpu())).clamp(-fmax, fmax).to(ops.fp8_dtypes[fmt])
    assert torch.equal(q3.cpu().view(torch.uint8), want3.view(torch.uint8)) and rec.item() == amax.item()
    # strided source (a column block of a wider matrix), transposed copy only
    wide = gen(M, K + 64, seed=7, scale=2.0).cuda()
    am2 = ops.absmax(wide[:, 8:8 + K])
    _, qt2, _ = ops.quant_fp8(wide[:, 8:8 + K], fmt, am2, rowmajor=False, transposed=True)
    sc2 = torch.tensor(fmax, dtype=f32) / am2.cpu()
    want2 = (wide[:, 8:8 + K].cpu().float() * sc2).clamp(-fmax, fmax).to(ops.fp8_dtypes[fmt])
    assert torch.equal(qt2.cpu().view(torch.uint8), want2.view(torch.uint8).t().contiguous())


def _small_ints(M, K, seed, vals):
    g = torch.Generator().manual_seed(seed)
    v = torch.tensor(vals, dtype=f32)
    return v[torch.randint(0, len(vals), (M, K), generator=g)]


@pytest.fixture(params=[0, 256], ids=["auto", "t256"])
def tile8(request, ops):
    """the fp8 GEMM under the dispatcher's choice and with the 256 x 256 output tile pinned"""
    ops.gemm_force_tile(request.param)
    yield request.param
    ops.gemm_force_tile(0)


@pytest.mark.parametrize("a_fmt", [0, 1])
@pytest.mark.parametrize("M,N,K", [(300, 264, 272), (256, 256, 128), (1000, 520, 1152), (16, 8, 16), (700, 1152, 400)])
def test_gemm_fp8_exact_on_small_integers(ops, tile8, a_fmt, M, N, K):
    """operands exactly representable in both fp8 formats, products and fp32 sums exact: the lane maps of
    v_mfma_f32_16x16x128_f8f6f4, the K tail and ragged tiles must give the exact matrix product"""
    from video_diffusion_speedrun_amd._lib import EPI_F32, EPI_STORE
    A = _small_ints(M, K, 1, [-2.0, -1.0, -0.5, 0.0, 0.5, 1.0, 2.0, 3.0])
    B = _small_ints(N, K, 2, [-1.5, -1.0, 0.0, 0.25, 1.0, 2.0])
    want = A @ B.t()
    Aq = A.to(ops.fp8_dtypes[a_fmt]).cuda()
    Bq = B.to(torch.float8_e4m3fn).cuda()
    assert torch.equal(Aq.float().cpu(), A) and torch.equal(Bq.float().cpu(), B)
    sa = torch.tensor([0.5], device="cuda")
    sb = torch.tensor([4.0], device="cuda")
    C32 = torch.zeros(M, N, dtype=f32, device="cuda")
    ops.gemm_fp8(EPI_F32, M, N, K, Aq, Bq, sa, sb, a_fmt, C32, N)
    assert torch.equal(C32.cpu(), want * 2.0)
    if K >= 1024:  # split-K with atomic accumulation (the weight-gradient form)
        C32.zero_()
        ops.gemm_fp8(EPI_F32, M, N, K, Aq, Bq, sa, sb, a_fmt, C32, N, split_k=3)
        assert torch.equal(C32.cpu(), want * 2.0)
    Cb = torch.empty(M, N, dtype=bf16, device="cuda")
    ops.gemm_fp8(EPI_STORE, M, N, K, Aq, Bq, sa, None, a_fmt, Cb, N)
    assert torch.equal(Cb.cpu(), (want * 0.5).to(bf16))


@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (1152, 1152, 1040), (144, 3456, 912), (272, 48, 4112), (16, 16, 16)])
def test_gemm_fp8_tn_weight_gradient_exact_on_small_integers(ops, M, N, K):
    """round 4: dW[M,N] = sum_k dy[k,m] x[k,n] read straight from the token-major fp8 copies (both operands k-major,
    ds_read_b64_tr_b8 fragments; dy e5m2, x e4m3): exact on small integers -- lane maps of the transposing reads, the
    swizzled [128 tokens][128 bytes] LDS image, ragged token counts (K not a multiple of 128), ragged M / N tiles,
    split-K with atomic accumulation"""
    from video_diffusion_speedrun_amd._lib import EPI_F32
    A = _small_ints(K, M, 5, [-2.0, -1.0, -0.5, 0.0, 0.5, 1.0, 2.0, 3.0])      # dy [tokens, out features]
    B = _small_ints(K, N, 6, [-1.5, -1.0, 0.0, 0.25, 1.0, 2.0])                # x  [tokens, in features]
    want = A.t() @ B
    Aq, Bq = A.to(torch.float8_e5m2).cuda(), B.to(torch.float8_e4m3fn).cuda()
    assert torch.equal(Aq.float().cpu(), A) and torch.equal(Bq.float().cpu(), B)
    sa, sb = torch.tensor([0.5], device="cuda"), torch.tensor([4.0], device="cuda")
    for split in (1, 3) if K >= 900 else (1,):  # (token counts are multiples of 16: fp8.supported)
        C32 = torch.zeros(M, N, dtype=f32, device="cuda")
        ops.gemm_fp8(EPI_F32, M, N, K, Aq, Bq, sa, sb, 1, C32, N, split_k=-split if split > 1 else 1, tn=True)
        assert torch.equal(C32.cpu(), want * 2.0), (split, (C32.cpu() - want * 2).abs().max())


def test_fp8_wgrad_from_token_major_copies_matches_the_transposed_path(ops):
    """fp8.wgrad on the row-major copies (TN, default) == the NT product of the transposed copies it replaces, bit for
    bit up to the fp32 atomic summation order; asymmetric operands (a transposed or permuted fragment map cannot pass)"""
    from video_diffusion_speedrun_amd import fp8 as F8
    M, N, K = 4112, 432, 144  # tokens, out features, in features
    dy, x = gen(M, N, seed=21), gen(M, K, seed=22)
    dyq = F8.Q(dy.cuda(), F8.E5M2, True, True, weight=True)   # (weight=True: keep both copies for this comparison)
    xq = F8.Q(x.cuda(), F8.E4M3, True, True, weight=True)
    assert dyq.t is not None and xq.t is not None
    got = torch.zeros(N, K, dtype=f32, device="cuda")
    F8.wgrad(dyq, xq, got)
    old, F8.TN = F8.TN, False
    try:
        ref = torch.zeros(N, K, dtype=f32, device="cuda")
        F8.wgrad(dyq, xq, ref)
    finally:
        F8.TN = old
    exact = (dyq.q.float() * dyq.s).t() @ (xq.q.float() * xq.s)
    assert ((got - exact).norm() / exact.norm()).item() <= 1e-4  # (`exact` is torch's fp32 matmul)
    assert ((got - ref).norm() / ref.norm()).item() <= 1e-5


def test_linear_fp8_close_to_bf16_linear(ops):
    """quantise -> fp8 GEMM -> dequantise against the fp32 product of the same bf16 inputs: error at the
    e4m3 quantisation level (2^-4 relative per element, averaged down by the contraction)"""
    from video_diffusion_speedrun_amd._lib import EPI_STORE
    M, N, K = 1024, 768, 1152
    x, W = gen(M, K, seed=3), gen(N, K, seed=4, scale=0.03)
    xq, _, sx = ops.quant_fp8(x.cuda(), 0, ops.absmax(x.cuda()))
    Wq, _, sw = ops.quant_fp8(W.cuda(), 0, ops.absmax(W.cuda()))
    y = torch.empty(M, N, dtype=bf16, device="cuda")
    ops.gemm_fp8(EPI_STORE, M, N, K, xq, Wq, sx, sw, 0, y, N)
    ref = x.float() @ W.float().t()
    err = ((y.cpu().float() - ref).norm() / ref.norm()).item()
    assert err <= 4e-2, err
    # and exactly the product of the dequantised operands (up to the bf16 store)
    ref_q = (xq.cpu().float() * sx.cpu()) @ (Wq.cpu().float() * sw.cpu()).t()
    err_q = ((y.cpu().float() - ref_q).norm() / ref_q.norm()).item()
    assert err_q <= 3e-3, err_q


@pytest.mark.parametrize("M,N,K", [(304, 264, 144), (1024, 4608, 1152)])
def test_gemm_epilogue_emits_fp8_copies(ops, tile8, M, N, K, monkeypatch):
    """vds_fp8_out: the fc1 epilogue (bias + GELU) and the fc2-dgrad epilogue (gelu') write their result as fp8
    row-major + transposed, record its amax and (dgrad) its column sums -- bit-identical to quantising the bf16
    result in a separate pass with the same scale.  (Since round 4 the model no longer asks for the transposed copies
    -- fp8.TN -- but the C ABI still produces them: the test runs with the pre-round-4 setting.)"""
    from video_diffusion_speedrun_amd import fp8 as F8
    monkeypatch.setattr(F8, "TN", False)
    x, W, b = gen(M, K, seed=11), gen(N, K, seed=12, scale=0.05), gen(N, seed=13, scale=0.1)
    xq, wq = F8.Q(x.cuda(), 0, True, True), F8.Q(W.cuda(), 0, True, True)
    pre, act = F8.fwd_gelu(xq, wq, b.cuda())
    amax = ops.absmax(act)
    rec = torch.zeros(1, device="cuda")
    pre2, qa = F8.fwd_gelu_emit(xq, wq, b.cuda(), amax, rec, True)
    q_ref, qt_ref, s_ref = ops.quant_fp8(act, 0, amax, True, True)
    assert torch.equal(pre2, pre)
    assert torch.equal(qa.q.view(torch.uint8), q_ref.view(torch.uint8))
    assert torch.equal(qa.t.view(torch.uint8), qt_ref.view(torch.uint8))
    assert qa.s.item() == s_ref.item() and rec.item() == amax.item()
    # a stale (smaller) amax saturates instead of overflowing
    half = amax * 0.5
    _, qs = F8.fwd_gelu_emit(xq, wq, b.cuda(), half, rec, False)
    assert qs.t is None and qs.q.float().abs().max().item() == 448.0 and torch.isfinite(qs.q.float()).all()
    # backward of the following linear: dh = (dy W2) * gelu'(pre), dy [M, N2], W2 [N2, N]
    N2 = 144
    dy, W2 = gen(M, N2, seed=14), gen(N2, N, seed=15, scale=0.05)
    dyq, w2q = F8.Q(dy.cuda(), 1, True, True), F8.Q(W2.cuda(), 0, True, True)
    dh = F8.dgrad(dyq, w2q, pre=pre)
    amax_dh = ops.absmax(dh)
    rec.zero_()
    cs = torch.zeros(N, device="cuda")
    qd = F8.dgrad_gelu_emit(dyq, w2q, pre, amax_dh, rec, cs)
    q_ref, qt_ref, s_ref = ops.quant_fp8(dh, 1, amax_dh, True, True)
    assert torch.equal(qd.q.view(torch.uint8), q_ref.view(torch.uint8))
    assert torch.equal(qd.t.view(torch.uint8), qt_ref.view(torch.uint8))
    assert qd.s.item() == s_ref.item() and rec.item() == amax_dh.item()
    want = dh.float().sum(0)
    assert ((cs - want).abs().max() / want.abs().max()).item() <= 1e-5


# ---- producers that emit fp8 themselves: bit-identical to vds_quant_fp8 of the plain kernel's bf16 result -----------
def _same_as_quant(ops, name, q, dq, rec, plain, fmt, old):
    """q / dq / recorded amax of an *_fp8 entry point against a quantisation pass over `plain` (bf16, on the GPU)"""
    rec2 = torch.zeros(1, device="cuda")
    q2, _, dq2 = ops.quant_fp8(plain, fmt, old, amax_out=rec2)
    assert torch.equal(q.view(torch.uint8), q2.view(torch.uint8)), name
    assert dq.item() == dq2.item() and rec.max().item() == rec2.item() == plain.float().abs().max().item(), name


@pytest.mark.parametrize("M,K", [(128, 128), (300, 144), (516, 1152), (4, 16)])
def test_transpose_fp8(ops, M, K):
    g = torch.Generator().manual_seed(M * K)
    q = torch.randint(0, 256, (M, K), generator=g, dtype=torch.uint8).cuda().view(torch.float8_e4m3fn)
    qt = ops.transpose_fp8(q)
    assert qt.shape == (K, M) and torch.equal(qt.view(torch.uint8), q.view(torch.uint8).t().contiguous())


@pytest.mark.parametrize("fmt", [0, 1])
@pytest.mark.parametrize("D,with_w", [(384, False), (1152, False), (1152, True), (2048, False)])
def test_rmsnorm_mod_fwd_emits_fp8(ops, fmt, D, with_w):
    B, L = 3, 75
    x = gen(B * L, D, seed=50).cuda()
    w = (1 + 0.1 * gen(D, seed=51).float()).to(bf16).cuda() if with_w else None
    mod = (gen(B, 9 * D, seed=52).float() * 0.5).cuda()
    y, rstd = ops.rmsnorm_mod_fwd(x, w, mod, 3 * D, 4 * D, B, L)
    old = y.float().abs().max().reshape(1) * 0.8  # an older, smaller amax: saturation is exercised
    rec = torch.zeros(B * L, device="cuda")
    q, dq, rstd2 = ops.rmsnorm_mod_fwd_fp8(x, w, mod, 3 * D, 4 * D, B, L, fmt, old, rec)
    assert torch.equal(rstd, rstd2)
    _same_as_quant(ops, "rmsnorm", q, dq, rec, y, fmt, old)


@pytest.mark.parametrize("fmt", [0, 1])
def test_gate_bwd_emits_fp8(ops, fmt):
    B, L, D = 2, 203, 1152
    dxn, y = gen(B * L, D, seed=60).cuda(), gen(B * L, D, seed=61).cuda()
    mod = gen(B, 9 * D, seed=62).float().cuda()
    dmod, dmod2 = (torch.zeros(B, 9 * D, dtype=f32, device="cuda") for _ in range(2))
    dbias, dbias2 = (torch.zeros(D, dtype=f32, device="cuda") for _ in range(2))
    dy = ops.gate_bwd(dxn, y, mod, 5 * D, dmod, dbias, B, L)
    old = dy.float().abs().max().reshape(1) * 0.9
    rec = torch.zeros(B * L, device="cuda")
    q, dq = ops.gate_bwd_fp8(dxn, y, mod, 5 * D, dmod2, dbias2, B, L, fmt, old, rec)
    _same_as_quant(ops, "gate_bwd", q, dq, rec, dy, fmt, old)
    assert torch.equal(dmod, dmod2) or (dmod - dmod2).abs().max().item() <= 1e-5 * dmod.abs().max().item()
    assert (dbias - dbias2).abs().max().item() <= 1e-5 * dbias.abs().max().item()


@pytest.mark.parametrize("hd,hdp,H", [(64, 64, 2), (72, 96, 16), (128, 128, 3)])
@pytest.mark.parametrize("mix", [True, False])
def test_qkv_rope_bwd_emits_fp8(ops, hd, hdp, H, mix):
    from oracle import dit_oracle as O
    B, thw = 2, (2, 4, 5)
    L = thw[0] * thw[1] * thw[2] + 16
    D = H * hd
    cos, sin = (t.cuda() for t in O.rope_cos_sin(hd, thw, (3, 7, 11)))
    qkv = gen(B * L, 3 * D, seed=70).cuda()
    v0 = gen(B, H, L, hdp, seed=71)
    v0[..., hd:] = 0
    v0 = v0.cuda()
    lam = torch.tensor([0.37]).to(bf16).cuda()
    dq, dk, dv = (gen(B, H, L, hdp, seed=s).cuda() for s in (72, 73, 74))
    outs = []
    for emit in (False, True):
        dv0 = torch.full((B, H, L, hdp), 0.25, dtype=f32, device="cuda")
        dlam = torch.zeros(1, dtype=f32, device="cuda")
        args = (dq, dk, dv, cos, sin, qkv if mix else None, v0 if mix else None, lam if mix else None, dv0,
                dlam if mix else None, mix, not mix, B, L, H, hd, hdp)
        if not emit:
            plain = ops.qkv_rope_bwd(*args)
            old = plain.float().abs().max().reshape(1) * 0.7
            outs.append((dv0, dlam))
        else:
            rec = torch.zeros(B * L, device="cuda")
            q, s = ops.qkv_rope_bwd_fp8(*args, 1, old, rec)
            _same_as_quant(ops, "qkv_rope_bwd", q, s, rec, plain, 1, old)
            assert torch.equal(dv0, outs[0][0])
            assert abs(dlam.item() - outs[0][1].item()) <= 1e-5 * max(1.0, abs(dlam.item()))
