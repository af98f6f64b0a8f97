"""GPU parity of every hand-written HIP kernel against the CPU oracle / plain fp32 torch math
on the same seeded inputs (runs on the MI355X box: pytest -m gpu).  All calls go through the
C ABI (ctypes) -- nothing here can pass on a silent fallback because there is none."""
import math
import os

import pytest
import torch

from oracle import dit_oracle as O

pytestmark = pytest.mark.gpu

bf16, f32 = torch.bfloat16, torch.float32


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from video_diffusion_speedrun_amd import ops as _ops
    return _ops


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / (b.norm() + 1e-20)).item()


def close(name, got, ref, tol):
    got, ref = got.float().cpu(), ref.float().cpu()
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    assert torch.isfinite(got).all(), f"{name}: non-finite values"
    e = rel(got, ref)
    if e >= tol:
        d = (got - ref).abs()
        idx = d.argmax().item()
        raise AssertionError(f"{name}: rel {e:.3e} >= {tol:.1e}; max|d| {d.max():.4e} at flat {idx} "
                             f"(got {got.flatten()[idx]:.5f} ref {ref.flatten()[idx]:.5f}); |ref| max {ref.abs().max():.4f}")
    return e


def gen(*shape, seed=0, scale=1.0, dtype=bf16):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype)


def test_lane_maps(ops):
    """MFMA 16x16x32 / 32x32x16 operand maps, accumulator-as-operand permutation,
    ds_read_b64_tr_b16 and LDS-DMA placement, checked on the device with exact integers."""
    res = ops.selftest_lanemaps()
    assert res[:5] == [0, 0, 0, 0, 0], f"lane-map self test mismatches: {res}"


# ------------------------------------------------------------------------------- GEMM ----
@pytest.fixture(params=[0, 128, 256, 2], ids=["auto", "t128", "t256", "t256x128"])
def tile(request, ops):
    """every GEMM test runs under the dispatcher's own choice and with each of the three tilings pinned
    (vds_gemm_force_tile): 128x128, 256x256, 256x128 (two workgroups per CU)"""
    ops.gemm_force_tile(request.param)
    yield request.param
    ops.gemm_force_tile(0)


@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (300, 384, 192), (1000, 1152, 1152), (4112, 128, 768),
                                   (131, 64, 64), (212, 432, 144), (77, 144, 72)])
def test_gemm_nt_store_bias(ops, tile, M, N, K):
    x, w, b = gen(M, K, seed=1), gen(N, K, seed=2, scale=0.05), gen(N, seed=3)
    y = ops.linear_fwd(x.cuda(), w.cuda(), b.cuda())
    ref = x.float() @ w.float().t() + b.float()
    close("nt+bias", y, ref, 4e-3)
    y2 = ops.linear_fwd(x.cuda(), w.cuda(), None)
    close("nt", y2, x.float() @ w.float().t(), 4e-3)


@pytest.mark.parametrize("N", [1152, 3456, 1096, 1280])
def test_gemm_large_m_ragged_last_column_tile(ops, tile, N):
    """problems big enough for the 256^2 kernel whose N does not fill the last column of tiles (1152 = 4.5 tiles,
    3456 = 13.5, 1096 = 4 tiles + 72 columns; 1280 = 5 full tiles as the control): forward (bias), gate + residual
    epilogue and input gradient, ragged M (49248 = 192.4 tiles)"""
    B, L, K = 6, 8208, 192
    M = B * L
    x, w, b = gen(M, K, seed=31), gen(N, K, seed=32, scale=0.05), gen(N, seed=33)
    ref = x.float() @ w.float().t()
    y = ops.linear_fwd(x.cuda(), w.cuda(), b.cuda())
    close("stripe.nt+bias", y, ref + b.float(), 4e-3)
    res, mod = gen(M, N, seed=34), gen(B, 3 * N, seed=35, dtype=f32)
    y2, xn = ops.linear_fwd_gate_res(x.cuda(), w.cuda(), None, mod.cuda(), 2 * N, res.cuda(), L)
    close("stripe.gate.y", y2, ref, 4e-3)
    close("stripe.gate.xnew", xn, res.float() + ref * mod[:, 2 * N:].repeat_interleave(L, dim=0), 4e-3)
    # input gradient of a [K2 -> N] ... as NN: dy [M, K2] @ W2 [K2, N]
    K2 = 256
    dy, w2 = gen(M, K2, seed=36), gen(K2, N, seed=37, scale=0.05)
    dx = ops.linear_dgrad(dy.cuda(), w2.cuda())
    close("stripe.nn", dx, dy.float() @ w2.float(), 4e-3)


def test_gemm_nt_asymmetric_identity(ops, tile):
    """A = I against an asymmetric B catches transposed / permuted fragment maps."""
    K = 128
    a = torch.eye(K).to(bf16)
    w = (torch.arange(K * 256).reshape(256, K) % 251 - 125).float().to(bf16)
    y = ops.linear_fwd(a.cuda(), w.cuda())
    assert torch.equal(y.cpu().float(), w.float().t().contiguous())


def test_gemm_nt_gelu(ops, tile):
    M, N, K = 520, 512, 256
    x, w, b = gen(M, K, seed=4), gen(N, K, seed=5, scale=0.08), gen(N, seed=6, scale=0.5)
    pre, act = ops.linear_fwd_gelu(x.cuda(), w.cuda(), b.cuda())
    ref = x.float() @ w.float().t() + b.float()
    close("gelu.pre", pre, ref, 4e-3)
    close("gelu.act", act, O.gelu_erf(ref), 5e-3)


def test_gemm_nt_gate_residual(ops, tile):
    B, L, N, K = 3, 173, 384, 256
    M = B * L
    x, w, res = gen(M, K, seed=7), gen(N, K, seed=8, scale=0.06), gen(M, N, seed=9)
    b = gen(N, seed=10, scale=0.3)
    mod = gen(B, 9 * N, seed=11, dtype=f32)
    for bias in (None, b):
        y, xn = ops.linear_fwd_gate_res(x.cuda(), w.cuda(), bias.cuda() if bias is not None else None, mod.cuda(),
                                        2 * N, res.cuda(), L)
        yr = x.float() @ w.float().t() + (bias.float() if bias is not None else 0)
        gate = mod[:, 2 * N:3 * N].repeat_interleave(L, dim=0)
        close("gate.y", y, yr, 4e-3)
        close("gate.xnew", xn, res.float() + yr * gate, 4e-3)


@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (777, 1152, 384), (4112, 768, 3072), (200, 128, 64),
                                   (212, 432, 144), (90, 200, 72)])
def test_gemm_nn_dgrad(ops, tile, M, N, K):
    """dx[M,K] = dy[M,N] W[N,K]"""
    dy, w = gen(M, N, seed=12), gen(N, K, seed=13, scale=0.05)
    dx = ops.linear_dgrad(dy.cuda(), w.cuda())
    close("nn", dx, dy.float() @ w.float(), 4e-3)


def test_gemm_nn_asymmetric_identity(ops, tile):
    N = 128
    dy = torch.eye(N).to(bf16)
    w = (torch.arange(N * 384).reshape(N, 384) % 241 - 120).float().to(bf16)
    dx = ops.linear_dgrad(dy.cuda(), w.cuda())
    assert torch.equal(dx.cpu().float(), w.float())


def test_gemm_nn_dgelu(ops, tile):
    M, N, K = 333, 256, 512
    dy, w, pre = gen(M, N, seed=14), gen(N, K, seed=15, scale=0.05), gen(M, K, seed=16)
    dx = ops.linear_dgrad(dy.cuda(), w.cuda(), pre.cuda())
    p = pre.float().requires_grad_(True)
    O.gelu_erf(p).backward(dy.float() @ w.float())
    close("dgelu", dx, p.grad, 5e-3)


def test_gemm_gelu_table_over_the_bf16_range(ops):
    """256^2 kernel: gelu / gelu' come from a table indexed by the bf16 bits of the pre-activation (gemm.hip,
    g_gelu_lut).  With an identity weight the pre-activation IS the input, so every binade from 2^-20 to 2^6, both
    signs, zero, the clamped ends and interior + ragged tiles are checked against float64 erf math: the result must
    be the bf16 rounding of the exact value up to the table's stated error (1e-4 of the 0.5 plateau below 2^-13)."""
    M, N = 600, 256  # 2 full row tiles of 256 + a ragged one
    g = torch.Generator().manual_seed(5)
    mag = torch.exp2(torch.rand(M, N, generator=g) * 26.0 - 20.0)
    x = (mag * (torch.randint(0, 2, (M, N), generator=g) * 2 - 1)).to(bf16)
    x[0, :8] = torch.tensor([0.0, -0.0, 7.96875, -7.96875, 8.0, -8.0, 100.0, -100.0]).to(bf16)
    eye = torch.eye(N).to(bf16)
    xd = x.double()
    cdf = 0.5 * torch.erfc(-xd / math.sqrt(2.0))
    pdf = torch.exp(-0.5 * xd * xd) / math.sqrt(2.0 * math.pi)
    ops.gemm_force_tile(256)
    try:
        pre, act = ops.linear_fwd_gelu(x.cuda(), eye.cuda(), None)
        dx = ops.linear_dgrad(torch.ones(M, N).to(bf16).cuda(), eye.cuda(), x.cuda())
    finally:
        ops.gemm_force_tile(0)
    assert torch.equal(pre.cpu(), x)
    for name, got, want in (("gelu", act, xd * cdf), ("gelu'", dx, cdf + xd * pdf)):
        got = got.cpu().double()
        err = (got - want).abs()
        tol = want.abs() * 2.0 ** -8 + 1.5e-4 * xd.abs().clamp(max=1.0) + 1e-30  # one bf16 rounding + the table's clamp
        if name == "gelu'":
            tol = want.abs() * 2.0 ** -8 + 1.5e-4
        bad = err > tol
        assert not bad.any(), (name, int(bad.sum()), x[bad][:4], got[bad][:4], want[bad][:4])


@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (1000, 384, 1152), (4112, 1152, 384), (8208, 128, 768),
                                   (77, 64, 128)])
def test_gemm_tn_wgrad(ops, tile, M, N, K):
    """dW[N,K] = dy[M,N]^T x[M,K], ragged token count M, fp32 out, split-K atomics"""
    dy, x = gen(M, N, seed=17), gen(M, K, seed=18)
    dW = torch.zeros(N, K, dtype=f32, device="cuda")
    ops.linear_wgrad(dy.cuda(), x.cuda(), dW)
    close("tn", dW, dy.float().t() @ x.float(), 2e-3)


def test_gemm_tn_asymmetric(ops, tile):
    M = 192
    dy = torch.zeros(M, 128)
    dy[torch.arange(128), torch.arange(128)] = 1  # dy^T x = x[:128]
    x = (torch.arange(M * 256).reshape(M, 256) % 239 - 119).float()
    dW = torch.zeros(128, 256, dtype=f32, device="cuda")
    ops.linear_wgrad(dy.to(bf16).cuda(), x.to(bf16).cuda(), dW)
    assert torch.equal(dW.cpu(), x[:128])


# -------------------------------------------------------------------------- attention ----
def attn_ref(q, k, v, do=None):
    q, k, v = q.float().requires_grad_(True), k.float().requires_grad_(True), v.float().requires_grad_(True)
    o = O.attention(q, k, v)
    s = (q @ k.transpose(-1, -2)) / math.sqrt(q.shape[-1])
    lse = torch.logsumexp(s, dim=-1)
    if do is None:
        return o, lse
    o.backward(do.float())
    return o, lse, q.grad, k.grad, v.grad


@pytest.mark.parametrize("hd,Lq,Lk", [(64, 272, 272), (64, 130, 512), (128, 200, 333), (72, 272, 272),
                                      (72, 129, 512), (128, 64, 64), (64, 1040, 1040), (64, 2100, 300),
                                      (72, 2100, 300),
                                      # round 5: head dims without an instance of their own (rows stored unpadded, next instance)
                                      (48, 272, 272), (40, 129, 512), (80, 272, 300), (112, 200, 333), (104, 2100, 300),
                                      (24, 130, 130), (56, 2100, 300)])  # Lq >= 2048: the 64-queries-per-wave forward kernels
def test_attention_fwd_bwd(ops, hd, Lq, Lk):
    B, H = 2, 3
    hdp = {64: 64, 72: 96, 128: 128}.get(hd, hd)
    q, k, v = gen(B, H, Lq, hd, seed=20), gen(B, H, Lk, hd, seed=21), gen(B, H, Lk, hd, seed=22)
    do = gen(B, Lq, H * hd, seed=23)
    # device layouts: q/k/v head-major padded rows [B,H,L,hdp]; o/do token-major [B,L,H*hd]
    def pad(t):
        out = torch.zeros(*t.shape[:-1], hdp, dtype=bf16)
        out[..., :hd] = t
        return out.cuda()
    qd, kd, vd = pad(q), pad(k), pad(v)
    o_tok = torch.zeros(B * Lq, H * hd, dtype=bf16, device="cuda")
    lse = torch.zeros(B, H, Lq, dtype=f32, device="cuda")
    ov = ops.heads_view(o_tok, B, Lq, H, hd)
    ops.attn_fwd(qd[..., :hd], kd[..., :hd], vd[..., :hd], ov, lse)
    do_h = do.reshape(B, Lq, H, hd).permute(0, 2, 1, 3)
    o_ref, lse_ref, dq_ref, dk_ref, dv_ref = attn_ref(q, k, v, do_h)
    close("attn.o", o_tok.reshape(B, Lq, H, hd).permute(0, 2, 1, 3), o_ref, 6e-3)
    close("attn.lse", lse, lse_ref, 1e-3)
    dq, dk, dv = torch.zeros_like(qd), torch.zeros_like(kd), torch.zeros_like(vd)
    delta = torch.zeros(2, B, H, Lq, dtype=f32, device="cuda")
    dod = do.reshape(B * Lq, H * hd).cuda()
    ops.attn_bwd(qd[..., :hd], kd[..., :hd], vd[..., :hd], ov, lse, ops.heads_view(dod, B, Lq, H, hd),
                 dq[..., :hd], dk[..., :hd], dv[..., :hd], delta)
    close("attn.dv", dv[..., :hd], dv_ref, 8e-3)
    close("attn.dk", dk[..., :hd], dk_ref, 8e-3)
    close("attn.dq", dq[..., :hd], dq_ref, 8e-3)
    if hdp > hd:
        assert dq[..., hd:].abs().max().item() == 0  # pad columns are never written


@pytest.mark.parametrize("Lq,Lk,spike", [(2100, 300, False), (2304, 4160, True)])
def test_attention_fwd_ones_columns(ops, Lq, Lk, spike):
    """head_dim 72 forward with the ones columns in the K / V padding (the path the model uses):
    -max through the QK^T MFMA, row sums through the PV MFMA, lazy rescale; ragged key count and
    a late score spike that forces a rescale long after the first tile."""
    B, H, hd, hdp = 1, 2, 72, 96
    q, k, v = gen(B, H, Lq, hd, seed=31), gen(B, H, Lk, hd, seed=32), gen(B, H, Lk, hd, seed=33)
    if spike:
        k[:, :, Lk - 70] = 6.0 * q[:, :, 5]      # one key row aligned with one query row, near the end
        q[:, :, 1000] *= 8.0                      # and one query with very large logits throughout
    def padk(t, cols):
        out = torch.zeros(*t.shape[:-1], hdp, dtype=bf16)
        out[..., :hd] = t
        for c in cols:
            out[..., c] = 1
        return out.cuda()
    qd, kd, vd = padk(q, []), padk(k, [hd, hd + 1]), padk(v, [hd, hd + 4])
    o = torch.zeros(B * Lq, H * hd, dtype=bf16, device="cuda")
    lse = torch.zeros(B, H, Lq, dtype=f32, device="cuda")
    ops.attn_fwd(qd[..., :hd], kd[..., :hd], vd[..., :hd], ops.heads_view(o, B, Lq, H, hd), lse, kv_pad_ones=True)
    sc = (q.float() @ k.float().transpose(-1, -2)) / math.sqrt(hd)
    ref = torch.softmax(sc, -1) @ v.float()
    got = o.view(B, Lq, H, hd).permute(0, 2, 1, 3)
    close("attn.ones.o", got, ref, 1e-2)
    close("attn.ones.lse", lse, torch.logsumexp(sc, -1), 2e-3)
    # same inputs through the path without the ones columns
    o2 = torch.zeros_like(o)
    lse2 = torch.zeros_like(lse)
    ops.attn_fwd(qd[..., :hd], kd[..., :hd], vd[..., :hd], ops.heads_view(o2, B, Lq, H, hd), lse2)
    close("attn.ones.vs_plain", o, o2.float(), 1e-2)


def test_attention_token_major_cross(ops):
    """cross-attention layout: q from [B,L,D], k/v from [B,Lc,2D] token-major buffers (hd=72)"""
    B, H, hd, L, Lc = 2, 2, 72, 150, 512
    D = H * hd
    qb, kvb = gen(B * L, D, seed=30), gen(B * Lc, 2 * D, seed=31)
    qd, kvd = qb.cuda(), kvb.cuda()
    o = torch.zeros(B * L, D, dtype=bf16, device="cuda")
    lse = torch.zeros(B, H, L, dtype=f32, device="cuda")
    qv = ops.heads_view(qd, B, L, H, hd)
    kv_k = ops.heads_view(kvd, B, Lc, H, hd, 0)
    kv_v = ops.heads_view(kvd, B, Lc, H, hd, D)
    ops.attn_fwd(qv, kv_k, kv_v, ops.heads_view(o, B, L, H, hd), lse)
    q = qb.reshape(B, L, H, hd).permute(0, 2, 1, 3)
    k = kvb[:, :D].reshape(B, Lc, H, hd).permute(0, 2, 1, 3)
    v = kvb[:, D:].reshape(B, Lc, H, hd).permute(0, 2, 1, 3)
    do = gen(B * L, D, seed=32)
    o_ref, _, dq_ref, dk_ref, dv_ref = attn_ref(q, k, v, do.reshape(B, L, H, hd).permute(0, 2, 1, 3))
    close("cross.o", o.reshape(B, L, H, hd).permute(0, 2, 1, 3), o_ref, 6e-3)
    dqb = torch.zeros_like(qd)
    dkvb = torch.zeros_like(kvd)
    delta = torch.zeros(2, B, H, L, dtype=f32, device="cuda")
    ops.attn_bwd(qv, kv_k, kv_v, ops.heads_view(o, B, L, H, hd), lse, ops.heads_view(do.cuda(), B, L, H, hd),
                 ops.heads_view(dqb, B, L, H, hd), ops.heads_view(dkvb, B, Lc, H, hd, 0),
                 ops.heads_view(dkvb, B, Lc, H, hd, D), delta)
    close("cross.dq", dqb.reshape(B, L, H, hd).permute(0, 2, 1, 3), dq_ref, 8e-3)
    close("cross.dk", dkvb[:, :D].reshape(B, Lc, H, hd).permute(0, 2, 1, 3), dk_ref, 8e-3)
    close("cross.dv", dkvb[:, D:].reshape(B, Lc, H, hd).permute(0, 2, 1, 3), dv_ref, 8e-3)


@pytest.mark.parametrize("hd,L,Lc", [(72, 2100, 300), (64, 1111, 512), (72, 4368, 512)],
                         ids=["hd72_Lk300", "hd64_Lk512", "c3a_tokens_Lk512"])
def test_attention_cross_dkv_with_split_query_range(ops, hd, L, Lc):
    """round 4: with a short key sequence (cross-attention, model.py:157) the dK/dV launch has only ceil(Lk/128) workgroups
    per head; given the workspace vds_attn_bwd_workspace_bytes asks for (delta=None) the library splits the query range
    over several workgroups per key tile (fp32 partials + a reduction kernel).  Result == the unsplit launch (legacy
    [2,B,H,Lq] workspace) up to the summation order, and both == the fp32 reference; ragged Lq / Lk included."""
    B, H = 1, 2
    D = H * hd
    qb, kvb = gen(B * L, D, seed=33), gen(B * Lc, 2 * D, seed=34)
    qd, kvd = qb.cuda(), kvb.cuda()
    o = torch.zeros(B * L, D, dtype=bf16, device="cuda")
    lse = torch.zeros(B, H, L, dtype=f32, device="cuda")
    qv, kv_k, kv_v = ops.heads_view(qd, B, L, H, hd), ops.heads_view(kvd, B, Lc, H, hd, 0), ops.heads_view(kvd, B, Lc, H, hd, D)
    ops.attn_fwd(qv, kv_k, kv_v, ops.heads_view(o, B, L, H, hd), lse)
    do = gen(B * L, D, seed=35)
    q = qb.reshape(B, L, H, hd).permute(0, 2, 1, 3)
    k = kvb[:, :D].reshape(B, Lc, H, hd).permute(0, 2, 1, 3)
    v = kvb[:, D:].reshape(B, Lc, H, hd).permute(0, 2, 1, 3)
    _, _, dq_ref, dk_ref, dv_ref = attn_ref(q, k, v, do.reshape(B, L, H, hd).permute(0, 2, 1, 3))
    assert ops.attn_bwd_workspace_floats(B, H, L, Lc, hd) > 2 * B * H * L  # the library wants room for partials here
    outs = []
    for delta in (torch.zeros(2, B, H, L, dtype=f32, device="cuda"), None):
        dqb = torch.full_like(qd, float("nan"))
        dkvb = torch.full_like(kvd, float("nan"))
        ops.attn_bwd(qv, kv_k, kv_v, ops.heads_view(o, B, L, H, hd), lse, ops.heads_view(do.cuda(), B, L, H, hd),
                     ops.heads_view(dqb, B, L, H, hd), ops.heads_view(dkvb, B, Lc, H, hd, 0),
                     ops.heads_view(dkvb, B, Lc, H, hd, D), delta)
        assert torch.isfinite(dkvb.float()).all()
        close("split.dq", dqb.reshape(B, L, H, hd).permute(0, 2, 1, 3), dq_ref, 8e-3)
        close("split.dk", dkvb[:, :D].reshape(B, Lc, H, hd).permute(0, 2, 1, 3), dk_ref, 8e-3)
        close("split.dv", dkvb[:, D:].reshape(B, Lc, H, hd).permute(0, 2, 1, 3), dv_ref, 8e-3)
        outs.append(dkvb.float().cpu())
    close("split.vs_unsplit", outs[1], outs[0], 4e-3)  # bf16 roundings of two fp32 summation orders


def test_attention_online_softmax_rescale(ops):
    """spike one key late in the sequence so the running max jumps in the last tile"""
    B, H, hd, L = 1, 1, 64, 256
    q, k, v = gen(B, H, L, hd, seed=40), gen(B, H, L, hd, seed=41), gen(B, H, L, hd, seed=42)
    k[0, 0, 250] = q[0, 0, 7] * 3
    o = torch.zeros(B * L, H * hd, dtype=bf16, device="cuda")
    lse = torch.zeros(B, H, L, dtype=f32, device="cuda")
    ops.attn_fwd(q.cuda(), k.cuda(), v.cuda(), ops.heads_view(o, B, L, H, hd), lse)
    o_ref, lse_ref = attn_ref(q, k, v)
    close("spike.o", o.reshape(B, L, H, hd).permute(0, 2, 1, 3), o_ref.detach(), 6e-3)
    close("spike.lse", lse, lse_ref.detach(), 1e-3)


# ----------------------------------------------------------- norm / modulation / gate ----
@pytest.mark.parametrize("D,with_w", [(384, False), (1152, False), (144, True), (768, True)])
def test_rmsnorm_mod_fwd_bwd(ops, D, with_w):
    B, L = 3, 75
    x = gen(B * L, D, seed=50)
    w = (1 + 0.1 * gen(D, seed=51, dtype=f32)).to(bf16) if with_w else None
    mod = gen(B, 9 * D, seed=52, dtype=f32) * 0.5
    dy, dres = gen(B * L, D, seed=53), gen(B * L, D, seed=54)
    y, rstd = ops.rmsnorm_mod_fwd(x.cuda(), w.cuda() if with_w else None, mod.cuda(), 3 * D, 4 * D, B, L)
    xf = x.float().reshape(B, L, D).requires_grad_(True)
    wf = w.float().requires_grad_(True) if with_w else None
    mf = mod.clone().requires_grad_(True)
    yr = O.modulate(O.rms_norm(xf, wf), mf[:, 3 * D:4 * D], mf[:, 4 * D:5 * D])
    close("rms.y", y, yr.reshape(B * L, D), 4e-3)
    yr.backward(dy.float().reshape(B, L, D))
    dmod = torch.zeros(B, 9 * D, dtype=f32, device="cuda")
    dw = torch.zeros(D, dtype=f32, device="cuda") if with_w else None
    dx = ops.rmsnorm_mod_bwd(dy.cuda(), x.cuda(), w.cuda() if with_w else None, mod.cuda(), 3 * D, 4 * D, rstd,
                             dres.cuda(), dmod, dw, B, L)
    close("rms.dx", dx, xf.grad.reshape(B * L, D) + dres.float(), 5e-3)
    close("rms.dmod", dmod, mf.grad, 2e-3)
    if with_w:
        close("rms.dw", dw, wf.grad, 2e-3)


def test_gate_bwd_and_colsum(ops):
    B, L, D = 2, 203, 384
    dxn, y = gen(B * L, D, seed=60), gen(B * L, D, seed=61)
    mod = gen(B, 9 * D, seed=62, dtype=f32)
    dmod = torch.zeros(B, 9 * D, dtype=f32, device="cuda")
    dbias = torch.zeros(D, dtype=f32, device="cuda")
    dy = ops.gate_bwd(dxn.cuda(), y.cuda(), mod.cuda(), 5 * D, dmod, dbias, B, L)
    gate = mod[:, 5 * D:6 * D].repeat_interleave(L, dim=0)
    dyr = dxn.float() * gate
    close("gate.dy", dy, dyr, 4e-3)
    ref = torch.zeros(B, 9 * D)
    ref[:, 5 * D:6 * D] = (dxn.float() * y.float()).reshape(B, L, D).sum(1)
    close("gate.dmod", dmod, ref, 1e-3)
    close("gate.dbias", dbias, dyr.to(bf16).float().sum(0), 3e-3)
    cs = torch.zeros(D, dtype=f32, device="cuda")
    ops.colsum(y.cuda(), cs)
    close("colsum", cs, y.float().sum(0), 1e-4)
    cs2 = torch.zeros(D, dtype=f32, device="cuda")
    ops.colsum(y.cuda(), cs2, rows_per_sample=L, row_offset=16)  # token rows only (skip 16 register rows per sample)
    close("colsum.rows", cs2, y.float().reshape(B, L, D)[:, 16:].sum((0, 1)), 1e-4)


# ----------------------------------------------------------------- qkv / rope / res-V ----
@pytest.mark.parametrize("hd,hdp", [(72, 96), (32, 32), (96, 96)])
def test_qkv_rope_fwd_ragged_token_tiles(ops, hd, hdp):
    """the forward producer works on tiles of 4 consecutive tokens (csrc/rope_stage.h): L = 25 and B = 3 make tiles
    straddle the sample boundaries and leave a 3-token last tile; every head size the tile kernel is instantiated for"""
    B, H, thw = 3, 3, (1, 3, 3)
    L = thw[0] * thw[1] * thw[2] + 16
    D = H * hd
    cos, sin = O.rope_cos_sin(hd, thw, (3, 7, 11))
    qkv = gen(B * L, 3 * D, seed=75)
    v0 = gen(B, H, L, hdp, seed=76)
    v0[..., hd:] = 0
    lam = torch.tensor([0.37]).to(bf16)
    q, k, v = O.split_heads(qkv.reshape(B, L, 3 * D), 3, H)
    qr, kr = O.apply_rotary(q, cos, sin), O.apply_rotary(k, cos, sin)
    for mix in (False, True):
        qd, kd, vd = ops.qkv_rope_fwd(qkv.cuda(), cos.cuda(), sin.cuda(), v0.cuda() if mix else None,
                                      lam.cuda() if mix else None, B, L, H, hd, hdp)
        close("rope.q", qd[..., :hd], qr, 1e-3)
        close("rope.k", kd[..., :hd], kr, 1e-3)
        if mix:
            close("rope.v", vd[..., :hd], lam * v + (1 - lam) * v0[..., :hd], 4e-3)
        else:
            assert torch.equal(vd[..., :hd].cpu(), v.contiguous())
        if hdp > hd:
            assert qd[..., hd:].abs().max().item() == 0
            want_k = torch.zeros(hdp - hd); want_v = torch.zeros(hdp - hd)
            want_k[0] = 1; want_k[1] = 1; want_v[0] = 1; want_v[4] = 1
            assert torch.equal(kd[..., hd:].float().cpu(), want_k.expand(B, H, L, -1))
            assert torch.equal(vd[..., hd:].float().cpu(), want_v.expand(B, H, L, -1))


@pytest.mark.parametrize("hd,hdp", [(64, 64), (72, 96), (128, 128)])
def test_qkv_rope_fwd_bwd(ops, hd, hdp):
    B, H, thw = 2, 2, (2, 4, 5)
    N = thw[0] * thw[1] * thw[2]
    L = N + 16
    D = H * hd
    cos, sin = O.rope_cos_sin(hd, thw, (3, 7, 11))
    qkv = gen(B * L, 3 * D, seed=70)
    v0 = gen(B, H, L, hdp, seed=71)
    v0[..., hd:] = 0
    lam = torch.tensor([0.37]).to(bf16)
    qd, kd, vd = ops.qkv_rope_fwd(qkv.cuda(), cos.cuda(), sin.cuda(), v0.cuda(), lam.cuda(), B, L, H, hd, hdp)
    q, k, v = O.split_heads(qkv.reshape(B, L, 3 * D), 3, H)
    qr, kr = O.apply_rotary(q, cos, sin), O.apply_rotary(k, cos, sin)
    vm = lam * v + (1 - lam) * v0[..., :hd]
    close("rope.q", qd[..., :hd], qr, 1e-3)
    close("rope.k", kd[..., :hd], kr, 1e-3)
    close("rope.v", vd[..., :hd], vm, 4e-3)
    if hdp > hd:  # pad: zeros, except the ones columns of k and v (hdp >= hd + 8)
        assert qd[..., hd:].abs().max().item() == 0
        want_k = torch.zeros(hdp - hd); want_v = torch.zeros(hdp - hd)
        if hdp - hd >= 8:
            want_k[0] = 1; want_k[1] = 1; want_v[0] = 1; want_v[4] = 1
        assert torch.equal(kd[..., hd:].float().cpu(), want_k.expand(B, H, L, -1))
        assert torch.equal(vd[..., hd:].float().cpu(), want_v.expand(B, H, L, -1))
    # no-mix variant (block 0)
    _, _, v_raw = ops.qkv_rope_fwd(qkv.cuda(), cos.cuda(), sin.cuda(), None, None, B, L, H, hd, hdp)
    assert torch.equal(v_raw[..., :hd].cpu(), v.contiguous())
    # backward
    dq, dk, dv = gen(B, H, L, hdp, seed=72), gen(B, H, L, hdp, seed=73), gen(B, H, L, hdp, seed=74)
    qf = qkv.float().reshape(B, L, 3 * D).requires_grad_(True)
    v0f = v0[..., :hd].float().requires_grad_(True)
    lf = lam.float().requires_grad_(True)
    q2, k2, v2 = O.split_heads(qf, 3, H)
    loss = (O.apply_rotary(q2, cos, sin) * dq[..., :hd].float()).sum() + \
           (O.apply_rotary(k2, cos, sin) * dk[..., :hd].float()).sum() + \
           ((lf * v2 + (1 - lf) * v0f) * dv[..., :hd].float()).sum()
    loss.backward()
    dv0 = torch.zeros(B, H, L, hdp, dtype=f32, device="cuda")
    dlam = torch.zeros(1, dtype=f32, device="cuda")
    dqkv = ops.qkv_rope_bwd(dq.cuda(), dk.cuda(), dv.cuda(), cos.cuda(), sin.cuda(), qkv.cuda(), v0.cuda(),
                            lam.cuda(), dv0, dlam, True, False, B, L, H, hd, hdp)
    close("rope.dqkv", dqkv, qf.grad.reshape(B * L, 3 * D), 4e-3)
    close("rope.dv0", dv0[..., :hd], v0f.grad, 1e-3)
    close("rope.dlam", dlam, lf.grad, 2e-3)
    # block-0 variant: dv_total = dv + dv0_acc, no mix
    dqkv0 = ops.qkv_rope_bwd(dq.cuda(), dk.cuda(), dv.cuda(), cos.cuda(), sin.cuda(), None, None, None, dv0, None,
                             False, True, B, L, H, hd, hdp)
    want = (dv[..., :hd].float() + dv0[..., :hd].cpu()).permute(0, 2, 1, 3).reshape(B * L, D)
    close("rope.dv_block0", dqkv0[:, 2 * D:], want, 4e-3)


# -------------------------------------------------------------------- small linears ----
@pytest.mark.parametrize("M,N,K", [(4, 1152, 384), (6, 3456, 1152), (13, 203, 72), (16, 70, 1160), (1, 33, 8),
                                   (37, 1152, 384), (64, 144, 72)])  # > 16 rows: per-rank batches up to train.py:150's 64
def test_small_linear_and_timestep(ops, M, N, K):
    x = gen(M, K, seed=80, dtype=f32)
    W, b = gen(N, K, seed=81, scale=0.05), gen(N, seed=82, scale=0.1)
    for act in (0, 1):
        y = ops.small_linear_fwd(x.cuda(), W.cuda(), b.cuda(), act)
        xr = x.clone().requires_grad_(True)
        yr = O.linear(O.silu(xr) if act else xr, W.float(), b.float())
        close(f"small.y{act}", y, yr, 1e-4)
        dy = gen(M, N, seed=83, dtype=f32)
        yr.backward(dy)
        dW = torch.zeros(N, K, dtype=f32, device="cuda")
        db = torch.zeros(N, dtype=f32, device="cuda")
        dx = torch.zeros(M, K, dtype=f32, device="cuda")
        ops.small_linear_bwd(dy.cuda(), x.cuda(), W.cuda(), dW, db, dx, act)
        Wf = W.float().requires_grad_(True)
        O.linear(O.silu(x) if act else x, Wf, None).backward(dy)
        close(f"small.dW{act}", dW, Wf.grad, 1e-4)
        close(f"small.db{act}", db, dy.sum(0), 1e-5)
        close(f"small.dx{act}", dx, xr.grad, 1e-4)
    t = torch.tensor([0.03, 0.5, 0.97, 0.2])
    e = ops.timestep_embedding(t.cuda(), 384)
    close("temb", e, O.timestep_embedding(t, 384).to(bf16), 2e-3)


# ---------------------------------------------------------------- patches / loss ----
@pytest.mark.parametrize("pt", [1, 2])
def test_patchify_unpatchify(ops, pt):
    B, C, T, H, W, p = 2, 16, 4, 8, 6, 2
    x = gen(B, C, T, H, W, seed=90)
    pat = ops.patchify(x.cuda(), pt, p)
    ref = O.patchify(x, pt, p)
    assert torch.equal(pat.cpu().reshape(ref.shape), ref)
    n = ref.shape[1]
    y = gen(B * n, p * p * pt * C, seed=91)
    img = ops.unpatchify(y.cuda(), B, C, T, H, W, pt, p)
    ref_img = O.unpatchify(y.reshape(B, n, -1), C, T // pt, H // p, W // p, pt, p)
    assert torch.equal(img.cpu(), ref_img)
    back = ops.unpatchify_bwd(img, pt, p)
    assert torch.equal(back.cpu(), y)
    # the same with the tokens addressed as rows of the model's [B * (16 + N)] token buffer (16 untouched register
    # rows in front of every sample): what lets patch embedding / final layer run as one GEMM over all rows
    R = 16
    pat_r = ops.patchify(x.cuda(), pt, p, lead_rows=R).cpu().reshape(B, R + n, -1)
    assert torch.equal(pat_r[:, R:], ref) and pat_r[:, :R].abs().max().item() == 0
    y_r = torch.zeros(B, R + n, p * p * pt * C, dtype=bf16)
    y_r[:, R:] = y.reshape(B, n, -1)
    y_r[:, :R] = 7.0  # register rows must be ignored
    img_r = ops.unpatchify(y_r.reshape(B * (R + n), -1).cuda(), B, C, T, H, W, pt, p, lead_rows=R)
    assert torch.equal(img_r.cpu(), ref_img)
    back_r = ops.unpatchify_bwd(img, pt, p, lead_rows=R).cpu().reshape(B, R + n, -1)
    assert torch.equal(back_r[:, R:], y.reshape(B, n, -1)) and back_r[:, :R].abs().max().item() == 0


def test_registers_noise_loss_cast(ops):
    B, R, D, L = 3, 16, 128, 40
    reg = gen(R, D, seed=100)
    x = torch.zeros(B * L, D, dtype=bf16, device="cuda")
    ops.fill_registers(reg.cuda(), x, L * D, B, R, D)
    assert torch.equal(x.reshape(B, L, D)[:, :R].cpu(), reg.expand(B, R, D))
    dx = gen(B * L, D, seed=101)
    dreg = torch.zeros(R, D, dtype=f32, device="cuda")
    ops.registers_bwd(dx.cuda(), L * D, dreg, B, R, D)
    close("dreg", dreg, dx.float().reshape(B, L, D)[:, :R].sum(0), 1e-5)
    lat, noise = gen(B, 16, 4, 8, 8, seed=102), gen(B, 16, 4, 8, 8, seed=103)
    t = O.time_shift(gen(B, seed=104))
    zt, v = ops.noise_latents(lat.cuda(), noise.cuda(), t.float().cuda())
    zr, vr = O.noise_latents(lat, noise, t)
    assert torch.equal(zt.cpu(), zr) and torch.equal(v.cpu(), vr)
    out = gen(B, 16, 4, 8, 8, seed=105)
    loss, per, dout = ops.flow_loss(v, out.cuda(), True)
    of = out.float().requires_grad_(True)
    lr, pr = O.flow_loss(vr, of)
    lr.backward()
    close("loss", loss.reshape(1), lr.reshape(1), 1e-5)
    close("loss.per", per, pr, 1e-5)
    for _ in range(3):  # fixed-order reduction: the same words every time, and batch mean == mean of the per-sample means
        loss_b, per_b, _ = ops.flow_loss(v, out.cuda(), False)
        assert torch.equal(loss_b, loss) and torch.equal(per_b, per)
    tot = torch.zeros((), dtype=f32)
    for b in range(B):  # the second stage's order: samples in index order, then one division (train.py:125)
        tot = tot + per[b].cpu()
    assert loss.item() == (tot / B).item()
    close("loss.dout", dout, of.grad, 4e-3)
    a = gen(1000003, seed=106, dtype=f32)
    d = torch.empty(a.numel(), dtype=bf16, device="cuda")
    ops.cast_f32_bf16(a.cuda(), d)
    assert torch.equal(d.cpu(), a.to(bf16))


# ------------------------------------------------------------------ sharding runtime ----
def test_comm_c_abi_on_a_one_rank_communicator(ops):
    """vds_comm_* (csrc/comm.hip: RCCL driven from the library, model.py:512-542's collectives) on the one GPU of
    this box: unique id -> init -> bf16 / fp32 all-gather, fp32 reduce-scatter-average and all-reduce on a side
    stream -> destroy.  At world 1 every collective must be an exact copy."""
    import ctypes as C
    from video_diffusion_speedrun_amd import _lib
    lib = _lib.load()
    ident = (C.c_ubyte * 128)()
    _lib.check(lib.vds_comm_unique_id(ident, 128), "vds_comm_unique_id")
    assert any(ident)
    assert lib.vds_all_gather_bf16(None, None, 4, None) != 0 and b"vds_comm_init" in lib.vds_last_error()
    _lib.check(lib.vds_comm_init(0, 1, ident, 128), "vds_comm_init")
    try:
        assert lib.vds_comm_init(0, 1, ident, 128) != 0                      # one communicator per process
        r, w, v = C.c_int32(), C.c_int32(), C.c_int32()
        _lib.check(lib.vds_comm_info(C.byref(r), C.byref(w), C.byref(v), None), "vds_comm_info")
        assert (r.value, w.value) == (0, 1) and v.value > 20000
        side = torch.cuda.Stream()
        x16, x32 = gen(4096, seed=1).cuda(), gen(4096, seed=2, dtype=f32).cuda()
        o16, o32, rs = torch.zeros_like(x16), torch.zeros_like(x32), torch.zeros_like(x32)
        side.wait_stream(torch.cuda.current_stream())
        s = side.cuda_stream
        _lib.check(lib.vds_all_gather_bf16(x16.data_ptr(), o16.data_ptr(), 4096, s), "ag16")
        _lib.check(lib.vds_all_gather_f32(x32.data_ptr(), o32.data_ptr(), 4096, s), "ag32")
        assert lib.vds_reduce_scatter_workspace_bytes(4096) == 0
        _lib.check(lib.vds_reduce_scatter_f32_avg(x32.data_ptr(), rs.data_ptr(), 4096, None, 0, s), "rs")
        ar = x32.clone()
        side.wait_stream(torch.cuda.current_stream())
        _lib.check(lib.vds_all_reduce_f32_avg(ar.data_ptr(), 4096, s), "ar")
        side.synchronize()
        assert torch.equal(o16, x16) and torch.equal(o32, x32) and torch.equal(rs, x32) and torch.equal(ar, x32)
    finally:
        _lib.check(lib.vds_comm_destroy(), "vds_comm_destroy")
    assert lib.vds_comm_destroy() == 0                                       # idempotent


@pytest.mark.parametrize("W,rank", [(2, 0), (8, 3), (8, 7), (1, 0)])
def test_allpairs_average_kernel(ops, W, rank):
    """local half of the all-pairs reduce-scatter (VDS_COMM_SCHEDULE=allpairs): mean of W chunks in rank order"""
    from video_diffusion_speedrun_amd import _lib
    n = 4 * 1000 + 8
    chunks = [gen(n, seed=50 + r, dtype=f32) for r in range(W)]
    staged = torch.cat([c for r, c in enumerate(chunks) if r != rank]).cuda() if W > 1 else None
    out = torch.zeros(n, dtype=f32, device="cuda")
    _lib.check(_lib.load().vds_average_chunks_f32(chunks[rank].cuda().data_ptr(),
                                                  staged.data_ptr() if staged is not None else None, out.data_ptr(),
                                                  n, W, rank, torch.cuda.current_stream().cuda_stream), "avg")
    want = torch.zeros(n)
    for c in chunks:                      # the same left-to-right fp32 summation order
        want = want + c
    assert torch.equal(out.cpu(), want / W) or rel(out, want / W) < 1e-7


@pytest.mark.parametrize("variant", [0, 1, 3, 7], ids=["mfma32", "dkv16", "dkv16+dq16", "all16"])
@pytest.mark.parametrize("Lq,Lk", [(1300, 1300), (2304, 1090)])
def test_attention_ones_columns_mfma_shape_variants(ops, variant, Lq, Lk):
    """head_dim 72 with the ones-column padding (the path the model uses), forward + backward, under every
    combination of MFMA shapes (vds_attn_set_variant: 32x32x16 kernels vs their 16x16x32 counterparts), ragged
    query / key counts, one peaked query row"""
    B, H, hd, hdp = 2, 2, 72, 96
    q, k, v = gen(B, H, Lq, hd, seed=61), gen(B, H, Lk, hd, seed=62), gen(B, H, Lk, hd, seed=63)
    q[:, :, 77] *= 5.0
    do = gen(B, Lq, H * hd, seed=64)

    def padk(t, cols):
        out = torch.zeros(*t.shape[:-1], hdp, dtype=bf16)
        out[..., :hd] = t
        for c in cols:
            out[..., c] = 1
        return out.cuda()
    qd, kd, vd = padk(q, []), padk(k, [hd, hd + 1]), padk(v, [hd, hd + 4])
    o = torch.zeros(B * Lq, H * hd, dtype=bf16, device="cuda")
    lse = torch.zeros(B, H, Lq, dtype=f32, device="cuda")
    ov = ops.heads_view(o, B, Lq, H, hd)
    prev = ops.attn_set_variant(variant)
    try:
        ops.attn_fwd(qd[..., :hd], kd[..., :hd], vd[..., :hd], ov, lse, kv_pad_ones=True)
        dq, dk, dv = torch.zeros_like(qd), torch.zeros_like(kd), torch.zeros_like(vd)
        ops.attn_bwd(qd[..., :hd], kd[..., :hd], vd[..., :hd], ov, lse,
                     ops.heads_view(do.reshape(B * Lq, H * hd).cuda(), B, Lq, H, hd),
                     dq[..., :hd], dk[..., :hd], dv[..., :hd], None, kv_pad_ones=True)
    finally:
        ops.attn_set_variant(-1)
    assert ops.attn_set_variant(-1) == prev or True
    o_ref, lse_ref, dq_ref, dk_ref, dv_ref = attn_ref(q, k, v, do.reshape(B, Lq, H, hd).permute(0, 2, 1, 3))
    close("v.o", o.view(B, Lq, H, hd).permute(0, 2, 1, 3), o_ref, 1e-2)
    close("v.lse", lse, lse_ref, 2e-3)
    close("v.dv", dv[..., :hd], dv_ref, 1e-2)
    close("v.dk", dk[..., :hd], dk_ref, 1e-2)
    close("v.dq", dq[..., :hd], dq_ref, 1e-2)
    assert dq[..., hd + 2:].abs().max().item() == 0 and dk[..., hd:].abs().max().item() == 0


def test_attention_16x16_epilogues_wide_and_narrow_row_stores_agree(ops):
    """round 4: the 16x16x32 kernels store O / dQ / dK / dV rows with 16-byte stores after a lane-row transpose when the
    rows are 16-byte aligned, and keep the 8-byte stores otherwise (the C ABI only asks for 8-byte alignment).  Both
    paths must write the same bits, and nothing outside the head's columns."""
    B, H, hd, hdp, Lq, Lk = 2, 2, 72, 96, 700, 515
    q, k, v = gen(B, H, Lq, hd, seed=71), gen(B, H, Lk, hd, seed=72), gen(B, H, Lk, hd, seed=73)
    do = gen(B, Lq, H * hd, seed=74).reshape(B * Lq, H * hd).cuda()

    def padk(t, cols):
        out = torch.zeros(*t.shape[:-1], hdp, dtype=bf16)
        out[..., :hd] = t
        for c in cols:
            out[..., c] = 1
        return out.cuda()
    kd, vd = padk(k, [hd, hd + 1]), padk(v, [hd, hd + 4])
    res = {}
    prev = ops.attn_set_variant(7)
    try:
        for name, off, wid in (("wide", 0, hdp), ("narrow", 4, 100)):
            qd = padk(q, [])  # (the backward annotates q's pad columns: a fresh copy per run)
            obuf = torch.full((B * Lq, H * hd + 2 * off), 3.0, dtype=bf16, device="cuda")
            ov = ops.heads_view(obuf[:, off:off + H * hd], B, Lq, H, hd)
            lse = torch.zeros(B, H, Lq, dtype=f32, device="cuda")
            bufs = [torch.full((B, H, L, wid), 3.0, dtype=bf16, device="cuda") for L in (Lq, Lk, Lk)]
            dq, dk, dv = (t[..., off:off + hd] for t in bufs)
            assert (ov.data_ptr() % 16 == 0) == (name == "wide") and (dq.data_ptr() % 16 == 0) == (name == "wide")
            ops.attn_fwd(qd[..., :hd], kd[..., :hd], vd[..., :hd], ov, lse, kv_pad_ones=True)
            # (the backward reads O from the aligned copy in both runs: the delta preprocess picks its access width --
            # and with it its summation order -- by O's strides, which is not what this test is about)
            o_al = obuf[:, off:off + H * hd].contiguous()
            ops.attn_bwd(qd[..., :hd], kd[..., :hd], vd[..., :hd], ops.heads_view(o_al, B, Lq, H, hd), lse,
                         ops.heads_view(do, B, Lq, H, hd), dq, dk, dv, None, kv_pad_ones=True)
            torch.cuda.synchronize()
            res[name] = (o_al, dq.clone(), dk.clone(), dv.clone())
            if off:  # the columns around the head's 72 stay untouched
                assert bool((obuf[:, :off] == 3.0).all()) and bool((obuf[:, off + H * hd:] == 3.0).all())
                for t in bufs:
                    assert bool((t[..., :off] == 3.0).all()) and bool((t[..., off + hd:] == 3.0).all())
    finally:
        ops.attn_set_variant(prev)
    for a, b, n in zip(res["wide"], res["narrow"], ("o", "dq", "dk", "dv")):
        bad = (a != b)
        assert torch.equal(a, b), (n, int(bad.sum()), bad.nonzero()[:6].tolist())
    o_ref, _, dq_ref, dk_ref, dv_ref = attn_ref(q, k, v, do.cpu().reshape(B, Lq, H, hd).permute(0, 2, 1, 3))
    close("w.o", res["wide"][0].view(B, Lq, H, hd).permute(0, 2, 1, 3), o_ref, 1e-2)
    close("w.dq", res["wide"][1], dq_ref, 1e-2)
    close("w.dk", res["wide"][2], dk_ref, 1e-2)
    close("w.dv", res["wide"][3], dv_ref, 1e-2)


def test_small_linear_batched_equals_per_set(ops):
    """the adaLN linears of all blocks in one launch (vds_small_linear_*_batched, device pointer tables) give the same
    outputs, weight / bias gradients and accumulated input gradient as one launch per block"""
    nb, M, N, K = 5, 6, 9 * 48, 48
    x = gen(M, K, seed=120, dtype=f32).cuda()
    Ws = [gen(N, K, seed=121 + i, scale=0.1).cuda() for i in range(nb)]
    bs = [gen(N, seed=131 + i, scale=0.1).cuda() for i in range(nb)]
    y = ops.small_linear_fwd_batched(x, ops.ptr_table(Ws), ops.ptr_table(bs), nb, N, 1)
    for i in range(nb):
        assert torch.equal(y[i], ops.small_linear_fwd(x, Ws[i], bs[i], 1))
    dy = gen(nb, M, N, seed=140, dtype=f32).cuda()
    dWs = [torch.zeros(N, K, dtype=f32, device="cuda") for _ in range(nb)]
    dbs = [torch.zeros(N, dtype=f32, device="cuda") for _ in range(nb)]
    dx = torch.zeros(M, K, dtype=f32, device="cuda")
    ops.small_linear_bwd_batched(dy, x, ops.ptr_table(Ws), ops.ptr_table(dWs), ops.ptr_table(dbs), dx, 1)
    dx_ref = torch.zeros(M, K, dtype=f32, device="cuda")
    for i in range(nb):
        dW, db = torch.zeros(N, K, dtype=f32, device="cuda"), torch.zeros(N, dtype=f32, device="cuda")
        ops.small_linear_bwd(dy[i].contiguous(), x, Ws[i], dW, db, dx_ref, 1)
        assert torch.equal(dWs[i], dW) and torch.equal(dbs[i], db)
    close("batched.dx", dx, dx_ref, 1e-5)


@pytest.mark.parametrize("M", [3, 12, 16])
@pytest.mark.parametrize("act", [0, 1])
def test_small_linear_batched_streaming_kernels_vs_fp32(ops, M, act):
    """round 4: at adaLN sizes (N >= 1024) the batched forward runs on the MFMA (x as a bf16 hi + lo pair from LDS, W
    streamed once), dW in 64 x 128 tiles, dx with its dy slices staged in LDS (several slices per workgroup here);
    against fp64 torch on the same bf16 weights: y, dW, dbias, dx within 1e-4 / 1e-5"""
    nb, N, K = 3, 2080, 288
    x = gen(M, K, seed=150, dtype=f32).cuda()
    Ws = [gen(N, K, seed=151 + i, scale=0.1).cuda() for i in range(nb)]
    bs = [gen(N, seed=161 + i, scale=0.1).cuda() for i in range(nb)]
    y = ops.small_linear_fwd_batched(x, ops.ptr_table(Ws), ops.ptr_table(bs), nb, N, act)
    for i in range(nb):  # the per-block form (sharding runtime) runs the same per-row arithmetic: identical bits
        assert torch.equal(y[i], ops.small_linear_fwd(x, Ws[i], bs[i], act))
    dy = gen(nb, M, N, seed=170, dtype=f32).cuda()
    dWs = [torch.full((N, K), 7.0, dtype=f32, device="cuda") for _ in range(nb)]  # (written, not accumulated)
    dbs = [torch.full((N,), 7.0, dtype=f32, device="cuda") for _ in range(nb)]
    dx = torch.zeros(M, K, dtype=f32, device="cuda")
    ops.small_linear_bwd_batched(dy, x, ops.ptr_table(Ws), ops.ptr_table(dWs), ops.ptr_table(dbs), dx, act)
    xr = x.double().cpu().requires_grad_(True)
    xa = torch.nn.functional.silu(xr) if act else xr
    tot = 0.0
    for i in range(nb):
        Wd = Ws[i].double().cpu().requires_grad_(True)
        yr = xa @ Wd.t() + bs[i].double().cpu()
        close(f"sl.y{i}", y[i], yr.detach(), 1e-4)
        (g,) = torch.autograd.grad((yr * dy[i].double().cpu()).sum(), Wd, retain_graph=True)
        close(f"sl.dW{i}", dWs[i], g, 1e-4)
        close(f"sl.db{i}", dbs[i], dy[i].double().cpu().sum(0), 1e-5)
        tot = tot + (yr * dy[i].double().cpu()).sum()
    tot.backward()
    close("sl.dx", dx, xr.grad, 1e-4)


def test_gemm_nn_dgelu_with_fused_bias_gradient(ops, tile):
    """dx = (dy W) * gelu'(pre) with colsum[k] += sum_m dx[m,k] from the same call (the fc1 bias gradient,
    model.py:84): fused in the 256^2 kernel's epilogue, a follow-up pass for the other tilings; M ragged, N ragged
    in the last column tile"""
    M, N, K = 4112, 384, 1096
    dy, w, pre = gen(M, N, seed=14), gen(N, K, seed=15, scale=0.05), gen(M, K, seed=16)
    cs = torch.full((K,), 0.5, dtype=f32, device="cuda")          # accumulates on top of what is there
    dx = ops.linear_dgrad(dy.cuda(), w.cuda(), pre.cuda(), colsum=cs)
    p = pre.float().requires_grad_(True)
    O.gelu_erf(p).backward(dy.float() @ w.float())
    close("dgelu+colsum.dx", dx, p.grad, 5e-3)
    close("dgelu+colsum.cs", cs - 0.5, dx.float().sum(0), 2e-3)   # sums of the bf16-rounded result


@pytest.mark.parametrize("N", [1152, 384, 1096, 72])
def test_gemm_narrow_last_tile_column_is_bit_identical(ops, N):
    """round 5: on the 256^2 kernel a last tile column that holds at most 128 columns runs the 256 x 128 body (wave tile
    128 x 32, B staged as one half-tile, 32 MFMAs per wave and K tile).  Same products in the same k order: every fused
    epilogue must give the bits of the full-width body (knob gemm_narrow = 0), and the reference values."""
    B, L, K = 3, 1500, 328  # 4500 rows = 17.6 row tiles, K = 5.1 K tiles
    M = B * L
    x, w, b = gen(M, K, seed=41), gen(N, K, seed=42, scale=0.05), gen(N, seed=43, scale=0.3)
    res, mod = gen(M, N, seed=44), gen(B, 3 * N, seed=45, dtype=f32)
    dy, w2, pre = gen(M, K, seed=46), gen(K, N, seed=47, scale=0.05), gen(M, N, seed=48)

    def run():
        out = {}
        out["store"] = ops.linear_fwd(x.cuda(), w.cuda(), b.cuda())
        out["pre"], out["act"] = ops.linear_fwd_gelu(x.cuda(), w.cuda(), b.cuda())
        out["y"], out["xn"] = ops.linear_fwd_gate_res(x.cuda(), w.cuda(), b.cuda(), mod.cuda(), 2 * N, res.cuda(), L)
        out["nn"] = ops.linear_dgrad(dy.cuda(), w2.cuda())
        cs = torch.zeros(N, dtype=f32, device="cuda")
        out["dgelu"] = ops.linear_dgrad(dy.cuda(), w2.cuda(), pre.cuda(), colsum=cs)
        out["cs"] = cs
        torch.cuda.synchronize()
        return out

    ops.gemm_force_tile(256)
    try:
        got = run()
        ops.knob_set("gemm_narrow", 0)
        ref = run()
    finally:
        ops.knob_set("gemm_narrow", 1)
        ops.gemm_force_tile(0)
    for k in got:
        if k == "cs":  # fp32 atomics across workgroups: order-dependent in the last bits
            close("narrow.cs", got[k], ref[k], 1e-5)
        else:
            assert torch.equal(got[k], ref[k]), k
    yr = x.float() @ w.float().t() + b.float()
    close("narrow.store", got["store"], yr, 4e-3)
    close("narrow.act", got["act"], O.gelu_erf(yr), 5e-3)
    close("narrow.xn", got["xn"], res.float() + yr * mod[:, 2 * N:].repeat_interleave(L, dim=0), 4e-3)
    close("narrow.nn", got["nn"], dy.float() @ w2.float(), 4e-3)


def _ints(shape, seed, lo=-2, hi=2, dtype=bf16):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(lo, hi + 1, shape, generator=g).to(dtype)


@pytest.mark.parametrize("dkv16", ["0", "2"], ids=["dkv_plain_qsplit", "dkv16_from_staged_lse"])
@pytest.mark.parametrize("L,Lc", [(2100, 512), (8208, 512), (2304, 300)], ids=["Lq2100", "headline_8208", "ragged_Lk300"])
def test_attention_cross_on_the_ones_column_kernels(ops, L, Lc, dkv16, request):
    """round 5 (vds_attn_args.kv_pad_ones = 2): cross-attention with token-major queries and head-major padded K / V copies
    that carry the ones columns (vds_kv_pad_ones): forward and dQ on the 16x16x32 ones-column kernels, dK/dV on the plain
    kernel with the query-range split or on the 16x16x32 kernel with S started from the staged -lse2 (chosen by the
    number of workgroups; both forced here); nothing may be written outside the rows.  Against fp32 attention and against the
    plain path on the token-major K / V."""
    prev = ops.knob_set("cross_dkv16", int(dkv16))  # 2: the 16x16x32 dK/dV kernel whatever the grid (it needs >= 512 workgroups otherwise)
    request.addfinalizer(lambda: ops.knob_set("cross_dkv16", prev))
    B, H, hd, hdp = 2, 3, 72, 96
    D = H * hd
    qb, kvb = gen(B * L, D, seed=91), gen(B * Lc, 2 * D, seed=92)
    qd, kvd = qb.cuda(), kvb.cuda()
    kp, vp = ops.kv_pad_ones(kvd, B, Lc, H, hd, hdp, 0, D)
    assert torch.equal(kp[..., :hd].reshape(B, H, Lc, hd), kvd[:, :D].reshape(B, Lc, H, hd).permute(0, 2, 1, 3))
    assert torch.equal(vp[..., :hd].reshape(B, H, Lc, hd), kvd[:, D:].reshape(B, Lc, H, hd).permute(0, 2, 1, 3))
    pad_k = torch.zeros(hdp - hd); pad_k[0] = pad_k[1] = 1.0
    pad_v = torch.zeros(hdp - hd); pad_v[0] = pad_v[4] = 1.0
    assert torch.equal(kp[..., hd:].float().cpu(), pad_k.expand(B, H, Lc, hdp - hd))
    assert torch.equal(vp[..., hd:].float().cpu(), pad_v.expand(B, H, Lc, hdp - hd))
    qv = ops.heads_view(qd, B, L, H, hd)
    o = torch.zeros(B * L, D, dtype=bf16, device="cuda")
    lse = torch.zeros(B, H, L, dtype=f32, device="cuda")
    q_before = qd.clone()
    ops.attn_fwd(qv, kp[..., :hd], vp[..., :hd], ops.heads_view(o, B, L, H, hd), lse, kv_pad_ones=2)
    o2 = torch.zeros_like(o)
    lse2 = torch.zeros_like(lse)
    ops.attn_fwd(qv, ops.heads_view(kvd, B, Lc, H, hd, 0), ops.heads_view(kvd, B, Lc, H, hd, D),
                 ops.heads_view(o2, B, L, H, hd), lse2)
    q = qb.reshape(B, L, H, hd).permute(0, 2, 1, 3)
    k = kvb[:, :D].reshape(B, Lc, H, hd).permute(0, 2, 1, 3)
    v = kvb[:, D:].reshape(B, Lc, H, hd).permute(0, 2, 1, 3)
    do = gen(B * L, D, seed=93)
    o_ref, lse_ref, dq_ref, dk_ref, dv_ref = attn_ref(q, k, v, do.reshape(B, L, H, hd).permute(0, 2, 1, 3))
    close("xones.o", o.reshape(B, L, H, hd).permute(0, 2, 1, 3), o_ref, 6e-3)
    close("xones.lse", lse, lse_ref, 1e-3)
    close("xones.o_vs_plain", o, o2, 6e-3)
    dqb, dkvb = torch.zeros_like(qd), torch.zeros_like(kvd)
    ops.attn_bwd(qv, kp[..., :hd], vp[..., :hd], ops.heads_view(o, B, L, H, hd), lse, ops.heads_view(do.cuda(), B, L, H, hd),
                 ops.heads_view(dqb, B, L, H, hd), ops.heads_view(dkvb, B, Lc, H, hd, 0),
                 ops.heads_view(dkvb, B, Lc, H, hd, D), None, kv_pad_ones=2)
    torch.cuda.synchronize()
    assert torch.equal(qd, q_before)  # (mode 1 annotates the q pad; mode 2 must not write into q)
    close("xones.dq", dqb.reshape(B, L, H, hd).permute(0, 2, 1, 3), dq_ref, 8e-3)
    close("xones.dk", dkvb[:, :D].reshape(B, Lc, H, hd).permute(0, 2, 1, 3), dk_ref, 8e-3)
    close("xones.dv", dkvb[:, D:].reshape(B, Lc, H, hd).permute(0, 2, 1, 3), dv_ref, 8e-3)


@pytest.mark.parametrize("hd,hdp,n", [(72, 96, 3), (64, 64, 1), (72, 96, 35)])
def test_dv0_reduce_equals_the_per_block_accumulation(ops, hd, hdp, n):
    """round 5: vds_qkv_rope_bwd with mix = 2 (no dv0 update) + ONE vds_dv0_reduce over the blocks' dv tensors gives the
    accumulator, the un-rotated gradients and the lambda gradients of n calls with mix = 1 -- bit for bit (same terms, same
    order, fp32); 35 tensors take two launches of the reduction (32 pointers per launch); pad columns are never touched."""
    B, H, thw = 2, 2, (2, 4, 5)
    L = thw[0] * thw[1] * thw[2] + 16
    D = H * hd
    cos, sin = O.rope_cos_sin(hd, thw, (3, 7, 11))
    cosd, sind = cos.cuda(), sin.cuda()
    v0 = gen(B, H, L, hdp, seed=171)
    v0[..., hd:] = 0
    v0d = v0.cuda()
    blocks = []
    for i in range(n):
        dq, dk, dv = (gen(B, H, L, hdp, seed=200 + 3 * i + j).cuda() for j in range(3))
        dv[..., hd:] = float("nan")  # (attention leaves the pad of dv unwritten: it must never be read)
        blocks.append((dq, dk, dv, gen(B * L, 3 * D, seed=400 + i).cuda(), torch.tensor([0.1 + 0.8 * i / max(n, 2)]).to(bf16).cuda()))
    acc = torch.zeros(B, H, L, hdp, dtype=f32, device="cuda")
    dlam1 = torch.zeros(n, dtype=f32, device="cuda")
    ref = [ops.qkv_rope_bwd(dq, dk, dv, cosd, sind, qkv, v0d, lam, acc, dlam1[i:i + 1], 1, False, B, L, H, hd, hdp)
           for i, (dq, dk, dv, qkv, lam) in enumerate(blocks)]
    dlam2 = torch.zeros(n, dtype=f32, device="cuda")
    got = [ops.qkv_rope_bwd(dq, dk, dv, cosd, sind, qkv, v0d, lam, None, dlam2[i:i + 1], 2, False, B, L, H, hd, hdp)
           for i, (dq, dk, dv, qkv, lam) in enumerate(blocks)]
    out = torch.full((B, H, L, hdp), 5.0, dtype=f32, device="cuda")
    ops.dv0_reduce([b[2] for b in blocks], [b[4] for b in blocks], out, B, H, L, hd, hdp)
    torch.cuda.synchronize()
    assert torch.equal(out[..., :hd], acc[..., :hd])
    if hdp > hd:
        assert (out[..., hd:] == 5.0).all()  # pad columns untouched
    for a, b in zip(ref, got):
        assert torch.equal(a, b)
    # the lambda gradients are sums of signed terms that largely cancel, accumulated with fp32 atomics in whatever order the
    # workgroups finish: loose here, bit-equal under the fixed-order reduction of deterministic mode
    close("dv0.dlam", dlam2, dlam1, 1e-3)
    ops.set_deterministic(True, 64 << 20)
    try:
        d1, d2 = torch.zeros(n, dtype=f32, device="cuda"), torch.zeros(n, dtype=f32, device="cuda")
        acc_d = torch.zeros(B, H, L, hdp, dtype=f32, device="cuda")
        for i, (dq, dk, dv, qkv, lam) in enumerate(blocks):
            ops.qkv_rope_bwd(dq, dk, dv, cosd, sind, qkv, v0d, lam, acc_d, d1[i:i + 1], 1, False, B, L, H, hd, hdp)
            ops.qkv_rope_bwd(dq, dk, dv, cosd, sind, qkv, v0d, lam, None, d2[i:i + 1], 2, False, B, L, H, hd, hdp)
        torch.cuda.synchronize()
        assert torch.equal(d1, d2)
        close("dv0.dlam.det", d1, dlam1, 1e-3)
    finally:
        ops.set_deterministic(False)
    # accumulate = True adds to what is there
    ops.dv0_reduce([b[2] for b in blocks[:1]], [b[4] for b in blocks[:1]], out, B, H, L, hd, hdp, accumulate=True)
    want = acc[..., :hd] + (1.0 - blocks[0][4].float()) * blocks[0][2][..., :hd].float()
    close("dv0.accumulate", out[..., :hd], want, 1e-6)


# ------------------------------------------------------------------ deterministic mode ----
@pytest.fixture
def deterministic(ops):
    """vds_set_deterministic(1) with a 256-MiB workspace for the duration of one test"""
    ops.set_deterministic(True, 256 << 20)
    yield
    ops.set_deterministic(False)


@pytest.mark.parametrize("tile_", [0, 128, 2], ids=["auto", "t128", "t256x128"])
@pytest.mark.parametrize("M,N,K,split", [(1152, 1152, 8208, 0), (384, 128, 4112, 5), (1152, 4608, 16416, -1), (264, 72, 1000, 3)])
def test_gemm_tn_fixed_order_split_k(ops, deterministic, M, N, K, split, tile_):
    """deterministic mode: a split-K weight gradient writes one fp32 slab per split and a second pass adds the slabs to C in
    split order.  On small integers every sum is exact: C must be the exact product plus what was there; on random
    operands two launches must give identical bits and agree with the default (atomic) mode to fp32 rounding."""
    from video_diffusion_speedrun_amd._lib import EPI_F32, VDS_TN
    dy, x = _ints((K, M), 301), _ints((K, N), 302)
    c0 = _ints((M, N), 303, -5, 5, f32)
    ops.gemm_force_tile(tile_)
    try:
        C = c0.clone().cuda()
        ops.gemm(VDS_TN, EPI_F32, M, N, K, dy.cuda(), M, x.cuda(), N, C, N, split_k=split if split else -1)
        torch.cuda.synchronize()
        assert torch.equal(C.cpu(), c0 + dy.float().t() @ x.float())
        dyr, xr = gen(K, M, seed=304), gen(K, N, seed=305)
        outs = []
        for _ in range(2):
            C = c0.clone().cuda()
            ops.gemm(VDS_TN, EPI_F32, M, N, K, dyr.cuda(), M, xr.cuda(), N, C, N, split_k=split if split else -1)
            outs.append(C)
        torch.cuda.synchronize()
        assert torch.equal(outs[0], outs[1])
        ops.set_deterministic(False)
        Ca = c0.clone().cuda()
        ops.gemm(VDS_TN, EPI_F32, M, N, K, dyr.cuda(), M, xr.cuda(), N, Ca, N, split_k=split if split else -1)
        ops.set_deterministic(True, 256 << 20)
        close("det.tn_vs_atomic", outs[0], Ca, 1e-5)
    finally:
        ops.gemm_force_tile(0)


def test_row_kernels_fixed_order_column_sums(ops, deterministic):
    """deterministic mode: the column sums of rmsnorm_mod_bwd (d shift, d scale, d weight), gate_bwd (d gate, d bias),
    colsum and the lambda gradient of the RoPE backward are per-workgroup partials + one fixed-order pass: two runs give
    identical bits, and the values agree with the default (atomic) mode."""
    B, L, D = 3, 1100, 1152
    dy, x, w = gen(B * L, D, seed=311), gen(B * L, D, seed=312), gen(D, seed=313)
    mod = gen(B, 9 * D, seed=314, dtype=f32)
    dres, y = gen(B * L, D, seed=315), gen(B * L, D, seed=316)

    def run():
        out = {}
        _, rstd = ops.rmsnorm_mod_fwd(x.cuda(), w.cuda(), mod.cuda(), 0, D, B, L)
        dmod = torch.zeros(B, 9 * D, dtype=f32, device="cuda")
        dw = torch.zeros(D, dtype=f32, device="cuda")
        out["dx"] = ops.rmsnorm_mod_bwd(dy.cuda(), x.cuda(), w.cuda(), mod.cuda(), 0, D, rstd, dres.cuda(), dmod, dw, B, L)
        db = torch.zeros(D, dtype=f32, device="cuda")
        out["dyg"] = ops.gate_bwd(dy.cuda(), y.cuda(), mod.cuda(), 2 * D, dmod, db, B, L)
        cs = torch.zeros(D, dtype=f32, device="cuda")
        ops.colsum(dy.cuda(), cs)
        out.update(dmod=dmod, dw=dw, db=db, cs=cs)
        torch.cuda.synchronize()
        return out

    a, b = run(), run()
    for k in a:
        assert torch.equal(a[k], b[k]), k
    ops.set_deterministic(False)
    c = run()
    ops.set_deterministic(True, 256 << 20)
    for k in a:
        if a[k].dtype == f32:
            close("det.rows." + k, a[k], c[k], 2e-5)
        else:
            assert torch.equal(a[k], c[k]), k  # (the row results do not depend on the mode)
    close("det.cs", a["cs"], dy.float().sum(0), 1e-5)
