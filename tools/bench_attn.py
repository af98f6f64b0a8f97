"""Attention-only micro-benchmark at the DiT-XL / seq-8k shape (HIP events, random data).
VDS_ATTN_VARIANT selects the forward variant (read once per process)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_diffusion_speedrun_amd import ops

bf16, f32 = torch.bfloat16, torch.float32
dev = "cuda"


def timeit(fn, iters=8, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


B = int(os.environ.get("B", 2))
tag = f"variant={os.environ.get('VDS_ATTN_VARIANT', 'default')}"
SHAPES = ((72, 16, 96), (64, 12, 64), (128, 16, 128))
if os.environ.get("ONLY72") == "1":
    SHAPES = SHAPES[:1]
for hd, H, hdp in SHAPES:
    Lq = 8208
    g = torch.Generator(device=dev).manual_seed(0)
    q, k, v = (torch.zeros(B, H, Lq, hdp, dtype=bf16, device=dev) for _ in range(3))
    ZERO = os.environ.get("ZERO") == "1"  # all-zero operands: same cycles, less switching power -> shows the DVFS share
    for t_ in (q, k, v):
        if not ZERO:
            t_[..., :hd] = torch.randn(B, H, Lq, hd, device=dev, generator=g).to(bf16)
    ones = os.environ.get("ONES", "1") == "1" and hdp - hd >= 8
    if ones:  # the pad layout vds_qkv_rope_fwd produces
        k[..., hd] = 1
        k[..., hd + 1] = 1
        v[..., hd] = 1
        v[..., hd + 4] = 1
    o = torch.empty(B * Lq, H * hd, dtype=bf16, device=dev)
    lse = torch.empty(B, H, Lq, dtype=f32, device=dev)
    ov = ops.heads_view(o, B, Lq, H, hd)
    fl = 4 * B * H * Lq * Lq * hd
    t = timeit(lambda: ops.attn_fwd(q[..., :hd], k[..., :hd], v[..., :hd], ov, lse, kv_pad_ones=ones))
    print(f"{tag} fwd hd{hd}: {t*1e3:8.3f} ms {fl/t/1e12:7.1f} TF/s")
    do = torch.randn(B * Lq, H * hd, device=dev, generator=g).to(bf16)
    if ZERO:
        do.zero_()
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    delta = torch.empty(2, B, H, Lq, dtype=f32, device=dev)
    dov = ops.heads_view(do, B, Lq, H, hd)
    ops.prof_enable()
    for _ in range(4):
        ops.attn_bwd(q[..., :hd], k[..., :hd], v[..., :hd], ov, lse, dov, dq[..., :hd], dk[..., :hd], dv[..., :hd], delta,
                     kv_pad_ones=ones)
    st = ops.prof_collect()
    ops.prof_enable(0)
    for kname in ("attn_bwd_dkv", "attn_bwd_dq", "attn_bwd_delta"):
        r = st.get(kname) or st[kname + "_plain"]
        print(f"{tag} {kname} hd{hd}: {r['ms']/r['launches']:8.3f} ms {r['flops']/r['ms']/1e9:7.1f} TF/s")

    if hd == 72 and ops.attn_fp8_supported(hd):  # the fp8 kernels (attention_fp8.hip) on the same problem
        E4, E5 = torch.float8_e4m3fn, torch.float8_e5m2

        def rows(x, fmt, target):
            a = target / x.abs().max().item()
            r = torch.zeros(B, H, Lq, 128, dtype=torch.uint8, device=dev)
            r[..., :hd] = (x[..., :hd].float() * a).to(fmt).view(torch.uint8)
            return r, 1.0 / a
        aq, ak, E = ops.attn_fp8_qk_factors(q[..., :hd].float().abs().max().item(), k[..., :hd].float().abs().max().item(), hd)
        q8, sq = rows(q, E4, aq * q[..., :hd].float().abs().max().item())
        k8, sk = rows(k, E4, 448.0)
        v8, sv = rows(v, E4, 448.0)
        v8[..., hd] = 0x38
        q8, k8, v8 = q8.view(E4), k8.view(E4), v8.view(E4)
        deq = torch.tensor([sq, sk, sv, 0.0, E, 0.0, 0.0, 0.0], dtype=f32, device=dev)
        t = timeit(lambda: ops.attn_fp8_fwd(q8, k8, v8, deq, ov, lse, hd))
        print(f"{tag} fp8 fwd hd{hd}: {t*1e3:8.3f} ms {fl/t/1e12:7.1f} TF/s")
        doq = torch.zeros(B, H, Lq, 128, dtype=E5, device=dev)
        ap = do.float().abs().max().reshape(1)
        ac = torch.zeros(1, dtype=f32, device=dev)
        stats = ops.attn_fp8_delta(o, do, lse, doq, ap, ac, deq, B, H, Lq, hd)
        t = timeit(lambda: ops.attn_fp8_delta(o, do, lse, doq, ap, ac, deq, B, H, Lq, hd))
        print(f"{tag} fp8 delta hd{hd}: {t*1e3:8.3f} ms")
        ops.prof_enable()
        for _ in range(4):
            ops.attn_fp8_bwd(q8, k8, v8, doq, stats, deq, dq[..., :hd], dk[..., :hd], dv[..., :hd], hd)
        st = ops.prof_collect()
        ops.prof_enable(0)
        for kname in ("attn_fp8_dkv", "attn_fp8_dq"):
            r = st[kname]
            print(f"{tag} {kname} hd{hd}: {r['ms']/r['launches']:8.3f} ms {r['flops']/r['ms']/1e9:7.1f} TF/s")
