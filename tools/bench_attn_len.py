"""bf16 attention kernels at L = 8192 (64 key / query tiles per head = one XCD round) vs L = 8208 (65 tiles: heads drift
across XCD rounds): time per launch and per (query, key) pair -- how much the 65th tile costs beyond its 1/65 of work."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_diffusion_speedrun_amd import ops
bf16, f32 = torch.bfloat16, torch.float32
dev = "cuda"
B, H, hd, hdp = int(os.environ.get("B", 6)), 16, 72, 96
for L in (8192, 8208, 8192, 8208):
    g = torch.Generator(device=dev).manual_seed(0)
    q, k, v = (torch.zeros(B, H, L, hdp, dtype=bf16, device=dev) for _ in range(3))
    for t_ in (q, k, v):
        t_[..., :hd] = torch.randn(B, H, L, hd, device=dev, generator=g).to(bf16)
    k[..., hd] = 1; k[..., hd + 1] = 1; v[..., hd] = 1; v[..., hd + 4] = 1
    o = torch.empty(B * L, H * hd, dtype=bf16, device=dev)
    lse = torch.empty(B, H, L, dtype=f32, device=dev)
    ov = ops.heads_view(o, B, L, H, hd)
    do = torch.randn(B * L, H * hd, device=dev, generator=g).to(bf16)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    delta = torch.empty(2, B, H, L, dtype=f32, device=dev)
    dov = ops.heads_view(do, B, L, H, hd)
    for _ in range(3):
        ops.attn_fwd(q[..., :hd], k[..., :hd], v[..., :hd], ov, lse, kv_pad_ones=True)
        ops.attn_bwd(q[..., :hd], k[..., :hd], v[..., :hd], ov, lse, dov, dq[..., :hd], dk[..., :hd], dv[..., :hd], delta, kv_pad_ones=True)
    ops.prof_enable()
    for _ in range(6):
        ops.attn_fwd(q[..., :hd], k[..., :hd], v[..., :hd], ov, lse, kv_pad_ones=True)
        ops.attn_bwd(q[..., :hd], k[..., :hd], v[..., :hd], ov, lse, dov, dq[..., :hd], dk[..., :hd], dv[..., :hd], delta, kv_pad_ones=True)
    st = ops.prof_collect()
    ops.prof_enable(0)
    for kname in ("attn_fwd", "attn_bwd_dkv", "attn_bwd_dq"):
        r = st[kname]
        ms = r["ms"] / r["launches"]
        print(f"L={L} {kname}: {ms:8.3f} ms   {ms * 1e6 / (B * H * L * L) * 1e3:8.4f} ps per (q,k) pair")
