"""cross-attention shape (Lq = 8208, Lk = 512, head_dim 72): the plain 32x32x16 kernels the model uses today (unpadded,
token-major operands) against the ones-column 16x16x32 kernels on padded head-major operands -- how much a pad-in-LDS
variant of the latter could buy."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_diffusion_speedrun_amd import ops
bf16, f32 = torch.bfloat16, torch.float32
dev = "cuda"
B, H, hd, hdp, Lq, Lk = int(os.environ.get("B", 12)), 16, 72, 96, 8208, 512
g = torch.Generator(device=dev).manual_seed(0)
q = torch.zeros(B, H, Lq, hdp, dtype=bf16, device=dev); q[..., :hd] = torch.randn(B, H, Lq, hd, device=dev, generator=g).to(bf16)
k = torch.zeros(B, H, Lk, hdp, dtype=bf16, device=dev); k[..., :hd] = torch.randn(B, H, Lk, hd, device=dev, generator=g).to(bf16)
v = torch.zeros(B, H, Lk, hdp, dtype=bf16, device=dev); v[..., :hd] = torch.randn(B, H, Lk, hd, device=dev, generator=g).to(bf16)
k[..., hd] = 1; k[..., hd + 1] = 1; v[..., hd] = 1; v[..., hd + 4] = 1
o = torch.empty(B * Lq, H * hd, dtype=bf16, device=dev)
lse = torch.empty(B, H, Lq, dtype=f32, device=dev)
ov = ops.heads_view(o, B, Lq, H, hd)
do = torch.randn(B * Lq, H * hd, device=dev, generator=g).to(bf16)
dov = ops.heads_view(do, B, Lq, H, hd)
dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
delta = torch.empty(2, B, H, Lq, dtype=f32, device=dev)
for ones in (False, True, False, True):
    f = lambda: ops.attn_fwd(q[..., :hd], k[..., :hd], v[..., :hd], ov, lse, kv_pad_ones=ones)
    b_ = lambda: ops.attn_bwd(q[..., :hd], k[..., :hd], v[..., :hd], ov, lse, dov, dq[..., :hd], dk[..., :hd], dv[..., :hd], delta, kv_pad_ones=ones)
    for _ in range(3):
        f(); b_()
    ops.prof_enable()
    for _ in range(10):
        f(); b_()
    st = ops.prof_collect()
    ops.prof_enable(0)
    line = [f"ones={ones}"]
    for kname, r in st.items():
        if r["launches"] and kname.startswith("attn"):
            line.append(f"{kname} {r['ms'] / r['launches'] * 1e3:7.1f} us")
    print("  ".join(line), flush=True)
