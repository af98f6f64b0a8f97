"""one launch of each attention kernel at the DiT-XL shape (for rocprofv3 --pmc runs)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_diffusion_speedrun_amd import ops
bf16, f32 = torch.bfloat16, torch.float32
dev = "cuda"
B, H, hd, hdp, Lq = int(os.environ.get("B", 6)), 16, 72, 96, 8208
g = torch.Generator(device=dev).manual_seed(0)
q, k, v = (torch.zeros(B, H, Lq, hdp, dtype=bf16, device=dev) for _ in range(3))
for t_ in (q, k, v):
    t_[..., :hd] = torch.randn(B, H, Lq, hd, device=dev, generator=g).to(bf16)
ones = os.environ.get("ONES", "1") == "1"
if ones:  # the pad layout vds_qkv_rope_fwd produces (vds_attn_args.kv_pad_ones)
    k[..., hd] = 1.0
    k[..., hd + 1] = 1.0
    v[..., hd] = 1.0
    v[..., hd + 4] = 1.0
o = torch.empty(B * Lq, H * hd, dtype=bf16, device=dev)
lse = torch.empty(B, H, Lq, dtype=f32, device=dev)
ov = ops.heads_view(o, B, Lq, H, hd)
do = torch.randn(B * Lq, H * hd, device=dev, generator=g).to(bf16)
dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
delta = torch.empty(2, B, H, Lq, dtype=f32, device=dev)
dov = ops.heads_view(do, B, Lq, H, hd)
for _ in range(int(os.environ.get("REPS", 2))):
    ops.attn_fwd(q[..., :hd], k[..., :hd], v[..., :hd], ov, lse, kv_pad_ones=ones)
    ops.attn_bwd(q[..., :hd], k[..., :hd], v[..., :hd], ov, lse, dov, dq[..., :hd], dk[..., :hd], dv[..., :hd], delta, kv_pad_ones=ones)
torch.cuda.synchronize()
