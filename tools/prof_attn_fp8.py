"""one launch of each fp8 attention kernel at the DiT-XL shape (for rocprofv3 --pmc runs)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_diffusion_speedrun_amd import ops
bf16, f32 = torch.bfloat16, torch.float32
E4, E5 = torch.float8_e4m3fn, torch.float8_e5m2
dev = "cuda"
B, H, hd, Lq = int(os.environ.get("B", 6)), 16, 72, 8208
g = torch.Generator(device=dev).manual_seed(0)


def rows(fmt, target, x=None):
    x = torch.randn(B, H, Lq, hd, device=dev, generator=g) if x is None else x
    a = target / x.abs().max().item()
    r = torch.zeros(B, H, Lq, 128, dtype=torch.uint8, device=dev)
    r[..., :hd] = (x * a).to(fmt).view(torch.uint8)
    return r, 1.0 / a


xq, xk = (torch.randn(B, H, Lq, hd, device=dev, generator=g) for _ in range(2))
aq, ak, E = ops.attn_fp8_qk_factors(xq.abs().max().item(), xk.abs().max().item(), hd)
q8, sq = rows(E4, aq * xq.abs().max().item(), xq)
k8, sk = rows(E4, 448.0, xk)
v8, sv = rows(E4, 448.0)
v8[..., hd] = 0x38
q8, k8, v8 = q8.view(E4), k8.view(E4), v8.view(E4)
deq = torch.tensor([sq, sk, sv, 0.0, E, 0.0, 0.0, 0.0], dtype=f32, device=dev)
o = torch.empty(B * Lq, H * hd, dtype=bf16, device=dev)
lse = torch.empty(B, H, Lq, dtype=f32, device=dev)
ov = ops.heads_view(o, B, Lq, H, hd)
do = torch.randn(B * Lq, H * hd, device=dev, generator=g).to(bf16)
doq = torch.zeros(B, H, Lq, 128, dtype=E5, device=dev)
ap = do.float().abs().max().reshape(1)
ac = torch.zeros(1, dtype=f32, device=dev)
dq, dk, dv = (torch.empty(B, H, Lq, 96, dtype=bf16, device=dev) for _ in range(3))
for _ in range(int(os.environ.get("REPS", 2))):
    ops.attn_fp8_fwd(q8, k8, v8, deq, ov, lse, hd)
    stats = ops.attn_fp8_delta(o, do, lse, doq, ap, ac, deq, B, H, Lq, hd)
    ops.attn_fp8_bwd(q8, k8, v8, doq, stats, deq, dq[..., :hd], dk[..., :hd], dv[..., :hd], hd)
torch.cuda.synchronize()
