"""Cross-attention (Lq = 8208 tokens, Lk = 512 context rows, 16 heads of 72) on the plain kernels against the
ones-column 16x16x32 kernels (round 5, kv_pad_ones = 2), same box, round-robin.   B=12 python tools/bench_cross_ones.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_diffusion_speedrun_amd import ops
bf16, f32 = torch.bfloat16, torch.float32
dev = "cuda"
B, L, Lc, H, hd, hdp = int(os.environ.get("B", 12)), 8208, 512, 16, 72, 96
D = H * hd
q = torch.randn(B * L, D, device=dev).to(bf16)
kv = torch.randn(B * Lc, 2 * D, device=dev).to(bf16)
do = (torch.randn(B * L, D, device=dev) * 0.05).to(bf16)
o = torch.empty(B * L, D, dtype=bf16, device=dev)
lse = torch.empty(B, H, L, dtype=f32, device=dev)
dq = torch.empty(B * L, D, dtype=bf16, device=dev)
dkv = torch.empty(B * Lc, 2 * D, dtype=bf16, device=dev)
hv = ops.heads_view
ws = torch.empty(ops.attn_bwd_workspace_floats(B, H, L, Lc, hd, 2), dtype=f32, device=dev)


def fwd_plain():
    ops.attn_fwd(hv(q, B, L, H, hd), hv(kv, B, Lc, H, hd, 0), hv(kv, B, Lc, H, hd, D), hv(o, B, L, H, hd), lse)


def fwd_ones():
    kp, vp = ops.kv_pad_ones(kv, B, Lc, H, hd, hdp, 0, D)
    ops.attn_fwd(hv(q, B, L, H, hd), kp[..., :hd], vp[..., :hd], hv(o, B, L, H, hd), lse, kv_pad_ones=2)


def bwd_plain():
    ops.attn_bwd(hv(q, B, L, H, hd), hv(kv, B, Lc, H, hd, 0), hv(kv, B, Lc, H, hd, D), hv(o, B, L, H, hd), lse,
                 hv(do, B, L, H, hd), hv(dq, B, L, H, hd), hv(dkv, B, Lc, H, hd, 0), hv(dkv, B, Lc, H, hd, D), ws)


def bwd_ones():
    kp, vp = ops.kv_pad_ones(kv, B, Lc, H, hd, hdp, 0, D)
    ops.attn_bwd(hv(q, B, L, H, hd), kp[..., :hd], vp[..., :hd], hv(o, B, L, H, hd), lse, hv(do, B, L, H, hd),
                 hv(dq, B, L, H, hd), hv(dkv, B, Lc, H, hd, 0), hv(dkv, B, Lc, H, hd, D), ws, kv_pad_ones=2)


def t(fn, n=10):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


fwd_plain()
fns = {"fwd plain": fwd_plain, "fwd ones": fwd_ones, "bwd plain": bwd_plain, "bwd ones": bwd_ones}
for f in fns.values():
    f()
res = {k: [] for k in fns}
for _ in range(7):
    for k, f in fns.items():
        res[k].append(t(f))
for k, v in res.items():
    print(f"B={B} {k:10s} {sorted(v)[len(v) // 2]:8.1f} us", flush=True)
ops.prof_enable()
fwd_ones(); bwd_ones(); fwd_plain(); bwd_plain()
for k, v in ops.prof_collect().items():
    print("   class", k, v["launches"], round(v["ms"] * 1e3, 1), "us")
ops.prof_enable(0)
