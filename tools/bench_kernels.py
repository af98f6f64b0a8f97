"""Micro-benchmarks of the hot kernels at the DiT-XL / seq-8k shapes (HIP events, random data)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_diffusion_speedrun_amd import ops

bf16, f32 = torch.bfloat16, torch.float32
dev = "cuda"


def timeit(fn, iters=10, warm=3):
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


def rnd(*shape, scale=1.0, dtype=bf16):
    return (torch.randn(*shape, device=dev) * scale).to(dtype)


res = []
B, L, D = int(os.environ.get("B", 2)), 8208, 1152
M = B * L
for name, N, K in (("qkv", 3 * D, D), ("proj", D, D), ("fc1", 4 * D, D), ("fc2", D, 4 * D)):
    x, w = rnd(M, K), rnd(N, K, scale=0.03)
    y = torch.empty(M, N, dtype=bf16, device=dev)
    t = timeit(lambda: ops.linear_fwd(x, w, None, out=y))
    res.append((f"NT {name} M{M} N{N} K{K}", t, 2 * M * N * K / t / 1e12))
    dy = rnd(M, N)
    t = timeit(lambda: ops.linear_dgrad(dy, w))
    res.append((f"NN {name} dgrad", t, 2 * M * N * K / t / 1e12))
    dW = torch.zeros(N, K, dtype=f32, device=dev)
    t = timeit(lambda: ops.linear_wgrad(dy, x, dW))
    res.append((f"TN {name} wgrad", t, 2 * M * N * K / t / 1e12))
xs, ws = rnd(8192, 8192), rnd(8192, 8192, scale=0.01)
ys = torch.empty(8192, 8192, dtype=bf16, device=dev)
t = timeit(lambda: ops.linear_fwd(xs, ws, None, out=ys))
res.append(("NT 8192^3", t, 2 * 8192 ** 3 / t / 1e12))

for hd, H, hdp in ((72, 16, 96), (64, 12, 64), (128, 16, 128)):
    Lq = 8208
    q, k, v = (torch.zeros(B, H, Lq, hdp, dtype=bf16, device=dev) for _ in range(3))
    for t_ in (q, k, v):
        t_[..., :hd] = rnd(B, H, Lq, hd)
    o = torch.empty(B * Lq, H * hd, dtype=bf16, device=dev)
    lse = torch.empty(B, H, Lq, dtype=f32, device=dev)
    ov = ops.heads_view(o, B, Lq, H, hd)
    fl = 4 * B * H * Lq * Lq * hd
    t = timeit(lambda: ops.attn_fwd(q[..., :hd], k[..., :hd], v[..., :hd], ov, lse), iters=5, warm=2)
    res.append((f"attn fwd hd{hd} H{H} L{Lq}", t, fl / t / 1e12))
    do = rnd(B * Lq, H * hd)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    delta = torch.empty(2, B, H, Lq, dtype=f32, device=dev)
    dov = ops.heads_view(do, B, Lq, H, hd)
    t = timeit(lambda: ops.attn_bwd(q[..., :hd], k[..., :hd], v[..., :hd], ov, lse, dov, dq[..., :hd], dk[..., :hd],
                                    dv[..., :hd], delta), iters=5, warm=2)
    res.append((f"attn bwd hd{hd} (algorithmic 2.5x fwd flops)", t, 2.5 * fl / t / 1e12))
    # cross attention Lk=512
    kc, vc = q[:, :, :512], k[:, :, :512]
    t = timeit(lambda: ops.attn_fwd(q[..., :hd], kc[..., :hd], vc[..., :hd], ov, lse), iters=5, warm=2)
    res.append((f"cross fwd hd{hd} Lk512", t, 4 * B * H * Lq * 512 * hd / t / 1e12))

x = rnd(M, D)
mod = rnd(B, 9 * D, dtype=f32)
t = timeit(lambda: ops.rmsnorm_mod_fwd(x, None, mod, 0, D, B, L))
res.append(("rmsnorm_mod fwd (GB/s in TF col)", t, 2 * M * D * 2 / t / 1e9))
for r in res:
    print(f"{r[0]:55s} {r[1]*1e3:9.3f} ms  {r[2]:9.1f}")
