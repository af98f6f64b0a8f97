"""Run one kernel class back to back for a few seconds (for tools/power_probe.sh): KIND=attn|gemm|rmsnorm, ZERO=1 for
all-zero operands, SECS=duration.  Prints launches per second."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_diffusion_speedrun_amd import ops

bf16, f32 = torch.bfloat16, torch.float32
kind, zero, secs = os.environ.get("KIND", "attn"), os.environ.get("ZERO") == "1", float(os.environ.get("SECS", 14))
dev = "cuda"
mk = (lambda *s: torch.zeros(*s, device=dev).to(bf16)) if zero else (lambda *s: torch.randn(*s, device=dev).to(bf16))
if kind == "attn":
    B, H, L, hd, hdp = 6, 16, 8208, 72, 96
    q, k, v = (torch.zeros(B, H, L, hdp, dtype=bf16, device=dev) for _ in range(3))
    for t in (q, k, v):
        t[..., :hd] = mk(B, H, L, hd)
    k[..., hd] = 1; k[..., hd + 1] = 1; v[..., hd] = 1; v[..., hd + 4] = 1
    o = torch.empty(B * L, H * hd, dtype=bf16, device=dev)
    lse = torch.empty(B, H, L, dtype=f32, device=dev)
    ov = ops.heads_view(o, B, L, H, hd)
    do = mk(B * L, H * hd)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    delta = torch.empty(2, B, H, L, dtype=f32, device=dev)
    dov = ops.heads_view(do, B, L, H, hd)
    ops.attn_fwd(q[..., :hd], k[..., :hd], v[..., :hd], ov, lse, kv_pad_ones=True)
    fn = lambda: ops.attn_bwd(q[..., :hd], k[..., :hd], v[..., :hd], ov, lse, dov, dq[..., :hd], dk[..., :hd],
                              dv[..., :hd], delta, kv_pad_ones=True)
elif kind == "gemm":
    M, N, K = 98496, 4608, 1152
    x, w = mk(M, K), (mk(N, K).float() * 0.03).to(bf16)
    y = torch.empty(M, N, dtype=bf16, device=dev)
    fn = lambda: ops.linear_fwd(x, w, out=y)
else:
    B, L, D = 12, 8208, 1152
    x, mod = mk(B * L, D), torch.randn(B, 9 * D, device=dev) * 0.3
    fn = lambda: ops.rmsnorm_mod_fwd(x, None, mod, 0, D, B, L)
fn(); torch.cuda.synchronize()
t0, n = time.time(), 0
while time.time() - t0 < secs:
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    n += 20
print(f"{kind} zero={int(zero)}: {n / (time.time() - t0):.1f} launches/s, {(time.time() - t0) / n * 1e3:.3f} ms each")
