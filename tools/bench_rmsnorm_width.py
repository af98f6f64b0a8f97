"""rmsnorm_mod_fwd / bwd and gate_bwd at D = 1152 (2.25 wave-instructions per row and stream) against D = 1024 and
1536 (2 and 3 full instructions): is the partly filled third instruction what holds these kernels at 4.7 TB/s?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_diffusion_speedrun_amd import ops
bf16, f32 = torch.bfloat16, torch.float32
dev = "cuda"
B, L = int(os.environ.get("B", 12)), 8208


def t(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e-3


for D in (1024, 1152, 1280, 1536):
    M = B * L
    x = torch.randn(M, D, device=dev).to(bf16)
    dy = torch.randn(M, D, device=dev).to(bf16)
    dres = torch.randn(M, D, device=dev).to(bf16)
    mod = torch.randn(B, 9 * D, device=dev, dtype=f32)
    dmod = torch.zeros(B, 9 * D, device=dev, dtype=f32)
    xn, rstd = ops.rmsnorm_mod_fwd(x, None, mod, 0, D, B, L)
    tf = t(lambda: ops.rmsnorm_mod_fwd(x, None, mod, 0, D, B, L))
    tb = t(lambda: ops.rmsnorm_mod_bwd(dy, x, None, mod, 0, D, rstd, dres, dmod, None, B, L))
    tg = t(lambda: ops.gate_bwd(dy, x, mod, 2 * D, dmod, None, B, L))
    print(f"D={D}: rmsnorm_mod_fwd {tf * 1e6:7.1f} us {4 * M * D / tf / 1e12:5.2f} TB/s | rmsnorm_mod_bwd {tb * 1e6:7.1f} us "
          f"{8 * M * D / tb / 1e12:5.2f} TB/s | gate_bwd {tg * 1e6:7.1f} us {6 * M * D / tg / 1e12:5.2f} TB/s", flush=True)
