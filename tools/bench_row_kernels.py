"""HBM-bound row kernels of a DiT-XL block at the bench shape (B = 12, L = 8208, D = 1152): microseconds per launch and
achieved TB/s (algorithmic bytes), for same-box A/B of library builds (VDS_LIB_PATH).
    B=12 python tools/bench_row_kernels.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_diffusion_speedrun_amd import ops

bf16, f32 = torch.bfloat16, torch.float32
dev = "cuda"
B, L, D = int(os.environ.get("B", 12)), int(os.environ.get("L", 8208)), int(os.environ.get("D", 1152))
M = B * L


def rnd(*shape, dtype=bf16):
    return torch.randn(*shape, device=dev).to(dtype)


def timeit(fn, iters=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            fn()
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / iters)
    return sorted(ts)[len(ts) // 2] * 1e-3


x, dy, dres, y = rnd(M, D), rnd(M, D), rnd(M, D), rnd(M, D)
mod = rnd(B, 9 * D, dtype=f32)
_, rstd = ops.rmsnorm_mod_fwd(x, None, mod, 0, D, B, L)
dmod = torch.zeros(B, 9 * D, dtype=f32, device=dev)
db = torch.zeros(D, dtype=f32, device=dev)
cases = [
    ("rmsnorm_mod_fwd", lambda: ops.rmsnorm_mod_fwd(x, None, mod, 0, D, B, L), 4.0 * M * D),
    ("rmsnorm_mod_bwd (+res)", lambda: ops.rmsnorm_mod_bwd(dy, x, None, mod, 0, D, rstd, dres, dmod, None, B, L), 8.0 * M * D),
    ("rmsnorm_mod_bwd (no res)", lambda: ops.rmsnorm_mod_bwd(dy, x, None, mod, 0, D, rstd, None, dmod, None, B, L), 6.0 * M * D),
    ("gate_bwd (+bias)", lambda: ops.gate_bwd(dy, y, mod, 2 * D, dmod, db, B, L), 6.0 * M * D),
    ("gate_bwd", lambda: ops.gate_bwd(dy, y, mod, 2 * D, dmod, None, B, L), 6.0 * M * D),
]
tag = os.environ.get("VDS_LIB_PATH", "product")
for name, fn, nbytes in cases:
    t = timeit(fn)
    print(f"lib={tag} B={B} {name:26s} {t * 1e6:8.1f} us  {nbytes / t / 1e12:5.2f} TB/s ({nbytes / t / 8e12:4.2f} of 8 TB/s)", flush=True)
