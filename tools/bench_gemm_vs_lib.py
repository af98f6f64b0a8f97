"""Yardstick only (never on the product path): our bf16 GEMMs vs the vendor library (hipBLASLt through
torch.matmul) at the DiT-XL shapes, same box, HIP events.  Shows how much headroom the 256x256 kernel has."""
import os, sys
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


B, L, D = int(os.environ.get("B", 6)), 8208, 1152
M = B * L
TILES = [int(t) for t in os.environ.get("TILES", "0").split(",")]  # 0 auto, 128, 256, 2 (= 256x128, 2 WG / CU)
NOLIB = os.environ.get("NOLIB") == "1"
print(f"{'shape':40s} " + " ".join(f"{'t' + str(t) + ' ms':>9s} {'TF/s':>7s}" for t in TILES) + f" {'lib ms':>9s} {'TF/s':>8s}")
for name, N, K in (("qkv", 3 * D, D), ("proj", D, D), ("fc1", 4 * D, D), ("fc2", D, 4 * D), ("ctxkv", 2 * D, 4096)):
    Mx = B * 512 if name == "ctxkv" else M
    x, w = rnd(Mx, K), rnd(N, K, scale=0.03)
    y = torch.empty(Mx, N, dtype=bf16, device=dev)
    dy = rnd(Mx, N)
    dW = torch.zeros(N, K, dtype=f32, device=dev)
    fl = 2 * Mx * N * K
    for tag, ours, lib in (
        ("NT fwd", lambda: ops.linear_fwd(x, w, None, out=y), lambda: torch.matmul(x, w.t())),
        ("NN dgrad", lambda: ops.linear_dgrad(dy, w), lambda: torch.matmul(dy, w)),
        ("TN wgrad", lambda: ops.linear_wgrad(dy, x, dW), lambda: torch.matmul(dy.t(), x)),
    ):
        cols = []
        for t in TILES:
            ops.gemm_force_tile(t)
            to = timeit(ours)
            cols.append(f"{to*1e3:9.3f} {fl/to/1e12:7.1f}")
        ops.gemm_force_tile(0)
        tl = float("nan") if NOLIB else timeit(lib)
        print(f"{tag:9s} {name:6s} M{Mx} N{N} K{K:5d}     " + " ".join(cols) + f" {tl*1e3:9.3f} {fl/tl/1e12:8.1f}", flush=True)
