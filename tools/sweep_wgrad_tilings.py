"""Weight-gradient GEMMs at the DiT-XL shapes, per-GPU batch B: the library's own choice (tiling + split) against every
tiling (128^2 / 256x128 / 256^2) x split count.    B=2 python tools/sweep_wgrad_tilings.py"""
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
    return s.elapsed_time(e) / iters * 1e3


B, L, D = int(os.environ.get("B", 2)), 8208, 1152
M = B * L
tot_auto = tot_best = 0.0
for name, N, K, cnt in (("qkv", 3 * D, D, 1), ("proj", D, D, 3), ("fc1", 4 * D, D, 1), ("fc2", D, 4 * D, 1)):
    x = torch.randn(M, K, device=dev).to(bf16)
    dy = torch.randn(M, N, device=dev).to(bf16)
    dW = torch.zeros(N, K, dtype=f32, device=dev)
    auto = timeit(lambda: ops.linear_wgrad(dy, x, dW))
    res = {}
    for tile in (128, 2, 256):
        for sp in (1, 2, 3, 4, 6, 8, 12, 16):
            ops.gemm_force_tile(tile)
            try:
                res[(tile, sp)] = timeit(lambda: ops.linear_wgrad(dy, x, dW, split_k=sp), iters=6, warm=2)
            finally:
                ops.gemm_force_tile(0)
    best = min(res, key=res.get)
    tot_auto += cnt * auto
    tot_best += cnt * res[best]
    top = sorted(res.items(), key=lambda kv: kv[1])[:4]
    print(f"{name:5s} {N}x{K}: auto {auto:6.1f} us  best t{best[0]}/s{best[1]} {res[best]:6.1f} us  | " +
          "  ".join(f"t{t}/s{s} {v:.1f}" for (t, s), v in top), flush=True)
print(f"per block: auto {tot_auto:.1f} us, best {tot_best:.1f} us  (x28: {tot_auto * 28e-3:.2f} / {tot_best * 28e-3:.2f} ms)")
