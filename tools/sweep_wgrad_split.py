"""Sweep the split-K factor of the weight-gradient GEMMs at the DiT-XL shapes against ops._wgrad_split's choice."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_diffusion_speedrun_amd import ops
from video_diffusion_speedrun_amd._lib import EPI_F32, VDS_TN
bf16, f32 = torch.bfloat16, torch.float32
dev = "cuda"
def timeit(fn, iters=8, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
B, L, D = 12, 8208, 1152
M = B * L
for name, N, K in (("qkv", 3*D, D), ("proj", D, D), ("fc1", 4*D, D), ("fc2", D, 4*D)):
    x = torch.randn(M, K, device=dev).to(bf16); dy = torch.randn(M, N, device=dev).to(bf16)
    dW = torch.zeros(N, K, dtype=f32, device=dev)
    tiles = ((N + 127) // 128) * ((K + 127) // 128)
    h = ops._wgrad_split(tiles, (M + 63) // 64, 512)
    res = []
    for s in (1, 2, 3, 4, 5, 6, 8):
        t = timeit(lambda: ops.gemm(VDS_TN, EPI_F32, N, K, M, dy, N, x, K, dW, K, split_k=-s))
        res.append(f"{s}:{t*1e3:.0f}")
    print(name, "tiles", tiles, "heuristic", h, " ".join(res))
