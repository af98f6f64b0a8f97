"""one launch of each linear-layer GEMM at the DiT-XL shapes (for rocprofv3 --pmc runs)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_diffusion_speedrun_amd import ops
bf16, f32 = torch.bfloat16, torch.float32
dev = "cuda"
B, L, D = int(os.environ.get("B", 12)), 8208, 1152
M = B * L
for name, N, K in (("fc2", D, 4 * D), ("qkv", 3 * D, D)):
    x = (torch.randn(M, K, device=dev)).to(bf16)
    w = (torch.randn(N, K, device=dev) * 0.03).to(bf16)
    dy = torch.randn(M, N, device=dev).to(bf16)
    y = torch.empty(M, N, dtype=bf16, device=dev)
    dW = torch.zeros(N, K, dtype=f32, device=dev)
    for _ in range(3):
        ops.linear_fwd(x, w, None, out=y)
        ops.linear_dgrad(dy, w)
        ops.linear_wgrad(dy, x, dW)
torch.cuda.synchronize()
