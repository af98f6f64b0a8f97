"""Is the TN (both operands k-major) main loop of the 256^2 kernel slow, or its split-K / fp32 epilogue at the weight-gradient
shapes?  A square problem without split-K: NT / NN / TN on the 256^2 and the 256x128 tilings."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_diffusion_speedrun_amd import ops
bf16, f32 = torch.bfloat16, torch.float32
dev = "cuda"


def t(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


for S in (4096, 8192):
    x = torch.randn(S, S, device=dev).to(bf16)
    w = (torch.randn(S, S, device=dev) * 0.03).to(bf16)
    y = torch.empty(S, S, dtype=bf16, device=dev)
    dW = torch.zeros(S, S, dtype=f32, device=dev)
    fl = 2.0 * S * S * S
    for tile in (256, 2):
        ops.gemm_force_tile(tile)
        r = {"NT": t(lambda: ops.linear_fwd(x, w, None, out=y)), "NN": t(lambda: ops.linear_dgrad(x, w)),
             "TN f32 out": t(lambda: ops.linear_wgrad(x, w, dW, split_k=1))}
        ops.gemm_force_tile(0)
        print(f"S={S} tile {tile}: " + "  ".join(f"{k} {ms:.3f} ms {fl / ms / 1e9:.0f} TF" for k, ms in r.items()), flush=True)
# the weight-gradient shapes with the split the library picks and forced splits, 256^2 tiling
M = 12 * 8208
for N, K in ((1152, 1152), (4608, 1152), (1152, 4608), (3456, 1152)):
    dy = torch.randn(M, N, device=dev).to(bf16)
    x = torch.randn(M, K, device=dev).to(bf16)
    dW = torch.zeros(N, K, dtype=f32, device=dev)
    fl = 2.0 * M * N * K
    out = []
    for tile in (2, 256):
        for sp in (None, 4, 8, 16, 32):
            ops.gemm_force_tile(tile)
            try:
                ms = t(lambda: ops.linear_wgrad(dy, x, dW, split_k=sp))
            finally:
                ops.gemm_force_tile(0)
            out.append(f"t{tile}/s{sp}: {ms:.3f} ms {fl / ms / 1e9:.0f} TF")
    print(f"wgrad {N}x{K}: " + "  ".join(out), flush=True)
