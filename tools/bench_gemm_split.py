"""Experiment: the N = 1152 linear layers at the DiT-XL step shape (M = 12 x 8208 = 384.75 row tiles of 256, 4.5 column
tiles) as ONE launch against a 256^2-aligned rectangle [98304 x 1024] on the 256^2 kernel (exactly 6 rounds of 256
workgroups) plus the remaining column strip and bottom rows on a smaller tiling."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_diffusion_speedrun_amd import ops

bf16 = torch.bfloat16
B, L, D = int(os.environ.get("B", 12)), 8208, 1152
M = B * L
M0, N0 = (M // 256) * 256, 1024


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


for name, K in (("proj K=1152", D), ("fc2 K=4608", 4 * D), ("dgrad-qkv-like K=3456", 3 * D)):
    x = torch.randn(M, K, device="cuda").to(bf16)
    w = (torch.randn(D, K, device="cuda") * 0.03).to(bf16)
    y = torch.empty(M, D, dtype=bf16, device="cuda")
    fl = 2.0 * M * D * K

    def full(tile=0):
        ops.gemm_force_tile(tile)
        ops.linear_fwd(x, w, out=y)
        ops.gemm_force_tile(0)

    def split(rest_tile):
        ops.gemm_force_tile(256)
        ops.linear_fwd(x[:M0], w[:N0], out=y[:M0, :N0])
        ops.gemm_force_tile(rest_tile)
        ops.linear_fwd(x, w[N0:], out=y[:, N0:])
        ops.gemm_force_tile(128)
        ops.linear_fwd(x[M0:], w[:N0], out=y[M0:, :N0])
        ops.gemm_force_tile(0)

    def big_only():
        ops.gemm_force_tile(256)
        ops.linear_fwd(x[:M0], w[:N0], out=y[:M0, :N0])
        ops.gemm_force_tile(0)

    t = {"auto": timeit(full), "256": timeit(lambda: full(256)), "big-rect only": timeit(big_only),
         "split+128": timeit(lambda: split(128)), "split+mid": timeit(lambda: split(2))}
    print(name, " ".join(f"{k}: {v:7.1f} us ({fl / v / 1e6:6.1f} TF/s)" if "only" not in k else f"{k}: {v:7.1f} us"
                         for k, v in t.items()))
