"""Probe of the persistent (stream-K) GEMM kernel's per-tile overhead: shapes whose tile count is a whole number of rounds
(no stream-K region: the persistent kernel runs data-parallel tiles only) against the plain launch, and the DiT shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_diffusion_speedrun_amd import ops
bf16 = torch.bfloat16
dev = "cuda"


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
    return s.elapsed_time(e) / n


for M, N, K in ((65536, 1024, 1152), (65536, 1024, 4608), (65536, 2048, 1152), (98496, 1152, 1152), (98496, 1024, 1152),
                (16384, 1024, 1152), (16416, 1152, 1152)):
    x = torch.randn(M, K, device=dev).to(bf16)
    w = (torch.randn(N, K, device=dev) * 0.03).to(bf16)
    y = torch.empty(M, N, dtype=bf16, device=dev)
    res = {}
    for rnd in range(3):
        for mode in (0, 1):
            ops.gemm_force_tile(256)
            ops.gemm_stream_k(mode)
            res.setdefault(mode, []).append(t(lambda: ops.linear_fwd(x, w, None, out=y)))
    ops.gemm_stream_k(-1)
    ops.gemm_force_tile(0)
    a, b = sorted(res[0])[1], sorted(res[1])[1]
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    print(f"M{M} N{N} K{K}: tiles {tiles} = {tiles / 256:.2f} rounds  plain {a * 1e3:7.1f} us  persistent {b * 1e3:7.1f} us  ratio {b / a:.3f}",
          flush=True)
