"""plain launch against the persistent kernel with the next tile's first K tile under the epilogue (VDS_GEMM_PK=1)"""
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


for M, N, K in ((65536, 1024, 1152), (65536, 1024, 4608), (98496, 1152, 1152), (98496, 3456, 1152), (98496, 1152, 4608),
                (98496, 4608, 1152), (32832, 1152, 1152)):
    x = torch.randn(M, K, device=dev).to(bf16)
    w = (torch.randn(N, K, device=dev) * 0.03).to(bf16)
    y = torch.empty(M, N, dtype=bf16, device=dev)
    res = {}
    for rnd in range(3):
        for mode in ("0", "1"):
            os.environ["VDS_GEMM_PK"] = mode
            res.setdefault(mode, []).append(t(lambda: ops.linear_fwd(x, w, None, out=y)))
    a, b = sorted(res["0"])[1], sorted(res["1"])[1]
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    print(f"M{M} N{N} K{K}: tiles {tiles} = {tiles / 256:.2f} rounds  plain {a * 1e3:7.1f} us  pk {b * 1e3:7.1f} us  ratio {b / a:.3f}",
          flush=True)
