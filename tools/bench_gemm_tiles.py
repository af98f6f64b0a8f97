"""Same-box A/B of the GEMM tilings at the DiT-XL shapes (per-GPU batch B): candidates are run round-robin
(ABAB...) so that clock / thermal drift hits all of them alike; the figure is the median over the rounds.
    B=12 python tools/bench_gemm_tiles.py            # NT / NN: 256 vs 256x128 vs 128; TN: 128 vs 256x128 (auto split)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_diffusion_speedrun_amd import ops

bf16, f32 = torch.bfloat16, torch.float32
dev = "cuda"
B, L, D = int(os.environ.get("B", 12)), int(os.environ.get("L", 8208)), int(os.environ.get("D", 1152))
ROUNDS, INNER = int(os.environ.get("ROUNDS", 9)), int(os.environ.get("INNER", 5))


def rnd(*shape, scale=1.0, dtype=bf16):
    return (torch.randn(*shape, device=dev) * scale).to(dtype)


def ab(fns):
    """fns: {name: callable}; returns {name: median ms}"""
    for f in fns.values():
        f()
    torch.cuda.synchronize()
    times = {k: [] for k in fns}
    for _ in range(ROUNDS):
        for k, f in fns.items():
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(INNER):
                f()
            e.record()
            torch.cuda.synchronize()
            times[k].append(s.elapsed_time(e) / INNER)
    return {k: sorted(v)[len(v) // 2] for k, v in times.items()}


def forced(tile, fn):
    def run():
        ops.gemm_force_tile(tile)
        fn()
        ops.gemm_force_tile(0)
    return run


M = B * L
for name, N, K in (("qkv", 3 * D, D), ("proj", D, D), ("fc1", 4 * D, D), ("fc2", D, 4 * D), ("ctxkv", 2 * D, 4096)):
    Mx = B * 512 if name == "ctxkv" else M
    x, w = rnd(Mx, K), rnd(N, K, scale=0.03)
    y = torch.empty(Mx, N, dtype=bf16, device=dev)
    dy = rnd(Mx, N)
    dW = torch.zeros(N, K, dtype=f32, device=dev)
    fl = 2 * Mx * N * K
    rows = (("NT fwd", lambda: ops.linear_fwd(x, w, None, out=y), (256, 2, 128, 0)),
            ("NN dgrad", lambda: ops.linear_dgrad(dy, w), (256, 2, 128, 0)),
            ("TN wgrad", lambda: ops.linear_wgrad(dy, x, dW), (128, 2, 256, 0)))
    for tag, fn, tiles in rows:
        res = ab({t: forced(t, fn) for t in tiles})
        print(f"{tag:9s} {name:6s} M{Mx} N{N} K{K:5d}  " +
              "  ".join(f"t{t}: {ms:7.3f} ms {fl / ms / 1e9:7.1f} TF/s" for t, ms in res.items()), flush=True)
