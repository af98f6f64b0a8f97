"""Same-process A/B of the two-phase K-tile variant of the 256^2 GEMM (default since round 4; VDS_GEMM_PHASES=4, read per call, selects
the four-phase loop: 16 instead of 32 MFMAs between barriers) at the DiT-XL plain-store shapes; checks that both produce identical results."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_diffusion_speedrun_amd import ops

bf16 = torch.bfloat16
dev = "cuda"
B, L, D = int(os.environ.get("B", 12)), 8208, 1152
M = B * L
ROUNDS, INNER = 11, 5


def rnd(*shape, scale=1.0):
    return (torch.randn(*shape, device=dev) * scale).to(bf16)


def ab(fns):
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


def with_env(val, fn):
    def run():
        os.environ["VDS_GEMM_PHASES"] = val
        r = fn()
        os.environ["VDS_GEMM_PHASES"] = "2"
        return r
    return run


x1 = rnd(M, D)
w = {n: rnd(*s, scale=0.03) for n, s in dict(qkv=(3 * D, D), proj=(D, D), fc1=(4 * D, D)).items()}
dy1, dy3, dy4 = rnd(M, D), rnd(M, 3 * D), rnd(M, 4 * D)
xs, ws = rnd(8192, 8192), rnd(8192, 8192, scale=0.02)
cases = [("NT qkv fwd    N3456 K1152", lambda: ops.linear_fwd(x1, w["qkv"], None), 2 * M * 3 * D * D),
         ("NT q_cross    N1152 K1152", lambda: ops.linear_fwd(x1, w["proj"], None), 2 * M * D * D),
         ("NN qkv dgrad  N1152 K3456", lambda: ops.linear_dgrad(dy3, w["qkv"]), 2 * M * 3 * D * D),
         ("NN proj dgrad N1152 K1152", lambda: ops.linear_dgrad(dy1, w["proj"]), 2 * M * D * D),
         ("NN fc1 dgrad  N1152 K4608", lambda: ops.linear_dgrad(dy4, w["fc1"]), 2 * M * 4 * D * D),
         ("NT 8192^3", lambda: ops.linear_fwd(xs, ws, None), 2 * 8192 ** 3)]
for name, fn, fl in cases:
    a = with_env("4", fn)().clone()
    b = with_env("2", fn)().clone()
    same = torch.equal(a, b)
    r = ab({"4 phases": with_env("4", fn), "2 phases": with_env("2", fn)})
    print(f"{name}  " + "  ".join(f"{k}: {ms:6.3f} ms {fl / ms / 1e9:6.0f} TF" for k, ms in r.items()) + f"  identical={same}",
          flush=True)
