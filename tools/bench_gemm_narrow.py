"""Same-process A/B of a GEMM knob of the library (csrc/config.h; set through vds_knob_set: 0 = off, 1 = on) at the DiT-XL
shapes with the fused epilogues the model uses, bf16 and fp8.  Candidates run round-robin, median over the rounds.
    B=12 python tools/bench_gemm_narrow.py                     # gemm_narrow: 256 x 128 body for a narrow last tile column
    B=12 KNOB=gemm_tn_joint python tools/bench_gemm_narrow.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_diffusion_speedrun_amd import ops, fp8 as F8

bf16, f32 = torch.bfloat16, torch.float32
dev = "cuda"
B, L, D = int(os.environ.get("B", 12)), int(os.environ.get("L", 8208)), int(os.environ.get("D", 1152))
ROUNDS, INNER = int(os.environ.get("ROUNDS", 9)), int(os.environ.get("INNER", 5))
M = B * L


def rnd(*shape, scale=1.0, dtype=bf16):
    return (torch.randn(*shape, device=dev) * scale).to(dtype)


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


KNOB = os.environ.get("KNOB", "gemm_narrow")


def with_env(val, fn):
    def run():
        ops.knob_set(KNOB, float(val))
        fn()
    return run


mod = rnd(B, 9 * D, dtype=f32)
res = rnd(M, D)
x1, x4 = rnd(M, D), rnd(M, 4 * D)
w = {n: rnd(*s, scale=0.03) for n, s in dict(qkv=(3 * D, D), proj=(D, D), fc1=(4 * D, D), fc2=(D, 4 * D)).items()}
b1, b4 = rnd(D, scale=0.1), rnd(4 * D, scale=0.1)
dy1, dy3, dy4 = rnd(M, D), rnd(M, 3 * D), rnd(M, 4 * D)
hpre = rnd(M, 4 * D)
cs = torch.zeros(4 * D, dtype=f32, device=dev)
cases = [("NT qkv fwd      N3456 K1152", lambda: ops.linear_fwd(x1, w["qkv"], None), 2 * M * 3 * D * D, 1),
         ("NT proj gate+res N1152 K1152", lambda: ops.linear_fwd_gate_res(x1, w["proj"], None, mod, 2 * D, res, L), 2 * M * D * D, 2),
         ("NT q_cross store N1152 K1152", lambda: ops.linear_fwd(x1, w["proj"], None), 2 * M * D * D, 1),
         ("NT fc1 bias+gelu N4608 K1152", lambda: ops.linear_fwd_gelu(x1, w["fc1"], b4), 2 * M * 4 * D * D, 1),
         ("NT fc2 gate+res  N1152 K4608", lambda: ops.linear_fwd_gate_res(x4, w["fc2"], b1, mod, 8 * D, res, L), 2 * M * 4 * D * D, 1),
         ("NN qkv dgrad     N1152 K3456", lambda: ops.linear_dgrad(dy3, w["qkv"]), 2 * M * 3 * D * D, 1),
         ("NN proj dgrad    N1152 K1152", lambda: ops.linear_dgrad(dy1, w["proj"]), 2 * M * D * D, 3),
         ("NN fc1 dgrad     N1152 K4608", lambda: ops.linear_dgrad(dy4, w["fc1"]), 2 * M * 4 * D * D, 1),
         ("NN fc2 dgelu+cs  N4608 K1152", lambda: ops.linear_dgrad(dy1, w["fc2"], pre=hpre, colsum=cs), 2 * M * 4 * D * D, 1)]
total = {"off": 0.0, "on": 0.0}
print(f"{KNOB} off / on; bf16, M = {M}")
for name, fn, fl, cnt in cases:
    r = ab({"off": with_env("0", fn), "on": with_env("1", fn)})
    for t in r:
        total[t] += r[t] * cnt
    print(f"{name}  " + "  ".join(f"{t}: {ms:6.3f} ms {fl / ms / 1e9:6.0f} TF" for t, ms in r.items()), flush=True)
print("per block (NT + NN): " + "  ".join(f"{t}: {v:7.3f} ms" for t, v in total.items()), " x28:",
      "  ".join(f"{28 * v:6.1f}" for v in total.values()))

if os.environ.get("FP8", "1") == "1":
    print("fp8 (e4m3 x e4m3 forward, e5m2 x e4m3 dgrad, e5m2^T x e4m3 wgrad)")
    tot8 = {"off": 0.0, "on": 0.0}
    q = lambda t, fmt: F8.Q(t, fmt, True, False)
    qx1, qx4 = q(x1, F8.E4M3), q(x4, F8.E4M3)
    qw = {n: F8.Q(t, F8.E4M3, True, True, weight=True) for n, t in w.items()}
    qdy1, qdy3, qdy4 = q(dy1, F8.E5M2), q(dy3, F8.E5M2), q(dy4, F8.E5M2)
    y3 = torch.empty(M, 3 * D, dtype=bf16, device=dev)
    y1 = torch.empty(M, D, dtype=bf16, device=dev)
    g11 = torch.zeros(D, D, dtype=f32, device=dev)
    g41 = torch.zeros(4 * D, D, dtype=f32, device=dev)
    g14 = torch.zeros(D, 4 * D, dtype=f32, device=dev)
    c8 = [("fp8 qkv fwd      N3456 K1152", lambda: F8.fwd(qx1, qw["qkv"], y3), 2 * M * 3 * D * D, 1),
          ("fp8 proj gate+res N1152 K1152", lambda: F8.fwd_gate_res(qx1, qw["proj"], None, mod, 2 * D, res, L), 2 * M * D * D, 2),
          ("fp8 q_cross store N1152 K1152", lambda: F8.fwd(qx1, qw["proj"], y1), 2 * M * D * D, 1),
          ("fp8 fc1 bias+gelu N4608 K1152", lambda: F8.fwd_gelu(qx1, qw["fc1"], b4), 2 * M * 4 * D * D, 1),
          ("fp8 fc2 gate+res  N1152 K4608", lambda: F8.fwd_gate_res(qx4, qw["fc2"], b1, mod, 8 * D, res, L), 2 * M * 4 * D * D, 1),
          ("fp8 qkv dgrad     N1152 K3456", lambda: F8.dgrad(qdy3, qw["qkv"]), 2 * M * 3 * D * D, 1),
          ("fp8 proj dgrad    N1152 K1152", lambda: F8.dgrad(qdy1, qw["proj"]), 2 * M * D * D, 3),
          ("fp8 fc1 dgrad     N1152 K4608", lambda: F8.dgrad(qdy4, qw["fc1"]), 2 * M * 4 * D * D, 1),
          ("fp8 fc2 dgelu     N4608 K1152", lambda: F8.dgrad(qdy1, qw["fc2"], pre=hpre), 2 * M * 4 * D * D, 1),
          ("fp8 proj wgrad  1152x1152", lambda: F8.wgrad(qdy1, qx1, g11), 2 * M * D * D, 3),
          ("fp8 fc1 wgrad   4608x1152", lambda: F8.wgrad(qdy4, qx1, g41), 2 * M * 4 * D * D, 1),
          ("fp8 fc2 wgrad   1152x4608", lambda: F8.wgrad(qdy1, qx4, g14), 2 * M * 4 * D * D, 1)]
    for name, fn, fl, cnt in c8:
        try:
            r = ab({"off": with_env("0", fn), "on": with_env("1", fn)})
        except Exception as ex:  # a case the fp8 front end does not take in this form
            print(f"{name}  skipped: {ex}")
            continue
        for t in r:
            tot8[t] += r[t] * cnt
        print(f"{name}  " + "  ".join(f"{t}: {ms:6.3f} ms {fl / ms / 1e9:6.0f} TF" for t, ms in r.items()), flush=True)
    print("per block: " + "  ".join(f"{t}: {v:7.3f} ms" for t, v in tot8.items()), " x28:",
          "  ".join(f"{28 * v:6.1f}" for v in tot8.values()))
