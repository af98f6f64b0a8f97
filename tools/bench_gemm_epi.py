"""The fused-epilogue GEMMs of the MLP at the DiT-XL step shape, bf16 and fp8 (HIP events, random data)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_diffusion_speedrun_amd import ops, fp8 as F8

bf16, f32 = torch.bfloat16, torch.float32
dev = "cuda"
B, L, D = int(os.environ.get("B", 12)), 8208, 1152
M = B * L


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
    return s.elapsed_time(e) / iters


def rnd(*shape, scale=1.0, dtype=bf16):
    return (torch.randn(*shape, device=dev) * scale).to(dtype)


x, xh = rnd(M, D), rnd(M, 4 * D)
w_fc1, w_fc2, w_proj = rnd(4 * D, D, scale=0.03), rnd(D, 4 * D, scale=0.03), rnd(D, D, scale=0.03)
b_fc1 = rnd(4 * D, scale=0.1)
mod = torch.randn(B, 9 * D, device=dev, dtype=f32)
pre, dy = rnd(M, 4 * D), rnd(M, D)
cs = torch.zeros(4 * D, dtype=f32, device=dev)
fl = 2 * M * 4 * D * D
rows = [
    ("bf16 fc1 fwd  (store, no bias)", fl, lambda: ops.linear_fwd(x, w_fc1)),
    ("bf16 fc1 fwd  (bias + gelu)", fl, lambda: ops.linear_fwd_gelu(x, w_fc1, b_fc1)),
    ("bf16 fc2 fwd  (gate + res)", fl, lambda: ops.linear_fwd_gate_res(xh, w_fc2, None, mod, 0, x, L)),
    ("bf16 proj fwd (gate + res)", fl / 4, lambda: ops.linear_fwd_gate_res(x, w_proj, None, mod, 0, x, L)),
    ("bf16 fc2 dgrad (store)", fl, lambda: ops.linear_dgrad(dy, w_fc2)),
    ("bf16 fc2 dgrad (dgelu + colsum)", fl, lambda: ops.linear_dgrad(dy, w_fc2, pre, cs)),
]
xq, w1q, w2q = F8.Q(x, 0, True, True), F8.Q(w_fc1, 0, True, True), F8.Q(w_fc2, 0, True, True)
dyq = F8.Q(dy, 1, True, True)
amax_a, amax_d, rec = torch.full((1,), 4.0, device=dev), torch.full((1,), 2.0, device=dev), torch.zeros(1, device=dev)
out = torch.empty(M, 4 * D, dtype=bf16, device=dev)
rows += [
    ("fp8  fc1 fwd  (store)", fl, lambda: F8.fwd(xq, w1q, out)),
    ("fp8  fc1 fwd  (bias + gelu -> e4m3 q + qt)", fl, lambda: F8.fwd_gelu_emit(xq, w1q, b_fc1, amax_a, rec, True)),
    ("fp8  fc2 dgrad (store)", fl, lambda: F8.dgrad(dyq, w2q)),
    ("fp8  fc2 dgrad (dgelu -> e5m2 q + qt + colsum)", fl, lambda: F8.dgrad_gelu_emit(dyq, w2q, pre, amax_d, rec, cs)),
]
for name, f, fn in rows:
    ms = timeit(fn)
    print(f"{name:48s} {ms:7.3f} ms  {f / ms / 1e9:7.1f} TF/s", flush=True)
