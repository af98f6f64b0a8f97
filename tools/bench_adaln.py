"""The batched adaLN modulation linears at the DiT-XL shape (28 blocks x [9D, D] weights against a B-row input):
forward, dW + dbias, dx.  Bytes = the 669 MB of bf16 weights (forward, dx) resp. the fp32 dW written (dW).
    B=12 python tools/bench_adaln.py         (VDS_LIB_PATH=<another build> for an A/B)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_diffusion_speedrun_amd import ops

dev, f32, bf16 = "cuda", torch.float32, torch.bfloat16
B, D, nb = int(os.environ.get("B", 12)), 1152, 28
N, K = 9 * D, D
g = torch.Generator(device=dev).manual_seed(1)
x = torch.randn(B, K, device=dev, generator=g)
Ws = [(torch.randn(N, K, device=dev, generator=g) * 0.03).to(bf16) for _ in range(nb)]
bs = [(torch.randn(N, device=dev, generator=g) * 0.1).to(bf16) for _ in range(nb)]
Wt, bt = ops.ptr_table(Ws), ops.ptr_table(bs)
dy = torch.randn(nb, B, N, device=dev, generator=g)
dWs = [torch.zeros(N, K, dtype=f32, device=dev) for _ in range(nb)]
dbs = [torch.zeros(N, dtype=f32, device=dev) for _ in range(nb)]
dWt, dbt = ops.ptr_table(dWs), ops.ptr_table(dbs)
dx = torch.zeros(B, K, dtype=f32, device=dev)


def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    t = []
    for _ in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n):
            fn()
        e.record(); torch.cuda.synchronize()
        t.append(s.elapsed_time(e) / n)
    return sorted(t)[2]


wb = nb * N * K * 2
t = timeit(lambda: ops.small_linear_fwd_batched(x, Wt, bt, nb, N, 1))
print(f"forward      {t * 1e3:8.1f} us  {wb / t / 1e9:7.2f} TB/s of weights")
t2 = timeit(lambda: ops.small_linear_bwd_batched(dy, x, Wt, dWt, dbt, dx, 1))
print(f"dW + dx      {t2 * 1e3:8.1f} us  ({wb / 1e6:.0f} MB of W read, {2 * wb / 1e6:.0f} MB of dW written)")
y = ops.small_linear_fwd_batched(x, Wt, bt, nb, N, 1)
ref = torch.nn.functional.silu(x).double() @ Ws[3].double().t() + bs[3].double()
print("forward rel err vs fp64:", ((y[3].double() - ref).norm() / ref.norm()).item())
dx.zero_(); ops.small_linear_bwd_batched(dy, x, Wt, dWt, dbt, dx, 1)
xr = x.double().requires_grad_(True)
tot = sum(((torch.nn.functional.silu(xr) @ Ws[i].double().t()) * dy[i].double()).sum() for i in range(nb))
tot.backward()
print("dx rel err vs fp64:", ((dx.double() - xr.grad).norm() / xr.grad.norm()).item())
