"""fp8 vs bf16 GEMM at the DiT-XL shapes (HIP events), plus the quantiser's GB/s."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_diffusion_speedrun_amd import ops
from video_diffusion_speedrun_amd._lib import EPI_F32, EPI_STORE

bf16, f32 = torch.bfloat16, torch.float32
dev = "cuda"


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
    return s.elapsed_time(e) / iters * 1e-3


def rnd(*shape, scale=1.0):
    return (torch.randn(*shape, device=dev) * scale).to(bf16)


B, L, D = int(os.environ.get("B", 6)), 8208, 1152
M = B * L
for name, N, K in (("qkv", 3 * D, D), ("proj", D, D), ("fc1", 4 * D, D), ("fc2", D, 4 * D)):
    x, w = rnd(M, K), rnd(N, K, scale=0.03)
    y = torch.empty(M, N, dtype=bf16, device=dev)
    xq, xt, sx = ops.quant_fp8(x, 0, ops.absmax(x), True, True)
    wq, wt, sw = ops.quant_fp8(w, 0, ops.absmax(w), True, True)
    dy = rnd(M, N)
    dyq, dyt, sd = ops.quant_fp8(dy, 1, ops.absmax(dy), True, True)
    dW = torch.zeros(N, K, dtype=f32, device=dev)
    dx = torch.empty(M, K, dtype=bf16, device=dev)
    fl = 2 * M * N * K
    tb = timeit(lambda: ops.linear_fwd(x, w, None, out=y))
    tf = timeit(lambda: ops.gemm_fp8(EPI_STORE, M, N, K, xq, wq, sx, sw, 0, y, N))
    print(f"fwd   {name:5s} M{M} N{N} K{K}: bf16 {tb*1e3:7.3f} ms {fl/tb/1e12:7.1f} TF/s | fp8 {tf*1e3:7.3f} ms {fl/tf/1e12:7.1f} TF/s  x{tb/tf:.2f}")
    tb = timeit(lambda: ops.linear_dgrad(dy, w))
    tf = timeit(lambda: ops.gemm_fp8(EPI_STORE, M, K, N, dyq, wt, sd, sw, 1, dx, K))
    print(f"dgrad {name:5s}                  : bf16 {tb*1e3:7.3f} ms {fl/tb/1e12:7.1f} TF/s | fp8 {tf*1e3:7.3f} ms {fl/tf/1e12:7.1f} TF/s  x{tb/tf:.2f}")
    tb = timeit(lambda: ops.linear_wgrad(dy, x, dW))
    tiles = ((N + 255) // 256) * ((K + 255) // 256)
    for split in sorted({max(1, 256 // tiles), max(1, 512 // tiles), max(1, 768 // tiles)}):
        tf = timeit(lambda: ops.gemm_fp8(EPI_F32, N, K, M, dyt, xt, sd, sx, 1, dW, K, split_k=-split))
        print(f"wgrad {name:5s} split {split:2d}         : bf16 {tb*1e3:7.3f} ms {fl/tb/1e12:7.1f} TF/s | fp8 {tf*1e3:7.3f} ms {fl/tf/1e12:7.1f} TF/s  x{tb/tf:.2f}")
    tq = timeit(lambda: ops.quant_fp8(x, 0, sx, True, True))
    ta = timeit(lambda: ops.absmax(x))
    print(f"quant+T {name:5s} [{M},{K}]: {tq*1e6:7.1f} us {4*M*K/tq/1e9:7.0f} GB/s ; absmax {ta*1e6:7.1f} us {2*M*K/ta/1e9:7.0f} GB/s")

# fused fp8 emission from the fc1 / fc2-dgrad epilogues
from video_diffusion_speedrun_amd import fp8 as F8
x, w1, b1 = rnd(M, D), rnd(4 * D, D, scale=0.03), rnd(4 * D, scale=0.1)
xq, w1q = F8.Q(x, 0, True, True), F8.Q(w1, 0, True, True)
pre, act = F8.fwd_gelu(xq, w1q, b1)
amax = ops.absmax(act)
rec = torch.zeros(1, device=dev)
t0 = timeit(lambda: F8.fwd_gelu(xq, w1q, b1))
t1 = timeit(lambda: F8.fwd_gelu_emit(xq, w1q, b1, amax, rec, False))
t2 = timeit(lambda: F8.fwd_gelu_emit(xq, w1q, b1, amax, rec, True))
t3 = timeit(lambda: F8.Q(act, 0, True, True))
print(f"fc1 fwd: plain {t0*1e6:.0f} us | emit q {t1*1e6:.0f} us | emit q+T {t2*1e6:.0f} us | separate absmax+quant {t3*1e6:.0f} us")
dy, w2 = rnd(M, D), rnd(D, 4 * D, scale=0.03)
dyq, w2q = F8.Q(dy, 1, True, True), F8.Q(w2, 0, True, True)
dh = F8.dgrad(dyq, w2q, pre=pre)
amax = ops.absmax(dh)
cs = torch.zeros(4 * D, device=dev)
t0 = timeit(lambda: F8.dgrad(dyq, w2q, pre=pre))
t1 = timeit(lambda: F8.dgrad_gelu_emit(dyq, w2q, pre, amax, rec, None))
t2 = timeit(lambda: F8.dgrad_gelu_emit(dyq, w2q, pre, amax, rec, cs))
t3 = timeit(lambda: ops.colsum(dh, cs))
print(f"fc2 dgrad: plain {t0*1e6:.0f} us | emit q+T {t1*1e6:.0f} us | emit q+T+colsum {t2*1e6:.0f} us | separate colsum {t3*1e6:.0f} us")
