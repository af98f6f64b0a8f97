"""host-side cost of one launch through ops.py (launch-bound workloads: C1 is ~1600 launches of a few microseconds)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_diffusion_speedrun_amd import ops, _lib

torch.cuda.init()
x = torch.zeros(64, 64, dtype=torch.bfloat16, device="cuda")
N = 200000


def t(fn, n=N):
    fn()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    return (time.perf_counter() - t0) / n * 1e6


print("torch.cuda.current_stream().cuda_stream : %.2f us" % t(lambda: torch.cuda.current_stream().cuda_stream))
print("torch._C._cuda_getCurrentRawStream(dev) : %.2f us" % t(lambda: torch._C._cuda_getCurrentRawStream(0)))
print("torch.cuda.current_device()             : %.2f us" % t(lambda: torch.cuda.current_device()))
print("tensor.data_ptr()                       : %.2f us" % t(lambda: x.data_ptr()))
print("torch.empty(64,64,bf16,cuda)            : %.2f us" % t(lambda: torch.empty(64, 64, dtype=torch.bfloat16, device=x.device), 50000))
lib = _lib.load()
print("lib.vds_prof_enable(0) (ctypes call)    : %.2f us" % t(lambda: lib.vds_prof_enable(0)))
y = torch.zeros(64, 64, dtype=torch.float32, device="cuda")
torch.cuda.synchronize()
print("ops.cast_f32_bf16 (one tiny launch, queue not drained): %.2f us" % t(lambda: ops.cast_f32_bf16(y.view(-1), x.view(-1)), 20000))
torch.cuda.synchronize()
