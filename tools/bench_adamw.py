"""AdamW kernel alone on DiT-XL-sized state (HIP events): GB/s of the 30 B/parameter stream."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_diffusion_speedrun_amd.optim import MuAdamW

n = int(os.environ.get("N", 675_000_000)) // 8 * 8
ps = []
for i in range(8):  # a few big tensors like the flat groups
    p = torch.nn.Parameter(torch.randn(n // 8, device="cuda"))
    p.grad = torch.randn(n // 8, device="cuda") * 1e-3
    p._vds_shadow = torch.empty(n // 8, dtype=torch.bfloat16, device="cuda")
    ps.append(p)
opt = MuAdamW([{"params": ps, "lr": 1e-4, "weight_decay": 0.1}], betas=(0.95, 0.99))
for _ in range(3):
    opt.step()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10):
    opt.step()
e.record()
torch.cuda.synchronize()
t = s.elapsed_time(e) / 10 * 1e-3
print(f"adamw {n/1e6:.0f}M params: {t*1e3:.3f} ms  {30*n/t/1e9:.0f} GB/s  lib={os.environ.get('VDS_LIB_PATH','default')}")
