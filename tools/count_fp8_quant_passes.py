"""Which fp8 operands of a steady-state C5 train step are still quantised by a SEPARATE pass over a bf16 tensor
(ops.quant_fp8 / ops.absmax calls of the 4th step, by shape and format)?    python tools/count_fp8_quant_passes.py"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from video_diffusion_speedrun_amd import ops
from video_diffusion_speedrun_amd.optim import MuAdamW
from video_diffusion_speedrun_amd.train import get_schedule, train_step

cnt = collections.Counter()
orig_q, orig_a = ops.quant_fp8, ops.absmax


def q(x, fmt, amax, rowmajor=True, transposed=False, **kw):
    cnt[("quant_fp8", tuple(x.shape), "e5m2" if fmt else "e4m3", "row-major" if rowmajor else "", "transposed" if transposed else "")] += 1
    return orig_q(x, fmt, amax, rowmajor, transposed, **kw)


def a(x, *args, **kw):
    cnt[("absmax", tuple(x.shape))] += 1
    return orig_a(x, *args, **kw)


ops.quant_fp8, ops.absmax = q, a
B = int(os.environ.get("B", 2))
device = torch.device("cuda", 0)
kw, latent_shape, _, desc = bench.WORKLOADS["c5"]
model = bench.build_model(kw, device, seed=1234)
model.enable_fp8()
groups, _ = model.get_mup_setup(1e-4, 0.1, ["patch_proj", "context_kv", "positional_embedding"])
opt = MuAdamW(groups, betas=(0.95, 0.99))
sched = get_schedule(opt, "cosine", 20, 10000)
gen = torch.Generator(device=device).manual_seed(1234)
batch = {"latent": torch.randn(B, *latent_shape, device=device, generator=gen).to(torch.bfloat16),
         "context": torch.randn(B, bench.LC, bench.CC, device=device, generator=gen).to(torch.bfloat16), "prompt": [""] * B}
for i in range(4):
    cnt.clear()
    train_step(model, opt, sched, batch, device, generator=gen)
torch.cuda.synchronize()
print(desc, "B =", B, "-- separate passes in step 4:")
for k, v in sorted(cnt.items(), key=lambda kv: -kv[1]):
    print(f"{v:5d} x", *[s for s in k if s != ""])
