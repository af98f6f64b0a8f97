"""Train for a while on a small fixed set of synthetic batches (bf16 and fp8 GEMMs, same seeds) and log the loss:
a stability / convergence check of the whole HIP train step (not a throughput number)."""
import os, sys, json, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from video_diffusion_speedrun_amd.optim import MuAdamW
from video_diffusion_speedrun_amd.train import get_schedule, train_step

steps = int(os.environ.get("STEPS", 120))
wl = os.environ.get("WORKLOAD", "c3a")
B = int(os.environ.get("B", 4))
kw, latent_shape, _, desc = bench.WORKLOADS[wl]
dev = torch.device("cuda", 0)
out = {"workload": desc, "per_gpu_batch": B, "steps": steps, "runs": {}}
MODES = os.environ.get("MODES", "bf16,fp8").split(",")  # fp8x = fp8 with the cross-attention products in fp8 too
for mode in MODES:
    model = bench.build_model(kw, dev, seed=1234)
    if mode.startswith("fp8"):
        model.enable_fp8(cross_attention=(mode == "fp8x"))
    groups, _ = model.get_mup_setup(float(os.environ.get("LR", 3e-4)), 0.1, ["patch_proj", "context_kv", "positional_embedding"])
    opt = MuAdamW(groups, betas=(0.95, 0.99))
    sched = get_schedule(opt, "cosine", 10, steps)
    g = torch.Generator(device=dev).manual_seed(7)
    batches = [{"latent": torch.randn(B, *latent_shape, device=dev, generator=g).to(torch.bfloat16),
                "context": torch.randn(B, bench.LC, bench.CC, device=dev, generator=g).to(torch.bfloat16)} for _ in range(4)]
    gen = torch.Generator(device=dev).manual_seed(11)
    torch.manual_seed(11)
    losses = []
    for s in range(steps):
        losses.append(train_step(model, opt, sched, batches[s % 4], dev, generator=gen).detach())
    losses = [float(l) for l in torch.stack(losses).cpu()]
    assert all(math.isfinite(l) for l in losses), "non-finite loss"
    out["runs"][mode] = {"first10_mean": sum(losses[:10]) / 10, "last10_mean": sum(losses[-10:]) / 10,
                         "every10": [round(sum(losses[i:i + 10]) / 10, 4) for i in range(0, steps, 10)]}
    del model, opt
    torch.cuda.empty_cache()
a = out["runs"][MODES[0]]["every10"]
for m in MODES[1:]:
    out["max_rel_gap_of_10step_means" + ("" if m == "fp8" else "_" + m)] = max(abs(x - y) / x for x, y in zip(a, out["runs"][m]["every10"]))
print(json.dumps(out))
