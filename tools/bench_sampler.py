"""Sampler throughput: DiT-XL/2, the reference's 512-px latent [1,16,16,64,64] (8192+16
tokens), Euler steps with classifier-free guidance 6.0, batched cond/uncond forward.  Prints one JSON line."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import build_model, WORKLOADS, CC, LC, PEAK_BF16_TFLOPS, step_flops
from video_diffusion_speedrun_amd.sampling import generate_latents

steps = int(os.environ.get("STEPS", 10))
kw, latent_shape, _, desc = WORKLOADS["c3b"]
dev = torch.device("cuda", 0)
model = build_model(kw, dev, seed=1234)
ctx = torch.randn(1, LC, CC, device=dev).to(torch.bfloat16)
generate_latents(model, ctx, inference_steps=2, cfg_scale=6.0, seed=0)  # warm-up
torch.cuda.synchronize()
t0 = time.perf_counter()
acc = generate_latents(model, ctx, inference_steps=steps, cfg_scale=6.0, seed=1)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
fwd_flops = step_flops(kw, latent_shape) / 3.0  # one forward, one sample
print(json.dumps({"metric": "sampler Euler steps/sec (CFG, 2 forwards per step)", "value": steps / dt, "unit": "steps/s",
                  "ms_per_step": dt / steps * 1e3, "latents_per_sec_at_50_steps": 1.0 / (dt / steps * 50),
                  "mfma_util": 2 * fwd_flops * steps / dt / (PEAK_BF16_TFLOPS * 1e12),
                  "config": {"workload": desc, "steps": steps, "cfg_scale": 6.0}, "latent_std": float(acc.std())}))
