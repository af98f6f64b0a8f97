"""Euler + classifier-free-guidance sampler on the HIP forward path (reference:
sampling/sample.py::generate_image, lines 77-159).

    acc = generate_latents(dit, prompt_embeds, inference_steps=50, cfg_scale=6.0, height=512, width=512, seed=42)

Returns the fp32 latent accumulator [1,16,16,2*(h//16),2*(w//16)] that the reference hands to the
Cosmos decoder (third-party, out of scope).  MI355X-first differences from the reference loop:
the prompt and the zeroed negative embeddings go through ONE batched forward (B=2) per step instead
of two forwards (they then share the step's random RoPE offsets -- the reference draws a second
random triple for the unconditional call), and the guidance + Euler update is one fused kernel.
`rope_starts` pins the offsets of every model call (cond, uncond per step) for parity tests; with
distinct offsets inside a step the two calls run separately, like the reference.
"""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import torch

from . import ops

bf16, f32 = torch.bfloat16, torch.float32
ALPHA = 8.0  # sample.py:131


def shifted_times(i: int, steps: int) -> Tuple[float, float]:
    """(t, t_next) of step i = steps..1 after the alpha = 8 shift (sample.py:126-134)."""
    t, t_next = i / steps, (i - 1) / steps
    t = t * ALPHA / (1 + (ALPHA - 1) * t)
    t_next = t_next * ALPHA / (1 + (ALPHA - 1) * t_next)
    return t, t_next


@torch.no_grad()
def generate_latents(model, prompt_embeds: torch.Tensor, negative_embeds: Optional[torch.Tensor] = None,
                     inference_steps: int = 50, cfg_scale: float = 6.0, height: int = 512, width: int = 512,
                     seed: int = 42, latents: Optional[torch.Tensor] = None,
                     rope_starts: Optional[Sequence[Tuple[int, int, int]]] = None, device="cuda") -> torch.Tensor:
    dev = torch.device(device)
    prompt_embeds = prompt_embeds.to(dev, bf16)
    guided = cfg_scale > 1
    if guided and negative_embeds is None:
        negative_embeds = torch.zeros_like(prompt_embeds)  # sample.py:105
    if latents is None:
        gen = torch.Generator(device=dev).manual_seed(seed)
        latents = torch.randn((1, 16, 16, 2 * (height // 16), 2 * (width // 16)), device=dev, dtype=bf16, generator=gen)
    lat = latents.to(dev, bf16).contiguous().clone()
    acc = lat.to(f32)
    call = 0
    for i in range(inference_steps, 0, -1):
        t, t_next = shifted_times(i, inference_steps)
        dt = t - t_next
        if guided:
            s_c = tuple(rope_starts[call]) if rope_starts is not None else None
            s_u = tuple(rope_starts[call + 1]) if rope_starts is not None else None
            call += 2
            if s_c == s_u:  # one batched forward: [cond, uncond]
                tt = torch.tensor([t, t], device=dev).to(bf16)
                out = model(torch.cat([lat, lat]), torch.cat([prompt_embeds, negative_embeds.to(dev, bf16)]), tt,
                            **({"rope_start": s_c} if s_c is not None else {}))
                cond, uncond = out[0:1].contiguous(), out[1:2].contiguous()
            else:
                tt = torch.tensor([t], device=dev).to(bf16)
                cond = model(lat, prompt_embeds, tt, rope_start=s_c)
                uncond = model(lat, negative_embeds.to(dev, bf16), tt, rope_start=s_u)
        else:
            tt = torch.tensor([t], device=dev).to(bf16)
            s_c = tuple(rope_starts[call]) if rope_starts is not None else None
            call += 1
            cond = model(lat, prompt_embeds, tt, **({"rope_start": s_c} if s_c is not None else {}))
            uncond = None
        ops.cfg_euler_step(cond, uncond, acc, lat, cfg_scale, dt)
    return acc
