"""Train-step harness with the reference's `train.py::forward` signature (train.py:51-145) and
the loop body of train.py:431-434, on the HIP path.

    total_loss, diffusion_loss = forward(dit_model, batch, text_encoder, tokenizer, device,
                                         global_step, master_process, generator=None, ...)
    optimizer.zero_grad(); total_loss.backward(); optimizer.step(); lr_scheduler.step()

`batch` = {"latent": Tensor[B,C,T,H,W], "prompt": list[str]}; the text encoder is the caller's
(frozen, third-party; out of scope) -- for synthetic runs pass `batch["context"]` (a pre-encoded
[B,Lc,Cc] tensor) and text_encoder=None.
"""
from __future__ import annotations

import logging
from typing import Optional

import torch

from . import ops

bf16, f32 = torch.bfloat16, torch.float32
ALPHA = 8.0  # time shift, train.py:95


def encode_prompt_with_t5(text_encoder, tokenizer, max_sequence_length=512, prompt=None, device=None,
                          return_index=-1):
    """Caption -> encoder hidden states [B, max_sequence_length, C] in the encoder's dtype, same call contract as
    the reference's helper (utils.py:38-80).  The encoder is the caller's frozen third-party model: host glue only."""
    prompts = [prompt] if isinstance(prompt, str) else list(prompt)
    tokens = tokenizer(prompts, padding="max_length", max_length=max_sequence_length, truncation=True,
                       return_length=False, return_overflowing_tokens=False, return_tensors="pt")
    states = text_encoder(tokens.input_ids.to(device), return_dict=True, output_hidden_states=True).hidden_states
    chosen = states[return_index]
    if return_index != -1:  # an inner layer: finish it the way the encoder finishes its last one
        chosen = text_encoder.encoder.dropout(text_encoder.encoder.final_layer_norm(chosen))
    return chosen.to(device=device, dtype=text_encoder.dtype)


class _FlowLoss(torch.autograd.Function):
    """mean_b mean_{chw} (v - out)^2 in fp32 (train.py:121-125); HIP kernels for value and gradient.  The
    backward honours the upstream gradient (`(loss / n).backward()`, loss scaling): it is read from device
    memory by the kernel, so nothing synchronises and a captured step stays replayable."""

    @staticmethod
    def forward(ctx, out, v):
        loss, per, _ = ops.flow_loss(v, out, want_grad=False)
        ctx.save_for_backward(out, v)
        ctx.mark_non_differentiable(per)
        return loss.reshape(()), per

    @staticmethod
    def backward(ctx, gloss, _gper):
        out, v = ctx.saved_tensors
        return ops.flow_loss_bwd(v, out, gloss.to(f32).reshape(1).contiguous()), None


def flow_loss(out, v):
    return _FlowLoss.apply(out, v)


def time_shift_from_normal(z: torch.Tensor) -> torch.Tensor:
    """t = sigmoid(z); t = a t / (1 + (a-1) t), in z.dtype (bf16) like train.py:93-96.
    B scalars: left to torch (host-side plumbing, not a hot-path kernel)."""
    t = torch.sigmoid(z)
    return t * ALPHA / (1 + (ALPHA - 1) * t)


def forward(dit_model, batch, text_encoder, tokenizer, device, global_step, master_process, generator=None,
            binnings=None, batch_size=None, return_index=-1, rope_start=None):
    logger = logging.getLogger(__name__)
    vae_latent = batch["latent"].to(device).to(bf16)
    with torch.no_grad():
        if "context" in batch and text_encoder is None:
            caption_encoded = batch["context"].to(device)
        else:
            caption_encoded = encode_prompt_with_t5(text_encoder, tokenizer, prompt=batch["prompt"], device=device,
                                                    return_index=return_index)
        caption_encoded = caption_encoded.to(bf16)
        do_zero_out = torch.rand(caption_encoded.shape[0], device=device) < 0.01  # train.py:86
        # same result as `caption_encoded[do_zero_out] = 0` without the host sync of a boolean index
        # (capturable) and without writing into the caller's batch["context"]
        caption_encoded = torch.where(do_zero_out[:, None, None], torch.zeros((), dtype=bf16, device=device),
                                      caption_encoded)
    B = vae_latent.size(0)
    z = torch.randn(B, device=device, dtype=bf16, generator=generator)
    t = time_shift_from_normal(z)
    noise = torch.randn(vae_latent.shape, device=device, dtype=bf16, generator=generator)
    z_t, v_objective = ops.noise_latents(vae_latent.contiguous(), noise, t.to(f32))
    output = dit_model(z_t, caption_encoded, t, rope_start=rope_start) if rope_start is not None \
        else dit_model(z_t, caption_encoded, t)
    diffusion_loss, _per = flow_loss(output, v_objective)
    total_loss = diffusion_loss
    if master_process:
        logger.debug("forward done (step %s)", global_step)
    return total_loss, diffusion_loss


def lr_lambda(step: int, kind: str, warmup: int, total: int) -> float:
    """multiplier of HF get_{cosine,linear}_schedule_with_warmup; "constant" is the reference's
    linear schedule with 1e10 total steps (train.py:349-364)."""
    if kind == "constant":
        kind, total = "linear", 10_000_000_000
    if step < warmup:
        return step / max(1, warmup)
    if kind == "linear":
        return max(0.0, (total - step) / max(1, total - warmup))
    if kind != "cosine":
        raise ValueError(f"unknown lr_scheduler_type {kind}")
    import math
    prog = (step - warmup) / max(1, total - warmup)
    return max(0.0, 0.5 * (1.0 + math.cos(math.pi * prog)))


def get_schedule(optimizer, lr_scheduler_type: str = "cosine", num_warmup_steps: int = 20,
                 num_training_steps: int = 10000):
    """the LR scheduler of train.py:349-364 as a torch LambdaLR (no transformers import needed)"""
    return torch.optim.lr_scheduler.LambdaLR(
        optimizer, lambda s: lr_lambda(s, lr_scheduler_type, num_warmup_steps, num_training_steps))


def train_step(dit_model, optimizer, lr_scheduler, batch, device, generator=None, rope_start=None):
    """train.py:412-434 for one batch with a pre-encoded context; returns the loss tensor."""
    total_loss, _ = forward(dit_model, batch, None, None, device, 0, False, generator=generator,
                            rope_start=rope_start)
    optimizer.zero_grad()
    total_loss.backward()
    optimizer.step()
    if lr_scheduler is not None:
        lr_scheduler.step()
    return total_loss
