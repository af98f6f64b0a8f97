"""muP AdamW: `torch.optim.AdamW(groups, betas=(0.95, 0.99), fused=True)` of the reference
(train.py:335-344,433) as ONE multi-tensor HIP launch over every (sharded) fp32 parameter, which
also refreshes the bf16 compute copy (saving the cast pass of the next forward / all-gather).

Use exactly like the reference:
    groups, settings = dit.get_mup_setup(lr, 0.1, ["patch_proj", "context_kv", "positional_embedding"])
    opt = MuAdamW(groups, betas=(0.95, 0.99))
It is a torch.optim.Optimizer, so HF / torch LR schedulers drive `param_groups[i]["lr"]` as usual.
"""
from __future__ import annotations

import ctypes as C
from typing import List

import torch

from . import _lib, ops

CHUNK = 1 << 16


class MuAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.95, 0.99), eps=1e-8, weight_decay=1e-2, fused=True):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self._step = 0
        self._table_key = None
        self._chunk_key = None
        self._dev = {}
        self._dev_scalars = None  # device float[3]: see use_device_scalars

    def use_device_scalars(self, device):
        """Keep the per-step scalars (bias corrections, LR multiplier) in device memory instead of kernel
        arguments, so that a captured step can be replayed (graph.py).  `step()` refreshes them with a
        stream-ordered copy unless the stream is capturing; a replay driver calls `advance()` itself."""
        if self._dev_scalars is None:
            self._dev_scalars = torch.zeros(3, dtype=torch.float32, device=device)
        return self._dev_scalars

    def _scalars(self, mult):
        import numpy as np
        b1, b2 = (np.float32(b) for b in self.param_groups[0]["betas"])
        one, st = np.float32(1), np.float32(self._step)
        bc1 = one - np.power(b1, st)
        bc2 = one - np.power(b2, st)
        return torch.tensor([bc1, one / np.sqrt(bc2), np.float32(mult)], dtype=torch.float32)

    def advance(self):
        """host half of one captured step: count it and push its scalars (call before the replay)"""
        plist = [(p, g) for g in self.param_groups for p in g["params"] if p.grad is not None and p.numel() > 0]
        lrs, mult = self._lr_plan(plist)
        if self._table_key is not None and tuple(k[2] for k in self._table_key) != tuple(lrs):
            raise RuntimeError("MuAdamW: the base learning rates changed under a captured step; capture again")
        self._step += 1
        self._dev_scalars.copy_(self._scalars(mult))

    def _build_chunks(self, plist, device):
        chunk_t: List[int] = []
        chunk_s: List[int] = []
        for i, (p, _) in enumerate(plist):
            st = self.state[p]
            if "exp_avg" not in st or st["exp_avg"].numel() != p.numel():
                st["exp_avg"] = torch.zeros(p.numel(), dtype=torch.float32, device=device)
                st["exp_avg_sq"] = torch.zeros(p.numel(), dtype=torch.float32, device=device)
            n = p.numel()
            chunk_t.extend([i] * ((n + CHUNK - 1) // CHUNK))
            chunk_s.extend(range(0, n, CHUNK))
        self._dev["ct"] = torch.tensor(chunk_t, dtype=torch.int32, device=device)
        self._dev["cs"] = torch.tensor(chunk_s, dtype=torch.int64, device=device)
        self._dev["n"] = len(chunk_t)

    def _lr_plan(self, plist):
        """(per-param base lr list, common multiplier): LR schedulers scale every group's lr by the same
        factor of its `initial_lr` (train.py:349-364), so the descriptor table keeps the base lr and the
        factor travels as a kernel argument -- no table rebuild / H2D copy per step."""
        mult = None
        for _, g in plist:
            base = g.get("initial_lr", None)
            if base is None or base == 0.0:
                mult = None
                break
            r = g["lr"] / base
            if mult is None:
                mult = r
            elif abs(r - mult) > 1e-9 * max(abs(mult), 1e-30):
                mult = None
                break
        if mult is None:
            return [float(g["lr"]) for _, g in plist], 1.0
        return [float(g["initial_lr"]) for _, g in plist], float(mult)

    def _build_descs(self, plist, device, lrs):
        descs = (_lib.AdamWTensor * len(plist))()
        for i, (p, group) in enumerate(plist):
            st = self.state[p]
            shadow = getattr(p, "_vds_shadow", None)
            d = descs[i]
            d.p, d.g = p.data.data_ptr(), p.grad.data_ptr()
            d.m, d.v = st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()
            d.p_bf16 = shadow.data_ptr() if shadow is not None and shadow.numel() == p.numel() else None
            d.numel = p.numel()
            d.lr, d.wd = lrs[i], float(group["weight_decay"])
        self._dev["desc"] = torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8).to(device)

    def state_dict(self):
        d = super().state_dict()
        d["vds_step"] = self._step
        return d

    def load_state_dict(self, state_dict):
        state_dict = dict(state_dict)
        self._step = int(state_dict.pop("vds_step", 0))
        super().load_state_dict(state_dict)
        self._table_key = self._chunk_key = None  # moments were replaced: rebuild the descriptor table

    @torch.no_grad()
    def step(self, closure=None):
        assert closure is None
        plist = []
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is None or p.numel() == 0:
                    continue
                if not p.is_cuda:
                    raise RuntimeError("MuAdamW is a HIP kernel: parameters must be on the GPU")
                assert p.dtype == torch.float32 and p.grad.dtype == torch.float32
                assert p.data.is_contiguous() and p.grad.is_contiguous()
                plist.append((p, group))
        if not plist:
            return None
        self._step += 1
        device = plist[0][0].device
        ckey = tuple(p.numel() for p, _ in plist)
        if ckey != self._chunk_key:
            self._build_chunks(plist, device)
            self._chunk_key, self._table_key = ckey, None
        lrs, mult = self._lr_plan(plist)
        key = tuple((p.data.data_ptr(), p.grad.data_ptr(), lr, g["weight_decay"]) for (p, g), lr in zip(plist, lrs))
        if key != self._table_key:  # pointers or the base lr changed: refresh the descriptors
            self._build_descs(plist, device, lrs)
            self._table_key = key
        b1, b2 = self.param_groups[0]["betas"]
        eps = self.param_groups[0]["eps"]
        if self._dev_scalars is not None:
            if not torch.cuda.is_current_stream_capturing():
                self._dev_scalars.copy_(self._scalars(mult))
            _lib.check(_lib.load().vds_adamw_multi_dev(self._dev["desc"].data_ptr(), self._dev["ct"].data_ptr(),
                                                       self._dev["cs"].data_ptr(), self._dev["n"], CHUNK, b1, b2, eps,
                                                       self._dev_scalars.data_ptr(), 1.0, ops._stream()),
                       "vds_adamw_multi_dev")
        else:
            _lib.check(_lib.load().vds_adamw_multi(self._dev["desc"].data_ptr(), self._dev["ct"].data_ptr(),
                                                   self._dev["cs"].data_ptr(), self._dev["n"], CHUNK, b1, b2, eps,
                                                   self._step, mult, 1.0, ops._stream()), "vds_adamw_multi")
        # the bf16 shadows of these flat groups are now current: the next forward skips its cast pass
        for g in {id(getattr(p, "_vds_group", None)): getattr(p, "_vds_group", None) for p, _ in plist}.values():
            if g is not None:
                g.mark_shadow_fresh()
        return None
