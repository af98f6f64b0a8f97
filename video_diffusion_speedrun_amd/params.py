"""Flat parameter groups: the unit of casting, all-gather, reduce-scatter and optimizer update.

The reference wraps every DiTBlock and the root module with FSDP2 `fully_shard`
(model.py:512-542): fp32 master parameters sharded over the ranks, bf16 all-gather before use,
fp32 reduce-scatter(avg) of the gradients.  MI355X-first re-design of the same contract:

  * every group (one per DiTBlock + one root group) is ONE flat buffer; parameters are
    16-element aligned slices of it, so a group needs one cast kernel, ONE all-gather and ONE
    reduce-scatter (few, large, contiguous collectives -- what point-to-point xGMI wants),
    instead of per-parameter dim-0 shards with copy-in / copy-out;
  * rank r owns the contiguous chunk [r*S, (r+1)*S) of the flat buffer: fp32 master + AdamW
    state exist only for that chunk; `named_parameters()` exposes, per name, the rank-local piece
    (the full tensor when world == 1, a 1-D slice -- possibly empty -- otherwise), which is all
    an element-wise optimizer and `get_mup_setup` need;
  * world == 1 degenerates to "cast fp32 -> bf16" with no communication (the reference cannot run
    at world_size 1 at all: model.py:489).

This file is pure host logic (no kernels): the cast function and the collectives are injected, so
the multi-rank behaviour is unit-tested on CPU with gloo.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

ALIGN = 16  # elements; keeps every slice 32-B (bf16) / 64-B (fp32) aligned


def _numel(shape) -> int:
    n = 1
    for s in shape:
        n *= int(s)
    return n


class FlatGroup:
    def __init__(self, name: str, named_params: Sequence[Tuple[str, torch.nn.Parameter]], world: int = 1,
                 rank: int = 0, process_group=None):
        self.name, self.world, self.rank = name, world, rank
        self.pg = process_group  # the shard group's communicator (None = the default group)
        self.names: List[str] = []
        self.shapes: Dict[str, Tuple[int, ...]] = {}
        self.offsets: Dict[str, int] = {}
        self.params: Dict[str, torch.nn.Parameter] = {}
        off = 0
        for n, p in named_params:
            self.names.append(n)
            self.shapes[n] = tuple(getattr(p, "_vds_full_shape", p.shape))
            self.offsets[n] = off
            self.params[n] = p
            off += (_numel(self.shapes[n]) + ALIGN - 1) // ALIGN * ALIGN
        self.total = off
        q = world * 256
        self.padded = (off + q - 1) // q * q
        self.shard = self.padded // world
        self.device: Optional[torch.device] = None
        self.master = self.shadow = self.full = self.gfull = self.gshard = None
        self.gathered = False
        self._fresh_versions = None  # see mark_shadow_fresh

    # ---- layout -------------------------------------------------------------------------
    def local_range(self, name: str) -> Tuple[int, int]:
        """[lo, hi) of the flat indices of `name` owned by this rank, relative to the shard."""
        lo, hi = self.offsets[name], self.offsets[name] + _numel(self.shapes[name])
        s0, s1 = self.rank * self.shard, (self.rank + 1) * self.shard
        lo, hi = max(lo, s0), min(hi, s1)
        if hi <= lo:
            return 0, 0
        return lo - s0, hi - s0

    def materialize(self, device, full_values: Optional[Dict[str, torch.Tensor]] = None, separate: bool = False,
                    gfull: Optional[torch.Tensor] = None):
        """Allocate the flat buffers on `device`, move the current parameter values into the fp32
        master chunk and re-point every nn.Parameter at its slice.  `full_values` supplies the
        full tensors when the parameters are already sharded pieces.  `gfull`: this group's `padded` floats of a
        gradient arena the caller owns (DiT keeps the full-size gradient buffers of all groups in ONE allocation, so
        that a backward pass clears them with one fill instead of one per group)."""
        device = torch.device(device)
        new_master = torch.zeros(self.shard, dtype=torch.float32, device=device)
        for n in self.names:
            lo, hi = self.local_range(n)
            if hi == lo:
                continue
            p = self.params[n]
            if full_values is not None:
                src = full_values[n]
            else:
                src = p.data
            src = src.detach().reshape(-1)
            if src.numel() == _numel(self.shapes[n]):  # full tensor: cut out this rank's piece
                g0 = self.rank * self.shard + lo - self.offsets[n]
                src = src[g0:g0 + (hi - lo)]
            assert src.numel() == hi - lo, (n, src.numel(), hi - lo)
            new_master[lo:hi].copy_(src.to(device=device, dtype=torch.float32))
        self.device = device
        self.master = new_master
        self.shadow = torch.zeros(self.shard, dtype=torch.bfloat16, device=device)
        if gfull is not None:
            assert gfull.numel() == self.padded and gfull.dtype == torch.float32 and gfull.device == device
            self.gfull = gfull
        else:
            self.gfull = torch.zeros(self.padded, dtype=torch.float32, device=device)
        if self.world == 1 and not separate:
            self.full, self.gshard = self.shadow, self.gfull
        else:
            self.full = torch.zeros(self.padded, dtype=torch.bfloat16, device=device)
            self.gshard = torch.zeros(self.shard, dtype=torch.float32, device=device)
        self._repoint()
        self.gathered = False
        self._fresh_versions = None

    def _repoint(self):
        for n in self.names:
            lo, hi = self.local_range(n)
            p = self.params[n]
            piece = self.master[lo:hi]
            if self.world == 1:
                piece = piece.view(self.shapes[n])
            p.data = piece
            p._vds_full_shape = self.shapes[n]
            p._vds_group = self
            p._vds_name = n
            gp = self.gshard[lo:hi]
            p._vds_grad_view = gp.view(self.shapes[n]) if self.world == 1 else gp
            p._vds_shadow = self.shadow[lo:hi]

    def is_current(self) -> bool:
        """True while every parameter still aliases its slice of the master buffer (a
        `.to()` / `load_state_dict(assign=True)` replaces the tensors and breaks that)."""
        if self.master is None:
            return False
        base = self.master.data_ptr()
        for n in self.names:
            lo, hi = self.local_range(n)
            p = self.params[n]
            if p.data.numel() != hi - lo or (hi > lo and p.data.data_ptr() != base + 4 * lo):
                return False
        return True

    # ---- bf16 compute copy bookkeeping ----------------------------------------------------
    def mark_shadow_fresh(self):
        """The writer of the fp32 master (MuAdamW's kernel) has written the bf16 shadow too, so the next
        `gather` may skip its cast pass.  The claim is tied to the autograd version counters of the
        parameters: any later in-place write through the nn.Parameters (`load_state_dict`, `p.mul_()`,
        `clip_grad`-style `p.copy_()`, an EMA swap ...) bumps a counter and voids it.  Writers that bypass the
        counters (`p.data.xxx_()`, raw pointers) must call `invalidate_shadow()` / `DiT.invalidate_compute_copy()`."""
        self._fresh_versions = tuple(p._version for p in self.params.values())

    def invalidate_shadow(self):
        self._fresh_versions = None

    @property
    def shadow_fresh(self) -> bool:
        return (self._fresh_versions is not None and
                self._fresh_versions == tuple(p._version for p in self.params.values()))

    def refresh_shadow(self, cast_fn) -> bool:
        """cast master -> shadow unless the shadow is known to be current; returns True when it cast"""
        if self.shadow_fresh:
            return False
        cast_fn(self.master, self.shadow)
        self.mark_shadow_fresh()
        # the parameters were written by something else than the fused optimizer: the fp8 path re-measures the weights'
        # amax instead of trusting its history (sticky: cleared by the training forward that did so, model.py)
        self.recast = True
        return True

    # ---- per-step operations ------------------------------------------------------------
    def gather(self, cast_fn: Callable[[torch.Tensor, torch.Tensor], None], group=None, skip_cast=False):
        """bf16 compute copy of the whole group: cast the local fp32 chunk, all-gather (C3)."""
        if not skip_cast:
            self.refresh_shadow(cast_fn)
        if self.world > 1 or (self.full is not self.shadow):
            all_gather_flat(self.full, self.shadow, group)
        self.gathered = True

    def reduce_grads(self, group=None):
        """fp32 reduce-scatter(avg) of the group's gradients into the local shard (C4)."""
        if self.world > 1 or (self.gshard is not self.gfull):
            reduce_scatter_avg(self.gshard, self.gfull, group)

    def publish_grads(self):
        """point `.grad` of every trainable parameter at its slice of the reduced gradient shard (frozen parameters --
        `requires_grad_(False)` -- get none: no optimizer would ever clear it again)"""
        for n in self.names:
            p = self.params[n]
            if p.requires_grad:
                p.grad = p._vds_grad_view

    def hold_grads(self) -> Optional[torch.Tensor]:
        """Gradients the caller already holds in `.grad` (a second backward before `zero_grad()`: micro-batch
        accumulation, like autograd's accumulate-into-.grad): a copy of the gradient shard in which the slices of
        parameters WITHOUT a held gradient (`.grad is None`: cleared by zero_grad, frozen, never used) are zero, to be
        added back by `add_held` after this backward has overwritten the shard.  None when nothing is held."""
        live = [n for n in self.names if self.params[n].grad is not None]
        if not live:
            return None
        for n in live:
            q = self.params[n]
            if q.grad.data_ptr() != q._vds_grad_view.data_ptr():
                raise RuntimeError("DiT backward: a parameter's .grad was replaced by a foreign tensor; "
                                   "call zero_grad(set_to_none=True) before backward")
        if len(live) == len(self.names):
            return self.gshard.clone()
        held = torch.zeros_like(self.gshard)
        for n in live:
            lo, hi = self.local_range(n)
            held[lo:hi].copy_(self.gshard[lo:hi])
        return held

    def add_held(self, held: Optional[torch.Tensor]):
        if held is not None:
            self.gshard.add_(held)

    def w(self, name: str) -> torch.Tensor:
        o = self.offsets[name]
        return self.full[o:o + _numel(self.shapes[name])].view(self.shapes[name])

    def g(self, name: str) -> torch.Tensor:
        o = self.offsets[name]
        return self.gfull[o:o + _numel(self.shapes[name])].view(self.shapes[name])

    def has(self, name: str) -> bool:
        return name in self.offsets

    def full_tensor(self, name: str) -> torch.Tensor:
        """fp32 full value of a parameter (all-gathers the master pieces when sharded)."""
        if self.world == 1:
            return self.params[name].data.detach().clone()
        return self.full_tensors([name])[name]

    def full_tensors(self, names: Optional[Sequence[str]] = None) -> Dict[str, torch.Tensor]:
        """fp32 full values of the group's parameters with ONE all-gather of the master shards (on the
        group's own communicator)."""
        names = list(self.names if names is None else names)
        if self.world == 1:
            return {n: self.params[n].data.detach().clone() for n in names}
        full = torch.empty(self.padded, dtype=torch.float32, device=self.device)
        all_gather_flat(full, self.master, self.pg)
        out = {}
        for n in names:
            o = self.offsets[n]
            out[n] = full[o:o + _numel(self.shapes[n])].view(self.shapes[n]).clone()
        return out


# ---- collectives -------------------------------------------------------------------------------
# GPU tensors: the library's own RCCL communicator (comm.py -> vds_all_gather_bf16 / vds_reduce_scatter_f32_avg,
# asynchronous on the current HIP stream).  CPU tensors (gloo; the tests of the host logic) and VDS_COMM=torch:
# torch.distributed.
def all_gather_flat(out: torch.Tensor, inp: torch.Tensor, group=None):
    from . import comm
    if out.is_cuda and comm.active_for(group):
        comm.all_gather(out, inp)
    elif dist.get_backend(group) == "gloo":
        parts = list(out.view(dist.get_world_size(group), -1).unbind(0))
        if inp.dtype == torch.bfloat16:  # gloo has no bf16: move the raw bits
            dist.all_gather([p.view(torch.uint8) for p in parts], inp.view(torch.uint8), group=group)
        else:
            dist.all_gather(parts, inp, group=group)
    else:
        dist.all_gather_into_tensor(out, inp, group=group)


def reduce_scatter_avg(out: torch.Tensor, inp: torch.Tensor, group=None):
    from . import comm
    if out.is_cuda and comm.active_for(group):
        comm.reduce_scatter_avg(out, inp)
    elif dist.get_backend(group) == "gloo":
        tmp = inp.clone()
        dist.all_reduce(tmp, group=group)
        w, r = dist.get_world_size(group), dist.get_rank(group)
        out.copy_(tmp.view(w, -1)[r] / w)
    else:
        dist.reduce_scatter_tensor(out, inp, op=dist.ReduceOp.AVG, group=group)
