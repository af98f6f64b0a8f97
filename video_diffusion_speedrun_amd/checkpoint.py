"""Checkpoint save / load with the reference's state-dict contract.

The reference saves `get_model_state_dict(dit_model)` with `dcp.save` (train.py:553,581-584) and
resumes by converting that directory with `dcp_to_torch_save` and `load_state_dict(assign=True,
strict=False)` before the FSDP wrap (train.py:292-320); it stores no optimizer / step state.

    save_checkpoint(dir, model, optimizer, step)      # all ranks call it; rank 0 writes the weights
    step = load_checkpoint(dir, model, optimizer)     # before OR after apply_fsdp

Weights are written as full fp32 tensors under the reference's parameter names, both as a torch
file (`model.pt`) and -- `dcp=True` -- as a torch.distributed.checkpoint directory (`dcp/`) that
the reference's own loader reads; a checkpoint directory written by the reference loads here.
On top of the reference: each rank's optimizer shard (AdamW moments, step) goes to
`optim_rank{r}_of{W}.pt`, so a run resumes exactly when world size and sharding are unchanged.
"""
from __future__ import annotations

import os
from typing import Dict, Optional

import torch
import torch.distributed as dist


def _rank_world(model=None):
    """(rank, world) of the communicator the model is sharded over (apply_fsdp's process_group; the
    default group when none was given)."""
    if model is not None and getattr(model, "_world", 1) > 1:
        return model._rank, model._world
    if dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def _barrier(model=None):
    if dist.is_initialized():
        dist.barrier(group=getattr(model, "_pg", None))


def save_checkpoint(path: str, model, optimizer=None, step: int = 0, dcp: bool = False) -> None:
    rank, world = _rank_world(model)
    os.makedirs(path, exist_ok=True)
    state = model.full_state_dict()  # collective when sharded: every rank calls it
    if rank == 0:
        state = {k: v.detach().to("cpu", torch.float32).contiguous() for k, v in state.items()}
        torch.save({"model": state, "step": int(step)}, os.path.join(path, "model.pt"))
        if dcp:
            import torch.distributed.checkpoint as dcp_mod
            dcp_mod.save(state, checkpoint_id=os.path.join(path, "dcp"), no_dist=True)
    if optimizer is not None:
        torch.save({"optimizer": optimizer.state_dict(), "step": int(step), "rank": rank, "world": world},
                   os.path.join(path, f"optim_rank{rank}_of{world}.pt"))
    _barrier(model)


def _read(path: str, model=None):
    """(state dict of full fp32 tensors, stored step) from model.pt, a reference `temp.pt` (plain state
    dict) or a DCP directory (ours or the reference's).  Every file is read with weights_only=True: the
    stored objects are tensors, numbers, tuples, lists and dicts only."""
    if os.path.isfile(path):
        obj = torch.load(path, map_location="cpu", weights_only=True)
    elif os.path.isfile(os.path.join(path, "model.pt")):
        obj = torch.load(os.path.join(path, "model.pt"), map_location="cpu", weights_only=True)
    else:
        from torch.distributed.checkpoint.format_utils import dcp_to_torch_save
        d = os.path.join(path, "dcp") if os.path.isdir(os.path.join(path, "dcp")) else path
        tmp = os.path.join(path, "temp.pt")  # the reference's own conversion target (train.py:298-300)
        if _rank_world(model)[0] == 0 and not os.path.exists(tmp):
            dcp_to_torch_save(d, tmp)
        _barrier(model)
        obj = torch.load(tmp, map_location="cpu", weights_only=True)
    wrapped = isinstance(obj, dict) and isinstance(obj.get("model"), dict)
    state = obj["model"] if wrapped else obj
    step = int(obj.get("step", 0)) if wrapped else 0
    # strip the wrappers' prefixes like train.py:305-310
    return {k.replace("module.", "").replace("_orig_mod.", ""): v for k, v in state.items() if torch.is_tensor(v)}, step


def read_model_state(path: str) -> Dict[str, torch.Tensor]:
    return _read(path)[0]


def load_checkpoint(path: str, model, optimizer=None) -> int:
    """Loads the weights into `model` (plain or already sharded) and, when present for this
    rank / world size, the optimizer shard.  Returns the stored step (0 for reference checkpoints)."""
    rank, world = _rank_world(model)
    state, step = _read(path, model)
    own = dict(model.named_parameters())
    missing = [k for k in own if k not in state]
    if missing:
        raise KeyError(f"checkpoint lacks parameters: {missing[:5]}{'...' if len(missing) > 5 else ''}")
    groups = getattr(model, "_groups", None)
    with torch.no_grad():
        if groups is None:  # not materialised yet: plain tensors
            for k, p in own.items():
                p.copy_(state[k].to(p.dtype).reshape(p.shape))
        else:  # flat groups (sharded or not): write this rank's piece of every tensor into the fp32 master
            for g in groups:
                for n in g.names:
                    lo, hi = g.local_range(n)
                    if hi == lo:
                        continue
                    g0 = g.rank * g.shard + lo - g.offsets[n]
                    g.master[lo:hi].copy_(state[n].reshape(-1)[g0:g0 + (hi - lo)].to(g.master.device, torch.float32))
                g.invalidate_shadow()  # the master was written behind the parameters' version counters
    if optimizer is not None and os.path.isdir(path):
        f = os.path.join(path, f"optim_rank{rank}_of{world}.pt")
        if os.path.isfile(f):
            blob = torch.load(f, map_location="cpu", weights_only=True)
            optimizer.load_state_dict(blob["optimizer"])
            step = int(blob.get("step", step))
    return step
