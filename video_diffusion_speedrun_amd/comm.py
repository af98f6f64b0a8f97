"""Host side of the sharding runtime's communicator: `vds_comm_*` of include/vds.h (csrc/comm.hip), i.e. RCCL
driven from the kernel library itself, one communicator per process, collectives asynchronous on the caller's HIP
stream.  Replaces what the reference gets from FSDP2's process-group plumbing (model.py:468-542).

torch.distributed is used for ONE thing here: shipping rank 0's 128-byte RCCL unique id to the other ranks of the
group at start-up (it is the rendezvous the launcher -- torchrun, train.py:214-220 -- has already set up).  After that
no data-path collective goes through torch.distributed on the GPU: `params.all_gather_flat` / `reduce_scatter_avg`
call `vds_all_gather_bf16` / `vds_reduce_scatter_f32_avg`.  (CPU tensors -- the gloo tests of the host logic -- and
`VDS_COMM=torch` keep the torch.distributed path.)
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch
import torch.distributed as dist

from . import _lib

ID_BYTES = 128
_state = {"group": None, "active": False, "rank": 0, "world": 1, "fallback": None}


def enabled() -> bool:
    return os.environ.get("VDS_COMM", "vds") != "torch"


def active_for(group) -> bool:
    """True when the library's communicator spans exactly this torch process group (None = the world)."""
    return _state["active"] and _state["group"] is group


def ensure(group=None) -> bool:
    """Create the library's RCCL communicator over the ranks of `group` (collective: every rank of the group calls
    it).  Returns False when the torch path was requested (`VDS_COMM=torch`) -- or, LOUDLY (stderr, every rank), when
    the library's communicator could not be created on some rank: the collectives then stay on torch.distributed's
    RCCL communicator over the same GPUs and links (same transport, same kernels; never a CPU path)."""
    if not enabled():
        return False
    if active_for(group):
        return True
    if _state["active"]:
        raise RuntimeError("vds communicator already spans another process group; comm.destroy() it first")
    lib = _lib.load()
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"

    def agree(ok: bool) -> bool:
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        return int(flag.item()) == 1

    def fall_back(err):
        import sys
        _state["fallback"] = err or "another rank failed"
        print(f"[vds comm] rank {rank}: the library's RCCL communicator is NOT in use ({_state['fallback']}); "
              "collectives run on torch.distributed's RCCL communicator instead", file=sys.stderr, flush=True)
        return False

    # phase 1 -- local, no communication inside the library: RCCL binds (dlopen + symbols) and rank 0 draws the id.
    # The ranks agree on it BEFORE anyone enters ncclCommInitRank (itself a collective: a rank that failed earlier
    # would otherwise leave the others blocked inside it).
    err = None
    buf = (C.c_ubyte * ID_BYTES)()
    rc = lib.vds_comm_available()
    if rc != 0:
        err = f"vds_comm_available -> {rc} {lib.vds_last_error().decode()}"
    elif rank == 0:
        rc = lib.vds_comm_unique_id(buf, ID_BYTES)
        if rc != 0:
            err = f"vds_comm_unique_id -> {rc} {lib.vds_last_error().decode()}"
    if not agree(err is None):
        return fall_back(err)
    box = [bytes(buf)]
    src = dist.get_global_rank(group, 0) if group is not None else 0
    dist.broadcast_object_list(box, src=src, group=group)
    # phase 2 -- the collective init, entered by every rank; then agree on its outcome
    ident = (C.c_ubyte * ID_BYTES).from_buffer_copy(box[0])
    rc = lib.vds_comm_init(rank, world, ident, ID_BYTES)
    ok = rc == 0
    if not ok:
        err = f"vds_comm_init(rank={rank}, world={world}) -> {rc} {lib.vds_last_error().decode()}"
    if not agree(ok):
        if ok:
            lib.vds_comm_destroy()
        return fall_back(err)
    _state["fallback"] = None
    _state.update(group=group, active=True, rank=rank, world=world)
    return True


def fallback_reason() -> Optional[str]:
    """why `ensure` last fell back to torch.distributed's communicator (None: it did not).  bench.py exits non-zero
    on it: a scaling number must not silently come from another communicator than the one it names."""
    return _state["fallback"]


def info() -> dict:
    r, w, v, ap = C.c_int32(-1), C.c_int32(0), C.c_int32(0), C.c_int32(0)
    _lib.load().vds_comm_info(C.byref(r), C.byref(w), C.byref(v), C.byref(ap))
    return {"rank": r.value, "world": w.value, "rccl_version": v.value,
            "schedule": "allpairs" if ap.value else "rccl", "active": _state["active"]}


def destroy():
    if _state["active"]:
        torch.cuda.synchronize()
        _lib.check(_lib.load().vds_comm_destroy(), "vds_comm_destroy")
        _state.update(group=None, active=False, rank=0, world=1)


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)  # private: resolved once, public fallback below


def _stream() -> int:
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def all_gather(out: torch.Tensor, inp: torch.Tensor):
    """out[world * n] <- every rank's inp[n]; bf16 (compute copies) or fp32 (masters, for checkpoints)"""
    assert out.is_cuda and inp.is_cuda and out.is_contiguous() and inp.is_contiguous()
    assert out.numel() == _state["world"] * inp.numel() and out.dtype == inp.dtype
    lib = _lib.load()
    if inp.dtype == torch.bfloat16:
        _lib.check(lib.vds_all_gather_bf16(inp.data_ptr(), out.data_ptr(), inp.numel(), _stream()), "vds_all_gather_bf16")
    elif inp.dtype == torch.float32:
        _lib.check(lib.vds_all_gather_f32(inp.data_ptr(), out.data_ptr(), inp.numel(), _stream()), "vds_all_gather_f32")
    else:
        raise TypeError(f"vds all-gather: bf16 or fp32 only, got {inp.dtype}")


def reduce_scatter_avg(out: torch.Tensor, inp: torch.Tensor):
    """out[n] <- mean over ranks of inp[rank*n : (rank+1)*n], fp32"""
    assert out.is_cuda and inp.is_cuda and out.dtype == inp.dtype == torch.float32
    assert inp.numel() == _state["world"] * out.numel() and out.is_contiguous() and inp.is_contiguous()
    lib = _lib.load()
    ws_bytes = lib.vds_reduce_scatter_workspace_bytes(out.numel())
    ws: Optional[torch.Tensor] = None
    if ws_bytes:
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=out.device)  # on the current (communication) stream
    _lib.check(lib.vds_reduce_scatter_f32_avg(inp.data_ptr(), out.data_ptr(), out.numel(),
                                              ws.data_ptr() if ws is not None else None, ws_bytes, _stream()),
               "vds_reduce_scatter_f32_avg")


def all_reduce_avg_(buf: torch.Tensor):
    assert buf.is_cuda and buf.dtype == torch.float32 and buf.is_contiguous()
    _lib.check(_lib.load().vds_all_reduce_f32_avg(buf.data_ptr(), buf.numel(), _stream()), "vds_all_reduce_f32_avg")
    return buf
