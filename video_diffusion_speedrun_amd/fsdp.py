"""Parameter / gradient sharding over the GPUs of one node: the build's `apply_fsdp`
(reference: model.py:468-542, FSDP2 `fully_shard` per DiTBlock + root, bf16 all-gather,
fp32 reduce-scatter-average; call site train.py:323-325).

MI355X-first re-design of the same contract:

  * one flat buffer per shard group (params.FlatGroup) => ONE RCCL all-gather (bf16) and ONE
    reduce-scatter (fp32, avg) per group per step: 29 + 29 large contiguous collectives for
    DiT-XL instead of per-parameter copy-in/copy-out.  xGMI is point-to-point; few, large
    messages are what it wants.  The collectives are the kernel library's own (`vds_all_gather_bf16`,
    `vds_reduce_scatter_f32_avg`: csrc/comm.hip drives RCCL on a communicator created from a unique id that
    torch.distributed ships once at start-up; comm.py);
  * 288 GB of HBM per GPU: by default the gathered bf16 copy of EVERY group (2.3 GB for DiT-XL) stays
    resident from forward to backward, so the reference's backward re-all-gather
    (`reshard_after_forward`, model.py:525) is not needed at all -- all-gather traffic is halved;
    `apply_fsdp(..., reshard_after_forward=True)` is the reference's memory-bounded behaviour (ReshardRuntime: the
    blocks' copies live in a ring of prefetch + 2 buffers and are gathered again in backward);
  * the gathers of a step are issued in use order on a dedicated communication stream; the compute stream
    waits on the per-group event right before the first kernel that reads the group.  By default ALL of them
    are issued up-front (prefetch depth = everything: the links are busy for the first 11-78 ms of the step and
    idle afterwards); `VDS_AG_PREFETCH=d` (or `ShardRuntime.prefetch = d`) bounds the window to d groups beyond
    the one in use, i.e. group i + d is issued when block i starts -- FSDP2's behaviour is d = 1 -- so that the
    first multi-GPU run can A/B the burst against a spread (RCCL's kernels take workgroup slots while they run);
  * each group's reduce-scatter is issued on the communication stream as soon as that group's
    last weight-gradient kernel is queued, overlapping with the backward of the next block;
  * world_size == 1 works (no communication, no streams) -- the reference cannot
    (model.py:489).

On CPU tensors (gloo; tests only) the same choreography runs synchronously without streams.
"""
from __future__ import annotations

from typing import Optional

import os

import torch
import torch.distributed as dist

from .params import FlatGroup


def get_device_mesh(world_size: Optional[int] = None):
    """Shape of the reference's mesh (model.py:475-498): dp_replicate 1 x dp_shard W, tp 1.
    There is no DeviceMesh object here: one process group spans the shard dimension."""
    w = world_size if world_size is not None else (dist.get_world_size() if dist.is_initialized() else 1)
    return {"dp_replicate": 1, "dp_shard": w, "tp": 1}


def reference_local_shape(shape, world: int, rank: int):
    """Shape of the piece of a parameter that rank `rank` holds under the REFERENCE's wrap (FSDP2 `fully_shard`,
    model.py:523-541: every parameter a `Shard(0)` DTensor, dim 0 cut into `world` chunks of ceil(dim0 / world)
    rows, trailing ranks possibly empty).  This build shards differently on purpose -- contiguous chunks of one
    flat buffer per group (params.FlatGroup) -- so this is only needed where the two layouts meet: reading or
    writing `torch.distributed.checkpoint` shards, and the parity test against the reference's recorded shapes
    (tests/golden/g6_fsdp.pt)."""
    shape = tuple(int(s) for s in shape)
    if not shape:
        return shape
    chunk = -(-shape[0] // world)
    rows = max(0, min(shape[0], (rank + 1) * chunk) - rank * chunk)
    return (rows,) + shape[1:]


class ShardRuntime:
    """Stream / event choreography of the per-group collectives around DiT's explicit
    forward / backward kernel sequences (model.py hooks `pre_forward_*` ... `post_backward_*`)."""

    def __init__(self, model, cast_fn, process_group=None):
        self.model = model
        self.cast_fn = cast_fn
        self.pg = process_group
        self.cuda = model._groups[0].device.type == "cuda"
        # default priority on purpose: a high-priority stream (FSDP2's choice; VDS_COMM_PRIORITY=-1 for A/B) cost 8 % of
        # the DiT-XL step at world size 1 -- RCCL's polling kernels then take workgroup slots from the attention / GEMM
        # kernels the moment they are enqueued
        prio = int(os.environ.get("VDS_COMM_PRIORITY", "0"))
        self.comm = torch.cuda.Stream(device=model._groups[0].device, priority=prio) if self.cuda else None
        self.gather_ev = [None] * len(model._groups)
        self.world = model._world
        # all-gather window: 0 = issue every group's gather up-front; d > 0 = at most d groups ahead of the one in use
        self.prefetch = max(0, int(os.environ.get("VDS_AG_PREFETCH", "0")))
        self._next_gather = len(model._groups)  # first group whose gather has not been issued in this forward
        self.n_all_gather = 0
        self.n_reduce_scatter = 0
        # measure=True: bracket every point where the compute stream waits for the communication stream with an
        # event pair ON THE COMPUTE STREAM; the time between the two is the stall, i.e. communication that was not
        # hidden under compute (bench.py's exposed-comm figure).  Off by default (no events).
        self.measure = False
        self._stalls = []

    def _stall_begin(self):
        if self.measure and self.cuda:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            return ev
        return None

    def _stall_end(self, ev0):
        if ev0 is not None:
            ev1 = torch.cuda.Event(enable_timing=True)
            ev1.record()
            self._stalls.append((ev0, ev1))

    def exposed_comm_ms(self) -> float:
        """sum of the compute stream's stalls on the communication stream since the last call (synchronises)"""
        if not self._stalls:
            return 0.0
        torch.cuda.synchronize()
        ms = sum(a.elapsed_time(b) for a, b in self._stalls)
        self._stalls = []
        return ms

    # ---- helpers --------------------------------------------------------------------------
    def _on_comm(self):
        return torch.cuda.stream(self.comm) if self.cuda else _Null()

    def _comm_waits_compute(self):
        if self.cuda:
            self.comm.wait_stream(torch.cuda.current_stream())

    def _compute_waits(self, gi: int):
        ev = self.gather_ev[gi]
        if ev is not None:
            t0 = self._stall_begin()
            torch.cuda.current_stream().wait_event(ev)
            self._stall_end(t0)
            self.gather_ev[gi] = None

    def _reduce(self, gi: int):
        g: FlatGroup = self.model._groups[gi]
        self._comm_waits_compute()  # every gradient kernel of this group is queued before here
        with self._on_comm():
            g.reduce_grads(self.pg)
        self.n_reduce_scatter += 1

    # ---- forward --------------------------------------------------------------------------
    def _issue_gathers(self, upto: int):
        """bf16 cast + all-gather of the groups [_next_gather, upto) on the comm stream, one event per group.  A
        gather reads the group's bf16 shadow (written by the optimizer step, which the comm stream has waited for in
        pre_forward_root) and writes its gathered copy, whose last readers -- the previous step's backward kernels --
        are older than that optimizer step: no further wait on the compute stream is needed"""
        upto = min(upto, len(self.model._groups))
        if self._next_gather >= upto:
            return
        with self._on_comm():
            for gi in range(self._next_gather, upto):
                self.model._groups[gi].gather(self.cast_fn, self.pg)
                self.n_all_gather += 1
                if self.cuda:
                    ev = torch.cuda.Event()
                    ev.record(self.comm)
                    self.gather_ev[gi] = ev
        self._next_gather = upto

    def pre_forward_root(self):
        """issue the bf16 cast + all-gather of the first groups (all of them unless `prefetch` bounds the window), in
        use order, on the comm stream"""
        self._comm_waits_compute()  # the optimizer step that wrote master / shadow is done
        self._next_gather = 0
        n = len(self.model._groups)
        self._issue_gathers(n if self.prefetch == 0 else 1 + self.prefetch)  # root + the first `prefetch` blocks
        self._compute_waits(0)

    def pre_forward_block(self, i: int):
        if self.prefetch:
            self._issue_gathers(1 + i + 1 + self.prefetch)  # keep `prefetch` groups beyond block i in flight
        self._compute_waits(1 + i)

    def post_forward_block(self, i: int):
        pass  # gathered copies stay resident (288 GB HBM): nothing to reshard

    def post_forward_root(self):
        pass

    # ---- backward -------------------------------------------------------------------------
    def pre_backward_root(self):
        for gi in range(len(self.model._groups)):  # no-grad forwards may have left events unconsumed
            self._compute_waits(gi)

    def pre_backward_block(self, i: int):
        pass  # no re-gather: see the module docstring

    def post_backward_block(self, i: int):
        self._reduce(1 + i)

    def post_backward_root(self):
        self._reduce(0)
        if self.cuda:
            t0 = self._stall_begin()
            torch.cuda.current_stream().wait_stream(self.comm)
            self._stall_end(t0)
        for g in self.model._groups:
            g.publish_grads()


class ReshardRuntime(ShardRuntime):
    """`reshard_after_forward` of the reference's wrap (model.py:525,541: every block but the last frees its gathered
    parameters after its forward and all-gathers them again in backward): the memory-bounded mode.  The gathered bf16
    copies of the BLOCKS live in a ring of `prefetch + 2` buffers of the largest block's size instead of one resident
    copy per block (DiT-XL: 3 x 80 MB instead of 2.3 GB; what pays for it is depth - 1 more all-gathers per step):

      forward   block i's copy is gathered into a free buffer at most `prefetch` blocks ahead (default 1 = FSDP2) and
                released when block i's forward kernels are queued -- the last block's stays, like the reference's;
      backward  block i is gathered again (`prefetch` blocks ahead, downwards), released behind its reduce-scatter.

    A buffer is handed to the next gather with an event recorded on the compute stream at release: the communication
    stream waits for it, so a gather never overwrites weights a queued kernel still reads.  The root group (patch
    embedding, time MLP, final layer: used at both ends of both passes) stays resident."""

    def __init__(self, model, cast_fn, process_group=None):
        super().__init__(model, cast_fn, process_group)
        groups = model._groups
        self.ahead = max(1, self.prefetch)
        self.n_slots = min(self.ahead + 2, len(groups) - 1)
        size = max(g.padded for g in groups[1:])
        dev = groups[0].device
        self.slots = [torch.zeros(size, dtype=torch.bfloat16, device=dev) for _ in range(self.n_slots)]
        self.slot_free_ev = [None] * self.n_slots
        self.free_slots = list(range(self.n_slots))
        self.slot_of = [None] * len(groups)
        for g in groups[1:]:  # the per-group gathered copies apply_fsdp allocated are not needed in this mode
            g.full = None
            g.gathered = False

    def _gather_group(self, gi: int):
        g: FlatGroup = self.model._groups[gi]
        if gi == 0:
            with self._on_comm():
                g.gather(self.cast_fn, self.pg)
                if self.cuda:
                    ev = torch.cuda.Event()
                    ev.record(self.comm)
                    self.gather_ev[0] = ev
            self.n_all_gather += 1
            return
        if self.slot_of[gi] is not None:
            return  # gathered (or in flight) already
        if not self.free_slots:
            raise RuntimeError("ReshardRuntime: no free parameter buffer (a block was not released)")
        sl = self.free_slots.pop(0)
        self.slot_of[gi] = sl
        g.full = self.slots[sl][:g.padded]
        with self._on_comm():
            if self.cuda and self.slot_free_ev[sl] is not None:
                self.comm.wait_event(self.slot_free_ev[sl])  # the kernels that read the previous tenant are done
                self.slot_free_ev[sl] = None
            g.gather(self.cast_fn, self.pg)
            if self.cuda:
                ev = torch.cuda.Event()
                ev.record(self.comm)
                self.gather_ev[gi] = ev
        self.n_all_gather += 1

    def _release_group(self, gi: int):
        sl = self.slot_of[gi]
        if sl is None:
            return
        self._compute_waits(gi)  # (a gather nobody consumed: order the release behind it)
        if self.cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self.slot_free_ev[sl] = ev
        self.slot_of[gi] = None
        self.free_slots.append(sl)
        g = self.model._groups[gi]
        g.full = None
        g.gathered = False

    # ---- forward --------------------------------------------------------------------------
    def pre_forward_root(self):
        self._comm_waits_compute()  # the optimizer step that wrote master / shadow is done
        n = len(self.model._groups)
        for gi in range(1, n):  # a forward without a backward (sampling, evaluation) left its last block gathered
            self._release_group(gi)
        self._gather_group(0)
        for gi in range(1, min(n, 1 + self.ahead)):
            self._gather_group(gi)
        self._compute_waits(0)

    def pre_forward_block(self, i: int):
        n = len(self.model._groups)
        for gi in range(1 + i, min(n, 1 + i + 1 + self.ahead)):
            self._gather_group(gi)
        self._compute_waits(1 + i)

    def post_forward_block(self, i: int):
        if i < len(self.model._groups) - 2:  # (model.py:525: the last block is not resharded after forward)
            self._release_group(1 + i)

    # ---- backward -------------------------------------------------------------------------
    def pre_backward_root(self):
        for gi in range(len(self.model._groups)):
            if gi == 0 or self.slot_of[gi] is not None:
                self._compute_waits(gi)

    def pre_backward_block(self, i: int):
        for gi in range(1 + i, max(0, 1 + i - 1 - self.ahead), -1):
            self._gather_group(gi)
        self._compute_waits(1 + i)

    def post_backward_block(self, i: int):
        self._reduce(1 + i)
        self._release_group(1 + i)


class _Null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def apply_fsdp(dit_model, param_dtype=torch.bfloat16, reduce_dtype=torch.float32, process_group=None,
               device=None, cast_fn=None, force_runtime=False, world_rank=None, reshard_after_forward=None):
    """Shard `dit_model` over the ranks of `process_group` (default: the world) and return it
    -- the same object, still callable, still exposing get_mup_setup / named_parameters
    (model.py:512-542).  After this call every nn.Parameter is this rank's 1-D fp32 piece of
    its tensor (possibly empty), which is what the element-wise optimizer consumes.

    param_dtype / reduce_dtype: the reference passes bf16 / fp32 (train.py:323-325); those
    are the only values the kernels implement.

    reshard_after_forward: False (default; also VDS_FSDP_RESHARD unset / 0) keeps every group's gathered bf16 copy
    resident from forward to backward (2.3 GB for DiT-XL: no re-gather in backward); True (VDS_FSDP_RESHARD=1) is the
    reference's memory-bounded behaviour (model.py:525,541) -- see ReshardRuntime."""
    if param_dtype != torch.bfloat16 or reduce_dtype != torch.float32:
        raise ValueError("apply_fsdp: the HIP path implements param_dtype=bf16, reduce_dtype=fp32 only")
    if world_rank is not None:
        # explicit (world, rank) without a torch.distributed group: the caller supplies the collectives by
        # replacing params.all_gather_flat / params.reduce_scatter_avg (single-GPU emulation of W ranks in
        # tests/test_model_gpu.py::test_two_emulated_ranks_on_one_gpu)
        world, rank = world_rank
    elif dist.is_initialized():
        world, rank = dist.get_world_size(process_group), dist.get_rank(process_group)
    else:
        world, rank = 1, 0
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
    device = torch.device(device)
    if dit_model._fsdp is not None:
        raise RuntimeError("apply_fsdp was already applied to this model")
    if cast_fn is None:
        if device.type == "cuda":
            from . import ops
            cast_fn = ops.cast_f32_bf16
        else:  # gloo / CPU tests of the host logic only (the model itself never computes on CPU)
            cast_fn = lambda src, dst: dst.copy_(src)
    full_values = {n: p.data.detach().clone() for n, p in dit_model.named_parameters()}
    dit_model._world, dit_model._rank, dit_model._pg = world, rank, process_group
    root, blocks = dit_model._group_members()
    pg = process_group if world_rank is None else None
    groups = [FlatGroup("root", root, world, rank, pg)]
    groups += [FlatGroup(f"blocks.{i}", m, world, rank, pg) for i, m in enumerate(blocks)]
    run = world > 1 or (force_runtime and (dist.is_initialized() or world_rank is not None))
    from .model import grad_arena
    dit_model._grad_arena = grad_arena(groups, device)
    off = 0
    for g in groups:  # (W=1 + runtime: real collectives into separate buffers)
        g.materialize(device, full_values, separate=run, gfull=dit_model._grad_arena[off:off + g.padded])
        off += g.padded
    for name, buf in dit_model.named_buffers():
        buf.data = buf.data.to(device)
    dit_model._groups = groups
    # force_runtime: run the stream / event / collective choreography even at world_size 1 (a 1-rank
    # process group must be initialised) -- used to test it on a single GPU
    if run:
        if device.type == "cuda" and world_rank is None and dist.is_initialized():
            from . import comm
            comm.ensure(process_group)  # the library's own RCCL communicator (vds_comm_*); collective call
        if reshard_after_forward is None:
            reshard_after_forward = os.environ.get("VDS_FSDP_RESHARD", "0") == "1"
        rt = ReshardRuntime if (reshard_after_forward and len(groups) > 2) else ShardRuntime
        dit_model._fsdp = rt(dit_model, cast_fn, process_group)
    return dit_model
