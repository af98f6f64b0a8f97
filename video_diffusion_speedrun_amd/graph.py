"""Whole-step HIP-graph replay of the train step (the reference's analogue is
`torch.compile(dit_model)` behind `--compile_models`, train.py:326-328, which removes Python and
launch overhead with a tracing compiler).

MI355X-first: nothing is traced or re-generated.  The explicit kernel sequence of
`train.forward -> loss.backward -> MuAdamW.step` (about 60 launches per DiT block) is captured once
into a HIP graph and replayed with one launch per step.  That needs every per-step value to live
in device memory rather than in kernel arguments:

  * the batch            -> static input buffers, refreshed with stream-ordered copies;
  * the RoPE offsets     -> device int32[3] read by `vds_rope_rows_dev` (drawn on the host from the
                            global CPU RNG exactly like the eager path, model.py:224-226);
  * the AdamW bias corrections and LR multiplier -> device float[3] read by `vds_adamw_multi_dev`;
  * z / noise / caption-dropout draws -> torch's default device generator (graph-safe Philox: the
                            captured kernels take seed/offset from device memory).

The first `eager_steps` calls run the ordinary path (they are real train steps: they build the
optimizer's descriptor tables and the allocator pool the capture then reuses); the next call
captures and from then on every call is a replay.  Large configurations are device-bound and gain
little; launch-bound ones (C1-size models, small batches) gain the most -- DESIGN.md 7e.

Sharded models (fsdp.ShardRuntime; the reference compiles the FSDP-wrapped model too, train.py:323-329 and
run_debug.sh:12-25): the capture covers the runtime's whole choreography -- the communication stream forks from the
capturing stream (`comm.wait_stream(compute)`), the per-group bf16 all-gathers and fp32 reduce-scatters are RCCL calls
on that stream (RCCL collectives are capturable), the compute stream joins on the per-group events, and the final
`wait_stream` closes the fork before the capture ends.  Every rank captures and replays the same sequence.  The
exposed-communication measurement (timing events) is off inside a capture.  Proven on one GPU with the runtime
forced on over a 1-rank RCCL group (tests/test_model_gpu.py::test_graph_replay_with_the_sharding_runtime) and, since round
6, at world size 2 on emulated ranks -- two sharded replicas in one process with the real shard layout, streams, events,
separate gathered / reduced buffers and sharded optimizers, both resident and `reshard_after_forward` runtimes, only the
two collectives replaced by an in-process exchange (test_graph_replay_with_two_emulated_ranks).  What no box of this pool
could run is the capture of RCCL collectives across several PROCESSES; RCCL supports it, every rank captures and replays
the same sequence.

A captured graph bakes in device pointers (static batch buffers, the fp8 amax tables, the activation buffers the
allocator handed out): `step()` refuses a batch whose latent / context shape differs from the captured one.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from . import train as _train

bf16 = torch.bfloat16


class GraphedTrainStep:
    """`step(batch) -> loss` with the semantics of `train.train_step` (train.py:412-434)."""

    def __init__(self, dit_model, optimizer, lr_scheduler, device, eager_steps: int = 2):
        if eager_steps < 1:
            raise ValueError("at least one eager step is needed before the capture (descriptor tables, shadows)")
        self.model, self.opt, self.sched = dit_model, optimizer, lr_scheduler
        self.device = torch.device(device)
        self.eager_left = eager_steps
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self.static: Dict[str, torch.Tensor] = {}
        self.rope_dev = torch.zeros(3, dtype=torch.int32, device=self.device)
        self.loss: Optional[torch.Tensor] = None
        self.n_replays = 0
        optimizer.use_device_scalars(self.device)

    # ------------------------------------------------------------------------------------------
    def _thw(self, latent):
        m = self.model
        return (latent.shape[2] // m.time_patch_size, latent.shape[3] // m.patch_size, latent.shape[4] // m.patch_size)

    def _stage(self, batch):
        """copy the batch into the static buffers (allocating them on first use)"""
        lat, ctx = batch["latent"], batch["context"]
        if not self.static:
            self.static["latent"] = torch.empty(lat.shape, dtype=bf16, device=self.device)
            self.static["context"] = torch.empty(ctx.shape, dtype=bf16, device=self.device)
        for k, src in (("latent", lat), ("context", ctx)):
            dst = self.static[k]
            if tuple(src.shape) != tuple(dst.shape):
                raise ValueError(f"GraphedTrainStep was captured for {k} {tuple(dst.shape)}, got {tuple(src.shape)}")
            dst.copy_(src, non_blocking=True)

    def _draw_rope(self):
        st = self.model.rope.draw_start(self._thw(self.static["latent"]))
        self.rope_dev.copy_(torch.tensor(st, dtype=torch.int32))
        return st

    def _body(self, rope_start):
        loss, _ = _train.forward(self.model, self.static, None, None, self.device, 0, False, rope_start=rope_start)
        self.opt.zero_grad()
        loss.backward()
        self.opt.step()
        return loss

    def _capture(self):
        step0 = self.opt._step
        fs = getattr(self.model, "_fsdp", None)
        measure = fs.measure if fs is not None else False
        if fs is not None:
            fs.measure = False  # timing events cannot be recorded into a graph
            counts = (fs.n_all_gather, fs.n_reduce_scatter)
        g = torch.cuda.CUDAGraph()
        torch.cuda.synchronize()
        try:
            with torch.cuda.graph(g):
                loss = self._body(self.rope_dev)
        finally:
            if fs is not None:
                fs.measure = measure
                # what one step issues (the memory-bounded runtime gathers most blocks twice): replays count it
                self._per_step = (fs.n_all_gather - counts[0], fs.n_reduce_scatter - counts[1])
                fs.n_all_gather, fs.n_reduce_scatter = counts  # the capture launched nothing
        self.opt._step = step0  # the capture launched nothing: `advance()` counts the step at replay time
        self.graph, self.loss = g, loss

    # ------------------------------------------------------------------------------------------
    def step(self, batch) -> torch.Tensor:
        if "context" not in batch:
            raise ValueError("GraphedTrainStep takes pre-encoded batches: {'latent', 'context'}")
        self._stage(batch)
        hist = getattr(self.model, "_fp8_hist", None) if getattr(self.model, "fp8", False) else None
        fp8_unarmed = getattr(self.model, "fp8", False) and (hist is None or not hist.ready or hist.part_tab is None)
        if self.eager_left > 0 or (self.graph is None and fp8_unarmed):
            # fp8: the capture must see delayed scaling armed (a complete recorded step) and the partial-maxima table
            # allocated, or the captured amax roll would differ from the one every later step needs
            self.eager_left = max(0, self.eager_left - 1)
            loss = self._body(self._draw_rope()).detach().clone()
        else:
            if self.graph is None:
                self._capture()
            self._draw_rope()
            self.opt.advance()
            # the captured step holds no fp32 -> bf16 cast (the optimizer kernel writes both copies): parameters
            # written from outside since the last step (load_state_dict, EMA swap, p.mul_()) are re-cast here
            from . import ops
            for g in self.model._groups:
                g.refresh_shadow(ops.cast_f32_bf16)
            fs = getattr(self.model, "_fsdp", None)
            if fs is not None:  # the counters a capture cannot bump
                fs.n_all_gather += self._per_step[0]
                fs.n_reduce_scatter += self._per_step[1]
            self.graph.replay()
            self.n_replays += 1
            loss = self.loss.detach()
        if self.sched is not None:
            self.sched.step()
        return loss
