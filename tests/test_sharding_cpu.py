"""Host logic of the sharding runtime (params.FlatGroup + fsdp.ShardRuntime) on CPU:
layout arithmetic in one process, and a world_size-2 gloo run in which each rank computes
the gradients of its own half batch with the CPU oracle from the all-gathered bf16 weights,
reduce-scatters them, and must end up with exactly its slice of the single-process
batch-mean gradient and of the single-process AdamW update (reference contract:
model.py:512-542 -- bf16 all-gather, fp32 reduce-scatter-avg, optimizer on local shards)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import dit_oracle as O
from video_diffusion_speedrun_amd.params import ALIGN, FlatGroup

CFG = dict(in_channels=16, patch_size=2, time_patch_size=2, hidden_size=128, depth=2, num_heads=2,
           cross_attn_input_size=64, residual_v=True, train_bias_and_rms=False)
CONSTS = ["patch_proj", "context_kv", "positional_embedding"]


def make_model(**over):
    from video_diffusion_speedrun_amd.model import DiT
    CFG = dict(globals()["CFG"], **over)
    cfg = O.DiTConfig(**CFG)
    P = O.init_params(cfg, seed=7, randomize_zero_init=True, init_std_factor=1.0)
    m = DiT(**CFG)
    m.load_state_dict(P, strict=True)
    return m, cfg, P


@pytest.mark.parametrize("world", [3, 4, 8])
def test_flat_group_layout_covers_every_element_once(world):
    m, cfg, P = make_model()
    named = [(n, p) for n, p in m.named_parameters() if n.startswith("blocks.1.")]
    groups = [FlatGroup("blocks.1", named, world, r) for r in range(world)]
    g0 = groups[0]
    assert g0.padded % (world * 256) == 0 and g0.shard * world == g0.padded
    for n in g0.names:
        assert g0.offsets[n] % ALIGN == 0
        numel = P[n].numel()
        covered = 0
        for r, g in enumerate(groups):
            lo, hi = g.local_range(n)
            covered += hi - lo
            assert 0 <= lo <= hi <= g.shard
        assert covered == numel, n
    # lambda_param (1 element) lives on exactly one rank, the others hold an empty piece
    owners = [r for r, g in enumerate(groups) if g.local_range("blocks.1.lambda_param") != (0, 0)]
    assert len(owners) == 1


def test_world1_materialize_aliases_parameters():
    m, cfg, P = make_model()
    named = [(n, p) for n, p in m.named_parameters() if not n.startswith("blocks.")]
    g = FlatGroup("root", named, 1, 0)
    g.materialize("cpu")
    assert g.is_current()
    for n, p in named:
        assert p.shape == P[n].shape and torch.equal(p.data, P[n])
        assert p.data.data_ptr() == g.master.data_ptr() + 4 * g.offsets[n]
    g.gather(lambda src, dst: dst.copy_(src))
    assert torch.equal(g.w("register_tokens"), P["register_tokens"].to(torch.bfloat16))
    g.g("register_tokens").fill_(2.0)
    g.publish_grads()
    assert m.register_tokens.grad.shape == P["register_tokens"].shape and float(m.register_tokens.grad.mean()) == 2.0


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _oracle_grads(cfg, weights, batch):
    Pg = {k: w.clone().float().requires_grad_(True) for k, w in weights.items()}
    loss = O.train_forward(Pg, cfg, batch["latent"], batch["context"], batch["z"], batch["noise"], (1, 2, 3),
                           compute_dtype=torch.float32)
    loss.backward()
    return {k: (w.grad if w.grad is not None else torch.zeros_like(w)) for k, w in Pg.items()}, loss.item()


def _batches(world=2):
    """a global batch of max(4, world) samples and its per-rank slices (equal sizes: the mean over ranks of the
    per-rank batch-mean gradients is the batch-mean gradient of the whole batch)"""
    n = max(4, world)
    g = torch.Generator().manual_seed(3)
    full = dict(latent=torch.randn(n, 16, 4, 8, 8, generator=g), context=torch.randn(n, 6, 64, generator=g),
                z=torch.randn(n, generator=g), noise=torch.randn(n, 16, 4, 8, 8, generator=g))
    per = n // world
    parts = [{k: v[per * r:per * (r + 1)] for k, v in full.items()} for r in range(world)]
    return full, parts


def _worker(rank, world, port, q, prefetch=0):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        torch.set_num_threads(1 if world > 2 else 2)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from video_diffusion_speedrun_amd.fsdp import apply_fsdp, get_device_mesh
        m, cfg, P = make_model()
        assert get_device_mesh() == {"dp_replicate": 1, "dp_shard": world, "tp": 1}
        m = apply_fsdp(m, torch.bfloat16, torch.float32, device="cpu")
        fs = m._fsdp
        assert fs is not None and m._world == world and fs.world == world
        fs.prefetch = prefetch  # all-gather window: 0 = everything up-front, d = d groups beyond the one in use
        # residual-V lambdas: 1-element tensors live on exactly one rank (FSDP2 leaves empty shards elsewhere)
        own = torch.tensor([float(m.block_group(1).local_range("blocks.1.lambda_param") != (0, 0))])
        dist.all_reduce(own)
        assert own.item() == 1.0
        # every parameter is now this rank's 1-D piece; the pieces of the two ranks tile the tensor
        for n, p in m.named_parameters():
            assert p.dim() == 1 and p.dtype == torch.float32
        groups_tbl, settings = m.get_mup_setup(1e-3, 0.1, CONSTS)  # still works on a sharded model
        ref_tbl = O.mup_settings(O.param_shapes(cfg), 1e-3, 0.1, CONSTS)
        assert {k: (v["lr"], v["wd"]) for k, v in settings.items()} == {k: (v["lr"], v["wd"]) for k, v in ref_tbl.items()}

        full, halves = _batches(world)
        # ---- forward side: groups gathered in use order (window = prefetch); read the full bf16 weights ----
        fs.pre_forward_root()
        assert fs.n_all_gather == (1 + cfg.depth if prefetch == 0 else min(1 + cfg.depth, 1 + prefetch))
        weights = {}
        for gi, g in enumerate(m._groups):
            if gi > 0:
                fs.pre_forward_block(gi - 1)
                if prefetch:
                    assert fs.n_all_gather == min(1 + cfg.depth, gi + 1 + prefetch)
            for n in g.names:
                weights[n] = g.w(n).clone()
                assert torch.equal(weights[n], P[n].to(torch.bfloat16)), n  # bf16 all-gather is exact
        grads, _ = _oracle_grads(cfg, weights, halves[rank])
        # ---- backward side: write this rank's gradients, reduce-scatter group by group ----------
        for g in m._groups:
            g.gfull.zero_()
        fs.pre_backward_root()
        for i in reversed(range(cfg.depth)):
            fs.pre_backward_block(i)
            g = m.block_group(i)
            for n in g.names:
                g.g(n).copy_(grads[n])
            fs.post_backward_block(i)
        for n in m.root_group.names:
            m.root_group.g(n).copy_(grads[n])
        fs.post_backward_root()
        assert fs.n_all_gather == 1 + cfg.depth and fs.n_reduce_scatter == 1 + cfg.depth

        # ---- single-process truth: the batch-mean gradient of the whole batch (= FSDP's AVG over ranks of the
        # per-rank batch means; at world 2 additionally formed from the two halves explicitly) ---------
        per_rank = [_oracle_grads(cfg, weights, h)[0] for h in halves]
        gfull = {n: sum(gr[n] for gr in per_rank) / world for n in per_rank[0]}
        table = ref_tbl
        for g in m._groups:
            for n in g.names:
                p = g.params[n]
                lo, hi = g.local_range(n)
                mean = gfull[n].reshape(-1)
                g0 = g.rank * g.shard + lo - g.offsets[n]
                ref = mean[g0:g0 + (hi - lo)]
                assert p.grad is not None and p.grad.shape == ref.shape, n
                assert torch.allclose(p.grad, ref, rtol=1e-4, atol=1e-7), n
                # optimizer on the local shard == the slice of the full-tensor update
                if hi > lo:
                    full_p = P[n].clone().reshape(-1)
                    mm, vv = torch.zeros_like(full_p), torch.zeros_like(full_p)
                    O.adamw_step(full_p, mean.clone(), mm, vv, 1, table[n]["lr"], table[n]["wd"])
                    loc = p.data.clone()
                    O.adamw_step(loc, p.grad.clone(), torch.zeros_like(loc), torch.zeros_like(loc), 1, table[n]["lr"],
                                 table[n]["wd"])
                    # (elements whose gradient is ~0 are left out: AdamW's first step is lr * g / (|g| + eps), and the
                    # ring's summation order may differ from the in-order mean in the last fp32 bit)
                    keep = ref.abs() > 1e-4 * mean.abs().max()
                    assert torch.allclose(loc[keep], full_p[g0:g0 + (hi - lo)][keep], rtol=1e-6, atol=1e-8), n
        # full_tensor() re-assembles the fp32 master across ranks
        assert torch.equal(m._groups[1].full_tensor("blocks.0.qkv.weight"), P["blocks.0.qkv.weight"])
        sd = m.full_state_dict()
        assert set(sd) == set(P) and all(torch.equal(sd[k], P[k]) for k in P)
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception as e:  # surface the failure in the parent
        import traceback
        q.put((rank, traceback.format_exc()))


def _g6_worker(rank, world, port, q, golden):
    """this build's 2-rank sharded step (gloo; the oracle in the reference's bf16 compute mode stands in for the HIP
    kernels) on the inputs of tests/golden/g6_fsdp.pt, against what the REFERENCE's own `apply_fsdp` run produced"""
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        torch.set_num_threads(2)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from video_diffusion_speedrun_amd.fsdp import apply_fsdp
        from video_diffusion_speedrun_amd.model import DiT
        fx = torch.load(os.path.join(golden, "g6_fsdp.pt"), weights_only=False)
        cfg = O.DiTConfig(**fx["cfg"])
        P = O.init_params(cfg, seed=fx["param_seed"], randomize_zero_init=True, init_std_factor=1.0)
        m = DiT(**fx["cfg"])
        m.load_state_dict(P, strict=True)
        m = apply_fsdp(m, torch.bfloat16, torch.float32, device="cpu")
        fs = m._fsdp
        _, settings = m.get_mup_setup(fx["lr"], fx["wd"], fx["consts"])
        assert {k: (v["lr"], v["wd"]) for k, v in settings.items()} == \
               {k: (v["lr"], v["wd"]) for k, v in fx["settings"].items()}
        fs.pre_forward_root()
        weights = {}
        for gi, g in enumerate(m._groups):
            if gi > 0:
                fs.pre_forward_block(gi - 1)
            for n in g.names:
                weights[n] = g.w(n).clone()          # bf16, as the reference's all-gathered parameters
        sl = slice(2 * rank, 2 * rank + 2)
        Pg = {k: w.clone().requires_grad_(True) for k, w in weights.items()}
        b = fx["batch"]
        loss = O.train_forward(Pg, cfg, b["latent"][sl], b["context"][sl], b["z"][sl], b["noise"][sl],
                               tuple(fx["rope_start"]), compute_dtype=torch.bfloat16)
        loss.backward()
        assert abs(loss.item() - fx["losses"][rank]) <= 1e-2 * fx["losses"][rank]
        for g in m._groups:
            g.gfull.zero_()
        fs.pre_backward_root()
        for i in reversed(range(cfg.depth)):
            g = m.block_group(i)
            for n in g.names:
                if Pg[n].grad is not None:
                    g.g(n).copy_(Pg[n].grad.float())
            fs.post_backward_block(i)
        for n in m.root_group.names:
            m.root_group.g(n).copy_(Pg[n].grad.float())
        fs.post_backward_root()
        # (1) reduced gradient: this rank's flat piece vs the same elements of the reference's full reduced
        # gradient (the reference's ranks hold Shard(0) rows instead: layouts differ, values must not);
        # (2) one AdamW step on the local shard vs the reference's parameters after its optimizer step
        worst_g, worst_p = 0.0, 0.0
        for g in m._groups:
            for n in g.names:
                lo, hi = g.local_range(n)
                if hi == lo or n not in fx["reduced_grads"]:
                    continue
                g0 = g.rank * g.shard + lo - g.offsets[n]
                ref_g = fx["reduced_grads"][n].reshape(-1)[g0:g0 + (hi - lo)]
                p = g.params[n]
                if hi - lo >= 64 and not n.endswith("lambda_param"):
                    e = ((p.grad - ref_g).norm() / (ref_g.norm() + 1e-30)).item()
                    worst_g = max(worst_g, e)
                    assert e <= 3e-2, (n, e)          # two bf16 executions of the same step (oracle vs torch ops)
                loc = p.data.clone()
                O.adamw_step(loc, ref_g.clone(), torch.zeros_like(loc), torch.zeros_like(loc), 1,
                             settings[n]["lr"], settings[n]["wd"])
                ref_p = fx["params_after_step"][n].reshape(-1)[g0:g0 + (hi - lo)]
                assert torch.allclose(loc, ref_p, rtol=1e-5, atol=1e-7), n   # same gradient -> same update
                worst_p = max(worst_p, (loc - ref_p).abs().max().item())
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception:
        import traceback
        q.put((rank, traceback.format_exc()))


def _run_two_ranks(target, *extra, world=2):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port, q) + extra) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=540) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in results:
        assert msg == "ok", f"rank {rank}:\n{msg}"


@pytest.mark.timeout(600)
def test_two_rank_step_matches_the_reference_fsdp_run(golden_dir):
    """the reference's own 2-rank `apply_fsdp` train step (oracle/make_golden_fsdp.py)"""
    _run_two_ranks(_g6_worker, golden_dir)


def test_reference_shard_shapes_and_reduction_contract(golden_dir):
    """what the reference's wrap does, as recorded from it: every parameter Shard(0) over the 2 ranks, fp32 reduced
    gradients, and the reduced gradient == mean of the ranks' single-process bf16 gradients (bit-identical)"""
    from video_diffusion_speedrun_amd.fsdp import reference_local_shape
    fx = torch.load(os.path.join(golden_dir, "g6_fsdp.pt"), weights_only=False)
    cfg = O.DiTConfig(**fx["cfg"])
    shapes = O.param_shapes(cfg)
    assert fx["grad_dtype"] == ["torch.float32"] and fx["reduced_vs_single_bf16_mean_worst_rel"] == 0.0
    assert all("Shard(dim=0)" in v for v in fx["placements"].values())
    for r in range(2):
        rec = fx[f"local_shapes_rank{r}"]
        assert set(rec) == set(shapes)
        for n, full in shapes.items():
            assert tuple(rec[n]) == reference_local_shape(full, 2, r), (n, r)
    assert tuple(fx["local_shapes_rank1"]["blocks.1.lambda_param"]) == (0,)   # a 1-element parameter
    # this build's flat layout holds the same elements, split evenly per group instead of per tensor
    m, _, _ = make_model()
    named = [(n, p) for n, p in m.named_parameters() if n.startswith("blocks.1.")]
    for r in range(2):
        g = FlatGroup("blocks.1", named, 2, r)
        mine = sum(hi - lo for lo, hi in (g.local_range(n) for n in g.names))
        theirs = sum(int(torch.Size(fx[f"local_shapes_rank{r}"][n]).numel()) for n in g.names)
        assert abs(mine - theirs) <= 2 * 256 + ALIGN * len(g.names)


@pytest.mark.timeout(600)
def test_two_rank_gloo_sharded_step_matches_single_process():
    _run_two_ranks(_worker)


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world,prefetch", [(4, 1), (8, 0), (8, 2)], ids=["w4_window1", "w8_all_upfront", "w8_window2"])
def test_four_and_eight_rank_gloo_sharded_step(world, prefetch):
    """the shard layout (flat pieces, lambda ownership), the per-group bf16 all-gathers -- all
    up-front or through a bounded window (`ShardRuntime.prefetch`, FSDP2-like at 1) -- and the fp32
    reduce-scatter-averages at the world sizes the driver's scaling run uses (model.py:523-541), on gloo: every
    rank must end up with its slice of the single-process batch-mean gradient and AdamW update."""
    _run_two_ranks(_worker, prefetch, world=world)


def _reshard_worker(rank, world, port, q, prefetch=1):
    """the memory-bounded mode (`apply_fsdp(..., reshard_after_forward=True)`, reference model.py:525,541): ring of
    parameter buffers, release after forward, re-gather in backward"""
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        torch.set_num_threads(1 if world > 2 else 2)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from video_diffusion_speedrun_amd.fsdp import ReshardRuntime, apply_fsdp
        os.environ["VDS_AG_PREFETCH"] = str(prefetch)
        m, cfg, P = make_model(depth=5)  # deeper than the ring: buffers are reused within a pass
        m = apply_fsdp(m, torch.bfloat16, torch.float32, device="cpu", reshard_after_forward=True)
        fs = m._fsdp
        assert isinstance(fs, ReshardRuntime) and fs.n_slots == min(max(1, prefetch) + 2, cfg.depth)
        assert all(g.full is None for g in m._groups[1:])  # no resident per-block copies
        full, halves = _batches(world)
        weights = {}
        for step in range(2):  # the second pass finds the ring in the state the first one left
            n0 = fs.n_all_gather
            fs.pre_forward_root()
            for gi, g in enumerate(m._groups):
                if gi > 0:
                    fs.pre_forward_block(gi - 1)
                    held = [k for k in range(1, len(m._groups)) if fs.slot_of[k] is not None]
                    assert gi in held and len(held) <= fs.n_slots and max(held) <= gi + max(1, prefetch), (gi, held)
                for n in g.names:
                    weights[n] = g.w(n).clone()
                    assert torch.equal(weights[n], P[n].to(torch.bfloat16)), n
                if gi > 0:
                    fs.post_forward_block(gi - 1)
                    assert (g.full is None) == (gi - 1 < cfg.depth - 1)  # every block but the last is released
            fs.post_forward_root()
            assert fs.n_all_gather - n0 == 1 + cfg.depth
            grads, _ = _oracle_grads(cfg, weights, halves[rank])
            for g in m._groups:
                g.gfull.zero_()
            fs.pre_backward_root()
            for i in reversed(range(cfg.depth)):
                fs.pre_backward_block(i)
                g = m.block_group(i)
                for n in g.names:  # gathered again: the same bf16 words
                    assert torch.equal(g.w(n), weights[n]), n
                    g.g(n).copy_(grads[n])
                fs.post_backward_block(i)
                assert g.full is None
            for n in m.root_group.names:
                m.root_group.g(n).copy_(grads[n])
            fs.post_backward_root()
            # forward: root + every block; backward: every block but the last again
            assert fs.n_all_gather - n0 == 1 + cfg.depth + (cfg.depth - 1)
            assert sorted(fs.free_slots) == list(range(fs.n_slots))
        per_rank = [_oracle_grads(cfg, weights, h)[0] for h in halves]
        gfull = {n: sum(gr[n] for gr in per_rank) / world for n in per_rank[0]}
        for g in m._groups:
            for n in g.names:
                lo, hi = g.local_range(n)
                g0 = g.rank * g.shard + lo - g.offsets[n]
                ref = gfull[n].reshape(-1)[g0:g0 + (hi - lo)]
                assert torch.allclose(g.params[n].grad, ref, rtol=1e-4, atol=1e-7), n
        # a forward without a backward (sampling) leaves the last block gathered; the next forward starts clean
        fs.pre_forward_root()
        for i in range(cfg.depth):
            fs.pre_forward_block(i)
            fs.post_forward_block(i)
        fs.pre_forward_root()
        assert len(fs.free_slots) == fs.n_slots - min(max(1, prefetch), cfg.depth)
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception:
        import traceback
        q.put((rank, traceback.format_exc()))


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world,prefetch", [(2, 1), (4, 2)], ids=["w2_window1", "w4_window2"])
def test_reshard_after_forward_mode(world, prefetch):
    """the reference's `reshard_after_forward` (model.py:525,541) as an optional memory-bounded mode of
    the sharding runtime: block copies in a ring of prefetch + 2 buffers, released after their forward (all but the last
    block's), gathered again in backward, same reduced gradients as the resident mode."""
    _run_two_ranks(_reshard_worker, prefetch, world=world)
