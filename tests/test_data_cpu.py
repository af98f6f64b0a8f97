"""Latent data path: dataset rows -> batches, rank sharding, scalar averaging."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from video_diffusion_speedrun_amd import data as D


def rows(n):
    g = torch.Generator().manual_seed(0)
    return [{"serialized_latent": D.serialize_tensor(torch.randn(16, 4, 8, 8, generator=g).to(torch.bfloat16) + i),
             "caption": f"clip {i}"} for i in range(n)]


def test_dataset_and_loader_match_the_reference_contract():
    r = rows(10)
    ds = D.LatentDataset(rows=r)
    assert len(ds) == 10
    item = ds[3]
    assert set(item) == {"latent", "prompt"} and item["prompt"] == "clip 3"
    assert torch.equal(item["latent"], D.deserialize_tensor(r[3]["serialized_latent"]))
    dl = D.create_dataloader("train", 4, 0, False, dataset=ds, pin_memory=False)
    batches = list(dl)
    assert [b["latent"].shape[0] for b in batches] == [4, 4, 2]          # no drop_last, like the reference
    assert batches[0]["latent"].shape == (4, 16, 4, 8, 8) and batches[0]["prompt"] == [f"clip {i}" for i in range(4)]
    dl2 = D.create_dataloader("train", 5, 2, True, prefetch_factor=2, dataset=ds, pin_memory=False)
    seen = sorted(p for b in dl2 for p in b["prompt"])
    assert seen == sorted(x["caption"] for x in r)


def test_rank_shard_sampler_is_a_partition():
    n, world = 23, 4
    parts = [list(D.RankShardSampler(n, r, world, True, seed=5)) for r in range(world)]
    flat = [i for p in parts for i in p]
    assert len(set(flat)) == len(flat) == (n // world) * world and all(len(p) == n // world for p in parts)
    s = D.RankShardSampler(n, 0, world, True, seed=5)
    a = list(s)
    s.set_epoch(1)
    assert list(s) != a
    assert D.avg_scalar_across_ranks(3.5) == 3.5  # no process group: passes through


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        avg = D.avg_scalar_across_ranks(float(rank + 1))
        ds = D.LatentDataset(rows=rows(8))
        dl = D.create_dataloader("train", 2, 0, True, dataset=ds, shard=True, pin_memory=False, seed=1)
        mine = sorted(p for b in dl for p in b["prompt"])
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, avg, mine))
    except Exception:
        import traceback
        q.put((rank, None, traceback.format_exc()))


def test_two_rank_average_and_disjoint_shards():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=240) for _ in ps)
    for p in ps:
        p.join(timeout=30)
    assert all(r[1] == 1.5 for r in res), res
    assert not set(res[0][2]) & set(res[1][2]) and len(res[0][2]) == len(res[1][2]) == 4
