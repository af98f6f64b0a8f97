"""Latent data path: the reference's `LatentDataset` (sharded_dataset.py:8-32),
`create_dataloader` and `avg_scalar_across_ranks` (utils.py:11-35), plus the piece the hot path
wants on an MI355X node: a device prefetcher that stages the next batch in pinned memory and copies
it to HBM on a side HIP stream while the current step computes.

Rows are dicts {"serialized_latent": bytes of torch.save(tensor), "caption": str}, exactly the
columns of `fal/cosmos-openvid-1m`; any indexable of such rows works (the HF dataset itself is only
loaded when no rows are given -- it needs network / a local cache).
"""
from __future__ import annotations

import io
from typing import Iterator, Optional, Sequence

import torch
import torch.distributed as dist
from torch.utils.data import DataLoader, Dataset, Sampler


HF_DATASET = "fal/cosmos-openvid-1m"
HF_ROWS_USED = 1979810 // 2     # the reference trains on the first half of the dataset ...
TEST_ROWS = 40                  # ... whose last 40 rows are its test split (sharded_dataset.py:18-19)


def deserialize_tensor(serialized_tensor: bytes, device=None) -> torch.Tensor:
    """bytes written by torch.save -> tensor (the row format of the dataset, sharded_dataset.py:8-13); tensors only"""
    where = torch.device(device) if device else None
    with io.BytesIO(serialized_tensor) as stream:
        return torch.load(stream, map_location=where, weights_only=True)


def serialize_tensor(t: torch.Tensor) -> bytes:
    buf = io.BytesIO()
    torch.save(t, buf)
    return buf.getvalue()


def split_indices(split: str) -> range:
    """row indices of the reference's train / test split"""
    first_test = HF_ROWS_USED - TEST_ROWS
    return range(first_test) if split == "train" else range(first_test, HF_ROWS_USED)


class LatentDataset(Dataset):
    """Same contract as the reference's class (sharded_dataset.py:16-32): item = {"latent": tensor on the CPU,
    "prompt": caption}.  `rows`: pre-loaded rows (any indexable); otherwise the HF dataset, which needs network or a
    local cache."""

    def __init__(self, split="train", cache_dir="./cache", rows: Optional[Sequence[dict]] = None):
        if rows is None:
            from datasets import load_dataset
            rows = load_dataset(HF_DATASET, split="train", cache_dir=cache_dir).select(split_indices(split))
        self.dataset = rows

    def __len__(self):
        return len(self.dataset)

    def __getitem__(self, idx):
        row = self.dataset[idx]
        return {"latent": deserialize_tensor(row["serialized_latent"], "cpu"), "prompt": row["caption"]}


def collate_fn(batch):
    """list of items -> {"latent": stacked [B, ...], "prompt": list of str} (utils.py:21-25)"""
    latents, prompts = zip(*((item["latent"], item["prompt"]) for item in batch))
    return {"latent": torch.stack(latents), "prompt": list(prompts)}


class RankShardSampler(Sampler):
    """Disjoint, equally sized index shards per rank with a per-epoch seeded shuffle.  The reference
    gives every rank its own independently shuffled full loader (utils.py:27-34, no sampler), so ranks
    may draw the same sample in one step; `create_dataloader(shard=True)` uses this instead."""

    def __init__(self, n: int, rank: int, world: int, shuffle: bool, seed: int = 0):
        self.n, self.rank, self.world, self.shuffle, self.seed, self.epoch = n, rank, world, shuffle, seed, 0

    def set_epoch(self, epoch: int):
        self.epoch = epoch

    def __len__(self):
        return self.n // self.world

    def __iter__(self) -> Iterator[int]:
        if self.shuffle:
            g = torch.Generator().manual_seed(self.seed + self.epoch)
            order = torch.randperm(self.n, generator=g).tolist()
        else:
            order = list(range(self.n))
        per = self.n // self.world
        return iter(order[self.rank * per:(self.rank + 1) * per])


def create_dataloader(split, batch_size, num_workers, do_shuffle, prefetch_factor=8, dataset: Optional[Dataset] = None,
                      shard: bool = False, pin_memory: bool = True, seed: int = 0) -> DataLoader:
    """utils.py:18-35 (same positional arguments).  Extras: an explicit dataset, rank sharding, pinned
    host buffers for the asynchronous copy to the GPU."""
    dset = dataset if dataset is not None else LatentDataset(split=split)
    sampler = None
    if shard and dist.is_initialized() and dist.get_world_size() > 1:
        sampler = RankShardSampler(len(dset), dist.get_rank(), dist.get_world_size(), do_shuffle, seed)
    return DataLoader(dset, batch_size=batch_size, num_workers=num_workers,
                      shuffle=(do_shuffle if sampler is None else None), sampler=sampler,
                      prefetch_factor=(prefetch_factor if num_workers > 0 else None), collate_fn=collate_fn,
                      pin_memory=pin_memory and torch.cuda.is_available(), drop_last=False)


def avg_scalar_across_ranks(scalar, device=None) -> float:
    """utils.py:11-15: mean of a Python scalar over the ranks (logging only); world size 1 passes through."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return float(scalar)
    dev = device if device is not None else (torch.device("cuda", torch.cuda.current_device())
                                             if dist.get_backend() == "nccl" else torch.device("cpu"))
    t = torch.tensor(float(scalar), device=dev)
    if dist.get_backend() == "gloo":  # no AVG on gloo
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        t /= dist.get_world_size()
    else:
        dist.all_reduce(t, op=dist.ReduceOp.AVG)
    return t.item()


class DevicePrefetcher:
    """Iterates a loader one batch ahead: the next batch's latents are cast to bf16 and copied to the
    GPU on a side stream (from pinned memory) while the caller trains on the current one; the current
    stream waits on the copy only when the batch is handed over (train.py:73 does a synchronous
    `.to(device).to(bf16)` inside the step instead)."""

    def __init__(self, loader, device, dtype=torch.bfloat16):
        self.loader, self.device, self.dtype = loader, torch.device(device), dtype
        self.stream = torch.cuda.Stream(device=self.device)

    def _stage(self, batch):
        with torch.cuda.stream(self.stream):
            lat = batch["latent"]
            if not lat.is_pinned():
                lat = lat.pin_memory()
            dev = lat.to(self.device, non_blocking=True).to(self.dtype)
        out = dict(batch)
        out["latent"] = dev
        ev = torch.cuda.Event()
        ev.record(self.stream)
        return out, ev

    def __iter__(self):
        it = iter(self.loader)
        try:
            nxt = self._stage(next(it))
        except StopIteration:
            return
        while nxt is not None:
            cur, ev = nxt
            try:
                nxt = self._stage(next(it))
            except StopIteration:
                nxt = None
            torch.cuda.current_stream(self.device).wait_event(ev)
            cur["latent"].record_stream(torch.cuda.current_stream(self.device))
            yield cur

    def __len__(self):
        return len(self.loader)
