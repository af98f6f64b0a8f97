"""Run under `torch.distributed.run --nproc-per-node 2` on a box with >= 2 GPUs (tests/test_multigpu_gpu.py):
a REAL 2-rank RCCL sharded train step of the HIP path -- each rank its own micro-batch -- against the unsharded
step on the concatenated batch computed by rank 0 (gradient = average over ranks, model.py:516-519)."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from oracle import dit_oracle as O  # noqa: E402  (test infrastructure: deterministic parameter init only)


def main():
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", device_id=dev)
    from video_diffusion_speedrun_amd import comm, model as M, optim, train
    from video_diffusion_speedrun_amd.fsdp import apply_fsdp
    bf16 = torch.bfloat16
    cfg = O.DiTConfig(in_channels=16, hidden_size=144, depth=3, num_heads=2, cross_attn_input_size=64,
                      residual_v=True, train_bias_and_rms=False)
    P = O.init_params(cfg, seed=61, randomize_zero_init=True, init_std_factor=1.0)
    g = torch.Generator().manual_seed(62)
    n = 2 * world
    x = torch.randn(n, 16, 4, 8, 8, generator=g).to(bf16).to(dev)
    ctx = torch.randn(n, 16, 64, generator=g).to(bf16).to(dev)
    t = torch.linspace(0.15, 0.9, n).to(bf16).to(dev)
    v = torch.randn(n, 16, 4, 8, 8, generator=g).to(bf16).to(dev)
    start = (1, 2, 3)
    consts = ["patch_proj", "context_kv", "positional_embedding"]

    def build():
        m = M.DiT(in_channels=16, hidden_size=144, depth=3, num_heads=2, cross_attn_input_size=64, residual_v=True,
                  train_bias_and_rms=False)
        m.load_state_dict(P, strict=True)
        return m.to(dev)

    def step(m, sl):
        groups, _ = m.get_mup_setup(3e-3, 0.1, consts)
        opt = optim.MuAdamW(groups, betas=(0.95, 0.99))
        out = m(x[sl], ctx[sl], t[sl], rope_start=start)
        loss, _ = train.flow_loss(out, v[sl])
        loss.backward()
        opt.step()
        return loss

    m = apply_fsdp(build(), torch.bfloat16, torch.float32)
    assert m._fsdp is not None and comm.info()["active"] and comm.info()["world"] == world
    loss = step(m, slice(2 * rank, 2 * rank + 2))
    got = m.full_state_dict()  # collective
    losses = [torch.zeros((), device=dev) for _ in range(world)]
    dist.all_gather(losses, loss.detach().float())
    if rank == 0:
        ref = build()
        loss_ref = step(ref, slice(0, n))
        want = ref.full_state_dict()
        assert abs(sum(l.item() for l in losses) / world - loss_ref.item()) <= 1e-5 * abs(loss_ref.item())
        for k in want:
            e = ((got[k] - want[k]).norm() / (want[k].norm() + 1e-20)).item()
            assert e <= 1e-4, (k, e)
        print("TWO_RANK_STEP_OK", flush=True)
    dist.barrier()
    comm.destroy()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
