"""Generate the sharding-level fixture (SURVEY.md §8(c) G5 recipe; stored as g6 because `g5_sampler.pt` took
the number) and a checkpoint directory WRITTEN BY THE REFERENCE's own save path.

Run in the build container only:   python oracle/make_golden_fsdp.py

It starts two CPU ranks (gloo) which import the reference from /root/reference and run, unmodified:
  * `model.apply_fsdp(dit, bf16, fp32)` (model.py:512-542; only `init_device_mesh` is pointed at "cpu" -- the
    reference hard-codes "cuda", model.py:498),
  * one train step on each rank's own micro-batch: DiT forward (bf16 compute under the FSDP policy), the flow
    loss of train.py:121-125, backward (fp32 reduce-scatter-average), `get_mup_setup` + `torch.optim.AdamW`
    (train.py:335-344; betas (0.95, 0.99)),
  * `get_model_state_dict(dit)` + `dcp.save(...)` (train.py:553,581-584) into tests/golden/g6_dcp/.

Written:
  tests/golden/g6_fsdp.pt   inputs (full batch of 4 = 2 ranks x 2), RoPE offsets, per-rank local shard shapes of
                            every parameter, the reduced gradient as full tensors, the single-process bf16 and
                            fp32 gradients of both micro-batches (the reduced gradient must equal their mean),
                            parameters after the optimizer step
  tests/golden/g6_dcp/      the reference-written DCP directory (weights AFTER the step, `Shard(0)` pieces of two
                            ranks, plus the persistent `rope.freqs_hwt_*` buffers)

To keep the directory at a few MB the reference's `ThreeDimRotary` is instantiated with an 8^3 position table
(the class, its buffer names and persistence are the reference's; DiT's ctor asks for 128^3 = 2 x 268 MB).
"""
import importlib.machinery
import os
import shutil
import socket
import sys
import types

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, "/root/reference")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402

GOLD = os.path.join(REPO, "tests", "golden")
CFG = dict(in_channels=16, patch_size=2, time_patch_size=2, hidden_size=128, depth=2, num_heads=2,
           cross_attn_input_size=64, residual_v=True, train_bias_and_rms=False)
CONSTS = ["patch_proj", "context_kv", "positional_embedding"]
LR, WD = 1e-3, 0.1
TABLE = 8  # positions per axis of the RoPE table in this fixture


def batches():
    g = torch.Generator().manual_seed(3)
    return dict(latent=torch.randn(4, 16, 4, 8, 8, generator=g), context=torch.randn(4, 6, 64, generator=g),
                z=torch.randn(4, generator=g), noise=torch.randn(4, 16, 4, 8, 8, generator=g))


def build_reference(P, dtype=None):
    import model as ref_model
    m = ref_model.DiT(use_rope=True, **CFG)
    m.rope = ref_model.ThreeDimRotary(CFG["hidden_size"] // (2 * CFG["num_heads"]), h=TABLE, w=TABLE, t=TABLE)
    missing, unexpected = m.load_state_dict(P, strict=False)
    assert not unexpected and all("rope" in k for k in missing), (missing, unexpected)
    return m if dtype is None else m.to(dtype)


def flow_loss(out, v):
    return ((v.float() - out.float()) ** 2).mean(dim=(1, 2, 3, 4)).mean()  # train.py:121-125


def step_inputs(O, full, sl, dtype):
    x = full["latent"][sl].to(dtype)
    t = O.time_shift(full["z"][sl].to(dtype))
    z_t, v = O.noise_latents(x, full["noise"][sl].to(dtype), t)
    return z_t, full["context"][sl].to(dtype), t, v


def worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(4)
    w = types.ModuleType("wandb")
    w.__spec__ = importlib.machinery.ModuleSpec("wandb", None)
    sys.modules["wandb"] = w
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import model as ref_model
    from torch.distributed.device_mesh import init_device_mesh
    import torch.distributed.checkpoint as dcp
    from torch.distributed.checkpoint.state_dict import get_model_state_dict
    from oracle import dit_oracle as O
    ref_model.init_device_mesh = lambda dev, **kw: init_device_mesh("cpu", **kw)  # model.py:498 says "cuda"

    cfg = O.DiTConfig(**CFG)
    P = O.init_params(cfg, seed=7, randomize_zero_init=True, init_std_factor=1.0)
    full = batches()
    sl = slice(2 * rank, 2 * rank + 2)
    rope_seed = 99

    m = build_reference(P)
    m = ref_model.apply_fsdp(m, param_dtype=torch.bfloat16, reduce_dtype=torch.float32)
    groups, settings = m.get_mup_setup(LR, WD, CONSTS)
    opt = torch.optim.AdamW(groups, betas=(0.95, 0.99))
    local_shapes = {n.replace("_fsdp_wrapped_module.", ""): tuple(p.to_local().shape)
                    for n, p in m.named_parameters()}
    placements = {n: str(p.placements) for n, p in m.named_parameters()}
    z_t, ctx, t, v = step_inputs(O, full, sl, torch.bfloat16)
    torch.manual_seed(rope_seed)
    thw = (z_t.shape[2] // 2, z_t.shape[3] // 2, z_t.shape[4] // 2)
    torch.randint(0, 1, (1,))  # placeholder so both ranks hold the same RNG state below
    torch.manual_seed(rope_seed)
    start_h = torch.randint(0, TABLE - thw[1] + 1, (1,)).item()
    start_w = torch.randint(0, TABLE - thw[2] + 1, (1,)).item()
    start_t = torch.randint(0, TABLE - thw[0] + 1, (1,)).item()
    torch.manual_seed(rope_seed)
    out = m(z_t, ctx, t)
    assert out.dtype == torch.bfloat16
    loss = flow_loss(out, v)
    opt.zero_grad()
    loss.backward()
    # blocks.0.lambda_param takes no part in the forward (block 0 has no v_0 to mix with): its grad stays None
    reduced = {n: p.grad.full_tensor().clone() for n, p in m.named_parameters() if p.grad is not None}  # collective
    grad_dtype = {str(p.grad.dtype) for p in m.parameters() if p.grad is not None}
    opt.step()
    after = {n: p.full_tensor().detach().clone() for n, p in m.named_parameters()}  # collective
    losses = [torch.zeros(()) for _ in range(world)]
    dist.all_gather(losses, loss.detach().float())

    # the reference's checkpoint path, unmodified (train.py:553,581-584)
    state_dict = get_model_state_dict(m)
    dcp.save(state_dict, checkpoint_id=os.path.join(tmp, "dcp"))
    dist.barrier()

    if rank == 0:
        # single-process reference gradients of both micro-batches: bf16 (what each FSDP rank computes) and fp32
        single = {}
        for dtype, key in ((torch.bfloat16, "bf16"), (torch.float32, "fp32")):
            per_rank = []
            for r in range(world):
                ms = build_reference(P, dtype)
                zt_r, ctx_r, t_r, v_r = step_inputs(O, full, slice(2 * r, 2 * r + 2), torch.bfloat16)
                torch.manual_seed(rope_seed)
                o = ms(zt_r.to(dtype), ctx_r.to(dtype), t_r.to(dtype))
                flow_loss(o, v_r).backward()
                per_rank.append({n: p.grad.float().clone() for n, p in ms.named_parameters() if p.grad is not None})
            single[key] = {n: sum(g[n] for g in per_rank) / world for n in per_rank[0]}

        def rel(a, b):
            return ((a.float() - b.float()).norm() / (b.float().norm() + 1e-30)).item()
        worst = max(rel(reduced[n], single["bf16"][n]) for n in reduced)
        worst32 = max(rel(reduced[n], single["fp32"][n]) for n in reduced if not n.endswith("lambda_param"))
        print(f"[g6] reduced FSDP gradient vs mean of single-process bf16 gradients: worst rel {worst:.2e}; "
              f"vs fp32: {worst32:.2e}; grad dtypes {grad_dtype}; losses {[float(l) for l in losses]}")
        assert worst < 2e-2
        fx = {"cfg": dict(CFG), "param_seed": 7, "lr": LR, "wd": WD, "consts": CONSTS, "world": world,
              "rope_table": TABLE, "rope_start": (start_t, start_h, start_w), "batch": full,
              "local_shapes_rank0": local_shapes, "placements": placements, "grad_dtype": sorted(grad_dtype),
              "losses": [float(l) for l in losses],
              # the mean of the two single-process bf16 gradients is NOT stored: it equals `reduced_grads` to the
              # worst relative error recorded here (0.0 = bit-identical); the fp32 truth is regenerated by the oracle
              "reduced_grads": reduced, "reduced_vs_single_bf16_mean_worst_rel": worst,
              "reduced_vs_single_fp32_mean_worst_rel": worst32, "params_after_step": after,
              "settings": {k.replace("_fsdp_wrapped_module.", ""): {"lr": s["lr"], "wd": s["wd"]}
                           for k, s in settings.items()}}
        torch.save(fx, os.path.join(tmp, "g6_fsdp.pt"))
    else:
        torch.save(local_shapes, os.path.join(tmp, "shapes_rank1.pt"))
    dist.barrier()
    dist.destroy_process_group()


def main():
    tmp = os.path.join(GOLD, "_g6_tmp")
    shutil.rmtree(tmp, ignore_errors=True)
    os.makedirs(tmp)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(worker, args=(2, port, tmp), nprocs=2, join=True)
    fx = torch.load(os.path.join(tmp, "g6_fsdp.pt"), weights_only=False)
    fx["local_shapes_rank1"] = torch.load(os.path.join(tmp, "shapes_rank1.pt"), weights_only=False)
    torch.save(fx, os.path.join(GOLD, "g6_fsdp.pt"))
    dst = os.path.join(GOLD, "g6_dcp")
    shutil.rmtree(dst, ignore_errors=True)
    shutil.move(os.path.join(tmp, "dcp"), dst)
    shutil.rmtree(tmp)
    for f in ["g6_fsdp.pt"] + [os.path.join("g6_dcp", x) for x in sorted(os.listdir(dst))]:
        print(f, os.path.getsize(os.path.join(GOLD, f)) // 1024, "KiB")


if __name__ == "__main__":
    main()
