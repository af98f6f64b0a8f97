#!/usr/bin/env python3
"""Train-step throughput of the HIP video-DiT hot path (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W                      # one GPU
    python bench.py --gpus N --steps K --warmup W                      # starts N ranks itself (child torchrun)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W         # N GPUs, one rank each (RCCL)

A step = noising -> DiT forward -> MSE loss -> backward (incl. the per-group all-gather /
reduce-scatter when N > 1) -> muP-AdamW step -> LR-schedule step, on one batch of synthetic
OpenVid-shaped latents already resident in HBM.  Weak scaling: the per-GPU batch is fixed.

Prints ONE JSON line on rank 0 (metric, value = whole-job samples/s, ms_per_step, roofline of the
dominant kernel measured live with HIP events on the launch stream, cpu_baseline = the CPU oracle
timed on the host cores on a bounded sample of the same workload).
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
# multi-process GPU work on this pool needs dmabuf IPC; the HSA runtime reads this when the first HIP call
# initialises it, so it is set before torch is imported
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")



def _spawn_ranks_if_needed():
    """`python bench.py --gpus N` with N > 1 and no launcher environment: start the N ranks ourselves as a CHILD
    `torch.distributed.run` (like run_debug.sh:12 of the reference) and relay its output and exit code.  This runs
    before torch is imported, i.e. before anything in this process can have touched the GPU."""
    if "RANK" in os.environ or "WORLD_SIZE" in os.environ:
        return
    n = 1
    for i, a in enumerate(sys.argv):
        if a == "--gpus" and i + 1 < len(sys.argv):
            n = int(sys.argv[i + 1])
        elif a.startswith("--gpus="):
            n = int(a.split("=", 1)[1])
    if n <= 1:
        return
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print(f"[bench] starting {n} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    sys.exit(subprocess.call(cmd))


if __name__ == "__main__":
    _spawn_ranks_if_needed()

import torch
import torch.distributed as dist

PEAK_BF16_TFLOPS = 2500.0  # dense bf16 MFMA peak, MI355X_MICROARCH.md "Chip-level parameters"
PEAK_FP8_TFLOPS = 5000.0   # dense fp8 MFMA peak
PEAK_HBM_GBS = 8000.0
TRAFFIC_JSONS = ("r06_traffic.json", "r05_traffic.json")  # committed rocprofv3 PMC passes over whole train steps, per kernel class (tools/collect_profiles.sh); newest first
FP8_CLASSES = ("gemm_fp8", "attn_fp8_fwd", "attn_fp8_dkv", "attn_fp8_dq")
# What a dense bf16 MFMA stream reaches on THIS chip on random data: a bare v_mfma_f32_16x16x32_bf16 loop, all operands in
# registers, no LDS / memory traffic, runs 1.79-1.84 PFLOP/s at the board's power limit (tools/mfma_power.hip,
# profiles/r02_power_clock_probe.txt; 2.16 on all-zero operands) -- the clock the 2.5 PFLOP/s datasheet peak assumes
# (2.4 GHz) is not held under MFMA load.  `frac_of_practical_ceiling` prices a kernel against this figure (fp8: twice
# it, same argument); `frac` stays against the datasheet peak.
PRACTICAL_BF16_TFLOPS = 1800.0
PROF_NAMES = ["gemm_nt", "gemm_nn", "gemm_tn", "attn_fwd", "attn_bwd_delta", "attn_bwd_dkv", "attn_bwd_dq",
              "rmsnorm_mod_fwd", "rmsnorm_mod_bwd", "adamw", "qkv_rope_fwd", "qkv_rope_bwd", "gate_bwd",
              "attn_fwd_plain", "attn_bwd_dkv_plain", "attn_bwd_dq_plain", "gemm_fp8", "attn_fp8_fwd", "attn_fp8_dkv",
              "attn_fp8_dq", "fp8_quant"]

# name -> (DiT kwargs, latent [C,T,H,W], default per-GPU batch, description)
WORKLOADS = {
    # the seq~8k DiT-XL step the metric's target is quoted on (fits one GPU)
    "c3b": (dict(hidden_size=1152, depth=28, num_heads=16, time_patch_size=2), (16, 16, 64, 64), 12,
            "C3b DiT-XL/2 bf16, latents [16,16,64,64] pt=2 -> 8192+16 tokens, ctx [512,4096]"),
    # BASELINE.json configs[1]
    "c2": (dict(hidden_size=768, depth=12, num_heads=12, time_patch_size=1), (16, 16, 32, 32), 16,
           "C2 DiT-B/2 bf16, latents [16,16,32,32] pt=1 -> 4096+16 tokens, ctx [512,4096]"),
    # BASELINE.json configs[2] literal shape
    "c3a": (dict(hidden_size=1152, depth=28, num_heads=16, time_patch_size=1), (16, 17, 32, 32), 16,
            "C3a DiT-XL/2 bf16, latents [16,17,32,32] pt=1 -> 4352+16 tokens, ctx [512,4096]"),
    # BASELINE.json configs[3]
    "c4": (dict(hidden_size=1152, depth=28, num_heads=16, time_patch_size=1), (16, 33, 64, 64), 2,
           "C4 DiT-XL/2 bf16, latents [16,33,64,64] pt=1 -> 33792+16 tokens, ctx [512,4096]"),
    # BASELINE.json configs[4]: the C3b shapes with every linear of the blocks and the attention products on the fp8 MFMA
    # path (fp8.py)
    "c5": (dict(hidden_size=1152, depth=28, num_heads=16, time_patch_size=2), (16, 16, 64, 64), 12,
           "C5 DiT-XL/2 fp8 (e4m3 activations/weights/P, e5m2 gradients: all seven linears of a block and the self- and "
           "cross-attention products on the fp8 MFMA; norms, residual stream, loss, optimizer bf16/fp32), "
           "latents [16,16,64,64] pt=2 -> 8192+16 tokens, ctx [512,4096]"),
    # BASELINE.json configs[0] shape, on the GPU
    "c1": (dict(hidden_size=384, depth=12, num_heads=6, time_patch_size=2), (16, 8, 16, 16), 4,
           "C1 DiT-S/2 bf16, latents [16,8,16,16] pt=2 -> 256+16 tokens, ctx [512,4096]"),
}
LC, CC = 512, 4096


def step_flops(kw, latent_shape) -> float:
    """algorithmic FLOPs of one train step per sample = 3 x forward"""
    C, T, H, W = latent_shape
    D, depth, pt, p = kw["hidden_size"], kw["depth"], kw["time_patch_size"], 2
    N = (T // pt) * (H // p) * (W // p)
    L = N + 16
    P = pt * p * p * C
    per_block = 28 * L * D * D + 4 * L * L * D + 4 * L * LC * D + 4 * LC * CC * D + 18 * D * D
    return 3.0 * (depth * per_block + 4 * N * P * D + 20 * D * D)


def build_model(kw, device, seed):
    from video_diffusion_speedrun_amd.model import DiT
    torch.manual_seed(seed)
    with torch.device(device):
        m = DiT(in_channels=16, patch_size=2, cross_attn_input_size=CC, residual_v=True, train_bias_and_rms=False,
                use_rope=True, **kw)
        with torch.no_grad():
            for n, p in m.named_parameters():  # train.py:247-251 (x0.1 on 2-D params)
                if p.dim() == 2:
                    p.mul_(0.1)
            # the reference zero-inits these (model.py:93-94,347-350), which makes every other gradient
            # vanish; re-draw them N(0, 0.02) so that every layer does real backward work
            for n, p in m.named_parameters():
                if "adaLN_modulation" in n or "final_modulation" in n or "final_proj" in n:
                    p.normal_(0.0, 0.02)
    return m


def cpu_baseline(kw, latent_shape, flops_per_sample):
    """the CPU oracle (oracle/dit_oracle.py, fp32 torch CPU ops) on a bounded sample of the same
    workload: ONE DiT block of the model (+ embed / final layers) at the full token count, B=1,
    forward + backward + AdamW; the per-sample rate is extrapolated x depth."""
    from oracle import dit_oracle as O
    # a 1-GPU box's CPU share is 16 cores: more threads than that only oversubscribe the host
    n_threads = min(os.cpu_count() or 1, len(os.sched_getaffinity(0)), 16)
    torch.set_num_threads(n_threads)
    depth = kw["depth"]
    cfg = O.DiTConfig(in_channels=16, patch_size=2, time_patch_size=kw["time_patch_size"],
                      hidden_size=kw["hidden_size"], depth=1, num_heads=kw["num_heads"], cross_attn_input_size=CC,
                      residual_v=True, train_bias_and_rms=False)
    P = O.init_params(cfg, seed=0)
    Pg = {k: w.clone().requires_grad_(True) for k, w in P.items()}
    g = torch.Generator().manual_seed(1)
    x = torch.randn(1, *latent_shape, generator=g)
    ctx = torch.randn(1, LC, CC, generator=g)
    z = torch.randn(1, generator=g)
    n = torch.randn(1, *latent_shape, generator=g)
    t0 = time.perf_counter()
    loss = O.train_forward(Pg, cfg, x, ctx, z, n, (0, 0, 0), compute_dtype=torch.float32)
    loss.backward()
    table = O.mup_settings(O.param_shapes(cfg), 1e-4, 0.1, ["patch_proj", "context_kv", "positional_embedding"])
    with torch.no_grad():
        for k, w in Pg.items():
            if w.grad is not None:
                O.adamw_step(w, w.grad, torch.zeros_like(w), torch.zeros_like(w), 1, table[k]["lr"], table[k]["wd"])
    dt = time.perf_counter() - t0
    one_block = flops_per_sample / depth
    return {"value": 1.0 / (dt * depth), "unit": "samples/s", "cores": n_threads, "kind": "port",
            "sample": f"1 of {depth} DiT blocks (+embed/final layers) at the full token count, B=1, fp32, "
                      f"fwd+bwd+AdamW in {dt:.1f} s (~{one_block / dt / 1e12:.2f} TFLOP/s); samples/s = 1/(depth x t)"}


def cpu_baseline_c1():
    """BASELINE.md section 4 / SURVEY 8(d): config C1 (DiT-S/2, 4 clips [16,8,16,16], context [512,4096]) MEASURED on
    the host cores -- the whole model, forward + backward + AdamW, fp32 and in the reference's bf16 compute mode, one
    warm-up step and the mean of the following steps (about 5 s in all)."""
    from oracle import dit_oracle as O
    n_threads = min(os.cpu_count() or 1, len(os.sched_getaffinity(0)), 16)
    torch.set_num_threads(n_threads)
    kw, lat = WORKLOADS["c1"][0], WORKLOADS["c1"][1]
    cfg = O.DiTConfig(in_channels=16, patch_size=2, time_patch_size=kw["time_patch_size"], hidden_size=kw["hidden_size"],
                      depth=kw["depth"], num_heads=kw["num_heads"], cross_attn_input_size=CC, residual_v=True,
                      train_bias_and_rms=False)
    P = O.init_params(cfg, seed=0, randomize_zero_init=True, init_std_factor=0.1)
    table = O.mup_settings(O.param_shapes(cfg), 1e-4, 0.1, ["patch_proj", "context_kv", "positional_embedding"])
    g = torch.Generator().manual_seed(1)
    B = 4
    x, ctx = torch.randn(B, *lat, generator=g), torch.randn(B, LC, CC, generator=g)
    z, n = torch.randn(B, generator=g), torch.randn(B, *lat, generator=g)
    out = {}
    for name, dt_ in (("fp32", torch.float32), ("bf16", torch.bfloat16)):
        master = {k: w.clone() for k, w in P.items()}  # fp32 masters; bf16 leg: bf16 compute copies (model.py:516-519)
        mom = {k: (torch.zeros_like(w), torch.zeros_like(w)) for k, w in P.items()}
        times = []
        for it in range(3):
            t0 = time.perf_counter()
            Pg = {k: w.to(dt_).requires_grad_(True) for k, w in master.items()}
            loss = O.train_forward(Pg, cfg, x, ctx, z, n, (0, 0, 0), compute_dtype=dt_)
            loss.backward()
            with torch.no_grad():
                for k, w in Pg.items():
                    if w.grad is not None:
                        O.adamw_step(master[k], w.grad.float(), mom[k][0], mom[k][1], it + 1, table[k]["lr"], table[k]["wd"])
            times.append(time.perf_counter() - t0)
        t = sum(times[1:]) / len(times[1:])
        out[name] = {"samples_per_s": round(B / t, 3), "ms_per_step": round(t * 1e3, 1), "loss": round(float(loss.detach()), 5)}
    return {"workload": WORKLOADS["c1"][3] + ", B=4 (BASELINE.json configs[0])", "cores": n_threads, "kind": "port",
            "sample": "the whole DiT-S model: 3 steps of fwd+bwd+AdamW per dtype, first one dropped", **out}


def cpu_baseline_c2():
    """BASELINE.md section 4 / SURVEY 8(d): config C2 (DiT-B/2, latents [16,16,32,32], pt = 1 -> 4096+16 tokens, context
    [512,4096]), ONE measured step on the host cores: the whole 12-block model, B=1, fp32, forward + backward + AdamW
    (4.78 TFLOP; no warm-up step: one step is the bounded sample)."""
    from oracle import dit_oracle as O
    n_threads = min(os.cpu_count() or 1, len(os.sched_getaffinity(0)), 16)
    torch.set_num_threads(n_threads)
    kw, lat = WORKLOADS["c2"][0], WORKLOADS["c2"][1]
    cfg = O.DiTConfig(in_channels=16, patch_size=2, time_patch_size=kw["time_patch_size"], hidden_size=kw["hidden_size"],
                      depth=kw["depth"], num_heads=kw["num_heads"], cross_attn_input_size=CC, residual_v=True,
                      train_bias_and_rms=False)
    P = O.init_params(cfg, seed=0, randomize_zero_init=True, init_std_factor=0.1)
    table = O.mup_settings(O.param_shapes(cfg), 1e-4, 0.1, ["patch_proj", "context_kv", "positional_embedding"])
    g = torch.Generator().manual_seed(1)
    x, ctx = torch.randn(1, *lat, generator=g), torch.randn(1, LC, CC, generator=g)
    z, n = torch.randn(1, generator=g), torch.randn(1, *lat, generator=g)
    Pg = {k: w.clone().requires_grad_(True) for k, w in P.items()}
    t0 = time.perf_counter()
    loss = O.train_forward(Pg, cfg, x, ctx, z, n, (0, 0, 0), compute_dtype=torch.float32)
    loss.backward()
    with torch.no_grad():
        for k, w in Pg.items():
            if w.grad is not None:
                O.adamw_step(w, w.grad, torch.zeros_like(w), torch.zeros_like(w), 1, table[k]["lr"], table[k]["wd"])
    dt = time.perf_counter() - t0
    fl = step_flops(kw, lat)
    return {"workload": WORKLOADS["c2"][3] + ", B=1 (BASELINE.json configs[1])", "cores": n_threads, "kind": "port",
            "sample": "the whole DiT-B model, ONE measured step (fwd+bwd+AdamW), fp32, no warm-up",
            "samples_per_s": round(1.0 / dt, 5), "ms_per_step": round(dt * 1e3, 1), "TFLOP/s": round(fl / dt / 1e12, 3),
            "loss": round(float(loss.detach()), 5)}


def measure(one_step, steps, warmup, world, kw, args, fs=None, graphed=None):
    """W untimed warm-up steps (the last one ranks the kernel classes by time with HIP events around every launch),
    then exactly K steps between barrier + synchronize with events around the dominant class only.
    -> dict(dt, median_ms, breakdown, dominant, dom, loss, n_ag0, n_rs0)"""
    from video_diffusion_speedrun_amd import ops

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    breakdown, dominant, loss = {}, None, None
    for w in range(warmup):
        last = (w == warmup - 1) and not args.no_prof
        if last:
            torch.cuda.synchronize()
            ops.prof_enable()
        loss = one_step()
        if last:
            breakdown = ops.prof_collect()
            ops.prof_enable(0)
    if breakdown:
        dominant = max(breakdown, key=lambda k: breakdown[k]["ms"])
    elif not args.no_prof:  # --warmup 0: no ranking step; time the kernel that dominates every attention-heavy config
        dominant = "attn_bwd_dkv" if (kw["hidden_size"] // kw["num_heads"]) == 72 else "attn_bwd_dkv_plain"
    sync()
    r = {"n_ag0": 0, "n_rs0": 0}
    if fs is not None:
        fs.exposed_comm_ms()  # drop the warm-up's stall records
        r["n_ag0"], r["n_rs0"] = fs.n_all_gather, fs.n_reduce_scatter
    if dominant is not None and graphed is None:
        ops.prof_enable(1 << PROF_NAMES.index(dominant))  # events around the dominant kernel's launches only
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]  # per-step times (median); no syncs
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(steps):
        loss = one_step()
        marks[i + 1].record()
    sync()
    dt = time.perf_counter() - t0
    step_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(steps))
    r["median_ms"] = (step_ms[(len(step_ms) - 1) // 2] + step_ms[len(step_ms) // 2]) / 2 if step_ms else float("nan")
    if graphed is not None:  # a replay has no per-launch events: the kernel's duration comes from the eager warmup step
        r["dom"] = breakdown.get(dominant)
    else:
        r["dom"] = ops.prof_collect().get(dominant) if dominant is not None else None
    ops.prof_enable(0)
    r.update(dt=dt, breakdown=breakdown, dominant=dominant, loss=loss)
    return r


_CSRC_DIGEST = []


def csrc_digest():
    """sha256 over csrc/*.hip, *.h of this tree (the same digest tools/pmc_class_traffic.py stamps a traffic file with)"""
    if not _CSRC_DIGEST:
        import glob
        import hashlib
        root = os.path.join(REPO, "video_diffusion_speedrun_amd", "csrc")
        h = hashlib.sha256()
        for f in sorted(glob.glob(os.path.join(root, "*.hip")) + glob.glob(os.path.join(root, "*.h"))):
            h.update(os.path.basename(f).encode())
            h.update(open(f, "rb").read())
        _CSRC_DIGEST.append(h.hexdigest())
    return _CSRC_DIGEST[0]


def roofline_of(dominant, dom, workload, B):
    """the JSON `roofline` object of one kernel class measured live (dom = its launches in the timed region)"""
    mfma_bound = dom["flops"] > 0
    fp8 = dominant in FP8_CLASSES
    if mfma_bound:
        ach = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
        peak, unit = (PEAK_FP8_TFLOPS if fp8 else PEAK_BF16_TFLOPS), "TFLOP/s"
    else:
        ach = dom["bytes"] / (dom["ms"] * 1e-3) / 1e9
        peak, unit = PEAK_HBM_GBS, "GB/s"
    # HBM bytes per launch: PMC counters cannot be read from inside this process (rocprofv3 wraps the command in
    # separate --pmc passes), so the figure is the committed pass of this kernel at this workload and batch,
    # stamped with the commit and kernel symbol it was taken on; null when there is none
    traffic, traffic_src, traffic_stale = None, None, None
    for tjname in TRAFFIC_JSONS:
        try:
            tj = json.load(open(os.path.join(REPO, "profiles", tjname)))
            # round 5: one section per workload ("c3b", "c5"), every kernel class of the step, averaged over the class's
            # launches of a profiled train step; round 4's file: the self-attention kernels only, one section
            sec = tj["workloads"][workload] if "workloads" in tj else (tj if workload in (tj["workload"], "c5") else None)
            if sec is not None and sec["per_gpu_batch"] == B and dominant in sec["kernels"]:
                k = sec["kernels"][dominant]
                traffic = (2 * k["fetch_kb"] + k["write_kb"]) * 1024
                traffic_src = (f"profiles/{tjname}: rocprofv3 PMC passes ({k.get('launches_averaged', 1)} launches of "
                               f"{k['symbol']}) at commit {tj['commit']}")
                # stale = the kernel sources of this tree are not the ones the passes were measured on (sha256 over
                # csrc/*.hip, *.h recorded by tools/pmc_class_traffic.py; files without it predate round 6: stale)
                traffic_stale = tj.get("csrc_sha256") != csrc_digest()
                if traffic_stale:
                    traffic_src = "stale: " + traffic_src
                break
        except (OSError, KeyError, ValueError, TypeError):
            pass
    out = {"bound": "mfma" if mfma_bound else "hbm", "achieved": ach, "peak": peak, "unit": unit,
           "frac": ach / peak, "traffic": traffic, "traffic_source": traffic_src, "traffic_stale": traffic_stale,
           "algorithmic_bytes": dom["bytes"] / dom["launches"],
           "kernel": dominant, "launches": dom["launches"], "avg_ms": dom["ms"] / dom["launches"]}
    if mfma_bound:
        ceil = PRACTICAL_BF16_TFLOPS * (2.0 if fp8 else 1.0)
        out["frac_of_practical_ceiling"] = ach / ceil
        out["practical_ceiling"] = {"TFLOP/s": ceil, "source": "bare register-resident MFMA loop on random data at the "
                                    "board power limit, tools/mfma_power.hip + profiles/r02_power_clock_probe.txt"}
    return out


def rates_of(breakdown):
    """the HBM-bound glue is reported separately, as achieved GB/s (algorithmic bytes: every operand read
    once, every result written once) against the HBM peak; MFMA classes as TFLOP/s"""
    return {
        k: ({"TFLOP/s": round(v["flops"] / v["ms"] / 1e9, 1),
             "frac_of_peak": round(v["flops"] / v["ms"] / 1e9 / (PEAK_FP8_TFLOPS if k in FP8_CLASSES else PEAK_BF16_TFLOPS), 4),
             "dtype": "fp8" if k in FP8_CLASSES else "bf16"}
            if v["flops"] > 0 and k != "attn_bwd_delta" else  # (the delta preprocess streams O and dO: HBM-bound)
            {"GB/s": round(v["bytes"] / v["ms"] / 1e6, 1), "frac_of_peak": round(v["bytes"] / v["ms"] / 1e6 / PEAK_HBM_GBS, 4)})
        for k, v in sorted(breakdown.items(), key=lambda kv: -kv[1]["ms"]) if v["ms"] > 0}


def secondary_c5(model, one_step, kw, args, B, flops):
    """BASELINE config 5 in the same run: the SAME model, batch and optimizer with `DiT.enable_fp8()` (every linear of
    the blocks and the self- and cross-attention products on the fp8 MFMA, fp8.py).  Three untimed steps -- the first
    records the amax history on the bf16 attention kernels, the second is the first armed one, the third ranks the
    kernel classes -- then 8 timed steps; its own roofline is priced against the 5 PFLOP/s fp8 peak, and `fp8_quant`
    is the time of the separate quantise / transpose / absmax passes."""
    model.enable_fp8()
    steps, warmup = 8, 3
    r = measure(one_step, steps, warmup, 1, kw, args)
    ms = r["dt"] / steps * 1e3
    value = B * steps / r["dt"]
    loss_val = float(r["loss"].item())
    out = {"metric": "train-step samples/sec (video latents)", "value": value, "unit": "samples/s", "steps": steps,
           "warmup": warmup, "ms_per_step": ms, "ms_per_step_median": r["median_ms"], "dtype": "fp8+bf16",
           "config": {"workload": WORKLOADS["c5"][3], "per_gpu_batch": B, "step_tflop_per_sample": flops / 1e12},
           "mfma_util_step_vs_bf16_peak": value * flops / (PEAK_BF16_TFLOPS * 1e12),
           "mfma_util_step_vs_fp8_peak": value * flops / (PEAK_FP8_TFLOPS * 1e12), "loss": loss_val,
           "finite": math.isfinite(loss_val)}
    if r["dom"]:
        out["roofline"] = roofline_of(r["dominant"], r["dom"], "c5", B)
    if r["breakdown"]:
        bd = r["breakdown"]
        out["kernel_breakdown_ms"] = {k: round(v["ms"], 3) for k, v in sorted(bd.items(), key=lambda kv: -kv[1]["ms"])}
        out["kernel_breakdown_ms"]["_sum"] = round(sum(v["ms"] for v in bd.values()), 3)
        out["kernel_rates"] = rates_of(bd)
    return out


def small_batch_leg(make_step, kw, args, flops, latent_shape, device, gen, B=2):
    """SURVEY 8(d) names B=2 for C3b (report 1, 2, 4): the same model, optimizer and schedule on a per-GPU batch of 2 in
    the same run (2 untimed steps -- the second one ranks the kernel classes -- then 6 timed ones, about 1.5 s)."""
    batch = {"latent": torch.randn(B, *latent_shape, device=device, generator=gen).to(torch.bfloat16),
             "context": torch.randn(B, LC, CC, device=device, generator=gen).to(torch.bfloat16), "prompt": [""] * B}
    steps, warmup = 6, 2
    r = measure(make_step(batch), steps, warmup, 1, kw, args)
    value = B * steps / r["dt"]
    out = {"per_gpu_batch": B, "value": value, "unit": "samples/s", "steps": steps, "warmup": warmup,
           "ms_per_step": r["dt"] / steps * 1e3, "ms_per_step_median": r["median_ms"],
           "mfma_util_step": value * flops / (PEAK_BF16_TFLOPS * 1e12), "loss": float(r["loss"].item())}
    if r["dom"]:  # its own dominant kernel, timed live like the headline's (traffic: the file's section for this batch, if any)
        out["roofline"] = roofline_of(r["dominant"], r["dom"], "c3b_b2" if B == 2 else "c3b", B)
    if r["breakdown"]:
        bd = r["breakdown"]
        out["kernel_breakdown_ms"] = {k: round(v["ms"], 3) for k, v in sorted(bd.items(), key=lambda kv: -kv[1]["ms"])}
        out["kernel_breakdown_ms"]["_sum"] = round(sum(v["ms"] for v in bd.values()), 3)
        out["kernel_rates"] = rates_of(bd)
    return out


def require_vds_comm(sharded: bool, comm_mod):
    """A sharded run must be on the library's own RCCL communicator: when `comm.ensure` fell back to
    torch.distributed's (it says so on stderr) the bench exits non-zero instead of reporting a number from a
    communicator the JSON line does not name.  `VDS_COMM=torch` measures that path on purpose."""
    if sharded and comm_mod.enabled() and comm_mod.fallback_reason() is not None:
        raise SystemExit(f"bench.py: the vds RCCL communicator could not be created ({comm_mod.fallback_reason()}); "
                         "refusing to report a number from the torch.distributed fallback (set VDS_COMM=torch to "
                         "measure that path on purpose)")


def comm_knobs(world, device):
    """everything that shapes the collectives of this run, for the JSON `comm` object"""
    env = {k: os.environ[k] for k in sorted(os.environ)
           if k.startswith(("VDS_COMM", "NCCL_", "RCCL_")) or k in ("HSA_ENABLE_IPC_MODE_LEGACY", "HIP_VISIBLE_DEVICES",
                                                                     "ROCR_VISIBLE_DEVICES")}
    devs = [torch.cuda.current_device()]
    if world > 1:
        t = torch.tensor([torch.cuda.current_device()], device=device)
        allr = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allr, t)
        devs = [int(x.item()) for x in allr]
    return {"comm_stream_priority": int(os.environ.get("VDS_COMM_PRIORITY", "0")),
            "schedule_env": os.environ.get("VDS_COMM_SCHEDULE", "rccl"), "env": env, "device_ids": devs}


def comm_only(model, args, world, rank, device, desc, B):
    """Time the step's collectives alone on the communication stream: per shard group one bf16 all-gather of the
    compute copy and one fp32 reduce-scatter-average of the gradient buffer, exactly the calls of a train step
    (params.FlatGroup.gather / reduce_grads) with no compute beside them.  One JSON line: bytes, ms and GB/s per
    collective and in total, per rank."""
    from video_diffusion_speedrun_amd import comm, ops
    fs = model._fsdp
    stream = fs.comm
    groups = model._groups
    res = {"all_gather": [], "reduce_scatter": []}

    def timed(fn, n):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        with torch.cuda.stream(stream):
            for _ in range(max(1, args.warmup)):
                fn()
            if world > 1:
                dist.barrier()
            stream.synchronize()
            ev[0].record(stream)
            for _ in range(n):
                fn()
            ev[1].record(stream)
        stream.synchronize()
        return ev[0].elapsed_time(ev[1]) / n

    n = max(1, args.steps)
    for g in groups:
        g.refresh_shadow(ops.cast_f32_bf16)
        ms = timed(lambda g=g: g.gather(ops.cast_f32_bf16, fs.pg, skip_cast=True), n)
        res["all_gather"].append({"group": g.name, "bytes": 2 * g.padded, "ms": round(ms, 4),
                                  "GB/s": round(2 * g.padded / ms / 1e6, 1)})
        ms = timed(lambda g=g: g.reduce_grads(fs.pg), n)
        res["reduce_scatter"].append({"group": g.name, "bytes": 4 * g.padded, "ms": round(ms, 4),
                                      "GB/s": round(4 * g.padded / ms / 1e6, 1)})
    tot = {k: {"collectives": len(v), "bytes": sum(x["bytes"] for x in v), "ms": round(sum(x["ms"] for x in v), 3)}
           for k, v in res.items()}
    for k in tot:
        tot[k]["GB/s"] = round(tot[k]["bytes"] / tot[k]["ms"] / 1e6, 1) if tot[k]["ms"] > 0 else None
    mine = torch.tensor([tot["all_gather"]["ms"], tot["reduce_scatter"]["ms"]], device=device, dtype=torch.float64)
    allr = [torch.zeros_like(mine) for _ in range(world)]
    if world > 1:
        dist.all_gather(allr, mine)
    else:
        allr = [mine]
    ci = comm.info()
    if rank == 0:
        _flush_c_stdio()
        print(json.dumps({
            "metric": "collectives of one train step, alone on the communication stream", "unit": "ms",
            "value": tot["all_gather"]["ms"] + tot["reduce_scatter"]["ms"], "higher_is_better": False, "n_gpus": world,
            "steps": n, "warmup": args.warmup, "config": {"workload": desc, "per_gpu_batch": B},
            "backend": "vds_comm (RCCL from csrc/comm.hip)" if ci["active"] else "torch.distributed nccl",
            "rccl_version": ci["rccl_version"], "reduce_scatter_schedule": ci["schedule"], "totals": tot,
            "per_rank_ms": [[round(float(t[0]), 3), round(float(t[1]), 3)] for t in allr],
            "largest_groups": {k: sorted(v, key=lambda x: -x["bytes"])[:2] for k, v in res.items()},
            **comm_knobs(world, device)}), flush=True)
    if world > 1:
        dist.barrier()
    if dist.is_initialized():
        comm.destroy()
        dist.destroy_process_group()


def _flush_c_stdio():
    """RCCL prints its version banner through C stdio when a communicator is created; with stdout on a pipe that text sits in
    the C buffer until exit and would land BEHIND the JSON line.  Flushing it first keeps the JSON line last on stdout."""
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:  # noqa: BLE001
        pass


def build_parser():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c3b", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default: per workload)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-prof", action="store_true", help="skip the live per-kernel HIP-event timing")
    ap.add_argument("--breakdown", action="store_true", help="also print the per-kernel-class table (stderr)")
    ap.add_argument("--graph", action="store_true",
                    help="replay the whole step as one captured HIP graph (1 GPU; launch-bound workloads)")
    ap.add_argument("--force-shard-runtime", action="store_true",
                    help="1 GPU only: run the sharding runtime (streams, events, RCCL collectives on a 1-rank group)")
    ap.add_argument("--comm-only", action="store_true",
                    help="time only the step's collectives (one bf16 all-gather + one fp32 reduce-scatter-average per "
                         "shard group) on the communication stream, under both reduce-scatter schedules; no compute")
    ap.add_argument("--no-fp8-attention", action="store_true", help="c5: keep the attention products in bf16")
    ap.add_argument("--no-secondary", action="store_true",
                    help="c3b on 1 GPU: skip the `secondary` measurement (the same model and batch with DiT.enable_fp8() = "
                         "BASELINE config 5)")
    ap.add_argument("--no-small-batch", action="store_true",
                    help="c3b on 1 GPU: skip the `small_batch` measurement (the same model at a per-GPU batch of 2, the batch "
                         "SURVEY 8(d) names)")
    ap.add_argument("--deterministic", action="store_true",
                    help="run with vds_set_deterministic(1): fixed-order reductions in every backward kernel (its cost on the step)")
    ap.add_argument("--no-fp8-cross-attention", action="store_true",
                    help="c5: keep the cross-attention products in bf16 (fp8 self-attention only)")
    return ap


def main():
    args = build_parser().parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        dist.init_process_group("nccl", device_id=device)
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} rank(s)")
    if world > 1 and dist.get_world_size() != world:
        raise SystemExit(f"bench.py: process group has {dist.get_world_size()} ranks, expected {world}")

    from video_diffusion_speedrun_amd import ops
    from video_diffusion_speedrun_amd.fsdp import apply_fsdp
    from video_diffusion_speedrun_amd.optim import MuAdamW
    from video_diffusion_speedrun_amd.train import get_schedule, train_step

    kw, latent_shape, B_default, desc = WORKLOADS[args.workload]
    B = args.batch or B_default
    flops = step_flops(kw, latent_shape)

    if args.deterministic:
        ops.set_deterministic(True, 2 << 30, device)
    model = build_model(kw, device, seed=1234)  # same init on every rank
    if args.workload == "c5":
        model.enable_fp8(attention=not args.no_fp8_attention, cross_attention=not args.no_fp8_cross_attention)
    if world > 1:
        model = apply_fsdp(model, torch.bfloat16, torch.float32)
        model._fsdp.measure = True
    elif args.force_shard_runtime:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
        model = apply_fsdp(model, torch.bfloat16, torch.float32, force_runtime=True)
        model._fsdp.measure = True
    from video_diffusion_speedrun_amd import comm as _comm
    require_vds_comm(getattr(model, "_fsdp", None) is not None, _comm)
    if args.comm_only:
        if getattr(model, "_fsdp", None) is None:
            raise SystemExit("bench.py --comm-only needs the sharding runtime: --gpus N > 1 or --force-shard-runtime")
        comm_only(model, args, world, rank, device, desc, B)
        return
    groups, _ = model.get_mup_setup(1e-4, 0.1, ["patch_proj", "context_kv", "positional_embedding"])
    opt = MuAdamW(groups, betas=(0.95, 0.99))
    sched = get_schedule(opt, "cosine", 20, 10000)

    gen = torch.Generator(device=device).manual_seed(1234 + rank)
    torch.manual_seed(1234 + rank)  # RoPE offsets / caption drop come from the global RNGs like the reference
    batch = {"latent": torch.randn(B, *latent_shape, device=device, generator=gen).to(torch.bfloat16),
             "context": torch.randn(B, LC, CC, device=device, generator=gen).to(torch.bfloat16),
             "prompt": [""] * B}

    def make_step(bt):
        return lambda: train_step(model, opt, sched, bt, device, generator=gen)

    one_step = make_step(batch)

    graphed = None
    if args.graph:  # whole-step HIP-graph replay (graph.py); world_size 1 only
        from video_diffusion_speedrun_amd.graph import GraphedTrainStep
        graphed = GraphedTrainStep(model, opt, sched, device, eager_steps=1)
        # eager step(s), then capture + first replay (untimed); fp8 needs a second eager step (armed amax history)
        for _ in range(3 if args.workload == "c5" else 2):
            graphed.step(batch)

    fs = getattr(model, "_fsdp", None)
    step_fn = (lambda: graphed.step(batch)) if graphed is not None else one_step
    r = measure(step_fn, args.steps, args.warmup, world, kw, args, fs=fs, graphed=graphed)
    dt, median_ms, breakdown, dominant, dom, loss = r["dt"], r["median_ms"], r["breakdown"], r["dominant"], r["dom"], r["loss"]
    n_ag0, n_rs0 = r["n_ag0"], r["n_rs0"]
    comm_info = None
    if fs is not None:  # sharded run: what the communicator saw, per rank
        from video_diffusion_speedrun_amd import comm
        mine = torch.tensor([dt / args.steps * 1e3, fs.exposed_comm_ms() / args.steps], device=device,
                            dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        if world > 1:
            dist.all_gather(allr, mine)
        else:
            allr = [mine]
        ci = comm.info()
        gbytes = sum(g.padded for g in model._groups)
        comm_info = {"backend": "vds_comm (RCCL from csrc/comm.hip)" if ci["active"] else "torch.distributed nccl",
                     "communicator_world": ci["world"] if ci["active"] else dist.get_world_size(),
                     "rccl_version": ci["rccl_version"], "reduce_scatter_schedule": ci["schedule"],
                     "runtime": type(fs).__name__,  # ShardRuntime (resident copies) | ReshardRuntime (reshard_after_forward)
                     "all_gathers_per_step": (fs.n_all_gather - n_ag0) / args.steps,
                     "reduce_scatters_per_step": (fs.n_reduce_scatter - n_rs0) / args.steps,
                     "all_gather_bytes_per_step": 2 * gbytes, "reduce_scatter_bytes_per_step": 4 * gbytes,
                     "per_rank_ms_per_step": [round(float(t[0]), 3) for t in allr],
                     "per_rank_exposed_comm_ms_per_step": [round(float(t[1]), 3) for t in allr],
                     **comm_knobs(world, device)}
    if world > 1:
        tt = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = tt.item()
    loss_val = float(loss.item())
    if not math.isfinite(loss_val):
        raise SystemExit(f"non-finite loss {loss_val}")

    if rank == 0:
        ms = dt / args.steps * 1e3
        value = world * B * args.steps / dt
        out = {
            "metric": "train-step samples/sec (video latents)", "value": value, "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
            # BASELINE.md §4 asks for the median of the timed steps: HIP events on the compute stream between the steps
            # of rank 0 (`value` / `ms_per_step` stay the contract's total over the K steps, max over ranks)
            "ms_per_step_median": median_ms, "value_at_median": world * B / (median_ms * 1e-3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "fp8+bf16" if args.workload == "c5" else "bf16",
            "data": "synthetic (N(0,1) latents/context, random-init weights, zero-init tensors re-drawn N(0,0.02))",
            "config": {"workload": desc, "per_gpu_batch": B, "global_batch": B * world,
                       "parallelism": f"fsdp{world}" if world > 1 else "single",
                       "launch": "hip-graph replay" if graphed is not None else "eager",
                       "deterministic": bool(args.deterministic),
                       "step_tflop_per_sample": flops / 1e12},
            "mfma_util_step": value * flops / (world * PEAK_BF16_TFLOPS * 1e12),
            "loss": loss_val,
            "peak_hbm_gb": torch.cuda.max_memory_allocated(device) / 1e9,
        }
        if comm_info is not None:
            assert comm_info["communicator_world"] == world, comm_info
            out["comm"] = comm_info
        if dom:
            out["roofline"] = roofline_of(dominant, dom, args.workload, B)
        if breakdown:
            tot = sum(v["ms"] for v in breakdown.values())
            out["kernel_breakdown_ms"] = {k: round(v["ms"], 3) for k, v in
                                          sorted(breakdown.items(), key=lambda kv: -kv[1]["ms"])}
            out["kernel_breakdown_ms"]["_sum"] = round(tot, 3)
            out["kernel_rates"] = rates_of(breakdown)
            if args.breakdown:
                for k, v in sorted(breakdown.items(), key=lambda kv: -kv[1]["ms"]):
                    rate = (f"{v['flops'] / v['ms'] / 1e9:8.1f} TFLOP/s" if v["flops"] > 0 else
                            f"{v['bytes'] / v['ms'] / 1e6:8.1f} GB/s")
                    print(f"[bench] {k:16s} {v['launches']:5d} launches {v['ms']:9.3f} ms  {rate}", file=sys.stderr)
        # the companion legs must never cost the headline its JSON line: a failure is reported in their place
        if world == 1 and args.workload == "c3b" and graphed is None and fs is None and not args.no_small_batch:
            try:  # (before the secondary: that one switches the model to fp8)
                out["small_batch"] = small_batch_leg(make_step, kw, args, flops, latent_shape, device, gen)
            except Exception as e:  # noqa: BLE001
                out["small_batch"] = {"error": f"{type(e).__name__}: {e}"[:500]}
        if world == 1 and args.workload == "c3b" and graphed is None and fs is None and not args.no_secondary:
            try:
                out["secondary"] = secondary_c5(model, one_step, kw, args, B, flops)
            except Exception as e:  # noqa: BLE001
                out["secondary"] = {"error": f"{type(e).__name__}: {e}"[:500]}
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(kw, latent_shape, flops)
                out["cpu_baseline"]["c1_measured"] = cpu_baseline_c1()
                out["cpu_baseline"]["c2_measured"] = cpu_baseline_c2()
            except Exception as e:  # noqa: BLE001
                out.setdefault("cpu_baseline", {})["error"] = f"{type(e).__name__}: {e}"[:500]
        _flush_c_stdio()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
    if dist.is_initialized():
        from video_diffusion_speedrun_amd import comm
        comm.destroy()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
