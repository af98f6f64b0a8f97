"""bench.py without a GPU: the workload table prices a step at the algorithmic FLOPs SURVEY.md 8(d) states, and the
CLI keeps the driver's contract (flags, defaults)."""
import importlib.util
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_bench():
    spec = importlib.util.spec_from_file_location("bench", os.path.join(REPO, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.mark.parametrize("name,tflop", [("c2", 4.78), ("c3a", 22.70), ("c3b", 54.15), ("c4", 555.5), ("c5", 54.15)])
def test_step_flops_match_the_survey_table(name, tflop):
    b = load_bench()
    kw, latent, batch, desc = b.WORKLOADS[name]
    assert abs(b.step_flops(kw, latent) / 1e12 - tflop) / tflop < 2e-3
    assert 1 <= batch <= 16 and desc.startswith(name.upper()[:2])


def test_default_workload_is_the_seq8k_dit_xl_step():
    b = load_bench()
    kw, latent, batch, desc = b.WORKLOADS["c3b"]
    assert kw == dict(hidden_size=1152, depth=28, num_heads=16, time_patch_size=2) and latent == (16, 16, 64, 64)
    assert b.PEAK_BF16_TFLOPS == 2500.0 and b.PEAK_HBM_GBS == 8000.0


def test_cli_flags_and_gpu_requirement():
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--help"], capture_output=True, text=True)
    assert out.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup", "--workload", "--graph"):
        assert flag in out.stdout
    import torch
    if not torch.cuda.is_available():  # no silent CPU path: the bench refuses to run without the HIP device
        out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "1", "--warmup", "0"],
                             capture_output=True, text=True)
        assert out.returncode != 0 and "needs a GPU" in (out.stderr + out.stdout)


def test_gpus_n_starts_n_ranks_itself():
    """`python bench.py --gpus 2` with no launcher environment must start 2 ranks (child torch.distributed.run) and
    relay their exit code; here, without a GPU, every rank refuses to run -- which proves they were started"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, env=env, timeout=300)
    text = out.stderr + out.stdout
    assert "starting 2 ranks" in text and "--nproc-per-node=2" in text
    import torch
    if not torch.cuda.is_available():
        assert out.returncode != 0 and text.count("needs a GPU") >= 2


def test_launcher_world_size_must_match_gpus_flag():
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    b = load_bench()
    assert b._spawn_ranks_if_needed.__doc__  # importable without side effects (the spawn runs under __main__ only)
    import torch
    if torch.cuda.is_available():
        out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1"],
                             capture_output=True, text=True, env=env, timeout=300)
        assert out.returncode != 0 and "WORLD_SIZE=1" in (out.stderr + out.stdout)


def test_a_fallen_back_communicator_fails_the_bench():
    """`bench.py --gpus N` must exit non-zero when comm.ensure fell back to torch.distributed"""
    b = load_bench()

    class Fake:
        def __init__(self, on, why):
            self._on, self._why = on, why

        def enabled(self):
            return self._on

        def fallback_reason(self):
            return self._why
    b.require_vds_comm(True, Fake(True, None))                      # the library's communicator is in use
    b.require_vds_comm(False, Fake(True, "dlopen failed"))          # not sharded: nothing to check
    b.require_vds_comm(True, Fake(False, "dlopen failed"))          # VDS_COMM=torch: the torch path on purpose
    with pytest.raises(SystemExit) as e:
        b.require_vds_comm(True, Fake(True, "vds_comm_init(rank=1, world=2) -> -3"))
    assert e.value.code not in (0, None) and "vds_comm_init" in str(e.value.code)
    for flag in ("--comm-only", "--force-shard-runtime", "--batch"):
        assert flag in subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--help"], capture_output=True,
                                      text=True).stdout


def test_c1_cpu_leg_and_line_helpers():
    """the measured C1 legs of `cpu_baseline` (DiT-S/2, 4 clips, fp32 and bf16 on the host
    cores), the per-kernel roofline object with its practical-ceiling fraction, and the profiler class table"""
    b = load_bench()
    c1 = b.cpu_baseline_c1()
    assert c1["kind"] == "port" and c1["cores"] >= 1
    for leg in ("fp32", "bf16"):
        assert c1[leg]["samples_per_s"] > 0 and c1[leg]["ms_per_step"] > 0
    from video_diffusion_speedrun_amd import _lib
    assert len(b.PROF_NAMES) == _lib.PROF_NCLASS and b.PROF_NAMES[-1] == "fp8_quant"
    r = b.roofline_of("attn_bwd_dkv", dict(flops=3.725e12 * 10, bytes=0.0, ms=63.6, launches=10), "c3b", 12)
    assert r["bound"] == "mfma" and abs(r["frac"] - 0.2343) < 1e-3
    assert abs(r["frac_of_practical_ceiling"] - r["achieved"] / 1800.0) < 1e-9
    r8 = b.roofline_of("attn_fp8_dkv", dict(flops=3.725e12, bytes=0.0, ms=4.0, launches=1), "c5", 12)
    assert r8["peak"] == 5000.0 and abs(r8["frac_of_practical_ceiling"] - r8["achieved"] / 3600.0) < 1e-9
    h = b.roofline_of("adamw", dict(flops=0.0, bytes=34e9, ms=6.6, launches=1), "c3b", 12)
    assert h["bound"] == "hbm" and "frac_of_practical_ceiling" not in h
