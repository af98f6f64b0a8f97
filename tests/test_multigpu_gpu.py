"""Real multi-rank RCCL execution of the sharded HIP step.  These tests need >= 2 GPUs in one box and skip
themselves otherwise (the 1-GPU boxes of the build pool); on 1 GPU the same code paths are covered by
test_model_gpu.py::test_shard_runtime_on_one_gpu_matches_unsharded (1-rank RCCL communicator through vds_comm_*),
::test_two_emulated_ranks_on_one_gpu and the 2-rank gloo tests of test_sharding_cpu.py."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _need_two_gpus():
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs in one box")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _clean_env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    return env


@pytest.mark.timeout(900)
def test_two_rank_rccl_sharded_step_equals_unsharded_step():
    _need_two_gpus()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(REPO, "tests", "helpers", "two_rank_step.py")]
    out = subprocess.run(cmd, capture_output=True, text=True, env=_clean_env(), timeout=800)
    assert out.returncode == 0 and "TWO_RANK_STEP_OK" in out.stdout, out.stdout[-3000:] + out.stderr[-3000:]


@pytest.mark.timeout(900)
def test_bench_gpus_2_reports_two_ranks_seen_by_rccl():
    _need_two_gpus()
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--workload", "c1", "--steps", "3",
           "--warmup", "2", "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, env=_clean_env(), timeout=800)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["parallelism"] == "fsdp2" and line["scaling"] == "weak"
    c = line["comm"]
    assert c["communicator_world"] == 2 and len(c["per_rank_ms_per_step"]) == 2
    assert c["all_gathers_per_step"] == 13 and c["reduce_scatters_per_step"] == 13   # root + 12 DiT-S blocks
