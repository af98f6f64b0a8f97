"""The C-ABI boundary without a GPU: libvds_hip.so loads and exports every symbol that
include/vds.h declares; the ctypes table binds exactly that set; argument checks that run
on the host reject bad calls with VDS_ERR_* (no kernel is launched here)."""
import ctypes as C
import os
import re

import pytest

from video_diffusion_speedrun_amd import _lib

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(REPO, "include", "vds.h")


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return set(re.findall(r"\b(vds_[a-z0-9_]+)\s*\(", src))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return _lib.load()


def test_header_and_ctypes_table_agree():
    assert declared_symbols() == set(_lib.SIGNATURES), declared_symbols() ^ set(_lib.SIGNATURES)


def test_library_exports_every_declared_symbol(lib):
    raw = C.CDLL(_lib.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(raw, name), name
    assert lib.vds_version() >= 1


def test_struct_layouts_match_header():
    # vds_gemm_args / vds_attn_args / vds_adamw_tensor: field counts and sizes as declared
    assert C.sizeof(_lib.AdamWTensor) == 5 * 8 + 8 + 4 + 4
    assert C.sizeof(_lib.GemmArgs) == 5 * 4 + 4 + 8 * 13 + 4 + 4 + 8  # 5 int32 + pad, 13 pointer/int64, 2 int32, colsum
    assert _lib.GemmArgs.colsum.offset == 136 and _lib.GemmArgs.split_k.offset == 132
    assert C.sizeof(_lib.AttnArgs) == 5 * 4 + 4 + 8 * (4 * 4 + 1 + 4 * 4 + 1) + 8 + 8  # + kv_pad_ones (padded) + ws_floats
    assert _lib.AttnArgs.ws_floats.offset == C.sizeof(_lib.AttnArgs) - 8
    assert C.sizeof(_lib.ProfStat) == 32
    assert C.sizeof(_lib.Fp8Out) == 7 * 8 + 4 + 4 + 8  # 7 pointer/int64, fmt + pad, colsum
    assert _lib.Fp8Out.colsum.offset == 64 and _lib.Fp8Out.fmt.offset == 56


def test_host_side_argument_checks(lib):
    a = _lib.GemmArgs()
    assert lib.vds_gemm_bf16(C.byref(a), None) == -1  # null operands
    b = _lib.AttnArgs()
    assert lib.vds_attn_fwd(C.byref(b), None) == -1
    assert lib.vds_rmsnorm_mod_fwd(None, 0, None, None, 0, 0, 0, None, 0, None, 1, 1, 8, 1e-6, None) == -1
    assert lib.vds_adamw_multi(None, None, None, 1, 1024, 0.9, 0.99, 1e-8, 1, 1.0, 1.0, None) == -1
    assert lib.vds_prof_collect(None) == -1
    assert lib.vds_prof_class_name(0) == b"gemm_nt"
    assert lib.vds_prof_class_name(_lib.PROF_NCLASS - 1) == b"fp8_quant" and lib.vds_prof_class_name(_lib.PROF_NCLASS) == b""
    # fp8 attention: null operands / unsupported head_dim are refused before any launch
    a8 = _lib.Attn8Args()
    assert lib.vds_attn_fp8_fwd(C.byref(a8), None) == -1 and lib.vds_attn_fp8_bwd(C.byref(a8), None) == -1
    assert lib.vds_attn_fp8_supported(72) == 1 and lib.vds_attn_fp8_supported(64) == 0
    assert lib.vds_attn_fp8_delta(None, 0, 0, None, 0, 0, None, None, None, None, None, None, 1, 1, 1, 72, None) == -1
    assert lib.vds_qkv_rope_fwd_fp8(None, None, None, None, None, None, None, None, None, None, None, 2, None, 1, 1, 1,
                                    72, 96, None) == -1
    # fp8 entry points: null operands / bad alignment / unsupported layout are refused before any launch
    assert lib.vds_gemm_fp8(C.byref(a), None, None, 0, 0, None, None) == -1
    g = _lib.GemmArgs(1, 0, 16, 16, 16, 8, 16, 8, 16, 8, 16)   # layout NN: fp8 operands are NT only
    assert lib.vds_gemm_fp8(C.byref(g), None, None, 0, 0, None, None) == -2
    g = _lib.GemmArgs(0, 0, 16, 16, 24, 8, 24, 8, 24, 8, 16)   # K not a multiple of 16
    assert lib.vds_gemm_fp8(C.byref(g), None, None, 0, 0, None, None) == -1
    assert lib.vds_quant_fp8(None, 0, 1, 8, 0, None, None, 0, None, 0, None, None, None) == -1
    assert lib.vds_absmax(None, 0, 1, 8, None, None) == -1
    assert lib.vds_rope_rows_dev(None, None, None, None, 1, 1, 1, 1, 1, None, 0, None, None, None) == -1
    assert lib.vds_adamw_multi_dev(None, None, None, 1, 1024, 0.9, 0.99, 1e-8, None, 1.0, None) == -1


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(REPO, "video_diffusion_speedrun_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(root, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt, f


def test_model_refuses_cpu_tensors():
    import torch
    from video_diffusion_speedrun_amd.model import DiT
    m = DiT(in_channels=16, hidden_size=128, depth=1, num_heads=2, cross_attn_input_size=64)
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 16, 2, 4, 4), torch.zeros(1, 4, 64), torch.zeros(1))


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    """no CPU / eager fallback: without libvds_hip.so the product raises instead of computing"""
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "libvds_hip.so"))
    with pytest.raises(_lib.VdsError):
        _lib.load()
    import torch
    from video_diffusion_speedrun_amd import ops
    with pytest.raises((_lib.VdsError, AssertionError)):
        ops.cast_f32_bf16(torch.zeros(8), torch.zeros(8, dtype=torch.bfloat16))


def test_bench_flop_model_matches_the_oracle_formula():
    """bench.py prices a step with BASELINE.md's formula; the oracle carries the same one"""
    import bench
    from oracle import dit_oracle as O
    for name in ("c3b", "c2", "c4"):
        kw, shape, _, _ = bench.WORKLOADS[name]
        cfg = O.DiTConfig(in_channels=16, patch_size=2, time_patch_size=kw["time_patch_size"], hidden_size=kw["hidden_size"],
                          depth=kw["depth"], num_heads=kw["num_heads"], cross_attn_input_size=4096)
        C, T, H, W = shape
        N = (T // kw["time_patch_size"]) * (H // 2) * (W // 2)
        assert abs(bench.step_flops(kw, shape) - O.train_step_flops(cfg, N)) <= 1e-9 * bench.step_flops(kw, shape)
    assert abs(bench.step_flops(*bench.WORKLOADS["c3b"][:2]) / 1e12 - 54.15) < 0.01


def test_schedule_and_wgrad_split_helpers():
    from oracle import dit_oracle as O
    from video_diffusion_speedrun_amd.train import lr_lambda
    from video_diffusion_speedrun_amd.ops import _wgrad_split
    for kind in ("cosine", "linear", "constant"):
        for s in (0, 1, 19, 20, 21, 500, 9999):
            assert abs(lr_lambda(s, kind, 20, 10000) - O.lr_lambda(s, kind, 20, 10000)) < 1e-12
    assert _wgrad_split(324, 16, 512) == 1          # short contraction: never split below 8 K tiles
    s = _wgrad_split(81, 513, 512)
    assert 2 <= s <= 16 and 81 * s <= 1024


def test_knob_table_and_deterministic_switch(lib):
    """round 6: the library's tuning knobs are ONE table (csrc/config.h), filled from the environment when the library is
    loaded and changed only through the C ABI; the deterministic mode needs its caller-allocated workspace (host-side
    argument checks only: nothing is launched)."""
    import math
    import re
    assert lib.vds_knob_get(b"gemm_narrow") == 1.0 and lib.vds_knob_get(b"gemm_group_m") == 4.0
    assert lib.vds_knob_get(b"attn_mfma16") == 7.0 and lib.vds_knob_get(b"deterministic") == 0.0
    assert math.isnan(lib.vds_knob_get(b"no_such_knob")) and lib.vds_knob_set(b"no_such_knob", 1.0) == -1
    assert lib.vds_knob_set(b"gemm_narrow", 0.0) == 0 and lib.vds_knob_get(b"gemm_narrow") == 0.0
    assert lib.vds_knob_set(b"gemm_narrow", 1.0) == 0
    assert lib.vds_knob_set(b"deterministic", 1.0) == -1  # only with its workspace: vds_set_deterministic
    assert lib.vds_set_deterministic(1, None, 0) == -1 and lib.vds_set_deterministic(2, None, 0) == -1
    assert lib.vds_set_deterministic(0, None, 0) == 0 and lib.vds_knob_get(b"deterministic") == 0.0
    assert lib.vds_flow_loss_workspace_floats(4, 1 << 20) == 4 * 256 and lib.vds_flow_loss_workspace_floats(3, 100) == 3
    # every knob of config.h is in the table, and no kernel file reads the environment
    csrc = os.path.join(REPO, "video_diffusion_speedrun_amd", "csrc")
    enum = re.search(r"enum Knob \{(.*?)N_KNOBS", open(os.path.join(csrc, "config.h")).read(), re.S).group(1)
    names = re.findall(r"^\s*([A-Z0-9_]+),", enum, re.M)
    assert len(names) >= 20
    for n in names:
        assert not math.isnan(lib.vds_knob_get(n.lower().encode())), n
    for f in os.listdir(csrc):
        if f.endswith(".hip") and f != "config.hip":
            assert "getenv" not in open(os.path.join(csrc, f)).read(), f
