"""Checkpoint contract on CPU: reference parameter names, torch and DCP formats,
a reference-style plain state dict with wrapper prefixes, and the round trip through flat groups."""
import os

import pytest
import torch

from oracle import dit_oracle as O
from video_diffusion_speedrun_amd import checkpoint as ck
from video_diffusion_speedrun_amd.model import DiT
from video_diffusion_speedrun_amd.params import FlatGroup

CFG = dict(in_channels=16, patch_size=2, time_patch_size=2, hidden_size=128, depth=2, num_heads=2,
           cross_attn_input_size=64, residual_v=True, train_bias_and_rms=True)


def make(seed):
    cfg = O.DiTConfig(**CFG)
    P = O.init_params(cfg, seed=seed, randomize_zero_init=True, init_std_factor=1.0)
    m = DiT(**CFG)
    m.load_state_dict(P, strict=True)
    return m, P


def test_save_load_roundtrip_torch_and_dcp(tmp_path):
    m, P = make(1)
    d = str(tmp_path / "ck")
    ck.save_checkpoint(d, m, None, step=7, dcp=True)
    assert set(torch.load(os.path.join(d, "model.pt"), weights_only=False)["model"]) == set(O.param_shapes(O.DiTConfig(**CFG)))
    m2, _ = make(2)
    assert ck.load_checkpoint(d, m2) == 7
    for k, p in m2.named_parameters():
        assert torch.equal(p.data, P[k]), k
    # the DCP directory alone (what the reference writes / reads): converted like train.py:298-300
    m3, _ = make(3)
    ck.load_checkpoint(os.path.join(d, "dcp"), m3)
    for k, p in m3.named_parameters():
        assert torch.equal(p.data, P[k]), k


def test_reference_style_state_dict_with_prefixes(tmp_path):
    m, P = make(4)
    f = str(tmp_path / "temp.pt")
    torch.save({("module._orig_mod." + k): v for k, v in P.items()}, f)  # train.py:305-310 strips these
    m2, _ = make(5)
    ck.load_checkpoint(f, m2)
    for k, p in m2.named_parameters():
        assert torch.equal(p.data, P[k]), k
    bad = {k: v for k, v in P.items() if k != "final_proj.bias"}
    torch.save(bad, f)
    with pytest.raises(KeyError):
        ck.load_checkpoint(f, m2)


def test_load_into_flat_groups(tmp_path):
    """after materialisation the parameters alias flat fp32 masters: loading writes through them"""
    m, P = make(6)
    d = str(tmp_path / "ck")
    ck.save_checkpoint(d, m)
    m2, _ = make(7)
    root, blocks = m2._group_members()
    groups = [FlatGroup("root", root, 1, 0)] + [FlatGroup(f"blocks.{i}", b, 1, 0) for i, b in enumerate(blocks)]
    for g in groups:
        g.materialize("cpu")
    m2._groups = groups
    ck.load_checkpoint(d, m2)
    for k, p in m2.named_parameters():
        assert torch.equal(p.data, P[k]), k
    assert all(g.is_current() for g in groups)
    sd = m2.full_state_dict()
    assert all(torch.equal(sd[k], P[k]) for k in P)


def test_loads_the_checkpoint_directory_the_reference_wrote(tmp_path, golden_dir):
    """tests/golden/g6_dcp was written by the reference's own save path -- `get_model_state_dict(dit_model)` +
    `dcp.save` on a 2-rank FSDP model (train.py:553,581-584; oracle/make_golden_fsdp.py) -- and holds Shard(0)
    pieces of both ranks plus the persistent `rope.freqs_hwt_*` buffers.  It must load here, before the flat
    groups exist, with the values the reference had after its optimizer step."""
    import shutil
    from video_diffusion_speedrun_amd import checkpoint as ck
    from video_diffusion_speedrun_amd.model import DiT
    fx = torch.load(os.path.join(golden_dir, "g6_fsdp.pt"), weights_only=False)
    d = str(tmp_path / "ref_ckpt")
    shutil.copytree(os.path.join(golden_dir, "g6_dcp"), d)   # the loader writes temp.pt next to the shards
    state = ck.read_model_state(d)
    assert {"rope.freqs_hwt_cos", "rope.freqs_hwt_sin"} <= set(state)          # reference buffers: present, ignored
    for n, want in fx["params_after_step"].items():
        assert torch.equal(state[n], want), n
    m = DiT(**fx["cfg"])
    assert ck.load_checkpoint(d, m) == 0                                        # the reference stores no step
    for n, p in m.named_parameters():
        assert torch.equal(p.detach(), fx["params_after_step"][n]), n
