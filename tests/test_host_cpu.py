"""Host-side logic of the product package that needs no GPU: the muP table against the values the
reference's own `DiT.get_mup_setup` produced (tests/golden/g4_optim.pt), and the bookkeeping that
decides when the bf16 compute copy of the parameters must be re-cast."""
import os

import torch

from video_diffusion_speedrun_amd.model import DiT
from video_diffusion_speedrun_amd.params import FlatGroup

CONSTS = ["patch_proj", "context_kv", "positional_embedding"]


def _dit(device, **kw):
    with torch.device(device):
        return DiT(in_channels=16, patch_size=2, cross_attn_input_size=4096, residual_v=True, **kw)


def test_get_mup_setup_equals_reference_tables(golden_dir):
    fx = torch.load(os.path.join(golden_dir, "g4_optim.pt"), weights_only=False)
    cases = {"dit_s": dict(hidden_size=384, depth=12, num_heads=6, train_bias_and_rms=False),
             "dit_xl": dict(hidden_size=1152, depth=28, num_heads=16, train_bias_and_rms=False)}
    for key, kw in cases.items():
        m = _dit("meta", **kw)  # shapes and names only: no 1.1 B-parameter allocation
        groups, table = m.get_mup_setup(1e-4, 0.1, CONSTS)
        ref = fx[key]["settings"]
        assert list(table) == list(ref)  # same names, same order
        for n in ref:
            assert table[n]["lr"] == ref[n]["lr"] and table[n]["wd"] == ref[n]["wd"], n
            assert tuple(table[n]["shape"]) == tuple(ref[n]["shape"]), n
        assert len(groups) == fx[key]["n_groups"]
        assert sum(len(g["params"]) for g in groups) == len(ref)
        for g in groups:
            assert set(g) == {"params", "lr", "weight_decay"}


def test_shadow_freshness_follows_parameter_writes():
    """FlatGroup.mark_shadow_fresh: the 'bf16 shadow is current' claim dies with any in-place write through
    the nn.Parameters (load_state_dict, p.mul_(), p.copy_()), so the next gather re-casts"""
    lin = torch.nn.Linear(8, 4)
    g = FlatGroup("g", list(lin.named_parameters()))
    g.materialize("cpu")
    casts = []

    def cast(src, dst):
        casts.append(1)
        dst.copy_(src)

    g.gather(cast)
    assert len(casts) == 1 and g.shadow_fresh
    g.gather(cast)                      # nothing was written: no second cast
    assert len(casts) == 1
    with torch.no_grad():
        lin.weight.mul_(2.0)            # in-place write through the parameter
    assert not g.shadow_fresh
    g.gather(cast)
    assert len(casts) == 2 and torch.equal(g.shadow.float()[:32].view(4, 8), lin.weight.detach().bfloat16().float())
    lin.load_state_dict({"weight": torch.ones(4, 8), "bias": torch.zeros(4)})
    assert not g.shadow_fresh and g.is_current()  # non-assign load writes into the flat master
    g.gather(cast)
    assert len(casts) == 3 and float(g.shadow[:32].float().sum()) == 32.0
    g.mark_shadow_fresh()
    lin.weight.data.zero_()             # bypasses the version counter: the documented escape hatch is explicit
    assert g.shadow_fresh
    g.invalidate_shadow()
    assert not g.shadow_fresh


def test_use_rope_false_keeps_the_reference_state_dict_key():
    """model.py:312-314: `use_rope=False` registers `positional_embedding` [1, 2048, D] (zeros) in place of the RoPE
    module; train.py:287 names it among the constant muP classes.  The reference's forward cannot run such a model
    (it calls self.rope unconditionally, model.py:364); a checkpoint of one must still load with strict=True here."""
    kw = dict(hidden_size=128, depth=1, num_heads=2, train_bias_and_rms=False)
    m = _dit("cpu", use_rope=False, **kw)
    sd = m.state_dict()
    assert "positional_embedding" in sd and tuple(sd["positional_embedding"].shape) == (1, 2048, 128)
    assert float(sd["positional_embedding"].abs().max()) == 0.0
    names = [n for n, _ in m.named_parameters()]
    assert names.index("positional_embedding") == names.index("register_tokens") - 1  # the reference's registration order
    ref_like = {k: torch.randn_like(v) for k, v in sd.items()}
    m.load_state_dict(ref_like, strict=True)
    assert torch.equal(m.positional_embedding.data, ref_like["positional_embedding"])
    _, table = m.get_mup_setup(1e-4, 0.1, CONSTS)
    assert table["positional_embedding"]["lr"] == 1e-4 * 0.01 and table["positional_embedding"]["wd"] == 0.0
    assert "positional_embedding" not in _dit("meta", use_rope=True, **kw).state_dict()
    import pytest
    with pytest.raises((AttributeError, RuntimeError)):
        m(torch.zeros(1, 16, 4, 8, 8), torch.zeros(1, 4, 4096), torch.zeros(1))
