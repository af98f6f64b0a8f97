"""Host-side logic of the fp8 path that needs no GPU: the delayed-scaling table and the split-K choice."""
import torch

from video_diffusion_speedrun_amd import fp8 as F8
from video_diffusion_speedrun_amd import ops


def test_amax_history_rolls_without_losing_scales():
    h = F8.AmaxHistory(3, "cpu")
    assert not h.ready
    h.roll()                       # first training forward: nothing recorded yet, still no history
    assert not h.ready and float(h.tab.abs().sum()) == 0
    h.cur(0).fill_(2.0)
    h.cur(1).fill_(5.0)            # row 2 recorded nothing
    h.backward_done()              # the step completed (DiT._backward_impl)
    h.roll()
    assert h.ready
    assert h.prev(0).item() == 2.0 and h.prev(1).item() == 5.0 and h.prev(2).item() == 0.0
    assert float(h.tab[:, 1].abs().sum()) == 0
    h.cur(0).fill_(3.0)            # rows 1, 2 record nothing this step: row 1 keeps its older scale
    h.roll()
    assert h.prev(0).item() == 3.0 and h.prev(1).item() == 5.0 and h.prev(2).item() == 0.0
    # views alias the table (the kernels write through them)
    torch.maximum(h.cur(2), torch.tensor([7.0]), out=h.cur(2))
    assert h.tab[2, 1].item() == 7.0


def test_amax_history_needs_a_completed_backward_before_delayed_scaling():
    """a grad-enabled forward whose backward never ran (validation loss outside no_grad, an exception) leaves
    the gradient rows without an amax: delayed scaling must stay off until a full step has been recorded"""
    h = F8.AmaxHistory(2, "cpu")
    h.roll()
    h.cur(0).fill_(1.5)            # forward row only; the backward (row 1) never happens
    h.roll()
    assert not h.ready and h.prev(0).item() == 1.5 and h.prev(1).item() == 0.0
    h.cur(0).fill_(2.5)
    h.cur(1).fill_(1e-5)
    h.backward_done()
    h.roll()
    assert h.ready and abs(h.prev(1).item() - 1e-5) < 1e-9


def test_alignment_rule_of_the_fp8_linears():
    assert F8.supported(49248, 3456, 1152) and F8.supported(96, 432, 144)
    assert not F8.supported(25, 432, 144) and not F8.supported(96, 432, 72)


def test_wgrad_split_fills_the_chip_without_starving_the_k_loop():
    # DiT-XL qkv weight gradient at batch 6: 70 tiles of 256^2, 385 K tiles of 128 tokens -> a few splits
    s = ops._wgrad_split(14 * 5, 385, 256)
    assert 2 <= s <= 8 and 385 // s >= 8
    assert ops._wgrad_split(1000, 385, 256) == 1          # already more tiles than slots
    assert ops._wgrad_split(4, 16, 256) <= 2              # never fewer than 8 K tiles per split
