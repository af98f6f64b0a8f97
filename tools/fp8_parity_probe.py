"""Prints how far the fp8 step is from the fp32 oracle and from the bf16 HIP path (calibrates the test tolerance)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import dit_oracle as O
import video_diffusion_speedrun_amd as pkg
from video_diffusion_speedrun_amd import model, train

bf16 = torch.bfloat16


def rel(a, b):
    a, b = a.float().cpu().flatten(), b.float().cpu().flatten()
    return ((a - b).norm() / (b.norm() + 1e-20)).item()


def cosine(a, b):
    a, b = a.float().cpu().flatten(), b.float().cpu().flatten()
    return (a @ b / (a.norm() * b.norm() + 1e-30)).item()


for (D, H, depth, lat) in ((144, 2, 3, (2, 16, 4, 8, 8)), (256, 4, 2, (2, 16, 4, 16, 16))):
    cfg = O.DiTConfig(in_channels=16, hidden_size=D, depth=depth, num_heads=H, cross_attn_input_size=64,
                      residual_v=True, train_bias_and_rms=False)
    P = O.init_params(cfg, seed=51, randomize_zero_init=True, init_std_factor=1.0)
    g = torch.Generator().manual_seed(52)
    x = torch.randn(*lat, generator=g).to(bf16)
    ctx = torch.randn(lat[0], 16, 64, generator=g).to(bf16)
    t = torch.tensor([0.3, 0.8]).to(bf16)
    v = torch.randn(*lat, generator=g).to(bf16)
    start = (1, 2, 3)
    Pg = {k: w.clone().requires_grad_(True) for k, w in P.items()}
    o_ref = O.dit_forward(Pg, cfg, x.float(), ctx.float(), t.float(), start)
    l_ref, _ = O.flow_loss(v, o_ref)
    l_ref.backward()
    res = {}
    for mode in ("bf16", "fp8"):
        m = model.DiT(in_channels=16, hidden_size=D, depth=depth, num_heads=H, cross_attn_input_size=64,
                      residual_v=True, train_bias_and_rms=False)
        m.load_state_dict(P)
        m = m.to("cuda")
        if mode == "fp8":
            m.enable_fp8()
        out = m(x.cuda(), ctx.cuda(), t.cuda(), rope_start=start)
        loss, _ = train.flow_loss(out, v.cuda())
        loss.backward()
        worst = (1.0, None, 0.0)
        for k, p in m.named_parameters():
            if Pg[k].grad is None or k.endswith("lambda_param") or float(Pg[k].grad.abs().max()) == 0:
                continue
            c, e = cosine(p.grad, Pg[k].grad), rel(p.grad, Pg[k].grad)
            if c < worst[0]:
                worst = (c, k, e)
        print(f"D{D} {mode}: out rel {rel(out, o_ref):.4f}  loss rel {abs(loss.item()-l_ref.item())/l_ref.item():.5f}  "
              f"worst grad cos {worst[0]:.5f} ({worst[1]}, rel {worst[2]:.4f})")
        for name in ("blocks.0.qkv.weight", "blocks.1.mlp.0.weight", "blocks.1.mlp.2.weight", "blocks.0.attn_proj.weight"):
            p = dict(m.named_parameters())[name]
            print(f"      {name}: cos {cosine(p.grad, Pg[name].grad):.5f} rel {rel(p.grad, Pg[name].grad):.4f}")
