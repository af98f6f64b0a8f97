"""What does the partly filled last key tile cost the fp8 attention kernels?  Lq fixed at 8208 (the DiT-XL token count),
Lk = 8192 (64 full key tiles) against Lk = 8208 (64 full + one 16-key tile); a free ragged tile would cost 65/64."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_diffusion_speedrun_amd import ops

dev, HD, ROW = "cuda", 72, 128
E4, E5 = torch.float8_e4m3fn, torch.float8_e5m2
B, H, Lq = int(os.environ.get("B", 12)), 16, 8208


def rows(L, fmt):
    x = torch.zeros(B, H, L, ROW, dtype=torch.uint8, device=dev)
    x[..., :HD] = (torch.randn(B, H, L, HD, device=dev) * 40).to(fmt).view(torch.uint8)
    return x


def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    t = []
    for _ in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n):
            fn()
        e.record(); torch.cuda.synchronize()
        t.append(s.elapsed_time(e) / n)
    return sorted(t)[2]


q8 = rows(Lq, E4).view(E4)
for Lk in (8192, 8208):
    k8, v8 = rows(Lk, E4), rows(Lk, E4)
    v8[..., HD] = 0x38
    k8, v8 = k8.view(E4), v8.view(E4)
    aq, ak, E = ops.attn_fp8_qk_factors(448.0 / 8, 448.0 / 8, HD)
    deq = torch.tensor([0.02, 0.02, 0.01, 0.001, E, 0, 0, 0], dtype=torch.float32, device=dev)
    o = torch.empty(B * Lq, H * HD, dtype=torch.bfloat16, device=dev)
    lse = torch.empty(B, H, Lq, dtype=torch.float32, device=dev)
    t = timeit(lambda: ops.attn_fp8_fwd(q8, k8, v8, deq, ops.heads_view(o, B, Lq, H, HD), lse, HD))
    print(f"attn8 fwd Lq {Lq} Lk {Lk}: {t * 1e3:8.1f} us   ({t * 1e3 / ((Lk + 127) // 128):6.2f} us per key tile)")
