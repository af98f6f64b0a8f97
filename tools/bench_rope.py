"""qkv -> (q, k, v) producers at the DiT-XL step shape (bf16 and fp8), HIP events; prints a digest of the outputs so
that runs with different VDS_ROPE_TILE settings (read once per process) can be compared bit for bit.
    for t in 0 2 4 8; do VDS_ROPE_TILE=$t python tools/bench_rope.py; done"""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_diffusion_speedrun_amd import ops

bf16, f32 = torch.bfloat16, torch.float32
B, L, H, hd, hdp = int(os.environ.get("B", 12)), int(os.environ.get("L", 8208)), 16, 72, 96
D = H * hd


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


def digest(*ts):
    h = hashlib.sha1()
    for t in ts:
        h.update(t.contiguous().view(torch.uint8).cpu().numpy().tobytes())
    return h.hexdigest()[:12]


torch.manual_seed(0)
qkv = torch.randn(B * L, 3 * D, device="cuda").to(bf16)
ang = torch.rand(L, hd // 2, device="cuda") * 6.28
cos, sin = ang.cos().contiguous(), ang.sin().contiguous()
v0 = torch.randn(B, H, L, hdp, device="cuda").to(bf16)
lam = torch.tensor([0.7], device="cuda").to(bf16)
tag = os.environ.get("VDS_ROPE_TILE", "default")
for name, a, b in (("block 0 (no mix)", None, None), ("mix", v0, lam)):
    q, k, v = ops.qkv_rope_fwd(qkv, cos, sin, a, b, B, L, H, hd, hdp)
    us = timeit(lambda: ops.qkv_rope_fwd(qkv, cos, sin, a, b, B, L, H, hd, hdp))
    gb = (B * L * 3 * D * 2 + 3 * B * H * L * hdp * 2 + (B * H * L * hd * 2 if a is not None else 0)) / 1e9
    print(f"tile={tag} qkv_rope_fwd bf16 {name:17s} {us:8.1f} us  {gb / us * 1e3:6.2f} TB/s  digest {digest(q, k, v)}", flush=True)
amax = torch.tensor([6.0, 0.0, 6.0, 0.0, 6.0, 0.0], device="cuda")
deq = torch.empty(8, device="cuda")
for name, a, b, wv in (("block 0 (+bf16 v)", None, None, True), ("mix", v0, lam, False)):
    f = lambda: ops.qkv_rope_fwd_fp8(qkv, cos, sin, a, b, B, L, H, hd, hdp, amax, amax[1:], 2, deq, want_v=wv)
    q8, k8, v8, vb = f()
    us = timeit(f)
    gb = (B * L * 3 * D * 2 + 3 * B * H * L * 128 + (B * H * L * hd * 2 if a is not None else 0) + (B * H * L * hdp * 2 if wv else 0)) / 1e9
    outs = [q8, k8, v8] + ([vb] if wv else [])
    print(f"tile={tag} qkv_rope_fwd fp8  {name:17s} {us:8.1f} us  {gb / us * 1e3:6.2f} TB/s  digest {digest(*outs)} amax {amax.tolist()}", flush=True)
    amax[1::2] = 0
