"""RMSNorm+modulate / gate backward at the DiT-XL step shape: bf16 form against the fp8-emitting form
(with and without amax recording).  HIP events, random data."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_diffusion_speedrun_amd import ops

bf16, f32 = torch.bfloat16, torch.float32
B, L, D = int(os.environ.get("B", 12)), 8208, 1152


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


x = torch.randn(B * L, D, device="cuda").to(bf16)
y = torch.randn(B * L, D, device="cuda").to(bf16)
mod = torch.randn(B, 9 * D, device="cuda") * 0.3
amax = torch.full((1,), 6.0, device="cuda")
slots = torch.zeros(B * L, device="cuda")
dmod = torch.zeros(B, 9 * D, device="cuda")
print(f"rmsnorm_mod_fwd bf16      {timeit(lambda: ops.rmsnorm_mod_fwd(x, None, mod, 0, D, B, L)):8.1f} us")
print(f"rmsnorm_mod_fwd fp8       {timeit(lambda: ops.rmsnorm_mod_fwd_fp8(x, None, mod, 0, D, B, L, 0, amax, slots)):8.1f} us")
slots.zero_()
print(f"rmsnorm_mod_fwd fp8 zeroed-slots each call "
      f"{timeit(lambda: (slots.zero_(), ops.rmsnorm_mod_fwd_fp8(x, None, mod, 0, D, B, L, 0, amax, slots))):8.1f} us")
print(f"rmsnorm_mod_fwd fp8 no-amax {timeit(lambda: ops.rmsnorm_mod_fwd_fp8(x, None, mod, 0, D, B, L, 0, amax, None)):8.1f} us")
print(f"gate_bwd bf16             {timeit(lambda: ops.gate_bwd(x, y, mod, 2 * D, dmod, None, B, L)):8.1f} us")
print(f"gate_bwd fp8              {timeit(lambda: ops.gate_bwd_fp8(x, y, mod, 2 * D, dmod, None, B, L, 1, amax, slots)):8.1f} us")
print(f"gate_bwd fp8 no-amax      {timeit(lambda: ops.gate_bwd_fp8(x, y, mod, 2 * D, dmod, None, B, L, 1, amax, None)):8.1f} us")
q = torch.empty(B * L, 3 * D, dtype=torch.uint8, device="cuda").view(torch.float8_e5m2)
print(f"transpose_fp8 [BL,3D]     {timeit(lambda: ops.transpose_fp8(q)):8.1f} us")
xq = torch.randn(B * L, 3 * D, device="cuda").to(bf16)
print(f"quant_fp8 q+t [BL,3D]     {timeit(lambda: ops.quant_fp8(xq, 1, amax, True, True, amax_out=slots[:1])):8.1f} us")
