mkdir -p gpurun_out/r03b
for t in 0 2 4 8; do VDS_ROPE_TILE=$t python tools/bench_rope.py; done > gpurun_out/r03b/rope.log 2>&1
python -m pytest tests/test_kernels_gpu.py tests/test_attn_fp8_gpu.py -x -q -m gpu -k "rope or producer or model" > gpurun_out/r03b/rope_tests.log 2>&1
tail -3 gpurun_out/r03b/rope_tests.log
grep fp8 gpurun_out/r03b/rope.log
