python -m pytest tests/test_kernels_gpu.py tests/test_attn_fp8_gpu.py -x -q -m gpu -k "rope" 2>&1 | tail -3
