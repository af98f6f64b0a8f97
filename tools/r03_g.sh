#!/bin/bash
mkdir -p gpurun_out/r03
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_fp8_gpu.py -x -q 2>&1 | tail -8
timeout 900 python -m pytest tests/test_model_gpu.py -x -q 2>&1 | tail -8
python bench.py --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/r03/c3b_gelu_bench.log 2>&1
tail -c 2200 gpurun_out/r03/c3b_gelu_bench.log | head -c 1200
python bench.py --workload c5 --steps 6 --warmup 3 --no-cpu-baseline > gpurun_out/r03/c5_gelu_bench.log 2>&1
tail -c 300 gpurun_out/r03/c5_gelu_bench.log
