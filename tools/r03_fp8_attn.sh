#!/bin/bash
# fp8 attention: parity tests + micro-benchmark at the C3b attention shape
mkdir -p gpurun_out/r03
timeout 600 python -m pytest tests/test_attn_fp8_gpu.py -x -q 2>&1 | tail -30 > gpurun_out/r03/fp8_attn_tests.log
cat gpurun_out/r03/fp8_attn_tests.log
B=6 ONLY72=1 timeout 300 python tools/bench_attn.py > gpurun_out/r03/fp8_attn_bench.log 2>&1
cat gpurun_out/r03/fp8_attn_bench.log
