#!/bin/bash
mkdir -p gpurun_out/r03
timeout 900 python -m pytest tests/test_attn_fp8_gpu.py tests/test_fp8_gpu.py -q 2>&1 | tail -8
timeout 900 python -m pytest tests/test_model_gpu.py -q -k "fp8 or head_dim or g1 or g2 or oracle" 2>&1 | tail -12
B=6 ONLY72=1 timeout 300 python tools/bench_attn.py 2>&1 | grep fp8 > gpurun_out/r03/fp8_attn_bench4.log
cat gpurun_out/r03/fp8_attn_bench4.log
python bench.py --workload c5 --steps 6 --warmup 3 --no-cpu-baseline > gpurun_out/r03/c5_fp8attn_bench4.log 2>&1
tail -c 1800 gpurun_out/r03/c5_fp8attn_bench4.log
timeout 2400 python -m pytest tests/test_workloads_gpu.py -x -q -k "headline or depth28_short" 2>&1 | tail -8
grep "headline_block_c5\|depth28" gpurun_out/parity_report.jsonl | tail -2 | cut -c1-1200
