#!/bin/bash
mkdir -p gpurun_out/r03
python bench.py --workload c5 --steps 6 --warmup 3 --no-cpu-baseline > gpurun_out/r03/c5_fp8attn_bench.log 2>&1
tail -c 2500 gpurun_out/r03/c5_fp8attn_bench.log
python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r03/c3b_same_box.log 2>&1
tail -c 300 gpurun_out/r03/c3b_same_box.log
timeout 2400 python -m pytest tests/test_workloads_gpu.py -x -q -k "headline or depth28_short or depth6" 2>&1 | tail -15 > gpurun_out/r03/workloads_tests.log
cat gpurun_out/r03/workloads_tests.log
