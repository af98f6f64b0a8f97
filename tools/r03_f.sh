#!/bin/bash
mkdir -p gpurun_out/r03
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -k "parity_random or fp8 or graph" 2>&1 | tail -40 > gpurun_out/r03/model_tests_f.log
cat gpurun_out/r03/model_tests_f.log
timeout 1200 python -m pytest tests/test_workloads_gpu.py -x -q -k "headline or depth28_short" 2>&1 | tail -8
grep "headline_block_c5\|depth28" gpurun_out/parity_report.jsonl | tail -2 | cut -c1-1500
WORKLOAD=c3b STEPS=200 B=4 timeout 900 python tools/soak.py > gpurun_out/r03/soak_c3b_bf16_vs_fp8.json 2> gpurun_out/r03/soak_err.log; cat gpurun_out/r03/soak_c3b_bf16_vs_fp8.json; tail -3 gpurun_out/r03/soak_err.log
for b in 1 2 4; do python bench.py --batch $b --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r03/c3b_b${b}_bench.log 2>&1; tail -c 250 gpurun_out/r03/c3b_b${b}_bench.log; echo; done
python bench.py --comm-only --gpus 1 --force-shard-runtime --steps 5 --warmup 2 > gpurun_out/r03/comm_only_w1.log 2>&1; tail -c 1500 gpurun_out/r03/comm_only_w1.log
