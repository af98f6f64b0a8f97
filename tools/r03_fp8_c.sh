#!/bin/bash
mkdir -p gpurun_out/r03
timeout 900 python -m pytest tests/test_attn_fp8_gpu.py -q 2>&1 | tail -40 > gpurun_out/r03/fp8_attn_tests3.log
tail -15 gpurun_out/r03/fp8_attn_tests3.log
B=6 ONLY72=1 timeout 300 python tools/bench_attn.py > gpurun_out/r03/fp8_attn_bench2.log 2>&1
cat gpurun_out/r03/fp8_attn_bench2.log
bash tools/r03_prof_fp8.sh v2 > /dev/null 2>&1
grep -A3 "^attn8" gpurun_out/r03/fp8_attn_sq_counters_v2.txt
tail -5 gpurun_out/parity_report.jsonl
