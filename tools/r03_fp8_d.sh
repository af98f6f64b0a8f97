#!/bin/bash
mkdir -p gpurun_out/r03
timeout 900 python -m pytest tests/test_attn_fp8_gpu.py -q 2>&1 | tail -30 > gpurun_out/r03/fp8_attn_tests4.log
tail -12 gpurun_out/r03/fp8_attn_tests4.log
B=6 ONLY72=1 timeout 300 python tools/bench_attn.py 2>&1 | grep -v amdgpu > gpurun_out/r03/fp8_attn_bench3.log
cat gpurun_out/r03/fp8_attn_bench3.log
python bench.py --workload c5 --steps 6 --warmup 3 --no-cpu-baseline > gpurun_out/r03/c5_fp8attn_bench3.log 2>&1
tail -c 1800 gpurun_out/r03/c5_fp8attn_bench3.log
bash tools/r03_prof_fp8.sh v3 > /dev/null 2>&1
grep -A3 "^attn8_[fb]" gpurun_out/r03/fp8_attn_sq_counters_v3.txt
grep "attn_fp8\|fp8_attention" gpurun_out/parity_report.jsonl | tail -8
