#!/bin/bash
mkdir -p gpurun_out/r03
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value tools/probe_mfma_scale.hip -o /tmp/probe_mfma_scale && /tmp/probe_mfma_scale > gpurun_out/r03/probe_mfma_scale.txt 2>&1
cat gpurun_out/r03/probe_mfma_scale.txt
python bench.py --workload c5 --steps 6 --warmup 3 --no-cpu-baseline > gpurun_out/r03/c5_fp8attn_bench2.log 2>&1
tail -c 1500 gpurun_out/r03/c5_fp8attn_bench2.log
B=6 ONLY72=1 timeout 300 python tools/bench_attn.py 2>&1 | grep fp8
timeout 900 python -m pytest tests/test_attn_fp8_gpu.py tests/test_model_gpu.py -q -k "fp8 or head_dim" 2>&1 | tail -8
