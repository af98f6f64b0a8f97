#!/bin/bash
mkdir -p gpurun_out/r03
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value tools/probe_cvt_u8.hip -o /tmp/probe_cvt_u8 && /tmp/probe_cvt_u8 > gpurun_out/r03/probe_cvt_u8.txt 2>&1
timeout 900 python -m pytest tests/test_attn_fp8_gpu.py -q 2>&1 | tail -40 > gpurun_out/r03/fp8_attn_tests2.log
cat gpurun_out/r03/fp8_attn_tests2.log | tail -15
bash tools/r03_prof_fp8.sh v1 > /dev/null 2>&1
cat gpurun_out/r03/fp8_attn_sq_counters_v1.txt
