#!/bin/bash
# round-3 start: fp8 probes + this box's starting numbers
set -x
mkdir -p gpurun_out/r03
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value tools/probe_tr8.hip -o /tmp/probe_tr8 && /tmp/probe_tr8 > gpurun_out/r03/probe_tr8.txt 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value tools/mfma_rate_fp8.hip -o /tmp/mfma_rate_fp8 && /tmp/mfma_rate_fp8 > gpurun_out/r03/mfma_rate_fp8.txt 2>&1
python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r03/start_c3b_bench.log 2>&1
python bench.py --workload c5 --steps 6 --warmup 3 --no-cpu-baseline > gpurun_out/r03/start_c5_bench.log 2>&1
tail -c 600 gpurun_out/r03/start_c3b_bench.log
