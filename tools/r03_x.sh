mkdir -p gpurun_out/r03b
python -m pytest tests/test_model_gpu.py tests/test_workloads_gpu.py -x -q -m gpu > gpurun_out/r03b/mw.log 2>&1; grep -E "passed|failed|Error|error" gpurun_out/r03b/mw.log | tail -5
python bench.py --workload c5 --graph --steps 6 --warmup 3 --no-cpu-baseline > gpurun_out/r03b/c5graph.log 2>&1; tail -5 gpurun_out/r03b/c5graph.log | cut -c1-400
