mkdir -p gpurun_out/r03b
python -m pytest tests/test_workloads_gpu.py -x -q -m "gpu and slow" -s > gpurun_out/r03b/slow_depth28_headline.log 2>&1
tail -15 gpurun_out/r03b/slow_depth28_headline.log
tail -3 gpurun_out/parity_report.jsonl
