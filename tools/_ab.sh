python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; grep -E "passed|failed|error|Error" gpurun_out/pytest_gpu.log | tail -5
for o in 0 1 0 1; do B=6 ONES=$o python tools/bench_attn.py 2>&1 | grep -v amdgpu | grep "hd72\|dkv hd64" | grep "dkv" | tr '\n' ' '; echo " (ones=$o)"; done
