#!/bin/bash
mkdir -p gpurun_out/r03
timeout 900 python -m pytest tests/test_model_gpu.py -x -q 2>&1 | tail -70 > gpurun_out/r03/model_fail.log
tail -5 gpurun_out/r03/model_fail.log
