mkdir -p gpurun_out/r03b
python -m pytest tests/test_fp8_gpu.py -x -q -m gpu > gpurun_out/r03b/fp8_tests.log 2>&1
tail -3 gpurun_out/r03b/fp8_tests.log
for i in 1 2; do
echo "== old"; VDS_LIB_PATH=$PWD/video_diffusion_speedrun_amd/libvds_hip_old.so python tools/bench_fp8_producers.py 2>&1 | grep -i "transpose\|quant"
echo "== new"; python tools/bench_fp8_producers.py 2>&1 | grep -i "transpose\|quant"
done
