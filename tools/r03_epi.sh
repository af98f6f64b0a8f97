mkdir -p gpurun_out/r03b
python -m pytest tests/test_kernels_gpu.py tests/test_fp8_gpu.py -x -q -m gpu -k "gemm or linear or emit" > gpurun_out/r03b/epi_tests.log 2>&1
tail -5 gpurun_out/r03b/epi_tests.log
for i in 1 2; do
echo "== old"; VDS_LIB_PATH=$PWD/video_diffusion_speedrun_amd/libvds_hip_old.so python tools/bench_gemm_epi.py
echo "== new"; python tools/bench_gemm_epi.py
done > gpurun_out/r03b/epi_ab2.log 2>&1
cat gpurun_out/r03b/epi_ab2.log
