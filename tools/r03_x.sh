python -m pytest tests/test_kernels_gpu.py tests/test_fp8_gpu.py -x -q -m gpu -k "gemm or linear or emit" 2>&1 | tail -2
for i in 1 2; do
echo "== base"; VDS_LIB_PATH=$PWD/video_diffusion_speedrun_amd/libvds_hip_old.so python tools/bench_gemm_epi.py 2>&1 | grep -v amdgpu
echo "== pipelined halves"; python tools/bench_gemm_epi.py 2>&1 | grep -v amdgpu
done
