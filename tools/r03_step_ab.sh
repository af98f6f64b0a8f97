mkdir -p gpurun_out/r03b
for lib in old new nt; do
  p=$PWD/video_diffusion_speedrun_amd/libvds_hip_$lib.so
  [ $lib = new ] && p=$PWD/video_diffusion_speedrun_amd/libvds_hip.so
  VDS_LIB_PATH=$p python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r03b/step_c3b_$lib.log 2>&1
  VDS_LIB_PATH=$p python bench.py --workload c5 --steps 6 --warmup 3 --no-cpu-baseline > gpurun_out/r03b/step_c5_$lib.log 2>&1
done
for f in gpurun_out/r03b/step_c*; do echo $f; tail -1 $f | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('ms_per_step_median'))"; done
