#!/bin/bash
# one bench line per BASELINE workload on one box (profiles/r03/final_<workload>_bench.log)
mkdir -p gpurun_out/r03final
for w in c3b c5 c2 c3a c4 c1; do
  python bench.py --workload $w --steps 8 --warmup 3 $( [ $w = c3b ] || echo --no-cpu-baseline ) 2>/dev/null | grep '^{' > gpurun_out/r03final/final_${w}_bench.log
  python -c "
import json,sys
d=json.loads(open('gpurun_out/r03final/final_${w}_bench.log').read())
print('$w', round(d['value'],3), round(d['ms_per_step'],1), round(d['mfma_util_step'],4), round(d['peak_hbm_gb'],1))"
done
python bench.py --workload c1 --graph --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | grep '^{' > gpurun_out/r03final/final_c1_graph_bench.log
python bench.py --workload c5 --no-fp8-attention --steps 6 --warmup 3 --no-cpu-baseline 2>/dev/null | grep '^{' > gpurun_out/r03final/c5_without_fp8_attention_bench.log
python bench.py --workload c5 --no-fp8-cross-attention --steps 6 --warmup 3 --no-cpu-baseline 2>/dev/null | grep "^{" > gpurun_out/r03final/c5_without_fp8_cross_attention_bench.log
python bench.py --force-shard-runtime --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | grep '^{' > gpurun_out/r03final/final_c3b_shard_runtime_w1_bench.log
python tools/bench_sampler.py > gpurun_out/r03final/sampler.log 2>&1; tail -3 gpurun_out/r03final/sampler.log
for f in final_c1_graph_bench c5_without_fp8_attention_bench c5_without_fp8_cross_attention_bench final_c3b_shard_runtime_w1_bench; do python -c "
import json
d=json.loads(open('gpurun_out/r03final/$f.log').read()); print('$f', round(d['value'],3), round(d['ms_per_step'],1))"; done
for b in 1 2 4; do
  python bench.py --batch $b --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | grep '^{' > gpurun_out/r03final/c3b_b${b}_bench.log
  python -c "
import json
d=json.loads(open('gpurun_out/r03final/c3b_b${b}_bench.log').read()); print('c3b B=$b', round(d['value'],3), round(d['ms_per_step'],1), round(d['mfma_util_step'],4))"
done
