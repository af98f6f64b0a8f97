#!/bin/bash
# one bench line per BASELINE workload on one box (copied to profiles/<round>/final_<workload>_bench.log)
O=gpurun_out/${1:-r06}final
mkdir -p $O
show() { python -c "
import json,sys
d=json.loads(open('$O/$1.log').read())
print('$1', round(d['value'],3), round(d['ms_per_step'],1), round(d.get('mfma_util_step',0),4), round(d.get('peak_hbm_gb',0),1))"; }
python bench.py --steps 20 --warmup 5 2>/dev/null | grep '^{' > $O/final_c3b_bench.log; show final_c3b_bench
python - <<EOF
import json
d=json.loads(open('$O/final_c3b_bench.log').read())
s=d['secondary']
print('  secondary c5', round(s['value'],3), round(s['ms_per_step'],1), 'roofline', round(d['roofline']['frac'],4), round(d['roofline']['frac_of_practical_ceiling'],4), 'c5 roofline', s['roofline']['kernel'], round(s['roofline']['frac'],4))
c=d['cpu_baseline']; print('  cpu', c['value'], c['c1_measured']['fp32'], c['c1_measured']['bf16'])
EOF
for w in c5 c2 c3a c4 c1; do
  python bench.py --workload $w --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | grep '^{' > $O/final_${w}_bench.log; show final_${w}_bench
done
python bench.py --workload c1 --graph --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | grep '^{' > $O/final_c1_graph_bench.log; show final_c1_graph_bench
python bench.py --force-shard-runtime --steps 6 --warmup 2 --no-cpu-baseline --no-secondary 2>/dev/null | grep '^{' > $O/final_c3b_shard_runtime_w1_bench.log; show final_c3b_shard_runtime_w1_bench
VDS_FSDP_RESHARD=1 python bench.py --force-shard-runtime --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --no-small-batch 2>/dev/null | grep '^{' > $O/c3b_shard_runtime_w1_reshard_bench.log; show c3b_shard_runtime_w1_reshard_bench
python bench.py --deterministic --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --no-small-batch 2>/dev/null | grep '^{' > $O/c3b_deterministic_bench.log; show c3b_deterministic_bench
python tools/bench_sampler.py > $O/sampler.log 2>&1; tail -3 $O/sampler.log
for b in 1 2 4; do
  python bench.py --batch $b --steps 8 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | grep '^{' > $O/c3b_b${b}_bench.log; show c3b_b${b}_bench
done
