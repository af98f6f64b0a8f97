#!/bin/bash
# one bench line per BASELINE workload on one box (copied to profiles/r05/final_<workload>_bench.log)
O=gpurun_out/r05final
mkdir -p $O
show() { python -c "
import json,sys
d=json.loads(open('$O/$1.log').read())
print('$1', round(d['value'],3), round(d['ms_per_step'],1), round(d.get('mfma_util_step',0),4), round(d.get('peak_hbm_gb',0),1))"; }
python bench.py --steps 20 --warmup 5 2>/dev/null | grep '^{' > $O/final_c3b_bench.log; show final_c3b_bench
python - <<PYEOF
import json
d=json.loads(open('$O/final_c3b_bench.log').read())
s=d['secondary']; sb=d['small_batch']
print('  secondary c5', round(s['value'],3), round(s['ms_per_step'],1), 'roofline', round(d['roofline']['frac'],4), round(d['roofline']['frac_of_practical_ceiling'],4), 'c5 roofline', s['roofline']['kernel'], round(s['roofline']['frac'],4), 'traffic', s['roofline']['traffic'])
print('  small_batch B=2', round(sb['value'],3), round(sb['ms_per_step'],1), round(sb['mfma_util_step'],4), {k: sb['kernel_rates'][k]['frac_of_peak'] for k in ('gemm_nt','gemm_nn','gemm_tn','attn_bwd_dkv_plain')})
c=d['cpu_baseline']; print('  cpu', c['value'], c['c1_measured']['fp32'], c['c1_measured']['bf16'], c['c2_measured'])
PYEOF
for w in c5 c2 c3a c4 c1; do
  python bench.py --workload $w --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | grep '^{' > $O/final_${w}_bench.log; show final_${w}_bench
done
python bench.py --workload c1 --graph --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | grep '^{' > $O/final_c1_graph_bench.log; show final_c1_graph_bench
python bench.py --force-shard-runtime --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --no-small-batch 2>/dev/null | grep '^{' > $O/final_c3b_shard_runtime_w1_bench.log; show final_c3b_shard_runtime_w1_bench
VDS_ADALN_BATCH=0 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --no-small-batch 2>/dev/null | grep '^{' > $O/c3b_adaln_per_block_bench.log; show c3b_adaln_per_block_bench
python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --no-small-batch 2>/dev/null | grep '^{' > $O/c3b_plain_same_flags_bench.log; show c3b_plain_same_flags_bench
python tools/bench_sampler.py > $O/sampler.log 2>&1; tail -3 $O/sampler.log
for b in 1 2 4; do
  python bench.py --batch $b --steps 8 --warmup 3 --no-cpu-baseline --no-secondary --no-small-batch 2>/dev/null | grep '^{' > $O/c3b_b${b}_bench.log; show c3b_b${b}_bench
done
