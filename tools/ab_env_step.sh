#!/bin/bash
# same-box A/B of one environment knob on a bench workload: tools/ab_env_step.sh <workload> <VAR> <value A> <value B> [steps]
W=$1; VAR=$2; A=$3; B=$4; S=${5:-10}
O=gpurun_out/ab_env; mkdir -p $O
for r in 1 2; do
for v in $A $B; do
  env $VAR=$v python bench.py --workload $W --steps $S --warmup 4 --no-cpu-baseline --no-secondary 2>/dev/null | grep '^{' > $O/${W}_${VAR}_${v}_$r.json
  python -c "
import json; d=json.load(open('$O/${W}_${VAR}_${v}_$r.json')); k=d['kernel_breakdown_ms']; print('$VAR=$v', round(d['value'],3), round(d['ms_per_step'],1), round(d['ms_per_step_median'],1), {n:v for n,v in k.items() if 'attn' in n and 'plain' not in n})"
done; done
