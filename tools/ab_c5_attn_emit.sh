#!/bin/bash
# same-box A/B of the fp8 emission from the attention epilogues (C5 step)
O=gpurun_out/ab_emit; mkdir -p $O
for r in 1 2; do
for v in 1 0; do
  VDS_FP8_ATTN_EMIT=$v python bench.py --workload c5 --steps 10 --warmup 4 --no-cpu-baseline 2>/dev/null | grep '^{' > $O/c5_emit${v}_$r.json
  python -c "
import json; d=json.load(open('$O/c5_emit${v}_$r.json')); k=d['kernel_breakdown_ms']; print('emit=$v', round(d['value'],3), round(d['ms_per_step'],1), round(d['ms_per_step_median'],1), 'fp8_quant', k.get('fp8_quant'), 'attn', {n:v for n,v in k.items() if 'attn' in n})"
done; done
