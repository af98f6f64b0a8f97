#!/bin/bash
# same-box A/B of two library builds on the C5 step: tools/ab_c5_lib.sh <other .so> [env assignments for both]
O=gpurun_out/ab_lib; mkdir -p $O
for r in 1 2; do
for v in new other; do
  if [ $v = other ]; then export VDS_LIB_PATH=$PWD/$1; else unset VDS_LIB_PATH; fi
  python bench.py --workload c5 --steps 10 --warmup 4 --no-cpu-baseline 2>/dev/null | grep '^{' > $O/c5_${v}_$r.json
  python -c "
import json; d=json.load(open('$O/c5_${v}_$r.json')); k=d['kernel_breakdown_ms']; print('$v', round(d['value'],3), round(d['ms_per_step'],1), round(d['ms_per_step_median'],1), {n:v for n,v in k.items() if 'attn' in n})"
done; done
