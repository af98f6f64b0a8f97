#!/bin/bash
# same-box A/B of two library builds on the C3b (bf16 headline) step: tools/ab_c3b_lib.sh <other .so> [steps]
O=gpurun_out/ab_lib3; mkdir -p $O; S=${2:-8}
for r in 1 2; do
for v in new other; do
  if [ $v = other ]; then export VDS_LIB_PATH=$PWD/$1; else unset VDS_LIB_PATH; fi
  python bench.py --steps $S --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | grep '^{' > $O/c3b_${v}_$r.json
  python -c "
import json,sys; d=json.load(open('$O/c3b_${v}_$r.json')); k=d['kernel_breakdown_ms']; print('$v', round(d['value'],3), round(d['ms_per_step'],1), round(d['ms_per_step_median'],1), {n:v for n,v in k.items() if (sys.argv[1] in n) or n=='_sum'})" "${3:-attn_}"
done; done
