#!/bin/bash
# the round's closing evidence on ONE box: full GPU suite, one bench line per workload, rocprof passes (SOAK=1: + fp8 soak)
#   COMMIT=$(git rev-parse --short HEAD) bash tools/r05_refresh_all.sh      (COMMIT stamps profiles/r05_traffic.json)
mkdir -p gpurun_out/r05final
python -m pytest tests -q -m gpu -x 2>&1 | tail -5 > gpurun_out/r05final/gpu_suite.log; cat gpurun_out/r05final/gpu_suite.log
bash tools/r05_final_benches.sh 2>&1 | tee gpurun_out/r05final/final_benches_summary.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
bash tools/collect_profiles.sh r05 > gpurun_out/r05final/collect_profiles.log 2>&1; tail -5 gpurun_out/r05final/collect_profiles.log
if [ "${SOAK:-0}" = 1 ]; then
WORKLOAD=c3b B=4 STEPS=200 MODES=bf16,fp8x python tools/soak.py > gpurun_out/r05final/soak200.json 2> gpurun_out/r05final/soak200.err; tail -c 400 gpurun_out/r05final/soak200.json
fi
