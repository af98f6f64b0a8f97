#!/bin/bash
# Collect the rocprofv3 evidence behind DESIGN.md / bench.py's roofline object.  Run on the GPU box from the repo root:
#     bash tools/collect_profiles.sh r02        (writes gpurun_out/prof_r02/..., summaries are then copied into profiles/)
# Counter passes are separate runs (--pmc with --kernel-trace only), as MI355X_MICROARCH.md prescribes.
set -u
TAG=${1:-r06}
OUT=gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
# The interpreter's ELF itself must follow `--`: under rocprofv3 the profiler's preloaded library has initialised the GPU
# before the program starts, so a python3 that is a shim / wrapper script (pyenv, conda) would exec the real interpreter
# from a GPU-initialised process -- the hop that takes a box of this pool down.
PY=$(python3 -c 'import os, sys; print(os.path.realpath(sys.executable))')
if ! head -c 4 "$PY" | grep -q ELF; then
  echo "collect_profiles.sh: $PY is not an ELF executable; refusing to run it under rocprofv3" >&2
  exit 2
fi
# 1. per-kernel time of the bench command itself
rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o c3b --output-format csv -- $PY bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-small-batch > "$OUT/bench_under_rocprof.log" 2>&1
# 1b. the same command with 4 more steps: the difference of the two kernel_stats tables = launches per train step (what
#     is left of torch's own kernels -- fills, copies, RNG -- in the step, as opposed to model / optimizer construction)
rocprofv3 --kernel-trace --stats -d "$OUT/stats7" -o c3b7 --output-format csv -- $PY bench.py --steps 7 --warmup 1 --no-cpu-baseline --no-secondary --no-small-batch > "$OUT/bench7_under_rocprof.log" 2>&1
find "$OUT/stats7" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/c3b_7steps_kernel_stats.csv"
# 1c. (round 5) HBM-side bytes of EVERY kernel class of the step: the bench command itself under the two PMC passes
#     (FETCH_SIZE and WRITE_SIZE cannot share a pass), bf16 (c3b) and fp8 (c5) -> profiles/${TAG}_traffic.json, the file
#     bench.py's roofline.traffic reads (per-launch mean over the class's dispatches)
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d "$OUT/step_c3b/$c" -o x --output-format csv -- $PY bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary --no-small-batch --no-prof > "$OUT/step_c3b_$c.log" 2>&1
  rocprofv3 --kernel-trace --pmc $c -d "$OUT/step_c5/$c" -o x --output-format csv -- $PY bench.py --workload c5 --steps 1 --warmup 3 --no-cpu-baseline --no-prof > "$OUT/step_c5_$c.log" 2>&1
  # (round 6) the batch SURVEY 8(d) names: bench.py's `small_batch` leg has its own roofline object
  rocprofv3 --kernel-trace --pmc $c -d "$OUT/step_c3b_b2/$c" -o x --output-format csv -- $PY bench.py --batch 2 --steps 1 --warmup 1 --no-cpu-baseline --no-secondary --no-small-batch --no-prof > "$OUT/step_c3b_b2_$c.log" 2>&1
done
$PY tools/pmc_class_traffic.py "$(git rev-parse --short HEAD 2>/dev/null || echo ${COMMIT:-unknown})" "c3b:$OUT/step_c3b" "c5:$OUT/step_c5" "c3b_b2:$OUT/step_c3b_b2:2" > "$OUT/${TAG}_traffic.json" 2> "$OUT/traffic_json.err"
# 2. HBM-side bytes of the attention kernels at the bench shape (B = 12): FETCH_SIZE and WRITE_SIZE cannot share a pass
for c in FETCH_SIZE WRITE_SIZE; do
  B=12 REPS=1 rocprofv3 --kernel-trace --pmc $c -d "$OUT/pmc_$c" -o x --output-format csv -- $PY tools/prof_attn.py > "$OUT/pmc_$c.log" 2>&1
done
# 2b. the same two passes over the linear-layer GEMMs at the DiT-XL shapes (B = 12)
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d "$OUT/pmcg_$c" -o x --output-format csv -- $PY tools/prof_gemm.py > "$OUT/pmcg_$c.log" 2>&1
done
# 3. SQ counters of the attention kernels (B = 6) and of the GEMMs (B = 12), 4 counters per pass
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_LDS" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU"; do
  i=$((i + 1))
  B=6 REPS=2 rocprofv3 --kernel-trace --pmc $set -d "$OUT/sq_attn_$i" -o x --output-format csv -- $PY tools/prof_attn.py > "$OUT/sq_attn_$i.log" 2>&1
  rocprofv3 --kernel-trace --pmc $set -d "$OUT/sq_gemm_$i" -o x --output-format csv -- $PY tools/prof_gemm.py > "$OUT/sq_gemm_$i.log" 2>&1
done
# 4. (round 3) the fp8 configuration: kernel stats of the C5 bench, HBM bytes (B = 12) and SQ counters (B = 6) of the
#    fp8 attention kernels
rocprofv3 --kernel-trace --stats -d "$OUT/stats_c5" -o c5 --output-format csv -- $PY bench.py --workload c5 --steps 3 --warmup 3 --no-cpu-baseline > "$OUT/c5_bench_under_rocprof.log" 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  B=12 REPS=1 rocprofv3 --kernel-trace --pmc $c -d "$OUT/pmc8_$c" -o x --output-format csv -- $PY tools/prof_attn_fp8.py > "$OUT/pmc8_$c.log" 2>&1
done
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_LDS" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"; do
  i=$((i + 1))
  B=6 REPS=2 rocprofv3 --kernel-trace --pmc $set -d "$OUT/sq_attn8_$i" -o x --output-format csv -- $PY tools/prof_attn_fp8.py > "$OUT/sq_attn8_$i.log" 2>&1
done
$PY tools/pmc_summary.py $(find "$OUT" -path "*pmc8_*" -name "*counter_collection.csv") > "$OUT/attn_fp8_hbm_traffic_pmc.txt" 2>&1
$PY tools/pmc_summary.py $(find "$OUT" -path "*sq_attn8_*" -name "*counter_collection.csv") > "$OUT/attn_fp8_sq_counters.txt" 2>&1
find "$OUT/stats_c5" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/c5_kernel_stats.csv"
$PY tools/pmc_summary.py $(find "$OUT" -path "*pmc_*" -name "*counter_collection.csv") > "$OUT/attn_hbm_traffic_pmc.txt" 2>&1
$PY tools/pmc_summary.py $(find "$OUT" -path "*pmcg_*" -name "*counter_collection.csv") > "$OUT/gemm_hbm_traffic_pmc.txt" 2>&1
$PY tools/pmc_summary.py $(find "$OUT" -path "*sq_attn_*" -name "*counter_collection.csv") > "$OUT/attn_sq_counters.txt" 2>&1
$PY tools/pmc_summary.py $(find "$OUT" -path "*sq_gemm_*" -name "*counter_collection.csv") > "$OUT/gemm_sq_counters.txt" 2>&1
find "$OUT/stats" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/c3b_kernel_stats.csv"
tail -c 600 "$OUT/bench_under_rocprof.log"; echo; cat "$OUT/attn_hbm_traffic_pmc.txt" | head -20
