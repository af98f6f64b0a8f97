#!/bin/bash
# SQ / LDS counters of the fp8 attention kernels (separate --pmc passes with --kernel-trace only)
set -u
OUT=gpurun_out/r03/prof_fp8_$1
mkdir -p "$OUT"
export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_LDS" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"; do
  i=$((i + 1))
  B=6 REPS=2 rocprofv3 --kernel-trace --pmc $set -d "$OUT/sq_$i" -o x --output-format csv -- python3 tools/prof_attn_fp8.py > "$OUT/sq_$i.log" 2>&1
done
python3 tools/pmc_summary.py $(find "$OUT" -name "*counter_collection.csv") > "$OUT/../fp8_attn_sq_counters_$1.txt" 2>&1
cat "$OUT/../fp8_attn_sq_counters_$1.txt"
