#!/bin/bash
# Sample socket power / shader clock (rocm-smi, read-only) while a kernel loop runs: evidence for DESIGN.md §7g.
#   bash tools/power_probe.sh <tag> <command...>
TAG=$1; shift
OUT=gpurun_out/power_$TAG.log
"$@" > gpurun_out/power_${TAG}_cmd.log 2>&1 &
PID=$!
sleep 6   # import + warm-up
for i in $(seq 1 12); do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Socket Graphics Package Power|sclk|Average Graphics Package Power|fclk|mclk" | tr -s ' ' | tr '\n' ';' >> $OUT
  echo >> $OUT
  sleep 0.5
done
wait $PID
tail -3 gpurun_out/power_${TAG}_cmd.log
echo "--- $TAG"; cat $OUT | head -14
