#!/bin/bash
# One command for the first run on a multi-GPU MI355X node: bench.py at N = 1, 2, 4, 8 x both reduce-scatter schedules x
# two all-gather windows, the collectives alone, one table (see tools/first_multigpu.py).
cd "$(dirname "$0")/.." && exec python3 tools/first_multigpu.py "$@"
