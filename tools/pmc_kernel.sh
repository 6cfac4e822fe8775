#!/bin/bash
# Runs ON THE GPU BOX: SQ counters of every kernel a command launches, one rocprofv3 pass (--kernel-trace + --pmc).
#   bash tools/pmc_kernel.sh <tag> <python script and args...>      -> gpurun_out/<tag>/pmc_counter_collection.csv
# then tools/pmc_table.py gpurun_out/<tag> <kernel-name substring> prints per-launch means and derived shares.
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
TAG=$1; shift
OUT="$ROOT/gpurun_out/$TAG"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS \
  -d "$OUT" -o pmc --output-format csv -- python3 "$ROOT/$1" "${@:2}" > "$OUT/run.log" 2>&1
echo "done: $OUT"
