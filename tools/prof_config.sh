#!/bin/bash
# Runs ON THE GPU BOX: rocprofv3 --kernel-trace --stats of one BASELINE config through bench.py (the program itself
# after `--`), at the per-GPU shard size.   bash tools/prof_config.sh <cfg> <tag> [extra bench args]
#   -> gpurun_out/<tag>/cfg<cfg>_kernel_stats.csv (+ the bench line in cfg<cfg>_bench.json)
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
CFG=$1; TAG=$2; shift 2
OUT="$ROOT/gpurun_out/$TAG"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$OUT/prof_cfg$CFG" -o bench --output-format csv -- \
  python3 "$ROOT/bench.py" --config "$CFG" --scaling weak --no-cpu-baseline "$@" > "$OUT/cfg${CFG}_bench.json" 2> "$OUT/cfg${CFG}_bench.err"
cp "$(find "$OUT/prof_cfg$CFG" -name '*kernel_stats.csv' | head -1)" "$OUT/cfg${CFG}_kernel_stats.csv"
echo "done: $OUT/cfg${CFG}_kernel_stats.csv"
