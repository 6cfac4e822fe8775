#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): kernel-trace statistics and the hardware-counter passes behind bench.py's
# roofline line, every pass in its own rocprofv3 run (--kernel-trace + --pmc only; the program itself after `--`).
#   bash tools/pmc_collect.sh <tag>      ->  gpurun_out/<tag>/{stats,sq,sq2,tcc,fetch,write,calib}/...
# then  python3 tools/pmc_summary.py gpurun_out/<tag> profiles/<round>_pmc_fused_kernel.json
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
TAG=${1:-pmc}
OUT="$ROOT/gpurun_out/$TAG"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 20 --warmup 5"
LIGHT="python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-train --no-roofline --no-configs-table"
echo "== kernel trace + stats of the default bench command"
rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o bench --output-format csv -- $BENCH > "$OUT/stats.log" 2>&1
echo "== SQ pass 1 (MFMA instruction counters, wave cycles)"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU \
  -d "$OUT/sq" -o pmc --output-format csv -- $LIGHT > "$OUT/sq.log" 2>&1
echo "== SQ pass 2 (LDS / memory instruction mix)"
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT \
  -d "$OUT/sq2" -o pmc --output-format csv -- $LIGHT > "$OUT/sq2.log" 2>&1
echo "== TCC pass (L2 hit rate, fabric read requests)"
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum \
  -d "$OUT/tcc" -o pmc --output-format csv -- $LIGHT > "$OUT/tcc.log" 2>&1
echo "== FETCH_SIZE pass"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/fetch" -o pmc --output-format csv -- $LIGHT > "$OUT/fetch.log" 2>&1
echo "== WRITE_SIZE pass"
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/write" -o pmc --output-format csv -- $LIGHT > "$OUT/write.log" 2>&1
echo "== calibration of the MFMA counters on a kernel with a known instruction count"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES \
  -d "$OUT/calib" -o pmc --output-format csv -- $ROOT/tools/_diag/mfma_count_calib > "$OUT/calib.log" 2>&1
echo "done: $OUT"
