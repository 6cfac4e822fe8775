"""Per-kernel table of the SQ counters collected by tools/pmc_kernel.sh:
    python3 tools/pmc_table.py gpurun_out/<tag> [kernel-name substring ...]
Quad-cycle counters (SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_*) are shown as shares of the wave cycles:
  valu = time a wave spends issuing VALU instructions, wait = parked on s_waitcnt / barriers (memory latency),
  stall = issue stalls (pipe busy / dependencies).  A stencil kernel that is VALU-bound shows valu >> wait."""
import csv
import glob
import os
import sys


def main():
    src, pats = sys.argv[1], sys.argv[2:]
    acc = {}
    for f in glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                name = row["Kernel_Name"]
                if pats and not any(p in name for p in pats):
                    continue
                key = (name.split("(")[0][-60:], row["Grid_Size"])
                acc.setdefault(key, {}).setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    trace = {}
    for f in glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                name = row["Kernel_Name"]
                if pats and not any(p in name for p in pats):
                    continue
                grid = row.get("Grid_Size") or str(int(row["Grid_Size_X"]) * int(row.get("Grid_Size_Y", 1) or 1)
                                                   * int(row.get("Grid_Size_Z", 1) or 1))
                key = (name.split("(")[0][-60:], grid)
                trace.setdefault(key, []).append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
    print(f"{'kernel':60s} {'grid':>9s} {'n':>4s} {'us':>9s} {'valu':>6s} {'wait':>6s} {'stall':>6s} {'lds-stall':>9s} "
          f"{'VALU insts/wave-launch':>22s}")
    for key in sorted(acc):
        c = {k: sum(v) / len(v) for k, v in acc[key].items()}
        wc = c.get("SQ_WAVE_CYCLES", 0.0) or 1.0
        us = trace.get(key, [0.0])
        us = sorted(us)[len(us) // 2]
        print(f"{key[0]:60s} {key[1]:>9s} {len(next(iter(acc[key].values()))):4d} {us:9.1f} "
              f"{c.get('SQ_ACTIVE_INST_VALU', 0) / wc:6.2f} {c.get('SQ_WAIT_ANY', 0) / wc:6.2f} "
              f"{c.get('SQ_WAIT_INST_ANY', 0) / wc:6.2f} {c.get('SQ_WAIT_INST_LDS', 0) / wc:9.3f} "
              f"{c.get('SQ_INSTS_VALU', 0):22.3e}")


if __name__ == "__main__":
    main()
