"""Diagnostic: per-(kernel, grid) launch durations from a rocprofv3 --kernel-trace CSV.
    python3 tools/kernel_trace_by_grid.py <dir with *kernel_trace.csv> [substring of the kernel name]"""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if pat in r["Kernel_Name"]:
        grid = tuple(r.get(k) for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z") if k in r) or (r.get("Grid_Size"),)
        d[(r["Kernel_Name"][:60], grid)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(d.items()):
    v.sort()
    print(f"{k[0]:60s} grid {k[1]}  n {len(v):5d}  median {v[len(v)//2]/1e3:8.2f} us  min {v[0]/1e3:8.2f}  mean {sum(v)/len(v)/1e3:8.2f}")
