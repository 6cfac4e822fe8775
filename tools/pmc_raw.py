"""Diagnostic: mean per launch of every counter in a rocprofv3 --pmc CSV, by (kernel, grid).
    python3 tools/pmc_raw.py <dir> [kernel-name substring]"""
import collections, csv, glob, os, sys
src, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f, newline="")):
        if pat in row["Kernel_Name"]:
            acc[(row["Kernel_Name"].split("(")[0][-50:], row["Grid_Size"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, c in sorted(acc.items()):
    print(k)
    for n, v in sorted(c.items()):
        print(f"    {n:36s} n {len(v):5d}  mean {sum(v)/len(v):16.1f}")
