"""tools/pmc_summary.py <dir> — per-kernel averages from rocprofv3 csv outputs (counter_collection + kernel_trace)."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
for path in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
    acc = defaultdict(lambda: defaultdict(list))
    for row in csv.DictReader(open(path)):
        acc[row["Kernel_Name"].split("(")[0][:70]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    print("==", os.path.relpath(path, root))
    for k, cs in acc.items():
        for c, v in cs.items():
            print(f"  {k:70s} {c:24s} n={len(v):4d} mean={sum(v)/len(v):.6g}")
for path in sorted(glob.glob(os.path.join(root, "**", "*kernel_stats.csv"), recursive=True)):
    print("==", os.path.relpath(path, root))
    for row in csv.DictReader(open(path)):
        print("  ", {k: row[k] for k in row if k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage")})
