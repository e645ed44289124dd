"""Summarise the counter CSV of tools/pmc_gemm.sh: per kernel name, mean of each counter per dispatch."""
import csv
import glob
import sys
from collections import defaultdict

rows = defaultdict(lambda: defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in rows.items():
    if "gemm" not in k and "splitk" not in k:
        continue
    print(k)
    for name, v in sorted(c.items()):
        v = v[len(v) // 2:]          # second half of the dispatches: warmed up
        print(f"   {name:32s} {sum(v) / len(v):16.0f}  (n={len(v)})")
