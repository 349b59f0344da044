#!/usr/bin/env python3
"""Summarise the counter_collection CSVs written by tools/pmc_passes.sh: per kernel, mean counter value per dispatch."""
import csv
import glob
import sys
from collections import defaultdict

root = sys.argv[1]
filt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(f"{root}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if filt and filt not in name:
            continue
        acc[name[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:36s} n={len(v):4d} mean={sum(v) / len(v):.4g} min={min(v):.4g} max={max(v):.4g}")
