#!/usr/bin/env python3
"""Sum rocprofv3 --pmc csv rows (counter_collection.csv) per kernel name and counter.  python scripts/pmc_by_kernel.py DIR [substr]"""
import csv
import glob
import sys
from collections import defaultdict

d, sub = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
acc, n = defaultdict(lambda: defaultdict(float)), defaultdict(int)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if sub and sub not in k:
            continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[(k, r["Counter_Name"])] += 1
for k in sorted(acc, key=lambda k: -max(acc[k].values())):
    print(k[:110])
    for c, v in sorted(acc[k].items()):
        print(f"    {c:32s} {v:16.0f}  ({n[(k, c)]} rows)")
