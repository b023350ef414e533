#!/usr/bin/env python
"""Summarise rocprofv3 --pmc CSV output: per kernel name, mean of each counter per dispatch.
usage: python scripts/pmc_summary.py <dir> [name-substring]"""
import csv, glob, sys, collections
d = sys.argv[1]; sub = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if sub and sub not in k: continue
        acc[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:36s} n={len(v):4d} mean={sum(v)/len(v):16.1f}")
