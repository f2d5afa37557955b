#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc counter_collection CSVs per kernel (mean per dispatch)."""
import csv, glob, sys, collections
root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row["Kernel_Name"].split("(")[0][:48]
            agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in sorted(agg.items()):
    if not any(x in k for x in ("gemm1", "gemm2", "mid", "dots", "sum_blocks")):
        continue
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:32s} n={len(v):5d} mean={sum(v)/len(v):.4g}")
