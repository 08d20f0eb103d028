#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs: mean counter value per kernel name (per dispatch)."""
import csv, sys, collections, glob, os
d = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(os.path.join(d, "*counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in agg.items():
    if not any(s in k for s in (sys.argv[2:] or ["scan_kernel", "select_kernel", "refine_list"])):
        continue
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:32s} n={len(v):3d} mean={sum(v)/len(v):.6g}")
