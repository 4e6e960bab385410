#!/usr/bin/env python3
"""Summarise rocprofv3 counter_collection.csv: mean counter value per kernel name."""
import csv, sys, collections, glob
for path in sys.argv[1:]:
    for f in glob.glob(path + "/**/*_counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:60]
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, d in acc.items():
            print(k)
            for c, v in sorted(d.items()):
                print(f"   {c:36s} n={len(v):4d} mean={sum(v)/len(v):.4g}")
