#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter_collection.csv per counter over the fig_* kernels."""
import csv, sys, glob, collections
tot = collections.defaultdict(float)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "fig_" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in sorted(tot.items()):
    print(f"{k:28s} {v:.4g}")
