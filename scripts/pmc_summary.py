#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel, per counter mean over dispatches."""
import csv
import glob
import sys
from collections import defaultdict

root = sys.argv[1]
for f in sorted(glob.glob(root + "/*/runc/*counter_collection.csv")):
    acc = defaultdict(lambda: defaultdict(list))
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"].split("(")[0].replace("void fpic::", "")
        acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
    print("==", f.split("/")[-3])
    for k in sorted(acc):
        if not any(s in k for s in ("push", "cell_sums", "bin_scatter", "stamp", "bin_count", "sort_scatter")):
            continue
        parts = []
        for c, v in sorted(acc[k].items()):
            parts.append("%s mean %.4g (n=%d, min %.4g)" % (c, sum(v) / len(v), len(v), min(v)))
        print("  %-34s %s" % (k[:34], "; ".join(parts)))
