#!/usr/bin/env python3
"""Per-kernel averages of the counters of one rocprofv3 --pmc pass.
    python tools/sq_counters.py DIR KERNEL_SUBSTRING [min_value_filter_counter=value]
prints, for the launches of the kernels whose name contains the substring, the mean of every counter collected."""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    d, pat = sys.argv[1], sys.argv[2]
    f = max(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    acc = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        print(k)
        for c, v in sorted(cs.items()):
            v = sorted(v)
            print(f"  {c:28s} launches {len(v):4d} mean {sum(v)/len(v):16.1f} min {v[0]:16.1f} max {v[-1]:16.1f}")


main()
