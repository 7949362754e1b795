#!/usr/bin/env python3
"""Per-kernel averages of the counters of one rocprofv3 --pmc pass.
    python tools/sq_counters.py DIR KERNEL_SUBSTRING [COUNTER>VALUE | COUNTER<VALUE]
prints, for the launches of the kernels whose name contains the substring, the mean of every counter collected.  The
optional condition keeps only the launches whose COUNTER satisfies it (k_loc_deposit has two sides: the resampling
launches are the ones with the many vector instructions)."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def main():
    d, pat = sys.argv[1], sys.argv[2]
    cond = re.match(r"(\w+)([<>])([\d.eE+]+)$", sys.argv[3]) if len(sys.argv) > 3 else None
    f = max(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    per = defaultdict(dict)                     # dispatch -> counter -> value
    name = {}
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            per[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
            name[r["Dispatch_Id"]] = r["Kernel_Name"].split("(")[0]
    acc = defaultdict(lambda: defaultdict(list))
    for disp, cs in per.items():
        if cond:
            v = cs.get(cond.group(1))
            if v is None or not ((v > float(cond.group(3))) if cond.group(2) == ">" else (v < float(cond.group(3)))):
                continue
        for c, v in cs.items():
            acc[name[disp]][c].append(v)
    for k, cs in acc.items():
        print(k + (f"   [launches with {sys.argv[3]}]" if cond else ""))
        for c, v in sorted(cs.items()):
            v = sorted(v)
            print(f"  {c:28s} launches {len(v):4d} mean {sum(v)/len(v):16.1f} min {v[0]:16.1f} max {v[-1]:16.1f}")


main()
