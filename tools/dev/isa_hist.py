#!/usr/bin/env python3
"""Histogram of the instructions of one kernel in a hipcc -S listing (tools/dev: `make -C slam3d_amd/csrc asm` writes
/tmp/pf_loc.s).  usage: isa_hist.py LISTING KERNEL_SUBSTRING [--blocks]
--blocks: per basic block (label), with instruction counts, to find the hot loop."""
import collections
import re
import sys


def main():
    path, key = sys.argv[1], sys.argv[2]
    blocks = "--blocks" in sys.argv
    lines = open(path).read().split("\n")
    start = None
    for i, l in enumerate(lines):
        if re.match(r"^_Z\w*:", l) and key in l:
            start = i
            break
    if start is None:
        raise SystemExit("kernel not found")
    end = start
    while not lines[end].startswith("\t.section") and ".Lfunc_end" not in lines[end]:
        end += 1
    hist = collections.Counter()
    cur, per = "entry", collections.OrderedDict()
    for l in lines[start + 1:end]:
        s = l.strip()
        if not s or s.startswith(";") or s.startswith("."):
            if re.match(r"^\.LBB\d+_\d+:", s):
                cur = s.split(":")[0]
            continue
        m = s.split()[0]
        hist[m] += 1
        per.setdefault(cur, collections.Counter())[m] += 1
    tot = sum(hist.values())
    print(lines[start][:100], "total", tot)
    if blocks:
        for b, h in per.items():
            n = sum(h.values())
            if n >= 40:
                v = sum(c for k, c in h.items() if k.startswith("v_"))
                print(f"{b}: {n} instr, {v} VALU; top:", ", ".join(f"{k}={c}" for k, c in h.most_common(14)))
    else:
        for k, c in hist.most_common(60):
            print(f"{c:6d} {k}")


main()
