"""Idle time between consecutive kernels of a rocprofv3 kernel trace (csv): python tools/dev/trace_gaps.py TRACE.csv [skip_first_n]"""
import collections, csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("pf::", "")[:34]) for r in rows)
seg = ev[len(ev) // 2:] if not skip else ev[skip:]
busy = sum(e[1] - e[0] for e in seg)
span = seg[-1][1] - seg[0][0]
print(f"{len(seg)} kernels, span {span / 1e3:.1f} us, busy {busy / 1e3:.1f} us, idle {100 * (span - busy) / span:.2f} %")
gaps = collections.defaultdict(list)
for p, q in zip(seg[:-1], seg[1:]):
    gaps[(p[2], q[2])].append((q[0] - p[1]) / 1e3)
for k, v in sorted(gaps.items(), key=lambda kv: -sum(kv[1])):
    print(f"  {k[0]:>34s} -> {k[1]:<34s} {len(v):5d} x  avg {sum(v) / len(v):6.2f} us  max {max(v):7.2f}  total {sum(v):8.1f}")
dur = collections.defaultdict(list)
for e in seg:
    dur[e[2]].append((e[1] - e[0]) / 1e3)
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    print(f"  {k:>34s} {len(v):5d} x  avg {sum(v) / len(v):7.2f} us")
