#!/usr/bin/env python3
"""Where the time of k_slam_gate_rows goes, by branch (BASELINE config 5: 512 tags x 4 beacons).

    rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/slam_branch_split.py run CENSUS.json
    python3 tools/slam_branch_split.py join DIR CENSUS.json > profiles/rNN_slam_branch_split.json

`run` drives the config-5 step of bench.py and asks for the row census after every deposit (pfx_slam_row_census: rows
of the measured beacon that only normalised / that resampled); `join` pairs deposit k with the k-th launch of
k_slam_gate_rows in the kernel trace and fits  duration = a + b * (fraction of rows that resampled)."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def run(path, F=512, B=4, K=64):
    import numpy as np
    import bench
    from slam3d_amd import slam
    batch = slam.DeviceSlamBatch(F, 100, 1000, max_bcns=B, seed0=1000003, device=0)
    for _ in range(B):
        batch.add_bcn_all()
    truth, dfun = bench._slam_inputs(B, K + B + 1)
    ones = np.ones(F, dtype=np.float32)
    census = []
    for k in range(B + K):
        batch.deposit_tag_vio_batch(np.full(F, 1000.0 + 0.1 * k), ones * truth[k, 0], ones * truth[k, 1], ones * truth[k, 2])
        batch.deposit_range_batch(np.full(F, k % B, dtype=np.int32), ones * dfun(k), ones * np.float32(0.1))
        if k >= B:
            census.append(batch.row_census())
    with open(path, "w") as fh:
        json.dump({"filters": F, "beacons": B, "census": census}, fh)


def join(d, path):
    f = max(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
    rows = [r for r in csv.DictReader(open(f)) if "k_slam_gate_rows" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3 for r in rows]      # us
    cen = json.load(open(path))["census"]
    dur = dur[-len(cen):]
    frac = [c[1] / max(1, c[0] + c[1]) for c in cen]
    n = len(frac)
    mx, my = sum(frac) / n, sum(dur) / n
    b = sum((x - mx) * (y - my) for x, y in zip(frac, dur)) / max(1e-30, sum((x - mx) ** 2 for x in frac))
    a = my - b * mx
    out = {"deposits": n, "fit_us": {"only_normalising_rows": a, "all_rows_resampling": a + b},
           "per_deposit": [{"rows_resampled_fraction": round(x, 4), "tag_resamplings": c[2], "k_slam_gate_rows_us": round(y, 1)}
                           for x, y, c in zip(frac, dur, cen)],
           "bytes_per_beacon_particle": {"normalising_row": 20, "resampling_row": 40}}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2])
    else:
        join(sys.argv[2], sys.argv[3])
