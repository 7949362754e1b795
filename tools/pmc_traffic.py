#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; each its own run, as MI355X_MICROARCH.md
'HBM' prescribes) into HBM bytes per launch for the hot kernels -> profiles/pmc_traffic.json.

Units and corrections (guide, section HBM): FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts
128-B requests as 64 B for wide coalesced streaming reads, so the read side is DOUBLED; WRITE_SIZE is exact for
16-B-per-lane streaming stores.  The resample kernels read their source tile with the same float4-per-lane loads
as the streaming kernels, so the same doubling applies.

    python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write [N]
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def per_kernel(d, counter):
    f = max(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)    # latest run
    acc = defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 16000000
    out = {}
    names = {"stream": "k_loc_stream<true, true>", "stream_static": "k_loc_stream<false, true>", "resample": "k_loc_resample<false>",
             "resample_stream": "k_loc_resample_deposit<true>", "resample_static": "k_loc_resample_deposit<false>"}
    for key, pat in names.items():
        fk = [k for k in fetch if pat in k]
        wk = [k for k in write if pat in k]
        if not fk or not wk:
            continue
        fv, wv = fetch[fk[0]], write[wk[0]]
        # drop the launches that exit at once (a resample launch after a normalise-only decision; with deferred
        # resampling whichever of k_loc_resample_deposit / k_loc_stream had nothing to do); a class that never did
        # any work in this run is left out
        fmax, wmax = max(fv), max(wv)
        if 2.0 * 1024.0 * fmax < 1.0 * n:       # < 1 B read per particle: only empty launches
            continue
        fv = [v for v in fv if v > 0.25 * fmax]
        wv = [v for v in wv if v > 0.25 * wmax]
        rd = 2.0 * 1024.0 * sum(fv) / len(fv)
        wr = 1024.0 * sum(wv) / len(wv)
        out[key] = {"kernel": fk[0].split("(")[0], "launches_fetch_pass": len(fv), "launches_write_pass": len(wv),
                    "fetch_size_KiB_avg_raw": sum(fv) / len(fv), "write_size_KiB_avg": sum(wv) / len(wv),
                    "read_bytes_per_launch": rd, "write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr,
                    "hbm_bytes_per_particle": (rd + wr) / n, "n": n,
                    "note": "FETCH_SIZE doubled (gfx950 wide-stream correction)" +
                            ("; reads are float4 per lane like the streaming kernels'" if key.startswith("resample") else "")}
    os.makedirs("profiles", exist_ok=True)
    with open("profiles/pmc_traffic.json", "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
