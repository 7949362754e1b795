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


def slam(fetch, write):
    """python tools/pmc_traffic.py FETCH_DIR WRITE_DIR slam -> profiles/pmc_traffic_slam.json (bench.py --workload config5)"""
    out = {}
    for pat, key in (("k_slam_gate_rows", "k_slam_gate_rows"), ("k_slam_tag(", "k_slam_tag"), ("k_slam_rows_forced", "k_slam_rows_forced")):
        fk = [k for k in fetch if pat in k]
        wk = [k for k in write if pat in k]
        if not fk or not wk:
            continue
        m = min(len(fetch[fk[0]]), len(write[wk[0]]))
        f = fetch[fk[0]][-m:]
        w = write[wk[0]][-m:]
        # the first launches of a run spawn the beacons: take the second half (steady deposits)
        f, w = f[m // 2:], w[m // 2:]
        rd = 2.0 * 1024.0 * sum(f) / len(f)
        wr = 1024.0 * sum(w) / len(w)
        out[key] = {"kernel": fk[0].split("(")[0], "launches": len(f), "read_bytes_per_launch": rd, "write_bytes_per_launch": wr,
                    "hbm_bytes_per_launch": rd + wr, "note": "FETCH_SIZE doubled (gfx950 wide-stream correction: float4 row loads)"}
    with open("profiles/pmc_traffic_slam.json", "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out, indent=1))


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    if len(sys.argv) > 3 and sys.argv[3] == "slam":
        return slam(fetch, write)
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 16000000
    out = {}

    def launches(pat):
        """(read bytes, written bytes) per launch of the kernel whose name contains pat; the two passes replay the
        same deterministic launch sequence, so launch k of one pass is launch k of the other"""
        fk = [k for k in fetch if pat in k]
        wk = [k for k in write if pat in k]
        if not fk or not wk:
            return None, []
        m = min(len(fetch[fk[0]]), len(write[wk[0]]))
        return fk[0].split("(")[0], [(2.0 * 1024.0 * fetch[fk[0]][i], 1024.0 * write[wk[0]][i]) for i in range(m)]

    def put(key, kernel, sel, note=""):
        if not sel:
            return
        rd = sum(r for r, _ in sel) / len(sel)
        wr = sum(w for _, w in sel) / len(sel)
        out[key] = {"kernel": kernel, "launches": len(sel), "read_bytes_per_launch": rd, "write_bytes_per_launch": wr,
                    "hbm_bytes_per_launch": rd + wr, "hbm_bytes_per_particle": (rd + wr) / n, "n": n,
                    "note": "FETCH_SIZE doubled (gfx950 wide-stream correction)" + note}

    # launches that exit at once (a resample launch after a normalise-only decision) read < 1 B per particle
    work = lambda ls: [x for x in ls if x[0] > 1.0 * n]
    k, ls = launches("k_loc_stream<true, true>")
    put("stream", k, work(ls))
    k, ls = launches("k_loc_stream<false, true>")
    put("stream_static", k, work(ls))
    k, ls = launches("k_loc_resample_dest<false")
    put("resample", k, work(ls))
    # k_loc_deposit is one kernel with two sides (deferred resampling + deposit, or streaming pass only; the previous
    # decision, on the device, picks).  Both sides write 20 B per particle with motion; the streaming side reads exactly
    # 20 B, the resampling side the source weights of ~3 tiles per destination tile on top of the gathered ancestors:
    # launches that read more than 22 B per particle are resampling launches.  Without motion the streaming side writes
    # only the weights (4 B).
    k, ls = launches("k_loc_deposit<true>")
    ls = work(ls)
    put("resample_stream", k, [x for x in ls if x[0] > 22.0 * n], "; resampling side of k_loc_deposit<motion> (source weights of the "
        "tiles a destination tile draws from + gathered ancestors read, 20 B written; the doubling of FETCH_SIZE is calibrated "
        "for float4 streams, the dword gathers are covered by it only roughly)")
    if "stream" not in out or out["stream"]["launches"] < 4:
        put("stream", k, [x for x in ls if x[0] <= 22.0 * n], "; streaming side of k_loc_deposit<motion>")
    k, ls = launches("k_loc_deposit<false>")
    ls = work(ls)
    put("resample_static", k, [x for x in ls if x[1] > 12.0 * n], "; resampling side of k_loc_deposit<static>")
    if "stream_static" not in out:
        put("stream_static", k, [x for x in ls if x[1] <= 12.0 * n], "; streaming side of k_loc_deposit<static>")
    # what the numbers were measured on: bench.py flags `traffic_stale` when the kernel sources have changed since
    import hashlib
    h = hashlib.sha256()
    for f in ("pf_common.h", "pf_loc_kernels.h"):
        with open(os.path.join("slam3d_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    out["_source"] = {"kernel_sources_sha": h.hexdigest()[:16], "command": " ".join(sys.argv)}
    os.makedirs("profiles", exist_ok=True)
    with open("profiles/pmc_traffic.json", "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
