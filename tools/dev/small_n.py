"""per-event host cost of a small Loc filter, the C++ caller (pfreplay) beside the Python wrapper.

    python tools/dev/small_n.py [N ...]        (on the GPU box; writes gpurun_out/small_n.txt)

Range-only replay = deposits enqueued back to back, one pose read at the end: the rate at which a host can feed the
filter.  Full replay = test.c's loop: a pose read (a stream synchronisation) after every VIO row.
"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
EXE = os.path.join(ROOT, "slam3d_amd", "bin", "pfreplay")
T1 = os.path.join(ROOT, "tests", "golden", "test1")


def cxx(n, range_only, repeat, warm):
    from slam3d_amd import replay as R
    cmd = [EXE, "loc", "--vio", os.path.join(T1, R.VIO_FILE), "--uwb", os.path.join(T1, R.UWB_FILE), "--deploy",
           os.path.join(T1, R.DEPLOY_FILE), "--out", "/tmp/small_n.csv", "--n", str(n), "--repeat", str(repeat), "--warm", str(warm)]
    if range_only:
        cmd.append("--range-only")
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    if p.returncode:
        raise SystemExit(p.stdout + p.stderr)
    line = [l for l in p.stdout.splitlines() if l.startswith("timing:")][0]
    return float(line.split("passes,")[1].split("us")[0])


def py(n, range_only, repeat, warm):
    from slam3d_amd import particlefilter as pfm, replay as R
    vio, uwb, dep = R.load_test1(T1)
    g = pfm.DeviceParticleFilterLoc(n, seed=123456789, stream_id=0)
    events = 0
    for rep in range(-warm, repeat):
        if rep == 0:
            g.sync()
            events = 0
            t0 = time.perf_counter()
        res = R.replay_loc(g, vio, uwb, dep, range_only=range_only)
        events += res.n_range + len(res.rows)
    g.getTagLoc()
    return 1e6 * (time.perf_counter() - t0) / max(1, events)


def main():
    ns = [int(a) for a in sys.argv[1:]] or [10000, 100000, 1000000]
    out = []
    for n in ns:
        for ro in (True, False):
            a = cxx(n, ro, 20, 3)
            b = py(n, ro, 5, 1)
            out.append(f"N = {n:>8}  {'range-only' if ro else 'test.c loop'}: C++ {a:8.2f} us per event, Python wrapper {b:8.2f} us per event")
            print(out[-1], flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "small_n.txt"), "w") as f:
        f.write("\n".join(out) + "\n")


if __name__ == "__main__":
    main()
