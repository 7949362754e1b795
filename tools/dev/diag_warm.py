"""Is the N = 1 M step faster behind a longer pre-roll because of the GPU's clocks or because of the filter's state? (diagnostic)"""
import os, sys, time, gc
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from slam3d_amd import particlefilter as pfm

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
bx, truth, rng = bench.scene(6000)


def fresh():
    pf = pfm.DeviceParticleFilterLoc(n, seed=123456789, stream_id=0, device=0)
    pf.depositVio(999.9, float(truth[0, 0]), float(truth[0, 1]), float(truth[0, 2]), 0.0)
    pf.depositRange(float(bx[0, 0]), float(bx[0, 1]), float(bx[0, 2]), float(rng[0, 0]), 0.1)
    return pf


def timed(pf, k0, K=100):
    pf.sync()
    t0 = time.perf_counter()
    for k in range(k0, k0 + K):
        bench.step(pf, k, bx, truth, rng)
    pf.sync()
    return (time.perf_counter() - t0) / K * 1e3


gc.collect(); gc.disable()
for label, other_steps, pre in (("cold, pre-roll 400", 0, 400), ("other filter busy for 2000 steps first, pre-roll 400", 2000, 400),
                                ("cold, pre-roll 2000", 0, 2000), ("other filter busy first, pre-roll 2000", 2000, 2000)):
    a = fresh()
    b = fresh() if other_steps else None
    a.sync()
    time.sleep(0.08)                       # the GPU idles, as behind gc.collect()
    if b is not None:
        for k in range(other_steps):
            bench.step(b, k, bx, truth, rng)
    for k in range(pre):
        bench.step(a, k, bx, truth, rng)
    ms = [timed(a, pre + 100 * i) for i in range(4)]
    print(f"N={n} {label}: ms per step over four consecutive 100-step windows {[round(v, 4) for v in ms]}", flush=True)
    del a, b
