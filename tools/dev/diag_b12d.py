"""The wrapper's calls with and without the garbage collector / fresh arrays (diagnostic)."""
import os, sys, time, gc
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from slam3d_amd import slam

F, B = int(sys.argv[1]), int(sys.argv[2])
mode = sys.argv[3]
K = 120
batch = slam.DeviceSlamBatch(F, 100, 1000, max_bcns=B, seed0=1000003, device=0)
for _ in range(B):
    batch.add_bcn_all()
truth, dfun = bench._slam_inputs(B, K + 64 + B)
ones = np.ones(F, dtype=np.float32)
if mode == "nogc":
    gc.disable()
if mode == "freeze":
    gc.collect(); gc.freeze()
pa = (np.full(F, 1000.0), ones * 1, ones * 2, ones * 3)
pb = (np.zeros(F, dtype=np.int32), ones * 3, ones * np.float32(0.1))
th = np.zeros(3)


def one(k):
    t0 = time.perf_counter()
    if mode == "persist":
        pa[0][:] = 1000.0 + 0.1 * k; pa[1][:] = truth[k, 0]; pa[2][:] = truth[k, 1]; pa[3][:] = truth[k, 2]
        pb[0][:] = k % B; pb[1][:] = dfun(k)
        a, b = pa, pb
    else:
        a = (np.full(F, 1000.0 + 0.1 * k), ones * truth[k, 0], ones * truth[k, 1], ones * truth[k, 2])
        b = (np.full(F, k % B, dtype=np.int32), ones * dfun(k), ones * np.float32(0.1))
    t1 = time.perf_counter()
    batch.deposit_tag_vio_batch(*a)
    t2 = time.perf_counter()
    batch.deposit_range_batch(*b)
    t3 = time.perf_counter()
    th[:] += (t1 - t0, t2 - t1, t3 - t2)


for k in range(B + 30):
    one(k)
batch.sync()
th[:] = 0
g0 = [s["collections"] for s in gc.get_stats()]
t0 = time.perf_counter()
for k in range(B + 30, B + 30 + K):
    one(k)
tq = time.perf_counter() - t0
batch.sync()
dt = time.perf_counter() - t0
g1 = [s["collections"] for s in gc.get_stats()]
print(f"F={F} B={B} mode={mode}: step {dt / K * 1e3:.4f} ms; loop {tq / K * 1e3:.4f}: numpy {th[0] / K * 1e3:.4f} "
      f"vio {th[1] / K * 1e3:.4f} range {th[2] / K * 1e3:.4f}; gc collections {[b - a for a, b in zip(g0, g1)]}", flush=True)
