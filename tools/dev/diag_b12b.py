"""Which call of a config-5 step waits, and for what (diagnostic)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from slam3d_amd import slam

F, B = int(sys.argv[1]), int(sys.argv[2])
mode = sys.argv[3] if len(sys.argv) > 3 else "twice"
K = 120
batch = slam.DeviceSlamBatch(F, 100, 1000, max_bcns=B, seed0=1000003, device=0)
for _ in range(B):
    batch.add_bcn_all()
truth, dfun = bench._slam_inputs(B, K + 64 + B)
ones = np.ones(F, dtype=np.float32)
th = np.zeros(6)


def one(k):
    t0 = time.perf_counter()
    a = (np.full(F, 1000.0 + 0.1 * k), ones * truth[k, 0], ones * truth[k, 1], ones * truth[k, 2])
    b = (np.full(F, k % B, dtype=np.int32), ones * dfun(k), ones * np.float32(0.1))
    t1 = time.perf_counter()
    if mode == "spin":
        while time.perf_counter() - t1 < 100e-6:
            pass
    t1b = time.perf_counter()
    batch.deposit_tag_vio_batch(*a)
    t2 = time.perf_counter()
    if mode == "twice":
        batch.deposit_tag_vio_batch(*a)
    t3 = time.perf_counter()
    batch.deposit_range_batch(*b)
    t4 = time.perf_counter()
    th[:5] += (t1 - t0, t1b - t1, t2 - t1b, t3 - t2, t4 - t3)


for k in range(B + 30):
    one(k)
batch.sync()
th[:] = 0
t0 = time.perf_counter()
for k in range(B + 30, B + 30 + K):
    one(k)
tq = time.perf_counter() - t0
batch.sync()
dt = time.perf_counter() - t0
print(f"F={F} B={B} mode={mode}: step {dt / K * 1e3:.4f} ms; loop {tq / K * 1e3:.4f}: numpy {th[0] / K * 1e3:.4f} spin {th[1] / K * 1e3:.4f} "
      f"vio#1 {th[2] / K * 1e3:.4f} vio#2 {th[3] / K * 1e3:.4f} range {th[4] / K * 1e3:.4f}", flush=True)
