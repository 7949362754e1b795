"""Where a config-5 step with B beacons spends its time: host call durations against the synchronised step (diagnostic)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from slam3d_amd import slam

F, B = int(sys.argv[1]), int(sys.argv[2])
K = 120
batch = slam.DeviceSlamBatch(F, 100, 1000, max_bcns=B, seed0=1000003, device=0)
for _ in range(B):
    batch.add_bcn_all()
truth, dfun = bench._slam_inputs(B, K + 64 + B)
ones = np.ones(F, dtype=np.float32)
th = [0.0, 0.0, 0.0]


def one(k, sync=False):
    t0 = time.perf_counter()
    a = (np.full(F, 1000.0 + 0.1 * k), ones * truth[k, 0], ones * truth[k, 1], ones * truth[k, 2])
    b = (np.full(F, k % B, dtype=np.int32), ones * dfun(k), ones * np.float32(0.1))
    t1 = time.perf_counter()
    batch.deposit_tag_vio_batch(*a)
    t2 = time.perf_counter()
    batch.deposit_range_batch(*b)
    t3 = time.perf_counter()
    if sync:
        batch.sync()
    th[0] += t1 - t0; th[1] += t2 - t1; th[2] += t3 - t2


for k in range(B + 30):
    one(k)
batch.sync()
th[:] = [0.0, 0.0, 0.0]
t0 = time.perf_counter()
for k in range(B + 30, B + 30 + K):
    one(k)
tq = time.perf_counter() - t0
batch.sync()
dt = time.perf_counter() - t0
print(f"F={F} B={B}: step {dt / K * 1e3:.4f} ms; host enqueue loop {tq / K * 1e3:.4f} ms (numpy {th[0] / K * 1e3:.4f}, "
      f"tag_vio {th[1] / K * 1e3:.4f}, range {th[2] / K * 1e3:.4f})", flush=True)
# each deposit synchronised: the GPU side alone
per = []
for k in range(B + 30 + K, B + 30 + K + 24):
    batch.sync()
    t0 = time.perf_counter()
    one(k, sync=True)
    per.append((time.perf_counter() - t0) * 1e3)
    c = batch.row_census()
    print(f"  k%B={k % B:2d} synchronised deposit {per[-1]:.3f} ms census {c}", flush=True)

# host-only: the VIO call alone with the GPU idle
batch.sync()
a = (np.full(F, 2000.0), ones * 1.0, ones * 2.0, ones * 3.0)
t0 = time.perf_counter()
for i in range(200):
    batch.deposit_tag_vio_batch(*a)
print(f"tag_vio alone, GPU idle: {(time.perf_counter() - t0) / 200 * 1e3:.4f} ms per call", flush=True)
import ctypes as C
L = batch._L
pt = [C.c_void_p(v.ctypes.data) for v in (a[0], a[1].astype(np.float32), a[2].astype(np.float32), a[3].astype(np.float32))]
keep = (a[1].astype(np.float32),)
xs = [np.ascontiguousarray(v, dtype=np.float32) for v in a[1:]]
pt = [C.c_void_p(a[0].ctypes.data)] + [C.c_void_p(v.ctypes.data) for v in xs]
t0 = time.perf_counter()
for i in range(200):
    L.pfx_slam_deposit_tag_vio_batch(batch._h, pt[0], pt[1], pt[2], pt[3], None)
print(f"raw C call alone: {(time.perf_counter() - t0) / 200 * 1e3:.4f} ms per call", flush=True)
print("affinity", len(os.sched_getaffinity(0)), "cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else "n/a", flush=True)
import threading
print("python threads", threading.active_count(), "os threads", len(os.listdir("/proc/self/task")), flush=True)
