"""Which part of the first library call after a range deposit waits (diagnostic)."""
import os, sys, time
import ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if len(sys.argv) > 4 and sys.argv[4] == "notorch":
    os.environ["PFX_PRELOAD_TORCH"] = "0"
from slam3d_amd import slam, _lib

F, B = int(sys.argv[1]), int(sys.argv[2])
mode = sys.argv[3]
K = 120
batch = slam.DeviceSlamBatch(F, 100, 1000, max_bcns=B, seed0=1000003, device=0)
for _ in range(B):
    batch.add_bcn_all()
L = batch._L
P = C.PyDLL(_lib.LIB_PATH)
P.pfx_slam_deposit_tag_vio_batch.argtypes = [C.c_void_p] * 6
P.pfx_slam_deposit_tag_vio_batch.restype = C.c_int
P.pfx_version.restype = C.c_char_p
L.pfx_version.restype = C.c_char_p
t = np.zeros(F); x = np.ones(F, dtype=np.float32); y = x.copy(); z = x.copy()
slot = np.zeros(F, dtype=np.int32); r = np.ones(F, dtype=np.float32); sd = np.full(F, 0.1, dtype=np.float32)
pt, px, py, pz, ps, pr, psd = [C.c_void_p(v.ctypes.data) for v in (t, x, y, z, slot, r, sd)]
h = batch._h
th = np.zeros(4)


def one(k):
    a = 0.025 * k
    t[:] = 1000.0 + 0.1 * k; x[:] = 2 * np.cos(a); y[:] = 2 * np.sin(a); z[:] = 1.5; slot[:] = k % B
    bb = ((5, 5, .5), (-5, 5, 3), (5, -5, 3), (-5, -5, .5), (5, 5, 3), (-5, 5, .5), (5, -5, .5), (-5, -5, 3))[(k % B) % 8]
    r[:] = np.sqrt((x[0] - bb[0]) ** 2 + (y[0] - bb[1]) ** 2 + (z[0] - bb[2]) ** 2)
    t0 = time.perf_counter()
    if mode == "sleep":
        time.sleep(0.0003)
    elif mode == "version":
        L.pfx_version()                 # a call that touches nothing, GIL released around it
    elif mode == "pyversion":
        P.pfx_version()                 # the same without releasing the GIL
    t1 = time.perf_counter()
    if mode == "pydll":
        P.pfx_slam_deposit_tag_vio_batch(h, pt, px, py, pz, None)
    else:
        L.pfx_slam_deposit_tag_vio_batch(h, pt, px, py, pz, None)
    t2 = time.perf_counter()
    L.pfx_slam_deposit_range_batch(h, ps, pr, psd)
    t3 = time.perf_counter()
    th[:3] += (t1 - t0, t2 - t1, t3 - t2)


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
print(f"F={F} B={B} mode={mode} {sys.argv[4:]}: step {dt / K * 1e3:.4f} ms; loop {tq / K * 1e3:.4f}: pre {th[0] / K * 1e3:.4f} "
      f"vio {th[1] / K * 1e3:.4f} range {th[2] / K * 1e3:.4f}", flush=True)
