"""per-launch duration of the O(N) kernels over the first steps of the bench scene (N = 16 M): is the slow start of a
short run the data (unconverged filter: respawns, concentrated weights) or the clock?"""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from slam3d_amd import particlefilter as pfm
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16000000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 80
bx, truth, rng = bench.scene(K + 1)
pf = pfm.DeviceParticleFilterLoc(n, seed=123456789, stream_id=0)
pf.depositVio(999.9, float(truth[0, 0]), float(truth[0, 1]), float(truth[0, 2]), 0.0)
pf.depositRange(float(bx[0, 0]), float(bx[0, 1]), float(bx[0, 2]), float(rng[0, 0]), 0.1)
pf.profile_only("stream", "stream_static", "resample", "deposit", "deposit_static")
pf.profile_stride(1)
pf.profile_begin(4 * K + 16)
stats = []
for k in range(K):
    bench.step(pf, k, bx, truth, rng)
kinds, ms = pf.profile_records()
log = pf.branch_log()[0]
names = pfm.DeviceParticleFilterLoc.KINDS
print("decisions:", "".join(str(int(b)) for b in log))
for i, (kd, t) in enumerate(zip(kinds, ms)):
    print(i // 2, names[kd], f"{t:.4f}", "prev-decision", int(log[i]) if i < len(log) else -1)
