"""Seed-to-seed spread of the reference SLAM filter (oracle, reference arithmetic) on the trimmed cic/0 trace:
final tag pose and beacon estimates against the glibc golden.  Sets the thresholds of tests/test_gpu_slam.py::test_cic0_replay."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as ora          # noqa: E402
from slam3d_amd import replay as R         # noqa: E402

d = os.path.join(ROOT, "tests", "golden", "cic0")
g = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_cic0_slam.json")))["glibc"]
vio, uwb = R.read_vio_verbose(os.path.join(d, "vio.csv")), R.read_uwb_verbose(os.path.join(d, "uwb.csv"))
gt = np.array(g["tag_rows"])
gb = np.array(g["bcn_rows"])
for mode, acc, name in ((ora.RNG_GLIBC, ora.ACC_F32, "glibc"), (ora.RNG_PHILOX, ora.ACC_F64, "philox")):
    for seed in (1, 2, 3):
        pf = ora.OracleSlam(100, 1000, rng_mode=mode, seed=seed, acc=acc)
        bc = [pf.addBcn() for _ in range(12)]
        res = R.replay_slam(pf, bc, vio, uwb)
        t = np.array(res.tag_rows)
        b = np.array(res.bcn_rows)
        dt = np.linalg.norm(t[:, 1:4] - gt[:, 1:4], axis=1)
        db = np.abs(b[:, 1:4] - gb[:, 1:4])
        print(name, seed, "tag max %.4f" % dt.max(), "theta max %.2e" % np.abs(t[:, 4] - gt[:, 4]).max(),
              "bcn |dx| max %.3f |dy| max %.3f |dz| max %.3f" % tuple(db.max(0)), "dist max %.3f" % np.linalg.norm(db, axis=1).max())
