"""Seed-to-seed spread of the reference filter (oracle in reference arithmetic: rand_r + f32, N = 10 000) on the
RANGE-ONLY replay of test/data (BASELINE.json configs[1]: range-only updates): final pose per seed.
Used to set the thresholds of tests/test_gpu_loc.py::test_config2_range_only_1m."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as ora          # noqa: E402
from slam3d_amd import replay as R         # noqa: E402

vio, uwb, dep = R.load_test1(os.path.join(ROOT, "tests", "golden", "test1"))
out = []
for seed in range(1, 13):
    pf = ora.OracleLoc(10000, rng_mode=ora.RNG_GLIBC, seed=seed, acc=ora.ACC_F32)
    R.replay_loc(pf, vio, uwb, dep, range_only=True)
    out.append([float(v) for v in pf.getTagLoc()[2:]])
a = np.array(out)
print("final x y z theta per seed:\n", a)
print("mean", a.mean(0), "std", a.std(0), "min", a.min(0), "max", a.max(0))
