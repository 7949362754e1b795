#!/usr/bin/env python3
"""Seed-to-seed spread of the reference filter on the test/data trace (CPU, oracle in reference arithmetic).

Used to set the statistical-equivalence thresholds of tests/test_gpu_loc.py::test_test1_replay_statistically_equivalent
and of the CPU twin in tests/test_oracle_golden.py.  Prints, for each run against the linux-gcc golden, the RMS and
max position difference and the RMS heading difference over rows 100..874.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O            # noqa: E402
from slam3d_amd import replay as R        # noqa: E402

G = os.path.join(ROOT, "tests", "golden", "test1")
exp = np.loadtxt(os.path.join(G, "test1_ParticleFilterLoc_expected_out_linux-gcc.csv"), delimiter=",", skiprows=1)[:, [0, 3, 1, 2, 4]]
vio, uwb, dep = R.load_test1(G)


def stats(a):
    d = np.linalg.norm(a[100:, 1:4] - exp[100:, 1:4], axis=1)
    dth = np.angle(np.exp(1j * (a[100:, 4] - exp[100:, 4])))
    return np.sqrt((d ** 2).mean()), d.max(), np.sqrt((dth ** 2).mean()), np.linalg.norm(a[-1, 1:4] - exp[-1, 1:4])


for mode, acc, name in ((O.RNG_GLIBC, O.ACC_F32, "glibc/f32"), (O.RNG_MUSL, O.ACC_F32, "musl/f32"),
                        (O.RNG_PHILOX, O.ACC_F64, "philox/f64")):
    for seed in (1, 2, 3, 123456789):
        pf = O.OracleLoc(10000, rng_mode=mode, seed=seed, acc=acc)
        a = R.replay_loc(pf, vio, uwb, dep).as_array()
        print("%-11s seed %-10d pos rms %.3f max %.3f  theta rms %.3f  final %.3f" % ((name, seed) + stats(a)))
