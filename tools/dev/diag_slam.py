"""How many particles / rows of the SLAM scenario differ between the GPU and the oracle, per deposit (diagnostic)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _synth
from slam3d_amd import slam
from oracle import oracle as ora
from test_gpu_slam import _drive


def bad_mask(got, exp):
    bad = np.abs(got[0] - exp[0]) > 1e-6 * np.maximum(1.0, np.abs(exp[0]))
    for k in range(1, 5):
        d = np.abs(got[k] - exp[k])
        if k == 4:
            d = np.minimum(d, np.abs(d - 6.2831855))
        bad |= d > 5e-6
    return bad


for n_tag, n_bcn in [(100, 1000), (7, 37), (64, 256), (5, 1024)]:
    sc = _synth.slam_scenario()
    g = slam.DeviceParticleFilterSlam(n_tag, n_bcn, seed=sc["seed"], max_bcns=4)
    o = ora.OracleSlam(n_tag, n_bcn, rng_mode=ora.RNG_PHILOX, seed=sc["seed"], acc=ora.ACC_F64)
    gb = [g.addBcn() for _ in range(sc["num_bcns"])]
    ob = [o.addBcn() for _ in range(sc["num_bcns"])]
    k = 0
    for ev in sc["events"]:
        dep = _drive(g, gb, ev, n_tag)
        _drive(o, ob, ev, n_tag)
        if not dep:
            continue
        k += 1
        bt = bad_mask(g.arrays(), o.arrays())
        line = f"n_tag={n_tag} n_bcn={n_bcn} dep {k:2d} ev={ev[0]}:{ev[1]} tagRS={int(o.last_tag_resampled)} tag bad {int(bt.sum())}"
        for i, (x, y) in enumerate(zip(gb, ob)):
            if y.initialized:
                b = bad_mask(x.arrays(), y.arrays())
                line += f" | b{i}: {int(b.sum())} in {int(b.any(axis=1).sum())} rows"
        print(line, flush=True)
