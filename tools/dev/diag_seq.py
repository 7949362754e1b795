"""diagnostic: where do the GPU filter and the oracle part ways in the 30-step scene at N = 10000?"""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _synth
from slam3d_amd import particlefilter as pfm
from oracle import oracle as ora
from test_gpu_loc import _scene_events, _deposit_only
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
g = pfm.DeviceParticleFilterLoc(n, seed=123456789, stream_id=0)
g.set_deferred(False)
g.record_ancestors(True)
o = ora.OracleLoc(n, rng_mode=ora.RNG_PHILOX, seed=123456789, acc=ora.ACC_F64)
k = 0
for e in _scene_events(30):
    for f in (g, o):
        _deposit_only(f, [e])
    if e[0] == "vio":
        continue
    k += 1
    got, exp = g.download(), o.arrays()
    wbad = np.nonzero(np.abs(got[0] - exp[0]) > 1e-5 * np.maximum(1.0, np.abs(exp[0])))[0]
    pd = max(np.abs(got[c] - exp[c]).max() for c in range(1, 4))
    line = f"deposit {k} {e[0]} wbad {wbad.size} maxposdiff {pd:.3g}"
    if k > 1:
        gs, os_ = g.last_stats(), o.last_stats()
        line += f" branch {int(gs['branch'])}/{int(os_['branch'])} S {gs['S']!r} {os_['S']!r}"
        if int(gs["branch"]) and int(os_["branch"]):
            line += f" ancdiff {int((g.last_ancestors() != o.last_ancestors()).sum())}"
    print(line)
    if wbad.size and wbad.size < 5:
        bx, by, bz, rng = e[1], e[2], e[3], (e[4] if e[0] == "range" else 1.5)
        std = 0.1 if e[0] == "range" else 0.5
        for i in wbad:
            d = [float(np.sqrt((a[1][i]-bx)**2 + (a[2][i]-by)**2 + (a[3][i]-bz)**2)) for a in (got, exp)]
            print("   particle", i, "w", got[0][i], exp[0][i], "|range-d|-3s", [abs(x - rng) - 3*std for x in d])
