"""Pin every function of the CPU oracle against goldens generated from the COMPILED reference
(tests/golden/make_golden.py ran oracle/_ref/libpf_ref_{glibc,musl}.so in the build container).

The reference's own tests pin only the Loc replay; depositRssi, the spawners, the resampler in
isolation and the whole particleFilterSlam API have no reference test --
these goldens are what pins them.  In rand_r modes the oracle must match bit for bit: the hashes
are sha256 over the float32 bytes.
"""
import hashlib
import json
import os

import numpy as np
import pytest

import _synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FLAVOURS = {"glibc": 0, "musl": 1}


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.float32).tobytes()).hexdigest()


def _aos(arrs):
    return np.stack(arrs, axis=1)


@pytest.fixture(scope="module")
def kat():
    return np.load(os.path.join(GOLDEN, "ref_rng_kat.npz"))


@pytest.mark.parametrize("flavour", list(FLAVOURS))
@pytest.mark.parametrize("seed", [1, 123456789])
def test_rng_kat(ora, kat, flavour, seed):
    mode = FLAVOURS[flavour]
    r = ora.OracleRng(mode, seed)
    got = np.array([r.uniform() for _ in range(1000)], dtype=np.float32)
    assert np.array_equal(got, kat[f"{flavour}_uniform_{seed}"])
    r = ora.OracleRng(mode, seed)
    got = np.array([r.normal2() for _ in range(500)], dtype=np.float32)
    assert np.array_equal(got, kat[f"{flavour}_normal2_{seed}"])
    for tag, (rng_, std) in {"a": (5.0, 0.1), "b": (0.2, 0.5)}.items():
        r = ora.OracleRng(mode, seed)
        got = np.array([r.sphere(rng_, std) for _ in range(400)], dtype=np.float32)
        assert np.array_equal(got, kat[f"{flavour}_sphere{tag}_{seed}"])
        assert r.state == int(kat[f"{flavour}_sphere{tag}_{seed}_seedafter"][0])


@pytest.mark.parametrize("flavour", list(FLAVOURS))
def test_loc_replay_steps(ora, test1, flavour):
    from slam3d_amd import replay as R
    with open(os.path.join(GOLDEN, "ref_loc_steps.json")) as fh:
        g = json.load(fh)[flavour]
    vio, uwb, dep = test1
    pf = ora.OracleLoc(10000, rng_mode=FLAVOURS[flavour], seed=123456789, acc=ora.ACC_F32)
    log, snaps = [], {}
    orig = pf.depositRange

    def tapped(bx, by, bz, r, s):
        orig(bx, by, bz, r, s)
        k = len(log) + 1
        st = pf.last_stats() if k > 1 else {"branch": 1}
        log.append({"seed": pf.rng.state, "resampled": bool(st["branch"]), "pose": pf.getTagLoc()})
        snaps[k] = None
        for name in ("after_init", "after_deposit2", "after_first_resample", "after_first_spawn"):
            want = {"after_init": 1, "after_deposit2": 2}.get(name, g["snapshots"].get(name + "_step"))
            if want == k:
                snaps[name] = _sha(_aos(pf.arrays()))
    pf.depositRange = tapped
    res = R.replay_loc(pf, vio, uwb, dep)
    assert (len(res.rows), res.n_vio, res.n_range) == (g["n_rows"], g["n_vio"], g["n_range"])
    assert pf.rng.state == g["final_seed"]
    assert "".join("A" if r["resampled"] else "B" for r in log) == g["branches"]
    assert [r["seed"] for r in log[::50]] == g["seeds_every50"]
    for got, exp in zip(log[:20], g["first20"]):
        assert got["seed"] == exp["seed"]
        assert [float(v) for v in got["pose"]] == pytest.approx(exp["pose"], abs=0, rel=0)
    for name in ("after_init", "after_deposit2", "after_first_resample", "after_first_spawn"):
        assert name in g["snapshots"], name
        assert snaps[name] == g["snapshots"][name]["sha256"], name
    assert _sha(_aos(pf.arrays())) == g["snapshots"]["final"]["sha256"]
    assert hashlib.sha256(res.csv_text().encode()).hexdigest() == g["csv_sha256"]


@pytest.mark.parametrize("flavour", list(FLAVOURS))
@pytest.mark.parametrize("case", _synth.LOC_FUNC_CASES, ids=[c["name"] for c in _synth.LOC_FUNC_CASES])
def test_loc_functions(ora, flavour, case):
    with open(os.path.join(GOLDEN, "ref_loc_funcs.json")) as fh:
        g = json.load(fh)[flavour][case["name"]]
    p = _synth.loc_particles(10000, case["pseed"], case["kind"])
    pf = ora.OracleLoc(10000, rng_mode=FLAVOURS[flavour], seed=case["seed"], acc=ora.ACC_F32)
    pf.set_particles(*[p[:, k] for k in range(5)])
    pf.set_initialized(1)
    bx, by, bz, r, s = case["beacon"]
    if "range" in case["ops"]:
        pf.apply_range(bx, by, bz, r, s)
        w = pf.arrays()[0]
        assert int(np.sum(w != p[:, 0])) == g["n_gated"]
        assert _sha(w) == g["after_range"]["sha256"]
    if "vio" in case["ops"]:
        pf.apply_vio(*case["vio"])
        assert pf.rng.state == g["seed_after_vio"]
        assert _sha(_aos(pf.arrays())) == g["after_vio"]["sha256"]
    if "resample" in case["ops"]:
        pf.resample(bx, by, bz, r, s)
        assert bool(pf.last_stats()["branch"]) == g["resampled"]
        assert pf.rng.state == g["seed_after_resample"]
        assert _sha(_aos(pf.arrays())) == g["after_resample"]["sha256"]
    # getTagLoc with an uncommitted VIO delta (particleFilter.c:280-288)
    pf.depositVio(12.5, 0.125, 0.25, 0.0625, 0.0)       # latches first == last
    pf.depositVio(12.5, 0.25, -0.5, 0.125, 0.0)
    got = pf.getTagLoc()
    assert [float(v) for v in got] == g["pose"]


@pytest.mark.parametrize("flavour", list(FLAVOURS))
def test_slam_scenario(ora, flavour):
    with open(os.path.join(GOLDEN, "ref_slam_steps.json")) as fh:
        g = json.load(fh)[flavour]
    sc = _synth.slam_scenario()
    pf = ora.OracleSlam(100, 1000, rng_mode=FLAVOURS[flavour], seed=sc["seed"], acc=ora.ACC_F32)
    bcns = [pf.addBcn() for _ in range(sc["num_bcns"])]
    k = 0
    clamp0 = ora.clamp_events()
    for ev in sc["events"]:
        if ev[0] == "vio":
            pf.depositTagVio(ev[1], ev[2], ev[3], ev[4], 0.0)
            continue
        if ev[0] == "bvio":
            bcns[ev[1]].depositVio(ev[2], ev[3], ev[4], ev[5], 0.0)
            continue
        if ev[0] == "tagw":
            pf.arrays()[0][:] = _synth.slam_tag_weights(ev[1], 100)
            continue
        if ev[0] == "range":
            pf.depositRange(bcns[ev[1]], ev[2], ev[3])
        else:
            pf.depositRssi(bcns[ev[1]], ev[2])
        exp = g["steps"][k]
        k += 1
        # the golden is meaningful only while no offspring fell off the end of a CDF (reference: UB)
        assert ora.clamp_events() == clamp0, "scenario seed must keep the reference in bounds"
        assert pf.rng.state == exp["seed"], f"step {k}"
        assert [float(v) for v in pf.getTagLoc()] == exp["tag"], f"step {k}"
        for b, eb in zip(bcns, exp["bcn"]):
            got = pf.getBcnLoc(b)
            if eb[0] == 0:
                assert got[0] == 0
            else:
                assert [int(got[0])] + [float(v) for v in got[2:]] == eb, f"step {k}"
        assert _sha(_aos(pf.arrays())) == exp["tag_sha256"], f"step {k}"
        assert pf.last_tag_resampled == exp["tag_resampled"], f"step {k}"
    assert k == len(g["steps"])
    assert sum(s["tag_resampled"] for s in g["steps"]) >= 2, "scenario should exercise the tag-resample path"
    for i, b in enumerate(bcns):
        a = _aos([v.reshape(-1) for v in b.arrays()])
        assert _sha(a) == g["final"][f"bcn{i}"]["sha256"]


@pytest.mark.parametrize("flavour", list(FLAVOURS))
def test_cic0_csvslam_replay(ora, flavour):
    """csvslam.c's replay (12 beacons, bias 0.4, skip to waypoint 4) of the trimmed sampledata/cic/0 trace: the oracle
    in reference arithmetic reproduces what the compiled reference wrote (tag.csv rows, bcn.csv rows, final PF_SEED)
    exactly.  The reference has no test of its own for this path."""
    from slam3d_amd import replay as R
    with open(os.path.join(GOLDEN, "ref_cic0_slam.json")) as fh:
        g = json.load(fh)[flavour]
    d = os.path.join(GOLDEN, "cic0")
    vio, uwb = R.read_vio_verbose(os.path.join(d, "vio.csv")), R.read_uwb_verbose(os.path.join(d, "uwb.csv"))
    assert (len(vio), len(uwb)) == (591, 110) and vio[0][0] == 1515283343.999549     # rows before waypoint 4 skipped
    pf = ora.OracleSlam(100, 1000, rng_mode=FLAVOURS[flavour], seed=g["seed"], acc=ora.ACC_F32)
    bcns = [pf.addBcn() for _ in range(R.SLAM_NUM_BCNS)]
    clamp0 = ora.clamp_events()
    res = R.replay_slam(pf, bcns, vio, uwb)
    assert ora.clamp_events() == clamp0
    assert (res.n_vio, res.n_range) == (g["n_vio"], g["n_range"])
    assert pf.rng.state == g["final_seed"]
    assert [[float(v) for v in r] for r in res.tag_rows] == g["tag_rows"]
    assert [[int(r[0])] + [float(v) for v in r[1:]] for r in res.bcn_rows] == g["bcn_rows"]
