"""GPU parity tests of the particleFilterSlam hot path: HIP kernels (through the C-ABI) vs the CPU oracle.

The oracle runs in Philox mode with the f64 accumulation rule (ORA_RNG_PHILOX, ORA_ACC_F64), the rule the
kernels implement; its restatement of the reference's SLAM path is pinned separately against goldens taken from
the compiled reference (tests/test_oracle_vs_ref_golden.py::test_slam_scenario; the reference itself has no SLAM
test).

Bars: branch decisions (tag resample yes/no, beacon initialised) identical; resampling indices identical (the weight
sums and CDFs are exact sums of multiples of 2^-20 on both sides, DESIGN.md section 4.1, so there is no summation
order left): EVERY particle of the tag cloud and of every beacon row, after EVERY deposit, has its weight within 1e-6
relative and its position/heading (behind Box-Muller and sin/cos) within 5e-6 absolute of the oracle's -- one differing
ancestor anywhere would show as a particle metres away; pose estimates within 2e-5.  (A particle within ~2e-6 m of the
3-sigma shell could be gated on one side only and make a row diverge, as the Loc tests describe; the seeded scenario
has none in any of the four shapes, so nothing is tolerated here.)
"""
import numpy as np
import pytest

import _synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def slam():
    from slam3d_amd import particlefilter as pfm
    from slam3d_amd import slam as m
    if pfm.device_count() < 1:
        pytest.fail("no HIP device visible: the GPU tests must run on an MI355X box")
    return m


def _drive(pf, bcns, ev, n_tag):
    """apply one scenario event to either implementation; returns True if it was a deposit"""
    if ev[0] == "vio":
        pf.depositTagVio(ev[1], ev[2], ev[3], ev[4], 0.0)
    elif ev[0] == "bvio":
        bcns[ev[1]].depositVio(ev[2], ev[3], ev[4], ev[5], 0.0)
    elif ev[0] == "tagw":
        w = _synth.slam_tag_weights(ev[1], n_tag)
        if hasattr(pf, "set_tag_weights"):
            pf.set_tag_weights(w)
        else:
            pf.arrays()[0][:] = w
    elif ev[0] == "range":
        pf.depositRange(bcns[ev[1]], ev[2], ev[3])
        return True
    else:
        pf.depositRssi(bcns[ev[1]], ev[2])
        return True
    return False


def _close_particles(got, exp, what):
    """weights relative 1e-6, x y z theta absolute 5e-6, every particle"""
    n = exp[0].size
    bad = np.abs(got[0] - exp[0]).reshape(-1) > 1e-6 * np.maximum(1.0, np.abs(exp[0]).reshape(-1))
    for k in range(1, 5):
        d = np.abs(got[k] - exp[k]).reshape(-1)
        if k == 4:
            d = np.minimum(d, np.abs(d - 6.2831855))
        bad |= d > 5e-6
    assert not bad.any(), f"{what}: {int(bad.sum())} of {n} particles differ"


@pytest.mark.parametrize("n_tag,n_bcn", [(100, 1000), (7, 37), (64, 256), (5, 1024)])
def test_slam_scenario_tracks_oracle(slam, ora, n_tag, n_bcn):
    sc = _synth.slam_scenario()
    g = slam.DeviceParticleFilterSlam(n_tag, n_bcn, seed=sc["seed"], max_bcns=4)
    o = ora.OracleSlam(n_tag, n_bcn, rng_mode=ora.RNG_PHILOX, seed=sc["seed"], acc=ora.ACC_F64)
    gb = [g.addBcn() for _ in range(sc["num_bcns"])]
    ob = [o.addBcn() for _ in range(sc["num_bcns"])]
    k = 0
    n_tag_rs = 0
    for ev in sc["events"]:
        dep = _drive(g, gb, ev, n_tag)
        _drive(o, ob, ev, n_tag)
        if not dep:
            continue
        k += 1
        assert g.last_tag_resampled == o.last_tag_resampled, f"deposit {k}"
        n_tag_rs += int(o.last_tag_resampled)
        a, b = g.getTagLoc(), o.getTagLoc()
        assert a[0] == b[0] == 1 and a[1] == b[1]
        assert np.abs(np.array(a[2:], dtype=np.float64) - np.array(b[2:], dtype=np.float64)).max() <= 2e-5, f"deposit {k}"
        for x, y in zip(gb, ob):
            assert x.initialized == y.initialized
            pa, pb = g.getBcnLoc(x), o.getBcnLoc(y)
            assert pa[0] == pb[0]
            if pa[0]:
                assert pa[1] == pb[1]
                d = np.abs(np.array(pa[2:], dtype=np.float64) - np.array(pb[2:], dtype=np.float64))
                d[3] = min(d[3], abs(d[3] - 2 * np.pi))
                assert d.max() <= 2e-5, f"deposit {k}: {pa} vs {pb}"
        _close_particles(g.arrays(), o.arrays(), f"tag particles after deposit {k}")
        for i, (x, y) in enumerate(zip(gb, ob)):
            if y.initialized:
                _close_particles(x.arrays(), y.arrays(), f"beacon {i} after deposit {k}")
    assert k == 36
    if n_tag == 100:
        assert n_tag_rs >= 2, "the scenario should exercise the tag-resample path (every beacon force-resampled)"


def test_slam_first_range_spawns_on_shell(slam):
    """pfInit.c:34-45: beacon particles of row i lie on the range shell around tag particle i, w = 1."""
    g = slam.DeviceParticleFilterSlam(100, 1000, seed=7)
    b = g.addBcn()
    assert g.getBcnLoc(b)[0] == 0 and not b.initialized
    g.depositTagVio(1.0, 0.0, 0.0, 0.0, 0.0)
    g.depositTagVio(1.5, 0.5, -0.25, 0.125, 0.0)
    g.depositRange(b, 3.0, 0.1)
    assert b.initialized
    tw, tx, ty, tz, tt = g.arrays()
    w, x, y, z, th = b.arrays()
    assert np.all(w == 1.0) and np.all(tw == 1.0)
    r = np.sqrt((x - tx[:, None]) ** 2 + (y - ty[:, None]) ** 2 + (z - tz[:, None]) ** 2)
    assert r.min() >= 3.0 - 0.3 - 1e-5 and r.max() <= 3.0 + 0.3 + 1e-5
    assert th.min() >= 0.0 and th.max() <= 6.2831855
    # the tag particles moved by the committed VIO delta (rotated by theta = 0, plus 1e-3 * sqrt(dist) noise)
    assert np.abs(tx - 0.5).max() < 0.01 and np.abs(ty + 0.25).max() < 0.01 and np.abs(tz - 0.125).max() < 0.01


def test_reference_layout_api_matches_extension(slam):
    """particleFilterSlam_* on caller-owned structs (4 056 B / 2 020 056 B) run the same kernels as the pfx_slam_*
    filter: identical seed -> identical particles, bit for bit, and pTag[] / pBcn[][] after pfx_slam_sync_host equal
    the downloaded SoA state."""
    from slam3d_amd import particlefilter as pfm
    sc = _synth.slam_scenario()
    pfm.setSeed(sc["seed"])
    legacy = slam.ParticleFilterSlam()
    ext = slam.DeviceParticleFilterSlam(100, 1000, seed=sc["seed"], max_bcns=8)
    lb = [legacy.addBcn() for _ in range(3)]
    eb = [ext.addBcn() for _ in range(3)]
    k = 0
    for ev in sc["events"]:
        if ev[0] == "tagw":
            continue
        dep = _drive(legacy, lb, ev, 100)
        _drive(ext, eb, ev, 100)
        if dep:
            k += 1
            assert legacy.getTagLoc() == ext.getTagLoc()
            for x, y in zip(lb, eb):
                assert x.initialized == y.initialized
                assert legacy.getBcnLoc(x) == ext.getBcnLoc(y)
        if k == 14:
            break
    p = legacy.tag_particles()
    soa = ext.arrays()
    for c in range(5):
        assert np.array_equal(p[:, c], soa[c])
    q = legacy.bcn_particles(lb[1])
    soa = eb[1].arrays()
    for c in range(5):
        assert np.array_equal(q[:, :, c], soa[c])


def test_batch_equals_single_filters(slam):
    """A batch of filters in one set of launches == the same filters one by one (filters never interact);
    filters that sit a deposit out (slot -1) keep accumulating VIO and are untouched."""
    F, n_tag, n_bcn = 5, 20, 200
    seeds = [11, 22, 33, 44, 55]
    batch = slam.DeviceSlamBatch(F, n_tag, n_bcn, max_bcns=3, seeds=seeds)
    singles = [slam.DeviceSlamBatch(1, n_tag, n_bcn, max_bcns=3, seeds=[s]) for s in seeds]
    for _ in range(2):
        batch.add_bcn_all()
        for s in singles:
            s.add_bcn(0)
    rng = np.random.default_rng(5)
    for step in range(14):
        t = 10.0 + 0.1 * step
        pos = rng.normal(size=(F, 3)).astype(np.float32) * 0.05 + np.float32(0.02 * step)
        batch.deposit_tag_vio_batch(np.full(F, t), pos[:, 0], pos[:, 1], pos[:, 2])
        for f, s in enumerate(singles):
            s.deposit_tag_vio(0, t, float(pos[f, 0]), float(pos[f, 1]), float(pos[f, 2]))
        slot = np.array([(step + f) % 2 for f in range(F)], dtype=np.int32)
        if step % 3 == 1:
            slot[step % F] = -1
        r = (2.0 + 0.1 * rng.random(F)).astype(np.float32)
        sd = np.full(F, 0.1, dtype=np.float32)
        batch.deposit_range_batch(slot, r, sd)
        for f, s in enumerate(singles):
            if slot[f] >= 0:
                s.deposit_range(0, int(slot[f]), float(r[f]), 0.1)
    bt = batch.get_tag_loc_batch()
    for f, s in enumerate(singles):
        for a, b in zip(batch.download_tag(f), s.download_tag(0)):
            assert np.array_equal(a, b)
        for c in range(2):
            for a, b in zip(batch.download_bcn(f, c), s.download_bcn(0, c)):
                assert np.array_equal(a, b)
        one = s.get_tag_loc(0)
        assert one[1] == bt[0][f] and one[2] == bt[1][f] and one[5] == bt[4][f]
        assert batch.get_bcn_loc(f, 1) == s.get_bcn_loc(0, 1)
    ok, _t, bx, _by, _bz, _bth = batch.get_bcn_loc_batch(np.ones(F, dtype=np.int32))
    assert ok.all() and bx[2] == batch.get_bcn_loc(2, 1)[2]


def test_row_filter_properties_full_size(slam):
    """Size-independent properties on BASELINE-sized filters (100 x 1000 per beacon, 4 beacons, 6 filters): after
    a deposit every row either holds w == 1 everywhere (resampled) or has mean weight 1 (normalised); tag weights
    have mean 1 or are all 1; beacons that were not measured and did not move are bitwise untouched."""
    F = 6
    batch = slam.DeviceSlamBatch(F, 100, 1000, max_bcns=4, seed0=900)
    for _ in range(4):
        batch.add_bcn_all()
    z = np.zeros(F, dtype=np.float32)
    for step in range(8):
        batch.deposit_tag_vio_batch(np.full(F, 5.0 + step), z + 0.05 * step, z, z)
        batch.deposit_range_batch(np.full(F, step % 4, dtype=np.int32), np.full(F, 2.5 + 0.05 * step, dtype=np.float32),
                                  np.full(F, 0.1, dtype=np.float32))
    before = batch.download_bcn(3, 2)
    batch.deposit_tag_vio_batch(np.full(F, 20.0), z + 0.5, z, z)
    batch.deposit_range_batch(np.full(F, 1, dtype=np.int32), np.full(F, 2.6, dtype=np.float32), np.full(F, 0.1, dtype=np.float32))
    for f in range(F):
        tw = batch.download_tag(f)[0]
        assert np.all(tw == 1.0) or abs(float(tw.mean()) - 1.0) < 1e-5
        w = batch.download_bcn(f, 1)[0]
        rows_one = np.all(w == 1.0, axis=1)
        rows_norm = np.abs(w.mean(axis=1) - 1.0) < 1e-4
        assert np.all(rows_one | rows_norm)
    if not batch.last_tag_branch(3):
        after = batch.download_bcn(3, 2)
        for a, b in zip(before, after):
            assert np.array_equal(a, b)


def test_cic0_replay_drop_in(slam):
    """The csvslam workload on real data (trimmed sampledata/cic/0, tests/golden/cic0) through the DROP-IN entry points
    (particleFilterSlam_* on caller-owned structs, 12 bcn_t of 2 MB each), against what the compiled reference wrote
    (tests/golden/ref_cic0_slam.json, glibc flavour).  Philox noise cannot follow the rand_r sample path, so the bars
    are the reference's own seed-to-seed spread on this trace (tests/tools/cic0_spread.py, 3 rand_r + 3 Philox seeds of
    the oracle against the golden: tag path <= 1.5 mm, heading <= 1e-6 rad, beacon estimates |dx| <= 0.16,
    |dy| <= 0.55, |dz| <= 0.24 m -- y is barely observable in the first 10 s of the walk), times 1.6."""
    import json
    import os
    from slam3d_amd import particlefilter as pfm, replay as R
    root = os.path.dirname(os.path.abspath(__file__))
    with open(os.path.join(root, "golden", "ref_cic0_slam.json")) as fh:
        g = json.load(fh)["glibc"]
    d = os.path.join(root, "golden", "cic0")
    vio, uwb = R.read_vio_verbose(os.path.join(d, "vio.csv")), R.read_uwb_verbose(os.path.join(d, "uwb.csv"))
    pfm.setSeed(g["seed"])
    pf = slam.ParticleFilterSlam()
    bcns = [pf.addBcn() for _ in range(R.SLAM_NUM_BCNS)]
    res = R.replay_slam(pf, bcns, vio, uwb)
    assert (res.n_vio, res.n_range) == (g["n_vio"], g["n_range"])
    t, gt = np.array(res.tag_rows), np.array(g["tag_rows"])
    assert t.shape == gt.shape and np.array_equal(t[:, 0], gt[:, 0])
    assert np.linalg.norm(t[:, 1:4] - gt[:, 1:4], axis=1).max() < 0.0025
    assert np.abs(t[:, 4] - gt[:, 4]).max() < 2e-6
    b, gb = np.array(res.bcn_rows), np.array(g["bcn_rows"])
    assert b.shape == gb.shape and np.array_equal(b[:, 0], gb[:, 0])        # the same ten beacons were heard
    db = np.abs(b[:, 1:4] - gb[:, 1:4]).max(axis=0)
    assert db[0] < 0.26 and db[1] < 0.88 and db[2] < 0.39, db
    # csvslam's output files, as text
    assert res.tag_csv().count("\n") == 1 + len(res.tag_rows) and res.bcn_csv().startswith("b,x,y,z,theta\n0,")


def _walk(edge, half):
    """2 * half + 1 consecutive float32 values centred on edge"""
    lo = np.float32(edge)
    for _ in range(half):
        lo = np.nextafter(lo, np.float32(-np.inf))
    out = [lo]
    for _ in range(2 * half):
        out.append(np.nextafter(out[-1], np.float32(np.inf)))
    return np.array(out, dtype=np.float32)


@pytest.mark.parametrize("moving", [False, True], ids=["static-kernel", "moving-kernel"])
@pytest.mark.parametrize("rng_,std", [(3.0, 0.1), (0.25, 0.1), (7.5, 0.0), (2.0, 1e-3)])
def test_slam_gate_at_its_boundaries(slam, ora, rng_, std, moving):
    """VERDICT r02 item 7b, the SLAM counterpart of test_gpu_loc.py::test_range_gate_at_its_boundaries.  Every row holds
    beacon particles whose distance to the row's tag particle walks float by float across an edge of the 3-sigma shell
    (pfMeasurement.c:118-123 through applyRangeSlam); rows with the tag particle at the origin resolve the edge to one
    ulp of the distance, the others to one ulp of the position.  Even row pairs carry uniform weights (a gated weight is
    halved, so the ESS stays high and the row is normalised), odd pairs carry their weight on 100 particles (the row
    resamples).  One particle gated on one side only changes a row sum, the tag weight and, in a resampled row,
    ancestors: all of it is compared.
    moving: the beacon has a VIO delta of 1e-30 m at dt = 0 -- a motion as far as the library is concerned, so
    k_slam_gate_rows_moving and k_slam_rows run, but one whose noise scales sqrtf(ddist) and sqrtf(dt) are 0 and whose
    translation is absorbed by rounding, so every distance is still the one laid out here.  Otherwise no beacon VIO:
    k_slam_gate_rows_static.  The tag particles take a zero translation over dt = 1 (heading noise only)."""
    n_tag, n_bcn = 100, 1000
    g = slam.DeviceParticleFilterSlam(n_tag, n_bcn, seed=99, max_bcns=2)
    o = ora.OracleSlam(n_tag, n_bcn, rng_mode=ora.RNG_PHILOX, seed=99, acc=ora.ACC_F64)
    gb, ob = g.addBcn(), o.addBcn()
    for pf, b in ((g, gb), (o, ob)):
        pf.depositTagVio(1.0, 0.0, 0.0, 0.0, 0.0)
        b.depositVio(1.0, 0.0, 0.0, 0.0, 0.0)
    g.depositRange(gb, 3.0, 0.1)                           # first range: spawns the rows (epoch 1 on both sides)
    o.depositRange(ob, 3.0, 0.1)
    rs = np.random.default_rng(5)
    tag = np.zeros((5, n_tag), dtype=np.float32)
    tag[0] = 1.0
    tag[1:4, n_tag // 4:] = rs.uniform(-20, 20, size=(3, n_tag - n_tag // 4)).astype(np.float32)   # first quarter at the origin
    tag[4] = rs.uniform(0, 6.28, size=n_tag).astype(np.float32)
    r, t = np.float32(rng_), np.float32(3.0) * np.float32(std)
    edges = [e for e in (r - t, r + t) if e > 0]
    bcn = np.zeros((5, n_tag, n_bcn), dtype=np.float32)
    bcn[0] = 1.0
    bcn[4] = rs.uniform(0, 6.28, size=(n_tag, n_bcn)).astype(np.float32)
    dirs = np.eye(3, dtype=np.float32)
    concentrated = np.zeros(n_tag, dtype=bool)
    for i in range(n_tag):
        e = edges[i % len(edges)]
        d = _walk(e, 250)                                  # 501 floats around the edge
        inside = np.float32(rng_)                          # on the shell
        outside = np.float32(rng_ + 10.0 * max(std, 0.01) + 1.0)
        filler = outside if (i // 4) % 2 else inside
        dist = np.concatenate([d, np.full(n_bcn - d.size, filler, dtype=np.float32)])
        rs.shuffle(dist)
        if i % 3 == 2:                                     # on a diagonal: d2 = h*h + h*h
            h = (dist * np.float32(0.70710678)).astype(np.float32)
            off = np.stack([h, np.zeros_like(h), h])
        else:
            off = dirs[i % 3][:, None] * dist[None, :] * np.float32(1.0 if i % 2 else -1.0)
        for k in range(3):
            bcn[1 + k, i] = tag[1 + k, i] + off[k].astype(np.float32)
        if (i // 2) % 2:
            concentrated[i] = True
            bcn[0, i] = np.float32(2.0 ** -7)
            bcn[0, i, rs.choice(n_bcn, size=100, replace=False)] = 1.0
    g.batch.upload_tag(0, *tag)
    g.batch.upload_bcn(0, gb.slot, *bcn, initialized=True)
    for dst, src in zip(o.arrays(), tag):
        dst[:] = src
    for dst, src in zip(ob.arrays(), bcn):
        dst[:] = src
    for pf, b in ((g, gb), (o, ob)):
        pf.depositTagVio(2.0, 0.0, 0.0, 0.0, 0.0)
        if moving:
            b.depositVio(1.0, 1e-30, 0.0, 0.0, 0.0)
    g.depositRange(gb, float(rng_), float(std))
    o.depositRange(ob, float(rng_), float(std))
    assert g.last_tag_resampled == o.last_tag_resampled
    ga, oa = g.arrays(), o.arrays()
    gw, ow = gb.arrays(), ob.arrays()
    _close_particles(ga, oa, "tag particles")
    _close_particles(gw, ow, "beacon particles")
    assert not o.last_tag_resampled
    # row by row: a resampled row has all weights 1, a normalised one keeps its pattern -- identical either way
    res_o = np.all(ow[0] == 1.0, axis=1)
    res_g = np.all(gw[0] == 1.0, axis=1)
    assert np.array_equal(res_o, res_g)
    assert np.all(res_o[concentrated]) and 0 < int(res_o.sum()) < n_tag, "both row branches should occur"
    keep = ~res_o
    # which particles were gated (weight below the row's maximum) must agree exactly
    gated_g = gw[0][keep] < gw[0][keep].max(axis=1, keepdims=True)
    gated_o = ow[0][keep] < ow[0][keep].max(axis=1, keepdims=True)
    assert np.array_equal(gated_g, gated_o)
    assert 0 < int(gated_o.sum()) < gated_o.size
    # positions of a row that was not resampled are what was uploaded (zero or absorbed beacon motion)
    # (a coordinate that was exactly 0 becomes 1e-30 * cos or sin of the heading: equal to the oracle's to ~1e-36 only)
    for k in (1, 2, 3):
        far = np.abs(bcn[k][keep]) > 1e-20
        assert np.array_equal(gw[k][keep][far], bcn[k][keep][far]) and np.array_equal(ow[k][keep][far], bcn[k][keep][far])
        assert np.abs(gw[k][keep][~far] - ow[k][keep][~far]).max(initial=0.0) <= 1e-29
    pa, pb = g.getBcnLoc(gb), o.getBcnLoc(ob)
    assert pa[0] == pb[0] == 1
    assert np.abs(np.array(pa[2:5], dtype=np.float64) - np.array(pb[2:5], dtype=np.float64)).max() <= 2e-5
