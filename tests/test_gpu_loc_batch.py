"""pfx_loc_batch_*: many independent particleFilterLoc filters in one set of launches ("batch of
independent filters ... also many Loc tags"; the reference's callers run one filter per tag, csvlocalize.c:64-88).

  * filter f of a batch == a single filter with its seed and stream id after the same calls, BIT FOR BIT (particles,
    decision scalars, poses) -- with per-filter masks, filters that start later than others, pose queries in between,
    for filters of one chunk (N = 10 000, the reference's size: k_locb_decide_small) and of several (k_locb_chunks +
    k_locb_decide_map);
  * one filter of a batch against the ORACLE directly (same Philox key, f64 accumulation rule), through a run with
    both branches: decisions and sums bit-exact, particles to 2e-6.
"""
import numpy as np
import pytest

import _synth

pytestmark = pytest.mark.gpu


def _scene_calls(F, steps, rs):
    """per step: (t, pos[F,3], beacon index[F], range[F], mask for the range deposit, mask for an rssi deposit or None)"""
    bx, truth, rng = _synth.synthetic_scene(steps + 70)
    off = (np.arange(F) * 7) % 64
    calls = []
    for k in range(steps):
        kk = k + off
        mask = np.ones(F, dtype=bool) if k % 5 else rs.rand(F) < 0.6          # every fifth deposit reaches some filters only
        if k < 3:
            mask &= np.arange(F) != (F - 1)                                   # the last filter starts (spawns) later
        rssi_mask = (rs.rand(F) < 0.5) if k % 3 == 2 else None
        calls.append((10.0 + 0.1 * k, truth[kk], kk % 8, rng[kk, 0], mask, rssi_mask, bx))
    return calls


@pytest.mark.parametrize("n", [777, 10000, 65536, 66000, 300000])     # 1 tile; 10 tiles; 64 tiles (last one-wavefront size); 65; two chunks
def test_batch_equals_single_filters(n):
    from slam3d_amd import particlefilter as pfm
    F, steps = 5, 26
    rs = np.random.RandomState(n % 1000)
    seeds = np.array([11, 11, 12, 13, 11], dtype=np.uint32)
    sids = np.array([0, 1, 0, 5, 2], dtype=np.uint32)
    batch = pfm.DeviceLocBatch(F, n, seeds=seeds, stream_ids=sids)
    singles = [pfm.DeviceParticleFilterLoc(n, seed=int(seeds[f]), stream_id=int(sids[f])) for f in range(F)]
    for k, (t, pos, b, r, mask, rssi_mask, bx) in enumerate(_scene_calls(F, steps, rs)):
        vio_mask = None if k % 4 else (np.arange(F) % 2 == 0)                 # now and then only some filters get VIO
        batch.deposit_vio(t, pos[:, 0], pos[:, 1], pos[:, 2], mask=vio_mask)
        batch.deposit_range(bx[b, 0], bx[b, 1], bx[b, 2], r, 0.1, mask=mask)
        for f, s in enumerate(singles):
            if vio_mask is None or vio_mask[f]:
                s.depositVio(t, float(pos[f, 0]), float(pos[f, 1]), float(pos[f, 2]), 0.0)
            if mask[f]:
                s.depositRange(float(bx[b[f], 0]), float(bx[b[f], 1]), float(bx[b[f], 2]), float(r[f]), 0.1)
        if rssi_mask is not None:
            b2 = (b + 1) % 8
            batch.deposit_rssi(bx[b2, 0], bx[b2, 1], bx[b2, 2], mask=rssi_mask)
            for f, s in enumerate(singles):
                if rssi_mask[f]:
                    s.depositRssi(float(bx[b2[f], 0]), float(bx[b2[f], 1]), float(bx[b2[f], 2]), -60)
        if k % 6 == 4 or k == steps - 1:
            ok, tt, x, y, z, th = batch.get_tag_loc()
            for f, s in enumerate(singles):
                e = s.getTagLoc()
                assert (int(ok[f]), tt[f], x[f], y[f], z[f], th[f]) == e, (k, f)
    dec, rsd = batch.branch_counts(per_filter=True)
    assert rsd.sum() > 0 and (dec - rsd).sum() > 0                            # both branches were taken
    for f, s in enumerate(singles):
        assert batch.last_stats(f) == s.last_stats(), f
        log, total = s.branch_log()
        assert total == dec[f] and int(log.sum()) == rsd[f]
        got, exp = batch.download(f), s.download()
        for c in range(5):
            assert np.array_equal(got[c], exp[c]), (f, c)


def test_batch_filter_tracks_the_oracle():
    from slam3d_amd import particlefilter as pfm
    from oracle import oracle as ora
    F, n, steps, f0 = 3, 20000, 14, 1
    batch = pfm.DeviceLocBatch(F, n, seed=777, stream_ids=np.zeros(F, dtype=np.uint32), seeds=np.array([5, 777, 9], dtype=np.uint32))
    o = ora.OracleLoc(n, rng_mode=ora.RNG_PHILOX, seed=777, acc=ora.ACC_F64)      # Philox key (777, stream 0), as filter 1
    bx, truth, rng = _synth.synthetic_scene(steps + 1)
    branches = []
    for k in range(steps):
        pos = truth[k]
        batch.deposit_vio(20.0 + 0.1 * k, pos[0], pos[1], pos[2])
        o.depositVio(20.0 + 0.1 * k, float(pos[0]), float(pos[1]), float(pos[2]), 0.0)
        b = bx[k % 8]
        if k % 4 == 3:
            batch.deposit_rssi(b[0], b[1], b[2])
            o.depositRssi(float(b[0]), float(b[1]), float(b[2]), -60)
        else:
            batch.deposit_range(b[0], b[1], b[2], rng[k, 0], 0.1)
            o.depositRange(float(b[0]), float(b[1]), float(b[2]), float(rng[k, 0]), 0.1)
        if k == 0:
            continue
        gs, es = batch.last_stats(f0), o.last_stats()
        assert int(gs["branch"]) == int(es["branch"]) and int(gs["numSpawn"]) == int(es["numSpawn"]), k
        if gs["S"] != es["S"]:
            pytest.skip(f"a gate flip at deposit {k} (a particle within 2e-6 m of the 3-sigma shell): see test_gpu_loc._track_oracle")
        assert gs["s"] == es["s"], k
        if int(gs["branch"]) == 1:
            assert gs["rStart"] == es["rStart"] and gs["rStep"] == es["rStep"] and gs["htheta"] == pytest.approx(es["htheta"], rel=1e-5), k
        else:
            assert gs["m"] == es["m"], k
        branches.append(int(gs["branch"]))
        pose_g = batch.get_tag_loc()
        pose_o = o.getTagLoc()
        assert pose_g[0][f0] == pose_o[0] == 1
        assert np.allclose([pose_g[2][f0], pose_g[3][f0], pose_g[4][f0]], pose_o[2:5], rtol=0, atol=1e-4), k
    assert 0 < sum(branches) < len(branches)
    got, exp = batch.download(f0), o.arrays()
    assert np.array_equal(got[0], exp[0])
    for c in range(1, 5):
        assert np.allclose(got[c], exp[c], rtol=0, atol=1e-5), c


def test_batch_rejects_bad_arguments():
    from slam3d_amd import _lib, particlefilter as pfm
    with pytest.raises(pfm.PfxError):
        pfm.DeviceLocBatch(0, 1000)
    with pytest.raises(pfm.PfxError):
        pfm.DeviceLocBatch(2, 0)
    b = pfm.DeviceLocBatch(2, 1000, seed=1)
    ok, *_ = b.get_tag_loc()
    assert not ok.any()                                                       # nothing initialised yet
    with pytest.raises(pfm.PfxError):
        b.download(5)
    with pytest.raises(ValueError):
        b.deposit_range(0.0, 0.0, 0.0, 1.0, 0.1, mask=[1, 0, 1])


def test_device_gate_bounds_equal_the_host_bounds():
    """k_locb_prepare derives the gate's squared-distance bounds on the device (bracketed bisection); single filters get
    them from the host (gate_of): the same floats for ordinary, tiny, huge, zero, negative, infinite and NaN gates"""
    import ctypes as C
    from slam3d_amd import _lib
    rs = np.random.RandomState(9)
    r = np.concatenate([rs.uniform(0.0, 30.0, 4000), rs.uniform(0.0, 0.5, 500), 10.0 ** rs.uniform(-30, 30, 500),
                        [0.0, 1.5, 3.0, 1e-20, 5.0, 5.0, np.inf, 5.0, -np.inf, np.nan, 1e19, 3e38, 1e-45, 2.0]]).astype(np.float32)
    s = np.concatenate([rs.uniform(0.0, 1.0, 4000), rs.uniform(0.0, 3.0, 500), 10.0 ** rs.uniform(-30, 30, 500),
                        [0.0, 0.5, 0.0, 1e-21, np.inf, np.nan, 0.1, -0.1, 0.1, 0.1, 1e18, 1e38, 1e-45, 1e30]]).astype(np.float32)
    n = r.shape[0]
    out = [np.empty(n, dtype=np.float32) for _ in range(4)]
    p = lambda a: a.ctypes.data_as(C.c_void_p)      # noqa: E731
    _lib.check(_lib.lib().pfx_gate_bounds_both(p(r), p(s), n, *[p(a) for a in out], 0), "pfx_gate_bounds_both")
    lo_h, hi_h, lo_d, hi_d = out
    assert np.array_equal(lo_h.view(np.uint32), lo_d.view(np.uint32))
    assert np.array_equal(hi_h.view(np.uint32), hi_d.view(np.uint32))
    assert np.isfinite(lo_h[:4000]).all() and (hi_h[:4000] > lo_h[:4000]).all()
