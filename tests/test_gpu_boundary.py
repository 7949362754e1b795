"""Boundary behaviour of the reference-layout entry points on the GPU:
  * particleFilterSeed_set + init + run reproduces its sample path however many filters the process initialised before
    (the reference reseeds its one global stream, pfRandom.c:51-60); filters initialised after one seed differ;
  * the Loc stage functions of the reference's internal headers (include/pfStages.h) forward to the GPU stages;
  * particleFilterSlam_addBcn resets a beacon without using up another of the filter's slots, and
    particleFilterSlam_init on a bound struct keeps the beacons (they are structs of their own in the reference);
  * pfx_set_abort_on_error(0): a failing call is reported and ignored instead of aborting the process."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
SEED = 424242


def _replay_csv(pfm, vio, uwb, dep, n_events=300):
    from slam3d_amd import replay as R
    f = pfm.ParticleFilterLoc()
    return R.replay_loc(f, vio, uwb, dep, max_events=n_events).csv_text()


def test_set_seed_then_run_is_reproducible_in_one_process(test1):
    from slam3d_amd import particlefilter as pfm
    vio, uwb, dep = test1
    pfm.setSeed(SEED)
    a = _replay_csv(pfm, vio, uwb, dep)
    other = _replay_csv(pfm, vio, uwb, dep)         # a second filter after the same seed: its own sub-stream
    pfm.setSeed(SEED)
    b = _replay_csv(pfm, vio, uwb, dep)
    assert a == b and len(a) > 1000
    assert other != a


def test_slam_filters_after_one_seed_do_not_share_noise():
    from slam3d_amd import particlefilter as pfm, slam
    def run():
        f = slam.ParticleFilterSlam()
        b = f.addBcn()
        for k in range(4):
            f.depositTagVio(1.0 + 0.1 * k, 0.05 * k, 0.0, 0.0, 0.0)
            f.depositRange(b, 2.0 + 0.1 * k, 0.1)
        return f.getTagLoc(), f.getBcnLoc(b)
    pfm.setSeed(SEED)
    first = run()
    second = run()
    pfm.setSeed(SEED)
    again = run()
    assert first == again
    assert first != second


def test_loc_stage_functions_forward_to_the_gpu_stages():
    from slam3d_amd import _lib, particlefilter as pfm
    L = _lib.lib()
    f = C.c_float
    for name in ("pfInit_initTagLoc", "pfMeasurement_applyRangeLoc", "pfResample_resampleLoc"):
        getattr(L, name).argtypes = [C.c_void_p, f, f, f, f, f]
        getattr(L, name).restype = None
    L.pfMeasurement_applyVioLoc.argtypes = [C.c_void_p, f, f, f, f, f]
    L.pfMeasurement_applyVioLoc.restype = None
    pfm.setSeed(SEED)
    legacy = pfm.ParticleFilterLoc()
    dev = legacy.device_filter()
    ext = pfm.DeviceParticleFilterLoc(10000, seed=dev.seed, stream_id=dev.stream_id)
    dev.epoch = ext.epoch = 3
    L.pfInit_initTagLoc(legacy._p, 1.0, 2.0, 0.5, 2.8, 0.1)
    ext.spawn_all(1.0, 2.0, 0.5, 2.8, 0.1)
    L.pfMeasurement_applyVioLoc(legacy._p, 0.1, 0.04, -0.02, 0.01, 0.05)
    ext.apply_vio(0.1, 0.04, -0.02, 0.01, 0.05)
    L.pfMeasurement_applyRangeLoc(legacy._p, 1.0, 2.0, 0.5, 2.85, 0.02)
    ext.apply_range(1.0, 2.0, 0.5, 2.85, 0.02)
    L.pfResample_resampleLoc(legacy._p, 1.0, 2.0, 0.5, 2.85, 0.02)
    ext.resample(1.0, 2.0, 0.5, 2.85, 0.02)
    assert int(ext.last_stats()["branch"]) == 1
    got = legacy.particles()
    exp = ext.download()
    for k in range(5):
        assert np.array_equal(got[:, k], exp[k])


def test_slam_add_bcn_resets_without_using_slots():
    from slam3d_amd import _lib, particlefilter as pfm, slam
    L = _lib.lib()
    pfm.setSeed(SEED)
    f = slam.ParticleFilterSlam()
    b = f.addBcn()
    h = L.pfx_slam_of(f._p)
    for k in range(40):                             # far more resets than the filter has beacon slots (32)
        f.depositTagVio(1.0 + 0.1 * k, 0.01 * k, 0.0, 0.0, 0.0)
        f.depositRange(b, 2.0, 0.1)
        assert b.initialized and f.getBcnLoc(b)[0] == 1
        assert L.pfx_slam_slot_of(f._p, b._p) == 0
        L.particleFilterSlam_addBcn(b._p)           # the reference's way of resetting a beacon (particleFilter.c:65-68)
        assert not b.initialized and f.getBcnLoc(b)[0] == 0
    # re-initialising the filter keeps its beacons (the reference's init touches the tag particles only)
    f.depositRange(b, 2.0, 0.1)
    before = f.bcn_particles(b)
    L.particleFilterSlam_init(f._p)
    assert L.pfx_slam_of(f._p) == h and b.initialized
    assert np.array_equal(f.bcn_particles(b), before)
    tag = f.tag_particles()
    assert np.all(tag[:, 0] == 1.0) and np.all(tag[:, 1:] == 0.0)


def test_errors_can_be_reported_instead_of_aborting():
    from slam3d_amd import _lib, particlefilter as pfm
    L = _lib.lib()
    unbound = C.create_string_buffer(pfm.SIZEOF_LOC)            # never initialised: no filter behind it
    L.pfx_set_abort_on_error(0)
    try:
        n0 = L.pfx_error_count()
        L.particleFilterLoc_depositRange(C.cast(unbound, C.c_void_p), 0.0, 0.0, 0.0, 1.0, 0.1)
        assert L.pfx_error_count() == n0 + 1
        assert b"not initialised" in L.pfx_last_error()
        t = C.c_double()
        x = C.c_float()
        assert L.particleFilterLoc_getTagLoc(C.cast(unbound, C.c_void_p), C.byref(t), C.byref(x), C.byref(x), C.byref(x), C.byref(x)) == 0
    finally:
        L.pfx_set_abort_on_error(1)
