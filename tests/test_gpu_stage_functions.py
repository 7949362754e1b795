"""The Loc stage functions of the reference's internal headers (include/pfStages.h) used the way the reference's own
code uses them: on the struct.

  * the reference's golden generator (tests/golden/make_golden.py gen_loc_funcs, run there against the COMPILED
    REFERENCE) fills pf->pTag[] by hand, sets pf->initialized, calls a stage and reads pf->pTag[]: the same flow
    through libparticlefilter.so must reproduce the reference's own outputs -- the gated weights bit for bit
    (sha256 of the reference's array), the pose to float tolerance;
  * particleFilter.c:188-189 calls pfInit_initTagLoc(...) and sets pf->initialized = 1 AFTERWARDS: the flag the caller
    writes into the struct must reach the device filter (getTagLoc answers, the next deposit does not respawn).
"""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

import _synth

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
N = 10000
OFF_INIT, OFF_FIRST_T, OFF_FIRST_X, OFF_LAST_T, OFF_LAST_X = 400000, 400008, 400016, 400032, 400040   # particleFilter.h:48-64


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.float32).tobytes()).hexdigest()


def _struct(pfm, L, seed):
    buf = C.create_string_buffer(pfm.SIZEOF_LOC)
    p = C.cast(buf, C.c_void_p)
    L.particleFilterSeed_set(seed)
    L.particleFilterLoc_init(p)
    return buf, p


def _put_particles(buf, arr):
    flat = np.ascontiguousarray(arr, dtype=np.float32)
    C.memmove(buf, flat.ctypes.data, flat.nbytes)


def _get_particles(buf):
    return np.frombuffer(buf, dtype=np.float32, count=N * 5).reshape(N, 5).copy()


def _pose(L, p):
    t = C.c_double()
    v = [C.c_float() for _ in range(4)]
    ok = L.particleFilterLoc_getTagLoc(p, C.byref(t), *[C.byref(q) for q in v])
    return [int(ok), t.value] + [q.value for q in v]


@pytest.mark.parametrize("case", [c for c in _synth.LOC_FUNC_CASES if c["ops"] == ["range"]], ids=lambda c: c["name"])
def test_reference_golden_flow_on_the_struct(case):
    """gen_loc_funcs of make_golden.py, line for line, against the reference's recorded outputs"""
    from slam3d_amd import _lib, particlefilter as pfm
    L = _lib.lib()
    with open(os.path.join(HERE, "golden", "ref_loc_funcs.json")) as fh:
        gold = json.load(fh)["glibc"][case["name"]]
    parts = _synth.loc_particles(N, case["pseed"], case["kind"])
    buf, p = _struct(pfm, L, case["seed"])
    _put_particles(buf, parts)
    buf[OFF_INIT] = b"\x01"                                          # pf.initialized = 1
    bx, by, bz, r, s = case["beacon"]
    L.pfMeasurement_applyRangeLoc(p, bx, by, bz, r, s)
    q = _get_particles(buf)
    assert int(np.sum(q[:, 0] != parts[:, 0])) == gold["n_gated"]
    assert _sha(q[:, 0]) == gold["after_range"]["sha256"]            # the reference's weights, bit for bit
    assert np.array_equal(q[:, 1:], parts[:, 1:])
    # pf.lastT, lastX.. / firstX.. set by hand, then the pose query (make_golden.py does exactly this)
    C.memmove(C.byref(buf, OFF_LAST_T), C.byref(C.c_double(12.5)), 8)
    C.memmove(C.byref(buf, OFF_LAST_X), (C.c_float * 3)(0.25, -0.5, 0.125), 12)
    C.memmove(C.byref(buf, OFF_FIRST_X), (C.c_float * 3)(0.125, 0.25, 0.0625), 12)
    got = _pose(L, p)
    assert got[0] == gold["pose"][0] == 1 and got[1] == gold["pose"][1]
    # the reference adds 10 000 float32 terms serially; here the same products are added in float64
    assert np.allclose(got[2:], gold["pose"][2:], rtol=0, atol=2e-4)
    L.pfx_loc_release(p)


def test_stage_functions_in_the_reference_call_order():
    """particleFilter.c:186-189: pfInit_initTagLoc(pf, ...); pf->initialized = 1; -- the caller's flag counts"""
    from slam3d_amd import _lib, particlefilter as pfm
    L = _lib.lib()
    buf, p = _struct(pfm, L, 31337)
    assert _pose(L, p)[0] == 0
    L.pfInit_initTagLoc(p, 1.0, 2.0, 0.5, 2.8, 0.1)
    assert buf[OFF_INIT] == b"\x00" and _pose(L, p)[0] == 0          # the stage leaves the flag to its caller
    spawned = _get_particles(buf)                                    # ... and pTag[] holds the spawned particles
    assert np.all(spawned[:, 0] == 1.0)
    d = np.linalg.norm(spawned[:, 1:4] - np.array([1.0, 2.0, 0.5], dtype=np.float32), axis=1)
    assert np.all(np.abs(d - 2.8) <= 0.3 + 1e-5)
    buf[OFF_INIT] = b"\x01"
    pose = _pose(L, p)
    assert pose[0] == 1 and abs(pose[2] - 1.0) < 0.2 and abs(pose[3] - 2.0) < 0.2
    # the next deposit works on those particles (an uninitialised filter would respawn them: every weight 1 again)
    L.particleFilterLoc_depositRange(p, 1.0, 2.0, 0.5, 2.5, 0.1)        # keeps about half: no resampling
    assert L.pfx_loc_sync_host(p) == 0
    after = _get_particles(buf)
    assert np.array_equal(after[:, 1:4], spawned[:, 1:4])
    assert 0 < int(np.sum(after[:, 0] != after[:, 0].max())) < N     # some gated, some not: it was a range update
    L.pfx_loc_release(p)


def test_caller_written_particles_reach_every_stage():
    """fill pTag[], run vio + range + resample through the stage functions: same particles as the extension API on an
    upload of the same array"""
    from slam3d_amd import _lib, particlefilter as pfm
    L = _lib.lib()
    case = next(c for c in _synth.LOC_FUNC_CASES if c["name"] == "full_step")
    parts = _synth.loc_particles(N, case["pseed"], case["kind"])
    buf, p = _struct(pfm, L, case["seed"])
    h = pfm.DeviceParticleFilterLoc._borrow(L.pfx_loc_of(p))
    ext = pfm.DeviceParticleFilterLoc(N, seed=h.seed, stream_id=h.stream_id)
    _put_particles(buf, parts)
    buf[OFF_INIT] = b"\x01"
    ext.upload(*[parts[:, k] for k in range(5)])
    bx, by, bz, r, s = case["beacon"]
    L.pfMeasurement_applyVioLoc(p, *case["vio"])
    ext.apply_vio(*case["vio"])
    L.pfMeasurement_applyRangeLoc(p, bx, by, bz, r, s)
    ext.apply_range(bx, by, bz, r, s)
    mid = _get_particles(buf)
    mid[:, 0] *= 0.5                                                 # the caller edits the array between two stages
    _put_particles(buf, mid)
    e = ext.download()
    ext.upload(e[0] * np.float32(0.5), *e[1:])
    L.pfResample_resampleLoc(p, bx, by, bz, r, s)
    ext.resample(bx, by, bz, r, s)
    got, exp = _get_particles(buf), ext.download()
    for k in range(5):
        assert np.array_equal(got[:, k], exp[k])
    L.pfx_loc_release(p)
