"""The HIP path against the COMPILED REFERENCE itself, live, on the same inputs -- not through the oracle.

oracle/_ref/libpf_ref_glibc.so is the reference's own sources built by oracle/Makefile in the build container (it travels to
the GPU box; nothing of /root/reference does).  Its stage functions are exported (pfMeasurement.h, pfResample.h), so every
stage whose result does not depend on the random stream can be compared with the GPU stage directly:

* pfMeasurement_applyRangeLoc (pfMeasurement.c:108-125): the gate is a pure function of the particle -- weights EQUAL;
* pfResample_resampleLoc, branch B (pfResample.c:80-82): w *= N / s.  The reference's s is a serial float32 sum, the GPU's an
  exact sum rounded once; on weights that are multiples of 2^-10 both are exact, so m and every weight are EQUAL;
* particleFilterLoc_getTagLoc (particleFilter.c:248-291): float32 serial sums against float64 tile sums -- 1e-4;
* pfMeasurement_applyVioLoc with dt = ddist = 0 (pfMeasurement.c:24-49): the noise is multiplied by an exact zero, what is
  left is the rotation by the heading, glibc's cosf / sinf against the kernel's bounded sincos -- 2e-6.
The stages that draw (spawn, jitter, rStart) are pinned through the oracle's Philox mode, not here.
"""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "libpf_ref_glibc.so")
N = 10000                                   # PF_N_TAG_LOC of the compiled reference (particleFilter.h:18)


class _P(C.Structure):
    _fields_ = [("w", C.c_float), ("x", C.c_float), ("y", C.c_float), ("z", C.c_float), ("theta", C.c_float)]


class _Loc(C.Structure):                    # particleFilter.h:48-64
    _fields_ = [("pTag", _P * N), ("pTagBuf", _P * N), ("initialized", C.c_uint8), ("firstT", C.c_double), ("firstX", C.c_float),
                ("firstY", C.c_float), ("firstZ", C.c_float), ("firstDist", C.c_float), ("lastT", C.c_double), ("lastX", C.c_float),
                ("lastY", C.c_float), ("lastZ", C.c_float), ("lastDist", C.c_float)]


assert C.sizeof(_Loc) == 400056


@pytest.fixture(scope="module")
def ref():
    if not os.path.exists(REF):
        pytest.skip("oracle/_ref/libpf_ref_glibc.so was not built (needs /root/reference at build time)")
    L = C.CDLL(REF)
    f = C.c_float
    L.pfMeasurement_applyRangeLoc.argtypes = [C.c_void_p, f, f, f, f, f]
    L.pfMeasurement_applyVioLoc.argtypes = [C.c_void_p, f, f, f, f, f]
    L.pfResample_resampleLoc.argtypes = [C.c_void_p, f, f, f, f, f]
    L.particleFilterLoc_getTagLoc.argtypes = [C.c_void_p, C.POINTER(C.c_double)] + [C.POINTER(f)] * 4
    L.particleFilterLoc_getTagLoc.restype = C.c_uint8
    L.particleFilterSeed_set.argtypes = [C.c_uint]
    L.particleFilterSeed_set(7)
    return L


def _fill(s, a):
    v = np.frombuffer(s, dtype=np.float32, count=5 * N).reshape(N, 5)
    for k in range(5):
        v[:, k] = a[k]
    s.initialized = 1


def _read(s):
    v = np.frombuffer(s, dtype=np.float32, count=5 * N).reshape(N, 5)
    return [v[:, k].copy() for k in range(5)]


def _cloud(seed, centre, radius, spread=0.35):
    r = np.random.default_rng(seed)
    d = r.normal(size=(N, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rad = radius + r.normal(scale=spread, size=N)           # straddles the 3-sigma shell on both sides
    p = (np.asarray(centre) + d * rad[:, None]).astype(np.float32)
    w = (r.integers(1, 2048, size=N) / 1024.0).astype(np.float32)       # multiples of 2^-10
    th = r.uniform(-6.2, 6.2, size=N).astype(np.float32)
    return [w, p[:, 0].copy(), p[:, 1].copy(), p[:, 2].copy(), th]


@pytest.mark.parametrize("rng_,std", [(4.0, 0.1), (2.5, 0.1), (1.5, 0.5), (7.25, 0.05), (0.3, 0.1), (12.0, 0.3)])
def test_range_gate_equals_the_compiled_reference(ref, rng_, std):
    from slam3d_amd import particlefilter as pfm
    b = (1.0, -2.0, 0.5)
    a = _cloud(int(rng_ * 100), b, rng_, spread=max(0.35, 3.0 * std))
    s = _Loc()
    _fill(s, a)
    ref.pfMeasurement_applyRangeLoc(C.byref(s), b[0], b[1], b[2], rng_, std)
    exp = _read(s)
    g = pfm.DeviceParticleFilterLoc(N, seed=1)
    g.upload(*a)
    g.apply_range(b[0], b[1], b[2], rng_, std)
    got = g.download()
    gated = exp[0] != a[0]
    assert 0.05 * N < gated.sum() < 0.95 * N, "the cloud must straddle the gate"
    assert np.array_equal(got[0], exp[0])
    for k in range(1, 5):
        assert np.array_equal(got[k], exp[k]) and np.array_equal(got[k], a[k])


def test_normalise_branch_equals_the_compiled_reference(ref):
    from slam3d_amd import particlefilter as pfm
    a = _cloud(5, (0.0, 0.0, 0.0), 3.0)
    s = _Loc()
    _fill(s, a)
    ref.pfResample_resampleLoc(C.byref(s), 0.0, 0.0, 0.0, 10.0, 0.1)       # range >= 4: no respawn; ESS/N ~ 0.75: branch B
    exp = _read(s)
    ssum = float(np.sum(a[0].astype(np.float64)))
    assert np.array_equal(exp[0], a[0] * np.float32(np.float32(N) / np.float32(ssum))), "the reference took branch A"
    g = pfm.DeviceParticleFilterLoc(N, seed=1)
    g.upload(*a)
    g.resample(0.0, 0.0, 0.0, 10.0, 0.1)
    st = g.last_stats()
    assert int(st["branch"]) == 0 and st["S"] == ssum
    got = g.download()
    for k in range(5):
        assert np.array_equal(got[k], exp[k]), k


def test_pose_matches_the_compiled_reference(ref):
    from slam3d_amd import particlefilter as pfm
    a = _cloud(9, (2.0, 1.0, 1.5), 0.8)
    a[4] = (0.7 + 0.2 * np.random.default_rng(3).normal(size=N)).astype(np.float32)
    s = _Loc()
    _fill(s, a)
    s.firstT, s.lastT = 3.0, 4.5
    s.firstX, s.firstY, s.firstZ = 0.5, 0.25, 0.0
    s.lastX, s.lastY, s.lastZ = 0.75, 0.125, 0.0625         # an uncommitted VIO delta (particleFilter.c:276-289)
    t = C.c_double()
    o = [C.c_float() for _ in range(4)]
    assert ref.particleFilterLoc_getTagLoc(C.byref(s), C.byref(t), *[C.byref(v) for v in o]) == 1
    g = pfm.DeviceParticleFilterLoc(N, seed=1)
    g.upload(*a)
    g.depositVio(3.0, 0.5, 0.25, 0.0, 0.0)
    g.depositVio(4.5, 0.75, 0.125, 0.0625, 0.0)
    p = g.getTagLoc()
    assert p[0] == 1 and p[1] == t.value
    assert np.abs(np.array(p[2:], dtype=np.float64) - np.array([v.value for v in o], dtype=np.float64)).max() < 1e-4


def test_noise_free_motion_matches_the_compiled_reference(ref):
    from slam3d_amd import particlefilter as pfm
    a = _cloud(11, (0.0, 0.0, 1.0), 2.0)
    s = _Loc()
    _fill(s, a)
    dx, dy, dz = 0.375, -0.21875, 0.0625
    ref.pfMeasurement_applyVioLoc(C.byref(s), 0.0, dx, dy, dz, 0.0)        # sigma_xyz = sigma_theta = 0
    exp = _read(s)
    g = pfm.DeviceParticleFilterLoc(N, seed=1)
    g.upload(*a)
    g.apply_vio(0.0, dx, dy, dz, 0.0)
    got = g.download()
    assert np.array_equal(got[0], exp[0])
    assert np.abs(got[1] - exp[1]).max() <= 2e-6 and np.abs(got[2] - exp[2]).max() <= 2e-6
    assert np.array_equal(got[3], exp[3])                   # z += dz: no transcendental in it
    assert np.array_equal(got[4], exp[4])                   # fmodf(theta + 0, 2 pi)


# ------------------------------------------------------------------ SLAM: a deposit in which nothing is drawn
N_TAG, N_BCN = 100, 1000


class _Slam(C.Structure):                   # particleFilter.h:66-82
    _fields_ = [("pTag", _P * N_TAG), ("pTagBuf", _P * N_TAG)] + _Loc._fields_[2:]


class _Bcn(C.Structure):                    # particleFilter.h:84-100
    _fields_ = [("pBcn", (_P * N_BCN) * N_TAG), ("pBcnBuf", _P * N_BCN)] + _Loc._fields_[2:]


assert C.sizeof(_Slam) == 4056 and C.sizeof(_Bcn) == 2020056


@pytest.mark.parametrize("rng_,std", [(3.5, 0.1), (5.0, 0.2)])
def test_slam_deposit_without_resampling_equals_the_compiled_reference(ref, rng_, std):
    """particleFilterSlam_depositRange (particleFilter.c:193-211) on given particles, no VIO since the last deposit, weights
    even enough that neither the tag particles nor any beacon row resample: what is left of the deposit is the gate of every
    beacon particle against its tag particle (pfMeasurement.c:127-152), the row sums into the tag weights, and the two
    normalisations (pfResample.c:132-138, 195-199) -- nothing is drawn from the random stream (the reference's zero-sigma
    motion passes consume draws and change nothing).  range >= 3, so a gated weight is halved, and weights that are multiples
    of 2^-10 keep every row sum exact in the reference's float32 as in the kernel's exact sums: beacon weights EQUAL, all
    positions and headings EQUAL (untouched), tag weights to 1e-6 (their sum is not exact in float32)."""
    from slam3d_amd import slam
    ref.particleFilterSlam_depositRange.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_int]
    r = np.random.default_rng(int(rng_ * 10))
    tag = [(r.integers(512, 1536, size=N_TAG) / 1024.0).astype(np.float32)] + \
          [r.normal(scale=0.3, size=N_TAG).astype(np.float32) for _ in range(3)] + [r.uniform(-3, 3, size=N_TAG).astype(np.float32)]
    d = r.normal(size=(N_TAG, N_BCN, 3))
    d /= np.linalg.norm(d, axis=2, keepdims=True)
    rad = rng_ + r.normal(scale=max(0.3, 3 * std) * 0.9, size=(N_TAG, N_BCN))      # ~25 % of every row outside the shell
    tp = np.stack(tag[1:4], axis=1)[:, None, :]
    bp = (tp + d * rad[:, :, None]).astype(np.float32)
    bcn = [(r.integers(768, 1280, size=(N_TAG, N_BCN)) / 1024.0).astype(np.float32), bp[:, :, 0].copy(), bp[:, :, 1].copy(),
           bp[:, :, 2].copy(), r.uniform(-3, 3, size=(N_TAG, N_BCN)).astype(np.float32)]
    # the compiled reference
    ref.particleFilterSeed_set(11)
    s, b = _Slam(), _Bcn()
    ref.particleFilterSlam_init(C.byref(s))
    ref.particleFilterSlam_addBcn(C.byref(b))
    tv = np.frombuffer(s, dtype=np.float32, count=5 * N_TAG).reshape(N_TAG, 5)
    bv = np.frombuffer(b, dtype=np.float32, count=5 * N_TAG * N_BCN).reshape(N_TAG, N_BCN, 5)
    for k in range(5):
        tv[:, k] = tag[k]
        bv[:, :, k] = bcn[k]
    b.initialized = 1
    ptrs = (C.c_void_p * 1)(C.addressof(b))
    ref.particleFilterSlam_depositRange(C.byref(s), C.byref(b), rng_, std, ptrs, 1)
    exp_tag = [tv[:, k].copy() for k in range(5)]
    exp_bcn = [bv[:, :, k].copy() for k in range(5)]
    assert np.array_equal(exp_tag[1], tag[1]) and np.array_equal(exp_bcn[1], bcn[1]), "the reference resampled: the case is not the one meant"
    # the GPU
    g = slam.DeviceParticleFilterSlam(N_TAG, N_BCN, seed=5, max_bcns=2)
    gb = g.addBcn()
    g.batch.upload_tag(0, *tag)
    g.batch.upload_bcn(0, gb.slot, *bcn, initialized=True)
    g.depositRange(gb, rng_, std)
    assert not g.last_tag_resampled
    got_tag, got_bcn = g.arrays(), gb.arrays()
    share = float((exp_bcn[0] / exp_bcn[0].sum(axis=1, keepdims=True) < bcn[0] / bcn[0].sum(axis=1, keepdims=True) * 0.8).mean())
    assert 0.05 < share < 0.6, share                        # a good part of every row was gated, and not all of it
    for k in range(5):
        assert np.array_equal(got_bcn[k], exp_bcn[k]), k
    for k in range(1, 5):
        assert np.array_equal(got_tag[k], exp_tag[k]), k
    assert np.abs(got_tag[0] / exp_tag[0] - 1.0).max() < 1e-6


def test_resampling_indices_equal_the_compiled_reference(ref):
    """north_star: "when fed identical uniform draws and weights, resampling indices must be bit-exact" -- against the
    reference's own loop (pfResample.c:41-50, 67-72), run by the compiled reference.  The draw: the reference takes rStart
    from its stream first thing in branch A, so the same seed set twice gives the value once through pfRandom_uniform and
    once inside resampleLoc.  The indices: the particles sit 10 m apart on the x axis and the jitter has sigma 0.1
    (pfInit.c:68-79), so every offspring names its ancestor.  The GPU side is pfx_resample_indices_f32: the reference's
    float32 expression on the reference's float32 CDF."""
    from slam3d_amd import particlefilter as pfm
    r = np.random.default_rng(21)
    w = np.full(N, 2.0 ** -10, dtype=np.float32)
    heavy = r.choice(N, size=N * 3 // 10, replace=False)
    w[heavy] = (r.integers(512, 2048, size=heavy.shape[0]) / 1024.0).astype(np.float32)
    x = (10.0 * np.arange(N)).astype(np.float32)
    a = [w, x, np.zeros(N, dtype=np.float32), np.zeros(N, dtype=np.float32), np.zeros(N, dtype=np.float32)]
    ref.pfRandom_uniform.restype = C.c_float
    ref.particleFilterSeed_set(4711)
    u = np.float32(ref.pfRandom_uniform())
    ref.particleFilterSeed_set(4711)
    s = _Loc()
    _fill(s, a)
    ref.pfResample_resampleLoc(C.byref(s), 0.0, 0.0, 0.0, 10.0, 0.1)       # range >= 4: no respawn
    out = _read(s)
    assert np.all(out[0] == 1.0), "the reference took branch B: the case is not the one meant"
    anc_ref = np.rint(out[1].astype(np.float64) / 10.0).astype(np.int64)
    assert np.abs(out[1] - 10.0 * anc_ref).max() < 1.0
    # the reference's CDF and scalars, in its arithmetic (pfResample.c:41-50, 67-68)
    cdf = np.empty(N, dtype=np.float32)
    acc = np.float32(0.0)
    for i in range(N):
        acc = np.float32(acc + w[i])
        cdf[i] = acc
    r_step = np.float32(np.float32(np.float32(1.0) / np.float32(N)) * acc)         # rStep = invN * s, invN = 1.0f / N (:53,67)
    r_start = np.float32(u * r_step)
    idx = pfm.resample_indices_f32(cdf, float(r_start), float(r_step))
    assert idx.shape == (N,) and np.array_equal(idx.astype(np.int64), anc_ref)
    assert np.unique(anc_ref).shape[0] < N * 0.4            # it was a real selection
