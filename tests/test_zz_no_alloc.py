"""pfx_pose_sums_noalloc (include/pfx.h): the pose reduction over a caller's device arrays with its per-tile partial sums
in the library's static device array instead of an allocated workspace, for callers that cannot allocate.

Checked through the C-ABI against the existing entry point that runs the same reduction with the filter's own allocated
partial-sum buffer, pfx_loc_shard_pose_sums (what pfx_loc_get_tag_loc reads): same kernels' code, same association, so
the six float64 sums must be EQUAL -- on two calls in a row, the second on a smaller input (the scratch array keeps the
first call's partials beyond the second call's tiles; none of them may leak into the result)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _sums_existing(f):
    out = (C.c_double * 6)()
    from slam3d_amd import _lib
    _lib.check(_lib.lib().pfx_loc_shard_pose_sums(f._h, out), "pfx_loc_shard_pose_sums")
    return np.array(list(out))


def _sums_noalloc(f, out_t, stream=None):
    from slam3d_amd import _lib
    L = _lib.lib()
    ptrs = (C.c_void_p * 5)()
    ws = C.c_float()
    _lib.check(L.pfx_loc_device_arrays(f._h, ptrs, C.byref(ws)), "pfx_loc_device_arrays")
    rc = L.pfx_pose_sums_noalloc(ptrs[0], ptrs[1], ptrs[2], ptrs[3], ptrs[4], f.n, ws.value, out_t.data_ptr(), None, 0)
    _lib.check(rc, "pfx_pose_sums_noalloc")
    import torch
    torch.cuda.synchronize()
    return out_t.cpu().numpy().copy()


def test_pose_sums_noalloc_matches_allocating_entry_point():
    import torch
    from slam3d_amd import _lib, particlefilter as pfm
    if pfm.device_count() < 1:
        pytest.fail("no HIP device visible: the GPU tests must run on an MI355X box")
    out_t = torch.zeros(6, dtype=torch.float64, device="cuda:0")
    rs = np.random.RandomState(3)
    for n in (3000001, 70001):                      # two calls in a row, the second smaller (and a ragged last tile)
        f = pfm.DeviceParticleFilterLoc(n, seed=11)
        w = rs.uniform(0.05, 1.5, n).astype(np.float32)
        x, y, z = (rs.normal(0, 3, n).astype(np.float32) for _ in range(3))
        th = rs.uniform(-6.2, 6.2, n).astype(np.float32)
        f.upload(w, x, y, z, th, initialized=True)
        # a resampling step: the filter's pose sums then come from the per-tile pass both entry points share
        f.set_force_branch(1)
        f.depositVio(1.0, 0.0, 0.0, 0.0, 0.0)
        f.depositRange(0.5, 0.2, 0.1, 3.0, 0.2)
        exp = _sums_existing(f)
        got = _sums_noalloc(f, out_t)
        assert np.array_equal(got, exp), (n, got, exp)
        assert got[0] == float(n)                   # weights are 1 after a resampling step
        # and on weighted particles.  After a normalise-only step the filter's own sums are those of the streaming pass,
        # taken BEFORE the normalisation factor m = N / s (pfResample.c:80-82) is applied, while the caller's arrays
        # are read with it (w * wscale, each product rounded to float32): the pose is the ratio of the sums, so the
        # two are compared as ratios, to float32 rounding of the weights
        f.set_force_branch(0)
        f.depositVio(1.1, 0.01, 0.0, 0.0, 0.0)
        f.depositRange(-1.0, 0.7, 0.0, 2.5, 0.1)
        exp = _sums_existing(f)
        got = _sums_noalloc(f, out_t)
        assert abs(got[0] - n) <= 1e-5 * n           # mean weight 1 after normalisation (float32 factor and products)
        assert np.allclose(got[1:] / got[0], exp[1:] / exp[0], rtol=0, atol=1e-6), (n, got, exp)
    # sizes beyond the static array are refused with a message, nothing is launched
    L = _lib.lib()
    rc = L.pfx_pose_sums_noalloc(1, 1, 1, 1, 1, (1 << 26) + 1, 1.0, out_t.data_ptr(), None, 0)
    assert rc < 0 and b"static scratch" in L.pfx_last_error()
