"""Zero-copy device views of the particle state: objects with __cuda_array_interface__ that torch
wraps without a copy must show exactly what download() returns -- after a normalise-only deposit (pending factor in
wscale / row_scale) and after a resampling one (other ping-pong buffer) -- for the Loc filter and the SLAM batch."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_loc_arrays_view_equals_download():
    import torch
    from slam3d_amd import particlefilter as pfm
    n = 100003
    f = pfm.DeviceParticleFilterLoc(n, seed=5)
    f.depositVio(1.0, 0.0, 0.0, 0.0, 0.0)
    f.depositRange(1.0, 1.0, 0.0, 2.0, 0.1)                 # spawn
    for force, rng in ((0, 2.1), (1, 1.9), (0, 2.0)):
        f.set_force_branch(force)
        f.depositVio(1.1, 0.05, 0.0, 0.0, 0.0)
        f.depositRange(0.5, 0.5, 0.0, rng, 0.1)
        v = f.arrays_view()
        t = [torch.as_tensor(a, device="cuda:0") for a in v]
        assert all(x.shape == (n,) and x.dtype == torch.float32 for x in t)
        assert t[0].data_ptr() == v.w.ptr                   # a view, not a copy
        got = [x.cpu().numpy() for x in t]
        exp = f.download()
        assert int(f.last_stats()["branch"]) == force
        assert (v.wscale != 1.0) == (force == 0)
        assert np.array_equal(got[0] * np.float32(v.wscale), exp[0])
        for k in range(1, 5):
            assert np.array_equal(got[k], exp[k])


def test_slam_batch_arrays_view_equals_download():
    import torch
    from slam3d_amd import slam
    F, nT, nB = 3, 20, 203
    b = slam.DeviceSlamBatch(F, nT, nB, max_bcns=2, seed0=77)
    for _ in range(2):
        b.add_bcn_all()
    ones = np.ones(F, dtype=np.float32)
    for k in range(6):
        b.deposit_tag_vio_batch(np.full(F, 10.0 + 0.1 * k), ones * 0.02 * k, ones * 0.01 * k, ones * 0.0)
        b.deposit_range_batch(np.full(F, k % 2, dtype=np.int32), ones * (2.0 + 0.05 * k), ones * 0.1)
    v = b.arrays_view()
    tag = [torch.as_tensor(a, device="cuda:0").cpu().numpy() for a in v["tag"]]
    bcn = [torch.as_tensor(a, device="cuda:0").cpu().numpy() for a in v["bcn"]]
    rs = torch.as_tensor(v["row_scale"], device="cuda:0").cpu().numpy()
    assert tag[0].shape == (F, nT) and bcn[0].shape == (F, 2, nT, nB) and rs.shape == (F, 2, nT)
    assert (rs != 1.0).any()                                # some rows carry a pending normalisation factor
    for f in range(F):
        exp = b.download_tag(f)
        for k in range(5):
            assert np.array_equal(tag[k][f], exp[k])
        for slot in range(2):
            exp = b.download_bcn(f, slot)
            assert np.array_equal((bcn[0][f, slot] * rs[f, slot][:, None]).reshape(-1), np.asarray(exp[0]).reshape(-1))
            for k in range(1, 5):
                assert np.array_equal(bcn[k][f, slot].reshape(-1), np.asarray(exp[k]).reshape(-1))
