"""particleFilterSlam on the GPU: host-side mirror of the reference's SLAM interface.

The reference's Cython module wraps only the Loc filter (particlefilter_wrapper.pyx:27-49); its SLAM API is
reached from C (csvslam/csvslam.c:54-85) and JNI (ArSlam/app/src/main/jni/slam3d-jni.c:15-83).  The classes
here follow that C interface one to one, with the wrapper's naming style:

* ``ParticleFilterSlam`` / ``Bcn``  -- drop-in: own reference-layout ``particleFilterSlam_t`` (4 056 B) and
  ``bcn_t`` (2 020 056 B) structs and call ``particleFilterSlam_*`` exactly as csvslam.c does.
* ``DeviceSlamBatch``               -- the pfx_slam_* extension: a batch of independent SLAM filters with
  runtime particle counts on one GPU (BASELINE.json configs[4]); ``DeviceParticleFilterSlam`` is the batch of one
  with the per-filter interface (depositTagVio / depositBcnVio / depositRange / depositRssi / getTagLoc / getBcnLoc).
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence

import numpy as np

from . import _lib
from ._lib import PfxError, check
from . import particlefilter as _pfm

PF_N_TAG_SLAM = 100
PF_N_BCN = 1000
SIZEOF_SLAM = 4056           # sizeof(particleFilterSlam_t), include/particleFilter.h
SIZEOF_BCN = 2020056         # sizeof(bcn_t)


def _pose(ok, t, x, y, z, th):
    return (int(ok), t.value, np.float32(x.value), np.float32(y.value), np.float32(z.value), np.float32(th.value))


def _outs():
    return C.c_double(), C.c_float(), C.c_float(), C.c_float(), C.c_float()


def _fp(a):
    return a.ctypes.data_as(C.c_void_p)


# ------------------------------------------------------------------ drop-in, reference-layout structs
class Bcn:
    """bcn_t owned by the caller, as csvslam.c:36 (`static bcn_t _bcns[NUM_BCNS]`)."""

    def __init__(self) -> None:
        self._buf = C.create_string_buffer(SIZEOF_BCN)      # zeroed, like static storage
        self._p = C.cast(self._buf, C.c_void_p)
        _lib.lib().particleFilterSlam_addBcn(self._p)       # csvslam.c:56

    @property
    def initialized(self) -> bool:
        return self._buf.raw[SIZEOF_BCN - 56] != 0          # offsetof(bcn_t, initialized)

    def depositVio(self, t, x, y, z, dist) -> None:
        _lib.lib().particleFilterSlam_depositBcnVio(self._p, t, x, y, z, dist)


class ParticleFilterSlam:
    """particleFilterSlam_t + the reference's entry points (particleFilter.h:102-114)."""

    def __init__(self) -> None:
        self._L = _lib.lib()
        self._buf = C.create_string_buffer(SIZEOF_SLAM)
        self._p = C.cast(self._buf, C.c_void_p)
        self._L.particleFilterSlam_init(self._p)
        self.bcns: List[Bcn] = []

    def __del__(self):
        try:
            if getattr(self, "_p", None) is not None:
                self._L.pfx_slam_release(self._p)
                self._p = None
        except Exception:
            pass

    def addBcn(self) -> Bcn:
        b = Bcn()
        self.bcns.append(b)
        return b

    def _all(self):
        return (C.c_void_p * len(self.bcns))(*[b._p.value for b in self.bcns]), len(self.bcns)

    def depositTagVio(self, t, x, y, z, dist) -> None:
        self._L.particleFilterSlam_depositTagVio(self._p, t, x, y, z, dist)

    def depositRange(self, bcn: Bcn, range, stdRange) -> None:       # noqa: A002
        arr, n = self._all()
        self._L.particleFilterSlam_depositRange(self._p, bcn._p, range, stdRange, arr, n)

    def depositRssi(self, bcn: Bcn, rssi) -> None:
        arr, n = self._all()
        self._L.particleFilterSlam_depositRssi(self._p, bcn._p, int(rssi), arr, n)

    def getTagLoc(self):
        t, x, y, z, th = _outs()
        ok = self._L.particleFilterSlam_getTagLoc(self._p, C.byref(t), C.byref(x), C.byref(y), C.byref(z), C.byref(th))
        return _pose(ok, t, x, y, z, th)

    def getBcnLoc(self, bcn: Bcn):
        t, x, y, z, th = _outs()
        ok = self._L.particleFilterSlam_getBcnLoc(self._p, bcn._p, C.byref(t), C.byref(x), C.byref(y), C.byref(z), C.byref(th))
        return _pose(ok, t, x, y, z, th)

    # -- additions
    def tag_particles(self) -> np.ndarray:
        """(100, 5) float32 copy of pTag[] refreshed from HBM"""
        check(self._L.pfx_slam_sync_host(self._p, None), "pfx_slam_sync_host")
        return np.frombuffer(self._buf, dtype=np.float32, count=PF_N_TAG_SLAM * 5).reshape(PF_N_TAG_SLAM, 5).copy()

    def bcn_particles(self, bcn: Bcn) -> np.ndarray:
        """(100, 1000, 5) float32 copy of bcn.pBcn[][] refreshed from HBM"""
        check(self._L.pfx_slam_sync_host(self._p, bcn._p), "pfx_slam_sync_host")
        n = PF_N_TAG_SLAM * PF_N_BCN
        return np.frombuffer(bcn._buf, dtype=np.float32, count=n * 5).reshape(PF_N_TAG_SLAM, PF_N_BCN, 5).copy()


# ------------------------------------------------------------------ the extension: batches, runtime sizes
class DeviceSlamBatch:
    """`num_filters` independent SLAM filters on one GPU (pfx_slam_*, include/pfx.h)."""

    def __init__(self, num_filters: int, n_tag: int = PF_N_TAG_SLAM, n_bcn: int = PF_N_BCN, max_bcns: int = 12,
                 seeds: Optional[Sequence[int]] = None, seed0: int = 0, device: int = 0):
        self._L = _lib.lib()
        sp = None
        if seeds is not None:
            self._seeds = np.ascontiguousarray(np.asarray(seeds, dtype=np.uint64) & 0xFFFFFFFF, dtype=np.uint32)
            if self._seeds.shape != (num_filters,):
                raise ValueError("one seed per filter")
            sp = _fp(self._seeds)
        self._h = self._L.pfx_slam_create(int(num_filters), int(n_tag), int(n_bcn), int(max_bcns), sp, int(seed0) & 0xFFFFFFFF,
                                          int(device))
        if not self._h:
            raise PfxError(f"pfx_slam_create: {_lib.last_error()}")
        self.F, self.n_tag, self.n_bcn, self.max_bcns = int(num_filters), int(n_tag), int(n_bcn), int(max_bcns)

    def close(self):
        if getattr(self, "_h", None):
            self._L.pfx_slam_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        check(self._L.pfx_slam_sync(self._h), "pfx_slam_sync")

    def arrays_view(self):
        """Zero-copy views of the whole batch in HBM (objects with __cuda_array_interface__, float32):
        tag  = (w, x, y, z, theta), each of shape (filters, n_tag);
        bcn  = (w, x, y, z, theta), each of shape (filters, max_bcns, n_tag, n_bcn) -- bcn[k][f, slot, i, j] is
               pBcn[i][j] of that beacon (particleFilter.h:86);
        row_scale of shape (filters, max_bcns, n_tag): true beacon weight = w * row_scale of its row (the normalisation
               of pfResample.c:195-199 is applied lazily).
        Rows are padded in memory (strides say so).  Synchronises the stream; valid for the life of the batch."""
        import ctypes as C
        from .particlefilter import DeviceArrayView
        tag, bcn = (C.c_void_p * 5)(), (C.c_void_p * 5)()
        rs = C.c_void_p()
        dims = (C.c_int32 * 6)()
        check(self._L.pfx_slam_device_arrays(self._h, tag, bcn, C.byref(rs), dims), "pfx_slam_device_arrays")
        F, B, nT, nB, tS, rS = [int(v) for v in dims]
        tags = tuple(DeviceArrayView(p, (F, nT), "<f4", (4 * tS, 4)) for p in tag)
        bcns = tuple(DeviceArrayView(p, (F, B, nT, nB), "<f4", (4 * B * nT * rS, 4 * nT * rS, 4 * rS, 4)) for p in bcn)
        return {"tag": tags, "bcn": bcns, "row_scale": DeviceArrayView(rs.value, (F, B, nT), "<f4")}

    def row_census(self):
        """(rows normalised, rows resampled, filters whose tags resampled, filters) of the last deposit"""
        import ctypes as C
        out = (C.c_int64 * 4)()
        check(self._L.pfx_slam_row_census(self._h, out), "pfx_slam_row_census")
        return tuple(int(v) for v in out)

    # -- per filter (the reference's operations)
    def reset(self, f: int, seed: int):
        check(self._L.pfx_slam_reset(self._h, f, int(seed) & 0xFFFFFFFF), "pfx_slam_reset")

    def add_bcn(self, f: int) -> int:
        return check(self._L.pfx_slam_add_bcn(self._h, f), "pfx_slam_add_bcn")

    def add_bcn_all(self) -> int:
        """one more beacon for every filter; returns its slot (the same everywhere when used consistently)"""
        slots = {self.add_bcn(f) for f in range(self.F)}
        if len(slots) != 1:
            raise PfxError("filters of the batch hold different numbers of beacons")
        return slots.pop()

    def bcn_initialized(self, f: int, slot: int) -> bool:
        return bool(self._L.pfx_slam_bcn_initialized(self._h, f, slot))

    def epoch(self, f: int) -> int:
        return int(self._L.pfx_slam_epoch(self._h, f))

    def set_epoch(self, f: int, e: int):
        self._L.pfx_slam_set_epoch(self._h, f, int(e))

    def deposit_tag_vio(self, f, t, x, y, z, dist=0.0):
        check(self._L.pfx_slam_deposit_tag_vio(self._h, f, t, x, y, z, dist), "pfx_slam_deposit_tag_vio")

    def deposit_bcn_vio(self, f, slot, t, x, y, z, dist=0.0):
        check(self._L.pfx_slam_deposit_bcn_vio(self._h, f, slot, t, x, y, z, dist), "pfx_slam_deposit_bcn_vio")

    def deposit_range(self, f, slot, range, stdRange):                # noqa: A002
        check(self._L.pfx_slam_deposit_range(self._h, f, slot, range, stdRange), "pfx_slam_deposit_range")

    def deposit_rssi(self, f, slot, rssi):
        check(self._L.pfx_slam_deposit_rssi(self._h, f, slot, int(rssi)), "pfx_slam_deposit_rssi")

    def get_tag_loc(self, f):
        t, x, y, z, th = _outs()
        ok = check(self._L.pfx_slam_get_tag_loc(self._h, f, C.byref(t), C.byref(x), C.byref(y), C.byref(z), C.byref(th)),
                   "pfx_slam_get_tag_loc")
        return _pose(ok, t, x, y, z, th)

    def get_bcn_loc(self, f, slot):
        t, x, y, z, th = _outs()
        ok = check(self._L.pfx_slam_get_bcn_loc(self._h, f, slot, C.byref(t), C.byref(x), C.byref(y), C.byref(z), C.byref(th)),
                   "pfx_slam_get_bcn_loc")
        return _pose(ok, t, x, y, z, th)

    def last_tag_branch(self, f) -> int:
        return check(self._L.pfx_slam_last_tag_branch(self._h, f), "pfx_slam_last_tag_branch")

    # -- whole batch
    def deposit_tag_vio_batch(self, t, x, y, z, dist=None):
        t = np.ascontiguousarray(t, dtype=np.float64)
        a = [np.ascontiguousarray(v, dtype=np.float32) for v in (x, y, z)]
        dp = _fp(np.ascontiguousarray(dist, dtype=np.float32)) if dist is not None else None
        check(self._L.pfx_slam_deposit_tag_vio_batch(self._h, _fp(t), _fp(a[0]), _fp(a[1]), _fp(a[2]), dp),
              "pfx_slam_deposit_tag_vio_batch")

    def deposit_range_batch(self, slot, range, stdRange):             # noqa: A002
        s = np.ascontiguousarray(slot, dtype=np.int32)
        r = np.ascontiguousarray(range, dtype=np.float32)
        sd = np.ascontiguousarray(stdRange, dtype=np.float32)
        check(self._L.pfx_slam_deposit_range_batch(self._h, _fp(s), _fp(r), _fp(sd)), "pfx_slam_deposit_range_batch")

    def get_tag_loc_batch(self):
        t = np.empty(self.F, dtype=np.float64)
        o = [np.empty(self.F, dtype=np.float32) for _ in range(4)]
        check(self._L.pfx_slam_get_tag_loc_batch(self._h, _fp(t), *[_fp(a) for a in o]), "pfx_slam_get_tag_loc_batch")
        return t, o[0], o[1], o[2], o[3]

    def get_bcn_loc_batch(self, slot):
        s = np.ascontiguousarray(slot, dtype=np.int32)
        ok = np.zeros(self.F, dtype=np.uint8)
        t = np.zeros(self.F, dtype=np.float64)
        o = [np.zeros(self.F, dtype=np.float32) for _ in range(4)]
        check(self._L.pfx_slam_get_bcn_loc_batch(self._h, _fp(s), _fp(ok), _fp(t), *[_fp(a) for a in o]),
              "pfx_slam_get_bcn_loc_batch")
        return ok, t, o[0], o[1], o[2], o[3]

    # -- state
    def download_tag(self, f):
        out = [np.empty(self.n_tag, dtype=np.float32) for _ in range(5)]
        check(self._L.pfx_slam_download_tag(self._h, f, *[_fp(a) for a in out]), "pfx_slam_download_tag")
        return tuple(out)

    def upload_tag(self, f, w, x, y, z, theta):
        a = [np.ascontiguousarray(v, dtype=np.float32) for v in (w, x, y, z, theta)]
        check(self._L.pfx_slam_upload_tag(self._h, f, *[_fp(v) for v in a]), "pfx_slam_upload_tag")

    def download_bcn(self, f, slot):
        out = [np.empty((self.n_tag, self.n_bcn), dtype=np.float32) for _ in range(5)]
        check(self._L.pfx_slam_download_bcn(self._h, f, slot, *[_fp(a) for a in out]), "pfx_slam_download_bcn")
        return tuple(out)

    def upload_bcn(self, f, slot, w, x, y, z, theta, initialized=True):
        a = [np.ascontiguousarray(v, dtype=np.float32).reshape(self.n_tag, self.n_bcn) for v in (w, x, y, z, theta)]
        check(self._L.pfx_slam_upload_bcn(self._h, f, slot, *[_fp(v) for v in a], int(initialized)), "pfx_slam_upload_bcn")


class _DevBcn:
    def __init__(self, owner: "DeviceParticleFilterSlam", slot: int):
        self.owner, self.slot = owner, slot

    @property
    def initialized(self) -> bool:
        return self.owner.batch.bcn_initialized(0, self.slot)

    def depositVio(self, t, x, y, z, dist) -> None:
        self.owner.batch.deposit_bcn_vio(0, self.slot, t, x, y, z, dist)

    def arrays(self):
        return self.owner.batch.download_bcn(0, self.slot)


class DeviceParticleFilterSlam:
    """One SLAM filter with runtime sizes, with the interface of ParticleFilterSlam above."""

    def __init__(self, n_tag: int = PF_N_TAG_SLAM, n_bcn: int = PF_N_BCN, seed: Optional[int] = None, max_bcns: int = 12,
                 device: int = 0):
        if seed is None:
            seed = _pfm._seed if _pfm._seed is not None else 0
        self.batch = DeviceSlamBatch(1, n_tag, n_bcn, max_bcns, seeds=[seed], device=device)
        self.bcns: List[_DevBcn] = []

    def addBcn(self) -> _DevBcn:
        b = _DevBcn(self, self.batch.add_bcn(0))
        self.bcns.append(b)
        return b

    def depositTagVio(self, t, x, y, z, dist) -> None:
        self.batch.deposit_tag_vio(0, t, x, y, z, dist)

    def depositRange(self, bcn: _DevBcn, range, stdRange) -> None:   # noqa: A002
        self.batch.deposit_range(0, bcn.slot, range, stdRange)

    def depositRssi(self, bcn: _DevBcn, rssi) -> None:
        self.batch.deposit_rssi(0, bcn.slot, rssi)

    def getTagLoc(self):
        return self.batch.get_tag_loc(0)

    def getBcnLoc(self, bcn: _DevBcn):
        return self.batch.get_bcn_loc(0, bcn.slot)

    @property
    def last_tag_resampled(self) -> bool:
        return bool(self.batch.last_tag_branch(0))

    def arrays(self):
        return self.batch.download_tag(0)

    def set_tag_weights(self, w):
        cur = list(self.batch.download_tag(0))
        cur[0] = np.ascontiguousarray(w, dtype=np.float32)
        self.batch.upload_tag(0, *cur)
