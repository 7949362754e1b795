"""Host-side mirror of the reference's Python module ``particlefilter``
(particlefilter/cython/particlefilter_wrapper.pyx:27-49): class ``ParticleFilterLoc`` with
``depositVio / depositRange / depositRssi / getTagLoc`` and the module function ``setSeed``,
same argument meaning and return shapes -- over libparticlefilter.so (HIP kernels) instead of
the C sources compiled into the extension.

Two classes share that interface:

* ``ParticleFilterLoc``        -- drop-in: owns a reference-layout ``particleFilterLoc_t`` (400 056 B)
  and calls the reference's own entry points ``particleFilterLoc_*`` exactly as the Cython wrapper
  does.  N is PF_N_TAG_LOC (10 000) unless the environment variable PF_HIP_N_TAG_LOC overrides it.
* ``DeviceParticleFilterLoc`` -- the pfx_* extension: particle count, device, seed chosen at run
  time; adds state upload/download and the single stages of a step.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import numpy as np

from . import _lib
from ._lib import PfxError, check

PF_N_TAG_LOC = 10000
SIZEOF_LOC = 400056          # sizeof(particleFilterLoc_t), include/particleFilter.h

_seed: Optional[int] = None


def setSeed(seed) -> None:
    """particlefilter_wrapper.pyx:48-49 -> particleFilterSeed_set (particleFilter.c:26-30)."""
    global _seed
    _seed = int(seed) & 0xFFFFFFFF
    _lib.lib().particleFilterSeed_set(_seed)


def _pose_tuple(ok, t, x, y, z, th):
    return (int(ok), t.value, np.float32(x.value), np.float32(y.value), np.float32(z.value), np.float32(th.value))


class ParticleFilterLoc:
    """The reference wrapper's class, over the reference-layout C entry points."""

    def __init__(self) -> None:
        self._L = _lib.lib()
        self._buf = C.create_string_buffer(SIZEOF_LOC)      # cdef particleFilterLoc_t pf (wrapper :28)
        self._p = C.cast(self._buf, C.c_void_p)
        self._L.particleFilterLoc_init(self._p)              # __cinit__ (wrapper :30-31)

    def __del__(self):
        try:
            if getattr(self, "_p", None) is not None:
                self._L.pfx_loc_release(self._p)
                self._p = None
        except Exception:
            pass

    def depositVio(self, t, x, y, z, dist) -> None:
        self._L.particleFilterLoc_depositVio(self._p, t, x, y, z, dist)

    def depositRange(self, bx, by, bz, range, stdRange) -> None:      # noqa: A002 (name kept from the wrapper)
        self._L.particleFilterLoc_depositRange(self._p, bx, by, bz, range, stdRange)

    def depositRssi(self, bx, by, bz, rssi) -> None:
        self._L.particleFilterLoc_depositRssi(self._p, bx, by, bz, int(rssi))

    def getTagLoc(self) -> Tuple[int, float, np.float32, np.float32, np.float32, np.float32]:
        t = C.c_double()
        x, y, z, th = C.c_float(), C.c_float(), C.c_float(), C.c_float()
        ok = self._L.particleFilterLoc_getTagLoc(self._p, C.byref(t), C.byref(x), C.byref(y), C.byref(z), C.byref(th))
        return _pose_tuple(ok, t, x, y, z, th)

    # -- additions
    def device_filter(self) -> "DeviceParticleFilterLoc":
        h = self._L.pfx_loc_of(self._p)
        if not h:
            raise PfxError("struct is not bound to a device filter")
        return DeviceParticleFilterLoc._borrow(h)

    def particles(self) -> np.ndarray:
        """(PF_N_TAG_LOC, 5) float32 copy of pTag[] (w, x, y, z, theta), refreshed from HBM."""
        check(self._L.pfx_loc_sync_host(self._p), "pfx_loc_sync_host")
        return np.frombuffer(self._buf, dtype=np.float32, count=PF_N_TAG_LOC * 5).reshape(PF_N_TAG_LOC, 5).copy()


def _fptr(a):
    return a.ctypes.data_as(C.c_void_p)


class DeviceArrayView:
    """library-owned device memory behind the CUDA array interface (version 3; also what HIP builds of torch / cupy read)"""

    def __init__(self, ptr: int, shape, typestr: str, strides=None):
        self.ptr, self.shape = int(ptr), tuple(int(v) for v in shape)
        self.__cuda_array_interface__ = {"shape": self.shape, "typestr": typestr, "data": (self.ptr, False), "version": 3,
                                         "strides": None if strides is None else tuple(int(v) for v in strides), "stream": None}


class ParticleViews:
    def __init__(self, w, x, y, z, theta, wscale=1.0):
        self.w, self.x, self.y, self.z, self.theta, self.wscale = w, x, y, z, theta, float(wscale)

    def __iter__(self):
        return iter((self.w, self.x, self.y, self.z, self.theta))


class DeviceParticleFilterLoc:
    """particleFilterLoc with runtime N on a chosen GPU (pfx_loc_*, include/pfx.h)."""

    def __init__(self, n: int = PF_N_TAG_LOC, seed: Optional[int] = None, stream_id: int = 0, device: int = 0):
        self._L = _lib.lib()
        if seed is None:
            seed = _seed if _seed is not None else 0
        self._h = self._L.pfx_loc_create(int(n), int(seed) & 0xFFFFFFFF, int(stream_id), int(device))
        if not self._h:
            raise PfxError(f"pfx_loc_create: {_lib.last_error()}")
        self._own = True
        self.n = int(n)

    @classmethod
    def _borrow(cls, handle) -> "DeviceParticleFilterLoc":
        self = cls.__new__(cls)
        self._L = _lib.lib()
        self._h = handle
        self._own = False
        self.n = int(self._L.pfx_loc_n(handle))
        return self

    def close(self):
        if getattr(self, "_h", None) and self._own:
            self._L.pfx_loc_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- the wrapper interface
    def depositVio(self, t, x, y, z, dist) -> None:
        self._L.pfx_loc_deposit_vio(self._h, t, x, y, z, dist)

    def depositRange(self, bx, by, bz, range, stdRange) -> None:      # noqa: A002
        check(self._L.pfx_loc_deposit_range(self._h, bx, by, bz, range, stdRange), "pfx_loc_deposit_range")

    def depositRssi(self, bx, by, bz, rssi) -> None:
        check(self._L.pfx_loc_deposit_rssi(self._h, bx, by, bz, int(rssi)), "pfx_loc_deposit_rssi")

    def getTagLoc(self):
        t = C.c_double()
        x, y, z, th = C.c_float(), C.c_float(), C.c_float(), C.c_float()
        ok = check(self._L.pfx_loc_get_tag_loc(self._h, C.byref(t), C.byref(x), C.byref(y), C.byref(z), C.byref(th)),
                   "pfx_loc_get_tag_loc")
        return _pose_tuple(ok, t, x, y, z, th)

    # -- control
    def reset(self, seed: int, stream_id: int = 0):
        check(self._L.pfx_loc_reset(self._h, int(seed) & 0xFFFFFFFF, int(stream_id)), "pfx_loc_reset")

    def sync(self):
        check(self._L.pfx_loc_sync(self._h), "pfx_loc_sync")

    def set_stream(self, hip_stream: int):
        check(self._L.pfx_loc_set_stream(self._h, C.c_void_p(hip_stream)), "pfx_loc_set_stream")

    def set_force_branch(self, force: int):
        check(self._L.pfx_loc_set_force_branch(self._h, int(force)), "pfx_loc_set_force_branch")

    def set_deferred(self, enable: bool):
        """Deferred resampling (default on): see pfx_loc_set_deferred in include/pfx.h."""
        check(self._L.pfx_loc_set_deferred(self._h, int(bool(enable))), "pfx_loc_set_deferred")

    @property
    def initialized(self) -> bool:
        return bool(self._L.pfx_loc_initialized(self._h))

    @property
    def seed(self) -> int:
        return int(self._L.pfx_loc_seed(self._h))

    @property
    def stream_id(self) -> int:
        return int(self._L.pfx_loc_stream_id(self._h))

    @property
    def epoch(self) -> int:
        return int(self._L.pfx_loc_epoch(self._h))

    @epoch.setter
    def epoch(self, e: int):
        self._L.pfx_loc_set_epoch(self._h, int(e))

    # -- state
    def upload(self, w, x, y, z, theta, initialized: bool = True):
        arrs = [np.ascontiguousarray(a, dtype=np.float32) for a in (w, x, y, z, theta)]
        for a in arrs:
            if a.shape != (self.n,):
                raise ValueError(f"expected {self.n} particles, got {a.shape}")
        check(self._L.pfx_loc_upload(self._h, *[_fptr(a) for a in arrs], self.n, int(initialized)), "pfx_loc_upload")

    def download(self):
        out = [np.empty(self.n, dtype=np.float32) for _ in range(5)]
        check(self._L.pfx_loc_download(self._h, *[_fptr(a) for a in out], self.n), "pfx_loc_download")
        return tuple(out)

    def device_arrays(self):
        ptrs = (C.c_void_p * 5)()
        ws = C.c_float()
        check(self._L.pfx_loc_device_arrays(self._h, ptrs, C.byref(ws)), "pfx_loc_device_arrays")
        return [int(p) for p in ptrs], float(ws.value)

    def arrays_view(self) -> "ParticleViews":
        """Zero-copy views of the current particles in HBM: five objects with __cuda_array_interface__ (float32, n
        elements; torch.as_tensor(v, device="cuda") / cupy.asarray(v) wrap them without a copy) and the pending
        normalisation factor: true weight = w * wscale (pfResample.c:80-82 is applied lazily).  A pending resampling is
        materialised first and the filter's stream is synchronised.  Valid until the next deposit, upload or reset on
        this filter (a resampling deposit switches to the other buffer of the ping-pong pair)."""
        ptrs, ws = self.device_arrays()
        return ParticleViews(*[DeviceArrayView(p, (self.n,), "<f4") for p in ptrs], wscale=ws)

    # -- the stages of a step
    def apply_vio(self, dt, dx, dy, dz, ddist):
        check(self._L.pfx_loc_apply_vio(self._h, dt, dx, dy, dz, ddist), "pfx_loc_apply_vio")

    def apply_range(self, bx, by, bz, range, stdRange):               # noqa: A002
        check(self._L.pfx_loc_apply_range(self._h, bx, by, bz, range, stdRange), "pfx_loc_apply_range")

    def resample(self, bx, by, bz, range, stdRange):                  # noqa: A002
        check(self._L.pfx_loc_resample(self._h, bx, by, bz, range, stdRange), "pfx_loc_resample")

    def spawn_all(self, bx, by, bz, range, stdRange):                 # noqa: A002
        check(self._L.pfx_loc_spawn_all(self._h, bx, by, bz, range, stdRange), "pfx_loc_spawn_all")

    def last_stats(self) -> dict:
        buf = (C.c_double * 12)()
        check(self._L.pfx_loc_last_stats(self._h, buf), "pfx_loc_last_stats")
        keys = ["branch", "numSpawn", "s", "ss", "csum", "ssum", "ess", "htheta", "rStart", "rStep", "m", "S"]
        return dict(zip(keys, list(buf)))

    def record_ancestors(self, enable: bool = True):
        check(self._L.pfx_loc_record_ancestors(self._h, int(enable)), "pfx_loc_record_ancestors")

    def last_ancestors(self) -> np.ndarray:
        out = np.empty(self.n, dtype=np.int32)
        check(self._L.pfx_loc_last_ancestors(self._h, _fptr(out), self.n), "pfx_loc_last_ancestors")
        return out

    # -- kernel timing
    def profile_begin(self, max_records: int = 4096):
        check(self._L.pfx_loc_profile_begin(self._h, int(max_records)), "pfx_loc_profile_begin")

    def profile_only(self, *kinds: str):
        """time only these kernel classes (names from KINDS); no argument = all"""
        mask = 0xFFFFFFFF if not kinds else sum(1 << self.KINDS.index(k) for k in kinds)
        check(self._L.pfx_loc_profile_mask(self._h, mask), "pfx_loc_profile_mask")

    def profile_stride(self, stride: int):
        """time launch 0, stride, 2*stride, ... of every class only"""
        check(self._L.pfx_loc_profile_stride(self._h, int(stride)), "pfx_loc_profile_stride")

    KINDS = ["stream", "decide", "resample", "spawn", "pose", "stream_static", "deposit", "deposit_static"]

    def profile_records(self, cap: int = 1 << 16):
        """(kind, ms) per recorded launch, in launch order; kind indexes KINDS."""
        kind = np.empty(cap, dtype=np.int32)
        ms = np.empty(cap, dtype=np.float32)
        k = check(self._L.pfx_loc_profile_records(self._h, _fptr(kind), _fptr(ms), cap), "pfx_loc_profile_records")
        return kind[:k].copy(), ms[:k].copy()

    def profile_read(self) -> dict:
        ms = (C.c_double * len(self.KINDS))()
        cnt = (C.c_int64 * len(self.KINDS))()
        check(self._L.pfx_loc_profile_read(self._h, ms, cnt), "pfx_loc_profile_read")
        return {k: {"ms": ms[i], "launches": int(cnt[i])} for i, k in enumerate(self.KINDS)}

    def branch_log(self, cap: int = 65536):
        """branches of the most recent deposits (1 resampled, 0 normalised) and the total count"""
        out = np.empty(cap, dtype=np.uint8)
        total = C.c_int64()
        k = check(self._L.pfx_loc_branch_log(self._h, _fptr(out), cap, C.byref(total)), "pfx_loc_branch_log")
        return out[:k].copy(), int(total.value)


class DeviceLocBatch:
    """F independent particleFilterLoc filters of n particles each on one GPU, every call one set of launches for the
    whole batch (pfx_loc_batch_*, include/pfx.h).  The methods are the wrapper's (depositVio / depositRange /
    depositRssi / getTagLoc) over arrays of F entries; `mask` (F booleans) selects the filters a call applies to."""

    def __init__(self, num_filters: int, n: int = PF_N_TAG_LOC, seed: Optional[int] = None, seeds=None, stream_ids=None,
                 device: int = 0):
        self._L = _lib.lib()
        if seed is None:
            seed = _seed if _seed is not None else 0
        self.F, self.n = int(num_filters), int(n)
        sa = None if seeds is None else np.ascontiguousarray(seeds, dtype=np.uint32)
        ia = None if stream_ids is None else np.ascontiguousarray(stream_ids, dtype=np.uint32)
        self._h = self._L.pfx_loc_batch_create(self.F, self.n, None if sa is None else _fptr(sa), int(seed) & 0xFFFFFFFF,
                                               None if ia is None else _fptr(ia), int(device))
        if not self._h:
            raise PfxError(f"pfx_loc_batch_create: {_lib.last_error()}")

    def close(self):
        if getattr(self, "_h", None):
            self._L.pfx_loc_batch_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _f32(self, a):
        a = np.ascontiguousarray(np.broadcast_to(np.asarray(a, dtype=np.float32), (self.F,)))
        return a

    def _mask(self, mask):
        if mask is None:
            return None, None
        m = np.ascontiguousarray(np.asarray(mask).astype(np.uint8))
        if m.shape != (self.F,):
            raise ValueError("mask must have one entry per filter")
        return m, _fptr(m)

    def seed_of(self, f: int) -> int:
        return int(self._L.pfx_loc_batch_seed(self._h, int(f)))

    def stream_id_of(self, f: int) -> int:
        return int(self._L.pfx_loc_batch_stream_id(self._h, int(f)))

    def set_force_branch(self, force: int):
        check(self._L.pfx_loc_batch_set_force_branch(self._h, int(force)), "pfx_loc_batch_set_force_branch")

    def sync(self):
        check(self._L.pfx_loc_batch_sync(self._h), "pfx_loc_batch_sync")

    def deposit_vio(self, t, x, y, z, dist=None, mask=None):
        ta = np.ascontiguousarray(np.broadcast_to(np.asarray(t, dtype=np.float64), (self.F,)))
        xa, ya, za = self._f32(x), self._f32(y), self._f32(z)
        da = None if dist is None else self._f32(dist)
        m, mp = self._mask(mask)
        check(self._L.pfx_loc_batch_deposit_vio(self._h, mp, _fptr(ta), _fptr(xa), _fptr(ya), _fptr(za),
                                                None if da is None else _fptr(da)), "pfx_loc_batch_deposit_vio")

    def deposit_range(self, bx, by, bz, range, stdRange, mask=None):   # noqa: A002
        a = [self._f32(v) for v in (bx, by, bz, range, stdRange)]
        m, mp = self._mask(mask)
        check(self._L.pfx_loc_batch_deposit_range(self._h, mp, *[_fptr(v) for v in a]), "pfx_loc_batch_deposit_range")

    def deposit_rssi(self, bx, by, bz, rssi=None, mask=None):
        a = [self._f32(v) for v in (bx, by, bz)]
        m, mp = self._mask(mask)
        check(self._L.pfx_loc_batch_deposit_rssi(self._h, mp, *[_fptr(v) for v in a], None), "pfx_loc_batch_deposit_rssi")

    depositVio, depositRange, depositRssi = deposit_vio, deposit_range, deposit_rssi

    def get_tag_loc(self):
        """(ok uint8[F], t float64[F], x, y, z, theta float32[F]); entries of uninitialised filters are 0"""
        ok = np.zeros(self.F, dtype=np.uint8)
        t = np.zeros(self.F, dtype=np.float64)
        out = [np.zeros(self.F, dtype=np.float32) for _ in range(4)]
        check(self._L.pfx_loc_batch_get_tag_loc(self._h, _fptr(ok), _fptr(t), *[_fptr(a) for a in out]), "pfx_loc_batch_get_tag_loc")
        return (ok, t, *out)

    getTagLoc = get_tag_loc

    def download(self, f: int):
        out = [np.empty(self.n, dtype=np.float32) for _ in range(5)]
        check(self._L.pfx_loc_batch_download(self._h, int(f), *[_fptr(a) for a in out], self.n), "pfx_loc_batch_download")
        return tuple(out)

    def last_stats(self, f: int) -> dict:
        buf = (C.c_double * 12)()
        check(self._L.pfx_loc_batch_last_stats(self._h, int(f), buf), "pfx_loc_batch_last_stats")
        keys = ["branch", "numSpawn", "s", "ss", "csum", "ssum", "ess", "htheta", "rStart", "rStep", "m", "S"]
        return dict(zip(keys, list(buf)))

    def branch_counts(self, per_filter: bool = False):
        """(deposits decided, of those resampled) since creation, summed over the batch -- or two int64 arrays"""
        d, r = C.c_int64(), C.c_int64()
        if per_filter:
            pd, pr = np.zeros(self.F, dtype=np.int64), np.zeros(self.F, dtype=np.int64)
            check(self._L.pfx_loc_batch_branch_counts(self._h, C.byref(d), C.byref(r), _fptr(pd), _fptr(pr)), "pfx_loc_batch_branch_counts")
            return pd, pr
        check(self._L.pfx_loc_batch_branch_counts(self._h, C.byref(d), C.byref(r), None, None), "pfx_loc_batch_branch_counts")
        return int(d.value), int(r.value)


def resample_indices_f32(cdf, r_start, r_step, device: int = 0) -> np.ndarray:
    """anc[i] = first j with r_start + r_step*i < cdf[j] (float32, pfResample.c:70-72), on the GPU."""
    cdf = np.ascontiguousarray(cdf, dtype=np.float32)
    out = np.empty(cdf.shape[0], dtype=np.int32)
    check(_lib.lib().pfx_resample_indices_f32(_fptr(cdf), cdf.shape[0], np.float32(r_start), np.float32(r_step),
                                              _fptr(out), device), "pfx_resample_indices_f32")
    return out


def resample_indices_f64(cdf, r_start, r_step, device: int = 0) -> np.ndarray:
    cdf = np.ascontiguousarray(cdf, dtype=np.float64)
    out = np.empty(cdf.shape[0], dtype=np.int32)
    check(_lib.lib().pfx_resample_indices_f64(_fptr(cdf), cdf.shape[0], float(r_start), float(r_step),
                                              _fptr(out), device), "pfx_resample_indices_f64")
    return out


def device_count() -> int:
    return int(_lib.lib().pfx_device_count())
