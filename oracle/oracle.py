"""ctypes binding of oracle/libpf_oracle.so -- TEST INFRASTRUCTURE ONLY.

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg;
never from slam3d_amd/.  The classes mirror the reference's Cython wrapper
interface (particlefilter/cython/particlefilter_wrapper.pyx:27-49) so the same
replay loop drives the oracle and the HIP path.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libpf_oracle.so")

RNG_GLIBC, RNG_MUSL, RNG_PHILOX = 0, 1, 2
ACC_F32, ACC_F64 = 0, 1
P_VIO, P_RSTART, P_JITTER, P_SPAWN = 1, 2, 3, 4

_lib = None


def build(force: bool = False) -> str:
    """Compile the oracle (and oracle/_ref when the reference tree is present)."""
    if force or not os.path.exists(_LIB_PATH) or \
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "pf_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "--no-print-directory"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        build()
    L = C.CDLL(_LIB_PATH)
    vp, f, d, i, u32, u64 = C.c_void_p, C.c_float, C.c_double, C.c_int, C.c_uint32, C.c_uint64
    pf_, pd_, pi_ = C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_int)

    def sig(name, res, *args):
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = list(args)

    sig("ora_rng_new", vp, i, C.c_uint)
    sig("ora_rng_free", None, vp)
    sig("ora_rng_state", C.c_uint, vp)
    sig("ora_rng_set_state", None, vp, C.c_uint)
    sig("ora_rng_begin", None, vp, u32, u32, u32, u64)
    sig("ora_rng_next31", i, vp)
    sig("ora_rng_uniform", f, vp)
    sig("ora_rng_normal2", None, vp, pf_, pf_)
    sig("ora_rng_sphere", None, vp, pf_, pf_, pf_, f, f)
    sig("ora_philox4x32_10", None, C.POINTER(u32), C.POINTER(u32), C.POINTER(u32))

    sig("ora_loc_new", vp, i, vp, i)
    sig("ora_loc_free", None, vp)
    sig("ora_loc_init", None, vp)
    sig("ora_loc_n", i, vp)
    sig("ora_loc_array", pf_, vp, i)
    sig("ora_loc_set_initialized", None, vp, i)
    sig("ora_loc_initialized", i, vp)
    sig("ora_loc_set_epoch", None, vp, u32)
    sig("ora_loc_epoch", u32, vp)
    sig("ora_loc_deposit_vio", None, vp, d, f, f, f, f)
    sig("ora_loc_deposit_range", None, vp, f, f, f, f, f)
    sig("ora_loc_deposit_rssi", None, vp, f, f, f, i)
    sig("ora_loc_get_tag_loc", i, vp, pd_, pf_, pf_, pf_, pf_)
    sig("ora_loc_apply_vio", None, vp, f, f, f, f, f)
    sig("ora_loc_apply_range", None, vp, f, f, f, f, f)
    sig("ora_loc_resample", None, vp, f, f, f, f, f)
    sig("ora_loc_spawn_all", None, vp, f, f, f, f, f)
    sig("ora_loc_last_stats", None, vp, pd_)
    sig("ora_loc_last_ancestors", pi_, vp)
    sig("ora_loc_last_cdf", pd_, vp)
    sig("ora_loc_set_index_offset", None, vp, u64)
    sig("ora_offspring_block", None, vp, u32, u32, C.c_int64, i, pi_, pf_, pf_, pf_, pf_, f, i, f, f, f, f, f,
        pf_, pf_, pf_, pf_)
    sig("ora_decide", None, vp, u32, u32, d, d, d, d, C.c_int64, f, i, pd_)
    sig("ora_clamp_events", C.c_long)
    sig("ora_resample_indices_f32", None, pf_, i, f, f, pi_)
    sig("ora_resample_indices_f64", None, pd_, i, d, d, pi_)

    sig("ora_slam_new", vp, i, i, vp, i)
    sig("ora_slam_free", None, vp)
    sig("ora_slam_init", None, vp)
    sig("ora_slam_array", pf_, vp, i)
    sig("ora_slam_last_tag_branch", i, vp)
    sig("ora_bcn_new", vp, vp, u32)
    sig("ora_bcn_free", None, vp)
    sig("ora_bcn_add", None, vp)
    sig("ora_bcn_array", pf_, vp, i)
    sig("ora_bcn_initialized", i, vp)
    sig("ora_slam_set_epoch", None, vp, u32)
    sig("ora_slam_deposit_tag_vio", None, vp, d, f, f, f, f)
    sig("ora_slam_deposit_bcn_vio", None, vp, d, f, f, f, f)
    sig("ora_slam_deposit_range", None, vp, vp, f, f, C.POINTER(vp), i)
    sig("ora_slam_deposit_rssi", None, vp, vp, i, C.POINTER(vp), i)
    sig("ora_slam_get_tag_loc", i, vp, pd_, pf_, pf_, pf_, pf_)
    sig("ora_slam_get_bcn_loc", i, vp, vp, pd_, pf_, pf_, pf_, pf_)
    _lib = L
    return L


def philox4x32_10(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().ora_philox4x32_10(c, k, o)
    return [int(v) for v in o]


class OracleRng:
    def __init__(self, mode: int, seed: int):
        self.h = lib().ora_rng_new(mode, seed & 0xFFFFFFFF)
        self.mode = mode

    def __del__(self):
        if getattr(self, "h", None):
            lib().ora_rng_free(self.h)
            self.h = None

    @property
    def state(self) -> int:
        return lib().ora_rng_state(self.h)

    def begin(self, stream, epoch, purpose, index):
        lib().ora_rng_begin(self.h, stream, epoch, purpose, index)

    def next31(self) -> int:
        return lib().ora_rng_next31(self.h)

    def uniform(self) -> np.float32:
        return np.float32(lib().ora_rng_uniform(self.h))

    def normal2(self):
        a, b = C.c_float(), C.c_float()
        lib().ora_rng_normal2(self.h, C.byref(a), C.byref(b))
        return np.float32(a.value), np.float32(b.value)

    def sphere(self, rng_, std):
        x, y, z = C.c_float(), C.c_float(), C.c_float()
        lib().ora_rng_sphere(self.h, C.byref(x), C.byref(y), C.byref(z), rng_, std)
        return np.float32(x.value), np.float32(y.value), np.float32(z.value)


def _view(ptr, n):
    return np.ctypeslib.as_array(ptr, shape=(n,))


class OracleLoc:
    """particleFilterLoc with runtime N.  Interface = the reference's ParticleFilterLoc wrapper."""

    def __init__(self, n: int = 10000, rng: OracleRng | None = None, acc: int = ACC_F32,
                 rng_mode: int = RNG_GLIBC, seed: int = 0):
        self.rng = rng if rng is not None else OracleRng(rng_mode, seed)
        self.n = n
        self.h = lib().ora_loc_new(n, self.rng.h, acc)

    def __del__(self):
        if getattr(self, "h", None):
            lib().ora_loc_free(self.h)
            self.h = None

    # -- wrapper interface (particlefilter_wrapper.pyx:33-46)
    def depositVio(self, t, x, y, z, dist):
        lib().ora_loc_deposit_vio(self.h, t, x, y, z, dist)

    def depositRange(self, bx, by, bz, rng_, std):
        lib().ora_loc_deposit_range(self.h, bx, by, bz, rng_, std)

    def depositRssi(self, bx, by, bz, rssi):
        lib().ora_loc_deposit_rssi(self.h, bx, by, bz, int(rssi))

    def getTagLoc(self):
        t = C.c_double()
        x, y, z, th = C.c_float(), C.c_float(), C.c_float(), C.c_float()
        ok = lib().ora_loc_get_tag_loc(self.h, C.byref(t), C.byref(x), C.byref(y), C.byref(z), C.byref(th))
        return (ok, t.value, np.float32(x.value), np.float32(y.value), np.float32(z.value), np.float32(th.value))

    # -- pieces and state access
    def arrays(self):
        """Live views (w, x, y, z, theta) into the oracle's particle storage."""
        return tuple(_view(lib().ora_loc_array(self.h, k), self.n) for k in range(5))

    def set_particles(self, w, x, y, z, th):
        for dst, src in zip(self.arrays(), (w, x, y, z, th)):
            dst[:] = np.asarray(src, dtype=np.float32)

    def set_initialized(self, v=1):
        lib().ora_loc_set_initialized(self.h, int(v))

    @property
    def initialized(self):
        return bool(lib().ora_loc_initialized(self.h))

    @property
    def epoch(self):
        return lib().ora_loc_epoch(self.h)

    @epoch.setter
    def epoch(self, e):
        lib().ora_loc_set_epoch(self.h, e)

    def apply_vio(self, dt, dx, dy, dz, ddist):
        lib().ora_loc_apply_vio(self.h, dt, dx, dy, dz, ddist)

    def apply_range(self, bx, by, bz, rng_, std):
        lib().ora_loc_apply_range(self.h, bx, by, bz, rng_, std)

    def resample(self, bx, by, bz, rng_, std):
        lib().ora_loc_resample(self.h, bx, by, bz, rng_, std)

    def spawn_all(self, bx, by, bz, rng_, std):
        lib().ora_loc_spawn_all(self.h, bx, by, bz, rng_, std)

    def last_stats(self):
        buf = (C.c_double * 12)()
        lib().ora_loc_last_stats(self.h, buf)
        keys = ["branch", "numSpawn", "s", "ss", "csum", "ssum", "ess", "htheta", "rStart", "rStep", "m", "S"]
        return dict(zip(keys, list(buf)[:12]))

    def last_ancestors(self):
        return _view(lib().ora_loc_last_ancestors(self.h), self.n).copy()

    def last_cdf(self):
        return _view(lib().ora_loc_last_cdf(self.h), self.n).copy()


def clamp_events() -> int:
    return int(lib().ora_clamp_events())


def resample_indices_f32(cdf, r_start, r_step):
    cdf = np.ascontiguousarray(cdf, dtype=np.float32)
    out = np.empty(cdf.shape[0], dtype=np.int32)
    lib().ora_resample_indices_f32(cdf.ctypes.data_as(C.POINTER(C.c_float)), cdf.shape[0],
                                   np.float32(r_start), np.float32(r_step),
                                   out.ctypes.data_as(C.POINTER(C.c_int)))
    return out


def resample_indices_f64(cdf, r_start, r_step):
    cdf = np.ascontiguousarray(cdf, dtype=np.float64)
    out = np.empty(cdf.shape[0], dtype=np.int32)
    lib().ora_resample_indices_f64(cdf.ctypes.data_as(C.POINTER(C.c_double)), cdf.shape[0],
                                   float(r_start), float(r_step), out.ctypes.data_as(C.POINTER(C.c_int)))
    return out


class OracleBcn:
    def __init__(self, slam: "OracleSlam", stream: int):
        self.slam = slam
        self.stream = stream
        self.h = lib().ora_bcn_new(slam.h, stream)
        lib().ora_bcn_add(self.h)

    def __del__(self):
        if getattr(self, "h", None):
            lib().ora_bcn_free(self.h)
            self.h = None

    def arrays(self):
        n = self.slam.n_tag * self.slam.n_bcn
        return tuple(_view(lib().ora_bcn_array(self.h, k), n).reshape(self.slam.n_tag, self.slam.n_bcn)
                     for k in range(5))

    @property
    def initialized(self):
        return bool(lib().ora_bcn_initialized(self.h))

    def depositVio(self, t, x, y, z, dist):
        lib().ora_slam_deposit_bcn_vio(self.h, t, x, y, z, dist)


class OracleSlam:
    """particleFilterSlam with runtime particle counts (reference: 100 tag x 1000 per beacon)."""

    def __init__(self, n_tag: int = 100, n_bcn: int = 1000, rng: OracleRng | None = None,
                 acc: int = ACC_F32, rng_mode: int = RNG_GLIBC, seed: int = 0):
        self.rng = rng if rng is not None else OracleRng(rng_mode, seed)
        self.n_tag, self.n_bcn = n_tag, n_bcn
        self.h = lib().ora_slam_new(n_tag, n_bcn, self.rng.h, acc)
        self.bcns = []

    def __del__(self):
        if getattr(self, "h", None):
            self.bcns = []
            lib().ora_slam_free(self.h)
            self.h = None

    @property
    def last_tag_resampled(self) -> bool:
        return bool(lib().ora_slam_last_tag_branch(self.h))

    def addBcn(self) -> OracleBcn:
        b = OracleBcn(self, 1 + len(self.bcns))
        self.bcns.append(b)
        return b

    def arrays(self):
        return tuple(_view(lib().ora_slam_array(self.h, k), self.n_tag) for k in range(5))

    def _all(self):
        arr = (C.c_void_p * len(self.bcns))(*[b.h for b in self.bcns])
        return arr, len(self.bcns)

    def depositTagVio(self, t, x, y, z, dist):
        lib().ora_slam_deposit_tag_vio(self.h, t, x, y, z, dist)

    def depositRange(self, bcn: OracleBcn, rng_, std):
        arr, n = self._all()
        lib().ora_slam_deposit_range(self.h, bcn.h, rng_, std, arr, n)

    def depositRssi(self, bcn: OracleBcn, rssi):
        arr, n = self._all()
        lib().ora_slam_deposit_rssi(self.h, bcn.h, int(rssi), arr, n)

    def getTagLoc(self):
        t = C.c_double()
        x, y, z, th = C.c_float(), C.c_float(), C.c_float(), C.c_float()
        ok = lib().ora_slam_get_tag_loc(self.h, C.byref(t), C.byref(x), C.byref(y), C.byref(z), C.byref(th))
        return (ok, t.value, np.float32(x.value), np.float32(y.value), np.float32(z.value), np.float32(th.value))

    def getBcnLoc(self, bcn: OracleBcn):
        t = C.c_double()
        x, y, z, th = C.c_float(), C.c_float(), C.c_float(), C.c_float()
        ok = lib().ora_slam_get_bcn_loc(self.h, bcn.h, C.byref(t), C.byref(x), C.byref(y), C.byref(z), C.byref(th))
        return (ok, t.value, np.float32(x.value), np.float32(y.value), np.float32(z.value), np.float32(th.value))
