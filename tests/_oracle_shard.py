"""CPU stand-in for slam3d_amd.sharded.GpuShard, built on the oracle (TEST INFRASTRUCTURE).

It restates, in numpy + oracle calls, what the local half of a sharded deposit does on the GPU (f64 sums,
global-index Philox, CDF span [base_g, base_{g+1}), offspring range [F_g, F_{g+1}), staging layout), so
that the HOST logic of ShardedLocFilter -- shard bounds, exchange plan, collectives, placement -- can be run
with world_size > 1 under gloo on a machine without a GPU.
"""
import ctypes as C

import numpy as np
import torch

from oracle import oracle as ora

f32 = np.float32


def cdf_weight(w):
    """oracle/pf_oracle.c ora_cdf_weight on a float32 array: nearest multiple of 2^-20 (ties to even), as float64"""
    return (np.rint(w.astype(np.float32) * f32(1048576.0)) * f32(9.5367431640625e-07)).astype(np.float64)


def _p(a, t=C.c_float):
    return a.ctypes.data_as(C.POINTER(t))


def first_offspring_at(b, r_start, r_step, n):
    """count of i in [0,n) with r_start + r_step*i < b (same function as the device's first_offspring_at)"""
    if not (r_step > 0.0 and r_step < 1e300):
        return 0
    q = np.ceil((b - r_start) / r_step)
    if not q == q:
        return 0
    f = 0 if q < 0 else (n if q > n else int(q))
    for _ in range(8):
        if f > 0 and not (r_start + r_step * float(f - 1) < b):
            f -= 1
    for _ in range(8):
        if f < n and (r_start + r_step * float(f) < b):
            f += 1
    return f


class OracleShard:
    def __init__(self, n_local, n_global, self_begin, rank, world, seed, stage_capacity):
        self.n, self.n_global, self.self_begin = n_local, n_global, self_begin
        self.rank, self.world = rank, world
        self.rng = ora.OracleRng(ora.RNG_PHILOX, seed)
        self.pf = ora.OracleLoc(n_local, rng=self.rng, acc=ora.ACC_F64)
        ora.lib().ora_loc_set_index_offset(self.pf.h, self_begin)
        self.stage_capacity = stage_capacity
        self._local = torch.zeros(8, dtype=torch.float64)
        self._all = torch.zeros(world * 8, dtype=torch.float64)
        self._stage = torch.zeros(4 * stage_capacity, dtype=torch.float32)      # (x, y, z, theta) records to send
        self._landing = torch.zeros(4 * stage_capacity, dtype=torch.float32)    # records received
        self.pose_local = None
        self.wscale = f32(1.0)
        self.force = -1
        self.branch = 1
        self.epoch = 0
        self.first = [0.0] * 5

    # VIO bookkeeping: reuse the oracle's accumulator through a 1-particle filter (commit on deposit)
    def deposit_vio(self, t, x, y, z, dist):
        self.pf.depositVio(t, x, y, z, dist)

    def set_force_branch(self, force):
        self.force = int(force)

    def _commit(self):
        # the oracle commits VIO inside deposit_range; here the pieces are driven one by one, so take the
        # delta from the public pose query trick: run apply_vio with the accumulator's delta
        L = ora.lib()
        if not hasattr(L, "_shard_sigs"):
            L.ora_loc_commit_vio.restype = None
            L.ora_loc_commit_vio.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
            L._shard_sigs = True
        d = (C.c_float * 5)()
        L.ora_loc_commit_vio(self.pf.h, d)
        return [f32(v) for v in d]

    def begin(self, bx, by, bz, rng, std):
        self.epoch += 1
        self.pf.epoch = self.epoch
        self._vio = self._commit()
        self._args = (bx, by, bz, rng, std)
        if not self.pf.initialized:
            self.pf.spawn_all(bx, by, bz, rng, std)
            self.pf.set_initialized(1)
            self.branch = 1
            self.pose_local = None
            self._outstanding = False
            return 1
        return 0

    def deposit(self):
        """acts on the outstanding decision as the device kernel does: after "resample" the offspring are produced
        (the deposit on them follows in land(), for the whole shard at once: same per-particle arithmetic), after
        "normalise" -- or with nothing outstanding -- the streaming pass"""
        if getattr(self, "_outstanding", False) and self.branch:
            self._outstanding = False
            self._resample()
        else:
            self._outstanding = False
            self.stream()

    def stage(self):
        """the device library produces the foreign offspring of a fused deposit here; _resample() has staged them already"""

    def stream(self):
        """VIO motion + gate over all local particles and the 8 local sums"""
        bx, by, bz, rng, std = self._args
        w, x, y, z, th = self.pf.arrays()
        w *= self.wscale                                  # pending normalisation (pfResample.c:80-82)
        self.wscale = f32(1.0)
        self.pf.apply_vio(*self._vio)
        self.pf.apply_range(bx, by, bz, rng, std)
        w64 = cdf_weight(w)
        c, s = np.cos(th).astype(np.float32), np.sin(th).astype(np.float32)
        sums = [w64.sum(), (w * w).astype(np.float64).sum(), (w * c).astype(np.float64).sum(),
                (w * s).astype(np.float64).sum(), (w * x).astype(np.float64).sum(), (w * y).astype(np.float64).sum(),
                (w * z).astype(np.float64).sum(), 0.0]
        self._local[:] = torch.tensor(sums, dtype=torch.float64)

    def local_sums(self):
        return self._local

    def all_sums(self):
        return self._all

    def decide(self):
        rng = self._args[3]
        a = self._all.numpy().reshape(self.world, 8)
        tot = np.zeros(7)
        base = [0.0]
        for g in range(self.world):
            tot += a[g, :7]
            base.append(float(tot[0]))
        out = (C.c_double * 6)()
        ora.lib().ora_decide(self.rng.h, 0, self.epoch, tot[0], tot[1], tot[2], tot[3], self.n_global, rng, self.force, out)
        self.branch, self.num_spawn, self.htheta = int(out[0]), int(out[1]), f32(out[2])
        self.r_start, self.r_step = out[4], out[5]
        self.base = base
        if self.branch:
            F = [0] + [first_offspring_at(base[g], self.r_start, self.r_step, self.n_global) for g in range(1, self.world)] \
                + [self.n_global]
        else:
            self.wscale = f32(out[3])
            l = self._local.numpy()
            self.pose_local = np.array([l[0], l[4], l[5], l[6], l[2], l[3]])
            F = [0] * (self.world + 1)
        self.F = F
        self._outstanding = True
        if self.branch:
            self._rs = (self.epoch, self._args)           # what the resampling will need

    def plan(self):
        return self.branch, self.F

    def produce(self):
        self._outstanding = False
        self._resample()

    def _resample(self):
        """own-range offspring into the arrays (weight 1), foreign ones staged as records"""
        epoch, (bx, by, bz, rng, std) = self._rs
        w, x, y, z, th = self.pf.arrays()
        cdf = self.base[self.rank] + np.cumsum(cdf_weight(w))
        ob, oe = self.F[self.rank], self.F[self.rank + 1]
        cnt = oe - ob
        i = np.arange(ob, oe, dtype=np.float64)
        u = self.r_start + self.r_step * i
        anc = np.minimum(np.searchsorted(cdf, u, side="right"), self.n - 1).astype(np.int32)
        ox, oy, oz, ot = (np.empty(cnt, dtype=np.float32) for _ in range(4))
        ora.lib().ora_offspring_block(self.rng.h, 0, epoch, ob, cnt, _p(anc, C.c_int), _p(x), _p(y), _p(z), _p(th),
                                      self.htheta, self.num_spawn, bx, by, bz, rng, std, _p(ox), _p(oy), _p(oz), _p(ot))
        # producer-side pose partials (w = 1)
        self.pose_local = np.array([cnt, ox.astype(np.float64).sum(), oy.astype(np.float64).sum(), oz.astype(np.float64).sum(),
                                    np.cos(ot).astype(np.float64).sum(), np.sin(ot).astype(np.float64).sum()])
        sb, se = self.self_begin, self.self_begin + self.n
        gi = np.arange(ob, oe)
        own = (gi >= sb) & (gi < se)
        below = gi < sb
        nb = int(below.sum())
        new = [np.zeros(self.n, dtype=np.float32) for _ in range(5)]
        new[0][gi[own] - sb] = 1.0
        st = self._stage.numpy().reshape(-1, 4)
        na = int((gi >= se).sum())
        for k, o in enumerate((ox, oy, oz, ot)):
            new[k + 1][gi[own] - sb] = o[own]
            st[:nb, k] = o[below]
            st[nb:nb + na, k] = o[gi >= se]
        self._new = new
        # install as the current arrays
        for dst, src in zip(self.pf.arrays(), new):
            dst[:] = src

    def stage_view(self, off, count):
        return self._stage[4 * off: 4 * (off + count)]

    def landing_view(self, off, count):
        return self._landing[4 * off: 4 * (off + count)]

    def land(self, deposit, lo, hi):
        """records of the landing area -> local elements [0, lo) and [hi, n), weight 1; deposit: then the deposit just
        begun on every local particle (the produced ones were left undeposited by produce())"""
        arrs = self.pf.arrays()
        rec = self._landing.numpy().reshape(-1, 4)
        na = self.n - hi
        arrs[0][:lo] = 1.0
        arrs[0][hi:] = 1.0
        for k in range(4):
            arrs[k + 1][:lo] = rec[:lo, k]
            arrs[k + 1][hi:] = rec[lo:lo + na, k]
        if deposit:
            self.stream()
        else:
            w, x, y, z, th = arrs
            self.pose_local = np.array([cdf_weight(w).sum(), x.astype(np.float64).sum(), y.astype(np.float64).sum(),
                                        z.astype(np.float64).sum(), np.cos(th).astype(np.float64).sum(),
                                        np.sin(th).astype(np.float64).sum()])

    def finish(self, pieces, lo=0, hi=0):
        pass                                              # land(deposit=True) has left the local sums

    def pose_sums(self):
        if self.pose_local is None:
            # freshly spawned
            w, x, y, z, th = self.pf.arrays()
            return np.array([self.n, x.astype(np.float64).sum(), y.astype(np.float64).sum(), z.astype(np.float64).sum(),
                             np.cos(th).astype(np.float64).sum(), np.sin(th).astype(np.float64).sum()])
        return self.pose_local.copy()

    def pose_from_sums(self, sums):
        L = ora.lib()
        L.ora_loc_pose_from_sums.restype = C.c_int
        L.ora_loc_pose_from_sums.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)] + [C.POINTER(C.c_float)] * 4
        arr = (C.c_double * 6)(*[float(v) for v in sums])
        t = C.c_double()
        x, y, z, th = C.c_float(), C.c_float(), C.c_float(), C.c_float()
        ok = L.ora_loc_pose_from_sums(self.pf.h, arr, C.byref(t), C.byref(x), C.byref(y), C.byref(z), C.byref(th))
        return (int(ok), t.value, f32(x.value), f32(y.value), f32(z.value), f32(th.value))

    def download(self):
        w, x, y, z, th = (a.copy() for a in self.pf.arrays())
        return (w * self.wscale, x, y, z, th)
