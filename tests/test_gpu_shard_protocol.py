"""The sharded deposit protocol (include/pfx.h, pfx_loc_shard_*) driven through the C-ABI with several shards in ONE
process on one GPU: the collectives are plain tensor copies here (all-gather = copy every shard's 8 sums into every
shard's table; shipping = copy a slice of one shard's staging area into another's landing area), so shard counts and
imbalances that the multi-process tests (tests/test_sharded.py: 2-3 ranks over gloo) cannot reach are covered, on the
real kernels.  Reference: the unsharded filter on the same GPU with the same seed -- the noise is keyed by global
particle index and every shard takes the decision from the same gathered sums, so the particles must agree: the
weight totals and the CDF are exact sums, hence weights and z are EQUAL whatever the shard count; the
heading's jitter bandwidth comes from circular sums that associate differently, so theta -- and x, y through the
rotation of the VIO delta -- may differ in their last bits.  One test compares the sharded kernels with the ORACLE
directly."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))

pytestmark = pytest.mark.gpu

SEED = 123456789


class InProcessShards:
    """ShardedLocFilter's deposit logic (slam3d_amd/sharded.py) over a list of GpuShard objects, phase by phase"""

    def __init__(self, n_total, world, seed=SEED, slack=1.0):
        import torch
        from slam3d_amd.sharded import GpuShard, shard_bounds
        self.torch = torch
        self.T = shard_bounds(n_total, world)
        self.world = world
        self.shards = [GpuShard(self.T[g + 1] - self.T[g], n_total, self.T[g], g, world, seed, 0,
                                int((self.T[g + 1] - self.T[g]) * slack) + 1024) for g in range(world)]
        self.outstanding = False
        self.plan = None
        self.moved = 0
        self.branches = []

    def _read_plan(self):
        if self.plan is None:
            plans = [s.plan() for s in self.shards]
            assert all(p == plans[0] for p in plans), "every shard must take the same decision"
            self.plan = plans[0]
            self.branches.append(self.plan[0])
        return self.plan

    def _ship(self, F, deposit):
        from slam3d_amd.sharded import exchange_plan
        self.torch.cuda.synchronize()
        ranges = []
        for g, sh in enumerate(self.shards):
            sends, recvs, _ = exchange_plan(F, self.T, g)
            lo = min(max(F[g], self.T[g]), self.T[g + 1]) - self.T[g]
            hi = max(max(min(F[g + 1], self.T[g + 1]), self.T[g]) - self.T[g], lo)
            ranges.append((lo, hi))
            if deposit and sends:
                sh.stage()
        self.torch.cuda.synchronize()
        for g, sh in enumerate(self.shards):
            sends, recvs, _ = exchange_plan(F, self.T, g)
            for peer, off, cnt in sends:
                # where the peer expects it: its own recv entry for this sender
                psends, precvs, _ = exchange_plan(F, self.T, peer)
                plo = min(max(F[peer], self.T[peer]), self.T[peer + 1]) - self.T[peer]
                phi = max(max(min(F[peer + 1], self.T[peer + 1]), self.T[peer]) - self.T[peer], plo)
                (poff, pcnt), = [(o, c) for h, o, c in precvs if h == g]
                assert pcnt == cnt
                dst = self.shards[peer].landing_view(poff if poff < plo else plo + (poff - phi), cnt)
                dst.copy_(sh.stage_view(off, cnt))
                self.moved += cnt
        self.torch.cuda.synchronize()
        for sh, (lo, hi) in zip(self.shards, ranges):
            sh.land(deposit, lo, hi)
        return ranges

    def settle(self):
        if not self.outstanding:
            return
        self.outstanding = False
        branch, F = self._read_plan()
        if branch:
            for sh in self.shards:
                sh.produce()
            self._ship(F, False)

    def depositVio(self, *a):
        for sh in self.shards:
            sh.deposit_vio(*a)

    def depositRange(self, bx, by, bz, rng, std):
        r = [sh.begin(bx, by, bz, rng, std) for sh in self.shards]
        if r[0] == 1:
            assert all(v == 1 for v in r)
            self.outstanding = False
            return
        for sh in self.shards:
            sh.deposit()
        branch, F = self._read_plan() if self.outstanding else (0, None)
        self.outstanding = False
        if branch:
            ranges = self._ship(F, True)
            for sh, (lo, hi) in zip(self.shards, ranges):
                sh.finish(True, lo, hi)
        else:
            for sh in self.shards:
                sh.finish(False)
        self.torch.cuda.synchronize()
        table = self.torch.cat([sh.local_sums().clone() for sh in self.shards])
        for sh in self.shards:
            sh.all_sums().copy_(table)
        for sh in self.shards:
            sh.decide()
        self.outstanding = True
        self.plan = None

    def depositRssi(self, bx, by, bz, rssi):
        self.depositRange(bx, by, bz, 1.5, 0.5)

    def getTagLoc(self):
        self.settle()
        sums = np.sum([sh.pose_sums() for sh in self.shards], axis=0)
        return self.shards[0].pose_from_sums(sums)

    def download(self):
        self.settle()
        parts = [sh.download() for sh in self.shards]
        return [np.concatenate([p[k] for p in parts]) for k in range(5)]


def _compare(got, ref, n, tol=1e-6):
    for k in (0, 3):
        assert np.array_equal(got[k], ref[k]), k
    for k in (1, 2, 4):
        d = np.abs(got[k] - ref[k])
        if k == 4:
            d = np.minimum(d, np.abs(d - 6.2831855))
        assert d.max() <= 4 * tol, (k, float(d.max()))


@pytest.mark.parametrize("direct", ["1", "0"], ids=["destination-owned", "source-owned"])
@pytest.mark.parametrize("world,n,pose_every", [(1, 50000, 3), (2, 50001, 1), (4, 100003, 4), (7, 200000, 5), (16, 70000, 2),
                                                (2, 700001, 4), (3, 1500007, 6)])
def test_in_process_shards_match_unsharded(world, n, pose_every, direct, monkeypatch):
    """direct: the fused pass of a shard is destination-owned (k_loc_deposit_shard + k_loc_stage_foreign, plan records by
    k_loc_shard_plan; the default) or source-owned (round 1's k_loc_deposit_sharded, PFX_SHARD_DIRECT=0, read when a shard is
    configured).  The two largest cases have shards of more than one chunk of 256 tiles."""
    from slam3d_amd import particlefilter as pfm
    import sharded_worker as W
    monkeypatch.setenv("PFX_SHARD_DIRECT", direct)
    ev = W.events(16)
    g = pfm.DeviceParticleFilterLoc(n, seed=SEED)
    ref_branches = []
    ref_poses = W.drive(g, ev, lambda f: ref_branches.append(int(f.last_stats()["branch"])), pose_every)
    s = InProcessShards(n, world)
    poses = W.drive(s, ev, None, pose_every)
    got = s.download()
    assert np.abs(poses[:, 2:] - ref_poses[:, 2:]).max() <= 1e-4
    _compare(got, g.download(), n)
    # decisions: one per deposit after the spawn (the last one is read by the final download)
    assert s.branches == ref_branches[1:]
    assert 0 < sum(s.branches) < len(s.branches)
    if world > 1:
        assert s.moved > 0, "no offspring ever crossed a shard boundary"


@pytest.mark.parametrize("world", [1, 4, 8])
def test_sharded_kernels_against_the_oracle(ora, world):
    """The sharded HIP filter against the CPU oracle itself (not against another HIP filter): identical uploaded
    particles, four deposits of which at least two resample (narrow gates on a wide cloud); those after a resampling run
    fused with it, offspring shipped between the shards.  Same decisions, same ancestors -- z, which is the ancestor's
    z plus 0.1 x a Box-Muller normal keyed by the global index, agrees to 5e-6 for EVERY particle, which a single
    differing ancestor would break -- weights equal, poses to 1e-4."""
    from slam3d_amd import particlefilter as pfm
    n = 40001
    rs = np.random.RandomState(8)
    x, y, z = (rs.uniform(-3, 3, n).astype(np.float32) for _ in range(3))
    th = rs.uniform(0, 6.28, n).astype(np.float32)
    w = np.where(rs.uniform(size=n) < 0.2, 1.0, 0.1).astype(np.float32)
    s = InProcessShards(n, world, seed=31, slack=2.0)
    o = ora.OracleLoc(n, rng_mode=ora.RNG_PHILOX, seed=31, acc=ora.ACC_F64)
    for k, sh in enumerate(s.shards):
        a, b = s.T[k], s.T[k + 1]
        up = pfm.DeviceParticleFilterLoc._borrow(sh._h)
        up.upload(w[a:b], x[a:b], y[a:b], z[a:b], th[a:b])
        up.epoch = 9
    o.set_particles(w, x, y, z, th)
    o.set_initialized(1)
    o.epoch = 9
    deposits = [(1.0, (0.0, 0.0, 0.0), (0.5, 0.2, 0.1, 3.0, 0.2)), (1.2, (0.1, -0.05, 0.02), (-1.0, 0.7, 0.0, 2.5, 0.1)),
                (1.4, (0.15, -0.02, 0.03), (0.3, 0.3, 0.0, 3.2, 0.3)), (1.6, (0.2, 0.0, 0.03), (1.0, -0.5, 0.2, 2.0, 0.1))]
    for f in (s, o):
        f.depositVio(0.9, 0.0, 0.0, 0.0, 0.0)
    ob = []
    for t, p, r in deposits:
        for f in (s, o):
            f.depositVio(t, p[0], p[1], p[2], 0.0)
            f.depositRange(*r)
        ob.append(int(o.last_stats()["branch"]))
    pa, pb = s.getTagLoc(), o.getTagLoc()
    got, exp = s.download(), o.arrays()
    assert s.branches == ob and sum(ob) >= 2, (s.branches, ob)
    assert np.abs(np.array(pa[2:], dtype=np.float64) - np.array(pb[2:], dtype=np.float64)).max() <= 1e-4
    assert np.array_equal(got[0], exp[0])
    for k in range(1, 4):
        assert np.abs(got[k] - exp[k]).max() <= 5e-6, k
    if world > 1:
        assert s.moved > 0


@pytest.mark.parametrize("n", [40000, 1200003])
def test_concentrated_weight_one_shard_feeds_all(n):
    """All the weight sits in shard 1 of 4: it produces every offspring, three quarters of them foreign (staged below
    and above its own range); the other shards produce nothing and receive everything.  With slack 3.0 the staging
    areas hold it.  Run once with the resampling materialised by a pose query and once fused into the next deposit.
    (The larger size: shards of more than one chunk of 256 tiles.)"""
    from slam3d_amd import particlefilter as pfm
    world = 4
    rs = np.random.RandomState(3)
    x, y, z = (rs.uniform(-3, 3, n).astype(np.float32) for _ in range(3))
    th = rs.uniform(0, 6.28, n).astype(np.float32)
    w = np.full(n, 1e-30, dtype=np.float32)
    w[n // 4 + 100: n // 2 - 100] = 1.0
    for settle_first in (True, False):
        g = pfm.DeviceParticleFilterLoc(n, seed=77)
        s = InProcessShards(n, world, seed=77, slack=3.0)
        g.upload(w, x, y, z, th)
        for k, sh in enumerate(s.shards):
            a, b = s.T[k], s.T[k + 1]
            up = pfm.DeviceParticleFilterLoc._borrow(sh._h)
            up.upload(w[a:b], x[a:b], y[a:b], z[a:b], th[a:b])
        for f in (g, s):
            f.depositVio(1.0, 0.0, 0.0, 0.0, 0.0)
        g.epoch = 5
        for sh in s.shards:
            pfm.DeviceParticleFilterLoc._borrow(sh._h).epoch = 5
        for f in (g, s):
            f.depositRange(0.5, 0.2, 0.1, 3.0, 0.2)          # ESS = n / 4: resamples
            if settle_first:
                f.getTagLoc()
            f.depositVio(1.2, 0.1, -0.05, 0.02, 0.0)
            f.depositRange(-1.0, 0.7, 0.0, 2.5, 0.1)
            f.depositRssi(0.3, 0.3, 0.0, -60)
        assert s.branches[0] == 1
        assert s.moved >= n // 2
        pa, pb = g.getTagLoc(), s.getTagLoc()
        assert np.abs(np.array(pa[2:], dtype=np.float64) - np.array(pb[2:], dtype=np.float64)).max() <= 1e-4
        _compare(s.download(), g.download(), n)


def test_reading_while_a_decision_is_outstanding_fails_loudly():
    from slam3d_amd import _lib
    s = InProcessShards(30000, 2)
    s.depositVio(1.0, 0.0, 0.0, 0.0, 0.0)
    s.depositRange(1.0, 1.0, 0.0, 2.0, 0.1)                    # spawn
    s.depositVio(1.1, 0.05, 0.0, 0.0, 0.0)
    s.depositRange(0.5, 0.5, 0.0, 1.5, 0.1)                    # a decision is now outstanding and unread
    with pytest.raises(_lib.PfxError):
        s.shards[0].download()
    with pytest.raises(_lib.PfxError):
        s.shards[0].pose_sums()
    s.settle()
    s.shards[0].download()
