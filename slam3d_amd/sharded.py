"""One particleFilterLoc sharded by particle range over the GPUs of a node (BASELINE.json configs[3]).

No counterpart in the reference (it has no parallelism of any kind); the interface is
still the wrapper's (depositVio / depositRange / depositRssi / getTagLoc), so the same drivers run it.

Rank g holds the global particles [T_g, T_{g+1}), T_g = floor(g*N/G).  Per deposit:

  1. local streaming pass (VIO + gate + sums) and local tile prefix            -- no communication
  2. all-gather of every rank's chunk sums (sum w, w^2, w cos, w sin, w x, w y, w z per 262 144 particles: 2 kB per
     rank at 8 M particles per shard)                                          -- RCCL, latency-bound
  3. every rank takes the SAME decision from the gathered sums (same order of additions), and derives
     its span [base_g, base_{g+1}) of the global CDF and the global offspring range [F_g, F_{g+1}) its
     particles produce.  rStart comes from Philox(epoch), identical everywhere: no broadcast.
  4. a "resample" decision stays pending until the next deposit (or the next read).  Then each rank resamples its
     own ancestors and applies the new deposit to the offspring in the same pass.  Offspring whose index falls in
     the rank's own range are written in place; the others are staged as (x, y, z, theta) records -- one
     contiguous message per neighbour -- shipped point-to-point (xGMI) into the owner's landing area, unpacked
     there and deposited on, so every shard ends the step with exactly T_{g+1}-T_g particles again.
     Particle noise is keyed by GLOBAL index, so the result does not depend on the number of shards.

A non-resampling step moves no particle data at all.  The pose query all-reduces 6 doubles.

`backend` is the local half: `GpuShard` (libparticlefilter.so) in production; the tests plug an
oracle-backed CPU stand-in with the same methods to exercise this host logic under gloo.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence, Tuple

import numpy as np

MAX_SHARDS = 16


def shard_bounds(n_total: int, world: int) -> List[int]:
    """T_0..T_G: equal split, remainder spread by floor(g*N/G)."""
    return [(g * n_total) // world for g in range(world + 1)]


def overlap(a0: int, a1: int, b0: int, b1: int) -> Tuple[int, int]:
    lo, hi = max(a0, b0), min(a1, b1)
    return (lo, hi) if hi > lo else (lo, lo)


def exchange_plan(F: Sequence[int], T: Sequence[int], rank: int):
    """Who sends what after a resampling decision.

    F[g]..F[g+1] = global offspring indices produced by rank g; T[g]..T[g+1] = indices rank g holds.
    Returns (sends, recvs, below): sends = [(peer, stage_offset, count)], recvs = [(peer, local_offset, count)],
    below = number of staged offspring that precede this rank's own range (the stage holds those first, then
    the ones above the range -- the layout k_loc_resample writes)."""
    G = len(T) - 1
    ob, oe = F[rank], F[rank + 1]
    sb, se = T[rank], T[rank + 1]
    below = max(0, min(sb, oe) - ob)
    sends, recvs = [], []
    for h in range(G):
        if h == rank:
            continue
        lo, hi = overlap(ob, oe, T[h], T[h + 1])
        if hi > lo:
            off = (lo - ob) if h < rank else (lo - max(se, ob)) + below
            sends.append((h, off, hi - lo))
        lo, hi = overlap(F[h], F[h + 1], sb, se)
        if hi > lo:
            recvs.append((h, lo - sb, hi - lo))
    return sends, recvs, below


# ------------------------------------------------------------------ the GPU backend
class _DevMem:
    """zero-copy view of library-owned device memory for torch (CUDA array interface)"""

    def __init__(self, ptr: int, n: int, typestr: str):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (int(ptr), False), "version": 2}


class GpuShard:
    """The local half of a sharded filter on one GPU (pfx_loc_shard_*, include/pfx.h)."""

    def __init__(self, n_local: int, n_global: int, self_begin: int, rank: int, world: int, seed: int,
                 device: int, stage_capacity: int):
        import torch
        from . import _lib
        self._torch = torch
        self._lib = _lib
        self._L = _lib.lib()
        self.n, self.device = int(n_local), int(device)
        self._h = self._L.pfx_loc_create(self.n, seed & 0xFFFFFFFF, 0, self.device)
        if not self._h:
            raise _lib.PfxError(f"pfx_loc_create: {_lib.last_error()}")
        self.stage_capacity = int(stage_capacity)
        _lib.check(self._L.pfx_loc_shard_config(self._h, rank, world, n_global, self_begin, self.stage_capacity),
                   "pfx_loc_shard_config")
        self.world = world
        # run the library's kernels on torch's current stream so that they are ordered with the collectives
        self._L.pfx_loc_set_stream(self._h, C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream))
        dev = f"cuda:{self.device}"
        # the exchange table: one row of chunk sums per shard (the shard's own row is what its prefix kernel leaves)
        self.row = int(self._L.pfx_loc_shard_row_doubles(self._h))
        self._local = torch.as_tensor(_DevMem(self._L.pfx_loc_shard_local_ptr(self._h), self.row, "<f8"), device=dev)
        self._all = torch.as_tensor(_DevMem(self._L.pfx_loc_shard_all_ptr(self._h), MAX_SHARDS * self.row, "<f8"), device=dev)
        st, ld, b0, b1 = C.c_void_p(), C.c_void_p(), (C.c_void_p * 5)(), (C.c_void_p * 5)()
        self._lib.check(self._L.pfx_loc_shard_ptrs_both(self._h, C.byref(st), C.byref(ld), b0, b1), "pfx_loc_shard_ptrs_both")
        self._stage = torch.as_tensor(_DevMem(st.value, 4 * self.stage_capacity, "<f4"), device=dev)
        self._landing = torch.as_tensor(_DevMem(ld.value, 4 * self.stage_capacity, "<f4"), device=dev)
        self._bufs = [[torch.as_tensor(_DevMem(p, self.n, "<f4"), device=dev) for p in b] for b in (b0, b1)]

    def close(self):
        if getattr(self, "_h", None):
            self._L.pfx_loc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def _curr(self):
        """w, x, y, z, theta views of the current buffer (the library tracks which one on the host: no sync)"""
        return self._bufs[self._L.pfx_loc_shard_cur(self._h)]

    # -- VIO bookkeeping and control are the library's
    def deposit_vio(self, t, x, y, z, dist):
        self._L.pfx_loc_deposit_vio(self._h, t, x, y, z, dist)

    def set_force_branch(self, force):
        self._L.pfx_loc_set_force_branch(self._h, int(force))

    # -- the local halves of a deposit (include/pfx.h, pfx_loc_shard_*)
    def begin(self, bx, by, bz, rng, std) -> int:
        return self._lib.check(self._L.pfx_loc_shard_begin(self._h, bx, by, bz, rng, std), "pfx_loc_shard_begin")

    def deposit(self):
        """the O(N) pass; acts on the previous decision on the device (resample-and-deposit, or stream)"""
        self._lib.check(self._L.pfx_loc_shard_deposit(self._h), "pfx_loc_shard_deposit")

    def stage(self):
        """after a fused deposit whose plan sends offspring to other shards: produce them into the staging area"""
        self._lib.check(self._L.pfx_loc_shard_stage(self._h), "pfx_loc_shard_stage")

    def local_sums(self):
        return self._local

    def all_sums(self):
        return self._all[: self.world * self.row]

    def decide(self):
        """the global decision, enqueued; nothing is waited for"""
        self._lib.check(self._L.pfx_loc_shard_decide(self._h), "pfx_loc_shard_decide")

    def plan(self) -> Tuple[int, List[int]]:
        """waits for the last decision only (it lands in pinned host memory)"""
        br = C.c_int()
        F = (C.c_int64 * (MAX_SHARDS + 1))()
        self._lib.check(self._L.pfx_loc_shard_plan(self._h, C.byref(br), F), "pfx_loc_shard_plan")
        return br.value, [int(F[k]) for k in range(self.world + 1)]

    def produce(self):
        self._lib.check(self._L.pfx_loc_shard_produce(self._h), "pfx_loc_shard_produce")

    def land(self, deposit: bool, lo: int, hi: int):
        self._lib.check(self._L.pfx_loc_shard_land(self._h, int(deposit), lo, hi), "pfx_loc_shard_land")

    def finish(self, pieces: bool, lo: int = 0, hi: int = 0):
        self._lib.check(self._L.pfx_loc_shard_finish(self._h, int(pieces), lo, hi), "pfx_loc_shard_finish")

    def stage_view(self, off, count):
        """records off .. off+count of the send half of the staging area, as one flat float32 tensor"""
        return self._stage[4 * off: 4 * (off + count)]

    def landing_view(self, off, count):
        return self._landing[4 * off: 4 * (off + count)]

    # -- the collectives inside the library (pfx_loc_shard_comm_*): one C call per deposit
    def comm_init(self, unique_id: bytes):
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        self._lib.check(self._L.pfx_loc_shard_comm_init(self._h, buf), "pfx_loc_shard_comm_init")

    def comm_active(self) -> bool:
        return bool(self._L.pfx_loc_shard_comm_active(self._h))

    def comm_deposit_range(self, bx, by, bz, rng, std) -> int:
        return self._lib.check(self._L.pfx_loc_shard_deposit_range(self._h, bx, by, bz, rng, std), "pfx_loc_shard_deposit_range")

    def comm_settle(self):
        self._lib.check(self._L.pfx_loc_shard_settle(self._h), "pfx_loc_shard_settle")

    def comm_get_tag_loc(self):
        t = C.c_double()
        x, y, z, th = C.c_float(), C.c_float(), C.c_float(), C.c_float()
        ok = self._lib.check(self._L.pfx_loc_shard_get_tag_loc(self._h, C.byref(t), C.byref(x), C.byref(y), C.byref(z), C.byref(th)),
                             "pfx_loc_shard_get_tag_loc")
        return (int(ok), t.value, np.float32(x.value), np.float32(y.value), np.float32(z.value), np.float32(th.value))

    def comm_exchanged(self) -> int:
        return int(self._L.pfx_loc_shard_exchanged(self._h))

    def pose_sums(self):
        out = (C.c_double * 6)()
        self._lib.check(self._L.pfx_loc_shard_pose_sums(self._h, out), "pfx_loc_shard_pose_sums")
        return np.array(list(out), dtype=np.float64)

    def pose_from_sums(self, sums):
        arr = (C.c_double * 6)(*[float(v) for v in sums])
        t = C.c_double()
        x, y, z, th = C.c_float(), C.c_float(), C.c_float(), C.c_float()
        ok = self._L.pfx_loc_pose_from_sums(self._h, arr, C.byref(t), C.byref(x), C.byref(y), C.byref(z), C.byref(th))
        return (int(ok), t.value, np.float32(x.value), np.float32(y.value), np.float32(z.value), np.float32(th.value))

    def download(self):
        out = [np.empty(self.n, dtype=np.float32) for _ in range(5)]
        self._lib.check(self._L.pfx_loc_download(self._h, *[a.ctypes.data_as(C.c_void_p) for a in out], self.n),
                        "pfx_loc_download")
        return tuple(out)

    # -- profiling passthrough (bench.py)
    def profile_begin(self, max_records):
        self._lib.check(self._L.pfx_loc_profile_begin(self._h, int(max_records)), "pfx_loc_profile_begin")

    def profile_stride(self, stride):
        self._lib.check(self._L.pfx_loc_profile_stride(self._h, int(stride)), "pfx_loc_profile_stride")

    def profile_only(self, *kinds):
        from .particlefilter import DeviceParticleFilterLoc as D
        mask = 0xFFFFFFFF if not kinds else sum(1 << D.KINDS.index(k) for k in kinds)
        self._lib.check(self._L.pfx_loc_profile_mask(self._h, mask), "pfx_loc_profile_mask")

    def profile_records(self, cap=1 << 16):
        kind = np.empty(cap, dtype=np.int32)
        ms = np.empty(cap, dtype=np.float32)
        k = self._lib.check(self._L.pfx_loc_profile_records(self._h, kind.ctypes.data_as(C.c_void_p),
                                                            ms.ctypes.data_as(C.c_void_p), cap), "pfx_loc_profile_records")
        return kind[:k].copy(), ms[:k].copy()

    def branch_log(self, cap=65536):
        out = np.empty(cap, dtype=np.uint8)
        total = C.c_int64()
        k = self._lib.check(self._L.pfx_loc_branch_log(self._h, out.ctypes.data_as(C.c_void_p), cap, C.byref(total)),
                            "pfx_loc_branch_log")
        return out[:k].copy(), int(total.value)


# ------------------------------------------------------------------ the sharded filter
class ShardedLocFilter:
    """particleFilterLoc of n_total particles over all ranks of a torch.distributed process group."""

    def __init__(self, n_total: int, seed: int = 0, device: int = 0, group=None, backend_factory=None,
                 stage_slack: Optional[float] = None):
        import torch
        import torch.distributed as dist
        self._torch, self._dist = torch, dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        if self.world > MAX_SHARDS:
            raise ValueError(f"at most {MAX_SHARDS} shards")
        self.n_total = int(n_total)
        self.T = shard_bounds(self.n_total, self.world)
        n_local = self.T[self.rank + 1] - self.T[self.rank]
        if n_local < 1:
            raise ValueError("fewer particles than ranks")
        # Staging area for the offspring a shard produces for others, in units of its own size.  Default: the worst case, one
        # shard producing every offspring of a resampling (all the weight inside its index range) -- world x its own size,
        # 32 B per record for the send and the landing half together.  Not a corner case: right after the spawn the cloud
        # collapses onto a fraction of the range shell, and in bench.py's scene one shard of four shipped 1.08 x its own
        # size in one of the first resamplings (found by the four-rank rehearsal over the loopback transport; with the
        # former default of 1.0 that run stopped with "shard imbalance beyond the configured slack").
        if stage_slack is None:
            stage_slack = float(self.world)
        cap = min(int(n_local * stage_slack) + 1024, (1 << 31) - 1024)        # (record indices are 32-bit on the device)
        if backend_factory is None:
            self.shard = GpuShard(n_local, self.n_total, self.T[self.rank], self.rank, self.world, seed, device, cap)
        else:
            self.shard = backend_factory(n_local, self.n_total, self.T[self.rank], self.rank, self.world, seed, cap)
        self.stage_capacity = cap
        self._pg_is_nccl = dist.get_backend(group) == "nccl"
        self._comm = False
        # PFX_SHARD_RCCL: "0" never, "force" also under a process group that is not nccl (the tests: gloo for the hand-shake,
        # PFX_RCCL_LIB = the shared-memory loopback for the exchange, several ranks on one GPU), default: when the group is nccl
        mode = os.environ.get("PFX_SHARD_RCCL", "1")
        if backend_factory is None and mode != "0" and (self._pg_is_nccl or mode == "force"):
            self._comm = self._init_library_comm()
        self.initialized = False
        self._outstanding = False               # a decision has been launched and not acted on yet
        self._decided = False                   # ... at least one has been launched
        self._plan_cache = None
        self.exchanged_particles = 0

    def _init_library_comm(self) -> bool:
        """One RCCL communicator inside the library for this filter (pfx_loc_shard_comm_init): a deposit is then one C call per
        rank.  Collective; every rank ends up with the same answer -- if any rank fails, all stay on the torch path."""
        torch, dist = self._torch, self._dist
        from . import _lib
        ok = 1
        ident = [None]
        if self.rank == 0:
            buf = (C.c_uint8 * 128)()
            if _lib.lib().pfx_rccl_unique_id(buf) != 0:
                ok = 0
            ident[0] = bytes(buf)
        dist.broadcast_object_list(ident, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0, group=self.group)
        fdev = f"cuda:{self.shard.device}" if self._pg_is_nccl else "cpu"
        flag = torch.tensor([ok], device=fdev, dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
        if int(flag.item()) == 0:
            return False
        try:
            self.shard.comm_init(ident[0])
            ok = 1
        except Exception as e:      # noqa: BLE001 -- whatever went wrong, the torch path still works
            import sys
            print(f"slam3d_amd.sharded: in-library RCCL not used on rank {self.rank}: {e}", file=sys.stderr)
            ok = 0
        flag = torch.tensor([ok], device=fdev, dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
        return int(flag.item()) == 1

    @property
    def uses_library_comm(self) -> bool:
        return self._comm

    # -- collectives (device tensors go straight to RCCL; under gloo they are staged through the host)
    def _all_gather(self, out_t, in_t):
        dist, torch = self._dist, self._torch
        if self._pg_is_nccl or not in_t.is_cuda:
            dist.all_gather_into_tensor(out_t, in_t, group=self.group)
        else:
            h_in = in_t.cpu()
            h_out = torch.empty(out_t.shape, dtype=out_t.dtype)
            dist.all_gather_into_tensor(h_out, h_in, group=self.group)
            out_t.copy_(h_out)

    def _p2p(self, sends, recvs):
        """sends/recvs: lists of (peer, tensor)"""
        dist, torch = self._dist, self._torch
        if not sends and not recvs:
            return
        staged = not self._pg_is_nccl and any(t.is_cuda for _, t in sends + recvs)
        ops, back = [], []
        for peer, t in recvs:
            if staged:
                h = torch.empty(t.shape, dtype=t.dtype)
                back.append((t, h))
                ops.append(dist.P2POp(dist.irecv, h, peer, group=self.group))
            else:
                ops.append(dist.P2POp(dist.irecv, t, peer, group=self.group))
        for peer, t in sends:
            ops.append(dist.P2POp(dist.isend, t.cpu() if staged else t, peer, group=self.group))
        for req in dist.batch_isend_irecv(ops):
            req.wait()
        for t, h in back:
            t.copy_(h)

    def _plan(self):
        """(branch, F) of the outstanding decision; read once, when first needed"""
        if self._comm:
            return self.shard.plan()            # (the library keeps the state; reading the plan twice is harmless)
        if self._plan_cache is None:
            self._plan_cache = self.shard.plan()
        return self._plan_cache

    @property
    def last_branch(self):
        return self._plan()[0] if self._decided else None

    @property
    def last_F(self):
        return self._plan()[1] if self._decided else None

    def _ship_and_land(self, F, deposit: bool):
        """after the offspring of a "resample" decision have been produced: the foreign ones go to their owners, what
        arrives is unpacked (deposit: and deposited on)"""
        sh, T, r = self.shard, self.T, self.rank
        sends, recvs, _below = exchange_plan(F, T, r)
        lo = min(max(F[r], T[r]), T[r + 1]) - T[r]              # the local elements [lo, hi) were produced here
        hi = max(max(min(F[r + 1], T[r + 1]), T[r]) - T[r], lo)
        if deposit and sends:
            sh.stage()                          # (without a deposit, produce() has staged them already)
        # record k of the landing area: local element k (k < lo) or hi + (k - lo)
        self._p2p([(p, sh.stage_view(off, cnt)) for p, off, cnt in sends],
                  [(p, sh.landing_view(off if off < lo else lo + (off - hi), cnt)) for p, off, cnt in recvs])
        sh.land(deposit, lo, hi)
        self.exchanged_particles += sum(c for _, _, c in sends)
        return lo, hi

    def _settle(self):
        """before anything reads the particles: produce a resampling that is still outstanding, without a deposit"""
        if self._comm:
            self.shard.comm_settle()
            return
        if not self._outstanding:
            return
        self._outstanding = False
        branch, F = self._plan()
        if branch:
            self.shard.produce()
            self._ship_and_land(F, False)

    # -- the wrapper interface
    def depositVio(self, t, x, y, z, dist_):
        self.shard.deposit_vio(t, x, y, z, dist_)

    def depositRange(self, bx, by, bz, range, stdRange):        # noqa: A002
        sh = self.shard
        if self._comm:                          # the whole deposit, collectives included, in one library call
            if sh.comm_deposit_range(bx, by, bz, range, stdRange) == 1:
                self.initialized = True         # first range: every shard spawned its own particles
                self._decided = False
            else:
                self._decided = True
            self.exchanged_particles = sh.comm_exchanged()
            return
        if sh.begin(bx, by, bz, range, stdRange) == 1:
            self.initialized = True             # first range: every shard spawned its own particles
            self._outstanding = self._decided = False
            return
        # The pass over the particles goes first: the kernel itself acts on the previous decision (resampling fused
        # with this deposit, or plain streaming).  Only then does the host look at that decision -- by now it has
        # long landed -- to find out whether offspring have to be shipped.
        sh.deposit()
        branch, F = self._plan() if self._outstanding else (0, None)
        self._outstanding = False
        if branch:
            lo, hi = self._ship_and_land(F, True)
            sh.finish(True, lo, hi)
        else:
            sh.finish(False)
        self._all_gather(sh.all_sums(), sh.local_sums())
        sh.decide()
        self._outstanding = self._decided = True
        self._plan_cache = None

    def depositRssi(self, bx, by, bz, rssi):
        self.depositRange(bx, by, bz, 1.5, 0.5)        # particleFilter.c:213-226

    def getTagLoc(self):
        torch, dist = self._torch, self._dist
        if not self.initialized:
            return (0, 0.0, np.float32(0), np.float32(0), np.float32(0), np.float32(0))
        if self._comm:
            return self.shard.comm_get_tag_loc()
        self._settle()
        sums = torch.from_numpy(self.shard.pose_sums())
        if self._pg_is_nccl:
            d = sums.cuda()
            dist.all_reduce(d, group=self.group)
            sums = d.cpu()
        else:
            dist.all_reduce(sums, group=self.group)
        return self.shard.pose_from_sums(sums.numpy())

    # -- passthroughs used by bench.py
    def set_force_branch(self, force):
        self.shard.set_force_branch(force)

    def profile_begin(self, max_records):
        self.shard.profile_begin(max_records)

    def profile_only(self, *kinds):
        self.shard.profile_only(*kinds)

    def profile_stride(self, stride):
        self.shard.profile_stride(stride)

    def profile_records(self):
        return self.shard.profile_records()

    def branch_log(self):
        return self.shard.branch_log()

    def download_local(self):
        self._settle()
        return self.shard.download()
