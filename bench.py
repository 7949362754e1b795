#!/usr/bin/env python3
"""bench.py -- BASELINE.json headline metric: particle-updates/sec (range update + resample) at N = 16 M.

Workload (BASELINE.json configs[2], SURVEY.md 8d "config 3", the roofline case): one particleFilterLoc of
N = 16 000 000 particles per GPU, 8 beacons at (+-5, +-5, {0.5, 3.0}) m, truth on a 2 m circle moving
5 cm per step, ranges = true distance + N(0, 0.1^2).  One STEP = depositVio + depositRange(beacon k%8) +
depositRssi(beacon (k+1)%8): two deposits, i.e. 2N particle-updates (one particle through one
depositRange/Rssi call: VIO commit, gate, sums, decision, resample-or-normalise).  The first deposit of a
step carries the VIO motion, the second has zero motion.  Particles live in HBM the whole time; the only
host->device traffic in the timed region is kernel arguments.

The scene is SURVEY's, and deliberately self-contradictory: depositRssi asserts a 1.5 m +- 1.5 m range
(particleFilter.c:213-226 ignores the rssi value) to a beacon that is at least 5 m from the truth, so every
RSSI deposit gates the whole converged cloud down and triggers the 5 % respawn (pfResample.c:53-56,75-76);
the filter does not track (the estimate ends 3-8 m from the truth; `final_position_error_m` in the record) and half
the deposits resample.  It is a traffic case, not a tracking demo; `extra.tracking` is the same step on ranges only.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--particles PER_GPU] [--force-branch -1|0|1]

Before the W warm-up steps the filter always runs PRE_ROLL = 400 untimed steps (0.1 s at N = 16 M): for the first
~300 steps after the spawn the cloud is still collapsing from the initial range shell (larger respawn share, clustered
weights, more source tiles per destination tile; the resampling launch takes 0.17 ms there against 0.15 ms once the
filter tracks) -- the state of the workload, not of the clock -- and a 20-step run would otherwise time that
transient while a 200-step run averages over its end: 115 vs 126 G particle-updates/s with a 40-step pre-roll,
133 vs 131 with 400.  The transient is still in the record: `extra.transient` is the same measurement behind a
40-step pre-roll.

N > 1 (launched by torch.distributed.run, one rank per GPU): ONE filter of N x 16 M particles sharded by
particle range; per deposit an all-gather of the per-GPU weight/pose sums (RCCL) gives every rank the
global decision and its CDF base; see slam3d_amd/sharded.py.  Weak scaling: per-GPU work fixed.

Prints ONE JSON line on rank 0.  At N = 1 the line also carries `extra` (N sweep 1 M / 2^24 / 64 M, the
always-resample run, BASELINE config 5 = batched SLAM) and the CPU baselines; --no-extras / --no-cpu-baseline
leave them out.
"""
import argparse
import contextlib
import ctypes as C
import gc
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

# SURVEY.md 8d / BASELINE.md: algorithmic (compulsory) HBM bytes per particle, SoA f32: a deposit's streaming pass 44
# (with motion) / 28 (without), a resampling 40 (source weights + ancestors read, offspring written).  The fused kernel
# does a resampling AND the following deposit in one launch; the offspring never leave the registers in between, so
# what such a launch MUST move is the resampling's 40 B and the roofline prices it at that.  (Priced as the two reference
# passes it replaces, ROWS_REPLACED, it would come out above the HBM peak: 84 B x 16 M / 0.155 ms = 8.7 TB/s.)  What it
# really moves is in "traffic" (rocprofv3 PMC, profiles/pmc_traffic.json).
ALG_BYTES = {"stream": 44, "stream_static": 28, "resample": 40, "resample_stream": 40, "resample_static": 40}
ROWS_REPLACED = {"resample_stream": 84, "resample_static": 68}
KERNEL_OF = {"stream": "k_loc_stream<motion,gate>", "stream_static": "k_loc_stream<gate>", "resample": "k_loc_resample_dest",
             "resample_stream": "k_loc_deposit<motion> (resampling side)", "resample_static": "k_loc_deposit<static> (resampling side)"}
# (a sharded filter runs the same two sides in k_loc_deposit_sharded)
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: 8 TB/s spec
PRE_ROLL = int(os.environ.get("BENCH_PRE_ROLL", "400"))     # untimed steps before the warm-up (see the module docstring)


@contextlib.contextmanager
def gc_paused():
    """Timed regions run with CPython's cyclic garbage collector paused (as `timeit` does).  With torch imported a
    full (generation-2) collection takes 40-60 ms; the wrappers allocate a few ctypes / numpy objects per call, so one
    such collection can land inside a 15-50 ms timed loop and double its time (measured: config 5, 120 steps, 0.231 ms
    per step with the collector paused, 0.50-0.76 ms when one full collection fell into the loop; the same loop from
    C++, tools/dev/slam_loop.cpp, 0.226 ms).  That is the interpreter's housekeeping, not the path under measurement.
    The pause begins BEFORE the warm-up steps: the collection it starts with leaves the GPU idle for those 40-60 ms, and a
    timed loop that starts right behind it runs its first milliseconds at the clocks of an idle part (a 20-step run came
    out 11 % slower that way, resampling side 0.181 ms against 0.158)."""
    gc.collect()
    was = gc.isenabled()
    gc.disable()
    try:
        yield
    finally:
        if was:
            gc.enable()


def scene(n_steps):
    import _synth
    return _synth.synthetic_scene(n_steps, seed=42)


POSE_EVERY_STEP = False     # --pose-every-step: the replay drivers' pattern (a getTagLoc per VIO row); not the headline metric


def step(pf, k, bx, truth, rng, ranges_only=False):
    """one STEP of the workload on any object with the wrapper interface.  ranges_only (extra.tracking): the second
    deposit is a depositRange with the measured range to beacon (k+1)%8 instead of SURVEY's depositRssi, whose fixed
    1.5 m +- 1.5 m contradicts every beacon of this scene -- the same two passes per step on a filter that tracks."""
    pf.depositVio(1000.0 + 0.1 * k, float(truth[k, 0]), float(truth[k, 1]), float(truth[k, 2]), 0.0)
    if POSE_EVERY_STEP:
        pf.getTagLoc()
    b = k % 8
    pf.depositRange(float(bx[b, 0]), float(bx[b, 1]), float(bx[b, 2]), float(rng[k, 0]), 0.1)
    b = (k + 1) % 8
    if ranges_only:
        pf.depositRange(float(bx[b, 0]), float(bx[b, 1]), float(bx[b, 2]), float(rng[k, 1]), 0.1)
    else:
        pf.depositRssi(float(bx[b, 0]), float(bx[b, 1]), float(bx[b, 2]), -60)


# ------------------------------------------------------------------ CPU baselines (rank 0, N = 1 only)
class _RefLoc:
    """The COMPILED REFERENCE (oracle/_ref/libpf_ref_glibc_n*.so: the reference's own sources built with
    PF_N_TAG_LOC = n) behind the wrapper interface."""

    def __init__(self, path, n):
        self.L = C.CDLL(path)
        f, d = C.c_float, C.c_double
        self.L.particleFilterLoc_depositVio.argtypes = [C.c_void_p, d, f, f, f, f]
        self.L.particleFilterLoc_depositRange.argtypes = [C.c_void_p, f, f, f, f, f]
        self.L.particleFilterLoc_depositRssi.argtypes = [C.c_void_p, f, f, f, C.c_int]
        self.L.particleFilterSeed_set.argtypes = [C.c_uint]
        self.buf = C.create_string_buffer(2 * n * 20 + 64)
        self.p = C.cast(self.buf, C.c_void_p)
        self.L.particleFilterSeed_set(123456789)
        self.L.particleFilterLoc_init(self.p)

    def depositVio(self, *a):
        self.L.particleFilterLoc_depositVio(self.p, *a)

    def depositRange(self, *a):
        self.L.particleFilterLoc_depositRange(self.p, *a)

    def depositRssi(self, *a):
        self.L.particleFilterLoc_depositRssi(self.p, *a)


def _timed_steps(pf, budget_s, max_steps, bx, truth, rng):
    pf.depositVio(999.9, float(truth[0, 0]), float(truth[0, 1]), float(truth[0, 2]), 0.0)
    pf.depositRange(float(bx[0, 0]), float(bx[0, 1]), float(bx[0, 2]), float(rng[0, 0]), 0.1)   # init, untimed
    t0 = time.perf_counter()
    k = 0
    while k < max_steps:
        step(pf, k, bx, truth, rng)
        k += 1
        if time.perf_counter() - t0 > budget_s:
            break
    return k, time.perf_counter() - t0


def _in_big_stack_thread(fn, stack_mb):
    """the reference keeps float weightCdf[N] on the stack (pfResample.c:35): 64 MB at N = 16 M"""
    out = {}
    old = threading.stack_size(stack_mb << 20)
    th = threading.Thread(target=lambda: out.update(r=fn()))
    th.start()
    th.join()
    threading.stack_size(old)
    return out.get("r")


def host_threads():
    """(threads for the all-cores baseline, what the host offers).  Every CPU this process may run on: os.cpu_count(),
    cut down to the scheduler affinity mask and to the cgroup CPU quota where the box sets one (a GPU box hands a
    one-GPU job a share of its CPUs; threads beyond the quota would only take turns).  BENCH_CPU_THREADS overrides."""
    avail = os.cpu_count() or 1
    info = {"cores_available": avail}
    n = avail
    try:
        info["affinity"] = len(os.sched_getaffinity(0))
        n = min(n, info["affinity"])
    except (AttributeError, OSError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            q, per = fh.read().split()[:2]
        if q != "max":
            info["cgroup_cpu_quota"] = float(q) / float(per)
            n = min(n, max(1, int(info["cgroup_cpu_quota"] + 0.5)))
    except (OSError, ValueError):
        pass
    if os.environ.get("BENCH_CPU_THREADS"):
        n = int(os.environ["BENCH_CPU_THREADS"])
        info["threads_from"] = "BENCH_CPU_THREADS"
    return max(1, min(n, 1000)), info


def kernel_sources_sha():
    """what profiles/pmc_traffic.json was measured on (tools/pmc_traffic.py records it): the kernel sources"""
    import hashlib
    h = hashlib.sha256()
    for f in ("pf_common.h", "pf_loc_kernels.h"):
        with open(os.path.join(ROOT, "slam3d_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def cpu_baseline(budget_s=12.0):
    """The same workload on the host: the primary record is ONE core at N = 1 M (a bounded sample of the GPU workload);
    `variants` adds one core at the full N = 16 M and all host cores.  Prefers the compiled reference (kind
    "reference"); else the oracle restatement ("port")."""
    bx, truth, rng = scene(64)
    cpu_model = ""
    try:
        with open("/proc/cpuinfo") as fh:
            cpu_model = next((l.split(":", 1)[1].strip() for l in fh if l.startswith("model name")), "")
    except OSError:
        pass

    def one_core(n, budget, max_steps):
        ref_so = os.path.join(ROOT, "oracle", "_ref", f"libpf_ref_glibc_n{n // 1000000}m.so")
        if os.path.exists(ref_so):
            make, kind = (lambda: _RefLoc(ref_so, n)), "reference"
        else:
            from oracle import oracle as ora
            make, kind = (lambda: ora.OracleLoc(n, rng_mode=ora.RNG_GLIBC, seed=123456789, acc=ora.ACC_F32)), "port"
        k, dt = _in_big_stack_thread(lambda: _timed_steps(make(), budget, max_steps, bx, truth, rng), 16 + n // 1000000 * 6)
        return {"value": 2.0 * n * k / dt, "unit": "particle-updates/s", "cores": 1, "kind": kind,
                "sample": f"same scene and step, N={n:,} particles, {k} steps, {dt:.1f} s, serial rand_r stream, float32 as the "
                          "reference"}

    primary = one_core(1000000, budget_s, 60)
    primary["sample"] = primary["sample"].replace("particles,", "particles (1/16 of the GPU workload),")
    primary["cpu_model"] = cpu_model
    variants = [one_core(16000000, 10.0, 3)]

    # all host cores: T independent oracle filters of N/T particles, one per thread (ctypes releases the GIL), each
    # with its own Philox stream -- what a threaded port of the reference could reach at best: the serial rand_r
    # stream cannot be split, and a real port would still have to join the threads for the global resampling
    from oracle import oracle as ora
    cores, cores_info = host_threads()
    n_each = 16000000 // cores
    filters = [ora.OracleLoc(n_each, rng_mode=ora.RNG_PHILOX, seed=1000 + i, acc=ora.ACC_F32) for i in range(cores)]
    res = [None] * cores

    def work(i):
        res[i] = _timed_steps(filters[i], 8.0, 4, bx, truth, rng)

    t0 = time.perf_counter()
    ths = [threading.Thread(target=work, args=(i,)) for i in range(cores)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    wall = time.perf_counter() - t0
    upd = sum(2.0 * n_each * r[0] for r in res)
    variants.append({"value": upd / max(r[1] for r in res), "unit": "particle-updates/s", "cores": cores, "kind": "port",
                     "sample": f"{cores} threads x one oracle filter of N={n_each:,} each (16 M in all), {res[0][0]} steps, {wall:.1f} s wall "
                               "incl. spawn; per-thread Philox streams, float32 sums; upper bound of a threaded port (no join for "
                               "the global resampling)", "cpu_model": cpu_model, **cores_info})
    primary["variants"] = variants
    return primary


# ------------------------------------------------------------------ config 5: batched SLAM filters, no comms
class _RefSlam:
    """particleFilterSlam of the COMPILED REFERENCE (oracle/_ref/libpf_ref_glibc.so): one tag, B beacons, one core."""

    def __init__(self, path, B):
        self.L = C.CDLL(path)
        f, d, vp = C.c_float, C.c_double, C.c_void_p
        self.L.particleFilterSlam_depositTagVio.argtypes = [vp, d, f, f, f, f]
        self.L.particleFilterSlam_depositRange.argtypes = [vp, vp, f, f, vp, C.c_int]
        self.L.particleFilterSeed_set.argtypes = [C.c_uint]
        self.pf = C.create_string_buffer(4056)
        self.b = [C.create_string_buffer(2020056) for _ in range(B)]
        self.L.particleFilterSeed_set(123456789)
        self.L.particleFilterSlam_init(C.cast(self.pf, vp))
        for b in self.b:
            self.L.particleFilterSlam_addBcn(C.cast(b, vp))
        self.all = (vp * B)(*[C.cast(b, vp) for b in self.b])
        self.B = B

    def vio(self, t, x, y, z):
        self.L.particleFilterSlam_depositTagVio(C.cast(self.pf, C.c_void_p), t, x, y, z, 0.0)

    def rng(self, b, r, s):
        self.L.particleFilterSlam_depositRange(C.cast(self.pf, C.c_void_p), C.cast(self.b[b], C.c_void_p), r, s, self.all, self.B)


def _slam_inputs(B, n_steps):
    bx, truth, rng = scene(n_steps)
    bpos = bx[:B] if B <= 8 else np.concatenate([bx, bx[:B - 8] * np.float32(0.6)])

    def dist(k):
        b = k % B
        return np.float32(np.linalg.norm(truth[k].astype(np.float64) - bpos[b].astype(np.float64)) + (rng[k, 0] - np.linalg.norm(
            truth[k].astype(np.float64) - bx[k % 8].astype(np.float64))))
    return truth, dist


def slam_cpu_baseline(B, budget_s=10.0):
    ref_so = os.path.join(ROOT, "oracle", "_ref", "libpf_ref_glibc.so")
    truth, dist = _slam_inputs(B, 400)
    if os.path.exists(ref_so):
        pf, kind = _RefSlam(ref_so, B), "reference"
        vio, rngf = pf.vio, pf.rng
    else:
        from oracle import oracle as ora
        o = ora.OracleSlam(100, 1000, rng_mode=ora.RNG_GLIBC, seed=123456789, acc=ora.ACC_F32)
        bs = [o.addBcn() for _ in range(B)]
        kind = "port"
        vio = lambda t, x, y, z: o.depositTagVio(t, x, y, z, 0.0)       # noqa: E731
        rngf = lambda b, r, s: o.depositRange(bs[b], r, s)              # noqa: E731
    for k in range(B):                      # first range of every beacon (spawn), untimed
        vio(1000.0 + 0.1 * k, float(truth[k, 0]), float(truth[k, 1]), float(truth[k, 2]))
        rngf(k % B, float(dist(k)), 0.1)
    t0 = time.perf_counter()
    k = B
    while k < 390 and time.perf_counter() - t0 < budget_s:
        vio(1000.0 + 0.1 * k, float(truth[k, 0]), float(truth[k, 1]), float(truth[k, 2]))
        rngf(k % B, float(dist(k)), 0.1)
        k += 1
    dt = time.perf_counter() - t0
    return {"value": 100.0 * 1000.0 * (k - B) / dt, "unit": "particle-updates/s", "cores": 1, "kind": kind,
            "sample": f"ONE tag (100 x {B} x 1000 particles) of the batch, {k - B} deposits, {dt:.1f} s, serial rand_r stream; a deposit "
                      f"also moves the other {B - 1} beacons' particles (particleFilter.c:198-199), not counted as updates"}


def measure_slam(F, B, K, W, world, rank, local, dist, torch, with_cpu=False, census_steps=16):
    """BASELINE.json configs[4]: independent particleFilterSlam tags (100 tag particles x B beacons x 1000), sharded
    over the GPUs with no communication.  One STEP = depositTagVio + depositRange(beacon k % B) for every tag of the
    rank's batch: F x 100 x 1000 beacon-particle updates (+ F x 100 tag particles).  Returns the record (rank 0) or None.

    Roofline: a deposit is priced by what its rows did (the census of `census_steps` further deposits after the timed
    loop, same steady state): a row that only normalises reads w, x, y, z and rewrites w = 20 B per beacon particle, a
    row that resamples reads and writes all five arrays = 40 B (pfResample.c:141-201), and a filter whose tag particles
    resampled force-resamples every initialised beacon (:122-130) = B x 40 B per tag-particle row."""
    from slam3d_amd import slam
    batch = slam.DeviceSlamBatch(F, 100, 1000, max_bcns=B, seed0=1000003 * (rank + 1), device=local)
    for _ in range(B):
        batch.add_bcn_all()
    truth, dfun = _slam_inputs(B, K + W + B + census_steps + 1)
    ones = np.ones(F, dtype=np.float32)

    # per-step arguments in arrays that live as long as the batch (a caller that feeds a batch keeps such arrays anyway)
    a_t, a_x, a_y, a_z = np.zeros(F), ones.copy(), ones.copy(), ones.copy()
    a_slot, a_r, a_sd = np.zeros(F, dtype=np.int32), ones.copy(), ones * np.float32(0.1)

    def one(k):
        a_t[:] = 1000.0 + 0.1 * k
        a_x[:] = truth[k, 0]; a_y[:] = truth[k, 1]; a_z[:] = truth[k, 2]
        batch.deposit_tag_vio_batch(a_t, a_x, a_y, a_z)
        a_slot[:] = k % B
        a_r[:] = dfun(k)
        batch.deposit_range_batch(a_slot, a_r, a_sd)

    def barrier():
        if world > 1:
            dist.barrier()
        batch.sync()
        torch.cuda.synchronize()

    with gc_paused():
        for k in range(B):          # first range of every beacon (spawn), untimed
            one(k)
        for k in range(B, B + W):
            one(k)
        barrier()
        t0 = time.perf_counter()
        for k in range(B + W, B + W + K):
            one(k)
        barrier()
        dt = time.perf_counter() - t0
    rows_norm = rows_rs = forced = 0
    for k in range(B + W + K, B + W + K + census_steps):
        one(k)
        c = batch.row_census()
        rows_norm += c[0]; rows_rs += c[1]; forced += c[2]
    if world > 1:
        t = torch.tensor([dt], device="cuda" if os.environ.get("BENCH_PG", "nccl") == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank != 0:
        return None
    upd = 100.0 * 1000.0 * F * world * K
    per_dep = 1000.0 * (20.0 * rows_norm + 40.0 * rows_rs + 40.0 * forced * 100 * B) / census_steps      # bytes, one GPU
    gbs = per_dep / (dt / K) / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic_slam.json")
    if os.path.exists(tpath) and F == 512:
        with open(tpath) as fh:
            traffic = json.load(fh).get("k_slam_gate_rows", {}).get("hbm_bytes_per_launch")
    return {
        "metric": "particle-updates/sec, batched particleFilterSlam (config 5)", "value": upd / dt, "unit": "particle-updates/s",
        "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"config5: {F} SLAM tags per GPU x 100 tag particles x {B} beacons x 1000, per step depositTagVio + "
                               "depositRange on one beacon of every tag", "filters_per_gpu": F, "beacons": B,
                   "filter_updates_per_s": F * world * K / dt, "parallelism": "tags sharded over GPUs, no communication",
                   "rows_resampled_fraction": rows_rs / max(1, rows_norm + rows_rs),
                   "tag_resamplings_per_deposit": forced / census_steps},
        "roofline": {"bound": "hbm", "kernel": "k_slam_gate_rows (one workgroup per beacon row)", "achieved": gbs, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": traffic,
                     "alg_bytes_per_deposit": per_dep,
                     "note": "whole-step rate; bytes priced per row by what it did (normalise 20 B, resample 40 B per beacon "
                             f"particle; census of {census_steps} deposits)"},
        "cpu_baseline": slam_cpu_baseline(B) if with_cpu else None}


# ------------------------------------------------------------------ config 3: one Loc filter
def measure_loc(pfm, torch, dist, world, rank, local, n, K, W, force, args, headline, pre_roll=None, ranges_only=False):
    """pre_roll (default PRE_ROLL) + W untimed steps, K timed; returns (record fields, per-kernel table) on rank 0,
    (None, None) elsewhere"""
    PRE_ROLL = globals()["PRE_ROLL"] if pre_roll is None else pre_roll
    pg = os.environ.get("BENCH_PG", "nccl")
    bx, truth, rng = scene(PRE_ROLL + K + W + 1)
    force_sharded = os.environ.get("BENCH_FORCE_SHARDED") == "1"      # rehearsal: the sharded host logic on one rank
    if world == 1 and not force_sharded:
        pf = pfm.DeviceParticleFilterLoc(n, seed=123456789, stream_id=0, device=local)
    else:
        from slam3d_amd.sharded import ShardedLocFilter
        pf = ShardedLocFilter(n * world, seed=123456789, device=local)
    exchange = None
    if hasattr(pf, "uses_library_comm"):
        exchange = ("in-library RCCL (pfx_loc_shard_deposit_range: one C call per deposit and rank)" if pf.uses_library_comm
                    else "torch.distributed collectives on views of the library's buffers")
    pf.set_force_branch(force)
    if args.undeferred and world == 1 and not force_sharded:
        pf.set_deferred(False)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # init (spawn) + pre-roll + warm-up, untimed
    pf.depositVio(999.9, float(truth[0, 0]), float(truth[0, 1]), float(truth[0, 2]), 0.0)
    pf.depositRange(float(bx[0, 0]), float(bx[0, 1]), float(bx[0, 2]), float(rng[0, 0]), 0.1)
    events = headline and not args.no_kernel_events
    with gc_paused():
        for k in range(PRE_ROLL + W):
            step(pf, k, bx, truth, rng, ranges_only)
        barrier()
        if events:
            # the O(N) kernels only: the two tiny prefix / decision launches of every deposit are not reported, and
            # timing them too costs ~12 us of GPU time per step
            pf.profile_only("stream", "stream_static", "resample", "deposit", "deposit_static")
            pf.profile_stride(args.event_stride)     # every event_stride-th launch of a class carries events (~3 us each)
            pf.profile_begin(4 * K * 4 + 16)
        barrier()
        sent0 = int(getattr(pf, "exchanged_particles", 0))
        t0 = time.perf_counter()
        for k in range(PRE_ROLL + W, PRE_ROLL + W + K):
            step(pf, k, bx, truth, rng, ranges_only)
        barrier()
        dt = time.perf_counter() - t0
        sent = int(getattr(pf, "exchanged_particles", 0)) - sent0       # offspring this rank shipped to other shards

    kinds, ms = pf.profile_records() if events else (np.zeros(0, dtype=np.int32), np.zeros(0, dtype=np.float32))
    full_log = pf.branch_log()[0]       # rank 0's log; a sharded filter takes the same decision on every rank
    branches = full_log[-2 * K:]
    pose = pf.getTagLoc()
    err = float(np.linalg.norm(np.array(pose[2:5], dtype=np.float64) - truth[PRE_ROLL + W + K - 1]))
    sent_max = sent_sum = sent
    if world > 1:
        t = torch.tensor([dt], device="cuda" if pg == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        c = torch.tensor([sent, sent], device="cuda" if pg == "nccl" else "cpu", dtype=torch.int64)
        dist.all_reduce(c[0:1], op=dist.ReduceOp.MAX)
        dist.all_reduce(c[1:2], op=dist.ReduceOp.SUM)
        sent_max, sent_sum = int(c[0].item()), int(c[1].item())
    del pf
    if rank != 0:
        return None, None
    rec = {"value": 2.0 * n * world * K / dt, "ms_per_step": dt / K * 1e3, "particles_per_gpu": n, "steps": K,
           "resample_fraction": float(branches.mean()) if branches.shape[0] else None, "final_position_error_m": err}
    if exchange:
        rec["exchange"] = exchange
        # what the neighbour exchange of the resampling deposits moved in the timed loop (16 B per offspring that changes
        # shard): in this scene the 5 % respawn lands on global indices [0, numSpawn) -- shard 0 -- with weight 1 against
        # ~0.1-0.5 elsewhere, every shard boundary of the offspring shifts, and ~3 % of ALL particles cross each boundary per
        # resampling (tools/dev/diag_migration.py): at 64 M that is ~30 MB per link and resampling
        rec["offspring_shipped_per_step"] = {"all_ranks": sent_sum / K, "busiest_rank": sent_max / K,
                                             "bytes_per_step_busiest_rank": 16.0 * sent_max / K}
    # every timed deposit by what its launch did: deposit j follows decision j - 1 (device-side branch log); even j carry
    # the VIO motion, odd j are static
    if full_log.shape[0] >= 2 * K + 1:
        prev = full_log[-(2 * K + 1):-1].astype(bool)
        rec["launch_classes"] = {"resample_stream": int(prev[0::2].sum()), "stream": int((~prev[0::2]).sum()),
                                 "resample_static": int(prev[1::2].sum()), "stream_static": int((~prev[1::2]).sum())}
    if not events:
        return rec, {}
    names = pfm.DeviceParticleFilterLoc.KINDS
    # Launch classes.  With deferred resampling a deposit is ONE launch of k_loc_deposit, which resamples-and-deposits or
    # only streams according to the previous decision (known on the device only): the launches of a class are told apart
    # by the device-side branch log (deposit j of the timed region follows decision j - 1) and, where that does not
    # line up, by duration.
    classes = {}
    for nm in ("stream", "stream_static", "resample"):
        sel = ms[kinds == names.index(nm)]
        if sel.shape[0]:
            sel = sel[sel > 0.25 * sel.max()]
        classes[nm] = sel
    for nm, fused, plain, phase in (("deposit", "resample_stream", "stream", 0), ("deposit_static", "resample_static", "stream_static", 1)):
        sel = ms[kinds == names.index(nm)]
        if not sel.shape[0]:
            continue
        prev = None
        if full_log is not None and full_log.shape[0] >= 2 * K + 1:
            # decision before each deposit of this phase; the timed launches are number 0, stride, 2 * stride ... of them
            cand = full_log[-(2 * K + 1):-1][phase::2][::args.event_stride].astype(bool)
            if cand.shape[0] == sel.shape[0]:
                prev = cand
        if prev is None:
            prev = sel > 0.5 * (sel.min() + sel.max())
        classes[fused] = sel[prev]
        classes[plain] = np.concatenate([classes.get(plain, sel[:0]), sel[~prev]])
    per = {}
    for nm, sel in classes.items():
        if sel.shape[0]:
            avg_ms = float(sel.mean())
            gbs = ALG_BYTES[nm] * n / (avg_ms * 1e-3) / 1e9
            per[nm] = {"launches": int(sel.shape[0]), "avg_ms": avg_ms, "alg_bytes_per_particle": ALG_BYTES[nm],
                       "achieved_GBs": gbs, "frac": gbs / HBM_PEAK_GBS, "total_ms": float(sel.sum())}
            if nm in ROWS_REPLACED:         # the same launch priced as the reference passes it stands for
                per[nm]["reference_passes_bytes_per_particle"] = ROWS_REPLACED[nm]
                per[nm]["reference_passes_GBs"] = ROWS_REPLACED[nm] * n / (avg_ms * 1e-3) / 1e9
    return rec, per


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--particles", dest="n", type=int, default=16000000, help="particles per GPU")
    ap.add_argument("--total-particles", type=int, default=0,
                    help="strong scaling (BASELINE config 4: 67108864): total particles of the filter, split over the GPUs; "
                         "overrides --particles")
    ap.add_argument("--force-branch", type=int, default=-1, help="-1 reference test, 0 never resample, 1 always")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="headline only: no N sweep, no always-resample run, no config 5")
    ap.add_argument("--event-stride", type=int, default=3,
                    help="time every n-th launch of each O(N) kernel class (3: coprime with the 8-beacon cycle of the scene)")
    ap.add_argument("--no-kernel-events", action="store_true", help="timed region without per-kernel HIP events (no roofline)")
    ap.add_argument("--pose-every-step", action="store_true", help="also call getTagLoc after every depositVio (driver pattern)")
    ap.add_argument("--undeferred", action="store_true", help="one resample kernel per deposit (pfx_loc_set_deferred(0))")
    ap.add_argument("--workload", default="config3", choices=["config3", "config5", "loc_batch"])
    ap.add_argument("--batch-filters", type=int, default=1600, help="loc_batch: filters in the batch (each of --particles)")
    ap.add_argument("--slam-filters", type=int, default=512, help="config5: SLAM tags per GPU")
    ap.add_argument("--slam-bcns", type=int, default=4, help="config5: beacons per tag")
    args = ap.parse_args()
    global POSE_EVERY_STEP
    POSE_EVERY_STEP = args.pose_every_step

    # ONE line on stdout: libraries print banners there (RCCL announces its version when a communicator is created), so
    # everything but the result goes to stderr and the result is written to the real stdout at the end
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    def emit(obj):
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(obj) + "\n").encode())

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    # BENCH_PG=gloo is a rehearsal mode for a one-GPU box: the ranks share the visible GPUs and the collectives are
    # staged through the host.  The driver's runs use the default (nccl = RCCL, one GPU per rank).
    pg = os.environ.get("BENCH_PG", "nccl")
    if pg != "nccl":
        local = local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    comm_note = None
    if world > 1:
        # The multi-rank paths have never run on more than one GPU (no such node was available to any round of this
        # build).  Two guards so that a stall there costs one run, not the series: a watchdog ends a run that has not
        # finished within BENCH_WATCHDOG_S seconds (rank 0 still prints a JSON line saying so), and leaves a marker on
        # the node; a later run that finds the marker drives the exchange with torch.distributed collectives instead of
        # the in-library RCCL path (PFX_SHARD_RCCL=0) and says so in its record.
        marker = os.path.join(os.environ.get("TMPDIR", "/tmp"), "pfx_bench_library_rccl_stalled")
        if os.path.exists(marker) and "PFX_SHARD_RCCL" not in os.environ:
            os.environ["PFX_SHARD_RCCL"] = "0"
            comm_note = "torch.distributed collectives (an earlier run on this node stalled with the in-library RCCL exchange)"
        limit = float(os.environ.get("BENCH_WATCHDOG_S", "480"))

        def watchdog():
            time.sleep(limit)
            sys.stderr.write(f"bench.py: rank {rank} not finished after {limit:.0f} s: giving up\n")
            try:
                if pg == "nccl" and os.environ.get("PFX_SHARD_RCCL", "1") != "0":
                    open(marker, "w").write("stalled\n")
                if rank == 0:
                    emit({"metric": "particle-updates/sec (range update+resample) at N=16M", "value": None, "unit": "particle-updates/s",
                          "n_gpus": world, "error": f"multi-rank run not finished after {limit:.0f} s (watchdog)",
                          "comm": comm_note or ("in-library RCCL" if pg == "nccl" else pg)})
            finally:
                os._exit(3)

        import threading
        threading.Thread(target=watchdog, daemon=True).start()
        if pg == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(pg)

    if args.workload == "config5":
        out = measure_slam(args.slam_filters, args.slam_bcns, args.steps, args.warmup, world, rank, local, dist, torch,
                           with_cpu=(world == 1 and not args.no_cpu_baseline))
        if rank == 0:
            emit(out)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    from slam3d_amd import particlefilter as pfm
    if args.workload == "loc_batch":        # many Loc filters in one set of launches, by itself (one GPU)
        nn = args.n if args.n != 16000000 else 10000
        emit(measure_loc_batch(pfm, torch, local, args.batch_filters, nn, args.steps, args.warmup))
        return
    if args.total_particles > 0:
        args.n = args.total_particles // world
    K, W, n = args.steps, args.warmup, args.n
    force_sharded = os.environ.get("BENCH_FORCE_SHARDED") == "1"
    if world == 1 and force_sharded:
        dist.init_process_group(pg, rank=0, world_size=1, init_method="tcp://127.0.0.1:29533",
                                **({"device_id": torch.device("cuda", local)} if pg == "nccl" else {}))

    rec, per = measure_loc(pfm, torch, dist, world, rank, local, n, K, W, args.force_branch, args, headline=True)

    # N > 1: the north star's other multi-GPU numbers ride on the same line (every rank takes part in measuring them)
    multi = None
    if world > 1 and not args.no_extras:
        multi = multi_gpu_extras(pfm, torch, dist, world, rank, local, args)

    if rank == 0:
        traffic_tab, traffic_stale = {}, None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            with open(tpath) as fh:
                traffic_tab = json.load(fh)
            src = traffic_tab.get("_source", {})
            traffic_stale = src.get("kernel_sources_sha") != kernel_sources_sha()
        dominant = max(per, key=lambda k_: per[k_]["total_ms"]) if per else None
        roof = None
        if dominant:
            d = per[dominant]
            # HBM bytes of one launch from the PMC counters (two rocprofv3 passes of this command, tools/pmc_traffic.py;
            # measured at N = 16 M, per particle, scaled to this run's N)
            traffic, tsrc = None, None
            t = traffic_tab.get(dominant, {})
            if "hbm_bytes_per_particle" in t:
                traffic = t["hbm_bytes_per_particle"] * n
                tsrc = ("profiles/pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), per particle at "
                        "N = 16 M; a constant of the profile, not re-measured in this run")
            roof = {"bound": "hbm", "kernel": KERNEL_OF[dominant],
                    "variant": dominant, "achieved": d["achieved_GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": d["frac"], "traffic": traffic, "traffic_source": tsrc,
                    "traffic_stale": traffic_stale,     # True: the kernel sources have changed since the PMC passes
                    "avg_launch_ms": d["avg_ms"],
                    "launches": d["launches"], "alg_bytes_per_particle": d["alg_bytes_per_particle"]}
            if traffic:
                roof["dram_GBs"] = traffic / (d["avg_ms"] * 1e-3) / 1e9      # what the kernel really moves per second
            if dominant.startswith("resample_"):
                roof["reference_passes_bytes_per_particle"] = d["reference_passes_bytes_per_particle"]
                roof["reference_passes_GBs"] = d["reference_passes_GBs"]
                roof["note"] = ("one launch does the reference's resample AND the next deposit's streaming pass; the offspring "
                                "stay in registers in between, so the launch is priced at the 40 B/particle it must move "
                                "(rounds before 2k priced it at the 84 B of the two passes it replaces: reference_passes_GBs, "
                                "above the HBM peak by now); `traffic` / dram_GBs is what it really moves.  Bound by how much of "
                                "its arithmetic (two Philox blocks, four Box-Muller pairs per offspring) the resident waves can "
                                "overlap with the dependent loads record -> weights -> ancestors, DESIGN.md 4.2")
        extra = None
        if world == 1 and not args.no_extras and not force_sharded:
            extra = {"note": f"same scene and step as the headline, {PRE_ROLL} pre-roll + 10 warm-up steps, no per-kernel events",
                     "n_sweep": [], "always_resample": None, "transient": None, "tracking": None, "loc_batch": None, "config5": None,
                     "config5_b12": None}
            for nn, kk in ((1000000, 100), (1 << 24, 50), (1 << 26, 30)):
                r, _ = measure_loc(pfm, torch, dist, 1, 0, local, nn, kk, 10, -1, args, headline=False)
                extra["n_sweep"].append(r)
            r, _ = measure_loc(pfm, torch, dist, 1, 0, local, n, 50, 10, 1, args, headline=False)
            r["force_branch"] = 1
            extra["always_resample"] = r
            r, _ = measure_loc(pfm, torch, dist, 1, 0, local, n, 20, 5, -1, args, headline=False, pre_roll=40)
            r["pre_roll_steps"] = 40
            extra["transient"] = r
            r, _ = measure_loc(pfm, torch, dist, 1, 0, local, n, 50, 10, -1, args, headline=False, ranges_only=True)
            r["note"] = ("the same step with the second deposit a depositRange on the measured range to beacon (k+1)%8 instead of "
                         "SURVEY's depositRssi (a fixed 1.5 m +- 1.5 m, which no beacon of this scene is within): a filter that "
                         "tracks -- see final_position_error_m -- and seldom resamples")
            extra["tracking"] = r
            if hasattr(pfm, "DeviceLocBatch"):
                extra["loc_batch"] = [measure_loc_batch(pfm, torch, local, F, nn, kk) for F, nn, kk in ((1600, 10000, 200), (16, 1000000, 100))]
            extra["config5"] = measure_slam(512, 4, 60, 10, 1, 0, local, dist, torch, with_cpu=not args.no_cpu_baseline)
            extra["config5_b12"] = measure_slam(512, 12, 60, 12, 1, 0, local, dist, torch, with_cpu=False)     # B = 4 and 12
        elif multi is not None:
            extra = multi
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline()
        # the whole step against the roofline, priced as the reference's passes: one deposit with motion, one
        # without, and the resamplings the decisions called for
        step_bytes = 44 + 28 + 40 * 2 * (rec["resample_fraction"] or 0.0)
        step_gbs = step_bytes * n * world / (rec["ms_per_step"] * 1e-3) / 1e9
        step_roof = {"alg_bytes_per_particle_step": step_bytes, "achieved": step_gbs, "peak": HBM_PEAK_GBS * world, "unit": "GB/s",
                     "frac": step_gbs / (HBM_PEAK_GBS * world),
                     "note": "depositRange (44 B/particle) + depositRssi (28) + resample_fraction x 2 resamplings (40 each), "
                             "over ms_per_step: launch gaps, the two small kernels per deposit and fusion gains all included"}
        # ... and in the bytes the launches really moved: PMC bytes per particle of each launch class (profiles/
        # pmc_traffic.json) x the launches of that class the device-side branch log counted in the timed region
        lc = rec.get("launch_classes")
        if lc and all(("hbm_bytes_per_particle" in traffic_tab.get(c, {})) or lc[c] == 0 for c in lc):
            real = sum(traffic_tab[c]["hbm_bytes_per_particle"] * cnt for c, cnt in lc.items() if cnt) / K
            real_gbs = real * n * world / (rec["ms_per_step"] * 1e-3) / 1e9
            step_roof["real_bytes_per_particle_step"] = real
            step_roof["real_GBs"] = real_gbs
            step_roof["real_frac"] = real_gbs / (HBM_PEAK_GBS * world)
            step_roof["real_note"] = ("HBM bytes the step's launches moved (rocprofv3 PMC per launch class x the class counts of "
                                      "this run), over ms_per_step; the fused launches never move the 44/28 B pass that the "
                                      "SURVEY pricing above charges them for")
            step_roof["traffic_stale"] = traffic_stale
        err = rec["final_position_error_m"]
        out = {
            "metric": "particle-updates/sec (range update+resample) at N=16M",
            "value": rec["value"], "unit": "particle-updates/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": rec["ms_per_step"],
            "higher_is_better": True, "scaling": "strong" if args.total_particles > 0 else "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "config3: single-tag particleFilterLoc, 8 beacons, per step depositVio + depositRange "
                                   "+ depositRssi (2 deposits = 2N particle-updates)",
                       "particles_per_gpu": n, "particles_total": n * world,
                       "parallelism": "single GPU" if world == 1 else f"one filter sharded by particle range over {world} GPUs, "
                                                                       "chunk sums exchanged per deposit, neighbour "
                                                                       "send/recv of migrating offspring per resampling",
                       "exchange": comm_note or rec.get("exchange"),
                       "force_branch": args.force_branch, "resample_fraction": rec["resample_fraction"],
                       "launch_classes": lc,
                       "deferred_resampling": not args.undeferred or world > 1 or force_sharded,
                       "pose_every_step": POSE_EVERY_STEP, "pre_roll_steps": PRE_ROLL,
                       "rng": "Philox4x32-10 key 123456789", "final_position_error_m": err,
                       "scene_note": "SURVEY's config-3 scene is self-contradictory on purpose of traffic: every depositRssi asserts "
                                     "1.5 +- 1.5 m to a beacon >= 5 m from the truth, so each one gates the whole cloud and forces "
                                     f"the 5 % respawn around that beacon; the filter does NOT track (estimate {err:.1f} m from the "
                                     "truth at the end of this run) and about half the deposits resample.  A traffic case; "
                                     "extra.tracking is the same step on ranges only, where the filter tracks"},
            "roofline": roof, "step_roofline": step_roof, "kernels": per, "cpu_baseline": cpu, "extra": extra,
        }
        emit(out)
    if world > 1 or force_sharded:
        if world > 1:
            dist.barrier()
        dist.destroy_process_group()


def multi_gpu_extras(pfm, torch, dist, world, rank, local, args):
    """world > 1: BASELINE config 4 (ONE filter of 67 108 864 particles over all ranks, strong scaling, against the same
    filter unsharded on rank 0's GPU, measured in this job while the other ranks wait) and config 5 (512 SLAM tags per
    GPU, no communication).  Returns the `extra` object on rank 0."""
    total = 1 << 26
    out = {"note": "measured in the same job after the headline; every rank takes part", "config4": None, "config5": None}
    single = None
    if rank == 0:
        single, _ = measure_loc(pfm, torch, dist, 1, 0, local, total, 30, 10, args.force_branch, args, headline=False)
    dist.barrier()
    r, _ = measure_loc(pfm, torch, dist, world, rank, local, total // world, 50, 10, args.force_branch, args, headline=False)
    if rank == 0:
        r.update({"scaling": "strong", "particles_total": (total // world) * world, "n_gpus": world,
                  "single_gpu_ms_per_step": single["ms_per_step"], "single_gpu_value": single["value"],
                  "speedup_vs_single_gpu": single["ms_per_step"] / r["ms_per_step"],
                  "note": "BASELINE config 4: one particleFilterLoc of 2^26 particles sharded by particle range; the single-GPU "
                          "figure is the unsharded filter of the same size on rank 0's GPU, same scene, same job"})
        out["config4"] = r
    dist.barrier()
    c5 = measure_slam(args.slam_filters, args.slam_bcns, 60, 10, world, rank, local, dist, torch, with_cpu=False)
    if rank == 0:
        out["config5"] = c5
    return out if rank == 0 else None


def measure_loc_batch(pfm, torch, local, F, n, K, W=20):
    """Many independent Loc filters in one set of launches ("also many Loc tags"): F filters of n
    particles each, every filter its own scene phase; one STEP = depositVio + depositRange + depositRssi on every filter
    (2 F n particle-updates), through pfx_loc_batch_*."""
    bx, truth, rng = scene(PRE_ROLL + K + W + F + 1)
    batch = pfm.DeviceLocBatch(F, n, seed=123456789, device=local)
    off = np.arange(F) % 64                   # filter f runs the scene from step off[f] on: the tags are not in lockstep

    def one(k):
        kk = k + off
        batch.deposit_vio(np.full(F, 1000.0 + 0.1 * k), truth[kk, 0], truth[kk, 1], truth[kk, 2], np.zeros(F, dtype=np.float32))
        b = kk % 8
        batch.deposit_range(bx[b, 0], bx[b, 1], bx[b, 2], rng[kk, 0], np.full(F, 0.1, dtype=np.float32))
        b = (kk + 1) % 8
        batch.deposit_rssi(bx[b, 0], bx[b, 1], bx[b, 2])

    batch.deposit_vio(np.full(F, 999.9), truth[off, 0], truth[off, 1], truth[off, 2], np.zeros(F, dtype=np.float32))
    batch.deposit_range(bx[off % 8, 0], bx[off % 8, 1], bx[off % 8, 2], rng[off, 0], np.full(F, 0.1, dtype=np.float32))
    pre = min(PRE_ROLL, 100)
    with gc_paused():
        for k in range(pre + W):
            one(k)
        d0, r0 = batch.branch_counts()          # (synchronises)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(pre + W, pre + W + K):
            one(k)
        batch.sync()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    d1, r1 = batch.branch_counts()
    frac = (r1 - r0) / max(1, d1 - d0)
    poses = batch.get_tag_loc()
    err = np.linalg.norm(np.stack([poses[2], poses[3], poses[4]], axis=1).astype(np.float64) - truth[pre + W + K - 1 + off], axis=1)
    step_bytes = 44 + 28 + 40 * 2 * frac
    gbs = step_bytes * n * F / (dt / K) / 1e9
    return {"filters": F, "particles_per_filter": n, "value": 2.0 * n * F * K / dt, "unit": "particle-updates/s", "steps": K,
            "ms_per_step": dt / K * 1e3, "us_per_filter_step": dt / K / F * 1e6,
            "resample_fraction": frac, "median_position_error_m": float(np.median(err)),
            "step_roofline_frac": gbs / HBM_PEAK_GBS,
            "note": "pfx_loc_batch_*: F independent particleFilterLoc filters of equal N, one set of launches per deposit of the "
                    "whole batch (grid dimension = filter), per-filter Philox stream, per-filter decision on the device"}


if __name__ == "__main__":
    main()
