#!/usr/bin/env python3
"""bench.py -- BASELINE.json headline metric: particle-updates/sec (range update + resample) at N = 16 M.

Workload (BASELINE.json configs[2], SURVEY.md 8d "config 3", the roofline case): one particleFilterLoc of
N = 16 000 000 particles per GPU, 8 beacons at (+-5, +-5, {0.5, 3.0}) m, truth on a 2 m circle moving
5 cm per step, ranges = true distance + N(0, 0.1^2).  One STEP = depositVio + depositRange(beacon k%8) +
depositRssi(beacon (k+1)%8): two deposits, i.e. 2N particle-updates (one particle through one
depositRange/Rssi call: VIO commit, gate, sums, decision, resample-or-normalise).  The first deposit of a
step carries the VIO motion, the second has zero motion.  Particles live in HBM the whole time; the only
host->device traffic in the timed region is kernel arguments.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--particles PER_GPU] [--force-branch -1|0|1]

N > 1 (launched by torch.distributed.run, one rank per GPU): ONE filter of N x 16 M particles sharded by
particle range; per deposit an all-gather of the per-GPU weight/pose sums (RCCL) gives every rank the
global decision and its CDF base; see slam3d_amd/sharded.py.  Weak scaling: per-GPU work fixed.

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

# SURVEY.md 8d / BASELINE.md: algorithmic (compulsory) HBM bytes per particle, SoA f32
# algorithmic bytes per particle.  The two fused kernels do the work of the resample (40) AND of
# the following deposit's streaming pass (44 with motion / 28 without) in one launch; what they really move is
# 40 B per particle (read w,x,y,z,theta of the ancestors, write them for the offspring) -- see "traffic".
ALG_BYTES = {"stream": 44, "stream_static": 28, "resample": 40, "resample_stream": 84, "resample_static": 68}
KERNEL_OF = {"stream": "k_loc_stream<motion,gate>", "stream_static": "k_loc_stream<gate>", "resample": "k_loc_resample",
             "resample_stream": "k_loc_deposit<motion> (resampling side)", "resample_static": "k_loc_deposit<static> (resampling side)"}
# (a sharded filter runs the same two sides in k_loc_deposit_sharded)
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: 8 TB/s spec


def scene(n_steps):
    import _synth
    return _synth.synthetic_scene(n_steps, seed=42)


POSE_EVERY_STEP = False     # --pose-every-step: the replay drivers' pattern (a getTagLoc per VIO row); not the headline metric


def step(pf, k, bx, truth, rng):
    """one STEP of the workload on any object with the wrapper interface"""
    pf.depositVio(1000.0 + 0.1 * k, float(truth[k, 0]), float(truth[k, 1]), float(truth[k, 2]), 0.0)
    if POSE_EVERY_STEP:
        pf.getTagLoc()
    b = k % 8
    pf.depositRange(float(bx[b, 0]), float(bx[b, 1]), float(bx[b, 2]), float(rng[k, 0]), 0.1)
    b = (k + 1) % 8
    pf.depositRssi(float(bx[b, 0]), float(bx[b, 1]), float(bx[b, 2]), -60)


# ------------------------------------------------------------------ CPU baseline (rank 0, N = 1 only)
class _RefLoc:
    """The COMPILED REFERENCE (oracle/_ref/libpf_ref_glibc_n1m.so: the reference's own sources built with
    PF_N_TAG_LOC = 1 000 000) behind the wrapper interface."""
    N = 1000000

    def __init__(self, path):
        self.L = C.CDLL(path)
        f, d = C.c_float, C.c_double
        self.L.particleFilterLoc_depositVio.argtypes = [C.c_void_p, d, f, f, f, f]
        self.L.particleFilterLoc_depositRange.argtypes = [C.c_void_p, f, f, f, f, f]
        self.L.particleFilterLoc_depositRssi.argtypes = [C.c_void_p, f, f, f, C.c_int]
        self.L.particleFilterSeed_set.argtypes = [C.c_uint]
        self.buf = C.create_string_buffer(2 * self.N * 20 + 64)
        self.p = C.cast(self.buf, C.c_void_p)
        self.L.particleFilterSeed_set(123456789)
        self.L.particleFilterLoc_init(self.p)

    def depositVio(self, *a):
        self.L.particleFilterLoc_depositVio(self.p, *a)

    def depositRange(self, *a):
        self.L.particleFilterLoc_depositRange(self.p, *a)

    def depositRssi(self, *a):
        self.L.particleFilterLoc_depositRssi(self.p, *a)


def cpu_baseline(budget_s=12.0):
    """Time the same workload on ONE host core, on a bounded sample (N = 1 M particles, as many steps as fit
    the budget).  Prefers the compiled reference (kind "reference"); else the oracle restatement ("port")."""
    n = 1000000
    bx, truth, rng = scene(64)
    ref_so = os.path.join(ROOT, "oracle", "_ref", "libpf_ref_glibc_n1m.so")
    if os.path.exists(ref_so):
        pf, kind = _RefLoc(ref_so), "reference"
    else:
        from oracle import oracle as ora
        pf, kind = ora.OracleLoc(n, rng_mode=ora.RNG_GLIBC, seed=123456789, acc=ora.ACC_F32), "port"
    pf.depositVio(999.9, float(truth[0, 0]), float(truth[0, 1]), float(truth[0, 2]), 0.0)
    pf.depositRange(float(bx[0, 0]), float(bx[0, 1]), float(bx[0, 2]), float(rng[0, 0]), 0.1)   # init, untimed
    t0 = time.perf_counter()
    k = 0
    while k < 60:
        step(pf, k, bx, truth, rng)
        k += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return {"value": 2.0 * n * k / dt, "unit": "particle-updates/s", "cores": 1, "kind": kind,
            "sample": f"same scene and step, N=1,000,000 particles (1/16 of the GPU workload), {k} steps, "
                      f"{dt:.1f} s, serial rand_r stream, float32 as the reference"}


# ------------------------------------------------------------------ config 5: batched SLAM filters, no comms
def run_slam(args, world, rank, local, dist, torch):
    """BASELINE.json configs[4]: independent particleFilterSlam tags (100 tag particles x B beacons x 1000), sharded
    over the GPUs with no communication.  One STEP = depositTagVio + depositRange(beacon k % B) for every tag of the
    rank's batch: F x 100 x 1000 beacon-particle updates (+ F x 100 tag particles).  Reported beside the headline
    metric; the default bench run stays the roofline case (config 3)."""
    from slam3d_amd import slam
    F, B, K, W = args.slam_filters, args.slam_bcns, args.steps, args.warmup
    batch = slam.DeviceSlamBatch(F, 100, 1000, max_bcns=B, seed0=1000003 * (rank + 1), device=local)
    for _ in range(B):
        batch.add_bcn_all()
    bx, truth, rng = scene(K + W + B + 1)
    bpos = bx[:B] if B <= 8 else np.concatenate([bx, bx[:B - 8] * np.float32(0.6)])
    ones = np.ones(F, dtype=np.float32)

    def one(k):
        batch.deposit_tag_vio_batch(np.full(F, 1000.0 + 0.1 * k), ones * truth[k, 0], ones * truth[k, 1], ones * truth[k, 2])
        b = k % B
        d = np.float32(np.linalg.norm(truth[k].astype(np.float64) - bpos[b].astype(np.float64)) + (rng[k, 0] - np.linalg.norm(
            truth[k].astype(np.float64) - bx[k % 8].astype(np.float64))))
        batch.deposit_range_batch(np.full(F, b, dtype=np.int32), ones * d, ones * np.float32(0.1))

    def barrier():
        if world > 1:
            dist.barrier()
        batch.sync()
        torch.cuda.synchronize()

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
    if world > 1:
        t = torch.tensor([dt], device="cuda" if os.environ.get("BENCH_PG", "nccl") == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        upd = 100.0 * 1000.0 * F * world * K
        # compulsory traffic of a deposit on the measured beacon: gate pass 20 B read + 4 B write, row pass 20 B read +
        # 20 B write (resample) or 4 B (normalise) -> 48..64 B per beacon particle; 64 used for the fraction
        print(json.dumps({
            "metric": "particle-updates/sec, batched particleFilterSlam (config 5)", "value": upd / dt, "unit": "particle-updates/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"config5: {F} SLAM tags per GPU x 100 tag particles x {B} beacons x 1000, per step depositTagVio + "
                                   "depositRange on one beacon of every tag", "filters_per_gpu": F, "beacons": B,
                       "filter_updates_per_s": F * world * K / dt, "parallelism": "tags sharded over GPUs, no communication"},
            "roofline": {"bound": "hbm", "achieved": 64.0 * upd / world / dt / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": 64.0 * upd / world / dt / 1e9 / HBM_PEAK_GBS, "traffic": None,
                         "note": "whole-step rate at 64 algorithmic B per beacon particle (gate 24 B + row resample 40 B)"},
            "cpu_baseline": None}))


# ------------------------------------------------------------------ GPU run
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
    ap.add_argument("--event-stride", type=int, default=3,
                    help="time every n-th launch of each O(N) kernel class (3: coprime with the 8-beacon cycle of the scene)")
    ap.add_argument("--no-kernel-events", action="store_true", help="timed region without per-kernel HIP events (no roofline)")
    ap.add_argument("--pose-every-step", action="store_true", help="also call getTagLoc after every depositVio (driver pattern)")
    ap.add_argument("--undeferred", action="store_true", help="one resample kernel per deposit (pfx_loc_set_deferred(0))")
    ap.add_argument("--workload", default="config3", choices=["config3", "config5"])
    ap.add_argument("--slam-filters", type=int, default=512, help="config5: SLAM tags per GPU")
    ap.add_argument("--slam-bcns", type=int, default=4, help="config5: beacons per tag")
    args = ap.parse_args()
    global POSE_EVERY_STEP
    POSE_EVERY_STEP = args.pose_every_step

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
    if world > 1:
        if pg == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(pg)

    if args.workload == "config5":
        run_slam(args, world, rank, local, dist, torch)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    from slam3d_amd import particlefilter as pfm
    if args.total_particles > 0:
        args.n = args.total_particles // world
    K, W, n = args.steps, args.warmup, args.n
    bx, truth, rng = scene(K + W + 1)

    force_sharded = os.environ.get("BENCH_FORCE_SHARDED") == "1"      # rehearsal: the sharded host logic on one rank
    if world == 1 and force_sharded:
        dist.init_process_group(pg, rank=0, world_size=1, init_method="tcp://127.0.0.1:29533",
                                **({"device_id": torch.device("cuda", local)} if pg == "nccl" else {}))
    if world == 1 and not force_sharded:
        pf = pfm.DeviceParticleFilterLoc(n, seed=123456789, stream_id=0, device=local)
    else:
        from slam3d_amd.sharded import ShardedLocFilter
        pf = ShardedLocFilter(n * world, seed=123456789, device=local)
    pf.set_force_branch(args.force_branch)
    if args.undeferred and world == 1 and not force_sharded:
        pf.set_deferred(False)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # init (spawn) + warm-up, untimed
    pf.depositVio(999.9, float(truth[0, 0]), float(truth[0, 1]), float(truth[0, 2]), 0.0)
    pf.depositRange(float(bx[0, 0]), float(bx[0, 1]), float(bx[0, 2]), float(rng[0, 0]), 0.1)
    for k in range(W):
        step(pf, k, bx, truth, rng)
    barrier()

    if not args.no_kernel_events:
        # the O(N) kernels only: the two tiny prefix / decision launches of every deposit are not reported, and timing
        # them too costs ~12 us of GPU time per step
        pf.profile_only("stream", "stream_static", "resample", "deposit", "deposit_static")
        pf.profile_stride(args.event_stride)     # every event_stride-th launch of a class carries events (~3 us each)
        pf.profile_begin(4 * K * 4 + 16)
    barrier()
    t0 = time.perf_counter()
    for k in range(W, W + K):
        step(pf, k, bx, truth, rng)
    barrier()
    dt = time.perf_counter() - t0

    kinds, ms = pf.profile_records()
    branches, total_dec = pf.branch_log()
    branches = branches[-2 * K:]
    pose = pf.getTagLoc()
    err = float(np.linalg.norm(np.array(pose[2:5], dtype=np.float64) - truth[W + K - 1]))

    if world > 1:
        t = torch.tensor([dt], device="cuda" if pg == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        names = pfm.DeviceParticleFilterLoc.KINDS
        per = {}
        # Launch classes.  k_loc_resample after a normalise-only decision exits at once: not counted.  With deferred
        # resampling a deposit is ONE launch of k_loc_deposit, which resamples-and-deposits or only streams according
        # to the previous decision (known on the device only); an empty launch takes a few microseconds and a streaming
        # pass well under the fused one, so the launches of a class are told apart by the device-side branch log
        # (deposit j of the timed region follows decision j - 1) and, where that does not line up, by duration.
        classes = {}
        for nm in ("stream", "stream_static", "resample"):
            sel = ms[kinds == names.index(nm)]
            if sel.shape[0]:
                sel = sel[sel > 0.25 * sel.max()]
            classes[nm] = sel
        full_log = pf.branch_log()[0]       # rank 0's log; a sharded filter takes the same decision on every rank
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
        for nm, sel in classes.items():
            if sel.shape[0]:
                avg_ms = float(sel.mean())
                gbs = ALG_BYTES[nm] * n / (avg_ms * 1e-3) / 1e9
                per[nm] = {"launches": int(sel.shape[0]), "avg_ms": avg_ms, "alg_bytes_per_particle": ALG_BYTES[nm],
                           "achieved_GBs": gbs, "frac": gbs / HBM_PEAK_GBS, "total_ms": float(sel.sum())}
        dominant = max(per, key=lambda k_: per[k_]["total_ms"]) if per else None
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if dominant and os.path.exists(tpath) and world == 1 and n == 16000000:
            with open(tpath) as fh:
                traffic = json.load(fh).get(dominant, {}).get("hbm_bytes_per_launch")
        roof = None
        if dominant:
            d = per[dominant]
            roof = {"bound": "hbm", "kernel": KERNEL_OF[dominant],
                    "variant": dominant, "achieved": d["achieved_GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": d["frac"], "traffic": traffic, "avg_launch_ms": d["avg_ms"], "launches": d["launches"],
                    "alg_bytes_per_particle": d["alg_bytes_per_particle"]}
            if traffic:
                roof["dram_GBs"] = traffic / (d["avg_ms"] * 1e-3) / 1e9      # what the kernel really moves per second
            if dominant.startswith("resample_"):
                roof["note"] = ("one launch does the reference's resample (40 algorithmic B/particle) and the next deposit's "
                                "streaming pass (44 / 28) while moving ~40.6 B; this side of the kernel is limited by "
                                "vector-ALU issue (SQ counters: 466 VALU instructions per offspring, ~99 % busy), not by HBM")
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline()
        frac_a = float(branches.mean()) if branches.shape[0] else None
        out = {
            "metric": "particle-updates/sec (range update+resample) at N=16M",
            "value": 2.0 * n * world * K / dt, "unit": "particle-updates/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": dt / K * 1e3,
            "higher_is_better": True, "scaling": "strong" if args.total_particles > 0 else "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "config3: single-tag particleFilterLoc, 8 beacons, per step depositVio + depositRange "
                                   "+ depositRssi (2 deposits = 2N particle-updates)",
                       "particles_per_gpu": n, "particles_total": n * world,
                       "parallelism": "single GPU" if world == 1 else f"one filter sharded by particle range over {world} GPUs, "
                                                                       "RCCL all-gather of per-GPU sums per deposit, neighbour "
                                                                       "send/recv of migrating offspring per resampling",
                       "force_branch": args.force_branch, "resample_fraction": frac_a,
                       "deferred_resampling": not args.undeferred or world > 1 or force_sharded,
                       "pose_every_step": POSE_EVERY_STEP,
                       "rng": "Philox4x32-10 key 123456789", "final_position_error_m": err},
            "roofline": roof, "kernels": per, "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
