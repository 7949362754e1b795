#!/usr/bin/env python3
"""One rank of a sharded-filter test run (launched by torch.distributed.run from tests/test_sharded*.py).

    sharded_worker.py --backend oracle|gpu --particles N --steps K --out DIR [--force F] [--pg gloo|nccl]

Every rank drives the same event list through a ShardedLocFilter and saves its shard (w,x,y,z,theta), the
poses after every deposit and the branch of every deposit to DIR/rank<r>.npz.
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def events(n_steps):
    import _synth
    bx, truth, rng = _synth.synthetic_scene(n_steps)
    ev = []
    for k in range(n_steps):
        ev.append(("vio", 10.0 + 0.1 * k, float(truth[k, 0]), float(truth[k, 1]), float(truth[k, 2]), 0.0))
        b = k % 8
        ev.append(("range", float(bx[b, 0]), float(bx[b, 1]), float(bx[b, 2]), float(rng[k, 0]), 0.1))
        if k % 3 == 2:
            b2 = (k + 1) % 8
            ev.append(("rssi", float(bx[b2, 0]), float(bx[b2, 1]), float(bx[b2, 2]), -55))
    return ev


def concentrate(w, n, first_global):
    """weight 1e-30 outside the global particles [n/8 + 10, n/4 - 10); w holds the particles from first_global on"""
    g = np.arange(first_global, first_global + w.shape[0])
    w[(g < n // 8 + 10) | (g >= n // 4 - 10)] = np.float32(1e-30)


def drive(pf, evs, on_deposit=None, pose_every=1):
    """pose_every = K: the pose is read after every K-th deposit only, so that the deposits in between run with
    their resampling deferred (fused with the next deposit)"""
    poses = []
    k = 0
    for e in evs:
        if e[0] == "vio":
            pf.depositVio(*e[1:])
            continue
        if e[0] == "range":
            pf.depositRange(*e[1:])
        else:
            pf.depositRssi(*e[1:])
        k += 1
        if k % pose_every == 0:
            poses.append([float(v) for v in pf.getTagLoc()])
        if on_deposit:
            on_deposit(pf)
    return np.array(poses)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", default="oracle")
    ap.add_argument("--pg", default="gloo")
    ap.add_argument("--particles", dest="n", type=int, default=20000)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--seed", type=int, default=123456789)
    ap.add_argument("--force", type=int, default=-1)
    ap.add_argument("--pose-every", type=int, default=1)
    ap.add_argument("--concentrate", type=int, default=0,
                    help="after the spawn, leave weight only on global particles [n/8 + 10, n/4 - 10): one shard of eight then "
                         "feeds all the others (oracle backend)")
    ap.add_argument("--stage-slack", type=float, default=1.0, help="staging area in units of the shard; 0 = the library's default (world)")
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    if a.stage_slack == 0:
        a.stage_slack = None

    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    if a.pg == "nccl":
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    dist.init_process_group(a.pg)
    from slam3d_amd.sharded import ShardedLocFilter
    factory = None
    if a.backend == "oracle":
        from _oracle_shard import OracleShard
        factory = OracleShard
    dev = int(os.environ.get("LOCAL_RANK", "0")) if a.pg == "nccl" else 0
    # staging area = slack x the shard's own size: a shard that feeds all eight needs room for seven shards' worth
    pf = ShardedLocFilter(a.n, seed=a.seed, device=dev, backend_factory=factory, stage_slack=8.0 if a.concentrate else a.stage_slack)
    pf.set_force_branch(a.force)
    branches, moved = [], []

    def tap(f):
        branches.append(-1 if f.last_branch is None else int(f.last_branch))
        moved.append(f.exchanged_particles)
        if a.concentrate and len(branches) == 1:
            concentrate(f.shard.pf.arrays()[0], a.n, f.T[rank])
    poses = drive(pf, events(a.steps), tap, a.pose_every)
    w, x, y, z, th = pf.download_local()
    np.savez(os.path.join(a.out, f"rank{rank}.npz"), w=w, x=x, y=y, z=z, th=th, poses=poses,
             branches=np.array(branches), moved=np.array(moved), T=np.array(pf.T),
             library_comm=int(getattr(pf, "uses_library_comm", False)))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
