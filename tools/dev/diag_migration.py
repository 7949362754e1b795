"""How many offspring change shard per resampling in bench.py's scene (config 3 / config 4)?  In-process shards on one GPU
(tests/test_gpu_shard_protocol.py::InProcessShards), world 2 / 4 / 8 (diagnostic)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from test_gpu_shard_protocol import InProcessShards
from slam3d_amd.sharded import exchange_plan


class Counting(InProcessShards):
    """... that also keeps, per resampling, the records every shard sends to every other"""

    def _ship(self, F, deposit):
        m = np.zeros((self.world, self.world), dtype=np.int64)
        for g in range(self.world):
            for peer, _off, cnt in exchange_plan(F, self.T, g)[0]:
                m[g, peer] += cnt
        self.pairs.append(m)
        return super()._ship(F, deposit)

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 460
bx, truth, rng = bench.scene(steps + 2)
for world in (2, 4, 8):
    s = Counting(n, world, slack=8.0)
    s.pairs = []
    s.depositVio(999.9, float(truth[0, 0]), float(truth[0, 1]), float(truth[0, 2]), 0.0)
    s.depositRange(float(bx[0, 0]), float(bx[0, 1]), float(bx[0, 2]), float(rng[0, 0]), 0.1)
    per = []
    for k in range(steps):
        m0, b0 = s.moved, len(s.branches)
        bench.step(s, k, bx, truth, rng)
        if k >= steps - 60:
            per.append((s.moved - m0, sum(s.branches[b0:])))
    s.settle()
    moved = np.array([p[0] for p in per], dtype=np.float64)
    rs = sum(p[1] for p in per)
    print(f"N={n} world={world}: last 60 steps: {rs} resamplings, offspring changing shard per step: mean {moved.mean() / n:.3f} N, "
          f"max {moved.max() / n:.3f} N; per resampling {moved.sum() / max(1, rs) / n:.3f} N = {moved.sum() / max(1, rs) * 16 / 1e6:.1f} MB "
          f"(at N = 64 M: {moved.sum() / max(1, rs) / n * 67108864 * 16 / 1e6:.0f} MB over all links)", flush=True)
    late = s.pairs[-rs:] if rs else []
    if late:
        out_max = np.mean([m.sum(axis=1).max() for m in late]) / n          # the busiest sender
        in_max = np.mean([m.sum(axis=0).max() for m in late]) / n           # the busiest receiver
        link_max = np.mean([m.max() for m in late]) / n                     # the busiest ordered pair (one xGMI link, one direction)
        hops = max(abs(a - b) for m in late for a, b in zip(*np.nonzero(m)))
        print(f"    per resampling: busiest sender {out_max:.3f} N, busiest receiver {in_max:.3f} N, busiest pair {link_max:.3f} N "
              f"(at N = 64 M: {link_max * 67108864 * 16 / 1e6:.0f} MB on one link = {link_max * 67108864 * 16 / 153e9 * 1e3:.2f} ms at 153 GB/s); "
              f"farthest peer {hops} shards away", flush=True)
    del s
