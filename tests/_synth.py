"""Deterministic synthetic inputs shared by tests/golden/make_golden.py and the tests.

Everything is derived from splitmix64 in explicit uint64 arithmetic, so the same arrays come out
on any numpy version (numpy's Generator streams carry no cross-version guarantee) and the goldens
need not store their inputs.
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(seed: int, n: int) -> np.ndarray:
    with np.errstate(over="ignore"):
        idx = np.arange(1, n + 1, dtype=np.uint64)
        z = np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def u01(seed: int, n: int) -> np.ndarray:
    """float64 uniforms in [0,1) with 53 random bits"""
    return (splitmix64(seed, n) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def gauss(seed: int, n: int) -> np.ndarray:
    u1 = 1.0 - u01(seed, n)
    u2 = u01(seed ^ 0x5DEECE66D, n)
    return np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)


def loc_particles(n: int, pseed: int, kind: str) -> np.ndarray:
    """(n,5) float32 columns w,x,y,z,theta"""
    p = np.empty((n, 5), dtype=np.float32)
    # cloud around (1, 2, 0.5): some within the gate of the test beacons, some not
    p[:, 1] = (1.0 + 1.5 * gauss(pseed + 1, n)).astype(np.float32)
    p[:, 2] = (2.0 + 1.5 * gauss(pseed + 2, n)).astype(np.float32)
    p[:, 3] = (0.5 + 0.5 * gauss(pseed + 3, n)).astype(np.float32)
    p[:, 4] = ((u01(pseed + 4, n) * 2.0 - 0.5) * np.pi).astype(np.float32)   # [-pi/2, 3pi/2)
    u = u01(pseed + 5, n)
    if kind == "ones":
        p[:, 0] = 1.0
    elif kind == "mixed":
        p[:, 0] = np.choose((u * 4).astype(int), [1.0, 0.5, 0.1, 0.05]).astype(np.float32)
    elif kind == "degenerate":          # few heavy particles: ESS far below N/2
        p[:, 0] = np.where(u < 0.01, 1.0, 1e-3).astype(np.float32)
    elif kind == "lowmean":             # mean weight < 0.4 -> respawn when range < 4
        p[:, 0] = (0.05 + 0.3 * u).astype(np.float32)
    elif kind == "tight":               # concentrated headings -> small theta bandwidth
        p[:, 0] = np.where(u < 0.3, 1.0, 0.1).astype(np.float32)
        p[:, 4] = (0.7 + 0.01 * gauss(pseed + 6, n)).astype(np.float32)
    else:
        raise ValueError(kind)
    return p


# beacon = (bx, by, bz, range, stdRange); vio = (dt, dx, dy, dz, ddist)
LOC_FUNC_CASES = [
    dict(name="range_far", pseed=11, kind="ones", seed=7, beacon=(1.0, 2.0, 0.5, 3.5, 0.1), ops=["range"]),
    dict(name="range_near", pseed=12, kind="mixed", seed=7, beacon=(0.5, 1.0, 0.0, 1.5, 0.5), ops=["range"]),
    dict(name="range_wide", pseed=13, kind="mixed", seed=7, beacon=(-3.0, 2.0, 1.0, 4.2, 0.4), ops=["range"]),
    dict(name="vio_step", pseed=14, kind="mixed", seed=42, beacon=(1.0, 2.0, 0.5, 2.0, 0.1),
         vio=(0.11, 0.04, -0.03, 0.01, 0.051), ops=["vio"]),
    dict(name="vio_zero", pseed=15, kind="ones", seed=43, beacon=(1.0, 2.0, 0.5, 2.0, 0.1),
         vio=(0.0, 0.0, 0.0, 0.0, 0.0), ops=["vio"]),
    dict(name="resample_mixed", pseed=16, kind="mixed", seed=99, beacon=(1.0, 2.0, 0.5, 5.0, 0.1), ops=["resample"]),
    dict(name="resample_degenerate", pseed=17, kind="degenerate", seed=100, beacon=(1.0, 2.0, 0.5, 5.0, 0.1), ops=["resample"]),
    dict(name="resample_spawn", pseed=18, kind="lowmean", seed=101, beacon=(1.0, 2.0, 0.5, 2.5, 0.1), ops=["resample"]),
    dict(name="normalise_only", pseed=19, kind="ones", seed=102, beacon=(1.0, 2.0, 0.5, 5.0, 0.1), ops=["resample"]),
    dict(name="resample_tight", pseed=20, kind="tight", seed=103, beacon=(1.0, 2.0, 0.5, 5.0, 0.1), ops=["resample"]),
    dict(name="full_step", pseed=21, kind="mixed", seed=104, beacon=(1.0, 2.0, 0.5, 2.2, 0.1),
         vio=(0.2, -0.05, 0.02, 0.0, 0.06), ops=["vio", "range", "resample"]),
    dict(name="full_step_rssi_like", pseed=22, kind="ones", seed=105, beacon=(1.0, 2.0, 0.5, 1.5, 0.5),
         vio=(0.1, 0.01, 0.01, 0.0, 0.02), ops=["vio", "range", "resample"]),
]


def slam_scenario():
    """A short synthetic particleFilterSlam run: 3 beacons, a tag walking a circle, ranges to the
    true beacon positions with small deterministic noise, one beacon seen through depositRssi and one
    beacon that moves (depositBcnVio).  Events:
      ("vio", t, x, y, z) | ("bvio", b, t, x, y, z) | ("range", b, r, std) | ("rssi", b, rssi)
      | ("tagw", pseed): overwrite the tag weights with slam_tag_weights(pseed) -- a degenerate set, so
      that the next deposit takes the tag-resample path (pfResample.c:110-131: every initialised beacon is
      force-resampled); in free running the 100 tag particles stay too alike for that to trigger."""
    bpos = np.array([[3.0, 0.5, 1.0], [-2.0, 2.5, 0.5], [0.5, -3.0, 1.5]])
    noise = gauss(777, 400) * 0.05
    ev = []
    t = 100.0
    k = 0
    for step in range(36):
        ang = 0.15 * step
        pos = np.array([1.5 * np.cos(ang) - 1.5, 1.5 * np.sin(ang), 0.02 * step])
        for sub in range(3):
            t += 0.033
            a2 = ang + 0.05 * sub
            p2 = np.array([1.5 * np.cos(a2) - 1.5, 1.5 * np.sin(a2), 0.02 * step + 0.005 * sub])
            ev.append(("vio", t, float(np.float32(p2[0])), float(np.float32(p2[1])), float(np.float32(p2[2]))))
        pos = p2
        if step % 6 == 5:
            ev.append(("bvio", 1, t, float(np.float32(0.01 * step)), 0.0, 0.0))
        if step in (20, 30):
            ev.append(("tagw", 900 + step))
        b = step % 3
        d = float(np.linalg.norm(pos - bpos[b]))
        if step in (10, 22):
            ev.append(("rssi", b, -60))
        else:
            ev.append(("range", b, float(np.float32(d + noise[k])), float(np.float32(0.1))))
        k += 1
    # seed chosen so that no resample of this run walks off the end of its CDF (the reference reads past
    # its stack array there -- undefined behaviour, pfResample.c:70-71; with seed 2024 that happens at
    # deposit 24 and the compiled reference's output from then on depends on stack garbage)
    return {"seed": 2025, "num_bcns": 3, "events": ev}


def slam_tag_weights(pseed: int, n: int = 100) -> np.ndarray:
    u = u01(pseed, n)
    return np.where(u < 0.15, 1.0, 0.01).astype(np.float32)


def synthetic_scene(n_steps: int, seed: int = 42):
    """BASELINE.json config 3: 8 beacons at (+-5, +-5, {0.5, 3.0}), truth on a 2 m circle at
    z = 1.5 moving 5 cm per step, ranges = true distance + N(0, 0.1^2).  Returns (beacons[8,3], truth[n,3],
    ranges[n,2]) where ranges[k] are to beacons k%8 and (k+1)%8."""
    bx = np.array([[sx * 5.0, sy * 5.0, z] for z in (0.5, 3.0) for sx in (-1, 1) for sy in (-1, 1)], dtype=np.float32)
    ang = np.arange(n_steps) * (0.05 / 2.0)
    truth = np.stack([2.0 * np.cos(ang), 2.0 * np.sin(ang), np.full(n_steps, 1.5)], axis=1)
    g = gauss(seed, 2 * n_steps).reshape(n_steps, 2) * 0.1
    rng = np.empty((n_steps, 2), dtype=np.float32)
    for k in range(n_steps):
        for j in range(2):
            b = (k + j) % 8
            rng[k, j] = np.float32(np.linalg.norm(truth[k] - bx[b]) + g[k, j])
    return bx, truth.astype(np.float32), rng
