"""CSV replay of a VIO + UWB trace through a particleFilterLoc-style object.

Host-side mirror of the reference's replay loops: test/test.c:90-105 (the golden
test), csvlocalize/csvlocalize.c:64-88 (same loop, hard-coded paths) and
test/test.py:107-121 (the same through the Cython wrapper).  The filter object
only needs the wrapper's interface (particlefilter_wrapper.pyx:27-46):
``depositVio(t,x,y,z,dist)``, ``depositRange(bx,by,bz,range,std)``,
``getTagLoc() -> (status,t,x,y,z,theta)``.

All scalar arithmetic that the reference does in float is done in numpy
float32 here, so a bit-faithful filter reproduces the reference's output file
byte for byte.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np

f32 = np.float32

# test/test.c:24-27 (csvlocalize.c:29-40 uses the same values)
NUM_BCNS = 4
UWB_STD = f32(0.1)
UWB_BIAS = f32(0.2)

VIO_FILE = "test1_ParticleFilterLoc_vio.csv"
UWB_FILE = "test1_ParticleFilterLoc_uwb.csv"
DEPLOY_FILE = "test1_ParticleFilterLoc_deploy.csv"


def read_vio(path: str) -> List[Tuple[float, np.float32, np.float32, np.float32]]:
    """Rows ``t,y,z,x[,quat...]`` -> (t, x, y, z).  Column order per test/test.c:170-173
    (iOS reports VIO as y, z, x)."""
    rows = []
    with open(path, "r") as fh:
        for line in fh:
            c = line.rstrip("\n").split(",")
            if len(c) < 4:
                continue
            rows.append((float(c[0]), f32(float(c[3])), f32(float(c[1])), f32(float(c[2]))))
    return rows


def read_uwb(path: str, num_bcns: int = NUM_BCNS) -> List[Tuple[float, int, np.float32]]:
    """Rows ``t,beacon,range[,...]`` (test/test.c:178-189)."""
    rows = []
    with open(path, "r") as fh:
        for line in fh:
            c = line.rstrip("\n").split(",")
            if len(c) < 3:
                continue
            b = int(c[1])
            if not 0 <= b < num_bcns:
                raise ValueError(f"beacon id {b} out of range in {path}")
            rows.append((float(c[0]), b, f32(float(c[2]))))
    return rows


def read_deployment(path: str, num_bcns: int = NUM_BCNS) -> np.ndarray:
    """Rows ``beacon,y,z,x`` -> array[beacon] = (x, y, z) (test/test.c:191-205)."""
    dep = np.zeros((num_bcns, 3), dtype=np.float32)
    with open(path, "r") as fh:
        for k, line in enumerate(fh):
            if k >= num_bcns:
                break
            c = line.rstrip("\n").split(",")
            b = int(c[0])
            if not 0 <= b < num_bcns:
                raise ValueError(f"beacon id {b} out of range in {path}")
            dep[b, 1] = f32(float(c[1]))
            dep[b, 2] = f32(float(c[2]))
            dep[b, 0] = f32(float(c[3]))
    return dep


def format_tag_row(t: float, x, y, z, theta) -> str:
    """One output row; test/test.c:213 prints ``t, y, z, x, theta`` with %lf/%f."""
    return "%f,%f,%f,%f,%f\n" % (float(t), float(y), float(z), float(x), float(theta))


@dataclass
class ReplayResult:
    rows: List[Tuple[float, float, float, float, float]] = field(default_factory=list)  # t,x,y,z,theta
    n_vio: int = 0
    n_range: int = 0

    def csv_text(self) -> str:
        out = ["t,x,y,z,theta\n"]
        out.extend(format_tag_row(*r) for r in self.rows)
        return "".join(out)

    def as_array(self) -> np.ndarray:
        return np.array(self.rows, dtype=np.float64).reshape(-1, 5)


def replay_loc(pf, vio: Sequence, uwb: Sequence, deployment: np.ndarray,
               uwb_bias=UWB_BIAS, uwb_std=UWB_STD, range_only: bool = False,
               max_events: Optional[int] = None, query_every_vio: bool = True) -> ReplayResult:
    """Merge the two traces by timestamp and feed them to ``pf`` (test/test.c:93-105).

    ``range_only`` skips the VIO rows (BASELINE.json config 2: range-only updates).
    """
    res = ReplayResult()
    iv = iu = 0
    nv, nu = (0 if range_only else len(vio)), len(uwb)
    events = 0
    while iv < nv or iu < nu:
        if max_events is not None and events >= max_events:
            break
        events += 1
        if iv < nv and (iu >= nu or vio[iv][0] < uwb[iu][0]):
            t, x, y, z = vio[iv]
            iv += 1
            pf.depositVio(t, x, y, z, 0.0)
            res.n_vio += 1
            if query_every_vio:
                st = pf.getTagLoc()
                if st[0]:
                    res.rows.append(tuple(st[1:]))
        else:
            t, b, r = uwb[iu]
            iu += 1
            r = f32(r - f32(uwb_bias))
            if r > f32(0.0) and r < f32(30.0):
                pf.depositRange(deployment[b, 0], deployment[b, 1], deployment[b, 2], r, f32(uwb_std))
                res.n_range += 1
    return res


def load_test1(folder: str):
    """The three input files of the reference's golden test (test/test.c:29-31)."""
    return (read_vio(os.path.join(folder, VIO_FILE)),
            read_uwb(os.path.join(folder, UWB_FILE)),
            read_deployment(os.path.join(folder, DEPLOY_FILE)))


# ------------------------------------------------------------------ csvslam: verbose trace format, SLAM replay
# csvslam/csvslam.c:16-27
SLAM_NUM_BCNS = 12
SLAM_UWB_STD = f32(0.1)
SLAM_UWB_BIAS = f32(0.4)


def _skip_to_waypoint(rows, key):
    """csvslam.c:104-116 / :125-135: the FIRST read skips rows while the first character of the waypoint field is < '4'
    (a character comparison: waypoints 10..19 would count as '1'; the skip only runs once, at the start)."""
    k = 0
    while k < len(rows) and key(rows[k]) < "4":
        k += 1
    return rows[k:]


def read_vio_verbose(path: str, skip_to_waypoint: bool = True):
    """Line pairs ``t,position,waypoint,accuracy,y,z,x`` / ``t,orientation,...`` -> (t, x, y, z) (csvslam.c:97-119)."""
    rows = []
    with open(path, "r") as fh:
        lines = fh.read().split("\n")
    k = 0
    while k < len(lines) and lines[k]:
        c = lines[k].split(",")
        rows.append((float(c[0]), f32(float(c[6])), f32(float(c[4])), f32(float(c[5])), c[2][0]))
        k += 2                                   # the orientation line is skipped unread (csvslam.c:113)
    if skip_to_waypoint:
        rows = _skip_to_waypoint(rows, lambda r: r[4])
    return [r[:4] for r in rows]


def read_uwb_verbose(path: str, skip_to_waypoint: bool = True, num_bcns: int = SLAM_NUM_BCNS):
    """Rows ``t,uwb_range,waypoint,<beacon name>,range,...`` -> (t, beacon, range); the beacon is the first letter of
    its host name minus 'a' (csvslam.c:121-140)."""
    rows = []
    with open(path, "r") as fh:
        for line in fh:
            c = line.rstrip("\n").split(",")
            if len(c) < 5:
                continue
            b = ord(c[3][0]) - ord("a")
            if not 0 <= b < num_bcns:
                raise ValueError(f"beacon {c[3]!r} out of range in {path}")          # the reference asserts (csvslam.c:138)
            rows.append((float(c[0]), b, f32(float(c[4])), c[2][0]))
    if skip_to_waypoint:
        rows = _skip_to_waypoint(rows, lambda r: r[3])
    return [r[:3] for r in rows]


def format_bcn_row(b: int, x, y, z, theta) -> str:
    """csvslam.c:158: ``%hhu,%f,%f,%f,%f``"""
    return "%d,%f,%f,%f,%f\n" % (int(b), float(x), float(y), float(z), float(theta))


@dataclass
class SlamReplayResult:
    tag_rows: List[Tuple[float, float, float, float, float]] = field(default_factory=list)   # t,x,y,z,theta per VIO row
    bcn_rows: List[Tuple[int, float, float, float, float]] = field(default_factory=list)     # b,x,y,z,theta at the end
    n_vio: int = 0
    n_range: int = 0

    def tag_csv(self) -> str:
        # csvslam.c:142-151 prints t,x,y,z,theta in THIS order (unlike test.c, which un-permutes to the file's y,z,x)
        return "t,x,y,z,theta\n" + "".join("%f,%f,%f,%f,%f\n" % tuple(float(v) for v in r) for r in self.tag_rows)

    def bcn_csv(self) -> str:
        return "b,x,y,z,theta\n" + "".join(format_bcn_row(*r) for r in self.bcn_rows)


def replay_slam(pf, bcns: Sequence, vio: Sequence, uwb: Sequence, uwb_bias=SLAM_UWB_BIAS, uwb_std=SLAM_UWB_STD,
                max_events: Optional[int] = None) -> SlamReplayResult:
    """csvslam.c:62-85: merge the traces by timestamp; VIO rows go to depositTagVio followed by a getTagLoc, range rows
    (minus the bias, if in (0, 30) m) to depositRange on the named beacon with every beacon in allBcns; at the end one
    getBcnLoc per beacon.  ``pf`` / ``bcns`` need the interface of slam3d_amd.slam.ParticleFilterSlam / its beacons."""
    res = SlamReplayResult()
    iv = iu = 0
    nv, nu = len(vio), len(uwb)
    events = 0
    while iv < nv or iu < nu:
        if max_events is not None and events >= max_events:
            break
        events += 1
        if iv < nv and (iu >= nu or vio[iv][0] < uwb[iu][0]):
            t, x, y, z = vio[iv]
            iv += 1
            pf.depositTagVio(t, x, y, z, 0.0)
            res.n_vio += 1
            st = pf.getTagLoc()
            if st[0]:
                res.tag_rows.append(tuple(st[1:]))
        else:
            t, b, r = uwb[iu]
            iu += 1
            r = f32(r - f32(uwb_bias))
            if r > f32(0.0) and r < f32(30.0):
                pf.depositRange(bcns[b], r, f32(uwb_std))
                res.n_range += 1
    for b, bcn in enumerate(bcns):
        st = pf.getBcnLoc(bcn)
        if st[0]:
            res.bcn_rows.append((b,) + tuple(st[2:]))
    return res
