"""Drop-in proof: the reference's OWN golden test program (test/test.c, compiled unchanged in the build container by
oracle/Makefile into oracle/_ref/ref_test_hip, linked against libparticlefilter.so instead of the reference's
sources) runs on the GPU box on the reference's test data.

Its byte-compare against the per-libc golden cannot pass (Philox noise, not the rand_r stream), so it is run with
--nofail and its output file is checked (a) byte for byte against the same replay driven through the Python mirror with
the same seed -- two independent callers of the same C-ABI -- and (b) against the linux-gcc golden within the
reference's own seed-to-seed spread (thresholds as in test_gpu_loc.py::test_test1_replay_statistically_equivalent)."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "oracle", "_ref", "ref_test_hip")
DATA = os.path.join(ROOT, "tests", "golden", "test1")


def test_reference_test_program_on_hip_library(tmp_path, test1):
    if not os.path.exists(EXE):
        pytest.skip("oracle/_ref/ref_test_hip was not built (needs /root/reference at build time)")
    out = tmp_path / "out.csv"
    p = subprocess.run([EXE, "--nofail", DATA + "/", str(out)], capture_output=True, text=True, timeout=110)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "Finished localization" in p.stdout
    text = out.read_text()
    got = np.loadtxt(str(out), delimiter=",", skiprows=1)
    assert got.shape == (875, 5)

    # (a) the same trace through the Python mirror of the wrapper: first filter of a fresh process -> same Philox key
    from slam3d_amd import particlefilter as pfm, replay as R
    vio, uwb, dep = test1
    g = pfm.DeviceParticleFilterLoc(10000, seed=123456789, stream_id=0)
    assert R.replay_loc(g, vio, uwb, dep).csv_text() == text

    # (b) against the reference's golden, statistically
    exp = np.loadtxt(os.path.join(DATA, "test1_ParticleFilterLoc_expected_out_linux-gcc.csv"), delimiter=",", skiprows=1)
    assert np.abs(got[:, 0] - exp[:, 0]).max() < 1e-6
    d = np.linalg.norm(got[100:, 1:4] - exp[100:, 1:4], axis=1)
    assert np.sqrt((d ** 2).mean()) < 0.30 and d.max() < 4.4
    assert np.linalg.norm(got[-1, 1:4] - exp[-1, 1:4]) < 0.10


CYDIR = os.path.join(ROOT, "oracle", "_ref", "cy")

_CY_DRIVER = r"""
import os, sys
sys.path.insert(0, {cydir!r}); sys.path.insert(0, {root!r})
import particlefilter as m                    # the reference's Cython module (particlefilter_wrapper.pyx, unchanged)
assert m.__file__.startswith({cydir!r}), m.__file__
from slam3d_amd import replay as R
vio, uwb, dep = R.load_test1(os.path.join({root!r}, "tests", "golden", "test1"))
m.setSeed(123456789)                          # test/test.py:91-93
pf = m.ParticleFilterLoc()
sys.stdout.write(R.replay_loc(pf, vio, uwb, dep).csv_text())
"""


def test_reference_cython_wrapper_on_hip_library(test1):
    """The reference's own Cython wrapper (particlefilter/cython/particlefilter_wrapper.pyx, unchanged, built by
    oracle/build_ref_cython.py with the one Extension(...) call of INTEGRATION.md section 2: libparticlefilter.so in place
    of particlefilter/src/*.c) drives the HIP library on the GPU through test/test.py's sequence (:91-121: setSeed,
    ParticleFilterLoc(), the merged trace with a getTagLoc per VIO row).  Its struct lives by value inside a Python object
    (particlefilter_wrapper.pyx:28) -- the third kind of storage next to static (test.c) and heap (JNI).  Output ==
    the same replay through this repository's ctypes mirror, byte for byte (same seed, first filter of a fresh process)."""
    import glob
    import sys
    if not glob.glob(os.path.join(CYDIR, "particlefilter*.so")):
        pytest.skip("oracle/_ref/cy was not built (needs /root/reference and Cython at build time)")
    code = _CY_DRIVER.format(cydir=CYDIR, root=ROOT)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=110)
    assert p.returncode == 0, p.stdout[-1000:] + p.stderr[-3000:]
    text = p.stdout
    assert text.count("\n") == 876                  # header + 875 rows
    from slam3d_amd import particlefilter as pfm, replay as R
    vio, uwb, dep = test1
    g = pfm.DeviceParticleFilterLoc(10000, seed=123456789, stream_id=0)
    assert R.replay_loc(g, vio, uwb, dep).csv_text() == text


def _lay_out(tmp_path, files):
    """a scratch tree around the reference drivers' relative path macros; returns the working directory"""
    import shutil
    cwd = tmp_path / "run"
    cwd.mkdir()
    for rel, src in files.items():
        dst = (cwd / rel).resolve()
        dst.parent.mkdir(parents=True, exist_ok=True)
        shutil.copy(src, dst)
    return cwd


def test_reference_csvlocalize_on_hip_library(tmp_path, test1):
    """csvlocalize/csvlocalize.c, unchanged, linked against the library (oracle/_ref/ref_csvlocalize_hip) and run on the
    GPU box.  Its inputs are macros relative to the working directory (csvlocalize.c:29-39: ../mqttlogger/{vio,uwb}.csv,
    ../sampledata/arena/deploy.csv -- byte-identical to test/data).  It never sets a seed (the
    reference then seeds from the clock, pfRandom.c:56-60); PF_HIP_SEED stands in for the clock, so its tag.csv must equal
    the same replay through the ctypes mirror byte for byte, and sit within the reference's own seed-to-seed spread of the
    linux-gcc golden, as test_reference_test_program_on_hip_library."""
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_csvlocalize_hip")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/ref_csvlocalize_hip was not built (needs /root/reference at build time)")
    cwd = _lay_out(tmp_path, {"../mqttlogger/vio.csv": DATA + "/test1_ParticleFilterLoc_vio.csv",
                              "../mqttlogger/uwb.csv": DATA + "/test1_ParticleFilterLoc_uwb.csv",
                              "../sampledata/arena/deploy.csv": DATA + "/test1_ParticleFilterLoc_deploy.csv"})
    p = subprocess.run([exe], cwd=cwd, capture_output=True, text=True, timeout=110, env=dict(os.environ, PF_HIP_SEED="123456789"))
    assert p.returncode == 0, p.stdout + p.stderr
    text = (tmp_path / "mqttlogger" / "tag.csv").read_text()
    from slam3d_amd import particlefilter as pfm, replay as R
    vio, uwb, dep = test1
    g = pfm.DeviceParticleFilterLoc(10000, seed=123456789, stream_id=0)
    assert R.replay_loc(g, vio, uwb, dep).csv_text() == text
    got = np.loadtxt(str(tmp_path / "mqttlogger" / "tag.csv"), delimiter=",", skiprows=1)
    exp = np.loadtxt(os.path.join(DATA, "test1_ParticleFilterLoc_expected_out_linux-gcc.csv"), delimiter=",", skiprows=1)
    assert got.shape == exp.shape == (875, 5)
    assert np.abs(got[:, 0] - exp[:, 0]).max() < 1e-6
    d = np.linalg.norm(got[100:, 1:4] - exp[100:, 1:4], axis=1)
    assert np.sqrt((d ** 2).mean()) < 0.30 and d.max() < 4.4
    assert np.linalg.norm(got[-1, 1:4] - exp[-1, 1:4]) < 0.10


def test_reference_csvslam_on_hip_library(tmp_path):
    """csvslam/csvslam.c, unchanged, linked against the library (oracle/_ref/ref_csvslam_hip) and run on the GPU box on the
    trimmed sampledata/cic/0 trace (tests/golden/cic0; the driver reads ../sampledata/cic/0/{vio,uwb}.csv, csvslam.c:16-26,
    and never sets a seed: PF_HIP_SEED).  tag.csv / bcn.csv equal, byte for byte, the same replay through the Python mirror
    of the SLAM interface, and sit within the bars of test_gpu_slam.py::test_cic0_replay_drop_in of what the compiled
    reference wrote (tests/golden/ref_cic0_slam.json)."""
    import json
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_csvslam_hip")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/ref_csvslam_hip was not built (needs /root/reference at build time)")
    with open(os.path.join(ROOT, "tests", "golden", "ref_cic0_slam.json")) as fh:
        g = json.load(fh)["glibc"]
    cic = os.path.join(ROOT, "tests", "golden", "cic0")
    cwd = _lay_out(tmp_path, {"../sampledata/cic/0/vio.csv": cic + "/vio.csv", "../sampledata/cic/0/uwb.csv": cic + "/uwb.csv"})
    p = subprocess.run([exe], cwd=cwd, capture_output=True, text=True, timeout=110, env=dict(os.environ, PF_HIP_SEED=str(g["seed"])))
    assert p.returncode == 0 and "Done" in p.stdout, p.stdout + p.stderr
    out = tmp_path / "sampledata" / "cic" / "0"
    from slam3d_amd import particlefilter as pfm, replay as R, slam
    pfm.setSeed(g["seed"])
    pf = slam.ParticleFilterSlam()
    bcns = [pf.addBcn() for _ in range(R.SLAM_NUM_BCNS)]
    res = R.replay_slam(pf, bcns, R.read_vio_verbose(cic + "/vio.csv"), R.read_uwb_verbose(cic + "/uwb.csv"))
    assert res.tag_csv() == (out / "tag.csv").read_text()
    assert res.bcn_csv() == (out / "bcn.csv").read_text()
    t, gt = np.loadtxt(str(out / "tag.csv"), delimiter=",", skiprows=1), np.array(g["tag_rows"])
    assert t.shape == gt.shape and np.abs(t[:, 0] - gt[:, 0]).max() < 1e-6
    assert np.linalg.norm(t[:, 1:4] - gt[:, 1:4], axis=1).max() < 0.0025 + 2e-6        # (+ the files' six decimals)
    assert np.abs(t[:, 4] - gt[:, 4]).max() < 4e-6
    b, gb = np.loadtxt(str(out / "bcn.csv"), delimiter=",", skiprows=1), np.array(g["bcn_rows"])
    assert b.shape == gb.shape and np.array_equal(b[:, 0], gb[:, 0])        # the same ten beacons were heard
    db = np.abs(b[:, 1:4] - gb[:, 1:4]).max(axis=0)
    assert db[0] < 0.26 and db[1] < 0.88 and db[2] < 0.39, db
