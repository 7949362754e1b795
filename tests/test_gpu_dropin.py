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
