"""The native replay driver (slam3d_amd/bin/pfreplay: csvlocalize / csvslam / test.c with run-time arguments) on the
GPU box: its output files equal, byte for byte, what the Python mirror's replay writes for the same seed -- two
independent hosts (C++ over the C-ABI, Python over ctypes) driving the same kernels."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "slam3d_amd", "bin", "pfreplay")
T1 = os.path.join(ROOT, "tests", "golden", "test1")
CIC = os.path.join(ROOT, "tests", "golden", "cic0")


def _run(*args):
    p = subprocess.run([EXE, *args], capture_output=True, text=True, timeout=110)
    assert p.returncode == 0, p.stdout + p.stderr
    return p.stdout


@pytest.mark.parametrize("n", [10000, 250000])
def test_pfreplay_loc_equals_python_replay(tmp_path, test1, n):
    from slam3d_amd import particlefilter as pfm, replay as R
    out = tmp_path / "tag.csv"
    log = _run("loc", "--vio", os.path.join(T1, R.VIO_FILE), "--uwb", os.path.join(T1, R.UWB_FILE), "--deploy",
               os.path.join(T1, R.DEPLOY_FILE), "--out", str(out), "--n", str(n), "--seed", "4242")
    assert "876 VIO rows" in log and "875 output rows" in log
    vio, uwb, dep = test1
    g = pfm.DeviceParticleFilterLoc(n, seed=4242, stream_id=0)
    assert R.replay_loc(g, vio, uwb, dep).csv_text() == out.read_text()


def test_pfreplay_loc_range_only(tmp_path):
    from slam3d_amd import replay as R
    out = tmp_path / "tag.csv"
    log = _run("loc", "--vio", os.path.join(T1, R.VIO_FILE), "--uwb", os.path.join(T1, R.UWB_FILE), "--deploy",
               os.path.join(T1, R.DEPLOY_FILE), "--out", str(out), "--n", "1048576", "--range-only")
    assert "0 VIO rows" in log and "final pose: x=-0.6" in log       # config 2: see test_config2_range_only_1m


def test_pfreplay_slam_equals_python_replay(tmp_path, ):
    from slam3d_amd import particlefilter as pfm, replay as R, slam
    tag, bcn = tmp_path / "tag.csv", tmp_path / "bcn.csv"
    log = _run("slam", "--vio", os.path.join(CIC, "vio.csv"), "--uwb", os.path.join(CIC, "uwb.csv"), "--out-tag", str(tag),
               "--out-bcn", str(bcn), "--seed", "20180107")
    assert "591 VIO rows, 110 ranges" in log
    vio, uwb = R.read_vio_verbose(os.path.join(CIC, "vio.csv")), R.read_uwb_verbose(os.path.join(CIC, "uwb.csv"))
    pfm.setSeed(20180107)
    pf = slam.ParticleFilterSlam()
    bcns = [pf.addBcn() for _ in range(R.SLAM_NUM_BCNS)]
    res = R.replay_slam(pf, bcns, vio, uwb)
    assert res.tag_csv() == tag.read_text()
    assert res.bcn_csv() == bcn.read_text()


def test_pfreplay_reports_errors():
    p = subprocess.run([EXE, "loc", "--vio", "/nonexistent", "--uwb", "x", "--deploy", "y", "--out", "/tmp/z"], capture_output=True, text=True)
    assert p.returncode == 2 and "cannot open" in p.stderr
