"""Pin the CPU oracle against the reference's own golden files.

test/test.c semantics on tests/golden/test1 (copies of the reference's test/data inputs and its four
expected_out CSVs): N = 10 000, seed 123456789, 876 VIO + 1006 UWB rows.
  - glibc rand_r + f32 sums  == expected_out_linux-gcc.csv, byte for byte (test.c:124-161 compare)
  - MUSL  rand_r + f32 sums  ~= expected_out_windows-{gcc,cl}.csv within 1.5e-6 (libm / printf last digit;
    the reference's own python test allows 5e-6, test.py:34)
"""
import os

import numpy as np
import pytest

from slam3d_amd import replay as R

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _expected(name):
    return os.path.join(GOLDEN, "test1", f"test1_ParticleFilterLoc_expected_out_{name}.csv")


@pytest.fixture(scope="module")
def glibc_run(ora, test1):
    vio, uwb, dep = test1
    pf = ora.OracleLoc(10000, rng_mode=ora.RNG_GLIBC, seed=123456789, acc=ora.ACC_F32)
    res = R.replay_loc(pf, vio, uwb, dep)
    return pf, res


def test_linux_gcc_golden_byte_exact(glibc_run):
    pf, res = glibc_run
    with open(_expected("linux-gcc"), "r") as fh:
        expected = fh.read()
    assert res.n_vio == 876 and res.n_range == 1006 and len(res.rows) == 875
    assert res.csv_text() == expected
    # whole-stream checksum of the rand_r sequence
    assert pf.rng.state == 1123155124


@pytest.mark.parametrize("name", ["windows-gcc", "windows-cl"])
def test_windows_goldens_musl(ora, test1, name):
    vio, uwb, dep = test1
    pf = ora.OracleLoc(10000, rng_mode=ora.RNG_MUSL, seed=123456789, acc=ora.ACC_F32)
    res = R.replay_loc(pf, vio, uwb, dep)
    got = res.as_array()[:, [0, 2, 3, 1, 4]]          # file column order: t, y, z, x, theta
    exp = np.loadtxt(_expected(name), delimiter=",", skiprows=1)
    assert got.shape == exp.shape == (875, 5)
    assert np.abs(got[:, 1:] - exp[:, 1:]).max() <= 1.5e-6
    assert np.abs(got[:, 0] - exp[:, 0]).max() <= 1e-6


def test_philox_known_answers(ora):
    # Random123 kat_vectors, philox4x32 10 rounds
    assert ora.philox4x32_10([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert ora.philox4x32_10([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert ora.philox4x32_10([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_f64_rule_stays_close_to_reference_arithmetic(ora, test1, glibc_run):
    """The f64-accumulation rule (what the HIP path implements) changes only the rounding of the sums
    and of the CDF.  That moves a few ancestor choices per resample (the reference's serial f32 CDF is
    itself only good to ~1e-3 relative at N = 10 000), after which the serial rand_r stream is dealt to
    different particles -- so the two runs are two draws of the same filter: identical before the first
    resample, then within the seed-to-seed spread of the estimate (a few cm while the cloud is wide)."""
    vio, uwb, dep = test1
    pf = ora.OracleLoc(10000, rng_mode=ora.RNG_GLIBC, seed=123456789, acc=ora.ACC_F64)
    res = R.replay_loc(pf, vio, uwb, dep, max_events=60)
    a = res.as_array()
    b = glibc_run[1].as_array()[: a.shape[0]]
    assert a.shape[0] > 20
    assert np.abs(a[0, 1:4] - b[0, 1:4]).max() < 1e-5
    assert np.abs(a[:, 1:4] - b[:, 1:4]).max() < 0.3


def test_resample_index_walk_matches_definition(ora):
    """anc[i] = first j with rStart + rStep*i < cdf[j], clamped to n-1 (pfResample.c:70-72)."""
    import _synth
    w = (0.05 + _synth.u01(5, 5000)).astype(np.float32)
    cdf = np.cumsum(w, dtype=np.float32)
    r_step = np.float32(cdf[-1] / np.float32(5000))
    r_start = np.float32(0.37) * r_step
    anc = ora.resample_indices_f32(cdf, r_start, r_step)
    u = (r_start + r_step * np.arange(5000, dtype=np.float32)).astype(np.float32)
    ref = np.minimum(np.searchsorted(cdf, u, side="right"), 4999)
    assert np.array_equal(anc, ref)
    # rStart == rStep (uniform() returned exactly 1.0): last offspring falls off the CDF -> clamp
    anc2 = ora.resample_indices_f32(cdf, r_step, r_step)
    assert anc2[-1] == 4999 and np.all(np.diff(anc2) >= 0)


def _exact_weights(n=8192):
    """weights whose running sums are exact in float32 AND in the 2^-20 fixed point of the f64 rule:
    512 x 4, 1024 x 1, 2048 x 1/2, the rest 0, shuffled.  s = 4096 = n/2 (so rStep = 1/2 exactly, no spawn:
    s/n >= 0.4 at a long range), ss = 9728, ESS/n = 0.21 (resample)."""
    w = np.zeros(n, dtype=np.float32)
    w[:512] = 4.0
    w[512:1536] = 1.0
    w[1536:3584] = 0.5
    np.random.default_rng(11).shuffle(w)
    return w


def test_f32_and_f64_rules_select_the_same_ancestors_on_exact_inputs(ora):
    """VERDICT r02 item 7a.  ORA_ACC_F32 (the reference's arithmetic, pfResample.c:52-72: serial float32 CDF, float32
    u_i) and ORA_ACC_F64 (what the HIP path implements: q(w) sums in f64) are the same function wherever neither
    rounds.  With the weights above both CDFs hold the same multiples of 1/2 exactly and rStep = 1/2 exactly."""
    n = 8192
    w = _exact_weights(n)
    cdf32 = np.cumsum(w, dtype=np.float32)
    cdf64 = np.cumsum(w.astype(np.float64))
    assert np.array_equal(cdf32.astype(np.float64), cdf64) and cdf64[-1] == n / 2
    r_step = 0.5
    # (1) rStart a power-of-two fraction of rStep: u_i = i/2 + 2^-(k+1) needs <= 24 bits up to k = 10 -> exact in float32
    for k in range(1, 11):
        r_start = r_step * 2.0 ** -k
        a32 = ora.resample_indices_f32(cdf32, r_start, r_step)
        a64 = ora.resample_indices_f64(cdf64, r_start, r_step)
        assert np.array_equal(a32, a64), f"k = {k}"
        u = r_start + r_step * np.arange(n)
        assert np.array_equal(a64, np.minimum(np.searchsorted(cdf64, u, side="right"), n - 1))
    # rStart = 0 and rStart = rStep: every u_i EQUAL to a CDF value -- "u < cdf[j]" is false at equality in both
    for r_start in (0.0, r_step):
        assert np.array_equal(ora.resample_indices_f32(cdf32, r_start, r_step), ora.resample_indices_f64(cdf64, r_start, r_step))
    # (2) the whole resample stage, Philox draw shared: uniform() * 1/2 is exact in both modes, so rStart is the same
    # number; float32 rounds u_i by at most 2^-12 (ulp at 4096 is 2^-11), which cannot carry it across a multiple of
    # 1/2 while rStart is further than that from 0 and from 1/2.
    done = 0
    for epoch in range(1, 9):
        pfs = []
        for acc in (ora.ACC_F32, ora.ACC_F64):
            pf = ora.OracleLoc(n, rng_mode=ora.RNG_PHILOX, seed=2024, acc=acc)
            z = np.zeros(n, dtype=np.float32)
            pf.set_particles(w, np.arange(n, dtype=np.float32), z, z, z)
            pf.set_initialized(1)
            pf.epoch = epoch
            pf.resample(0.0, 0.0, 0.0, 10.0, 0.1)
            pfs.append(pf)
        s32, s64 = pfs[0].last_stats(), pfs[1].last_stats()
        assert s32["branch"] == s64["branch"] == 1 and s32["numSpawn"] == s64["numSpawn"] == 0
        assert s32["rStep"] == s64["rStep"] == 0.5 and s32["rStart"] == s64["rStart"]
        assert s32["s"] == s64["S"] == n / 2
        if not (2.0 ** -10 < s64["rStart"] < 0.5 - 2.0 ** -10):
            continue
        done += 1
        assert np.array_equal(pfs[0].last_ancestors(), pfs[1].last_ancestors()), f"epoch {epoch}"
    assert done >= 6
