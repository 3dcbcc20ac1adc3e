"""Property tests of the library's arithmetic helpers on the device (atomdev_selftest_math)."""
import ctypes as C
import math
import os
import subprocess

import numpy as np
import pytest

import atom_amd
import harness as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def host_cr(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("cr") / "libcrhost.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-I", os.path.join(H.ROOT, "atom_amd", "csrc"),
                           os.path.join(H.ROOT, "tests", "cr_math_host.cpp"), "-o", so])
    lib = C.CDLL(so)
    dp = C.POINTER(C.c_double)
    lib.cr_host_eval.argtypes = [C.c_int, C.c_size_t, dp, dp, dp]
    lib.cr_host_eval.restype = None

    def run(kind, x, y=None):
        out = np.empty_like(x)
        lib.cr_host_eval(kind, x.size, x.ctypes.data_as(dp), y.ctypes.data_as(dp) if y is not None else None, out.ctypes.data_as(dp))
        return out
    return run


def _libm_differs(got, fn, *args, sample=200_000):
    """Share of the first `sample` results that differ from the C library's (math.* calls it; numpy has its own vector code)."""
    cols = [a[:sample].tolist() for a in args]
    ref = np.array([fn(*v) for v in zip(*cols)])
    return (got[:sample] != ref).mean()


def test_cr_math_same_bits_on_device_and_host(host_cr):
    """exp, pow, x^4 and the fourth root of cr_math.h: the device gives the bits the host gives from the same source
    (tests/test_abi.py holds the host's against long-double arithmetic), and glibc's functions differ from them in the
    last place in less than 0.2 % of the calls."""
    rng = np.random.default_rng(7)
    n = 2_000_000
    d = atom_amd.DeviceModel(H.ATM, 5, 7, 7)
    x = rng.uniform(-40., 40., n); x[::4] *= 17.
    dev = d.selftest_math(0, x)
    assert np.array_equal(dev, host_cr(0, x))
    assert _libm_differs(dev, math.exp, x) < 2e-3
    ys = np.array([1. / 3., 2. / 3., 7. / 9., 13. / 9., 1. / 6., 4. / 9., 5. / 26., 8. / 13., 1.5, 5.2558, 3.4161, -2.5, 0.286, 17.3])
    base = np.exp2(rng.uniform(-12., 6., n)); base[::3] = 1. + rng.uniform(-12., 6., base[::3].size) * 1e-3
    y = ys[rng.integers(0, ys.size, n)]; y[::5] = rng.uniform(-12., 6., y[::5].size)
    dev = d.selftest_math(1, base, y)
    assert np.array_equal(dev, host_cr(1, base, y))
    assert _libm_differs(dev, math.pow, base, y) < 2e-3
    t = rng.uniform(150., 330., n)
    dev = d.selftest_math(2, t)
    assert np.array_equal(dev, host_cr(2, t)) and _libm_differs(dev, math.pow, t, np.full(n, 4.)) < 2e-3
    r = rng.uniform(1e6, 2e10, n)
    dev = d.selftest_math(3, r)
    assert np.array_equal(dev, host_cr(3, r)) and _libm_differs(dev, math.pow, r, np.full(n, 0.25)) < 2e-3
    d.close()


def test_div_by_is_the_ieee_quotient():
    """The sweeps divide by grid constants through x * (1/y) and two fused corrections (div_by): on 10 000 000 operand
    pairs - dividends over the fields' magnitudes, divisors over the grid constants' range, divisors with every
    significand bit set and just above a power of two among them - the result has the bits of the hardware's IEEE
    division."""
    rng = np.random.default_rng(8)
    n = 10_000_000
    d = atom_amd.DeviceModel(H.ATM, 5, 7, 7)
    a = rng.uniform(-1., 1., n) * np.exp2(rng.integers(-40, 40, n))
    b = rng.uniform(1e-6, 1e6, n)
    ones = np.nextafter(np.exp2(rng.integers(-20, 20, n // 10)), 0.)       # 1.111...1 x 2^k
    b[:ones.size] = ones
    b[ones.size:2 * ones.size] = np.nextafter(np.exp2(rng.integers(-20, 20, ones.size)), np.inf)
    b[2 * ones.size:3 * ones.size] = np.array([0.025, 0.0174532925199432958, 0.00001, 1000., 717.9, 610., 960.])[rng.integers(0, 7, ones.size)]
    a[::7] = rng.integers(-1000, 1000, a[::7].size).astype(np.float64)     # exact quotients among them
    got = d.selftest_math(4, a, b)
    ref = d.selftest_math(5, a, b)
    assert np.array_equal(ref, a / b)
    bad = np.nonzero(got != ref)[0]
    assert bad.size == 0, (bad.size, a[bad[:5]], b[bad[:5]])
    d.close()


def test_a_failure_a_kernel_reports_surfaces_with_the_next_result():
    """The hydrosphere sweep's bounded wait between workgroups reports a give-up through a word in page-locked host memory:
    whichever call hands the next result to the host fails, once."""
    for take in ("download", "residuum", "get_columns", "sync"):
        d = atom_amd.DeviceModel(H.HYD, 9, 21, 31)
        d.op("hyd_bc_theta")
        d._check(d.lib.atomdev_selftest_raise_kernel_error(d._h))
        with pytest.raises(atom_amd.AtomDevError, match="gave up waiting"):
            if take == "download":
                d.download("T")
            elif take == "residuum":
                d.residuum("hyd_residuum3d")
            elif take == "get_columns":
                d.get_columns("T", 0, 4)
            else:
                d.sync()
        d.download("T"); d.sync()      # reported once
        d.close()


def test_deferred_files_for_one_path_are_written_in_order(tmp_path):
    """Inside a one-call slice the text files are finished by threads of their own.  A checkpoint and the final write can
    fall on the same file name (the reference's defaults: 2 x 2 iterations, checkpoint 2), and the reference writes one
    after the other, so the later content must stand: two jobs for one path, the first held back, leave the second's text."""
    d = atom_amd.DeviceModel(H.ATM, 5, 7, 9)
    path = str(tmp_path / "same_name.vtk")
    d._check(d.lib.atomdev_selftest_deferred_file_order(d._h, path.encode()))
    assert open(path).read() == "second\n"
    d.close()
