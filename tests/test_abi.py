"""CPU-side checks of the drop-in boundary: the library loads and exports every
symbol include/atomdev.h declares; argument errors come back as codes + text."""
import ctypes as C
import os
import re

import pytest

import atom_amd
import harness as H


def _declared():
    src = open(os.path.join(H.ROOT, "include", "atomdev.h")).read()
    return sorted(set(re.findall(r"\b(atomdev_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported():
    lib = atom_amd.load_library()
    names = _declared()
    assert len(names) >= 35
    for n in names:
        assert hasattr(lib, n), n


def test_default_params_match_reference_defaults():
    p = atom_amd.default_params(atom_amd.ATM)
    assert (p.re, p.pr, p.sc_WaterVapour, p.sc_CO2, p.L_atm, p.u_0) == (1000., .7179, .61, .96, 16000., 8.)
    assert p.dr == 0.025 and p.dt == 0.00001
    q = atom_amd.default_params(atom_amd.HYD)
    assert (q.re, q.pr, q.sc, q.L_hyd, q.u_0, q.c_0) == (10., 6.957, 1.7329, 1000., .25, 34.6)
    o = H.default_params(H.ATM)
    for name, _ in atom_amd.Params._fields_:
        assert getattr(p, name) == getattr(o, name), name


def test_default_atm_start_matches_reference_defaults():
    lib = atom_amd.load_library()
    s = atom_amd.AtmStart()
    lib.atomdev_default_atm_start(C.byref(s))
    ref, keep = H.atm_start(181)
    for name, _ in atom_amd.AtmStart._fields_[:-1]:
        assert getattr(s, name) == getattr(ref, name), name
    assert not s.im_tropopause
    assert lib.atomdev_atm_co2_coeff(1., .98, 330., 320., 280.) == .98 * ((330. / 280.) / (320. / 280.))
    assert lib.atomdev_atm_co2_coeff(0., .98, 330., 320., 280.) == 1.


def test_bad_arguments_are_reported_not_crashed():
    lib = atom_amd.load_library()
    h = C.c_void_p()
    p = atom_amd.default_params(atom_amd.ATM)
    assert lib.atomdev_create(7, 41, 181, 361, C.byref(p), -1, C.byref(h)) != 0
    assert b"unknown model" in lib.atomdev_last_error()
    assert lib.atomdev_create(0, 3, 181, 361, C.byref(p), -1, C.byref(h)) != 0
    assert b"too small" in lib.atomdev_last_error()
    assert lib.atomdev_sync(None) != 0
    assert lib.atomdev_atm_rk3d(None) != 0


def test_no_cpu_fallback_in_product():
    """The product never imports or links the oracle."""
    for root, _, files in os.walk(os.path.join(H.ROOT, "atom_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", "Makefile")):
                txt = open(os.path.join(root, f)).read()
                assert "oracle" not in txt.lower() or f == "__init__.py" and False, (root, f)


def test_text_numbers_match_stream_extraction(tmp_path):
    """The readers' direct decimal conversion (atom_amd/csrc/text_numbers.h) gives the bits of `stream >> double` on
    300 000 tokens in ten formats plus the edge cases, and refuses every token a stream would treat differently."""
    import subprocess
    exe = str(tmp_path / "text_numbers_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-I", os.path.join(H.ROOT, "atom_amd", "csrc"),
                           os.path.join(H.ROOT, "tests", "text_numbers_check.cpp"), "-o", exe])
    r = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and " 0 mismatches" in r.stdout, r.stdout + r.stderr


def test_fixed4_is_printf(tmp_path):
    """The writers' four-decimal formatter (atom_amd/csrc/fixed4.h) gives the bytes of printf("%.4f") on 5 000 000 values:
    random ones over sixty binades, dyadic values (the only ones that can sit on a half of the last place) and their
    neighbours, (k + 1/2) / 10^4 and its neighbours, zeros of both signs, huge values, infinities and NaNs."""
    import subprocess
    exe = str(tmp_path / "fixed4_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(H.ROOT, "atom_amd", "csrc"),
                           os.path.join(H.ROOT, "tests", "fixed4_check.cpp"), "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and " 0 mismatches" in r.stdout, r.stdout + r.stderr


def test_cr_math_is_correctly_rounded(tmp_path):
    """The double-double exp, pow, x^4 and fourth root of atom_amd/csrc/cr_math.h on 4 000 000 arguments each in the ranges
    the column physics uses: never on the wrong side of a case the x87 long-double functions decide; glibc's own double
    functions return the other neighbour in less than 0.2 % of the calls."""
    import subprocess
    exe = str(tmp_path / "cr_math_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(H.ROOT, "atom_amd", "csrc"),
                           os.path.join(H.ROOT, "tests", "cr_math_check.cpp"), "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = r.stdout.strip().split("\n")
    assert len(lines) == 3 and all(" wrong in 0 decidable cases" in ln for ln in lines), r.stdout
    for ln in lines:
        assert float(ln.rsplit("(", 1)[1].split("%")[0]) < 0.2, ln
