"""cAtmosphereModel::RunTimeSlice (atmosphere/cAtmosphereModel.cpp:140-341) as a whole, on the 1-degree grid with the
reference's own input files (tests/golden/atm_slice_inputs.npz writes them back), the plate reconstruction off.

* the oracle's operators composed in the reference's order against the compiled model object's own RunTimeSlice: every
  field bit for bit, every output file byte for byte (CPU, where the reference is);
* the oracle against the golden record of three time slices (0, 40 and 140 Ma) at the reference's iteration counts
  (tests/golden/atm_time_slice.npz, made by oracle/make_golden.py from the compiled reference);
* the device's atomdev_atm_run_time_slice against the oracle and against that record, and the hydrosphere's time slice
  of the same Ma behind it (atmosphere -> transfer file -> hydrosphere), device against oracle.

Bars.  The slice runs pow/exp in the sweep and in the column physics, and the saturation adjustment iterates to a 1e-5
exit test: last-place differences of libm grow with every call (tests/test_loops.py measures it).  The device is held to
DEVICE_TOL per field (max |a-b| / max |b|) after the reference's full iteration set; the error of every field is printed.
"""
import ctypes as C
import hashlib
import os

import numpy as np
import pytest

import atom_amd
import harness as H

DIMS = (41, 181, 361)
DEFAULT_COUNTS = (2, 10, 2, 2)      # param.py:33-36
DEVICE_TOL = 1e-10      # measured: 1.7e-11 (S_c at 0 Ma), 8.7e-12 (S_c at 140 Ma), 1.7e-12 (c at 40 Ma)


def _oracle_slice(tmp_path, Ma, prev, counts, curve):
    p = H.default_params(H.ATM)
    f, g = H.atm_reset_arrays(DIMS, p)
    o = H.CpuModel("oracle", H.ATM, *DIMS, p, f, g)
    s, i_trop = H.atm_slice_start(Ma, prev, curve)
    out = str(tmp_path / ("oracle_%d" % Ma))
    os.makedirs(out)
    H.run_oracle_atm_slice(o, s, i_trop, counts, str(tmp_path), out, curve)
    return p, o, s, out


def test_slice_inputs_are_written_back(tmp_path):
    temp, prec, curve = H.write_atm_slice_inputs(str(tmp_path))
    t = np.loadtxt(temp)
    assert t.shape == (181 * 361, 3) and t[0].tolist() == [0., 90., -14.89]
    assert sorted(curve) == [0, 30, 40, 130, 140]
    for Ma, _ in H.ATM_SLICE_CASES:
        assert np.loadtxt(str(tmp_path / ("%dMa_Golonka.xyz" % Ma))).shape == (181 * 361, 3)


@pytest.mark.skipif(not H.have_ref(), reason="oracle/_ref/libatomref.so not built")
def test_atm_time_slice_oracle_vs_reference(tmp_path):
    """A paleo slice (40 Ma after 30 Ma: the coast correction, the paleo increments) with 2 x 2 surface and 1 x 2 volume
    iterations (the saturation adjustment and the ice scheme run once, the checkpoint files are written)."""
    counts = (2, 2, 2, 1)
    _, _, curve = H.write_atm_slice_inputs(str(tmp_path))
    p, o, s, out_o = _oracle_slice(tmp_path, 40, 30, counts, curve)
    f, g = H.atm_reset_arrays(DIMS, p)
    r = H.CpuModel("ref", H.ATM, *DIMS, p, f, g)
    out_r = str(tmp_path / "ref_40")
    os.makedirs(out_r)
    H.run_ref_atm_slice(r, s, counts, str(tmp_path), out_r)
    for k in list(o.f) + list(o.g):
        if k == "RESIDUUM":
            continue
        a, b = (o.f[k], r.f[k]) if k in o.f else (o.g[k], r.g[k])
        assert H.bits_equal(a, b), (k, H.max_rel_err(a, b))
    for k in ("T", "U", "V", "W", "C", "CLOUD", "ICE", "CO2", "P_DYN", "P_RAIN", "S_C_C", "Q_LATENT", "RADIATION_3D"):
        assert not np.array_equal(o.f[k], f[k]), k
    names = sorted(os.listdir(out_r))
    assert len(names) == 5 and names == sorted(os.listdir(out_o))     # 1 pressure iteration: no checkpoint at 2
    for fn in names:
        assert open(os.path.join(out_r, fn), "rb").read() == open(os.path.join(out_o, fn), "rb").read(), fn


def _check_against_record(m, out, tag, z):
    rec = H.atm_slice_record(m, out, tag)
    keys = [k for k in z.files if k.startswith(tag + "/")]
    assert sorted(keys) == sorted(rec)
    for k in keys:
        if k.endswith("/sha256") or "/file/" in k:
            assert str(rec[k]) == str(z[k]), k


def test_atm_time_slice_oracle_golden(tmp_path):
    """The oracle's slice at 0 Ma with the reference's iteration counts: SHA-256 of every field and of the six output files
    as the compiled reference left them."""
    _, _, curve = H.write_atm_slice_inputs(str(tmp_path))
    p, o, s, out = _oracle_slice(tmp_path, 0, None, DEFAULT_COUNTS, curve)
    _check_against_record(o, out, "Ma0", np.load(H.ATM_SLICE_GOLDEN))


def _coupled_hydrosphere(tmp_path, Ma, o_atm, transfer_file):
    """The hydrosphere's time slice of the same Ma behind the atmosphere's (cHydrosphereModel.cpp:234-610, the reference's
    iteration counts): the device reads the transfer file the device atmosphere wrote, the oracle takes the oracle
    atmosphere's surface through the same text round trip; the same Golonka file as sea ground, the surface salinity
    file at 0 Ma."""
    dims = DIMS
    p = H.default_params(H.HYD)
    f, g = H.hyd_reset_arrays(dims, p)
    oh = H.CpuModel("oracle", H.HYD, *dims, p, f, g)
    dh = atom_amd.DeviceModel(H.HYD, *dims, atom_amd.Params.from_buffer_copy(bytes(p)))
    dh.upload_all(f, g)
    bath = str(tmp_path / ("%dMa_Golonka.xyz" % Ma))
    salt = str(tmp_path / "SurfaceSalinity_NASA.xyz")
    oh.lib.oracle_transfer_atm_to_hyd(o_atm.ctx, oh.ctx)
    st = H.hyd_start(dims[0], 0, Ma)
    H.hyd_start_of_slice(oh, st, bath, salt)
    H.hyd_2d_loop(oh, 10, 2)
    H.hyd_3d_loop(oh, 2, 2, run_data=True)
    dh.run_time_slice(bath, salt if Ma == 0 else None, transfer_file, Ma=Ma)
    errs = {}
    for k in list(f) + list(g):
        if k == "RESIDUUM":
            continue
        ref = oh.f[k] if k in f else oh.g[k]
        assert np.isfinite(ref).all(), k
        errs[k] = H.max_rel_err(dh.download(k), ref)
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:4]
    print("  coupled hydrosphere slice, device against oracle, the four largest: " + ", ".join("%s %.2e" % kv for kv in worst))
    for k, e in errs.items():
        assert e <= DEVICE_TOL, (k, e)
    for k in ("T", "U", "V", "W", "C", "P_DYN", "UPWELLING"):
        assert not np.array_equal(oh.f[k] if k in f else oh.g[k], f[k] if k in f else g[k]), k
    dh.close()


def _same_up_to_last_digit(path_a, path_b):
    """Two text files with the same tokens, numbers allowed to differ by one unit of their last printed digit.  Returns
    the number of tokens that differ at all."""
    ta, tb = open(path_a).read().split(), open(path_b).read().split()
    assert len(ta) == len(tb), (path_a, len(ta), len(tb))
    differing = 0
    for x, y in zip(ta, tb):
        if x == y:
            continue
        differing += 1
        fx, fy = float(x), float(y)      # raises on a differing word
        mant = x.lower().split("e")[0]
        decimals = len(mant.split(".")[1]) if "." in mant else 0
        expo = int(x.lower().split("e")[1]) if "e" in x.lower() else 0
        assert abs(fx - fy) <= 1.000001 * 10.0 ** (expo - decimals), (path_a, x, y)
    return differing


@pytest.mark.gpu
@pytest.mark.parametrize("Ma,prev", H.ATM_SLICE_CASES)
def test_atm_time_slice_device(Ma, prev, tmp_path):
    """atomdev_atm_run_time_slice with the reference's iteration counts (2 x 10 surface, 2 x 2 volume iterations, the
    saturation adjustment and the ice scheme twice each, the checkpoint files) against the oracle's slice, which is
    checked against the golden record here as well - with the sweep in its exact form (variant 0) and in its tolerance
    form (variant 2, the atmosphere's default).  Output files: the reference's bytes with variant 0; with variant 2 the
    same tokens up to one unit of a number's last printed digit."""
    _, _, curve = H.write_atm_slice_inputs(str(tmp_path))
    z = np.load(H.ATM_SLICE_GOLDEN)
    tag = "Ma%d" % Ma
    p, o, s, out_o = _oracle_slice(tmp_path, Ma, prev, DEFAULT_COUNTS, curve)
    _check_against_record(o, out_o, tag, z)
    name = "%dMa_Golonka.xyz" % Ma
    for variant in (0, 2):
        d = atom_amd.DeviceModel(H.ATM, *DIMS, atom_amd.Params.from_buffer_copy(bytes(p)))
        if variant == 2:
            assert d.rk_variant() == 2 or "ATOMDEV_RK_VARIANT" in os.environ      # the default of an atmosphere handle
        d.set_rk_variant(variant)
        out_d = str(tmp_path / ("device_%d_v%d" % (Ma, variant)))
        os.makedirs(out_d)
        sl = d.atm_slice(str(tmp_path / name), str(tmp_path / "SurfaceTemperature_NASA.xyz"), str(tmp_path / "SurfacePrecipitation_NASA.xyz"),
                         out_d, name, coast_correction=1 if Ma > 0 else 0, mean_temperature=float(np.float32(curve[Ma])))
        for fld, _ in atom_amd.AtmStart._fields_[:-1]:
            setattr(sl.start, fld, getattr(s, fld))
        state = d.atm_run_time_slice(sl)
        errs = {}
        for k in list(o.f) + list(o.g):
            if k == "RESIDUUM":
                continue
            ref = o.f[k] if k in o.f else o.g[k]
            assert np.isfinite(ref).all(), k
            errs[k] = H.max_rel_err(d.download(k), ref)
        print("\n%d Ma, sweep variant %d, device against oracle, max |a-b| / max |b| per field:" % (Ma, variant))
        for k, e in sorted(errs.items(), key=lambda kv: -kv[1]):
            print("  %-20s %.3e" % (k, e))
        for k, e in errs.items():
            assert e <= DEVICE_TOL, (variant, k, e)
        # samples of the compiled reference's own fields
        for k in errs:
            a = d.download(k)
            sample = a[::4, 62, ::8] if a.ndim == 3 else a[62, ::8]
            scale = max(np.abs(o.f[k] if k in o.f else o.g[k]).max(), 1e-300)
            assert np.abs(sample - z["%s/%s/sample" % (tag, k)]).max() / scale <= DEVICE_TOL, (variant, k)
        # output files: the reference's names; its bytes (variant 0) or its numbers up to the last printed digit (variant 2)
        names = sorted(os.listdir(out_o))
        assert names == sorted(os.listdir(out_d)) and len(names) == 6
        for fn in names:
            if variant == 0:
                digest = hashlib.sha256(open(os.path.join(out_d, fn), "rb").read()).hexdigest()
                assert digest == str(z["%s/file/%s" % (tag, fn)]), fn
            else:
                n = _same_up_to_last_digit(os.path.join(out_d, fn), os.path.join(out_o, fn))
                if n:
                    print("  %s: %d number(s) differ in their last printed digit" % (fn, n))
        print("  output files: %s; carried members: q_v_hyp %.6e S_nuc %.6e S_i_dep %.6e"
              % ("the reference's bytes" if variant == 0 else "the reference's numbers", state.q_v_hyp, state.S_nuc, state.S_i_dep))
        if variant == 0:
            _coupled_hydrosphere(tmp_path, Ma, o, os.path.join(out_d, "[%s]_Transfer_Atm.vw" % name))
        d.close()
