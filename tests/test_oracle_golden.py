"""The CPU restatement (oracle/atom_oracle.c) against the committed golden
vectors, which oracle/make_golden.py produced with the compiled reference.
Bar: bit-exact (every op here is +,-,*,/ and one glibc pow on the same host libm)."""
import hashlib
import os

import numpy as np
import pytest

import harness as H

GOLD = os.path.join(H.ROOT, "tests", "golden")


def _digest(fields, names):
    h = hashlib.sha256()
    for k in names:
        h.update(np.ascontiguousarray(fields[k]).tobytes())
    return h.hexdigest()


def _small(model, fname, seq, level=None):
    z = np.load(os.path.join(GOLD, fname))
    im, jm, km = (int(x) for x in z["dims"])
    p = H.default_params(model, im, jm, km)
    f, g = H.make_inputs(model, im, jm, km, int(z["seed"]), "blobs", p)
    names = H.ATM_FIELDS if model == H.ATM else H.HYD_FIELDS
    f = {k: f[k] for k in names}
    assert _digest(f, names) + _digest(g, H.FIELDS2D) == str(z["input_sha256"]), "input generator drifted"
    m = H.CpuModel("oracle", model, im, jm, km, p, f, g)
    lvl = None if level is None else im - 1
    snaps = H.run_sequence(m, seq, names, lvl)
    keys = [k for k in z.files if "/" in k]
    assert sorted(keys) == sorted(snaps.keys())
    for k in keys:
        assert H.bits_equal(snaps[k], z[k]), k
    return m, z


def test_atm_small_sequence_bitexact():
    _small(H.ATM, "atm_small.npz", H.ATM_SEQ)


def test_hyd_small_sequence_bitexact():
    _small(H.HYD, "hyd_small.npz", H.HYD_SEQ)


def test_hyd_surface_sequence_bitexact():
    m, z = _small(H.HYD, "hyd_surface.npz", H.HYD_SURFACE_SEQ, level="top")
    # Pressure_Hyd.cpp:304 zeroes p_dyn[0][jm-1][:] from the *surface* routine
    assert H.bits_equal(m.f["P_DYN"][0, m.jm - 1, :], z["p_dyn_level0_lastrow"])


def test_residuum_small():
    z = np.load(os.path.join(GOLD, "residuum_small.npz"))
    im, jm, km = 7, 13, 17
    p = H.default_params(H.ATM, im, jm, km)
    f, g = H.make_inputs(H.ATM, im, jm, km, 77, "blobs", p)
    f = {k: f[k] for k in H.ATM_FIELDS}
    m = H.CpuModel("oracle", H.ATM, im, jm, km, p, f, g)
    mn, loc = m.residuum("atm_residuum3d")
    assert mn == float(z["res3d_min"]) and loc == tuple(z["res3d_loc"])
    assert H.bits_equal(m.f["RESIDUUM"][1:-1, 1:-1, 1:-1], z["res3d_field"][1:-1, 1:-1, 1:-1])
    # two queries on one Accuracy_Atm object: state carried between the calls
    m2 = H.CpuModel("oracle", H.ATM, im, jm, km, p,
                    dict(f, U=-0.5 * f["U"], V=0.25 * f["V"], W=-0.125 * f["W"]), g)
    mn2, loc2 = m2.residuum("atm_residuum3d", mn, loc)
    assert np.array_equal([mn, mn2], z["res3d_twice_min"])
    assert np.array_equal(list(loc) + list(loc2), z["res3d_twice_loc"])
    mn, loc = m.residuum("atm_residuum2d")
    assert mn == float(z["res2d_min"]) and loc == tuple(z["res2d_loc"])
    assert H.bits_equal(m.f["RESIDUUM"][0, 1:-1, 1:-1], z["res2d_field"][1:-1, 1:-1])
    ph = H.default_params(H.HYD, im, jm, km)
    fh, gh = H.make_inputs(H.HYD, im, jm, km, 78, "blobs", ph)
    mh = H.CpuModel("oracle", H.HYD, im, jm, km, ph, {k: fh[k] for k in H.HYD_FIELDS}, gh)
    mn, loc = mh.residuum("hyd_residuum3d")
    assert mn == float(z["hyd_res3d_min"]) and loc == tuple(z["hyd_res3d_loc"])


@pytest.mark.parametrize("model,tag,op", [(H.ATM, "atm", "atm_rk3d"), (H.HYD, "hyd", "hyd_rk3d")])
def test_full_1deg_real_mask_rk3d(model, tag, op):
    """One RK sweep on the full 41x181x361 grid with the real 0 Ma land mask:
    per-field SHA-256 of the result equals the compiled reference's."""
    z = np.load(os.path.join(GOLD, "full_1deg_%s_rk3d.npz" % tag))
    im, jm, km = 41, 181, 361
    p = H.default_params(model)
    f, g = H.make_inputs(model, im, jm, km, 1234, "allfluid", p)
    names = H.ATM_FIELDS if model == H.ATM else H.HYD_FIELDS
    f = {k: f[k] for k in names}
    f["H"], g["TOPOGRAPHY"] = H.load_mask_0Ma(model)
    assert _digest(f, names) + _digest(g, H.FIELDS2D) == str(z["input_sha256"]), "input generator drifted"
    m = H.CpuModel("oracle", model, im, jm, km, p, f, g)
    m.op(op)
    changed = [k[:-7] for k in z.files if k.endswith("/sha256")]
    assert len(changed) >= 9
    for k in changed:
        assert hashlib.sha256(m.f[k].tobytes()).hexdigest() == str(z[k + "/sha256"]), k
        for pk, pv in H.sample_planes(m.f[k]).items():
            assert H.bits_equal(pv, z[k + "/" + pk]), (k, pk)


def test_bc_thermo_golden():
    """The oracle's restatements of BC_Thermo's in-loop functions (limiter, latent heat, multi-layer radiation, ice
    scheme, saturation adjustment; two chained calls where members are carried) against what the compiled BC_Thermo
    class produced on the same seeded inputs: every changed field's SHA-256, the carried members bit for bit."""
    z = np.load(os.path.join(H.ROOT, "tests", "golden", "bc_thermo.npz"))
    out = H.thermo_golden_run("oracle")
    assert sorted(out) == sorted(z.files)
    for k in z.files:
        if k.endswith("sha256"):
            assert str(out[k]) == str(z[k]), k
        else:
            assert H.bits_equal(out[k], z[k]), k
    assert sum(k.endswith("sha256") for k in z.files) >= 40


def test_steady_queries_against_numpy():
    im, jm, km = 7, 13, 17
    p = H.default_params(H.ATM, im, jm, km)
    f, g = H.make_inputs(H.ATM, im, jm, km, 6, "blobs", p)
    m = H.CpuModel("oracle", H.ATM, im, jm, km, p, {k: f[k] for k in H.ATM_FIELDS}, g)
    out, loc = m.steady("atm_steady3d", 9)
    for n, k in enumerate(["U", "V", "W", "T", "C", "CLOUD", "ICE", "CO2", "P_DYN"]):
        d = np.abs(f[k] - f[k + "N"])
        assert out[n] == d.max()
        assert tuple(loc[n]) == np.unravel_index(np.argmax(d), d.shape)  # first maximum
    out, loc = m.steady("atm_steady2d", 3)
    for n, k in enumerate(["V", "W", "P_DYN"]):
        d = np.abs(f[k][0] - f[k + "N"][0])
        assert out[n] == d.max() and tuple(loc[n][1:]) == np.unravel_index(np.argmax(d), d.shape)
    ph = H.default_params(H.HYD, im, jm, km)
    fh, gh = H.make_inputs(H.HYD, im, jm, km, 7, "blobs", ph)
    fh["UN"] = fh["U"].copy(); fh["UN"][2, 3, 4] += 1.0; fh["UN"][5, 6, 7] += 1.0  # a tie: >= keeps the LAST
    mh = H.CpuModel("oracle", H.HYD, im, jm, km, ph, {k: fh[k] for k in H.HYD_FIELDS}, gh)
    out, loc = mh.steady("hyd_steady3d", 6)
    assert out[0] == 1.0 and tuple(loc[0]) == (5, 6, 7)


def test_atm_msl_data_golden_bitexact():
    """Results_MSL_Atm::run_MSL_data: the oracle against the compiled reference's outputs."""
    import ctypes as C
    im, jm, km = 5, 181, 361
    p = H.default_params(H.ATM, im, jm, km)
    f, g = H.make_msl_inputs(im, jm, km, 51, p)
    z = np.load(os.path.join(GOLD, "atm_msl_data.npz"))
    assert _digest(f, sorted(f)) + _digest(g, sorted(g)) == str(z["input_sha256"]), "input generator drifted"
    m = H.CpuModel("oracle", H.ATM, im, jm, km, p, f, g)
    m.op("atm_msl_data", C.c_int(1))
    assert H.check_msl_golden(lambda k: m.f[k], lambda k: m.g[k], exact=True) >= 14


def _same_with_sign(a, b):
    return np.array_equal(a, b) and np.array_equal(np.signbit(a), np.signbit(b))


def test_transfer_golden_bitexact():
    """Atmosphere -> hydrosphere transfer: the oracle against what the reference's writer + reader delivered."""
    z = np.load(os.path.join(GOLD, "transfer_small.npz"))
    (pa, fa, ga), (ph, fh, gh) = H.make_transfer_inputs((5, 13, 17), (6, 13, 17), 61)
    assert (_digest(fa, sorted(fa)) + _digest(ga, sorted(ga)) + _digest(fh, sorted(fh)) + _digest(gh, sorted(gh))
            == str(z["input_sha256"])), "input generator drifted"
    oa, ob = H.CpuModel("oracle", H.ATM, 5, 13, 17, pa, fa, ga), H.CpuModel("oracle", H.HYD, 6, 13, 17, ph, fh, gh)
    oa.lib.oracle_transfer_atm_to_hyd(oa.ctx, ob.ctx)
    for k in ("V", "W", "T", "P_DYN"):
        assert _same_with_sign(ob.f[k][5], z[k]), k
    for k in ("EVAPORATION_DALTON", "PRECIPITATION"):
        assert _same_with_sign(ob.g[k], z[k]), k
