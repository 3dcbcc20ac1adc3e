"""Parity of the HIP path (through the C-ABI) with the oracle on the same seeded
inputs, and with the committed golden vectors of the compiled reference.

Bars.  Everything except the atmosphere 3-D RK sweep is +,-,*,/ only and must be
BIT-EXACT (up to the sign of zero).  The atmosphere RHS calls pow() once per
point (RHS_Atm.cpp:388): glibc's and the device's pow differ in the last place,
so that sweep is held to the stated fp64 tolerance, 1e-10 relative per field
(max |a-b| / max |b|); the test also prints how far below that it actually is.
"""
import os

import numpy as np
import pytest

import atom_amd
import harness as H

pytestmark = pytest.mark.gpu

TOL = 1e-10
GOLD = os.path.join(H.ROOT, "tests", "golden")


def _pair(model, dims, seed, mask, real_mask=False, variant=None):
    """Oracle and device model on the same inputs.  variant: the form of the device's 3-D sweep; None is the handle's
    default for the atmosphere and the exact form for the hydrosphere, whose comparisons here are bit for bit;
    "default" leaves the hydrosphere on its default (the pipelined tolerance form) as well."""
    im, jm, km = dims
    p = H.default_params(model, im, jm, km)
    f, g = H.make_inputs(model, im, jm, km, seed, mask, p)
    names = H.ATM_FIELDS if model == H.ATM else H.HYD_FIELDS
    f = {k: f[k] for k in names}
    if real_mask:
        f["H"], g["TOPOGRAPHY"] = H.load_mask_0Ma(model)
    o = H.CpuModel("oracle", model, im, jm, km, p, f, g)
    dp = atom_amd.Params.from_buffer_copy(bytes(p))
    d = atom_amd.DeviceModel(model, im, jm, km, dp)
    d.upload_all(f, g)
    if variant is None and model == H.HYD:
        variant = 0
    if variant not in (None, "default"):
        d.set_rk_variant(variant)
    return o, d, names


def _dev_op(d, op):
    if op == "move3d":
        d.move_data_to_new_arrays(-1)
    elif op == "move2d":
        d.move_data_to_new_arrays(0 if d.model == atom_amd.ATM else d.im - 1)
    else:
        d.op(op)


def _compare(o, d, names, op, exact):
    worst = 0.0
    for k in names:
        if k == "RESIDUUM":
            continue
        got = d.download(k)
        if exact:
            assert H.bits_equal(got, o.f[k]), (op, k, H.max_rel_err(got, o.f[k]))
        else:
            e = H.max_rel_err(got, o.f[k])
            worst = max(worst, e)
            assert e <= TOL, (op, k, e)
    return worst


def _run_sequence(model, dims, seed, mask, seq, pow_ops=("atm_rk3d",), variant=None):
    """variant: see _pair.  With the sweeps in a tolerance form (variants 2 and 3: the 3-D and the surface sweeps alike)
    the fields are held to the tolerance from the first sweep on."""
    o, d, names = _pair(model, dims, seed, mask, variant=variant)
    if d.rk_variant() >= 2:
        pow_ops = tuple(pow_ops) + ("atm_rk2d", "hyd_rk2d", "hyd_rk3d")
    tainted = False  # once a pow-dependent op ran, later fields inherit its last-place differences
    for op in seq:
        H.run_op(o, op)
        _dev_op(d, op)
        tainted = tainted or op in pow_ops
        w = _compare(o, d, names, op, exact=not tainted)
        if tainted:
            print("%s: max rel err %.3e" % (op, w))
    d.close()


@pytest.mark.parametrize("dims,seed,mask", [((7, 13, 17), 11, "blobs"), ((9, 21, 70), 21, "blobs"),
                                            ((6, 9, 131), 22, "allfluid")])
def test_atm_sequence(dims, seed, mask):
    _run_sequence(H.ATM, dims, seed, mask, H.ATM_SEQ + ["atm_value_limitation", "move2d", "atm_rk2d", "atm_pressure2d"])


@pytest.mark.parametrize("dims,seed,mask", [((7, 13, 17), 12, "blobs"), ((9, 21, 70), 23, "blobs"),
                                            ((6, 9, 131), 24, "allfluid")])
def test_hyd_sequence_bitexact(dims, seed, mask):
    _run_sequence(H.HYD, dims, seed, mask, H.HYD_SEQ)


@pytest.mark.parametrize("dims,seed,mask", [((9, 21, 70), 23, "blobs"), ((6, 9, 131), 24, "allfluid"), ((7, 13, 200), 25, "blobs")])
def test_hyd_sequence_two_blocks_per_group_bitexact(dims, seed, mask, monkeypatch):
    """The sweep's form for grids with more blocks of lines than the chip holds side by side (two blocks per workgroup,
    ten waves), forced on small grids: two, three (one workgroup half empty) and four blocks per level."""
    monkeypatch.setenv("ATOMDEV_HYD_BLOCKS_PER_GROUP", "2")
    _run_sequence(H.HYD, dims, seed, mask, H.HYD_SEQ)


@pytest.mark.parametrize("variant", [0, "default"])
def test_hyd_surface_sequence_bitexact(variant):
    """The hydrosphere's surface sequence: bit for bit with the sweeps in their exact form, within the tolerance in the
    handle's default form (the surface sweep follows the same knob as the 3-D sweep)."""
    _run_sequence(H.HYD, (41, 43, 19), 13, "blobs", H.HYD_SURFACE_SEQ + ["hyd_pressure2d"], variant=variant)


@pytest.mark.parametrize("variant", [0, "default"])
def test_atm_everything_but_rk3d_bitexact(variant):
    """The same operators with the pow-dependent sweep left out: bit-exact with the surface sweep in its exact form,
    within the tolerance in the handle's default form."""
    seq = ["atm_bc_radius", "atm_bc_theta", "atm_bc_phi", "atm_value_limitation", "atm_solid_ground",
           "atm_rk2d", "atm_pressure2d", "move2d", "atm_rk2d", "atm_pressure3d", "move3d", "atm_pressure3d"]
    _run_sequence(H.ATM, (9, 23, 67), 31, "blobs", seq, variant=variant)


@pytest.mark.parametrize("variant", [0, "default"])
@pytest.mark.parametrize("dims", [(5, 7, 7), (5, 8, 7), (5, 34, 7), (5, 35, 40), (5, 66, 9), (5, 7, 130)])
def test_atm_surface_sweep_small_and_ragged_bitexact(dims, variant):
    """The surface sweep on surfaces around its block sizes: fewer rows than a wave holds (32), one row more than a
    wave holds, one more than two waves hold, the smallest grids the library takes, a long thin one; in its exact form
    (bit for bit) and in its tolerance form."""
    _run_sequence(H.ATM, dims, 60 + dims[1], "blobs", ["atm_rk2d", "move2d", "atm_rk2d"], variant=variant)


@pytest.mark.parametrize("variant", [0, "default"])
@pytest.mark.parametrize("jm", [301, 801, 1101])
def test_atm_surface_sweep_many_rows_bitexact(jm, variant):
    """The surface sweep runs as a pipeline of waves of 32 rows each: 10, 25 and 35 waves here (the 1-degree grid has
    six), in its exact and in its tolerance form; variant 1 is the plain form, one workgroup walking the diagonals."""
    seq = ["atm_rk2d", "move2d", "atm_rk2d"]
    _run_sequence(H.ATM, (5, jm, 11), 40 + jm, "blobs", seq, variant=variant)
    o, d, names = _pair(H.ATM, (5, 45, 37), 47, "blobs")
    d.set_rk_variant(1)  # the looping form on a small grid as well
    H.run_op(o, "atm_rk2d"); _dev_op(d, "atm_rk2d")
    _compare(o, d, names, "atm_rk2d", exact=True)
    d.close()


@pytest.mark.parametrize("dims,radiation_model,seed", [((5, 181, 361), 1, 51), ((7, 13, 17), 0, 53), ((9, 21, 70), 1, 54),
                                                       ((70, 9, 13), 1, 55), ((129, 8, 9), 1, 56)])
def test_atm_msl_data(dims, radiation_model, seed):
    """Results_MSL_Atm::run_MSL_data on the device: exp() appears in the surface pressure and the saturation
    pressure (last-place differences between glibc and the device), so those outputs are held to 1e-10;
    everything without exp (clamps, buoyancy boundary values, precipitation, co2_total, heat fluxes) is bit-exact."""
    import ctypes as C
    im, jm, km = dims
    p = H.default_params(H.ATM, im, jm, km)
    f, g = H.make_msl_inputs(im, jm, km, seed, p)
    o = H.CpuModel("oracle", H.ATM, im, jm, km, p, f, g)
    d = atom_amd.DeviceModel(H.ATM, im, jm, km, atom_amd.Params.from_buffer_copy(bytes(p)))
    d.upload_all(f, g)
    o.op("atm_msl_data", C.c_int(radiation_model))
    d.msl_data(radiation_model)
    with_exp = {"P_STAT", "EVAPORATION_DALTON", "EVAPORATION_PENMAN", "PRECIPITABLE_WATER"}
    worst = 0.0
    for k in list(f) + list(g):
        got, ref = d.download(k), (o.f[k] if k in f else o.g[k])
        if k in with_exp:
            e = H.max_rel_err(got, ref)
            worst = max(worst, e)
            assert e <= TOL, (k, e)
        else:
            assert H.bits_equal(got, ref), (k, H.max_rel_err(got, ref))
    print("msl_data: max rel err of the exp-dependent fields %.3e" % worst)
    d.close()


def test_atm_msl_data_golden():
    """HIP path against the compiled reference's own outputs (tests/golden/atm_msl_data.npz), 1e-10 on the slices kept."""
    im, jm, km = 5, 181, 361
    p = H.default_params(H.ATM, im, jm, km)
    f, g = H.make_msl_inputs(im, jm, km, 51, p)
    d = atom_amd.DeviceModel(H.ATM, im, jm, km, atom_amd.Params.from_buffer_copy(bytes(p)))
    d.upload_all(f, g)
    d.msl_data(1)
    assert H.check_msl_golden(d.download, d.download, exact=False, tol=TOL) >= 14
    d.close()


def test_transfer_atm_to_hyd_bitexact(tmp_path):
    """The coupling without the text file: bit-exact (sign of zero included) against the oracle's print-and-parse
    and against what the reference's writer + reader delivered (tests/golden/transfer_small.npz); the file written
    by the library has the reference file's bytes, and reading it back gives the same fields again."""
    import hashlib
    z = np.load(os.path.join(GOLD, "transfer_small.npz"))
    (pa, fa, ga), (ph, fh, gh) = H.make_transfer_inputs((5, 13, 17), (6, 13, 17), 61)
    oa, ob = H.CpuModel("oracle", H.ATM, 5, 13, 17, pa, fa, ga), H.CpuModel("oracle", H.HYD, 6, 13, 17, ph, fh, gh)
    oa.lib.oracle_transfer_atm_to_hyd(oa.ctx, ob.ctx)
    da = atom_amd.DeviceModel(H.ATM, 5, 13, 17, atom_amd.Params.from_buffer_copy(bytes(pa)))
    da.upload_all(fa, ga)

    def same(a, b):
        return np.array_equal(a, b) and np.array_equal(np.signbit(a), np.signbit(b))

    def check(dh, what):
        for k in fh:
            assert same(dh.download(k), ob.f[k]), (what, k)
        for k in gh:
            assert same(dh.download(k), ob.g[k]), (what, k)
        for k in ("V", "W", "T", "P_DYN"):
            assert same(dh.download(k)[5], z[k]), (what, k)
        for k in ("EVAPORATION_DALTON", "PRECIPITATION"):
            assert same(dh.download(k), z[k]), (what, k)

    dh = atom_amd.DeviceModel(H.HYD, 6, 13, 17, atom_amd.Params.from_buffer_copy(bytes(ph)))
    dh.upload_all(fh, gh)
    da.transfer_to_hyd(dh)
    check(dh, "device to device")
    path = str(tmp_path / "t.vw")
    da.write_transfer_file(path)
    assert hashlib.sha256(open(path, "rb").read()).hexdigest() == str(z["file_sha256"])
    dh2 = atom_amd.DeviceModel(H.HYD, 6, 13, 17, atom_amd.Params.from_buffer_copy(bytes(ph)))
    dh2.upload_all(fh, gh)
    dh2.read_transfer_file(path)
    check(dh2, "through the file")
    with pytest.raises(atom_amd.AtomDevError):
        dh2.read_transfer_file(str(tmp_path / "missing.vw"))
    with pytest.raises(atom_amd.AtomDevError):
        dh.transfer_to_hyd(da)
    for m in (da, dh, dh2):
        m.close()


def test_transfer_wide_range_values_bitexact():
    """390 000 surface values over twelve decades of magnitude, both signs, through the device's decimal
    rounding and through the oracle's print-and-parse: identical bits."""
    im_a, im_h, jm, km = 5, 6, 181, 361
    (pa, fa, ga), (ph, fh, gh) = H.make_transfer_inputs((im_a, jm, km), (im_h, jm, km), 62)
    rng = np.random.default_rng(63)
    wide = lambda: rng.choice([-1.0, 1.0], (jm, km)) * 10.0 ** rng.uniform(-8, 4, (jm, km))
    fa["V"][0], fa["W"][0], fa["T"][0] = wide() / 8.0, wide() / 8.0, wide()
    ga["EVAPORATION_DALTON"], ga["PRECIPITATION"] = wide(), wide()
    oa = H.CpuModel("oracle", H.ATM, im_a, jm, km, pa, fa, ga)
    ob = H.CpuModel("oracle", H.HYD, im_h, jm, km, ph, fh, gh)
    oa.lib.oracle_transfer_atm_to_hyd(oa.ctx, ob.ctx)
    da = atom_amd.DeviceModel(H.ATM, im_a, jm, km, atom_amd.Params.from_buffer_copy(bytes(pa)))
    dh = atom_amd.DeviceModel(H.HYD, im_h, jm, km, atom_amd.Params.from_buffer_copy(bytes(ph)))
    da.upload_all(fa, ga); dh.upload_all(fh, gh)
    da.transfer_to_hyd(dh)
    for k in ("V", "W", "T", "P_DYN", "EVAPORATION_DALTON", "PRECIPITATION"):
        got, ref = dh.download(k), (ob.f[k] if k in ob.f else ob.g[k])
        assert np.array_equal(got, ref) and np.array_equal(np.signbit(got), np.signbit(ref)), k
    da.close(); dh.close()


@pytest.mark.parametrize("max_precipitation", [7.5, 0.0])
def test_atm_vegetation_distribution_bitexact(max_precipitation):
    import ctypes as C
    o, d, _ = _pair(H.ATM, (7, 13, 17), 81, "blobs")
    o.f["T"][0, ::2, :] = 1.0
    d.upload("T", o.f["T"])
    o.g["VEGETATION"] = np.full((13, 17), -1.0)
    getattr(o.lib, o.pre + "bind2d")(o.ctx, H.F2["VEGETATION"], o.g["VEGETATION"].ctypes.data_as(C.c_void_p))
    o.op("atm_vegetation_distribution", C.c_double(max_precipitation))
    d.vegetation_distribution(max_precipitation)
    assert H.bits_equal(d.download("VEGETATION"), o.g["VEGETATION"])
    d.close()


@pytest.mark.parametrize("model", [H.ATM, H.HYD])
def test_minmax_searches(model):
    """Values and locations of the MinMax_* scans (the reference exposes the values only; the locations follow the
    oracle's literal restatement of the loops): ties, all-negative fields, zeros, a seed element that wins."""
    import ctypes as C
    from test_oracle_vs_reference import minmax_cases
    o, d, _ = _pair(model, (7, 13, 17), 91, "blobs")
    cases3, cases2 = minmax_cases(o.f, o.g)
    for k, v in cases3.items():
        o.f[k][...] = v; d.upload(k, v)
    for k, v in cases2.items():
        o.g[k][...] = v; d.upload(k, v)
    for k in list(cases3) + list(cases2):
        two_d = k in cases2
        out = (C.c_double * 2)(); loc = (C.c_int * 6)()
        getattr(o.lib, "oracle_minmax2d" if two_d else "oracle_minmax3d")(o.ctx, (H.F2 if two_d else H.F3)[k], out, loc)
        (vmax, lmax), (vmin, lmin) = d.minmax(k)
        assert (vmax, vmin) == (out[0], out[1]), (k, vmax, vmin, out[0], out[1])
        assert lmax + lmin == tuple(loc), (k, lmax, lmin, tuple(loc))
    d.close()


@pytest.mark.parametrize("variant", [0, 2])
def test_rk2d_batch_same_bits(variant):
    """Surface sweeps of five independent models (three atmospheres, two hydrospheres) in one launch: each gets
    the bits of its own single-model sweep; work enqueued before and after the batch keeps its order.  In the exact form
    of the sweep those are the oracle's bits, in the tolerance form the fields are within the tolerance of it."""
    dims = (41, 43, 19)
    singles, batched, oracles = [], [], []
    for n, model in enumerate([H.ATM, H.HYD, H.ATM, H.ATM, H.HYD]):
        o, d1, names = _pair(model, dims, 100 + n, "blobs", variant=variant)
        d2 = atom_amd.DeviceModel(model, *dims, atom_amd.Params.from_buffer_copy(bytes(o.params)))
        d2.upload_all(o.f, o.g)
        d2.set_rk_variant(variant)
        singles.append(d1); batched.append(d2); oracles.append((o, names))
    pre = lambda m: "atm" if m.model == atom_amd.ATM else "hyd"
    for rep in range(2):
        for d in singles:
            d.op(pre(d) + "_bc_theta"); d.op(pre(d) + "_rk2d"); d.op(pre(d) + "_pressure2d")
        for d in batched:
            d.op(pre(d) + "_bc_theta")
        atom_amd.rk2d_batch(batched)
        for d in batched:
            d.op(pre(d) + "_pressure2d")
    for d1, d2, (o, names) in zip(singles, batched, oracles):
        for rep in range(2):
            for op in ("_bc_theta", "_rk2d", "_pressure2d"):
                H.run_op(o, pre(d1) + op)
        for k in names:
            if k != "RESIDUUM":
                a, b = d1.download(k), d2.download(k)
                assert H.bits_equal(a, b), k
                if variant == 0:
                    assert H.bits_equal(a, o.f[k]), k
                else:
                    assert H.max_rel_err(a, o.f[k]) <= TOL, (k, H.max_rel_err(a, o.f[k]))
    with pytest.raises(atom_amd.AtomDevError):
        atom_amd.rk2d_batch([batched[0], batched[0]])
    batched[1].set_rk_variant(2 - variant)   # the two forms of the sweep do not go into one batch
    with pytest.raises(atom_amd.AtomDevError):
        atom_amd.rk2d_batch(batched)
    for d in singles + batched:
        d.close()


@pytest.mark.parametrize("dims,seed", [((9, 21, 70), 57), ((7, 13, 17), 58)])
def test_atm_latent_heat_bitexact(dims, seed):
    """BC_Thermo::Latent_Heat: +, -, *, / only, so bit-exact against the oracle's restatement (itself pinned bit
    for bit by the compiled BC_Thermo class: test_oracle_vs_reference.py, golden/bc_thermo.npz)."""
    im, jm, km = dims
    p, f, g = H.make_latent_heat_inputs(im, jm, km, seed)
    o = H.CpuModel("oracle", H.ATM, im, jm, km, p, f, g)
    d = atom_amd.DeviceModel(H.ATM, im, jm, km, atom_amd.Params.from_buffer_copy(bytes(p)))
    d.upload_all(f, g)
    o.op("atm_latent_heat"); d.op("atm_latent_heat")
    for k in ("Q_LATENT", "Q_SENSIBLE"):
        assert H.bits_equal(d.download(k), o.f[k]), k
        assert not np.array_equal(o.f[k], f[k])
    d.close()


@pytest.mark.parametrize("dims,seed,trop", [((41, 19, 23), 59, 30), ((12, 13, 17), 60, 8), ((9, 21, 70), 61, 3),
                                            ((41, 37, 45), 76, 12), ((60, 9, 11), 77, 40)])   # 60 levels: one thread per column
def test_atm_radiation_multi_layer(dims, seed, trop):
    """BC_Thermo::BC_Radiation_multi_layer against the oracle's statement-by-statement restatement (which builds
    the reference's full transmission matrix): pow() and sqrt() on both sides, 1e-10 per field; the oracle is pinned bit
    for bit by the compiled BC_Thermo class (test_oracle_vs_reference.py, golden/bc_thermo.npz).  trop = 3 puts
    mountains above the tropopause."""
    import ctypes as C
    im, jm, km = dims
    p, f, g, i_trop = H.make_radiation_inputs(im, jm, km, seed, trop)
    o = H.CpuModel("oracle", H.ATM, im, jm, km, p, f, g)
    d = atom_amd.DeviceModel(H.ATM, im, jm, km, atom_amd.Params.from_buffer_copy(bytes(p)))
    d.upload_all(f, g)
    r = atom_amd.Radiation(); d.lib.atomdev_default_radiation(C.byref(r))
    o.lib.oracle_atm_radiation_multi_layer(o.ctx, C.byref(r), i_trop.ctypes.data_as(C.POINTER(C.c_int)))
    d.radiation_multi_layer(i_trop, r)
    worst = 0.0
    for k in list(f) + list(g):
        ref = o.f[k] if k in f else o.g[k]
        e = H.max_rel_err(d.download(k), ref)
        worst = max(worst, e)
        assert e <= TOL, (k, e)
    for k in ("T", "RADIATION_3D", "EPSILON_3D", "ALBEDO", "EPSILON", "RADIATION_SURFACE"):
        assert not np.array_equal(o.f[k] if k in f else o.g[k], f[k] if k in f else g[k]), k
    assert np.isfinite(o.f["T"]).all()
    print("radiation: max rel err %.3e" % worst)
    d.close()


@pytest.mark.parametrize("dims,seed,first", [((41, 19, 23), 62, "warm"), ((12, 13, 17), 63, "icy"), ((9, 21, 70), 64, "warm"),
                                             ((41, 37, 45), 68, "icy"), ((70, 7, 9), 69, "warm"),
                                             ((130, 7, 9), 70, "icy")])   # 130 levels: the one-thread-per-column kernels
def test_atm_two_category_ice_scheme(dims, seed, first):
    """BC_Thermo::Two_Category_Ice_Scheme against the oracle's restatement in the reference's serial visiting
    order (exp() and pow() on both sides, 1e-10 per field); the oracle is pinned bit for bit by the compiled
    BC_Thermo class (test_oracle_vs_reference.py, golden/bc_thermo.npz).  S_nuc and S_i_dep are assigned only under
    conditions and otherwise keep the value of the grid point visited before - across columns, passes and calls:
    the first columns of the visiting order are built so that they assign nothing ("warm": S_nuc, "icy":
    S_i_dep), which makes them work with the caller's state in the first pass and with the grid's last
    assignment afterwards; two calls in a row hand the state on."""
    import ctypes as C
    im, jm, km = dims
    p, f, g, rng = H.make_ice_scheme_inputs(im, jm, km, seed, first)
    shp = (im, jm, km)
    o = H.CpuModel("oracle", H.ATM, im, jm, km, p, f, g)
    d = atom_amd.DeviceModel(H.ATM, im, jm, km, atom_amd.Params.from_buffer_copy(bytes(p)))
    d.upload_all(f, g)
    o.lib.oracle_atm_two_category_ice_scheme.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    o.lib.oracle_atm_two_category_ice_scheme.restype = None
    st_o = (C.c_double * 2)(3.0e-7, -2.0e-7)
    st_d = (3.0e-7, -2.0e-7)
    worst = 0.0
    for call in range(2):
        o.lib.oracle_atm_two_category_ice_scheme(o.ctx, st_o)
        st_d = d.two_category_ice_scheme(st_d)
        for n in range(2):
            assert abs(st_d[n] - st_o[n]) <= TOL * max(abs(st_o[n]), 1e-300), (call, n, st_d[n], st_o[n])
        for k in list(f):
            e = H.max_rel_err(d.download(k), o.f[k])
            worst = max(worst, e)
            assert e <= TOL, (call, k, e)
        if call == 0:   # between the calls the model moves on: new moisture, same geometry
            for k in ("C", "ICE"):
                x = o.f[k] * rng.uniform(0.5, 1.5, shp)
                o.f[k][...] = x; d.upload(k, x)
    for k in ("S_V", "S_C", "S_I", "S_R", "S_S", "P_RAIN", "P_SNOW", "C", "ICE", "CLOUD"):
        assert not np.array_equal(o.f[k], f[k]), k
    assert (o.f["P_RAIN"] > 0).any() and (o.f["P_SNOW"] > 0).any()
    print("two-category ice scheme: max rel err %.3e, state %s" % (worst, st_d))
    d.close()


@pytest.mark.parametrize("dims,seed", [((41, 19, 23), 65), ((12, 13, 17), 66), ((9, 21, 70), 67),
                                       ((130, 9, 9), 71)])   # 130 levels: the one-thread-per-column kernel
def test_atm_ice_water_saturation_adjustment(dims, seed):
    """BC_Thermo::Ice_Water_Saturation_Adjustment against the oracle's restatement in the reference's serial
    visiting order (exp() and pow() inside an iteration with a 1e-5 exit test; 1e-10 per field); the oracle is
    pinned bit for bit by the compiled BC_Thermo class (test_oracle_vs_reference.py, golden/bc_thermo.npz).  The member q_v_hyp
    is inherited by cold points whose cloud water or ice comes in negative: the inputs hold such points, whole
    columns that start with one (three of them in a row in visiting order, and the very first column, which
    takes the caller's value), columns without a cold point, and warm points below, at and above saturation."""
    import ctypes as C
    im, jm, km = dims
    p, f, g, rng = H.make_adjustment_inputs(im, jm, km, seed)
    shp = (im, jm, km)
    o = H.CpuModel("oracle", H.ATM, im, jm, km, p, f, g)
    d = atom_amd.DeviceModel(H.ATM, im, jm, km, atom_amd.Params.from_buffer_copy(bytes(p)))
    d.upload_all(f, g)
    o.lib.oracle_atm_ice_water_saturation_adjustment.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    o.lib.oracle_atm_ice_water_saturation_adjustment.restype = None
    st_o = C.c_double(4.0e-3)
    st_d = 4.0e-3
    worst = 0.0
    for call in range(2):
        o.lib.oracle_atm_ice_water_saturation_adjustment(o.ctx, C.byref(st_o))
        st_d = d.ice_water_saturation_adjustment(st_d)
        assert abs(st_d - st_o.value) <= TOL * abs(st_o.value), (call, st_d, st_o.value)
        for k in list(f):
            assert np.isfinite(o.f[k]).all(), k
            e = H.max_rel_err(d.download(k), o.f[k])
            worst = max(worst, e)
            assert e <= TOL, (call, k, e)
        if call == 0:   # the model moves on: moisture changes, some of it to negative values
            for k, lo, hi in (("C", 0.7, 1.02), ("CLOUD", 0.5, 1.5), ("ICE", 0.5, 1.5)):
                x = o.f[k] * rng.uniform(lo, hi, shp) - 2e-5
                o.f[k][...] = x; d.upload(k, x)
    for k in ("T", "C", "CLOUD", "ICE", "CN", "CLOUDN", "ICEN", "S_C_C"):
        assert not np.array_equal(o.f[k], f[k]), k
    for k in ("C", "CLOUD", "ICE"):
        assert (o.f[k] >= 0).all()
    assert o.f["T"].max() < 1.2 and o.f["CLOUD"].max() < 0.1          # the iterations stayed bounded
    print("saturation adjustment: max rel err %.3e, q_v_hyp %r" % (worst, st_d))
    d.close()


@pytest.mark.parametrize("dims,L_hyd", [((41, 19, 23), 1000.0), ((41, 19, 23), 6000.0), ((9, 21, 70), 1000.0),
                                        ((12, 13, 17), 2000.0)])
def test_hyd_run_data_bitexact(dims, L_hyd):
    """Results_Hyd::run_data: every 3-D and surface field, bits and sign of zero (the oracle is pinned by the compiled
    class in tests/test_oracle_vs_reference.py).  (12, ., .) with 2000 m: odd number of Simpson intervals."""
    im, jm, km = dims
    p, f, g = H.make_run_data_inputs(im, jm, km, 94, L_hyd)
    o = H.CpuModel("oracle", H.HYD, im, jm, km, p, f, g)
    d = atom_amd.DeviceModel(H.HYD, im, jm, km, atom_amd.Params.from_buffer_copy(bytes(p)))
    d.upload_all(f, g)
    o.op("hyd_run_data"); d.op("hyd_run_data")
    for k in list(f) + list(g):
        got, ref = d.download(k), (o.f[k] if k in f else o.g[k])
        assert np.array_equal(got, ref) and np.array_equal(np.signbit(got), np.signbit(ref)), k
    for k in H.RUN_DATA_FIELDS2D:
        assert not np.array_equal(o.g[k], g[k]), k
    d.close()


@pytest.mark.parametrize("model", [H.ATM, H.HYD])
def test_plot_data_file_bytes(model, tmp_path):
    """The PlotData table written from device-resident fields is byte for byte the oracle's file, which
    tests/test_oracle_vs_reference.py pins to the file of the reference's own writer."""
    import ctypes as C
    im, jm, km = 7, 13, 17
    if model == H.ATM:
        p = H.default_params(H.ATM, im, jm, km)
        f, g = H.make_msl_inputs(im, jm, km, 92, p)
    else:
        p, f, g = H.make_run_data_inputs(im, jm, km, 93)
    f["V"][0 if model == H.ATM else im - 1, 3, 4] = -2e-6     # prints as -0.0000
    o = H.CpuModel("oracle", model, im, jm, km, p, f, g)
    d = atom_amd.DeviceModel(model, im, jm, km, atom_amd.Params.from_buffer_copy(bytes(p)))
    d.upload_all(f, g)
    o.lib.oracle_write_plot_data.argtypes = [C.c_void_p, C.c_char_p]
    assert o.lib.oracle_write_plot_data(o.ctx, str(tmp_path / "oracle.xyz").encode()) == 0
    d.write_plot_data(str(tmp_path / "device.xyz"))
    mine, theirs = (tmp_path / "device.xyz").read_bytes(), (tmp_path / "oracle.xyz").read_bytes()
    assert mine == theirs and mine.count(b"\n") == 1 + jm * km and b"-0.0000" in mine
    with pytest.raises(atom_amd.AtomDevError):
        d.write_plot_data(str(tmp_path / "no_such_dir" / "x.xyz"))
    d.close()


def _first_difference(a, b):
    la, lb = a.split(b"\n"), b.split(b"\n")
    for n, (x, y) in enumerate(zip(la, lb)):
        if x != y:
            return "line %d: %r != %r (after %r)" % (n + 1, x, y, la[max(0, n - 3):n])
    return "lengths %d != %d lines" % (len(la), len(lb))


@pytest.mark.parametrize("Ma", [0, 30])
@pytest.mark.parametrize("section,index", [(0, 0), (0, 3), (1, 5), (2, 9)])
@pytest.mark.parametrize("model", [H.ATM, H.HYD])
def test_vtk_section_file_bytes(model, section, index, Ma, tmp_path):
    """The ParaView section files written from device-resident fields are byte for byte the oracle's files (pinned to
    the reference's writers in tests/test_oracle_vs_reference.py), and the scratch the writers leave behind in
    aux_u / aux_v / aux_w and the Celsius copy of temperature_NASA sits in the same device fields, same bits."""
    import ctypes as C
    if model == H.HYD and Ma:
        pytest.skip("the hydrosphere writers take no Ma")
    im, jm, km = 7, 13, 17
    p, f, g = H.make_vtk_inputs(model, im, jm, km, 95)
    o = H.CpuModel("oracle", model, im, jm, km, p, f, g)
    d = atom_amd.DeviceModel(model, im, jm, km, atom_amd.Params.from_buffer_copy(bytes(p)))
    d.upload_all(f, g)
    o.lib.oracle_write_vtk.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_char_p]
    assert o.lib.oracle_write_vtk(o.ctx, section, index, Ma, str(tmp_path / "oracle.vtk").encode()) == 0
    d.write_vtk(section, index, str(tmp_path / "device.vtk"), Ma)
    mine, theirs = (tmp_path / "device.vtk").read_bytes(), (tmp_path / "oracle.vtk").read_bytes()
    assert mine == theirs, _first_difference(mine, theirs)
    assert b"-0.0000" in mine
    for k in list(f) + list(g):
        got, ref = d.download(k), (o.f[k] if k in f else o.g[k])
        assert np.array_equal(got, ref) and np.array_equal(np.signbit(got), np.signbit(ref)), k
    with pytest.raises(atom_amd.AtomDevError):
        d.write_vtk(section, index, str(tmp_path / "no_such_dir" / "x.vtk"), Ma)
    with pytest.raises(atom_amd.AtomDevError):
        d.write_vtk(section, (im, jm, km)[section], str(tmp_path / "outside.vtk"), Ma)
    d.close()


@pytest.mark.parametrize("model", [H.ATM, H.HYD])
def test_panorama_vts_file_bytes(model, tmp_path):
    """paraview_panorama_vts from device-resident fields: byte for byte the oracle's file (pinned to the reference's
    writers in tests/test_oracle_vs_reference.py)."""
    import ctypes as C
    im, jm, km = 7, 13, 17
    p, f, g = H.make_vtk_inputs(model, im, jm, km, 97)
    o = H.CpuModel("oracle", model, im, jm, km, p, f, g)
    d = atom_amd.DeviceModel(model, im, jm, km, atom_amd.Params.from_buffer_copy(bytes(p)))
    d.upload_all(f, g)
    o.lib.oracle_write_panorama_vts.argtypes = [C.c_void_p, C.c_char_p]
    assert o.lib.oracle_write_panorama_vts(o.ctx, str(tmp_path / "oracle.vts").encode()) == 0
    d.write_panorama_vts(str(tmp_path / "device.vts"))
    mine, theirs = (tmp_path / "device.vts").read_bytes(), (tmp_path / "oracle.vts").read_bytes()
    assert mine == theirs, _first_difference(mine, theirs)
    with pytest.raises(atom_amd.AtomDevError):
        d.write_panorama_vts(str(tmp_path / "no_such_dir" / "x.vts"))
    d.close()


@pytest.mark.parametrize("model", [H.ATM, H.HYD])
def test_vtk_sections_full_grid(model, tmp_path):
    """The three sections each model writes per time slice — atmosphere i = 0, j = 62, k = 87
    (cAtmosphereModel.cpp:610-626), hydrosphere j = 75, k = 185, i = 40 (cHydrosphereModel.cpp:745-759) — on the
    1-degree grid with the 0 Ma mask: byte for byte the oracle's files."""
    import ctypes as C
    import hashlib
    im, jm, km = 41, 181, 361
    p, f, g = H.make_vtk_inputs(model, im, jm, km, 96)
    f["H"], g["TOPOGRAPHY"] = H.load_mask_0Ma(model)
    o = H.CpuModel("oracle", model, im, jm, km, p, f, g)
    d = atom_amd.DeviceModel(model, im, jm, km, atom_amd.Params.from_buffer_copy(bytes(p)))
    d.upload_all(f, g)
    o.lib.oracle_write_vtk.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_char_p]
    for section, index in (((0, 0), (1, 62), (2, 87)) if model == H.ATM else ((1, 75), (2, 185), (0, 40))):
        assert o.lib.oracle_write_vtk(o.ctx, section, index, 0, str(tmp_path / "oracle.vtk").encode()) == 0
        d.write_vtk(section, index, str(tmp_path / "device.vtk"), 0)
        mine, theirs = (tmp_path / "device.vtk").read_bytes(), (tmp_path / "oracle.vtk").read_bytes()
        assert hashlib.sha256(mine).digest() == hashlib.sha256(theirs).digest(), _first_difference(mine, theirs)
    for k in ("AUX_U", "AUX_V", "AUX_W"):
        assert np.array_equal(d.download(k), o.f[k]), k
    d.close()


@pytest.mark.parametrize("dims,i_beg,seed", [((9, 181, 23), 0, 71), ((12, 181, 70), 3, 72)])
def test_hyd_start_of_slice(dims, i_beg, seed, tmp_path):
    """cHydrosphereModel.cpp:248-292 call by call (the oracle is pinned by the compiled classes in
    tests/test_oracle_vs_reference.py): both readers, IC_CircumPolar_Current, BC_Temperature_Salinity and
    BC_Pressure_Density bit-exact with the sign of zero; IC_v_w_EkmanSpiral (asin, cos, sin of the device's math
    library at the surface) within 1e-10 relative in v and w, every other field bit-exact."""
    im, jm, km = dims
    p, f, g = H.make_hyd_start_inputs(im, jm, km, seed)
    o = H.CpuModel("oracle", H.HYD, im, jm, km, p, f, g)
    d = atom_amd.DeviceModel(H.HYD, im, jm, km, atom_amd.Params.from_buffer_copy(bytes(p)))
    d.upload_all(f, g)
    bath, salt = str(tmp_path / "bath.xyz"), str(tmp_path / "salt.xyz")
    H.write_bathymetry_file(bath, jm, km, seed)
    H.write_salinity_file(salt, jm, km, seed + 1)
    st = H.hyd_start(im, i_beg)
    dst = atom_amd.HydStart.from_buffer_copy(bytes(st))
    loose = set()

    def check(what):
        for k in list(f) + list(g):
            got, ref = d.download(k), (o.f[k] if k in f else o.g[k])
            if k in loose:
                e = H.max_rel_err(got, ref)
                assert e <= TOL, (what, k, e)
                print("%s: %s max rel err %.3e" % (what, k, e))
            else:
                assert np.array_equal(got, ref) and np.array_equal(np.signbit(got), np.signbit(ref)), (what, k)

    o.op("hyd_read_sea_ground", bath.encode()); d.read_sea_ground(bath)
    check("BC_SeaGround")
    o.op("hyd_read_surface_salinity", salt.encode()); d.read_surface_salinity(salt)
    check("BC_Surface_Salinity_NASA")
    before = o.f["V"].copy()
    o.op("hyd_ic_ekman_spiral", H.C.c_int(i_beg)); d.ic_ekman_spiral(i_beg)
    loose |= {"V", "W"}
    check("IC_v_w_EkmanSpiral")
    assert not np.array_equal(o.f["V"][i_beg], before[i_beg])
    o.op("hyd_ic_circumpolar_current", H.C.c_int(i_beg)); d.ic_circumpolar_current(i_beg)
    check("IC_CircumPolar_Current")
    o.op("hyd_bc_temperature_salinity", H.C.byref(st)); d.bc_temperature_salinity(dst)
    check("BC_Temperature_Salinity")
    o.op("hyd_pressure_density"); d.op("hyd_pressure_density")
    check("BC_Pressure_Density")
    with pytest.raises(atom_amd.AtomDevError):
        d.read_sea_ground(str(tmp_path / "missing.xyz"))
    (tmp_path / "short.xyz").write_text("0 0 -5\n")
    with pytest.raises(atom_amd.AtomDevError):
        d.read_sea_ground(str(tmp_path / "short.xyz"))     # the reference aborts on a file of the wrong size
    d.close()


def test_atm_read_mountain_surface(tmp_path):
    """BC_MountainSurface from the device library: h and Topography bit for bit the oracle's (pinned by the compiled
    class in tests/test_oracle_vs_reference.py); a short file is an error, as the reference aborts on it."""
    im, jm, km = 9, 21, 31
    o, d, names = _pair(H.ATM, (im, jm, km), 83, "blobs")
    topo = str(tmp_path / "topo.xyz")
    H.write_topography_file(topo, jm, km, 84)
    o.op("atm_read_mountain_surface", topo.encode()); d.read_mountain_surface(topo)
    for k in ("H", "T", "U"):
        assert np.array_equal(d.download(k), o.f[k]), k
    assert np.array_equal(d.download("TOPOGRAPHY"), o.g["TOPOGRAPHY"])
    d.op("atm_solid_ground"); H.run_op(o, "atm_solid_ground")      # the new mask is the one the operators see
    for k in ("U", "CO2"):
        assert H.bits_equal(d.download(k), o.f[k]), k
    (tmp_path / "short.xyz").write_text("0 0 5\n")
    with pytest.raises(atom_amd.AtomDevError):
        d.read_mountain_surface(str(tmp_path / "short.xyz"))
    d.close()


def test_hyd_start_needs_the_one_degree_rows():
    p = H.default_params(H.HYD, 9, 21, 31)
    d = atom_amd.DeviceModel(H.HYD, 9, 21, 31, atom_amd.Params.from_buffer_copy(bytes(p)))
    with pytest.raises(atom_amd.AtomDevError):
        d.ic_ekman_spiral(0)
    with pytest.raises(atom_amd.AtomDevError):
        d.ic_circumpolar_current(0)
    d.close()


@pytest.mark.parametrize("variant", [None, 0])
def test_hyd_sweeps_of_concurrent_handles_same_bits(variant):
    """Volume sweeps of four hydrosphere handles enqueued side by side on their streams (the slices of a series share
    one GPU this way): level im-2 of each waits for rows another workgroup of its own launch stores, while the
    launches compete for the compute units - with each other and with the surface sweeps of two more models, whose
    waves wait for each other's rows in the same way and are enqueued first.  Every handle must end with the bits a
    handle running alone gets, and no wait may give up (sync reports that).  In the handle's default form of the sweep
    (the pipelined one) and in the exact form."""
    im, jm, km = 41, 181, 361
    p = H.default_params(H.HYD, im, jm, km)
    f, g = H.make_inputs(H.HYD, im, jm, km, 77, "allfluid", p)
    f = {k: f[k] for k in H.HYD_FIELDS}
    f["H"], g["TOPOGRAPHY"] = H.load_mask_0Ma(H.HYD)
    models = []
    for _ in range(6):
        d = atom_amd.DeviceModel(H.HYD, im, jm, km, atom_amd.Params.from_buffer_copy(bytes(p)))
        d.upload_all(f, g)
        if variant is not None:
            d.set_rk_variant(variant)
        models.append(d)
    models, others = models[:4], models[4:]
    for _ in range(3):
        models[0].op("hyd_rk3d")
    models[0].sync()
    alone = {k: models[0].download(k) for k in ("T", "U", "V", "W", "C", "AUX_U", "SALT_BALANCE")}
    for _ in range(3):
        others[0].op("hyd_rk2d")
    others[0].sync()
    surface_alone = {k: others[0].download(k) for k in ("V", "W")}
    models[0].upload_all(f, g)
    others[0].upload_all(f, g)
    for _ in range(3):
        for d in others:
            d.op("hyd_rk2d")
        for d in models:
            d.op("hyd_rk3d")
    for d in models + others:
        d.sync()
    for n, d in enumerate(models):
        for k, ref in alone.items():
            assert H.bits_equal(d.download(k), ref), (n, k)
        d.close()
    for n, d in enumerate(others):
        for k, ref in surface_alone.items():
            assert H.bits_equal(d.download(k), ref), ("surface", n, k)
        d.close()


def test_golden_small_hyd_bitexact():
    """HIP path against the compiled reference's own outputs (tests/golden/hyd_small.npz)."""
    z = np.load(os.path.join(GOLD, "hyd_small.npz"))
    im, jm, km = (int(x) for x in z["dims"])
    o, d, names = _pair(H.HYD, (im, jm, km), int(z["seed"]), "blobs")
    for step, op in enumerate(H.HYD_SEQ):
        _dev_op(d, op)
        for key in [k for k in z.files if k.startswith("%02d:%s/" % (step, op))]:
            assert H.bits_equal(d.download(key.split("/")[1]), z[key]), key
    d.close()


def test_golden_small_atm():
    z = np.load(os.path.join(GOLD, "atm_small.npz"))
    im, jm, km = (int(x) for x in z["dims"])
    o, d, names = _pair(H.ATM, (im, jm, km), int(z["seed"]), "blobs")
    tainted = False
    for step, op in enumerate(H.ATM_SEQ):
        _dev_op(d, op)
        tainted = tainted or op == "atm_rk3d"
        for key in [k for k in z.files if k.startswith("%02d:%s/" % (step, op))]:
            got = d.download(key.split("/")[1])
            if tainted:
                assert H.max_rel_err(got, z[key]) <= TOL, key
            else:
                assert H.bits_equal(got, z[key]), key
    d.close()


@pytest.mark.parametrize("model,tag,op,variant", [(H.ATM, "atm", "atm_rk3d", None), (H.ATM, "atm", "atm_rk3d", 2),
                                                  (H.ATM, "atm", "atm_rk3d", 3), (H.ATM, "atm", "atm_rk3d", 0),
                                                  (H.HYD, "hyd", "hyd_rk3d", None), (H.HYD, "hyd", "hyd_rk3d", 0),
                                                  (H.HYD, "hyd", "hyd_rk3d", 2), (H.HYD, "hyd", "hyd_rk3d", 3)])
def test_full_1deg_real_mask_rk3d(model, tag, op, variant):
    """One sweep on 41x181x361 with the real 0 Ma land mask, against the oracle
    (whole fields) and against the planes the compiled reference left in the goldens; the atmosphere in its
    default form and in each of the wave-specialised (0), tolerance (2) and pipelined (3) forms; the hydrosphere in its
    default form (the pipelined one) and in its tolerance form within the tolerance, in its exact form bit for bit."""
    o, d, names = _pair(model, (41, 181, 361), 1234, "allfluid", real_mask=True, variant="default" if variant is None else variant)
    assert d.rk_variant() == ((2 if model == H.ATM else 3) if variant is None else variant) or "ATOMDEV_RK_VARIANT" in os.environ
    z = np.load(os.path.join(GOLD, "full_1deg_%s_rk3d.npz" % tag))
    o.op(op)
    d.op(op)
    exact = model == H.HYD and variant == 0
    nbits = 0
    for k in names:
        if k == "RESIDUUM":
            continue
        got = d.download(k)
        if exact:
            assert H.bits_equal(got, o.f[k]), k
        else:
            e = H.max_rel_err(got, o.f[k])
            nbits += int(np.sum(got != o.f[k]))
            assert e <= TOL, (k, e)
            print("%s: max rel err %.3e" % (k, e))
        if k + "/j62" in z.files:
            for pk, pv in H.sample_planes(got).items():
                if exact:
                    assert H.bits_equal(pv, z[k + "/" + pk]), (k, pk)
                else:
                    assert H.max_rel_err(pv, z[k + "/" + pk]) <= TOL, (k, pk)
    if not exact:
        print("values differing in any bit: %d" % nbits)
    d.close()


def test_multiple_sweeps_one_time_slice_atm():
    """The default slice runs 2 x 2 velocity iterations + 2 pressure steps
    (cAtmosphereModel.cpp:756-874); the hot-path operators in that order, 4 sweeps,
    stay within tolerance of the oracle on a mid-size grid."""
    o, d, names = _pair(H.ATM, (11, 45, 91), 41, "blobs")
    for pit in range(2):
        for vit in range(2):
            for op in ["atm_bc_radius", "atm_bc_theta", "atm_bc_phi", "atm_value_limitation", "atm_solid_ground",
                       "atm_rk3d", "atm_value_limitation", "move3d"]:
                H.run_op(o, op); _dev_op(d, op)
        H.run_op(o, "atm_pressure3d"); _dev_op(d, "atm_pressure3d")
    w = _compare(o, d, names, "slice", exact=False)
    print("after one slice of hot-path operators: max rel err %.3e" % w)
    d.close()


def test_residuum_and_steady_queries():
    o, d, names = _pair(H.ATM, (9, 21, 70), 51, "blobs")
    r1 = o.residuum("atm_residuum3d"); g1 = d.residuum("atm_residuum3d")
    assert r1 == g1
    assert H.bits_equal(d.download("RESIDUUM")[1:-1, 1:-1, 1:-1], o.f["RESIDUUM"][1:-1, 1:-1, 1:-1])
    # second query on the same object: state carried in
    o.op("atm_bc_radius"); d.op("atm_bc_radius")
    assert o.residuum("atm_residuum3d", *r1) == d.residuum("atm_residuum3d", *g1)
    assert o.residuum("atm_residuum2d") == d.residuum("atm_residuum2d")
    a, la = o.steady("atm_steady3d", 9); b, lb = d.steady("atm_steady3d", 9)
    assert np.array_equal(a, b) and np.array_equal(la, lb)
    a, la = o.steady("atm_steady2d", 3); b, lb = d.steady("atm_steady2d", 3)
    assert np.array_equal(a, b) and np.array_equal(la, lb)
    d.close()
    o, d, names = _pair(H.HYD, (9, 21, 70), 52, "blobs")
    assert o.residuum("hyd_residuum3d") == d.residuum("hyd_residuum3d")
    a, la = o.steady("hyd_steady3d", 6); b, lb = d.steady("hyd_steady3d", 6)
    assert np.array_equal(a, b) and np.array_equal(la, lb)
    d.close()


def test_residuum_full_grid_many_chunks():
    """Full grid: 2.5 M points = 1224 scan chunks; zero velocities over land give long runs of ties."""
    o, d, names = _pair(H.ATM, (41, 181, 361), 61, "allfluid", real_mask=True)
    for m in (o,):
        H.run_op(m, "atm_value_limitation")
    d.op("atm_value_limitation")
    assert o.residuum("atm_residuum3d") == d.residuum("atm_residuum3d")
    d.close()


def test_errors_surface_as_exceptions():
    d = atom_amd.DeviceModel(atom_amd.ATM, 7, 13, 17)
    with pytest.raises(atom_amd.AtomDevError):
        d.op("hyd_rk3d")
    with pytest.raises(atom_amd.AtomDevError):
        d.upload("R_WATER", np.zeros((7, 13, 17)))
    d.close()


@pytest.mark.parametrize("model,op,dims,seed", [(H.ATM, "atm_rk3d", (9, 21, 70), 41), (H.ATM, "atm_rk3d", (6, 9, 131), 42),
                                                (H.HYD, "hyd_rk3d", (9, 21, 70), 43), (H.HYD, "hyd_rk3d", (6, 9, 131), 44)])
def test_rk_variants_same_bits(model, op, dims, seed):
    """The wave-specialised sweep (variant 0) and the one-thread-per-line sweep (variant 1) are the same
    arithmetic on different mappings: every field bit for bit, over two consecutive sweeps (the hydrosphere
    swaps its buffer pair in between).  Variant 2 is variant 0 in its tolerance form (reciprocal
    multiplications, fused multiply-adds): every field within 1e-12 of it."""
    out = []
    variants = (0, 1, 2, 3)   # 3: the pipelined sweep (tolerance form)
    for variant in variants:
        o, d, names = _pair(model, dims, seed, "blobs")
        o.close()
        d.set_rk_variant(variant)
        d.op(op)
        d.op(op)
        out.append({k: d.download(k) for k in names})
        d.close()
    for k in out[0]:
        assert H.bits_equal(out[0][k], out[1][k]), (op, k, H.max_rel_err(out[0][k], out[1][k]))
        if k != "RESIDUUM":
            for n in range(2, len(variants)):
                assert H.max_rel_err(out[n][k], out[0][k]) <= 1e-12, (op, variants[n], k, H.max_rel_err(out[n][k], out[0][k]))


@pytest.mark.parametrize("dims,seed,real_mask", [((9, 21, 70), 51, False), ((6, 9, 131), 52, False), ((5, 7, 7), 53, False),
                                                 ((41, 181, 361), 54, True)])
def test_atm_rk3d_limited_is_sweep_then_limiter(dims, seed, real_mask):
    """atomdev_atm_rk3d_limited against atomdev_atm_rk3d followed by atomdev_atm_value_limitation, and
    atomdev_atm_limit_ground_rk3d_limit against limiter, solid ground, sweep, limiter: every field bit for bit,
    in every form of the sweep, over two rounds (the second starts from limited fields).  The inputs are stretched so
    that every limit is exceeded somewhere, p_dyn is nonzero on land, and values on the six faces need limiting too."""
    im, jm, km = dims
    p = H.default_params(H.ATM, im, jm, km)
    f, g = H.make_inputs(H.ATM, im, jm, km, seed, "blobs", p)
    f = {k: f[k] for k in H.ATM_FIELDS}
    if real_mask:
        f["H"], g["TOPOGRAPHY"] = H.load_mask_0Ma(H.ATM)
    rng = np.random.default_rng(seed)
    for k, scale in (("U", 4.), ("V", 4.), ("W", 40.), ("T", 1.3), ("C", 6.), ("CLOUD", 30.), ("ICE", 30.), ("CO2", 6.)):
        stretch = np.where(rng.random(f[k].shape) < .2, scale, 1.)
        sign = np.where(rng.random(f[k].shape) < .1, -1., 1.)
        f[k] = f[k] * stretch * sign
        f[k + "N"] = f[k].copy()
    f["P_DYN"] = f["P_DYN"] + 1.
    hit = {}
    seqs = {"two": ["atm_rk3d", "atm_value_limitation"], "two in one": ["atm_rk3d_limited"],
            "four": ["atm_value_limitation", "atm_solid_ground", "atm_rk3d", "atm_value_limitation"],
            "four in one": ["atm_limit_ground_rk3d_limit"]}
    for variant in (0, 1, 2, 3):
        out = {}
        for name, seq in seqs.items():
            d = atom_amd.DeviceModel(H.ATM, im, jm, km, atom_amd.Params.from_buffer_copy(bytes(p)))
            d.upload_all(f, g)
            d.set_rk_variant(variant)
            for _ in range(2):
                for op in seq:
                    d.op(op)
                    if op == "atm_rk3d" and name == "two" and not hit:
                        u = d.download("U"); hit["u"] = int((np.abs(u[1:-1, 1:-1, 1:-1]) > .106).sum())
            out[name] = {k: d.download(k) for k in H.ATM_FIELDS}
            d.close()
        for k in H.ATM_FIELDS:
            if k != "RESIDUUM":
                for n in ("two", "four"):
                    assert H.bits_equal(out[n][k], out[n + " in one"][k]), (variant, n, k, H.max_rel_err(out[n + " in one"][k], out[n][k]))
    assert hit["u"] > 0   # the limiter had something to do on interior points


@pytest.mark.parametrize("dims,seed", [((7, 7, 7), 71), ((5, 8, 9), 72), ((9, 21, 70), 73), ((70, 9, 13), 74), ((41, 181, 361), 75)])
def test_atm_bc_fills_in_one_launch_same_bits(dims, seed):
    """atomdev_atm_bc_radius_theta_phi against atomdev_atm_bc_radius, _bc_theta, _bc_phi: every field bit for bit, twice in
    a row (the second call starts from filled faces), from fields whose faces hold values the fills must replace."""
    im, jm, km = dims
    p = H.default_params(H.ATM, im, jm, km)
    f, g = H.make_inputs(H.ATM, im, jm, km, seed, "blobs", p)
    f = {k: f[k] for k in H.ATM_FIELDS}
    out = []
    for fused in (False, True):
        d = atom_amd.DeviceModel(H.ATM, im, jm, km, atom_amd.Params.from_buffer_copy(bytes(p)))
        d.upload_all(f, g)
        for _ in range(2):
            if fused:
                d.op("atm_bc_radius_theta_phi")
            else:
                d.op("atm_bc_radius"); d.op("atm_bc_theta"); d.op("atm_bc_phi")
        out.append({k: d.download(k) for k in H.ATM_FIELDS})
        d.close()
    changed = 0
    for k in H.ATM_FIELDS:
        if k != "RESIDUUM":
            assert np.array_equal(out[0][k].view(np.int64), out[1][k].view(np.int64)), (k, H.max_rel_err(out[1][k], out[0][k]))
            changed += int((out[0][k] != f[k]).sum())
    assert changed > 0


@pytest.mark.parametrize("dims,seed", [((7, 7, 7), 76), ((5, 8, 9), 77), ((9, 21, 70), 78), ((41, 43, 19), 79), ((41, 181, 361), 80)])
def test_hyd_bc_fills_in_one_launch_same_bits(dims, seed):
    """atomdev_hyd_bc_radius_theta_phi against atomdev_hyd_bc_radius, _bc_theta, _bc_phi: every field bit for bit, twice in
    a row, from fields whose faces hold values the fills must replace."""
    im, jm, km = dims
    p = H.default_params(H.HYD, im, jm, km)
    f, g = H.make_inputs(H.HYD, im, jm, km, seed, "blobs", p)
    f = {k: f[k] for k in H.HYD_FIELDS}
    out = []
    for fused in (False, True):
        d = atom_amd.DeviceModel(H.HYD, im, jm, km, atom_amd.Params.from_buffer_copy(bytes(p)))
        d.upload_all(f, g)
        for _ in range(2):
            if fused:
                d.op("hyd_bc_radius_theta_phi")
            else:
                d.op("hyd_bc_radius"); d.op("hyd_bc_theta"); d.op("hyd_bc_phi")
        out.append({k: d.download(k) for k in H.HYD_FIELDS})
        d.close()
    changed = 0
    for k in H.HYD_FIELDS:
        if k != "RESIDUUM":
            assert np.array_equal(out[0][k].view(np.int64), out[1][k].view(np.int64)), (k, H.max_rel_err(out[1][k], out[0][k]))
            changed += int((out[0][k] != f[k]).sum())
    assert changed > 0


@pytest.mark.parametrize("model,dims,seed", [(H.ATM, (7, 7, 7), 81), (H.HYD, (5, 8, 9), 82), (H.ATM, (9, 21, 70), 83),
                                             (H.HYD, (41, 43, 19), 84), (H.ATM, (41, 181, 361), 85), (H.HYD, (41, 181, 361), 86)])
def test_pressure_fills_in_one_launch_same_bits(model, dims, seed, monkeypatch):
    """The pressure step with its six boundary fills as two launches (the default) against the same step with the fills
    one by one (ATOMDEV_PRESSURE_FILLS_ONE_BY_ONE): p_dyn and aux_u/v/w bit for bit, faces included."""
    im, jm, km = dims
    p = H.default_params(model, im, jm, km)
    f, g = H.make_inputs(model, im, jm, km, seed, "blobs", p)
    names = H.ATM_FIELDS if model == H.ATM else H.HYD_FIELDS
    f = {k: f[k] for k in names}
    op = "atm_pressure3d" if model == H.ATM else "hyd_pressure3d"
    out = []
    for one_by_one in (True, False):
        if one_by_one:
            monkeypatch.setenv("ATOMDEV_PRESSURE_FILLS_ONE_BY_ONE", "1")
        else:
            monkeypatch.delenv("ATOMDEV_PRESSURE_FILLS_ONE_BY_ONE", raising=False)
        d = atom_amd.DeviceModel(model, im, jm, km, atom_amd.Params.from_buffer_copy(bytes(p)))
        d.upload_all(f, g)
        d.op(op); d.move_data_to_new_arrays(-1); d.op(op)
        out.append({k: d.download(k) for k in ("P_DYN", "AUX_U", "AUX_V", "AUX_W")})
        d.close()
    for k in out[0]:
        assert np.array_equal(out[0][k].view(np.int64), out[1][k].view(np.int64)), (k, H.max_rel_err(out[1][k], out[0][k]))


def test_residuum_scan_adversarial_fields():
    """The residuum scan (signed store, unsigned compare) on fields built to stress its jump logic:
    many sign changes, long runs of negative values, exact ties and zeros, magnitudes that shrink or
    grow along the sequence, and a carried-in state that nothing or everything beats.  Grid 6 x 9 x 1100:
    every level is several scan chunks, with a ragged last one."""
    im, jm, km = 6, 9, 1100
    rng = np.random.default_rng(77)
    kk = np.arange(km, dtype=np.float64)[None, None, :]
    ii = np.arange(im, dtype=np.float64)[:, None, None]
    shapes = {
        "random": lambda: rng.uniform(-1, 1, (im, jm, km)),
        "integers_with_ties": lambda: rng.integers(-3, 4, (im, jm, km)).astype(np.float64),
        "mostly_zero": lambda: np.where(rng.uniform(0, 1, (im, jm, km)) < .01, rng.uniform(-1, 1, (im, jm, km)), 0.),
        "shrinking_negative": lambda: -np.exp(-kk / 300.) * (1. + ii) * np.ones((im, jm, km)),
        "growing_alternating": lambda: (1. + kk / km) * np.where((np.arange(km) % 2) == 0, 1., -1.)[None, None, :] * np.ones((im, jm, km)),
        "all_negative_random": lambda: -rng.uniform(0, 1, (im, jm, km)),
    }
    for model, op, ge in ((H.ATM, "atm_residuum3d", False), (H.HYD, "hyd_residuum3d", True)):
        p = H.default_params(model, im, jm, km)
        f, g = H.make_inputs(model, im, jm, km, 5, "allfluid", p)
        names = H.ATM_FIELDS if model == H.ATM else H.HYD_FIELDS
        f = {k: f[k] for k in names}
        o = H.CpuModel("oracle", model, im, jm, km, p, f, g)
        d = atom_amd.DeviceModel(model, im, jm, km, atom_amd.Params.from_buffer_copy(bytes(p)))
        d.upload_all(f, g)
        for name, make in shapes.items():
            for fld in ("U", "V", "W"):
                a = make() if fld == "U" else (make() * .1 if name == "random" else np.zeros((im, jm, km)))
                o.f[fld][...] = a
                d.upload(fld, a)
            for carried in ((0.0, (0, 0, 0)), (1e30, (1, 2, 3)), (-5.0, (1, 1, 1))):
                ro = o.residuum(op, *carried)
                rd = d.residuum(op, *carried)
                assert ro == rd, (op, name, carried, ro, rd)
        o.close()
        d.close()
