"""The reference's 3-D iteration loops run operator by operator, several iterations deep: the oracle against the
compiled reference classes (CPU, where oracle/_ref exists) and the HIP path against the oracle (GPU).  What a
single-operator test cannot show is that nothing drifts when the operators feed each other."""
import ctypes as C

import numpy as np
import pytest

import atom_amd
import harness as H


def _same_queries(a, b, exact):
    assert len(a) == len(b)
    for x, y in zip(a, b):
        if isinstance(x[1], tuple):  # residuum: (min, loc)
            assert x[1] == y[1], (x, y)
            assert x[0] == y[0] if exact else abs(x[0] - y[0]) <= 1e-10 * max(1.0, abs(y[0])), (x, y)
        else:  # steady: (values, locations)
            assert np.array_equal(x[1], y[1]), (x, y)
            assert np.array_equal(x[0], y[0]) if exact else np.allclose(x[0], y[0], rtol=1e-10, atol=0), (x, y)


def _hyd_inputs(dims, seed):
    im, jm, km = dims
    p = H.default_params(H.HYD, im, jm, km)
    f, g = H.make_inputs(H.HYD, im, jm, km, seed, "blobs", p)
    return p, {k: f[k] for k in H.HYD_FIELDS}, g


def _ref_steady_missing():
    return not hasattr(H.load_ref(), "ref_hyd_steady3d")


@pytest.mark.skipif(not H.have_ref(), reason="oracle/_ref/libatomref.so not built")
def test_hyd_loop_oracle_vs_reference():
    """steadyQuery results are printed only by the reference, so the loop is compared through its fields."""
    dims = (9, 21, 31)
    p, f, g = _hyd_inputs(dims, 71)
    a = H.CpuModel("oracle", H.HYD, *dims, p, f, g)
    b = H.CpuModel("ref", H.HYD, *dims, p, f, g)
    b.steady = lambda name, n: (np.zeros(n), np.zeros((n, 3), dtype=int))
    la, lb = H.hyd_3d_loop(a), H.hyd_3d_loop(b)
    for x, y in zip(la, lb):
        if isinstance(x[1], tuple):
            assert x == y
    for k in H.HYD_FIELDS:
        if k != "RESIDUUM":
            assert H.bits_equal(a.f[k], b.f[k]), k


@pytest.mark.gpu
@pytest.mark.parametrize("dims,seed", [((9, 21, 70), 72), ((41, 43, 19), 73)])
def test_hyd_loop_device_bitexact(dims, seed):
    """Every operator of the hydrosphere's velocity iteration on the device, Results_Hyd::run_data included."""
    p, f, g = _hyd_inputs(dims, seed)
    for k in H.RUN_DATA_FIELDS2D:
        g[k] = np.zeros(dims[1:])
    o = H.CpuModel("oracle", H.HYD, *dims, p, f, g)
    d = atom_amd.DeviceModel(H.HYD, *dims, atom_amd.Params.from_buffer_copy(bytes(p)))
    d.upload_all(f, g)
    d.set_rk_variant(0)   # the exact form of the sweep
    _same_queries(H.hyd_3d_loop(d, run_data=True), H.hyd_3d_loop(o, run_data=True), exact=True)
    for k in H.HYD_FIELDS:
        if k != "RESIDUUM":
            assert H.bits_equal(d.download(k), o.f[k]), k
    for k in H.RUN_DATA_FIELDS2D:
        assert H.bits_equal(d.download(k), o.g[k]), k
    d.close()


@pytest.mark.gpu
def test_atm_loop_device():
    """Four iterations of the atmosphere loop (pow in the sweep, exp in run_MSL_data): 1e-10 per field."""
    dims = (9, 21, 70)
    im, jm, km = dims
    p = H.default_params(H.ATM, im, jm, km)
    f, g = H.make_msl_inputs(im, jm, km, 74, p)
    o = H.CpuModel("oracle", H.ATM, im, jm, km, p, f, g)
    d = atom_amd.DeviceModel(H.ATM, im, jm, km, atom_amd.Params.from_buffer_copy(bytes(p)))
    d.upload_all(f, g)
    _same_queries(H.atm_3d_loop(d), H.atm_3d_loop(o), exact=False)
    worst = 0.0
    for k in list(f) + list(g):
        if k == "RESIDUUM":
            continue
        e = H.max_rel_err(d.download(k), o.f[k] if k in f else o.g[k])
        worst = max(worst, e)
        assert e <= 1e-10, (k, e)
    print("atm loop, 4 iterations: max rel err %.3e" % worst)
    d.close()


def physics_loop_inputs(im, jm, km, seed):
    """A state the column physics keeps bounded: moisture at most 5 % above saturation over water."""
    p = H.default_params(H.ATM, im, jm, km)
    f, g = H.make_msl_inputs(im, jm, km, seed, p)
    rng = np.random.default_rng(seed)
    shp = (im, jm, km)
    lev = np.arange(im).reshape(im, 1, 1) / (im - 1)
    f["T"] = 1.06 - 0.25 * lev + rng.uniform(-0.01, 0.01, shp)
    f["TN"] = f["T"].copy()
    t_u = f["T"] * p.t_0
    p_h = ((t_u - p.gam * lev * p.L_atm * 1e-2) / t_u) ** (p.g / (1e-2 * p.gam * p.R_Air)) * .01 * p.r_air * p.R_Air * t_u[0]
    e_rain = p.hp * np.exp(17.2694 * (t_u - 273.15) / (t_u - 35.86))
    f["C"] = np.minimum(p.ep * e_rain / (p_h - e_rain) * rng.uniform(0.3, 1.05, shp), 0.02)
    f["CN"] = f["C"].copy()
    for k, hi in (("CLOUD", 2e-4), ("ICE", 1e-4)):
        f[k] = np.where(rng.random(shp) < 0.5, 0.0, rng.uniform(0.0, hi, shp)); f[k + "N"] = f[k].copy()
    f["S_C_C"] = np.zeros(shp)
    f["EPSILON_3D"] = rng.uniform(0.3, 0.9, shp)
    for k in ("ALBEDO", "EPSILON", "RADIATION_SURFACE", "VEGETATION"):
        g[k] = rng.uniform(0.0, 1.0, (jm, km))
    i_trop = np.full(jm, im - 3, dtype=np.int32)
    return p, f, g, i_trop


@pytest.mark.gpu
def test_atm_loop_with_column_physics_device():
    """The whole velocity iteration of cAtmosphereModel.cpp:756-874 on the device, the BC_Thermo column physics
    included (saturation adjustment, radiation, latent heat, vegetation, ice scheme, compared with the oracle's
    restatements, which the compiled BC_Thermo class pins bit for bit): no operator of the loop is left to the host.  One pressure iteration of two
    velocity iterations (each operator at least once), 1e-10 per field, on a state the physics keeps bounded.
    Longer loops are not held to 1e-10: the saturation adjustment iterates to a 1e-5 exit test and, on the oracle
    itself, returns a 1e-12 relative change of t as 2.5e-10 in c - every further call multiplies what the
    libm difference of the call before left (measured 1.6e-12 after one call, 9.8e-11 after two)."""
    im, jm, km = 12, 21, 34
    p, f, g, i_trop = physics_loop_inputs(im, jm, km, 75)
    o = H.CpuModel("oracle", H.ATM, im, jm, km, p, f, g)
    d = atom_amd.DeviceModel(H.ATM, im, jm, km, atom_amd.Params.from_buffer_copy(bytes(p)))
    d.upload_all(f, g)
    _same_queries(H.atm_3d_loop(d, 1, 2, physics=True, i_trop=i_trop), H.atm_3d_loop(o, 1, 2, physics=True, i_trop=i_trop),
                  exact=False)
    worst, errs = 0.0, {}
    for k in list(f) + list(g):
        if k == "RESIDUUM":
            continue
        ref = o.f[k] if k in f else o.g[k]
        assert np.isfinite(ref).all(), k
        e = H.max_rel_err(d.download(k), ref)
        errs[k] = e
        worst = max(worst, e)
    print(sorted(errs.items(), key=lambda kv: -kv[1])[:6])
    for k, e in errs.items():
        assert e <= 1e-10, (k, e)
    for k in ("P_RAIN", "S_V", "Q_LATENT", "RADIATION_3D", "S_C_C"):
        assert not np.array_equal(o.f[k], f[k]), k
    print("atm loop with column physics, 2 iterations: max rel err %.3e" % worst)
    d.close()


def _hyd_slice(m, start, bath, salt):
    H.hyd_start_of_slice(m, start, bath, salt)
    H.hyd_2d_loop(m)
    return H.hyd_3d_loop(m, run_data=True)


def _hyd_slice_inputs(tmp_path, km, seed):
    im, jm = 41, 181          # the surface RHS hard-codes im = 41 (RHS_Hyd.cpp:466), the Ekman bands the rows of 1 degree
    p, f, g = H.make_hyd_start_inputs(im, jm, km, seed)
    rng = np.random.default_rng(seed)
    for k in H.RUN_DATA_FIELDS2D:
        g[k] = rng.uniform(-7.0, 7.0, (jm, km))
    bath, salt = str(tmp_path / "bath.xyz"), str(tmp_path / "salt.xyz")
    H.write_bathymetry_file(bath, jm, km, seed)
    H.write_salinity_file(salt, jm, km, seed + 1)
    return (im, jm, km), p, f, g, bath, salt


@pytest.mark.skipif(not H.have_ref(), reason="oracle/_ref/libatomref.so not built")
def test_hyd_time_slice_oracle_vs_reference(tmp_path):
    """A hydrosphere time slice as cHydrosphereModel::RunTimeSlice runs it (cHydrosphereModel.cpp:248-610) - readers,
    initial conditions, 2 x 2 surface iterations, 2 x 2 volume iterations with Results_Hyd::run_data - operator by
    operator through the oracle and through the compiled classes: every field and every query result, same bits."""
    dims, p, f, g, bath, salt = _hyd_slice_inputs(tmp_path, 185, 81)
    a, b = H.CpuModel("oracle", H.HYD, *dims, p, f, g), H.CpuModel("ref", H.HYD, *dims, p, f, g)
    st = H.hyd_start(dims[0])
    b.steady = lambda name, n: (np.zeros(n), np.zeros((n, 3), dtype=int))  # printed only by the reference
    for x, y in zip(_hyd_slice(a, st, bath, salt), _hyd_slice(b, st, bath, salt)):
        if isinstance(x[1], tuple):
            assert x == y
    for k in f:
        if k != "RESIDUUM":
            assert H.bits_equal(a.f[k], b.f[k]), k
    for k in g:
        assert H.bits_equal(a.g[k], b.g[k]), k
    assert not np.array_equal(a.f["T"], f["T"]) and not np.array_equal(a.g["UPWELLING"], g["UPWELLING"])


@pytest.mark.gpu
def test_hyd_time_slice_device(tmp_path):
    """The same time slice with every operator on the device, from the bathymetry file to the last pressure step: no
    field is uploaded after the transfer-file state.  The Ekman step's asin / cos / sin are the device library's, so
    the slice is held to the stated 1e-10 relative per field (iteration counts are fixed, query locations identical)."""
    dims, p, f, g, bath, salt = _hyd_slice_inputs(tmp_path, 185, 81)
    o = H.CpuModel("oracle", H.HYD, *dims, p, f, g)
    d = atom_amd.DeviceModel(H.HYD, *dims, atom_amd.Params.from_buffer_copy(bytes(p)))
    d.upload_all(f, g)
    st = H.hyd_start(dims[0])
    _same_queries(_hyd_slice(d, atom_amd.HydStart.from_buffer_copy(bytes(st)), bath, salt), _hyd_slice(o, st, bath, salt), exact=False)
    worst = 0.0
    for k in list(f) + list(g):
        if k == "RESIDUUM":
            continue
        e = H.max_rel_err(d.download(k), o.f[k] if k in f else o.g[k])
        worst = max(worst, e)
        assert e <= 1e-10, (k, e)
    print("hydrosphere time slice on the device: max rel err %.3e" % worst)
    d.close()


@pytest.mark.gpu
def test_hyd_time_slice_device_bitexact_after_ekman(tmp_path):
    """The same slice with the sweep in its exact form and the Ekman step's v and w taken from the oracle (the only operator whose bits may differ):
    every field of the surface loop and of the volume loop must then carry the oracle's bits - level im-2 of the volume
    sweep included, which takes its salinity rows from another workgroup's stores while both are running."""
    dims, p, f, g, bath, salt = _hyd_slice_inputs(tmp_path, 185, 82)
    o = H.CpuModel("oracle", H.HYD, *dims, p, f, g)
    d = atom_amd.DeviceModel(H.HYD, *dims, atom_amd.Params.from_buffer_copy(bytes(p)))
    d.upload_all(f, g)
    d.set_rk_variant(0)   # the exact form of the sweep
    st = H.hyd_start(dims[0])
    H.hyd_start_of_slice(o, st, bath, salt)
    H.hyd_start_of_slice(d, atom_amd.HydStart.from_buffer_copy(bytes(st)), bath, salt)
    for k in ("V", "W", "VN", "WN"):
        d.upload(k, o.f[k])
    for m in (o, d):
        H.hyd_2d_loop(m)
    _same_queries(H.hyd_3d_loop(d, 3, 2, run_data=True), H.hyd_3d_loop(o, 3, 2, run_data=True), exact=True)
    for k in list(f) + list(g):
        if k != "RESIDUUM":
            assert H.bits_equal(d.download(k), o.f[k] if k in f else o.g[k]), k
    d.close()


@pytest.mark.gpu
def test_hyd_time_slice_one_degree_reference_iteration_counts(tmp_path):
    """The hydrosphere run at 0 Ma on the 1-degree grid: 0 Ma bathymetry, the reference's iteration counts (10 x 2
    surface, 2 x 2 volume iterations; param.py:135-138), every operator on the device.  Every field within the stated
    1e-10 relative after the slice (measured 1e-15), query results and their locations alike, everything finite."""
    dims, p, f, g = H.make_hyd_slice_inputs_1deg(91)
    bath, salt = str(tmp_path / "bath.xyz"), str(tmp_path / "salt.xyz")
    H.write_bathymetry_file_0Ma(bath)
    H.write_salinity_file(salt, dims[1], dims[2], 92)
    o = H.CpuModel("oracle", H.HYD, *dims, p, f, g)
    d = atom_amd.DeviceModel(H.HYD, *dims, atom_amd.Params.from_buffer_copy(bytes(p)))
    d.upload_all(f, g)
    st = H.hyd_start(dims[0])
    logs = []
    for m, start in ((d, atom_amd.HydStart.from_buffer_copy(bytes(st))), (o, st)):
        H.hyd_start_of_slice(m, start, bath, salt)
        H.hyd_2d_loop(m, 10, 2)
        logs.append(H.hyd_3d_loop(m, 2, 2, run_data=True))
    _same_queries(logs[0], logs[1], exact=False)
    h_mask, _ = H.load_mask_0Ma(H.HYD)
    # the file written from the golden bathymetry reads back as the golden mask, up to a dozen needles the first
    # reading had removed from the solid cells but not from the depths
    assert np.count_nonzero(o.f["H"] != h_mask) < 100
    worst = 0.0
    for k in list(f) + list(g):
        if k == "RESIDUUM":
            continue
        ref = o.f[k] if k in f else o.g[k]
        assert np.isfinite(ref).all(), k
        e = H.max_rel_err(d.download(k), ref)
        worst = max(worst, e)
        assert e <= 1e-10, (k, e)
    print("hydrosphere 1-degree time slice on the device: max rel err %.3e" % worst)
    d.close()


@pytest.mark.gpu
def test_hyd_run_time_slice_one_call_same_bits(tmp_path):
    """atomdev_hyd_run_time_slice - the whole slice in one call - leaves every field with the bits of the same
    operators called one by one (which the tests above hold against the oracle and the compiled reference), and
    returns the last pair of residuum queries."""
    dims, p, f, g, bath, salt = _hyd_slice_inputs(tmp_path, 185, 83)
    a = atom_amd.DeviceModel(H.HYD, *dims, atom_amd.Params.from_buffer_copy(bytes(p)))
    b = atom_amd.DeviceModel(H.HYD, *dims, atom_amd.Params.from_buffer_copy(bytes(p)))
    a.upload_all(f, g); b.upload_all(f, g)
    st = atom_amd.HydStart.from_buffer_copy(bytes(H.hyd_start(dims[0])))
    H.hyd_start_of_slice(a, st, bath, salt)
    H.hyd_2d_loop(a, 3, 2)
    log = H.hyd_3d_loop(a, 2, 2, run_data=True)
    before, after, loc = b.run_time_slice(bath, salt, pressure_iter_max_2D=3)
    for k in list(f) + list(g):
        x, y = a.download(k), b.download(k)
        assert np.array_equal(x, y, equal_nan=True) and np.array_equal(np.signbit(x), np.signbit(y)), k
    assert (before, (after, loc)) == (log[-3][0], log[-2])
    b.run_time_slice(bath, salt, pressure_iter_max_2D=3, queries=False)
    with pytest.raises(atom_amd.AtomDevError):
        b.run_time_slice(str(tmp_path / "missing.xyz"))
    a.close(); b.close()


@pytest.mark.gpu
def test_hyd_run_time_slices_lockstep_same_bits(tmp_path):
    """Three time slices (different fields, bathymetries and Ma) advanced in lockstep by one call - their surface sweeps
    share a launch, their other operators overlap on their streams - end with the bits of three calls of their own."""
    runs = []
    for n, (seed, Ma) in enumerate(((84, 0), (85, 30), (86, 60))):
        sub = tmp_path / str(n)
        sub.mkdir()
        dims, p, f, g, bath, salt = _hyd_slice_inputs(sub, 185, seed)
        runs.append((dims, p, f, g, bath, salt, Ma))
    def models():
        out = []
        for dims, p, f, g, *_ in runs:
            d = atom_amd.DeviceModel(H.HYD, *dims, atom_amd.Params.from_buffer_copy(bytes(p)))
            d.upload_all(f, g)
            out.append(d)
        return out
    alone, together = models(), models()
    single = [d.run_time_slice(r[4], r[5], Ma=r[6], pressure_iter_max_2D=2) for d, r in zip(alone, runs)]
    batch = atom_amd.hyd_run_time_slices(together, [(r[4], r[5], None, r[6]) for r in runs], pressure_iter_max_2D=2)
    assert [tuple(x) for x in batch] == [tuple(x) for x in single]
    for a, b, r in zip(alone, together, runs):
        for k in list(r[2]) + list(r[3]):
            x, y = a.download(k), b.download(k)
            assert np.array_equal(x, y, equal_nan=True) and np.array_equal(np.signbit(x), np.signbit(y)), k
        a.close(); b.close()


@pytest.mark.gpu
def test_atm_run_loops_one_call_same_bits():
    """atomdev_atm_run_2d_loop and atomdev_atm_run_3d_loop (column physics included) leave every field with the bits of
    the same operators called one by one, which the tests above hold against the oracle; the carried members of
    BC_Thermo and the last residuum pair come back in the state."""
    im, jm, km = 12, 21, 34
    p, f, g, i_trop = physics_loop_inputs(im, jm, km, 76)
    a = atom_amd.DeviceModel(H.ATM, im, jm, km, atom_amd.Params.from_buffer_copy(bytes(p)))
    b = atom_amd.DeviceModel(H.ATM, im, jm, km, atom_amd.Params.from_buffer_copy(bytes(p)))
    a.upload_all(f, g); b.upload_all(f, g)
    # surface loop, operator by operator (cAtmosphereModel.cpp:685-728)
    for _ in range(2):
        for _ in range(2):
            a.op("atm_bc_theta"); a.op("atm_bc_phi")
            first = a.residuum("atm_residuum2d")
            for op in ("atm_value_limitation", "atm_solid_ground", "atm_rk2d"):
                a.op(op)
            a.residuum("atm_residuum2d", *first)
            a.steady("atm_steady2d", 3)
            a.move_data_to_new_arrays(0)
        a.op("atm_pressure2d")
    log = H.atm_3d_loop(a, 1, 2, physics=True, i_trop=i_trop)
    r = atom_amd.Radiation(albedo_pole=.7, albedo_equator=.15, epsilon_pole=.59, epsilon_equator=.594, epsilon_tropopause=.001,
                           rad_pole=40., rad_equator=230., co2_coeff=.98 * (330. / 320.), sigma=5.670280e-8)
    L = b.atm_loop(i_trop, r, pressure_iter_max_2D=2, pressure_iter_max=1)
    b.run_2d_loop(L)
    st = b.run_3d_loop(L)
    for k in list(f) + list(g):
        x, y = a.download(k), b.download(k)
        assert np.array_equal(x, y, equal_nan=True) and np.array_equal(np.signbit(x), np.signbit(y)), k
    assert (st.q_v_hyp, st.S_nuc, st.S_i_dep) == (log[-3][0], log[-2][0], log[-1][0])
    assert (st.residuum[1], tuple(st.residuum_loc)) == log[-5]
    a.close(); b.close()


@pytest.mark.gpu
def test_atm_run_2d_loops_lockstep_same_bits():
    """The surface loops of three atmosphere slices in lockstep (one launch per sweep for all of them): the bits of
    three loops of their own."""
    im, jm, km = 9, 21, 70
    alone, together = [], []
    for seed in (77, 78, 79):
        p = H.default_params(H.ATM, im, jm, km)
        f, g = H.make_msl_inputs(im, jm, km, seed, p)
        for group in (alone, together):
            d = atom_amd.DeviceModel(H.ATM, im, jm, km, atom_amd.Params.from_buffer_copy(bytes(p)))
            d.upload_all(f, g)
            group.append(d)
    L = alone[0].atm_loop(pressure_iter_max_2D=3)
    for d in alone:
        d.run_2d_loop(L)
    atom_amd.atm_run_2d_loops(together, L)
    for a, b in zip(alone, together):
        a.sync(); b.sync()
        for k in ("V", "W", "P_DYN", "VN", "WN", "P_DYNN", "AUX_V", "AUX_W", "T", "C"):
            x, y = a.download(k), b.download(k)
            assert np.array_equal(x, y, equal_nan=True) and np.array_equal(np.signbit(x), np.signbit(y)), k
        a.close(); b.close()
