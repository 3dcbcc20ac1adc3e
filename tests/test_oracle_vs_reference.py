"""atom_oracle.c against the reference's own compiled classes (oracle/_ref),
on fresh seeds, masks and grid shapes.  Runs only where the reference was
compiled (the build container); the committed goldens cover the GPU box."""
import ctypes as C

import numpy as np
import pytest

import harness as H

pytestmark = pytest.mark.skipif(not H.have_ref(), reason="oracle/_ref/libatomref.so not built")


def _pair(model, dims, seed, mask):
    im, jm, km = dims
    p = H.default_params(model, im, jm, km)
    f, g = H.make_inputs(model, im, jm, km, seed, mask, p)
    names = H.ATM_FIELDS if model == H.ATM else H.HYD_FIELDS
    f = {k: f[k] for k in names}
    return (H.CpuModel("oracle", model, im, jm, km, p, f, g),
            H.CpuModel("ref", model, im, jm, km, p, f, g), names)


@pytest.mark.parametrize("dims,seed,mask", [((9, 21, 31), 1, "blobs"), ((6, 9, 8), 2, "blobs"),
                                            ((11, 41, 31), 3, "allfluid"), ((5, 7, 7), 4, "blobs")])
def test_atm_ops(dims, seed, mask):
    a, b, names = _pair(H.ATM, dims, seed, mask)
    for op in H.ATM_SEQ + ["move2d", "atm_rk2d"]:
        H.run_op(a, op); H.run_op(b, op)
        for k in names:
            if k != "RESIDUUM":
                assert H.bits_equal(a.f[k], b.f[k]), (op, k)


@pytest.mark.parametrize("dims,seed,mask", [((9, 21, 31), 1, "blobs"), ((6, 9, 8), 2, "blobs"),
                                            ((11, 41, 31), 3, "allfluid")])
def test_hyd_ops(dims, seed, mask):
    a, b, names = _pair(H.HYD, dims, seed, mask)
    for op in H.HYD_SEQ:
        H.run_op(a, op); H.run_op(b, op)
        for k in names:
            if k != "RESIDUUM":
                assert H.bits_equal(a.f[k], b.f[k]), (op, k)


@pytest.mark.parametrize("dims,seed", [((41, 43, 19), 5), ((41, 45, 12), 6)])
def test_hyd_surface_ops(dims, seed):
    """RK_RHS_2D_Hydrosphere hard-codes im = 41 (RHS_Hyd.cpp:466), so these grids keep im = 41."""
    a, b, names = _pair(H.HYD, dims, seed, "blobs")
    for op in H.HYD_SURFACE_SEQ + ["hyd_pressure2d"]:
        H.run_op(a, op); H.run_op(b, op)
        for k in names:
            if k != "RESIDUUM":
                assert H.bits_equal(a.f[k], b.f[k]), (op, k)


def test_residuum_queries():
    a, b, _ = _pair(H.ATM, (9, 21, 31), 9, "blobs")
    assert a.residuum("atm_residuum3d") == b.residuum("atm_residuum3d")
    assert a.residuum("atm_residuum2d") == b.residuum("atm_residuum2d")
    a, b, _ = _pair(H.HYD, (9, 21, 31), 10, "blobs")
    assert a.residuum("hyd_residuum3d") == b.residuum("hyd_residuum3d")


@pytest.mark.parametrize("radiation_model,seed", [(1, 51), (0, 52)])
def test_atm_msl_data(radiation_model, seed):
    """Results_MSL_Atm::run_MSL_data prints values at fixed 1-degree locations, so the grid keeps the
    1-degree surface (181 x 361); five levels keep it cheap."""
    import ctypes as C
    im, jm, km = 5, 181, 361
    p = H.default_params(H.ATM, im, jm, km)
    f, g = H.make_msl_inputs(im, jm, km, seed, p)
    a = H.CpuModel("oracle", H.ATM, im, jm, km, p, f, g)
    b = H.CpuModel("ref", H.ATM, im, jm, km, p, f, g)
    a.op("atm_msl_data", C.c_int(radiation_model)); b.op("atm_msl_data", C.c_int(radiation_model))
    changed = 0
    for k in f:
        assert H.bits_equal(a.f[k], b.f[k]), k
        changed += not np.array_equal(a.f[k], f[k])
    for k in g:
        assert H.bits_equal(a.g[k], b.g[k]), k
        changed += not np.array_equal(a.g[k], g[k])
    assert changed >= 14  # p_stat, buoyancy, 5 clamped fields, 7-8 surface fields, evaporation, precipitation


def test_transfer_atm_to_hyd(tmp_path):
    """The decimal round trip of the atmosphere -> hydrosphere transfer file, through the reference's own
    writer and reader, against the oracle's print-and-parse restatement."""
    (pa, fa, ga), (ph, fh, gh) = H.make_transfer_inputs((5, 13, 17), (6, 13, 17), 61)
    oa, ob = H.CpuModel("oracle", H.ATM, 5, 13, 17, pa, fa, ga), H.CpuModel("oracle", H.HYD, 6, 13, 17, ph, fh, gh)
    ra, rb = H.CpuModel("ref", H.ATM, 5, 13, 17, pa, fa, ga), H.CpuModel("ref", H.HYD, 6, 13, 17, ph, fh, gh)
    oa.lib.oracle_transfer_atm_to_hyd(oa.ctx, ob.ctx)
    ra.lib.ref_transfer_atm_to_hyd(ra.ctx, rb.ctx, str(tmp_path).encode())
    for k in fh:
        assert np.array_equal(ob.f[k], rb.f[k]) and np.array_equal(np.signbit(ob.f[k]), np.signbit(rb.f[k])), k
    for k in gh:
        assert np.array_equal(ob.g[k], rb.g[k]) and np.array_equal(np.signbit(ob.g[k]), np.signbit(rb.g[k])), k
    assert not np.array_equal(ob.f["V"][5], fh["V"][5])


@pytest.mark.parametrize("dims,L_hyd", [((41, 91, 181), 1000.0), ((41, 91, 181), 6000.0), ((9, 91, 181), 1000.0),
                                        ((12, 91, 181), 2000.0), ((10, 91, 181), 1500.0)])
def test_hyd_run_data(dims, L_hyd):
    """Results_Hyd::run_data against the compiled class: Ekman layers that start at level 20 of 40, 36 of 40, 4 of 8,
    8 of 11 (odd number of intervals: Simpson refused, integrals stay 0) and 6 of 9.  The class prints the values at
    (j, k) = (90, 180) whatever the grid (Results_Hyd.cpp:344-373), so the surface has at least 91 x 181 points."""
    im, jm, km = dims
    p, f, g = H.make_run_data_inputs(im, jm, km, 91, L_hyd)
    a, b = H.CpuModel("oracle", H.HYD, im, jm, km, p, f, g), H.CpuModel("ref", H.HYD, im, jm, km, p, f, g)
    a.op("hyd_run_data"); b.op("hyd_run_data")
    for k in f:
        assert np.array_equal(a.f[k], b.f[k]) and np.array_equal(np.signbit(a.f[k]), np.signbit(b.f[k])), k
    for k in g:
        assert np.array_equal(a.g[k], b.g[k]) and np.array_equal(np.signbit(a.g[k]), np.signbit(b.g[k])), k
    for k in H.RUN_DATA_FIELDS2D + ["SALT_FINGER", "BUOYANCY"]:
        assert not np.array_equal(a.g[k] if k in g else a.f[k], g[k] if k in g else f[k]), k


@pytest.mark.parametrize("model", [H.ATM, H.HYD])
def test_plot_data_file(model, tmp_path):
    """The PlotData table of a time slice: the oracle's file is byte for byte the file of the reference's writer."""
    im, jm, km = 7, 13, 17
    if model == H.ATM:
        p = H.default_params(H.ATM, im, jm, km)
        f, g = H.make_msl_inputs(im, jm, km, 92, p)
        name = "[bridge]_PlotData_Atm.xyz"
    else:
        p, f, g = H.make_run_data_inputs(im, jm, km, 93)
        name = "[bridge]_PlotData_Hyd.xyz"
    f["V"][0 if model == H.ATM else im - 1, 3, 4] = -2e-6     # prints as -0.0000
    a, b = H.CpuModel("oracle", model, im, jm, km, p, f, g), H.CpuModel("ref", model, im, jm, km, p, f, g)
    mine = tmp_path / "mine.xyz"
    a.lib.oracle_write_plot_data.argtypes = [C.c_void_p, C.c_char_p]
    b.lib.ref_write_plot_data.argtypes = [C.c_void_p, C.c_char_p]
    assert a.lib.oracle_write_plot_data(a.ctx, str(mine).encode()) == 0
    b.lib.ref_write_plot_data(b.ctx, str(tmp_path).encode())
    theirs = (tmp_path / name).read_bytes()
    assert mine.read_bytes() == theirs
    assert theirs.count(b"\n") == 1 + jm * km and b"-0.0000" in theirs


@pytest.mark.parametrize("Ma", [0, 30])
@pytest.mark.parametrize("section,index", [(0, 0), (0, 3), (1, 5), (2, 9)])
@pytest.mark.parametrize("model", [H.ATM, H.HYD])
def test_vtk_section_file(model, section, index, Ma, tmp_path):
    """The ParaView section files (radial / longal / zonal): the oracle's file is byte for byte the file of the
    reference's writer, and the scratch the writers leave in aux_u / aux_v / aux_w and in the Celsius copy of
    temperature_NASA is the same, bits and sign of zero."""
    if model == H.HYD and Ma:
        pytest.skip("the hydrosphere writers take no Ma")
    im, jm, km = 7, 13, 17
    p, f, g = H.make_vtk_inputs(model, im, jm, km, 95)
    a, b = H.CpuModel("oracle", model, im, jm, km, p, f, g), H.CpuModel("ref", model, im, jm, km, p, f, g)
    mine = tmp_path / "mine.vtk"
    a.lib.oracle_write_vtk.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_char_p]
    b.lib.ref_write_vtk.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_char_p]
    assert a.lib.oracle_write_vtk(a.ctx, section, index, Ma, str(mine).encode()) == 0
    b.lib.ref_write_vtk(b.ctx, section, index, Ma, str(tmp_path).encode())
    name = "[bridge]_%s_%s_%d_0.vtk" % ("Atm" if model == H.ATM else "Hyd", H.VTK_SECTIONS[section], index)
    theirs = (tmp_path / name).read_bytes()
    assert mine.read_bytes() == theirs
    assert b"-0.0000" in theirs and theirs.count(b"SCALARS") >= 14
    for k in f:
        assert np.array_equal(a.f[k], b.f[k]) and np.array_equal(np.signbit(a.f[k]), np.signbit(b.f[k])), k
    for k in g:
        assert np.array_equal(a.g[k], b.g[k]) and np.array_equal(np.signbit(a.g[k]), np.signbit(b.g[k])), k
    changed = [k for k in ("AUX_U", "AUX_V", "AUX_W") if not np.array_equal(a.f[k], f[k])]
    if model == H.ATM:
        assert changed == {0: ["AUX_V", "AUX_W"] if Ma == 0 else [], 1: ["AUX_W"], 2: ["AUX_U", "AUX_V", "AUX_W"]}[section]
    else:
        assert changed == (["AUX_V"] if section == 0 else [])


def _same_bits(a, b, names):
    for k in names:
        x, y = (a.f[k], b.f[k]) if k in a.f else (a.g[k], b.g[k])
        assert np.array_equal(x, y) and np.array_equal(np.signbit(x), np.signbit(y)), k


@pytest.mark.parametrize("dims,i_beg,seed", [((9, 181, 23), 0, 71), ((12, 181, 17), 3, 72)])
def test_hyd_start_of_slice(dims, i_beg, seed, tmp_path):
    """cHydrosphereModel.cpp:248-292 in its order - BC_SeaGround, BC_Surface_Salinity_NASA, IC_v_w_EkmanSpiral,
    IC_CircumPolar_Current, BC_Temperature_Salinity, BC_Pressure_Density - oracle against the compiled classes after
    every call, bits and sign of zero.  (The latitude bands of the Ekman step are rows of the 1-degree grid: jm = 181.)"""
    im, jm, km = dims
    p, f, g = H.make_hyd_start_inputs(im, jm, km, seed)
    a, b = H.CpuModel("oracle", H.HYD, im, jm, km, p, f, g), H.CpuModel("ref", H.HYD, im, jm, km, p, f, g)
    bath, salt = str(tmp_path / "bath.xyz").encode(), str(tmp_path / "salt.xyz").encode()
    H.write_bathymetry_file(bath.decode(), jm, km, seed)
    H.write_salinity_file(salt.decode(), jm, km, seed + 1)
    names = list(f) + list(g)
    st = H.hyd_start(im, i_beg)
    for m in (a, b):
        m.lib.__getattr__(m.pre + "hyd_read_sea_ground").argtypes = [C.c_void_p, C.c_char_p]
        m.lib.__getattr__(m.pre + "hyd_read_surface_salinity").argtypes = [C.c_void_p, C.c_char_p]
    a.op("hyd_read_sea_ground", bath); b.op("hyd_read_sea_ground", bath)
    _same_bits(a, b, names)
    land = a.f["H"] == 1.0
    assert 0.2 < land.mean() < 0.9 and land[im - 1].any() and not land[im - 1].all()
    a.op("hyd_read_surface_salinity", salt); b.op("hyd_read_surface_salinity", salt)
    _same_bits(a, b, names)
    assert not np.array_equal(a.f["C"][im - 1], f["C"][im - 1])
    a.op("hyd_ic_ekman_spiral", C.c_int(i_beg)); b.op("hyd_ic_ekman_spiral", C.c_int(i_beg))
    _same_bits(a, b, names)
    assert not np.array_equal(a.f["V"][i_beg], f["V"][i_beg]) and (i_beg == 0 or np.array_equal(a.f["V"][i_beg - 1], f["V"][i_beg - 1]))
    a.op("hyd_ic_circumpolar_current", C.c_int(i_beg)); b.op("hyd_ic_circumpolar_current", C.c_int(i_beg))
    _same_bits(a, b, names)
    a.op("hyd_bc_temperature_salinity", C.byref(st)); b.op("hyd_bc_temperature_salinity", C.byref(st))
    _same_bits(a, b, names)
    a.op("hyd_pressure_density"); b.op("hyd_pressure_density")
    _same_bits(a, b, names)


def test_atm_read_mountain_surface(tmp_path):
    """BC_MountainSurface: h and Topography of the oracle's reader are those of the compiled class; the other fields
    pass through untouched."""
    a, b, names = _pair(H.ATM, (9, 21, 31), 83, "blobs")
    topo = str(tmp_path / "topo.xyz").encode()
    H.write_topography_file(topo.decode(), 21, 31, 84)
    for m in (a, b):
        m.lib.__getattr__(m.pre + "atm_read_mountain_surface").argtypes = [C.c_void_p, C.c_char_p]
    before = a.f["H"].copy()
    a.op("atm_read_mountain_surface", topo); b.op("atm_read_mountain_surface", topo)
    _same_bits(a, b, list(a.f) + list(a.g))
    assert not np.array_equal(a.f["H"], before) and 0.2 < a.f["H"][0].mean() < 0.8 and a.f["H"][4].any()


@pytest.mark.parametrize("model", [H.ATM, H.HYD])
def test_panorama_vts_file(model, tmp_path):
    """paraview_panorama_vts: the oracle's file is byte for byte the file of the reference's writer."""
    im, jm, km = 7, 13, 17
    p, f, g = H.make_vtk_inputs(model, im, jm, km, 97)
    a, b = H.CpuModel("oracle", model, im, jm, km, p, f, g), H.CpuModel("ref", model, im, jm, km, p, f, g)
    mine = tmp_path / "mine.vts"
    a.lib.oracle_write_panorama_vts.argtypes = [C.c_void_p, C.c_char_p]
    b.lib.ref_write_panorama_vts.argtypes = [C.c_void_p, C.c_char_p]
    assert a.lib.oracle_write_panorama_vts(a.ctx, str(mine).encode()) == 0
    b.lib.ref_write_panorama_vts(b.ctx, str(tmp_path).encode())
    theirs = (tmp_path / ("[bridge]_%s_panorama_0.vts" % ("Atm" if model == H.ATM else "Hyd"))).read_bytes()
    assert mine.read_bytes() == theirs
    assert b"-0.0000" in theirs and theirs.count(b"<DataArray") == (19 if model == H.ATM else 16)


@pytest.mark.parametrize("max_precipitation", [7.5, 0.0])
def test_atm_vegetation_distribution(max_precipitation):
    import ctypes as C
    a, b, _ = _pair(H.ATM, (7, 13, 17), 81, "blobs")
    for m in (a, b):
        m.f["T"][0, ::2, :] = 1.0          # exactly the threshold: counts as warm enough
        m.g["VEGETATION"] = np.full((13, 17), -1.0)
        getattr(m.lib, m.pre + "bind2d")(m.ctx, H.F2["VEGETATION"], m.g["VEGETATION"].ctypes.data_as(C.c_void_p))
        m.op("atm_vegetation_distribution", C.c_double(max_precipitation))
    assert H.bits_equal(a.g["VEGETATION"], b.g["VEGETATION"])
    assert (a.g["VEGETATION"] > 0).any() == (max_precipitation > 0)


def _minmax(m, name, two_d):
    import ctypes as C
    out = (C.c_double * 2)(); loc = (C.c_int * 6)()
    fid = H.F2[name] if two_d else H.F3[name]
    if m.kind == "oracle":
        getattr(m.lib, "oracle_minmax2d" if two_d else "oracle_minmax3d")(m.ctx, fid, out, loc)
        return (out[0], out[1]), tuple(loc)
    m.lib.ref_minmax(m.ctx, fid, 1 if two_d else 0, out)
    return (out[0], out[1]), None


def minmax_cases(f, g):
    """Field variants that reach every seed rule of the four scans."""
    shp, shp2 = f["T"].shape, g["PRECIPITATION"].shape
    rng = np.random.default_rng(5)
    cases3 = {"T": f["T"].copy(), "U": -np.abs(f["U"]) - 0.01,           # all negative: the hydrosphere maximum stays 0
              "V": np.where(rng.uniform(0, 1, shp) < 0.5, 0.25, -0.5),   # ties everywhere
              "W": np.zeros(shp)}
    cases3["C"] = f["C"].copy(); cases3["C"][0, 0, 0] = 9.0               # the seed element is the maximum
    e = rng.uniform(-1.0, 2.0, shp2); e[rng.uniform(0, 1, shp2) < 0.3] = 0.0
    cases2 = {"PRECIPITATION": rng.uniform(0.0, 25.0, shp2), "EVAPORATION_DALTON": e,
              "TOPOGRAPHY": -rng.uniform(0.5, 1.0, shp2)}
    cases2["PRECIPITATION"][0, 0] = 99.0                                   # larger than every interior value
    return cases3, cases2


@pytest.mark.parametrize("model", [H.ATM, H.HYD])
def test_minmax_values(model):
    a, b, _ = _pair(model, (7, 13, 17), 91, "blobs")
    cases3, cases2 = minmax_cases(a.f, a.g)
    for m in (a, b):
        for k, v in cases3.items():
            m.f[k][...] = v
        for k, v in cases2.items():
            m.g[k][...] = v
    for k in cases3:
        assert _minmax(a, k, False)[0] == _minmax(b, k, False)[0], k
    for k in cases2:
        assert _minmax(a, k, True)[0] == _minmax(b, k, True)[0], k


# --------------------------------------------------------------------------
# BC_Thermo (atmosphere/BC_Thermo.cpp), compiled with the headers the reference's own param.py generates
# --------------------------------------------------------------------------

def _thermo_pair(dims, p, f, g):
    im, jm, km = dims
    return H.CpuModel("oracle", H.ATM, im, jm, km, p, f, g), H.CpuModel("ref", H.ATM, im, jm, km, p, f, g)


def _all_same_bits(a, b, tag):
    for k in a.f:
        assert H.bits_equal(a.f[k], b.f[k]), (tag, k, H.max_rel_err(a.f[k], b.f[k]))
    for k in a.g:
        assert H.bits_equal(a.g[k], b.g[k]), (tag, k, H.max_rel_err(a.g[k], b.g[k]))


@pytest.mark.parametrize("dims,seed", [((7, 13, 17), 5), ((9, 21, 31), 8)])
def test_atm_value_limitation(dims, seed):
    im, jm, km = dims
    p = H.default_params(H.ATM, im, jm, km)
    f, g = H.make_inputs(H.ATM, im, jm, km, seed, "blobs", p)
    f = {k: f[k] for k in H.ATM_FIELDS}
    rng = np.random.default_rng(seed)
    for k in ["U", "V", "W", "T", "C", "CLOUD", "ICE", "CO2"]:
        f[k] = f[k] + rng.normal(0, 2.0 * np.abs(f[k]).max(), f[k].shape)  # push values past the bounds
    a, b = _thermo_pair(dims, p, f, g)
    a.op("atm_value_limitation"); b.op("atm_value_limitation")
    _all_same_bits(a, b, "value_limitation")
    assert not np.array_equal(a.f["U"], f["U"])


@pytest.mark.parametrize("dims,seed", [((9, 21, 70), 57), ((7, 13, 17), 58), ((41, 19, 23), 81)])
def test_atm_latent_heat(dims, seed):
    p, f, g = H.make_latent_heat_inputs(*dims, seed)
    f["H"][dims[0] - 2] = 0.0   # topography up to level im-2 makes the reference read level im, outside its arrays
    a, b = _thermo_pair(dims, p, f, g)
    a.op("atm_latent_heat"); b.op("atm_latent_heat")
    _all_same_bits(a, b, "latent_heat")
    assert not np.array_equal(a.f["Q_LATENT"], f["Q_LATENT"])


@pytest.mark.parametrize("dims,seed,trop", [((41, 19, 23), 59, 30), ((12, 13, 17), 60, 8), ((9, 21, 70), 61, 3),
                                            ((41, 37, 45), 76, 12)])
def test_atm_radiation_multi_layer(dims, seed, trop):
    import atom_amd
    p, f, g, i_trop = H.make_radiation_inputs(*dims, seed, trop)
    a, b = _thermo_pair(dims, p, f, g)
    r = atom_amd.Radiation(albedo_pole=.7, albedo_equator=.15, epsilon_pole=.59, epsilon_equator=.594,
                           epsilon_tropopause=.001, rad_pole=40., rad_equator=230., co2_coeff=.98 * (330. / 320.),
                           sigma=5.670280e-8)
    it = i_trop.ctypes.data_as(C.POINTER(C.c_int))
    a.lib.oracle_atm_radiation_multi_layer.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
    b.lib.ref_atm_radiation_multi_layer.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
    b.lib.ref_atm_radiation_multi_layer.restype = C.c_int
    for call in range(2):
        a.lib.oracle_atm_radiation_multi_layer(a.ctx, C.byref(r), it)
        assert b.lib.ref_atm_radiation_multi_layer(b.ctx, C.byref(r), it) == 0
        _all_same_bits(a, b, "radiation call %d" % call)
    assert not np.array_equal(a.f["T"], f["T"])


@pytest.mark.parametrize("dims,seed,first", [((41, 19, 23), 62, "warm"), ((12, 13, 17), 63, "icy"), ((9, 21, 70), 64, "warm"),
                                             ((41, 37, 45), 68, "icy")])
def test_atm_two_category_ice_scheme(dims, seed, first):
    p, f, g, rng = H.make_ice_scheme_inputs(*dims, seed, first)
    a, b = _thermo_pair(dims, p, f, g)
    dp = C.POINTER(C.c_double)
    a.lib.oracle_atm_two_category_ice_scheme.argtypes = [C.c_void_p, dp]
    b.lib.ref_atm_two_category_ice_scheme.argtypes = [C.c_void_p, dp]
    sa = (C.c_double * 2)(3.0e-7, -2.0e-7); sb = (C.c_double * 2)(3.0e-7, -2.0e-7)
    for call in range(2):
        a.lib.oracle_atm_two_category_ice_scheme(a.ctx, sa)
        b.lib.ref_atm_two_category_ice_scheme(b.ctx, sb)
        assert (sa[0], sa[1]) == (sb[0], sb[1]), call
        _all_same_bits(a, b, "ice scheme call %d" % call)
        for k in ("C", "ICE"):
            x = a.f[k] * rng.uniform(0.5, 1.5, dims)
            a.f[k][...] = x; b.f[k][...] = x
    assert (a.f["P_RAIN"] > 0).any() and (a.f["P_SNOW"] > 0).any()


@pytest.mark.parametrize("dims,seed", [((41, 19, 23), 65), ((12, 13, 17), 66), ((9, 21, 70), 67)])
def test_atm_ice_water_saturation_adjustment(dims, seed):
    p, f, g, rng = H.make_adjustment_inputs(*dims, seed)
    a, b = _thermo_pair(dims, p, f, g)
    dp = C.POINTER(C.c_double)
    a.lib.oracle_atm_ice_water_saturation_adjustment.argtypes = [C.c_void_p, dp]
    b.lib.ref_atm_ice_water_saturation_adjustment.argtypes = [C.c_void_p, dp]
    sa, sb = C.c_double(4.0e-3), C.c_double(4.0e-3)
    for call in range(2):
        a.lib.oracle_atm_ice_water_saturation_adjustment(a.ctx, C.byref(sa))
        b.lib.ref_atm_ice_water_saturation_adjustment(b.ctx, C.byref(sb))
        assert sa.value == sb.value, call
        _all_same_bits(a, b, "adjustment call %d" % call)
        for k, lo, hi in (("C", 0.7, 1.02), ("CLOUD", 0.5, 1.5), ("ICE", 0.5, 1.5)):
            x = a.f[k] * rng.uniform(lo, hi, dims) - 2e-5
            a.f[k][...] = x; b.f[k][...] = x
    assert not np.array_equal(a.f["T"], f["T"])


# ---- the atmosphere's start of a time slice ----

def _ref_curve():
    fn = H.load_ref().ref_atm_mean_temperature_from_curve
    fn.restype = C.c_float
    fn.argtypes = [C.c_float]
    return lambda Ma: fn(float(Ma))


def _start_pair(dims, seed):
    p, f, g = H.make_atm_start_inputs(*dims, seed)
    return p, f, g, _thermo_pair(dims, p, f, g)


def _call(m, name, *args):
    fn = getattr(m.lib, m.pre + name)
    fn.restype = C.c_int
    return fn(m.ctx, *args)


@pytest.mark.parametrize("dims,seed,trop", [((41, 181, 5), 91, 30), ((41, 181, 3), 92, 24), ((17, 181, 4), 93, 12)])
def test_atm_ic_cell_structure(dims, seed, trop):
    im, jm, km = dims
    p, f, g, (o, r) = _start_pair(dims, seed)
    i_trop = np.full(jm, trop, dtype=np.int32)
    i_trop[60:120] = min(trop + 3, im - 2)      # a higher tropopause between the tropics
    for m in (o, r):
        # the reference's caller passes the reset values (no wind); wind on entry shows what is overwritten and what is kept
        assert _call(m, "atm_ic_cell_structure", i_trop.ctypes.data_as(C.POINTER(C.c_int))) == 0
    _all_same_bits(o, r, "IC_CellStructure")
    assert np.abs(o.f["U"]).max() > 0 and np.abs(o.f["V"]).max() > 0 and np.abs(o.f["W"]).max() > 0


def test_atm_surface_files(tmp_path):
    dims = (9, 21, 31)
    p, f, g, (o, r) = _start_pair(dims, 94)
    rng = np.random.default_rng(94)
    temp, prec = tmp_path / "t.xyz", tmp_path / "p.xyz"
    H.write_lat_lon_value_file(str(temp), rng.uniform(-40., 32., dims[1:]), "%.2f")
    H.write_lat_lon_value_file(str(prec), rng.uniform(0., 12., dims[1:]), "%.3f")
    for m in (o, r):
        assert _call(m, "atm_read_surface_temperature", str(temp).encode()) == 0
        assert _call(m, "atm_read_surface_precipitation", str(prec).encode()) == 0
    _all_same_bits(o, r, "NASA files")
    assert not np.array_equal(o.g["PRECIPITATION_NASA"], g["PRECIPITATION_NASA"])


def test_atm_surface_files_of_the_reference():
    """The reference's own 1-degree files."""
    dims = (3, 181, 361)
    p, f, g, (o, r) = _start_pair(dims, 95)
    for m in (o, r):
        assert _call(m, "atm_read_surface_temperature", b"/root/reference/data/SurfaceTemperature_NASA.xyz") == 0
        assert _call(m, "atm_read_surface_precipitation", b"/root/reference/data/SurfacePrecipitation_NASA.xyz") == 0
    _all_same_bits(o, r, "NASA files")
    assert o.g["TEMPERATURE_NASA"].min() > 0.7 and o.g["PRECIPITATION_NASA"].max() > 1.0


START_CASES = [dict(Ma=0, prev=None), dict(Ma=10, prev=0), dict(Ma=10, prev=0, NASATemperature=0),
               dict(Ma=50, prev=40, sun=1), dict(Ma=100, prev=90, RadiationModel=0), dict(Ma=140, prev=130, t_land=0.003661),
               dict(Ma=0, prev=None, NASATemperature=0, t_land=0.003661)]


def _start_struct(case, jm, trop):
    kw = {k: v for k, v in case.items() if k not in ("Ma", "prev")}
    s, keep = H.atm_start(jm, trop, **kw)
    H.set_paleo_time(s, case["Ma"], case["prev"], _ref_curve())
    return s, keep


@pytest.mark.parametrize("case", START_CASES)
@pytest.mark.parametrize("dims,seed,trop", [((41, 19, 25), 96, 30), ((12, 13, 27), 97, 8)])
def test_atm_bc_temperature(dims, seed, trop, case):
    p, f, g, (o, r) = _start_pair(dims, seed)
    i_trop = np.full(dims[1], trop, dtype=np.int32)
    i_trop[::4] -= 2
    s, keep = _start_struct(case, dims[1], i_trop)
    for m in (o, r):
        m.op("atm_bc_temperature", C.byref(s))
    _all_same_bits(o, r, "BC_Temperature %r" % (case,))
    assert not np.array_equal(o.f["T"], f["T"])


@pytest.mark.parametrize("dims,seed", [((5, 19, 25), 98), ((4, 33, 41), 99)])
def test_atm_ic_temperature_west_east_coast(dims, seed):
    p, f, g, (o, r) = _start_pair(dims, seed)
    for m in (o, r):
        m.op("atm_ic_temperature_west_east_coast")
    _all_same_bits(o, r, "IC_Temperature_WestEastCoast")
    assert not np.array_equal(o.f["T"], f["T"])


@pytest.mark.parametrize("dims,seed", [((41, 19, 25), 100), ((12, 13, 27), 101)])
def test_atm_bc_pressure(dims, seed):
    p, f, g, (o, r) = _start_pair(dims, seed)
    for m in (o, r):
        m.op("atm_bc_pressure")
    _all_same_bits(o, r, "BC_Pressure")
    assert o.f["P_STAT"].min() > 50.


@pytest.mark.parametrize("case", [dict(Ma=0, prev=None), dict(Ma=100, prev=90), dict(Ma=140, prev=130, c_ocean=.78, c_land=.84)])
@pytest.mark.parametrize("dims,seed,trop", [((41, 19, 25), 102, 30), ((12, 13, 27), 103, 8)])
def test_atm_bc_water_vapour(dims, seed, trop, case):
    p, f, g, (o, r) = _start_pair(dims, seed)
    s, keep = _start_struct(case, dims[1], trop)
    for m in (o, r):
        m.op("atm_bc_temperature", C.byref(s))
        m.op("atm_bc_water_vapour", C.byref(s))
    _all_same_bits(o, r, "BC_WaterVapour %r" % (case,))
    assert o.f["C"].max() > 0.003


@pytest.mark.parametrize("case", [dict(Ma=0, prev=None), dict(Ma=100, prev=90), dict(Ma=140, prev=130, co2_ocean=2., co2_land=1.)])
@pytest.mark.parametrize("dims,seed,trop", [((41, 19, 25), 104, 30), ((12, 13, 27), 105, 8)])
def test_atm_bc_co2(dims, seed, trop, case):
    p, f, g, (o, r) = _start_pair(dims, seed)
    s, keep = _start_struct(case, dims[1], trop)
    for m in (o, r):
        m.op("atm_bc_co2", C.byref(s))
    _all_same_bits(o, r, "BC_CO2 %r" % (case,))
    assert o.f["CO2"].max() > 1.1
