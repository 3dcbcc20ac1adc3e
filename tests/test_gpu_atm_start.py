"""Parity of the atmosphere's start of a time slice on the device (BC_Thermo's initial conditions, through the
C-ABI) with the oracle, which the compiled BC_Thermo class pins bit for bit (test_oracle_vs_reference.py).

Bars: IC_CellStructure, the surface files, BC_Temperature, IC_Temperature_WestEastCoast and BC_CO2 are +, -, *, /
only and must be BIT-EXACT; BC_Pressure (pow) and BC_WaterVapour (exp, sqrt) go through libm on both sides and are held
to 1e-10 relative per field (max |a-b| / max |b|).
"""
import ctypes as C

import numpy as np
import pytest

import atom_amd
import harness as H

pytestmark = pytest.mark.gpu

TOL = 1e-10


def curve(Ma):
    """A stand-in for the model's temperature curve (the reference's data file does not travel)."""
    return 15. + 0.0371 * Ma - 1.1e-4 * Ma * Ma


def _pair(dims, seed, real_mask=False):
    p, f, g = H.make_atm_start_inputs(*dims, seed)
    if real_mask:
        f["H"], g["TOPOGRAPHY"] = H.load_mask_0Ma(H.ATM)
        g["VEGETATION"] = g["VEGETATION"] * 0 + np.random.default_rng(seed).uniform(0, 1, dims[1:]) * (f["H"][0] > 0)
    o = H.CpuModel("oracle", H.ATM, *dims, p, f, g)
    d = atom_amd.DeviceModel(H.ATM, *dims, atom_amd.Params.from_buffer_copy(bytes(p)))
    d.upload_all(f, g)
    return p, f, g, o, d


def _compare(o, d, tag, exact=True):
    worst = 0.0
    for k in list(o.f) + list(o.g):
        ref = o.f[k] if k in o.f else o.g[k]
        got = d.download(k)
        if exact:
            assert H.bits_equal(got, ref), (tag, k, H.max_rel_err(got, ref))
        else:
            e = H.max_rel_err(got, ref)
            worst = max(worst, e)
            assert e <= TOL, (tag, k, e)
    return worst


def _start(case, jm, trop):
    kw = {k: v for k, v in case.items() if k not in ("Ma", "prev")}
    s, keep = H.atm_start(jm, trop, **kw)
    H.set_paleo_time(s, case["Ma"], case["prev"], curve)
    return s, keep


@pytest.mark.parametrize("dims,seed,trop", [((41, 181, 70), 91, 30), ((41, 181, 9), 92, 24), ((17, 181, 65), 93, 12)])
def test_atm_ic_cell_structure_bitexact(dims, seed, trop):
    im, jm, km = dims
    p, f, g, o, d = _pair(dims, seed)
    i_trop = np.full(jm, trop, dtype=np.int32)
    i_trop[60:120] = min(trop + 3, im - 2)
    assert o.lib.oracle_atm_ic_cell_structure(o.ctx, i_trop.ctypes.data_as(C.POINTER(C.c_int))) == 0
    d.ic_cell_structure(i_trop)
    _compare(o, d, "IC_CellStructure")
    for k in "UVW":
        assert not np.array_equal(o.f[k], f[k])
    d.close()


def test_atm_ic_cell_structure_needs_the_one_degree_rows():
    d = atom_amd.DeviceModel(H.ATM, 9, 21, 31)
    with pytest.raises(atom_amd.AtomDevError, match="181"):
        d.ic_cell_structure(np.full(21, 5, dtype=np.int32))
    d.close()


def test_atm_surface_files_bitexact(tmp_path):
    dims = (9, 21, 31)
    p, f, g, o, d = _pair(dims, 94)
    rng = np.random.default_rng(94)
    temp, prec = tmp_path / "t.xyz", tmp_path / "p.xyz"
    H.write_lat_lon_value_file(str(temp), rng.uniform(-40., 32., dims[1:]), "%.2f")
    H.write_lat_lon_value_file(str(prec), rng.uniform(0., 12., dims[1:]), "%.3f")
    assert o.lib.oracle_atm_read_surface_temperature(o.ctx, str(temp).encode()) == 0
    assert o.lib.oracle_atm_read_surface_precipitation(o.ctx, str(prec).encode()) == 0
    d.read_surface_temperature(str(temp))
    d.read_surface_precipitation(str(prec))
    _compare(o, d, "surface files")
    assert not np.array_equal(o.g["PRECIPITATION_NASA"], g["PRECIPITATION_NASA"])
    with pytest.raises(atom_amd.AtomDevError, match="could not open"):
        d.read_surface_temperature(str(tmp_path / "absent.xyz"))
    d.close()


START_CASES = [dict(Ma=0, prev=None), dict(Ma=10, prev=0), dict(Ma=10, prev=0, NASATemperature=0),
               dict(Ma=50, prev=40, sun=1), dict(Ma=50, prev=40, sun=1, sun_position_lat=4, sun_position_lon=12),
               dict(Ma=100, prev=90, RadiationModel=0), dict(Ma=140, prev=130, t_land=0.003661),
               dict(Ma=0, prev=None, NASATemperature=0, t_land=0.003661)]


@pytest.mark.parametrize("case", START_CASES)
@pytest.mark.parametrize("dims,seed,trop", [((41, 19, 70), 96, 30), ((12, 13, 27), 97, 8)])
def test_atm_bc_temperature_bitexact(dims, seed, trop, case):
    p, f, g, o, d = _pair(dims, seed)
    i_trop = np.full(dims[1], trop, dtype=np.int32)
    i_trop[::4] -= 2
    s, keep = _start(case, dims[1], i_trop)
    o.op("atm_bc_temperature", C.byref(s))
    d.bc_temperature(s)
    _compare(o, d, "BC_Temperature %r" % (case,))
    assert not np.array_equal(o.f["T"], f["T"])
    d.close()


@pytest.mark.parametrize("dims,seed", [((5, 19, 25), 98), ((5, 33, 141), 99), ((6, 140, 70), 106)])
def test_atm_ic_temperature_west_east_coast_bitexact(dims, seed):
    p, f, g, o, d = _pair(dims, seed)
    o.op("atm_ic_temperature_west_east_coast")
    d.ic_temperature_west_east_coast()
    _compare(o, d, "IC_Temperature_WestEastCoast")
    assert not np.array_equal(o.f["T"], f["T"])
    d.close()


@pytest.mark.parametrize("dims,seed", [((41, 19, 70), 100), ((12, 13, 27), 101)])
def test_atm_bc_pressure(dims, seed):
    p, f, g, o, d = _pair(dims, seed)
    o.op("atm_bc_pressure")
    d.bc_pressure()
    print("BC_Pressure: max rel err %.3e" % _compare(o, d, "BC_Pressure", exact=False))
    assert not np.array_equal(o.f["P_STAT"], f["P_STAT"])
    d.close()


@pytest.mark.parametrize("case", [dict(Ma=0, prev=None), dict(Ma=100, prev=90), dict(Ma=140, prev=130, c_ocean=.78, c_land=.84)])
@pytest.mark.parametrize("dims,seed,trop", [((41, 19, 70), 102, 30), ((12, 13, 27), 103, 8)])
def test_atm_bc_water_vapour(dims, seed, trop, case):
    p, f, g, o, d = _pair(dims, seed)
    s, keep = _start(case, dims[1], trop)
    o.op("atm_bc_temperature", C.byref(s)); d.bc_temperature(s)
    o.op("atm_bc_water_vapour", C.byref(s)); d.bc_water_vapour(s)
    print("BC_WaterVapour: max rel err %.3e" % _compare(o, d, "BC_WaterVapour %r" % (case,), exact=False))
    assert not np.array_equal(o.f["C"], f["C"])
    d.close()


def test_atm_bc_water_vapour_land_before_the_first_ocean_column():
    """Land columns at the start of the visiting order add no evaporation term."""
    dims = (12, 13, 27)
    p, f, g = H.make_atm_start_inputs(*dims, 107)
    f["H"][0, :3, 0] = 1.0
    o = H.CpuModel("oracle", H.ATM, *dims, p, f, g)
    d = atom_amd.DeviceModel(H.ATM, *dims, atom_amd.Params.from_buffer_copy(bytes(p)))
    d.upload_all(f, g)
    s, keep = _start(dict(Ma=0, prev=None), dims[1], 8)
    o.op("atm_bc_water_vapour", C.byref(s)); d.bc_water_vapour(s)
    _compare(o, d, "BC_WaterVapour", exact=False)
    d.close()


@pytest.mark.parametrize("case", [dict(Ma=0, prev=None), dict(Ma=100, prev=90), dict(Ma=140, prev=130, co2_ocean=2., co2_land=1.)])
@pytest.mark.parametrize("dims,seed,trop", [((41, 19, 70), 104, 30), ((12, 13, 27), 105, 8)])
def test_atm_bc_co2_bitexact(dims, seed, trop, case):
    p, f, g, o, d = _pair(dims, seed)
    s, keep = _start(case, dims[1], trop)
    o.op("atm_bc_co2", C.byref(s))
    d.bc_co2(s)
    _compare(o, d, "BC_CO2 %r" % (case,))
    assert not np.array_equal(o.f["CO2"], f["CO2"])
    d.close()


@pytest.mark.parametrize("case", [dict(Ma=0, prev=None), dict(Ma=70, prev=60)])
def test_atm_start_of_slice_full_grid(case, tmp_path):
    """The calls of cAtmosphereModel::RunTimeSlice (cAtmosphereModel.cpp:260-292) in its order on the 1-degree grid with the
    0 Ma topography: everything before BC_Pressure bit-exact, the rest within the libm tolerance."""
    dims = (41, 181, 361)
    p, f, g, o, d = _pair(dims, 108, real_mask=True)
    rng = np.random.default_rng(108)
    temp, prec = tmp_path / "t.xyz", tmp_path / "p.xyz"
    lat = np.cos(np.linspace(-np.pi / 2, np.pi / 2, 181)).reshape(181, 1)
    H.write_lat_lon_value_file(str(temp), -25. + 52. * lat + rng.uniform(-2., 2., dims[1:]), "%.2f")
    H.write_lat_lon_value_file(str(prec), rng.uniform(0., 12., dims[1:]), "%.3f")
    s, i_trop = _start(case, 181, 30)
    it = i_trop.ctypes.data_as(C.POINTER(C.c_int))
    assert o.lib.oracle_atm_ic_cell_structure(o.ctx, it) == 0
    d.ic_cell_structure(i_trop)
    assert o.lib.oracle_atm_read_surface_temperature(o.ctx, str(temp).encode()) == 0
    assert o.lib.oracle_atm_read_surface_precipitation(o.ctx, str(prec).encode()) == 0
    d.read_surface_temperature(str(temp)); d.read_surface_precipitation(str(prec))
    o.op("atm_bc_temperature", C.byref(s)); d.bc_temperature(s)
    o.op("atm_ic_temperature_west_east_coast"); d.ic_temperature_west_east_coast()
    o.op("atm_bc_co2", C.byref(s)); d.bc_co2(s)
    _compare(o, d, "start of slice, arithmetic part")
    o.op("atm_bc_pressure"); d.bc_pressure()
    o.op("atm_bc_water_vapour", C.byref(s)); d.bc_water_vapour(s)
    print("start of slice %r: max rel err %.3e" % (case, _compare(o, d, "start of slice", exact=False)))
    d.close()
