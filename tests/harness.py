"""Shared test plumbing: ctypes views of the three libraries and seeded inputs.

* ``libatomoracle.so``  — oracle/atom_oracle.c, our CPU restatement (checker)
* ``_ref/libatomref.so`` — the reference's own classes, compiled from
  /root/reference by oracle/Makefile (checker; absent where the reference is)
* ``libatomdev.so``     — the product: HIP kernels behind include/atomdev.h

The CPU libraries work in place on numpy buffers; the device library owns HBM
and is driven through upload/op/download.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "libatomoracle.so")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libatomref.so")
DEV_SO = os.path.join(ROOT, "atom_amd", "libatomdev.so")

ATM, HYD = 0, 1

# enum atomdev_field3d (include/atomdev.h)
F3 = dict(T=0, U=1, V=2, W=3, P_DYN=4, C=5, CLOUD=6, ICE=7, CO2=8,
          TN=9, UN=10, VN=11, WN=12, P_DYNN=13, CN=14, CLOUDN=15, ICEN=16, CO2N=17,
          H=18, AUX_U=19, AUX_V=20, AUX_W=21, BUOYANCY=22,
          S_V=23, S_C=24, S_I=25, S_R=26, S_S=27,
          SALT_BALANCE=28, SALT_FINGER=29, SALT_DIFFUSION=30, R_WATER=31, R_SALT_WATER=32,
          RESIDUUM=33, P_STAT=34, Q_LATENT=35, Q_SENSIBLE=36, RADIATION_3D=37, P_RAIN=38, P_SNOW=39, EPSILON_3D=40, S_C_C=41)
F2 = dict(TOPOGRAPHY=0, EVAPORATION_DALTON=1, PRECIPITATION=2, Q_RADIATION=3, PRECIPITABLE_WATER=4,
          EVAPORATION_PENMAN=5, Q_EVAPORATION=6, Q_LATENT_2D=7, Q_SENSIBLE_2D=8, Q_BOTTOM=9, CO2_TOTAL=10, VEGETATION=11,
          ALBEDO=12, EPSILON=13, RADIATION_SURFACE=14, UPWELLING=15, DOWNWELLING=16,
          BOTTOM_WATER=17, SALT_FINGER_2D=18, SALT_DIFFUSION_2D=19, BUOYANCY_FORCE_2D=20, SALT_TOTAL=21,
          TEMPERATURE_NASA=22, PRECIPITATION_NASA=23, TEMP_NASA_CELSIUS=24)

ATM_FIELDS = ["T", "U", "V", "W", "P_DYN", "C", "CLOUD", "ICE", "CO2",
              "TN", "UN", "VN", "WN", "P_DYNN", "CN", "CLOUDN", "ICEN", "CO2N",
              "H", "AUX_U", "AUX_V", "AUX_W", "BUOYANCY", "S_V", "S_C", "S_I", "S_R", "S_S", "RESIDUUM"]
HYD_FIELDS = ["T", "U", "V", "W", "P_DYN", "C", "TN", "UN", "VN", "WN", "P_DYNN", "CN",
              "H", "AUX_U", "AUX_V", "AUX_W", "BUOYANCY",
              "SALT_BALANCE", "SALT_FINGER", "SALT_DIFFUSION", "R_WATER", "R_SALT_WATER", "RESIDUUM", "P_STAT"]
FIELDS2D = ["TOPOGRAPHY", "EVAPORATION_DALTON", "PRECIPITATION"]


class Params(C.Structure):
    _fields_ = [(n, C.c_double) for n in (
        "dt", "dr", "dthe", "dphi", "r0", "the0", "phi0",
        "re", "pr", "g", "Buoyancy", "u_0", "t_0", "c_0",
        "sc_WaterVapour", "sc_CO2", "gam", "r_air", "R_Air", "R_WaterVapour", "L_atm", "cp_l", "t_tropopause",
        "sc", "L_hyd", "r_0_water", "ta", "pa", "ca", "p_0", "ep", "hp", "lv", "ls", "r_water_vapour", "co2_0")]


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "all"])
    if os.path.isdir("/root/reference"):
        subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "ref"])


_libs = {}


def load_oracle():
    if "oracle" not in _libs:
        if not os.path.exists(ORACLE_SO):
            build_oracle()
        lib = C.CDLL(ORACLE_SO)
        lib.oracle_create.restype = C.c_void_p
        lib.oracle_create.argtypes = [C.c_int] * 4 + [C.POINTER(Params)]
        _libs["oracle"] = lib
    return _libs["oracle"]


def have_ref():
    return os.path.exists(REF_SO)


def load_ref():
    if "ref" not in _libs:
        lib = C.CDLL(REF_SO)
        lib.ref_create.restype = C.c_void_p
        lib.ref_create.argtypes = [C.c_int] * 4 + [C.POINTER(Params)]
        _libs["ref"] = lib
    return _libs["ref"]


def default_params(model, im=41, jm=181, km=361):
    """Reference defaults; dr/dthe/dphi follow the grid as on the refined grids: dthe = dphi = pi / (jm - 1), dr = 1 / (im - 1)."""
    p = Params()
    load_oracle().oracle_default_params(C.c_int(model), C.byref(p))
    if (im, jm, km) != (41, 181, 361):
        p.dr = 1.0 / (im - 1)
        p.dthe = np.pi / (jm - 1)
        p.dphi = p.dthe
    return p


class CpuModel:
    """A model whose fields are numpy arrays, advanced in place by the oracle or
    by the compiled reference (same call surface: ``m.op("atm_rk3d")``)."""

    def __init__(self, kind, model, im, jm, km, params, fields, fields2d):
        self.kind = kind
        self.lib = load_oracle() if kind == "oracle" else load_ref()
        self.pre = "oracle_" if kind == "oracle" else "ref_"
        self.model, self.im, self.jm, self.km = model, im, jm, km
        self.params = params
        create = getattr(self.lib, self.pre + "create")
        self.ctx = C.c_void_p(create(model, im, jm, km, C.byref(params)))
        self.f = {k: np.ascontiguousarray(v, dtype=np.float64).copy() for k, v in fields.items()}
        self.g = {k: np.ascontiguousarray(v, dtype=np.float64).copy() for k, v in fields2d.items()}
        for k, a in self.f.items():
            assert a.shape == (im, jm, km), (k, a.shape)
            getattr(self.lib, self.pre + "bind3d")(self.ctx, F3[k], a.ctypes.data_as(C.c_void_p))
        for k, a in self.g.items():
            assert a.shape == (jm, km)
            getattr(self.lib, self.pre + "bind2d")(self.ctx, F2[k], a.ctypes.data_as(C.c_void_p))

    def op(self, name, *args):
        fn = getattr(self.lib, self.pre + name)
        fn.restype = None
        fn(self.ctx, *args)

    def residuum(self, name, min0=0.0, loc0=(0, 0, 0)):
        mn = C.c_double(min0)
        loc = (C.c_int * 3)(*loc0)
        self.op(name, C.byref(mn), loc)
        return mn.value, tuple(loc)

    def steady(self, name, n):
        out = (C.c_double * n)()
        loc = (C.c_int * (3 * n))()
        self.op(name, out, loc)
        return np.array(out), np.array(loc).reshape(n, 3)

    def close(self):
        if self.ctx:
            getattr(self.lib, self.pre + "destroy")(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# --------------------------------------------------------------------------
# seeded synthetic inputs
# --------------------------------------------------------------------------

from atom_amd.synthetic import load_mask_0Ma, make_inputs, shrink_noise, synthetic_mask  # noqa: E402,F401


MSL_FIELDS3D = ["P_STAT", "Q_LATENT", "Q_SENSIBLE", "RADIATION_3D", "P_RAIN", "P_SNOW"]
MSL_FIELDS2D = ["Q_RADIATION", "PRECIPITABLE_WATER", "EVAPORATION_PENMAN", "Q_EVAPORATION", "Q_LATENT_2D",
                "Q_SENSIBLE_2D", "Q_BOTTOM", "CO2_TOTAL"]


def make_msl_inputs(im, jm, km, seed, params):
    """Atmosphere inputs plus what Results_MSL_Atm::run_MSL_data reads and writes: negative moisture and
    precipitation values for its clamps, precipitation above its 25 mm/d cap, no land on the top level
    (the reference reads one level above the array there)."""
    f, g = make_inputs(ATM, im, jm, km, seed, "blobs", params)
    f = {k: f[k] for k in ATM_FIELDS}
    rng = np.random.default_rng(seed + 4242)
    shp = (im, jm, km)
    f["H"][im - 1] = 0.0
    f["P_STAT"] = rng.uniform(200.0, 1013.0, shp)
    f["Q_LATENT"] = rng.uniform(-50.0, 150.0, shp)
    f["Q_SENSIBLE"] = rng.uniform(-50.0, 50.0, shp)
    f["RADIATION_3D"] = rng.uniform(100.0, 400.0, shp)
    f["P_RAIN"] = rng.uniform(-1e-5, 4e-4, shp)
    f["P_SNOW"] = rng.uniform(-1e-5, 1e-4, shp)
    f["BUOYANCY"] = rng.uniform(-1.0, 1.0, shp)
    for k in ("C", "CLOUD", "ICE"):
        f[k] = f[k] - 0.1 * np.abs(f[k]).max() * (rng.uniform(0, 1, shp) < 0.05)
    for k in MSL_FIELDS2D:
        g[k] = rng.uniform(0.0, 5.0, (jm, km))
    return f, g



def make_latent_heat_inputs(im, jm, km, seed):
    """Inputs of BC_Thermo::Latent_Heat: make_msl_inputs plus topography up to level im-2 in some columns, where the
    bottom formula is skipped."""
    p = default_params(ATM, im, jm, km)
    f, g = make_msl_inputs(im, jm, km, seed, p)
    f["H"][im - 2, ::3, ::4] = 1.0
    return p, f, g


def make_radiation_inputs(im, jm, km, seed, trop):
    """Inputs of BC_Thermo::BC_Radiation_multi_layer and the tropopause level of every row (trop, one less in every
    fifth row); a low trop puts mountains above the tropopause."""
    p = default_params(ATM, im, jm, km)
    f, g = make_msl_inputs(im, jm, km, seed, p)
    rng = np.random.default_rng(seed)
    f["EPSILON_3D"] = rng.uniform(0.3, 1.3, (im, jm, km))
    for k in ("ALBEDO", "EPSILON", "RADIATION_SURFACE"):
        g[k] = rng.uniform(0.0, 1.0, (jm, km))
    i_trop = np.full(jm, trop, dtype=np.int32)
    i_trop[::5] -= 1
    return p, f, g, i_trop


def make_ice_scheme_inputs(im, jm, km, seed, first):
    """Inputs of BC_Thermo::Two_Category_Ice_Scheme.  The first columns of the visiting order assign nothing to one of
    the carried members ("warm": S_nuc, "icy": S_i_dep); returns the generator for what the caller changes between
    two calls."""
    p = default_params(ATM, im, jm, km)
    f, g = make_msl_inputs(im, jm, km, seed, p)
    rng = np.random.default_rng(seed)
    shp = (im, jm, km)
    lev = np.arange(im).reshape(im, 1, 1) / (im - 1)
    f["T"] = 1.07 - 0.25 * lev + rng.uniform(-0.03, 0.03, shp)          # 295 K at the ground to 224 K at the top
    f["C"] = rng.uniform(-1e-4, 2e-2, shp) * (1.0 - 0.95 * lev)
    f["CLOUD"] = np.where(rng.random(shp) < 0.5, 0.0, rng.uniform(-1e-5, 1e-3, shp))
    f["ICE"] = np.where(rng.random(shp) < 0.5, 0.0, rng.uniform(-1e-5, 2e-3, shp))
    f["S_C_C"] = rng.uniform(-1e-7, 1e-7, shp)
    f["H"][im - 2, ::3, ::4] = 1.0
    f["H"][im - 3, ::3, ::4] = 1.0
    if first == "warm":   # no ice, above t_nuc: S_nuc is not assigned
        f["T"][:, :4, 0] = 1.05; f["ICE"][:, :4, 0] = 0.0
    else:                 # cold, dry, more ice than the deficit: S_i_dep is not assigned
        f["T"][:, :4, 0] = 0.9; f["C"][:, :4, 0] = 0.0; f["ICE"][:, :4, 0] = 0.01
    f["T"][:, 5:8, 2:4] = 1.05; f["ICE"][:, 5:8, 2:4] = 0.0               # more columns without an S_nuc of their own
    return p, f, g, rng


def make_adjustment_inputs(im, jm, km, seed):
    """Inputs of BC_Thermo::Ice_Water_Saturation_Adjustment: points that inherit the member q_v_hyp (cloud water or ice
    negative on entry), whole columns that start with one, columns without a cold point, warm points below, at and
    above saturation; returns the generator for what the caller changes between two calls."""
    p = default_params(ATM, im, jm, km)
    f, g = make_msl_inputs(im, jm, km, seed, p)
    rng = np.random.default_rng(seed)
    shp = (im, jm, km)
    lev = np.arange(im).reshape(im, 1, 1) / (im - 1)
    f["T"] = 1.07 - 0.25 * lev + rng.uniform(-0.03, 0.03, shp)
    f["T"][:, 7, 3] = 1.06                                   # a column without a cold point
    f["T"][0, ::4, ::5] = 1.0                                # exactly 0 degrees Celsius counts as warm
    cold_all = [(0, 0), (2, 1), (3, 1), (4, 1), (jm - 1, km - 1)]   # (j, k): columns cold from the ground up ...
    for j, k in cold_all:
        f["T"][:, j, k] = np.minimum(f["T"][:, j, k], 0.97)
    # water vapour between 20 % and 115 % of saturation over water (far above it the reference's iteration diverges)
    t_u = f["T"] * p.t_0
    p_h = ((t_u - p.gam * lev * p.L_atm * 1e-2) / t_u) ** (p.g / (1e-2 * p.gam * p.R_Air)) * .01 * p.r_air * p.R_Air * t_u[0]
    e_rain = p.hp * np.exp(17.2694 * (t_u - 273.15) / (t_u - 35.86))
    f["C"] = p.ep * e_rain / (p_h - e_rain) * rng.uniform(0.2, 1.15, shp)
    f["C"] = np.minimum(f["C"], 0.02)
    f["C"][rng.random(shp) < 0.02] = -1e-5
    f["CLOUD"] = np.where(rng.random(shp) < 0.4, 0.0, rng.uniform(-1e-4, 1e-3, shp))
    f["ICE"] = np.where(rng.random(shp) < 0.4, 0.0, rng.uniform(-1e-4, 1e-3, shp))
    f["CN"] = f["C"] + rng.uniform(-1e-5, 1e-5, shp)
    f["S_C_C"] = rng.uniform(-1e-7, 1e-7, shp)
    for j, k in cold_all:
        f["ICE"][0, j, k] = -1e-4                            # ... whose first point cannot assign q_v_hyp itself
    f["CLOUD"][0, 3, 1] = -2e-4; f["ICE"][0, 3, 1] = 3e-4
    return p, f, g, rng


RUN_DATA_FIELDS2D = ["UPWELLING", "DOWNWELLING", "BOTTOM_WATER", "SALT_FINGER_2D", "SALT_DIFFUSION_2D",
                     "BUOYANCY_FORCE_2D", "SALT_TOTAL"]


def make_run_data_inputs(im, jm, km, seed, L_hyd=None):
    """Hydrosphere inputs of Results_Hyd::run_data: the diagnostic 3-D fields of the RK sweep filled, the seven
    surface fields it writes pre-set to rubbish.  L_hyd decides where the Ekman layer starts (500 m below the top)."""
    p = default_params(HYD, im, jm, km)
    if L_hyd is not None:
        p.L_hyd = L_hyd
    f, g = make_inputs(HYD, im, jm, km, seed, "blobs", p)
    f = {k: f[k] for k in HYD_FIELDS}
    rng = np.random.default_rng(seed + 99)
    for k in ("BUOYANCY", "SALT_BALANCE", "SALT_FINGER", "SALT_DIFFUSION"):
        f[k] = rng.uniform(-1.0, 1.0, (im, jm, km))
    for k in RUN_DATA_FIELDS2D:
        g[k] = rng.uniform(-7.0, 7.0, (jm, km))
    return p, f, g


VTK_SECTIONS = {0: "radial", 1: "longal", 2: "zonal"}


def make_vtk_inputs(model, im, jm, km, seed):
    """Every field the ParaView section writers of a model read, all non-trivial; values that print as -0.0000;
    temperatures on both sides of t = 1 (the zonal atmosphere section keeps the ice saturation ratio only below)."""
    rng = np.random.default_rng(seed + 31337)
    shp = (im, jm, km)
    if model == ATM:
        p = default_params(ATM, im, jm, km)
        f, g = make_msl_inputs(im, jm, km, seed, p)
        for k in ("EPSILON_3D", "S_C_C"):
            f[k] = rng.uniform(-1e-3, 1.0, shp)
        f["T"][:, ::3, :] = rng.uniform(0.7, 0.99, (im, len(range(0, jm, 3)), km))
        for k in ("VEGETATION", "ALBEDO", "EPSILON", "TEMPERATURE_NASA", "PRECIPITATION_NASA", "TEMP_NASA_CELSIUS"):
            g[k] = rng.uniform(0.0, 1.5, (jm, km))
    else:
        p, f, g = make_run_data_inputs(im, jm, km, seed)
    f["V"][:, 3, :] = -2e-6     # a row and a column of -0.0000, so that every section holds some
    f["W"][:, :, 1] = -4e-5
    return p, f, g


class HydStart(C.Structure):
    """atomdev_hyd_start (include/atomdev.h)"""
    _fields_ = [("i_beg", C.c_int), ("i_max", C.c_int), ("Ma", C.c_int), ("Ma_max", C.c_int), ("Ma_max_half", C.c_int),
                ("t_average", C.c_double), ("t_cretaceous_max", C.c_double)]


def hyd_start(im, i_beg=0, Ma=0):
    return HydStart(i_beg, im - 1, Ma, 300, 150, 15.0, 10.0)


def make_hyd_start_inputs(im, jm, km, seed):
    """The hydrosphere after Atmosphere_TransferFile_read: surface winds in m/s at the sea level (some exactly zero,
    alone and in pairs), surface temperatures on both sides of 4 C, everything else as make_inputs leaves it."""
    p = default_params(HYD, im, jm, km)
    f, g = make_inputs(HYD, im, jm, km, seed, "blobs", p)
    f = {k: f[k] for k in HYD_FIELDS}
    rng = np.random.default_rng(seed + 2718)
    f["V"][im - 1] = rng.uniform(-10.0, 10.0, (jm, km))
    f["W"][im - 1] = rng.uniform(-10.0, 10.0, (jm, km))
    f["V"][im - 1][rng.uniform(0, 1, (jm, km)) < 0.03] = 0.0
    f["W"][im - 1][rng.uniform(0, 1, (jm, km)) < 0.03] = 0.0
    both = rng.uniform(0, 1, (jm, km)) < 0.02
    f["V"][im - 1][both] = 0.0
    f["W"][im - 1][both] = 0.0
    f["T"][im - 1] = rng.uniform(0.99, 1.1, (jm, km))
    return p, f, g


def write_bathymetry_file(path, jm, km, seed, L_hyd=1000.0):
    """A bathymetry file in the reference's format (lon lat depth, jm x km lines, longitudes from -180): smooth
    continents, trenches deeper than the model ocean, and single-cell islands for the needle smoothing."""
    rng = np.random.default_rng(seed)
    jj, kk = np.meshgrid(np.arange(jm), np.arange(km), indexing="ij")
    z = 900.0 * np.sin(jj / 9.0 + 1.0) * np.cos(kk / 11.0) + rng.uniform(-400.0, 250.0, (jm, km)) - 300.0
    z[rng.uniform(0, 1, (jm, km)) < 0.01] = -1.6 * L_hyd
    z[rng.uniform(0, 1, (jm, km)) < 0.02] = 50.0
    with open(path, "w") as fh:
        for j in range(jm):
            for k in range(km):
                fh.write("%d %d %.2f\n" % (k - 180, 90 - j, z[j, k]))


def write_bathymetry_file_0Ma(path):
    """The 0 Ma bathymetry of the golden mask file on the 1-degree grid, written back in the reference's format
    (longitudes from -180; the reader moves them to 0..360 and repeats the first column as the last)."""
    jm, km = 181, 361
    _, topo = load_mask_0Ma(HYD)
    with open(path, "w") as fh:
        for j in range(jm):
            row = np.concatenate([topo[j, km // 2:2 * (km // 2)], topo[j, :km // 2], topo[j, :1]])
            for k in range(km):
                fh.write("%d %d %.1f\n" % (k - 180, 90 - j, -row[k]))


def make_hyd_slice_inputs_1deg(seed):
    """A hydrosphere on the 1-degree grid as the transfer file leaves it: smooth surface winds in m/s, no dynamic
    pressure, the ocean at rest, smooth temperature and salinity - a state the reference's 20 surface and 4 volume
    iterations keep bounded on the 0 Ma bathymetry (rough random fields diverge within a few pressure iterations, in
    the reference's arithmetic as much as in ours)."""
    im, jm, km = 41, 181, 361
    p, f, g = make_hyd_start_inputs(im, jm, km, seed)
    rng = np.random.default_rng(seed)
    for k in RUN_DATA_FIELDS2D:
        g[k] = rng.uniform(-7.0, 7.0, (jm, km))
    tj, tk = np.meshgrid(np.linspace(0.0, np.pi, jm), np.linspace(0.0, 2.0 * np.pi, km), indexing="ij")
    for k in ("U", "V", "W", "P_DYN", "UN", "VN", "WN", "P_DYNN"):
        f[k] = np.zeros((im, jm, km))
    f["V"][im - 1] = 4.0 * np.sin(6.0 * tj) * (1.0 + 0.2 * np.cos(3.0 * tk))
    f["W"][im - 1] = 7.0 * np.cos(3.0 * tj) * (1.0 + 0.2 * np.sin(2.0 * tk))
    f["T"][:] = 1.0 + 0.09 * np.sin(tj)[None] ** 2
    f["TN"] = f["T"].copy()
    f["C"][:] = 1.0 + 0.02 * np.sin(tj)[None]
    f["CN"] = f["C"].copy()
    return (im, jm, km), p, f, g


def write_topography_file(path, jm, km, seed):
    """A topography file of the atmosphere in the reference's format: sea (negative), plains, mountains up to 9 km."""
    rng = np.random.default_rng(seed)
    jj, kk = np.meshgrid(np.arange(jm), np.arange(km), indexing="ij")
    z = 3000.0 * np.sin(jj / 5.0) * np.cos(kk / 7.0) + rng.uniform(-1500.0, 1500.0, (jm, km))
    z[rng.uniform(0, 1, (jm, km)) < 0.02] = 9000.0
    with open(path, "w") as fh:
        for j in range(jm):
            for k in range(km):
                fh.write("%d %d %.2f\n" % (k - 180, 90 - j, z[j, k]))


def write_salinity_file(path, jm, km, seed):
    """Surface salinity in the reference's format (lon lat psu, km x jm lines, j fastest; negative = no data)."""
    rng = np.random.default_rng(seed)
    s = rng.uniform(30.0, 38.0, (km, jm))
    s[rng.uniform(0, 1, (km, jm)) < 0.2] = -99.0
    with open(path, "w") as fh:
        for k in range(km):
            for j in range(jm):
                fh.write("%.1f %.1f %.4f\n" % (k - 179.5, 89.5 - j, s[k, j]))


def bits_equal(a, b):
    """Bitwise equality up to the sign of zero (NaN == NaN)."""
    a = np.asarray(a); b = np.asarray(b)
    return bool(np.all((a == b) | (np.isnan(a) & np.isnan(b))))


def max_rel_err(a, b):
    """max |a-b| / max(|b|) over the field — the per-field relative error the
    1e-10 target is stated in."""
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    scale = np.max(np.abs(b))
    if scale == 0:
        return float(np.max(np.abs(a - b)))
    return float(np.max(np.abs(a - b)) / scale)


# --------------------------------------------------------------------------
# op sequences shared by oracle/make_golden.py and the parity tests
# --------------------------------------------------------------------------

ATM_SEQ = ["atm_bc_radius", "atm_bc_theta", "atm_bc_phi", "atm_solid_ground", "atm_rk3d",
           "atm_pressure3d", "atm_rk2d", "atm_pressure2d", "move3d", "atm_rk3d"]
HYD_SEQ = ["hyd_bc_radius", "hyd_bc_theta", "hyd_bc_phi", "hyd_pressure_density", "hyd_value_limitation", "hyd_rk3d",
           "hyd_solid_ground", "hyd_pressure3d", "move3d", "hyd_rk3d"]
HYD_SURFACE_SEQ = ["hyd_rk2d", "hyd_pressure2d", "move2d", "hyd_rk2d"]


def run_op(m, op):
    if op == "move3d":
        m.op("move_data_to_new_arrays", C.c_int(-1))
    elif op == "move2d":
        m.op("move_data_to_new_arrays", C.c_int(0 if m.model == ATM else m.im - 1))
    else:
        m.op(op)


def run_sequence(m, seq, names, level=None):
    """Runs seq on model m; after each op records every field the op changed
    (whole field, or one level if ``level`` is given).  Keys: '<step>:<op>/<FIELD>'."""
    snaps = {}
    prev = {k: m.f[k].copy() for k in names}
    for step, op in enumerate(seq):
        run_op(m, op)
        for k in names:
            if not np.array_equal(m.f[k], prev[k], equal_nan=True):
                a = m.f[k] if level is None else m.f[k][level]
                snaps["%02d:%s/%s" % (step, op, k)] = a.copy()
                prev[k] = m.f[k].copy()
    return snaps


def sample_planes(a):
    """The slices of a full 1-degree field kept in the golden file."""
    return {"j62": a[:, 62, :].copy(), "k87": a[:, :, 87].copy(), "i1s8": a[1, ::8, ::8].copy()}


def check_msl_golden(get3d, get2d, exact, tol=1e-10):
    """Compares the fields of a model that ran atm_msl_data(1) on make_msl_inputs(5, 181, 361, 51) with
    tests/golden/atm_msl_data.npz (outputs of the compiled reference).  exact: SHA-256 of every changed
    field; otherwise the sample slices within tol (relative to the field's largest magnitude)."""
    import hashlib
    z = np.load(os.path.join(ROOT, "tests", "golden", "atm_msl_data.npz"))
    n = 0
    for key in z.files:
        if not key.endswith("/sha256"):
            continue
        k = key.split("/")[0]
        a = get2d(k) if k in F2 else get3d(k)
        n += 1
        if exact:
            assert hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest() == str(z[key]), k
        else:
            sl, ref = (a[::4, ::4], z[k + "/s4"]) if k in F2 else (a[:, 62, :], z[k + "/j62"])
            scale = z[k + "/stats"][3]
            assert np.max(np.abs(sl - ref)) <= tol * scale, (k, np.max(np.abs(sl - ref)) / scale)
    return n


def make_transfer_inputs(dims_atm, dims_hyd, seed):
    """Atmosphere and hydrosphere inputs for the surface transfer; the atmosphere's surface values include the
    hard cases of a 4-decimal round trip: exact ties (k + 1/2) / 10^4 that are representable (odd multiples of
    1/32 ...), values one ulp either side of a tie, negative values that print as -0.0000, zeros of both signs."""
    ia, jm, km = dims_atm
    ih = dims_hyd[0]
    pa, ph = default_params(ATM, ia, jm, km), default_params(HYD, ih, jm, km)
    fa, ga = make_inputs(ATM, ia, jm, km, seed, "blobs", pa)
    fh, gh = make_inputs(HYD, ih, jm, km, seed + 1, "blobs", ph)
    fa = {k: fa[k] for k in ATM_FIELDS}
    fh = {k: fh[k] for k in HYD_FIELDS}
    ties = np.array([0.03125, -0.03125, 0.09375, 2.15625, -7.28125, 0.00005, 1.5, 1e-5, -1e-5, -4e-5, 6e-5, 0.0, -0.0,
                     0.00015, 0.00025, 123.45675, 1000.00005, 0.99995, -0.99995, 5e-324, 24.99995])
    hard = np.concatenate([ties, np.nextafter(ties, np.inf), np.nextafter(ties, -np.inf)])
    n = min(hard.size, km * (jm - 1))
    for name, arr in (("T", fa["T"][0]), ("P_DYN", fa["P_DYN"][0]), ("E", ga["EVAPORATION_DALTON"]), ("P", ga["PRECIPITATION"])):
        arr.reshape(-1)[:n] = hard[:n]
    # v and w are multiplied by u_0 = 8 before they are written: put the hard cases in the product
    fa["V"][0].reshape(-1)[:n] = hard[:n] / 8.0
    fa["W"][0].reshape(-1)[km:km + n] = hard[:n] / 8.0
    return (pa, fa, ga), (ph, fh, gh)


# --------------------------------------------------------------------------
# the reference's 3-D iteration loops, operator by operator
# --------------------------------------------------------------------------

def _move3d(m):
    if hasattr(m, "move_data_to_new_arrays"):
        m.move_data_to_new_arrays(-1)
    else:
        m.op("move_data_to_new_arrays", C.c_int(-1))


def hyd_3d_loop(m, pressure_iters=2, velocity_iters=2, run_data=False):
    """cHydrosphereModel.cpp:420-610 without its printing: returns the queries' results in call order.
    run_data=True adds Results_Hyd::run_data where the reference calls it (:597); its seven surface fields must be
    bound then."""
    log = []
    for _ in range(pressure_iters):
        for _ in range(velocity_iters):
            log.append(m.residuum("hyd_residuum3d"))
            for op in ("hyd_bc_radius", "hyd_bc_theta", "hyd_bc_phi", "hyd_pressure_density", "hyd_value_limitation",
                       "hyd_rk3d", "hyd_solid_ground"):
                m.op(op)
            log.append(m.residuum("hyd_residuum3d"))
            log.append(m.steady("hyd_steady3d", 6))
            if run_data:
                m.op("hyd_run_data")
            _move3d(m)
        m.op("hyd_pressure3d")
    return log


def _move2d(m):
    level = 0 if m.model == ATM else m.im - 1
    if hasattr(m, "move_data_to_new_arrays"):
        m.move_data_to_new_arrays(level)
    else:
        m.op("move_data_to_new_arrays", C.c_int(level))


def hyd_start_of_slice(m, start, bathymetry_file, salinity_file):
    """cHydrosphereModel.cpp:248-297 for a model whose surface already holds the transfer file's values: the two
    readers, the initial conditions of BC_Thermohalin in the reference's order, the first old -> new copies."""
    dev = not hasattr(m, "ctx")
    if dev:
        m.read_sea_ground(bathymetry_file)
        if start.Ma == 0:
            m.read_surface_salinity(salinity_file)
        m.ic_ekman_spiral(start.i_beg)
        if start.Ma <= 41:
            m.ic_circumpolar_current(start.i_beg)
        m.bc_temperature_salinity(start)
    else:
        m.op("hyd_read_sea_ground", bathymetry_file.encode())
        if start.Ma == 0:
            m.op("hyd_read_surface_salinity", salinity_file.encode())
        m.op("hyd_ic_ekman_spiral", C.c_int(start.i_beg))
        if start.Ma <= 41:
            m.op("hyd_ic_circumpolar_current", C.c_int(start.i_beg))
        m.op("hyd_bc_temperature_salinity", C.byref(start))
    m.op("hyd_pressure_density")
    _move3d(m)
    _move2d(m)


def hyd_2d_loop(m, pressure_iters=2, velocity_iters=2):
    """cHydrosphereModel.cpp:314-404 without the two queries, whose results are printed only."""
    for _ in range(pressure_iters):
        for _ in range(velocity_iters):
            for op in ("hyd_bc_theta", "hyd_bc_phi", "hyd_value_limitation", "hyd_rk2d"):
                m.op(op)
            _move2d(m)
        m.op("hyd_pressure2d")


def _is_device(m):
    return not hasattr(m, "ctx")


def _atm_physics(m, radiation_constants=None):
    """The BC_Thermo column physics and the two searches between them, the same calls for the oracle and the device."""
    import atom_amd
    dev = _is_device(m)
    dp = C.POINTER(C.c_double)

    def adjust(q):
        if dev:
            return m.ice_water_saturation_adjustment(q)
        st = C.c_double(q)
        fn = m.lib.oracle_atm_ice_water_saturation_adjustment; fn.argtypes = [C.c_void_p, dp]; fn.restype = None
        fn(m.ctx, C.byref(st))
        return st.value

    def ice_scheme(state):
        if dev:
            return m.two_category_ice_scheme(state)
        st = (C.c_double * 2)(*state)
        fn = m.lib.oracle_atm_two_category_ice_scheme; fn.argtypes = [C.c_void_p, dp]; fn.restype = None
        fn(m.ctx, st)
        return (st[0], st[1])

    def radiation(i_trop):
        r = radiation_constants or atom_amd.Radiation(
            albedo_pole=.7, albedo_equator=.15, epsilon_pole=.59, epsilon_equator=.594, epsilon_tropopause=.001, rad_pole=40.,
            rad_equator=230., co2_coeff=.98 * (330. / 320.), sigma=5.670280e-8)   # param.py:60-70, 123-130
        if dev:
            m.radiation_multi_layer(i_trop, r)
        else:
            fn = m.lib.oracle_atm_radiation_multi_layer
            fn.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]; fn.restype = None
            fn(m.ctx, C.byref(r), i_trop.ctypes.data_as(C.POINTER(C.c_int)))

    def max_precipitation():
        if dev:
            return m.minmax("PRECIPITATION")[0][0]
        out = (C.c_double * 2)(); loc = (C.c_int * 6)()
        m.lib.oracle_minmax2d.argtypes = [C.c_void_p, C.c_int, dp, C.POINTER(C.c_int)]
        m.lib.oracle_minmax2d(m.ctx, F2["PRECIPITATION"], out, loc)
        return out[0]

    def vegetation(maxp):
        if dev:
            m.vegetation_distribution(maxp)
        else:
            m.op("atm_vegetation_distribution", C.c_double(maxp))

    return adjust, ice_scheme, radiation, max_precipitation, vegetation


def atm_3d_loop(m, pressure_iters=2, velocity_iters=2, physics=False, i_trop=None, radiation_constants=None, radiation_model=1,
                after_pressure=None):
    """cAtmosphereModel.cpp:756-874 with the operators of this library: returns the queries' results in call order.
    physics=True adds the BC_Thermo column physics where the reference calls it (:787-789, :828-831, :844-866;
    the saturation adjustment and the ice scheme on even velocity iterations) with the carried class members
    kept across the iterations as the reference's circulation object keeps them."""
    log = []
    if physics:
        adjust, ice_scheme, radiation, max_precipitation, vegetation = _atm_physics(m, radiation_constants)
        q_v_hyp, ice_state = 0.0, (0.0, 0.0)
    _move3d(m)
    for pressure_iter in range(1, pressure_iters + 1):
        for velocity_iter in range(1, velocity_iters + 1):
            mn, loc = m.residuum("atm_residuum3d")
            log.append((mn, loc))
            for op in ("atm_bc_radius", "atm_bc_theta", "atm_bc_phi"):
                m.op(op)
            if physics and velocity_iter % 2 == 0:
                q_v_hyp = adjust(q_v_hyp)
            for op in ("atm_value_limitation", "atm_solid_ground", "atm_rk3d", "atm_value_limitation"):
                m.op(op)
            if physics and radiation_model == 1:
                radiation(i_trop)
            log.append(m.residuum("atm_residuum3d", mn, loc))  # same Accuracy_Atm object: state carried
            log.append(m.steady("atm_steady3d", 9))
            if physics:
                m.op("atm_latent_heat")
                vegetation(max_precipitation())
            if hasattr(m, "msl_data"):
                m.msl_data(radiation_model)
            else:
                m.op("atm_msl_data", C.c_int(radiation_model))
            if physics and velocity_iter % 2 == 0:
                ice_state = ice_scheme(ice_state)
            _move3d(m)
        m.op("atm_pressure3d")
        if after_pressure:
            after_pressure(pressure_iter)
    if physics:
        log.append((q_v_hyp, (0, 0, 0)))
        log.append((ice_state[0], (0, 0, 0)))
        log.append((ice_state[1], (0, 0, 0)))
    return log


# --------------------------------------------------------------------------
# BC_Thermo: the run behind tests/golden/bc_thermo.npz (oracle/make_golden.py with the compiled reference,
# tests/test_oracle_golden.py with the oracle)
# --------------------------------------------------------------------------

def default_radiation():
    """Constants of BC_Radiation_multi_layer at their defaults (param.py:60-70, 123-130) with CO2 = 1."""
    import atom_amd
    return atom_amd.Radiation(albedo_pole=.7, albedo_equator=.15, epsilon_pole=.59, epsilon_equator=.594,
                              epsilon_tropopause=.001, rad_pole=40., rad_equator=230., co2_coeff=.98 * (330. / 320.),
                              sigma=5.670280e-8)


def thermo_golden_run(kind):
    """Runs Value_Limitation_Atm, Latent_Heat, BC_Radiation_multi_layer (twice), Two_Category_Ice_Scheme (twice, state
    handed on) and Ice_Water_Saturation_Adjustment (twice) of ``kind`` ("ref" or "oracle") on seeded inputs.  Returns
    {key: SHA-256 of a changed field | carried state | sample slice}."""
    import hashlib
    out = {}
    dp = C.POINTER(C.c_double)

    def record(tag, m, f, g):
        for k in sorted(f):
            if not np.array_equal(m.f[k], f[k]):
                out["%s/%s/sha256" % (tag, k)] = np.array(hashlib.sha256(m.f[k].tobytes()).hexdigest())
                out["%s/%s/j5" % (tag, k)] = m.f[k][:, 5, :].copy()
        for k in sorted(g):
            if not np.array_equal(m.g[k], g[k]):
                out["%s/%s/sha256" % (tag, k)] = np.array(hashlib.sha256(m.g[k].tobytes()).hexdigest())

    pre = "oracle_" if kind == "oracle" else "ref_"
    dims = (12, 13, 17)
    # limiter
    p = default_params(ATM, *dims)
    f, g = make_inputs(ATM, *dims, 5, "blobs", p)
    f = {k: f[k] for k in ATM_FIELDS}
    rng = np.random.default_rng(5)
    for k in ["U", "V", "W", "T", "C", "CLOUD", "ICE", "CO2"]:
        f[k] = f[k] + rng.normal(0, 2.0 * np.abs(f[k]).max(), f[k].shape)
    m = CpuModel(kind, ATM, *dims, p, f, g)
    m.op("atm_value_limitation")
    record("limit", m, f, g)
    # latent heat (no topography on level im-2: the reference would read outside its arrays)
    p, f, g = make_latent_heat_inputs(*dims, 58)
    f["H"][dims[0] - 2] = 0.0
    m = CpuModel(kind, ATM, *dims, p, f, g)
    m.op("atm_latent_heat")
    record("latent", m, f, g)
    # radiation, two calls
    p, f, g, i_trop = make_radiation_inputs(*dims, 60, 8)
    m = CpuModel(kind, ATM, *dims, p, f, g)
    r = default_radiation()
    fn = getattr(m.lib, pre + "atm_radiation_multi_layer")
    fn.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
    for call in range(2):
        fn(m.ctx, C.byref(r), i_trop.ctypes.data_as(C.POINTER(C.c_int)))
        record("radiation%d" % call, m, f, g)
    # ice scheme, two calls
    p, f, g, rng = make_ice_scheme_inputs(*dims, 63, "icy")
    m = CpuModel(kind, ATM, *dims, p, f, g)
    fn = getattr(m.lib, pre + "atm_two_category_ice_scheme"); fn.argtypes = [C.c_void_p, dp]
    st = (C.c_double * 2)(3.0e-7, -2.0e-7)
    for call in range(2):
        fn(m.ctx, st)
        record("ice%d" % call, m, f, g)
        out["ice%d/state" % call] = np.array([st[0], st[1]])
        for k in ("C", "ICE"):
            m.f[k][...] = m.f[k] * rng.uniform(0.5, 1.5, dims)
    # saturation adjustment, two calls
    p, f, g, rng = make_adjustment_inputs(*dims, 66)
    m = CpuModel(kind, ATM, *dims, p, f, g)
    fn = getattr(m.lib, pre + "atm_ice_water_saturation_adjustment"); fn.argtypes = [C.c_void_p, dp]
    q = C.c_double(4.0e-3)
    for call in range(2):
        fn(m.ctx, C.byref(q))
        record("adjust%d" % call, m, f, g)
        out["adjust%d/state" % call] = np.array([q.value])
        for k, lo, hi in (("C", 0.7, 1.02), ("CLOUD", 0.5, 1.5), ("ICE", 0.5, 1.5)):
            m.f[k][...] = m.f[k] * rng.uniform(lo, hi, dims) - 2e-5
    return out


# --------------------------------------------------------------------------
# the atmosphere's start of a time slice (BC_Thermo's initial conditions)
# --------------------------------------------------------------------------

def atm_start(jm, trop=30, **kw):
    """atomdev_atm_start with param.py's defaults (:45-58, 108-129), the tropopause at level ``trop`` in every row (an
    array of jm levels is taken as it is), and whatever ``kw`` overrides.  Returns (struct, the array it points to)."""
    import atom_amd
    s = atom_amd.AtmStart(Ma=0, first_time_slice=1, Ma_previous=0, sun=0, RadiationModel=1, NASATemperature=1, declination=0,
                          sun_position_lat=60, sun_position_lon=180, t_cretaceous=0., t_cretaceous_add=0., t_average=15.,
                          t_equator=1.0842, t_pole=.945, t_land=0., c_tropopause=.001, c_ocean=.58, c_land=.64,
                          co2_average=280., co2_equator=330., co2_tropopause=320., co2_pole=320., co2_cretaceous=330.,
                          co2_vegetation=3., co2_ocean=0., co2_land=0.)
    for k, v in kw.items():
        setattr(s, k, v)
    i_trop = np.ascontiguousarray(np.broadcast_to(np.asarray(trop, dtype=np.int32), (jm,)))
    s.im_tropopause = i_trop.ctypes.data_as(C.POINTER(C.c_int))
    return s, i_trop


def set_paleo_time(s, Ma, Ma_previous, curve, t_0=273.15):
    """Fills Ma, first_time_slice, Ma_previous, t_cretaceous and t_cretaceous_add of an atomdev_atm_start the way
    BC_Thermo derives them (BC_Thermo.cpp:93-94, 446-451); ``curve(Ma)`` is the mean temperature of the model's curve, a
    float like the reference's.  Ma_previous None: the first slice."""
    f32 = np.float32
    s.Ma = Ma
    s.first_time_slice = 1 if Ma_previous is None else 0
    s.Ma_previous = 0 if Ma_previous is None else Ma_previous
    s.t_cretaceous = float(f32(curve(Ma)) - f32(curve(0)))
    s.t_cretaceous_add = 0. if Ma_previous is None else float(f32(curve(Ma)) - f32(curve(Ma_previous))) / t_0


def write_lat_lon_value_file(path, values_jk, fmt="%.4f"):
    """A file in the layout of the NASA surface files: one 'lat lon value' line per cell, longitude outer, latitude
    inner (BC_Thermo.cpp:2329-2341)."""
    jm, km = values_jk.shape
    with open(path, "w") as fh:
        for k in range(km):
            for j in range(jm):
                fh.write(("%d %d " + fmt + "\n") % (90 - j, k - 180, values_jk[j, k]))


ATM_START_FIELDS2D = ["VEGETATION", "TEMPERATURE_NASA", "PRECIPITATION_NASA"]


def make_atm_start_inputs(im, jm, km, seed):
    """Atmosphere fields as reset_arrays leaves them (cAtmosphereModel.cpp:384-472: t, c, co2 at their initial values, no
    wind) with a seeded topography (none on the top level), surface wind for Dalton's evaporation, vegetation, and a
    temperature file's content in t level 0 / temperature_NASA."""
    p = default_params(ATM, im, jm, km)
    rng = np.random.default_rng(seed)
    h, topo = synthetic_mask(ATM, im, jm, km, p.L_atm, rng, "blobs")
    h[im - 1] = 0.0
    shp = (im, jm, km)
    f = {k: np.zeros(shp) for k in ATM_FIELDS + ["P_STAT"]}
    f["H"] = h
    f["T"][...] = 1.0; f["TN"][...] = 1.0
    f["CO2"][...] = 1.0; f["CO2N"][...] = 1.0
    f["V"] = rng.uniform(-0.02, 0.02, shp); f["W"] = rng.uniform(-0.05, 0.05, shp)
    g = {k: np.zeros((jm, km)) for k in FIELDS2D + ATM_START_FIELDS2D}
    g["TOPOGRAPHY"] = topo
    g["VEGETATION"] = rng.uniform(0.0, 1.0, (jm, km)) * (h[0] > 0)
    lat = np.cos(np.linspace(-np.pi / 2, np.pi / 2, jm)).reshape(jm, 1)
    celsius = np.round(-20. + 48. * lat + rng.uniform(-3., 3., (jm, km)), 2)
    g["TEMPERATURE_NASA"] = (celsius + p.t_0) / p.t_0
    f["T"][0] = g["TEMPERATURE_NASA"]
    return p, f, g


# --------------------------------------------------------------------------
# cAtmosphereModel::RunTimeSlice (cAtmosphereModel.cpp:140-341) composed of this library's operators
# --------------------------------------------------------------------------

ATM_SLICE_FIELDS3D = ATM_FIELDS + ["P_STAT", "Q_LATENT", "Q_SENSIBLE", "RADIATION_3D", "P_RAIN", "P_SNOW", "EPSILON_3D", "S_C_C"]
ATM_SLICE_FIELDS2D = ["TOPOGRAPHY", "EVAPORATION_DALTON", "PRECIPITATION", "Q_RADIATION", "PRECIPITABLE_WATER", "EVAPORATION_PENMAN",
                      "Q_EVAPORATION", "Q_LATENT_2D", "Q_SENSIBLE_2D", "Q_BOTTOM", "CO2_TOTAL", "VEGETATION", "ALBEDO", "EPSILON",
                      "RADIATION_SURFACE", "TEMPERATURE_NASA", "PRECIPITATION_NASA", "TEMP_NASA_CELSIUS"]


def atm_reset_arrays(dims, p):
    """reset_arrays (cAtmosphereModel.cpp:384-472): t, c, co2 and their new-level twins at ta, ca, coa (param.py:97-103),
    everything else 0."""
    f = {k: np.zeros(dims) for k in ATM_SLICE_FIELDS3D}
    for k, v in (("T", p.ta), ("C", p.ca), ("CO2", 1.0)):
        f[k][...] = v; f[k + "N"][...] = v
    for k in ("P_DYN", "P_DYNN", "P_STAT"):
        f[k][...] = p.pa
    g = {k: np.zeros(dims[1:]) for k in ATM_SLICE_FIELDS2D}
    return f, g


def atm_2d_loop(m, pressure_iters=10, velocity_iters=2):
    """cAtmosphereModel.cpp:667-736 without its printing."""
    for _ in range(pressure_iters):
        for _ in range(velocity_iters):
            m.op("atm_bc_theta"); m.op("atm_bc_phi")
            first = m.residuum("atm_residuum2d")
            for op in ("atm_value_limitation", "atm_solid_ground", "atm_rk2d"):
                m.op(op)
            m.residuum("atm_residuum2d", *first)
            m.steady("atm_steady2d", 3)
            _move2d(m)
        m.op("atm_pressure2d")


def slice_radiation(s, co2_0=280., CO2=1., co2_factor=.98):
    """The radiation constants of a time slice: param.py:60-70 with co2_coeff as BC_Radiation_multi_layer derives it after
    BC_CO2 has divided the members by co2_0 (BC_Thermo.cpp:251-256, 806-812)."""
    import atom_amd
    coeff = co2_factor * ((s.co2_equator / co2_0) / (s.co2_tropopause / co2_0)) if CO2 == 1. else 1.
    return atom_amd.Radiation(albedo_pole=.7, albedo_equator=.15, epsilon_pole=.59, epsilon_equator=.594, epsilon_tropopause=.001,
                              rad_pole=40., rad_equator=230., co2_coeff=coeff, sigma=5.670280e-8)


def tropopause_add(t_cretaceous, t_0=273.15):
    """BC_Thermo::GetTropopauseHightAdd as its callers use it (BC_Thermo.cpp:3298-3302); C's round()."""
    x = ((t_cretaceous / t_0) * t_0) / 2.6
    return int(np.floor(abs(x) + 0.5) * np.sign(x))


def atm_time_slice(m, s, i_trop, files, counts, mean_temperature, coast_correction, checkpoint=2):
    """cAtmosphereModel::RunTimeSlice on a CPU model whose fields atm_reset_arrays made.  files: bathymetry, temperature,
    precipitation inputs, then the output directory and the bathymetry name that goes into the output files' names; counts:
    (velocity_iter_max_2D, pressure_iter_max_2D, velocity_iter_max, pressure_iter_max).  Returns the 3-D loop's log."""
    bathymetry, temperature, precipitation, output_path, bathymetry_name = files
    ip = C.POINTER(C.c_int)

    def write_files(n, final):   # cAtmosphereModel::write_file (:599-652)
        stem = "%s/[%s]" % (output_path, bathymetry_name)
        for section, index, tag in ((0, 0, "radial"), (1, 62, "longal"), (2, 87, "zonal")):
            fn = m.lib.oracle_write_vtk; fn.restype = C.c_int
            assert fn(m.ctx, section, index, s.Ma, ("%s_Atm_%s_%d_%d.vtk" % (stem, tag, index, n)).encode()) == 0
        call("atm_write_transfer_file", (stem + "_Transfer_Atm.vw").encode())
        fn = m.lib.oracle_write_plot_data; fn.restype = C.c_int
        assert fn(m.ctx, (stem + "_PlotData_Atm" + ("" if final else "_%d" % n) + ".xyz").encode()) == 0

    def call(name, *args):
        fn = getattr(m.lib, m.pre + name)
        fn.restype = C.c_int
        assert fn(m.ctx, *args) == 0, name

    call("atm_read_mountain_surface", bathymetry.encode())
    call("atm_ic_cell_structure", i_trop.ctypes.data_as(ip))
    call("atm_read_surface_temperature", temperature.encode())
    call("atm_read_surface_precipitation", precipitation.encode())
    m.op("atm_bc_temperature", C.byref(s))
    if coast_correction:
        m.op("atm_ic_temperature_west_east_coast")
    m.op("atm_bc_pressure")
    m.op("atm_bc_water_vapour", C.byref(s))
    m.op("atm_bc_co2", C.byref(s))
    r = slice_radiation(s, m.params.co2_0)
    levels = (i_trop + tropopause_add(s.t_cretaceous, m.params.t_0)).astype(np.int32)
    if s.RadiationModel == 1:
        fn = m.lib.oracle_atm_radiation_multi_layer
        fn.argtypes = [C.c_void_p, C.c_void_p, ip]; fn.restype = None
        fn(m.ctx, C.byref(r), levels.ctypes.data_as(ip))
    _move3d(m)
    _move2d(m)
    atm_2d_loop(m, counts[1], counts[0])
    def after_pressure(pressure_iter):   # :886-888
        if checkpoint and pressure_iter % checkpoint == 0:
            write_files(pressure_iter * counts[2], False)

    log = atm_3d_loop(m, counts[3], counts[2], physics=True, i_trop=levels, radiation_constants=r, radiation_model=s.RadiationModel,
                      after_pressure=after_pressure)
    m.op("atm_restrain_temperature", C.c_double(mean_temperature))
    write_files(counts[3] * counts[2], True)
    return log


# ---- the whole atmosphere time slice: inputs that travel, the reference's run, the golden record ----

ATM_SLICE_INPUTS = os.path.join(ROOT, "tests", "golden", "atm_slice_inputs.npz")
ATM_SLICE_GOLDEN = os.path.join(ROOT, "tests", "golden", "atm_time_slice.npz")
HYD_SLICE_GOLDEN = os.path.join(ROOT, "tests", "golden", "hyd_time_slice.npz")


def text_file_numbers(path, stride=53):
    """A digest of the numbers of a text file (an ASCII VTK section, a PlotData table): how many tokens read as numbers,
    and every stride-th of them.  Two files that agree in it up to the last printed digit hold the same table."""
    import re
    tokens = re.findall(r"(?<![\w.])[-+]?(?:\d+\.\d*|\.\d+|\d+)(?:[eE][-+]?\d+)?(?![\w.])", open(path).read())
    v = np.array([float(t) for t in tokens])
    return np.concatenate(([float(v.size)], v[::stride]))
ATM_SLICE_CASES = [(0, None), (40, 30), (140, 130)]     # (Ma, the slice before it)


def write_atm_slice_inputs(directory):
    """Writes the reference's input files (the 1-degree surface temperature, precipitation and salinity, the Golonka
    topography of the golden cases' time slices) from tests/golden/atm_slice_inputs.npz, in the reference's text
    formats (the surface files digit for digit, the topography as numbers that read back to the same doubles).  Returns (temperature file, precipitation file, {Ma: mean temperature of the curve})."""
    z = np.load(ATM_SLICE_INPUTS)
    jm, km = 181, 361
    for name, key in (("SurfaceTemperature_NASA.xyz", "temperature_c100"), ("SurfacePrecipitation_NASA.xyz", "precipitation_c100"),
                      ("SurfaceSalinity_NASA.xyz", "salinity_c100")):
        v = z[key].astype(np.int64)
        with open(os.path.join(directory, name), "w") as fh:
            n = 0
            for k in range(km):
                for j in range(jm):
                    c = int(v[n]); n += 1
                    fh.write("%.2f %.2f %s%d.%02d\n" % (k, 90 - j, "-" if c < 0 else "", abs(c) // 100, abs(c) % 100))
    for key in z.files:
        if key.startswith("topography_"):
            Ma = int(key.split("_")[1])
            v = z[key].astype(np.float64)
            with open(os.path.join(directory, "%dMa_Golonka.xyz" % Ma), "w") as fh:
                n = 0
                for j in range(jm):
                    for k in range(km):
                        fh.write("%d\t%d\t%r\n" % (k - 180, 90 - j, float(v[n]))); n += 1   # shortest text that reads back as v[n]
    curve = {int(a): float(b) for a, b in zip(z["curve_Ma"], z["curve_temperature"])}
    return (os.path.join(directory, "SurfaceTemperature_NASA.xyz"), os.path.join(directory, "SurfacePrecipitation_NASA.xyz"), curve)


class RefSliceFiles(C.Structure):
    _fields_ = [(n, C.c_char_p) for n in ("bathymetry_path", "temperature_file", "precipitation_file", "output_path")]


def atm_slice_start(Ma, prev, curve):
    """atomdev_atm_start of a golden time slice; curve: {Ma: mean temperature} or a callable."""
    s, i_trop = atm_start(181, 30)
    set_paleo_time(s, Ma, prev, curve if callable(curve) else (lambda a: curve[int(a)]))
    return s, i_trop


def run_ref_atm_slice(r, s, counts, inputs_dir, output_path):
    """cAtmosphereModel::RunTimeSlice of the compiled reference on the CpuModel r (fields bound for copying out)."""
    import atom_amd
    lib = load_ref()
    fl = RefSliceFiles(inputs_dir.encode(), os.path.join(inputs_dir, "SurfaceTemperature_NASA.xyz").encode(),
                       os.path.join(inputs_dir, "SurfacePrecipitation_NASA.xyz").encode(), output_path.encode())
    L = atom_amd.AtmLoop(velocity_iter_max_2D=counts[0], pressure_iter_max_2D=counts[1], velocity_iter_max=counts[2],
                         pressure_iter_max=counts[3], column_physics=1, RadiationModel=s.RadiationModel,
                         radiation=slice_radiation(s), queries=1, nm=200)
    assert lib.ref_atm_run_time_slice(r.ctx, C.byref(s), C.byref(L), C.byref(fl)) == 0


def run_oracle_atm_slice(o, s, i_trop, counts, inputs_dir, output_path, curve):
    name = "%dMa_Golonka.xyz" % s.Ma
    value = curve(s.Ma) if callable(curve) else curve[int(s.Ma)]
    return atm_time_slice(o, s, i_trop, (os.path.join(inputs_dir, name), os.path.join(inputs_dir, "SurfaceTemperature_NASA.xyz"),
                                         os.path.join(inputs_dir, "SurfacePrecipitation_NASA.xyz"), output_path, name),
                          counts, float(np.float32(value)), coast_correction=(s.Ma > 0 and s.NASATemperature == 1))


def atm_slice_record(m, output_path, tag):
    """What the golden file keeps of a finished slice: SHA-256 and a sample of every field, SHA-256 of every output file."""
    import hashlib
    out = {}
    for k in sorted(list(m.f) + list(m.g)):
        if k == "RESIDUUM":
            continue
        a = m.f[k] if k in m.f else m.g[k]
        out["%s/%s/sha256" % (tag, k)] = np.array(hashlib.sha256(a.tobytes()).hexdigest())
        out["%s/%s/sample" % (tag, k)] = a[::4, 62, ::8].copy() if a.ndim == 3 else a[62, ::8].copy()
    for fn in sorted(os.listdir(output_path)):
        out["%s/file/%s" % (tag, fn)] = np.array(hashlib.sha256(open(os.path.join(output_path, fn), "rb").read()).hexdigest())
    return out


def hyd_reset_arrays(dims, p):
    """cHydrosphereModel::reset_arrays (hydrosphere/cHydrosphereModel.cpp:636-694): t and c at ta and ca, the rest 0."""
    f = {k: np.zeros(dims) for k in HYD_FIELDS}
    for k, v in (("T", p.ta), ("C", p.ca)):
        f[k][...] = v; f[k + "N"][...] = v
    for k in ("P_DYN", "P_DYNN", "P_STAT"):
        f[k][...] = p.pa
    g = {k: np.zeros(dims[1:]) for k in FIELDS2D + RUN_DATA_FIELDS2D}
    return f, g
