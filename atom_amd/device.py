"""ctypes binding of include/atomdev.h.

``DeviceModel`` mirrors what the reference's per-slice loops do with their
``Array`` members (atmosphere/cAtmosphereModel.cpp:654-898,
hydrosphere/cHydrosphereModel.cpp:309-622): fields live on the device, each
method is one reference operator.
"""
import ctypes as C
import os

import numpy as np

ATM, HYD = 0, 1

# enum atomdev_field3d / atomdev_field2d
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


class Radiation(C.Structure):
    """struct atomdev_radiation"""
    _fields_ = [(n, C.c_double) for n in ("albedo_pole", "albedo_equator", "epsilon_pole", "epsilon_equator",
                                          "epsilon_tropopause", "rad_pole", "rad_equator", "co2_coeff", "sigma")]


class Params(C.Structure):
    """struct atomdev_params"""
    _fields_ = [(n, C.c_double) for n in (
        "dt", "dr", "dthe", "dphi", "r0", "the0", "phi0",
        "re", "pr", "g", "Buoyancy", "u_0", "t_0", "c_0",
        "sc_WaterVapour", "sc_CO2", "gam", "r_air", "R_Air", "R_WaterVapour", "L_atm", "cp_l", "t_tropopause",
        "sc", "L_hyd", "r_0_water", "ta", "pa", "ca", "p_0", "ep", "hp", "lv", "ls", "r_water_vapour", "co2_0")]


class HydStart(C.Structure):
    """atomdev_hyd_start: what BC_Thermohalin's constructor receives beyond Params."""
    _fields_ = [("i_beg", C.c_int), ("i_max", C.c_int), ("Ma", C.c_int), ("Ma_max", C.c_int), ("Ma_max_half", C.c_int),
                ("t_average", C.c_double), ("t_cretaceous_max", C.c_double)]


class HydSlice(C.Structure):
    """atomdev_hyd_slice: a hydrosphere time slice as cHydrosphereModel::RunTimeSlice runs it."""
    _fields_ = [("start", HydStart), ("velocity_iter_max_2D", C.c_int), ("pressure_iter_max_2D", C.c_int),
                ("velocity_iter_max", C.c_int), ("pressure_iter_max", C.c_int), ("transfer_file", C.c_char_p),
                ("bathymetry_file", C.c_char_p), ("salinity_file", C.c_char_p), ("queries", C.c_int),
                ("checkpoint", C.c_int), ("paraview_panorama_vts", C.c_int), ("output_path", C.c_char_p),
                ("bathymetry_name", C.c_char_p)]


class AtmStart(C.Structure):
    """atomdev_atm_start: what BC_Thermo's constructor receives beyond Params."""
    _fields_ = ([(n, C.c_int) for n in ("Ma", "first_time_slice", "Ma_previous", "sun", "RadiationModel", "NASATemperature",
                                        "declination", "sun_position_lat", "sun_position_lon")] +
                [(n, C.c_double) for n in ("t_cretaceous", "t_cretaceous_add", "t_average", "t_equator", "t_pole", "t_land",
                                           "c_tropopause", "c_ocean", "c_land", "co2_average", "co2_equator",
                                           "co2_tropopause", "co2_pole", "co2_cretaceous", "co2_vegetation", "co2_ocean",
                                           "co2_land")] +
                [("im_tropopause", C.POINTER(C.c_int))])


class AtmLoop(C.Structure):
    """atomdev_atm_loop: the iteration loops of cAtmosphereModel (run_2D_loop, run_3D_loop)."""
    _fields_ = [("velocity_iter_max_2D", C.c_int), ("pressure_iter_max_2D", C.c_int), ("velocity_iter_max", C.c_int),
                ("pressure_iter_max", C.c_int), ("column_physics", C.c_int), ("RadiationModel", C.c_int),
                ("radiation", Radiation), ("i_trop", C.POINTER(C.c_int)), ("queries", C.c_int), ("nm", C.c_int)]


class AtmLoopState(C.Structure):
    """atomdev_atm_loop_state"""
    _fields_ = [("q_v_hyp", C.c_double), ("S_nuc", C.c_double), ("S_i_dep", C.c_double), ("residuum", C.c_double * 2),
                ("residuum_loc", C.c_int * 3)]


class AtmSlice(C.Structure):
    """atomdev_atm_slice: an atmosphere time slice as cAtmosphereModel::RunTimeSlice runs it."""
    _fields_ = [("start", AtmStart), ("loop", AtmLoop), ("tropopause_equator", C.c_int), ("coast_correction", C.c_int),
                ("checkpoint", C.c_int), ("CO2", C.c_double), ("co2_factor", C.c_double), ("ua", C.c_double), ("va", C.c_double),
                ("wa", C.c_double), ("coa", C.c_double), ("mean_temperature", C.c_double), ("bathymetry_file", C.c_char_p),
                ("temperature_file", C.c_char_p), ("precipitation_file", C.c_char_p), ("output_path", C.c_char_p),
                ("bathymetry_name", C.c_char_p)]


class AtomDevError(RuntimeError):
    pass


_OPS = ["atm_bc_radius", "atm_bc_theta", "atm_bc_phi", "atm_bc_radius_theta_phi", "atm_value_limitation", "atm_solid_ground",
        "atm_rk3d", "atm_rk3d_limited", "atm_limit_ground_rk3d_limit", "atm_rk2d", "atm_pressure3d", "atm_pressure2d",
        "hyd_bc_radius", "hyd_bc_theta", "hyd_bc_phi", "hyd_bc_radius_theta_phi", "hyd_value_limitation", "hyd_solid_ground",
        "hyd_rk3d", "hyd_rk2d", "hyd_pressure3d", "hyd_pressure2d", "hyd_pressure_density", "atm_latent_heat", "hyd_run_data"]

_lib = None


def library_path():
    """In-tree libatomdev.so; ATOMDEV_LIBRARY points experiments at another build of it."""
    return os.environ.get("ATOMDEV_LIBRARY") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libatomdev.so")


def load_library():
    """Loads libatomdev.so; raises AtomDevError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise AtomDevError("%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(there is no CPU fallback)" % path)
    lib = C.CDLL(path)
    vp, ip, dp = C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_double)
    lib.atomdev_last_error.restype = C.c_char_p
    lib.atomdev_default_params.argtypes = [C.c_int, C.POINTER(Params)]
    lib.atomdev_default_params.restype = None
    lib.atomdev_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(Params), C.c_int, C.POINTER(vp)]
    lib.atomdev_destroy.argtypes = [vp]
    lib.atomdev_set_params.argtypes = [vp, C.POINTER(Params)]
    lib.atomdev_dims.argtypes = [vp, ip, ip, ip]
    for name in ("atomdev_upload", "atomdev_download", "atomdev_upload2d", "atomdev_download2d"):
        getattr(lib, name).argtypes = [vp, C.c_int, vp]
    for name in ("atomdev_get_columns", "atomdev_set_columns"):
        getattr(lib, name).argtypes = [vp, C.c_int, C.c_int, C.c_int, vp]
    lib.atomdev_fill.argtypes = [vp, C.c_int, C.c_double]
    lib.atomdev_sync.argtypes = [vp]
    lib.atomdev_move_data_to_new_arrays.argtypes = [vp, C.c_int]
    lib.atomdev_atm_msl_data.argtypes = [vp, C.c_int]
    lib.atomdev_atm_vegetation_distribution.argtypes = [vp, C.c_double]
    lib.atomdev_atm_two_category_ice_scheme.argtypes = [vp, dp]
    lib.atomdev_write_vtk.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_char_p]
    lib.atomdev_write_panorama_vts.argtypes = [vp, C.c_char_p]
    lib.atomdev_default_hyd_start.argtypes = [C.c_int, C.POINTER(HydStart)]
    lib.atomdev_default_hyd_start.restype = None
    lib.atomdev_hyd_read_sea_ground.argtypes = [vp, C.c_char_p]
    lib.atomdev_default_atm_loop.argtypes = [C.POINTER(AtmLoop)]
    lib.atomdev_default_atm_loop.restype = None
    lib.atomdev_atm_run_2d_loop.argtypes = [vp, C.POINTER(AtmLoop)]
    lib.atomdev_atm_run_2d_loops.argtypes = [C.POINTER(vp), C.c_int, C.POINTER(AtmLoop)]
    lib.atomdev_atm_run_3d_loop.argtypes = [vp, C.POINTER(AtmLoop), C.POINTER(AtmLoopState)]
    lib.atomdev_default_hyd_slice.argtypes = [C.c_int, C.POINTER(HydSlice)]
    lib.atomdev_default_hyd_slice.restype = None
    lib.atomdev_hyd_run_time_slice.argtypes = [vp, C.POINTER(HydSlice), dp, ip]
    lib.atomdev_hyd_run_time_slices.argtypes = [C.POINTER(vp), C.POINTER(HydSlice), C.c_int, dp, ip]
    lib.atomdev_atm_read_mountain_surface.argtypes = [vp, C.c_char_p]
    lib.atomdev_default_atm_start.argtypes = [C.POINTER(AtmStart)]
    lib.atomdev_default_atm_start.restype = None
    lib.atomdev_atm_co2_coeff.argtypes = [C.c_double] * 5
    lib.atomdev_atm_co2_coeff.restype = C.c_double
    lib.atomdev_atm_ic_cell_structure.argtypes = [vp, ip]
    lib.atomdev_atm_read_surface_temperature.argtypes = [vp, C.c_char_p]
    lib.atomdev_atm_read_surface_precipitation.argtypes = [vp, C.c_char_p]
    for name in ("atomdev_atm_bc_temperature", "atomdev_atm_bc_water_vapour", "atomdev_atm_bc_co2"):
        getattr(lib, name).argtypes = [vp, C.POINTER(AtmStart)]
    lib.atomdev_atm_ic_temperature_west_east_coast.argtypes = [vp]
    lib.atomdev_atm_restrain_temperature.argtypes = [vp, C.c_double]
    lib.atomdev_default_atm_slice.argtypes = [C.POINTER(AtmSlice)]
    lib.atomdev_default_atm_slice.restype = None
    lib.atomdev_atm_run_time_slice.argtypes = [vp, C.POINTER(AtmSlice), C.POINTER(AtmLoopState)]
    lib.atomdev_fill2d.argtypes = [vp, C.c_int, C.c_double]
    lib.atomdev_selftest_math.argtypes = [vp, C.c_int, C.c_size_t, dp, dp, dp]
    lib.atomdev_selftest_raise_kernel_error.argtypes = [vp]
    lib.atomdev_selftest_deferred_file_order.argtypes = [vp, C.c_char_p]
    lib.atomdev_atm_calculate_mean_temperature.argtypes = [vp, C.POINTER(C.c_float)]
    ip = C.POINTER(C.c_int)
    lib.atomdev_slabs_local_km.argtypes = [C.c_int, C.c_int, C.c_int]
    lib.atomdev_slabs_columns.argtypes = [C.c_int, C.c_int, C.c_int, ip, ip, ip]
    lib.atomdev_slabs_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    lib.atomdev_slabs_destroy.argtypes = [vp]
    lib.atomdev_slabs_begin_exchange.argtypes = [vp, ip, C.c_int]
    lib.atomdev_slabs_finish_exchange.argtypes = [vp]
    lib.atomdev_slabs_residuum3d.argtypes = [vp, C.POINTER(C.c_double), ip]
    lib.atomdev_slabs_steady3d.argtypes = [vp, C.POINTER(C.c_double), ip]
    for name in ("atomdev_pack_columns", "atomdev_unpack_columns"):
        getattr(lib, name).argtypes = [vp, ip, C.c_int, C.c_int, C.c_int, vp]
    lib.atomdev_residuum3d_field.argtypes = [vp]
    lib.atomdev_residuum3d_suffix.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, dp]
    lib.atomdev_steady3d_part.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, dp, dp]
    lib.atomdev_atm_bc_pressure.argtypes = [vp]
    lib.atomdev_hyd_read_surface_salinity.argtypes = [vp, C.c_char_p]
    lib.atomdev_hyd_ic_ekman_spiral.argtypes = [vp, C.c_int]
    lib.atomdev_hyd_ic_circumpolar_current.argtypes = [vp, C.c_int]
    lib.atomdev_hyd_bc_temperature_salinity.argtypes = [vp, C.POINTER(HydStart)]
    lib.atomdev_atm_write_plot_data.argtypes = [vp, C.c_char_p]
    lib.atomdev_hyd_write_plot_data.argtypes = [vp, C.c_char_p]
    lib.atomdev_atm_ice_water_saturation_adjustment.argtypes = [vp, dp]
    lib.atomdev_default_radiation.argtypes = [C.POINTER(Radiation)]
    lib.atomdev_default_radiation.restype = None
    lib.atomdev_atm_radiation_multi_layer.argtypes = [vp, C.POINTER(Radiation), ip]
    lib.atomdev_rk2d_batch.argtypes = [C.POINTER(vp), C.c_int]
    lib.atomdev_transfer_atm_to_hyd.argtypes = [vp, vp]
    lib.atomdev_atm_write_transfer_file.argtypes = [vp, C.c_char_p]
    lib.atomdev_hyd_read_transfer_file.argtypes = [vp, C.c_char_p]
    for op in _OPS:
        getattr(lib, "atomdev_" + op).argtypes = [vp]
    for name in ("atomdev_atm_residuum3d", "atomdev_atm_residuum3d_noalloc", "atomdev_atm_residuum2d", "atomdev_hyd_residuum3d"):
        getattr(lib, name).argtypes = [vp, dp, ip]
    for name in ("atomdev_atm_steady3d", "atomdev_atm_steady2d", "atomdev_hyd_steady3d"):
        getattr(lib, name).argtypes = [vp, dp, ip]
    for name in ("atomdev_minmax3d", "atomdev_minmax2d"):
        getattr(lib, name).argtypes = [vp, C.c_int, dp, ip]
    lib.atomdev_timer_start.argtypes = [vp]
    lib.atomdev_timer_stop.argtypes = [vp, C.POINTER(C.c_float)]
    lib.atomdev_profile_rk.argtypes = [vp, C.c_int]
    lib.atomdev_set_rk_variant.argtypes = [vp, C.c_int]
    lib.atomdev_get_rk_variant.argtypes = [vp]
    lib.atomdev_profile_read.argtypes = [vp, C.POINTER(C.c_float), ip]
    _lib = lib
    return lib


def default_params(model, im=41, jm=181, km=361):
    """Reference defaults (param.py); on grids other than 41x181x361 the steps follow the grid."""
    p = Params()
    load_library().atomdev_default_params(model, C.byref(p))
    if (im, jm, km) != (41, 181, 361):
        p.dr = 1.0 / (im - 1)
        p.dthe = np.pi / (jm - 1)
        p.dphi = p.dthe
    return p


def atm_run_2d_loops(models, loop):
    """atomdev_atm_run_2d_loops: the surface loops of several atmosphere handles in lockstep."""
    hs = (C.c_void_p * len(models))(*[m._h for m in models])
    models[0]._check(models[0].lib.atomdev_atm_run_2d_loops(hs, len(models), C.byref(loop)))


def hyd_run_time_slices(models, files, queries=True, **counts):
    """atomdev_hyd_run_time_slices: the time slices of several hydrosphere handles in lockstep.  files: one
    (bathymetry_file, salinity_file or None, transfer_file or None, Ma) per model; counts as in run_time_slice."""
    n = len(models)
    lib = models[0].lib
    slices = (HydSlice * n)()
    keep = []
    for m, (bath, salt, transfer, Ma) in enumerate(files):
        lib.atomdev_default_hyd_slice(models[m].im, C.byref(slices[m]))
        for k, v in counts.items():
            setattr(slices[m], k, v)
        slices[m].start.Ma = Ma
        keep += [os.fsencode(bath), os.fsencode(salt) if salt else None, os.fsencode(transfer) if transfer else None]
        slices[m].bathymetry_file, slices[m].salinity_file, slices[m].transfer_file = keep[-3:]
        slices[m].queries = 1 if queries else 0
    hs = (C.c_void_p * n)(*[m._h for m in models])
    res = (C.c_double * (2 * n))()
    loc = (C.c_int * (3 * n))()
    models[0]._check(lib.atomdev_hyd_run_time_slices(hs, slices, n, res, loc))
    return [(res[2 * m], res[2 * m + 1], tuple(loc[3 * m:3 * m + 3])) for m in range(n)]


def rk2d_batch(models):
    """The surface sweeps (atm_rk2d / hyd_rk2d) of several models on one GPU in one launch."""
    lib = load_library()
    arr = (C.c_void_p * len(models))(*[m._h for m in models])
    if lib.atomdev_rk2d_batch(arr, len(models)) != 0:
        raise AtomDevError(lib.atomdev_last_error().decode())


class DeviceModel:
    """One atmosphere or hydrosphere field set resident in HBM."""

    def __init__(self, model, im, jm, km, params=None, device=-1):
        self.lib = load_library()
        self.model, self.im, self.jm, self.km = model, im, jm, km
        self.params = params if params is not None else default_params(model, im, jm, km)
        self._h = C.c_void_p()
        self._check(self.lib.atomdev_create(model, im, jm, km, C.byref(self.params), device, C.byref(self._h)))

    def _check(self, rc):
        if rc != 0:
            raise AtomDevError(self.lib.atomdev_last_error().decode())

    # ---- field store -----------------------------------------------------
    def upload(self, name, array):
        a = np.ascontiguousarray(array, dtype=np.float64)
        if name in F2:
            if a.shape != (self.jm, self.km):
                raise AtomDevError("upload %s: shape %s != %s" % (name, a.shape, (self.jm, self.km)))
            self._check(self.lib.atomdev_upload2d(self._h, F2[name], a.ctypes.data_as(C.c_void_p)))
        else:
            if a.shape != (self.im, self.jm, self.km):
                raise AtomDevError("upload %s: shape %s != %s" % (name, a.shape, (self.im, self.jm, self.km)))
            self._check(self.lib.atomdev_upload(self._h, F3[name], a.ctypes.data_as(C.c_void_p)))

    def download(self, name):
        if name in F2:
            a = np.empty((self.jm, self.km))
            self._check(self.lib.atomdev_download2d(self._h, F2[name], a.ctypes.data_as(C.c_void_p)))
        else:
            a = np.empty((self.im, self.jm, self.km))
            self._check(self.lib.atomdev_download(self._h, F3[name], a.ctypes.data_as(C.c_void_p)))
        return a

    def upload_all(self, fields3d, fields2d=None):
        for k, v in fields3d.items():
            self.upload(k, v)
        for k, v in (fields2d or {}).items():
            self.upload(k, v)

    @staticmethod
    def _buffer_pointer(buf, nbytes):
        """Address of a dense fp64 buffer: a numpy array or anything with data_ptr() (a torch tensor, host or device)."""
        if isinstance(buf, np.ndarray):
            if buf.dtype != np.float64 or not buf.flags["C_CONTIGUOUS"] or buf.nbytes != nbytes:
                raise AtomDevError("column buffer must be C-contiguous float64 of %d bytes" % nbytes)
            return buf.ctypes.data_as(C.c_void_p)
        if not buf.is_contiguous() or buf.element_size() != 8 or buf.numel() * 8 != nbytes:
            raise AtomDevError("column buffer must be a contiguous 8-byte tensor of %d bytes" % nbytes)
        return C.c_void_p(buf.data_ptr())

    def get_columns(self, name, k0, nk, out=None):
        """Columns k0..k0+nk-1 of a 3-D field as an (im, jm, nk) block (halo send side of a phi-slab split)."""
        if out is None:
            out = np.empty((self.im, self.jm, nk))
        ptr = self._buffer_pointer(out, self.im * self.jm * nk * 8)
        self._check(self.lib.atomdev_get_columns(self._h, F3[name], k0, nk, ptr))
        return out

    def set_columns(self, name, k0, block):
        """Overwrites columns k0.. of a 3-D field with an (im, jm, nk) block (halo receive side)."""
        nk = int(block.shape[-1])
        ptr = self._buffer_pointer(block, self.im * self.jm * nk * 8)
        self._check(self.lib.atomdev_set_columns(self._h, F3[name], k0, nk, ptr))

    def set_params(self, params):
        """Replaces the scalars of the model; the fields stay."""
        self._check(self.lib.atomdev_set_params(self._h, C.byref(params)))
        self.params = params

    def fill(self, name, value):
        self._check(self.lib.atomdev_fill(self._h, F3[name], float(value)))

    def sync(self):
        self._check(self.lib.atomdev_sync(self._h))

    # ---- operators ---------------------------------------------------------
    def op(self, name):
        """One reference operator, e.g. 'atm_rk3d', 'hyd_pressure3d'."""
        self._check(getattr(self.lib, "atomdev_" + name)(self._h))

    def atm_loop(self, i_trop=None, radiation=None, queries=True, **counts):
        """An atomdev_atm_loop with the reference's defaults; counts: velocity_iter_max[_2D], pressure_iter_max[_2D]."""
        L = AtmLoop()
        self.lib.atomdev_default_atm_loop(C.byref(L))
        for k, v in counts.items():
            setattr(L, k, v)
        if radiation is not None:
            L.radiation = radiation
        if i_trop is not None:
            self._i_trop = np.ascontiguousarray(i_trop, dtype=np.int32)
            if self._i_trop.size != self.jm:
                raise AtomDevError("i_trop must have jm entries")
            L.i_trop = self._i_trop.ctypes.data_as(C.POINTER(C.c_int))
            L.column_physics = 1
        L.queries = 1 if queries else 0
        return L

    def run_2d_loop(self, loop):
        """cAtmosphereModel::run_2D_loop on the device."""
        self._check(self.lib.atomdev_atm_run_2d_loop(self._h, C.byref(loop)))

    def run_3d_loop(self, loop, state=None):
        """cAtmosphereModel::run_3D_loop on the device; returns the carried state (AtmLoopState)."""
        state = state or AtmLoopState()
        self._check(self.lib.atomdev_atm_run_3d_loop(self._h, C.byref(loop), C.byref(state)))
        return state

    def read_mountain_surface(self, path):
        """BC_Bathymetry_Atmosphere::BC_MountainSurface: h and Topography from the topography file."""
        self._check(self.lib.atomdev_atm_read_mountain_surface(self._h, os.fsencode(path)))

    def ic_cell_structure(self, im_tropopause):
        """BC_Thermo::IC_CellStructure; im_tropopause: tropopause level of every row (jm ints)."""
        it = np.ascontiguousarray(im_tropopause, dtype=np.int32)
        if it.shape != (self.jm,):
            raise AtomDevError("im_tropopause must have jm entries")
        self._check(self.lib.atomdev_atm_ic_cell_structure(self._h, it.ctypes.data_as(C.POINTER(C.c_int))))

    def read_surface_temperature(self, path):
        """BC_Thermo::BC_Surface_Temperature_NASA: t at the sea level and temperature_NASA from the file."""
        self._check(self.lib.atomdev_atm_read_surface_temperature(self._h, os.fsencode(path)))

    def read_surface_precipitation(self, path):
        """BC_Thermo::BC_Surface_Precipitation_NASA: precipitation_NASA from the file."""
        self._check(self.lib.atomdev_atm_read_surface_precipitation(self._h, os.fsencode(path)))

    def bc_temperature(self, start):
        """BC_Thermo::BC_Temperature; start: AtmStart with im_tropopause set."""
        self._check(self.lib.atomdev_atm_bc_temperature(self._h, C.byref(start)))

    def ic_temperature_west_east_coast(self):
        self._check(self.lib.atomdev_atm_ic_temperature_west_east_coast(self._h))

    def bc_pressure(self):
        self._check(self.lib.atomdev_atm_bc_pressure(self._h))

    def bc_water_vapour(self, start):
        self._check(self.lib.atomdev_atm_bc_water_vapour(self._h, C.byref(start)))

    def bc_co2(self, start):
        self._check(self.lib.atomdev_atm_bc_co2(self._h, C.byref(start)))

    def pack_columns(self, names, k0, nk, buf):
        """atomdev_pack_columns: columns k0 .. k0+nk-1 of the named 3-D fields into buf, a dense (len(names), im, jm, nk)
        float64 buffer on the host (numpy) or on the device (anything with data_ptr())."""
        ids = (C.c_int * len(names))(*[F3[n] for n in names])
        nbytes = len(names) * self.im * self.jm * nk * 8
        self._check(self.lib.atomdev_pack_columns(self._h, ids, len(names), k0, nk, self._buffer_pointer(buf, nbytes)))

    def unpack_columns(self, names, k0, buf):
        ids = (C.c_int * len(names))(*[F3[n] for n in names])
        nk = buf.shape[-1]
        nbytes = len(names) * self.im * self.jm * nk * 8
        self._check(self.lib.atomdev_unpack_columns(self._h, ids, len(names), k0, nk, self._buffer_pointer(buf, nbytes)))

    # ---- the queries of a slab of a wider grid (atom_amd/slabs.py) ----
    def residuum_field(self):
        self._check(self.lib.atomdev_residuum3d_field(self._h))

    def residuum_suffix(self, k0, nk, kg0, km_global, start):
        out = (C.c_double * 7)()
        self._check(self.lib.atomdev_residuum3d_suffix(self._h, k0, nk, kg0, km_global, float(start), out))
        return list(out)

    def steady_part(self, k0, nk, kg0, km_global):
        nf = 9 if self.model == ATM else 6
        v, p = (C.c_double * nf)(), (C.c_double * nf)()
        self._check(self.lib.atomdev_steady3d_part(self._h, k0, nk, kg0, km_global, v, p))
        return list(v), list(p)

    def selftest_math(self, kind, x, y=None):
        """atomdev_selftest_math: kind 0 exp, 1 pow, 2 x^4, 3 fourth root, 4 division by a known reciprocal, 5 division."""
        x = np.ascontiguousarray(x, dtype=np.float64)
        out = np.empty_like(x)
        dp = C.POINTER(C.c_double)
        yp = None
        if y is not None:
            y = np.ascontiguousarray(y, dtype=np.float64)
            assert y.shape == x.shape
            yp = y.ctypes.data_as(dp)
        self._check(self.lib.atomdev_selftest_math(self._h, kind, x.size, x.ctypes.data_as(dp), yp, out.ctypes.data_as(dp)))
        return out

    def calculate_mean_temperature(self):
        """cAtmosphereModel::calculate_mean_temperature of the handle's temperature field (C, a float)."""
        out = C.c_float()
        self._check(self.lib.atomdev_atm_calculate_mean_temperature(self._h, C.byref(out)))
        return out.value

    def restrain_temperature(self, mean_temperature):
        """cAtmosphereModel::restrain_temperature; mean_temperature: the curve's value for the time slice in C."""
        self._check(self.lib.atomdev_atm_restrain_temperature(self._h, mean_temperature))

    def atm_slice(self, bathymetry_file, temperature_file, precipitation_file, output_path=None, bathymetry_name=None, **values):
        """An AtmSlice at the reference's defaults; values: fields of the slice, of its start (Ma, t_cretaceous, ...) or of
        its loop (velocity_iter_max, ...) by name."""
        s = AtmSlice()
        self.lib.atomdev_default_atm_slice(C.byref(s))
        top, start, loop = {n for n, _ in AtmSlice._fields_}, {n for n, _ in AtmStart._fields_}, {n for n, _ in AtmLoop._fields_}
        for k, v in values.items():
            if k in top:
                setattr(s, k, v)
            elif k in start:
                setattr(s.start, k, v)
            elif k in loop:
                setattr(s.loop, k, v)
            else:
                raise AtomDevError("atm_slice: unknown value " + k)
        s.bathymetry_file = os.fsencode(bathymetry_file)
        s.temperature_file = os.fsencode(temperature_file)
        s.precipitation_file = os.fsencode(precipitation_file)
        s.output_path = os.fsencode(output_path) if output_path else None
        s.bathymetry_name = os.fsencode(bathymetry_name) if bathymetry_name else None
        return s

    def atm_run_time_slice(self, slice_):
        """cAtmosphereModel::RunTimeSlice on the device; returns the carried state (AtmLoopState)."""
        state = AtmLoopState()
        self._check(self.lib.atomdev_atm_run_time_slice(self._h, C.byref(slice_), C.byref(state)))
        return state

    def run_time_slice(self, bathymetry_file, salinity_file=None, transfer_file=None, queries=True, **counts):
        """cHydrosphereModel::RunTimeSlice on the device; counts: velocity_iter_max_2D, pressure_iter_max_2D,
        velocity_iter_max, pressure_iter_max, Ma, i_beg.  Returns (residuum before, after, location of the latter)."""
        s = HydSlice()
        self.lib.atomdev_default_hyd_slice(self.im, C.byref(s))
        for k, v in counts.items():
            if k in ("Ma", "i_beg"):
                setattr(s.start, k, v)
            else:
                setattr(s, k, v)
        s.bathymetry_file = os.fsencode(bathymetry_file)
        s.salinity_file = os.fsencode(salinity_file) if salinity_file else None
        s.transfer_file = os.fsencode(transfer_file) if transfer_file else None
        s.queries = 1 if queries else 0
        res = (C.c_double * 2)()
        loc = (C.c_int * 3)()
        self._check(self.lib.atomdev_hyd_run_time_slice(self._h, C.byref(s), res, loc))
        return res[0], res[1], tuple(loc)

    def read_sea_ground(self, path):
        """BC_Bathymetry_Hydrosphere::BC_SeaGround: h and Bathymetry from the bathymetry file."""
        self._check(self.lib.atomdev_hyd_read_sea_ground(self._h, os.fsencode(path)))

    def read_surface_salinity(self, path):
        """BC_Thermohalin::BC_Surface_Salinity_NASA: c at the sea level from the salinity file."""
        self._check(self.lib.atomdev_hyd_read_surface_salinity(self._h, os.fsencode(path)))

    def ic_ekman_spiral(self, i_beg=0):
        self._check(self.lib.atomdev_hyd_ic_ekman_spiral(self._h, i_beg))

    def ic_circumpolar_current(self, i_beg=0):
        self._check(self.lib.atomdev_hyd_ic_circumpolar_current(self._h, i_beg))

    def bc_temperature_salinity(self, start=None):
        if start is None:
            start = HydStart()
            self.lib.atomdev_default_hyd_start(self.im, C.byref(start))
        self._check(self.lib.atomdev_hyd_bc_temperature_salinity(self._h, C.byref(start)))

    def msl_data(self, radiation_model=1):
        """Results_MSL_Atm::run_MSL_data, the part that changes model fields."""
        self._check(self.lib.atomdev_atm_msl_data(self._h, radiation_model))

    def radiation_multi_layer(self, i_trop, radiation=None):
        """BC_Thermo::BC_Radiation_multi_layer; i_trop: tropopause level of every row (jm ints)."""
        if radiation is None:
            radiation = Radiation()
            self.lib.atomdev_default_radiation(C.byref(radiation))
        it = np.ascontiguousarray(i_trop, dtype=np.int32)
        if it.shape != (self.jm,):
            raise AtomDevError("i_trop must have jm entries")
        self._check(self.lib.atomdev_atm_radiation_multi_layer(self._h, C.byref(radiation), it.ctypes.data_as(C.POINTER(C.c_int))))

    def write_vtk(self, section, index, path, Ma=0):
        """paraview_vtk_radial / _longal / _zonal (section 0 / 1 / 2) of the handle's model at level / row / column index."""
        self._check(self.lib.atomdev_write_vtk(self._h, section, index, Ma, os.fsencode(path)))

    def write_panorama_vts(self, path):
        """paraview_panorama_vts of the handle's model: the whole grid as an ASCII .vts file."""
        self._check(self.lib.atomdev_write_panorama_vts(self._h, os.fsencode(path)))

    def write_plot_data(self, path):
        """Atmosphere_PlotData / Hydrosphere_PlotData: the surface table of the slice as text."""
        fn = self.lib.atomdev_atm_write_plot_data if self.model == 0 else self.lib.atomdev_hyd_write_plot_data
        self._check(fn(self._h, os.fsencode(path)))

    def ice_water_saturation_adjustment(self, q_v_hyp=0.0):
        """BC_Thermo::Ice_Water_Saturation_Adjustment; q_v_hyp as the previous call left it. Returns the new value."""
        st = C.c_double(q_v_hyp)
        self._check(self.lib.atomdev_atm_ice_water_saturation_adjustment(self._h, C.byref(st)))
        return st.value

    def two_category_ice_scheme(self, state=(0.0, 0.0)):
        """BC_Thermo::Two_Category_Ice_Scheme; state = (S_nuc, S_i_dep) as the previous call left them. Returns the new state."""
        st = (C.c_double * 2)(*state)
        self._check(self.lib.atomdev_atm_two_category_ice_scheme(self._h, st))
        return (st[0], st[1])

    def vegetation_distribution(self, max_precipitation):
        self._check(self.lib.atomdev_atm_vegetation_distribution(self._h, float(max_precipitation)))

    def transfer_to_hyd(self, hyd):
        """Surface velocities, temperature, evaporation and precipitation of this atmosphere into the top level of
        a hydrosphere model, rounded as the reference's 4-decimal transfer file rounds them."""
        self._check(self.lib.atomdev_transfer_atm_to_hyd(self._h, hyd._h))

    def write_transfer_file(self, path):
        self._check(self.lib.atomdev_atm_write_transfer_file(self._h, os.fsencode(path)))

    def read_transfer_file(self, path):
        self._check(self.lib.atomdev_hyd_read_transfer_file(self._h, os.fsencode(path)))

    def move_data_to_new_arrays(self, level=-1):
        self._check(self.lib.atomdev_move_data_to_new_arrays(self._h, level))

    def residuum(self, name, min0=0.0, loc0=(0, 0, 0)):
        mn = C.c_double(min0)
        loc = (C.c_int * 3)(*loc0)
        self._check(getattr(self.lib, "atomdev_" + name)(self._h, C.byref(mn), loc))
        return mn.value, tuple(loc)

    def steady(self, name, n):
        out = (C.c_double * n)()
        loc = (C.c_int * (3 * n))()
        self._check(getattr(self.lib, "atomdev_" + name)(self._h, out, loc))
        return np.array(out), np.array(loc).reshape(n, 3)

    def minmax(self, name):
        """((max, (i, j, k)), (min, (i, j, k))) of a field as MinMax_Atm / MinMax_Hyd search it."""
        out = (C.c_double * 2)()
        loc = (C.c_int * 6)()
        if name in F2:
            self._check(self.lib.atomdev_minmax2d(self._h, F2[name], out, loc))
        else:
            self._check(self.lib.atomdev_minmax3d(self._h, F3[name], out, loc))
        return (out[0], tuple(loc[0:3])), (out[1], tuple(loc[3:6]))

    def timer_start(self):
        self._check(self.lib.atomdev_timer_start(self._h))

    def timer_stop(self):
        ms = C.c_float()
        self._check(self.lib.atomdev_timer_stop(self._h, C.byref(ms)))
        return ms.value

    def set_rk_variant(self, variant):
        """0: wave-specialised RK sweep, exact form; 1: one thread per theta line; 2: the tolerance form of 0."""
        self._check(self.lib.atomdev_set_rk_variant(self._h, variant))

    def rk_variant(self):
        """atomdev_get_rk_variant: the form of the 3-D sweeps in use."""
        return self.lib.atomdev_get_rk_variant(self._h)

    def profile_rk(self, enable=True):
        """True / 1: bracket the sweep kernels; 2: the sweep with the kernel that prepares it; False: off."""
        self._check(self.lib.atomdev_profile_rk(self._h, int(enable)))

    def profile_read(self):
        """(summed RK sweep kernel milliseconds, number of sweeps) since the last read."""
        ms, n = C.c_float(), C.c_int()
        self._check(self.lib.atomdev_profile_read(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def close(self):
        if self._h:
            self.lib.atomdev_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
