/*
 * atomdev — C-ABI of the MI355X dynamical-core library (libatomdev.so).
 *
 * One handle = one model (atmosphere or hydrosphere) on one GPU: it owns every
 * field of that model as a flat fp64 array in HBM, index (i*jm + j)*km + k
 * (i = radius, j = theta, k = phi, phi fastest), plus a derived 1-byte land
 * mask.  The entry points below are what the reference's per-slice loops call
 * on its jagged host `Array`s; each comment names the reference method it
 * replaces.  All functions return 0 on success, non-zero on error (the text is
 * available from atomdev_last_error()); nothing throws across this boundary.
 * One caller thread per handle; all work of a handle is ordered on the
 * handle's own HIP stream, and only atomdev_sync / download / the query
 * functions wait for it.
 */
#ifndef ATOMDEV_H_
#define ATOMDEV_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct atomdev atomdev_t;

enum atomdev_model {
    ATOMDEV_ATM = 0, /* cAtmosphereModel fields and operators */
    ATOMDEV_HYD = 1  /* cHydrosphereModel fields and operators */
};

/* 3-D fields (im*jm*km doubles each).  Names follow the members of
 * cAtmosphereModel (atmosphere/cAtmosphereModel.h:179-228) and
 * cHydrosphereModel (hydrosphere/cHydrosphereModel.h:62-96). */
enum atomdev_field3d {
    ATOMDEV_T = 0, ATOMDEV_U, ATOMDEV_V, ATOMDEV_W, ATOMDEV_P_DYN, ATOMDEV_C,
    ATOMDEV_CLOUD, ATOMDEV_ICE, ATOMDEV_CO2,                       /* 0..8  state            */
    ATOMDEV_TN = 9, ATOMDEV_UN, ATOMDEV_VN, ATOMDEV_WN, ATOMDEV_P_DYNN, ATOMDEV_CN,
    ATOMDEV_CLOUDN, ATOMDEV_ICEN, ATOMDEV_CO2N,                    /* 9..17 "new" copies     */
    ATOMDEV_H = 18,                                                /* 1.0 = solid, else fluid */
    ATOMDEV_AUX_U = 19, ATOMDEV_AUX_V, ATOMDEV_AUX_W,
    ATOMDEV_BUOYANCY = 22,       /* BuoyancyForce (atm) / BuoyancyForce_3D (hyd) */
    ATOMDEV_S_V = 23, ATOMDEV_S_C, ATOMDEV_S_I, ATOMDEV_S_R, ATOMDEV_S_S, /* atm sources */
    ATOMDEV_SALT_BALANCE = 28, ATOMDEV_SALT_FINGER, ATOMDEV_SALT_DIFFUSION,
    ATOMDEV_R_WATER, ATOMDEV_R_SALT_WATER,                         /* hyd only */
    ATOMDEV_RESIDUUM = 33,       /* continuity residuum written by *_residuum3d/2d */
    ATOMDEV_P_STAT = 34,         /* hydrostatic pressure: bar in the hydrosphere, hPa in the atmosphere */
    ATOMDEV_Q_LATENT = 35, ATOMDEV_Q_SENSIBLE, ATOMDEV_RADIATION_3D,
    ATOMDEV_P_RAIN, ATOMDEV_P_SNOW,                                /* atm only: inputs of run_MSL_data */
    ATOMDEV_EPSILON_3D = 40,                                       /* atm only: emissivity of the radiation model */
    ATOMDEV_S_C_C = 41,                                            /* atm only: condensation rate (cloud physics) */
    ATOMDEV_N3D = 42
};

/* 2-D fields (jm*km doubles each, index j*km + k). */
enum atomdev_field2d {
    ATOMDEV_TOPOGRAPHY = 0,        /* Topography (atm) / Bathymetry (hyd), metres */
    ATOMDEV_EVAPORATION_DALTON = 1,
    ATOMDEV_PRECIPITATION = 2,
    /* atmosphere surface diagnostics written by run_MSL_data (atmosphere/cAtmosphereModel.h:206-227) */
    ATOMDEV_Q_RADIATION = 3, ATOMDEV_PRECIPITABLE_WATER, ATOMDEV_EVAPORATION_PENMAN, ATOMDEV_Q_EVAPORATION,
    ATOMDEV_Q_LATENT_2D, ATOMDEV_Q_SENSIBLE_2D, ATOMDEV_Q_BOTTOM, ATOMDEV_CO2_TOTAL,
    ATOMDEV_VEGETATION = 11,
    ATOMDEV_ALBEDO = 12, ATOMDEV_EPSILON, ATOMDEV_RADIATION_SURFACE,   /* written by the radiation model */
    /* hydrosphere surface diagnostics written by Results_Hyd::run_data (hydrosphere/cHydrosphereModel.h) */
    ATOMDEV_UPWELLING = 15, ATOMDEV_DOWNWELLING, ATOMDEV_BOTTOM_WATER, ATOMDEV_SALT_FINGER_2D,
    ATOMDEV_SALT_DIFFUSION_2D, ATOMDEV_BUOYANCY_FORCE_2D, ATOMDEV_SALT_TOTAL,
    /* atmosphere: surface data read once per time slice and the Celsius copy the radial VTK writer keeps of it */
    ATOMDEV_TEMPERATURE_NASA = 22, ATOMDEV_PRECIPITATION_NASA, ATOMDEV_TEMP_NASA_CELSIUS,
    ATOMDEV_N2D = 25
};

/* Scalars the hot path reads.  Names and meaning are those of the reference's
 * parameters (param.py:9-178) and grid constants
 * (atmosphere/cAtmosphereModel.cpp:38-55, hydrosphere/cHydrosphereModel.cpp:145-156). */
typedef struct atomdev_params {
    double dt, dr, dthe, dphi;  /* step sizes                                     */
    double r0, the0, phi0;      /* coordinate origins; rad/the/phi are built by
                                   repeated addition like Array_1D::Coordinates   */
    double re, pr, g, Buoyancy, u_0, t_0, c_0;
    /* atmosphere */
    double sc_WaterVapour, sc_CO2, gam, r_air, R_Air, R_WaterVapour, L_atm, cp_l,
           t_tropopause;
    /* hydrosphere */
    double sc, L_hyd, r_0_water, ta, pa, ca;
    /* both models */
    double p_0;                 /* reference pressure in hPa */
    /* atmosphere, surface fluxes (Results_MSL_Atm): gas-constant ratio, vapour pressure at 0 C in hPa,
       latent heats of evaporation and sublimation in J/kg */
    double ep, hp, lv, ls;
    double r_water_vapour;      /* density of saturated water vapour in kg/m3 (param.py:82) */
    double co2_0;               /* preindustrial CO2 in ppm (param.py:95): scale of the CO2 column in the VTK files */
} atomdev_params;

/* Fills *p with the reference defaults for `model` on the 1-degree grid. */
void atomdev_default_params(int model, atomdev_params* p);

const char* atomdev_last_error(void);

/* Field store: replaces lib/Array.{h,cpp} + Array_2D + Array_1D for this path.
 * device < 0 selects the current HIP device. */
int atomdev_create(int model, int im, int jm, int km, const atomdev_params* p,
                   int device, atomdev_t** out);
int atomdev_destroy(atomdev_t* h);
/* Replaces the scalars of an existing handle (fields stay as they are).  The reference passes its
 * scalars with every operator call (e.g. u_0, r_air to Pressure_Atm::computePressure_3D,
 * atmosphere/Pressure_Atm.cpp:39-41); a caller that mirrors those signatures sets them here. */
int atomdev_set_params(atomdev_t* h, const atomdev_params* p);
int atomdev_dims(const atomdev_t* h, int* im, int* jm, int* km);
int atomdev_upload(atomdev_t* h, int field3d, const double* host);
int atomdev_download(atomdev_t* h, int field3d, double* host);
int atomdev_upload2d(atomdev_t* h, int field2d, const double* host);
int atomdev_download2d(atomdev_t* h, int field2d, double* host);
int atomdev_fill(atomdev_t* h, int field3d, double value); /* Array::initArray */
int atomdev_fill2d(atomdev_t* h, int field2d, double value); /* Array_2D::initArray_2D */
/* Columns k0 .. k0+nk-1 (all i, j) of a 3-D field to / from a dense buffer of im*jm rows of nk
 * doubles, in host or device memory.  This is the halo transfer of a phi-slab decomposition
 * (one slab of longitudes per GPU); the reference is single-domain and has no counterpart, the
 * operators it would feed are Pressure_Atm.cpp:140-155 (k+-1, k+-2 reads) and the phi-seam fills
 * BC_Atm.cpp:115-150.  Both calls wait for the handle's stream. */
int atomdev_get_columns(atomdev_t* h, int field3d, int k0, int nk, void* dst);
int atomdev_set_columns(atomdev_t* h, int field3d, int k0, int nk, const void* src);
int atomdev_sync(atomdev_t* h);

/* AtomUtils::move_data_to_new_arrays (lib/Utils.cpp:41-68), coeff = 1.
 * level < 0: the 3-D lists (atm: u v w t p_dyn c cloud ice co2; hyd: u v w t
 * p_dyn c), every level.  level >= 0: the 2-D lists (v w p_dyn) at that level. */
int atomdev_move_data_to_new_arrays(atomdev_t* h, int level);

/* ---- atmosphere ---- */
/* BC_Atmosphere::BC_radius / BC_theta / BC_phi (atmosphere/BC_Atm.cpp:39-150) */
int atomdev_atm_bc_radius(atomdev_t* h);
int atomdev_atm_bc_theta(atomdev_t* h);
int atomdev_atm_bc_phi(atomdev_t* h);
/* The three of them in the order the 3-D loop calls them (atmosphere/cAtmosphereModel.cpp:782-784) as one launch:
 * every boundary point works out what it would have read from the earlier fills itself.  Same bits as the three calls. */
int atomdev_atm_bc_radius_theta_phi(atomdev_t* h);
/* BC_Thermo::Value_Limitation_Atm (atmosphere/BC_Thermo.cpp:3235-3275) */
int atomdev_atm_value_limitation(atomdev_t* h);
/* BC_Bathymetry_Atmosphere::BC_SolidGround (atmosphere/BC_Bath_Atm.cpp:94-119) */
int atomdev_atm_solid_ground(atomdev_t* h);
/* RungeKutta_Atmosphere::solveRungeKutta_3D_Atmosphere + RHS_Atmosphere::
 * RK_RHS_3D_Atmosphere (atmosphere/RungeKutta_Atm.cpp:32-156, RHS_Atm.cpp:68-493) */
int atomdev_atm_rk3d(atomdev_t* h);
/* atomdev_atm_rk3d and the atomdev_atm_value_limitation that cAtmosphereModel::run_3D_loop calls right after it
 * (atmosphere/cAtmosphereModel.cpp:809-825) as one call: every field ends with the bits the two calls leave.  The
 * sweep stores its results already limited (it goes on computing with the unlimited ones, as it does when the limiter
 * runs afterwards), and one small kernel limits the points the sweep does not store and clears p_dyn on land. */
int atomdev_atm_rk3d_limited(atomdev_t* h);
/* The four calls that cAtmosphereModel::run_3D_loop makes in a row - Value_Limitation_Atm, BC_SolidGround,
 * solveRungeKutta_3D_Atmosphere, Value_Limitation_Atm (atmosphere/cAtmosphereModel.cpp:791, 793, 809, 825) - as one:
 * every field ends with the bits atomdev_atm_value_limitation, atomdev_atm_solid_ground, atomdev_atm_rk3d and
 * atomdev_atm_value_limitation leave.  The first limiter runs in the kernel that prepares the sweep's per-point
 * inputs (BC_SolidGround after it writes what the limiter's land rule has just written), the last one in the sweep's
 * stores (outside the sweep's points the first limiter has left it nothing to do).  With the handle's default form of
 * the sweep (rk_variant 2) the first limiter touches memory on the six faces only and the sweep passes the old values
 * of its own points - which it overwrites anyway - through the limiter as it loads them: same bits, one pass over the
 * fields less.  ATOMDEV_NO_LOAD_LIMIT=1 in the environment keeps the pass over all points. */
int atomdev_atm_limit_ground_rk3d_limit(atomdev_t* h);
/* solveRungeKutta_2D_Atmosphere + RK_RHS_2D_Atmosphere
 * (atmosphere/RungeKutta_Atm.cpp:163-217, RHS_Atm.cpp:498-678) */
int atomdev_atm_rk2d(atomdev_t* h);
/* Pressure_Atm::computePressure_3D / _2D (atmosphere/Pressure_Atm.cpp:39-301) */
int atomdev_atm_pressure3d(atomdev_t* h);
int atomdev_atm_pressure2d(atomdev_t* h);
/* Accuracy_Atm::residuumQuery_3D / _2D (atmosphere/Accuracy_Atm.cpp:54-116).
 * *min and loc carry the object's state (min, i_res, j_res, k_res) in and out,
 * as the reference reuses one object for the before/after calls.  The residuum
 * field is left in ATOMDEV_RESIDUUM. loc = {i, j, k} ({0, j, k} for 2-D). */
int atomdev_atm_residuum3d(atomdev_t* h, double* min, int loc[3]);
int atomdev_atm_residuum2d(atomdev_t* h, double* min, int loc[3]);
/* atomdev_atm_residuum3d for callers that must not allocate device memory while they run (tight serving loops,
 * harnesses that forbid hipMalloc; the call still synchronises the handle's stream to hand its result back, so it is
 * not itself something to record into a HIP graph).  No counterpart in the reference.  The query's
 * intermediate results (the compact residuum sequence, its per-chunk summaries, the result words) live in a
 * static array of the library's device image, 20 MiB per GPU for as long as the process lives, instead of the
 * handle's reduction buffer that atomdev_atm_residuum3d grows on demand; the result comes back through page-locked
 * memory the handle received at atomdev_create.  The call allocates and frees nothing.  Grids with more than
 * 2 621 440 interior points ((im-2)(jm-2)(km-2); the 41 x 181 x 361 grid has 2 506 179) are refused with an error.
 * Same kernels, same results as atomdev_atm_residuum3d.  All handles of one GPU share the array: calls that
 * overlap in time on one GPU (two host threads with a handle each) race; calls one after the other do not,
 * whatever the earlier call left in the array. */
int atomdev_atm_residuum3d_noalloc(atomdev_t* h, double* min, int loc[3]);
/* Accuracy_Atm::steadyQuery_3D (atmosphere/Accuracy_Atm.cpp:192-337): max |x - xn|
 * and its first location, order u v w t c cloud ice co2 p_dyn; loc = 9 x {i,j,k}. */
int atomdev_atm_steady3d(atomdev_t* h, double out[9], int loc[27]);
/* Accuracy_Atm::steadyQuery_2D (:118-177) on level 0: order v w p_dyn. */
int atomdev_atm_steady2d(atomdev_t* h, double out[3], int loc[9]);

/* Results_MSL_Atm::run_MSL_data (atmosphere/Results_Atm.cpp:118-353), the part that changes model
 * fields (the printed averages that follow it are not produced): surface pressure, evaporation after
 * Dalton and Penman and the heat fluxes at the top of the topography and on the sea surface of every
 * column (writes p_stat there, the 2-D fields Q_radiation [if RadiationModel >= 1], Q_Evaporation,
 * Q_latent, Q_sensible, Q_bottom, Evaporation_Dalton, Evaporation_Penman); boundary values of
 * BuoyancyForce; c, cloud, ice, P_rain, P_snow clamped at zero; co2_total, precipitable_water and
 * Precipitation.  Reads t, c, v, w, Q_Latent, Q_Sensible, radiation_3D, co2, h. */
int atomdev_atm_msl_data(atomdev_t* h, int RadiationModel);
/* Scalars of BC_Thermo::BC_Radiation_multi_layer; names as in param.py:60-70.  co2_coeff is what the
 * function derives from the model's CO2 switch (BC_Thermo.cpp:252-257): co2_factor * (co2_equator /
 * co2_tropopause) if CO2 == 1, else 1. */
typedef struct atomdev_radiation {
    double albedo_pole, albedo_equator, epsilon_pole, epsilon_equator, epsilon_tropopause, rad_pole, rad_equator;
    double co2_coeff, sigma;
} atomdev_radiation;
void atomdev_default_radiation(atomdev_radiation* r);
/* BC_Thermo::BC_Radiation_multi_layer (atmosphere/BC_Thermo.cpp:166-424): albedo, emissivity and long-wave
 * radiation of every column and four rounds of the multi-layer balance that relaxes t between the top of the
 * topography and the tropopause.  i_trop[j] (jm entries, host memory) is the tropopause level of row j as the
 * function computes it: im_tropopause[j] + GetTropopauseHightAdd(t_cretaceous / t_0) (:226, :3299-3303).
 * Reads p_stat, h; changes t, c, cloud, ice (zero clamps below the tropopause), epsilon_3D, radiation_3D and
 * the 2-D fields albedo, epsilon, radiation_surface. */
int atomdev_atm_radiation_multi_layer(atomdev_t* h, const atomdev_radiation* r, const int* i_trop);
/* BC_Thermo::Two_Category_Ice_Scheme (atmosphere/BC_Thermo.cpp:2803-3143): rain and snow by the two-category
 * ice scheme - one preparatory and five full passes over the grid, each integrating P_rain and P_snow down
 * every column from the sources S_r, S_s.  Reads t, h, S_c_c; writes S_v, S_c, S_i, S_r, S_s, P_rain, P_snow,
 * clamps c, cloud, ice at zero and zeroes S_c_c inside the ground.  The reference keeps S_nuc and S_i_dep as
 * class members that are assigned only under conditions, so a grid point may inherit them from the point
 * visited before it (k outer, j, i downwards), from the previous pass or from the previous call:
 * state = {S_nuc, S_i_dep} carries the two members in and out. */
int atomdev_atm_two_category_ice_scheme(atomdev_t* h, double state[2]);

/* BC_Thermo::Ice_Water_Saturation_Adjustment (atmosphere/BC_Thermo.cpp:2509-2796): water vapour, cloud water and
 * cloud ice brought into saturation equilibrium at every grid point (up to 20 iterations each), t changed by the
 * latent heat; writes c, cloud, ice, their new-level twins, t and S_c_c.  The class member q_v_hyp is assigned before
 * a cold point's iteration only where its cloud water and cloud ice come in non-negative (:2658-2661) and otherwise
 * keeps what the point visited before left behind: *q_v_hyp is the member, read on entry and written on return. */
int atomdev_atm_ice_water_saturation_adjustment(atomdev_t* h, double* q_v_hyp);
/* BC_Thermo::Latent_Heat (atmosphere/BC_Thermo.cpp:2418-2502): Q_Latent and Q_Sensible from the radial
 * gradients of c, ice and t, one-sided at the top of the topography (i_topography, :128-139), central above it
 * up to level im-3.  Reads t, c, ice, p_stat, h.  A column whose topography reaches level im-2 is left as it is
 * (the reference reads level im, outside its arrays, there). */
int atomdev_atm_latent_heat(atomdev_t* h);
/* Results_Hyd::run_data (hydrosphere/Results_Hyd.cpp:102-320) without its printout: Ekman-layer integrals of v and w
 * (Simpson's rule over the array as the reference indexes it: entries 0..i_max-i_Ekman of a buffer that is filled at
 * i_Ekman..im-2), BottomWater / Upwelling / Downwelling from them, the column sums SaltFinger, SaltDiffusion,
 * BuoyancyForce_2D, Salt_total over the upper half, and the boundary values (r, then theta, then phi) of
 * BuoyancyForce_3D, Salt_Finger, Salt_Diffusion and Salt_Balance.  The phi term of BottomWater is the reference's
 * (aux_w[j][k+1] - aux_w[j][k+1]) divided by a member the class never sets: it is taken as the +0 it is for any
 * finite positive value of that member. */
int atomdev_hyd_run_data(atomdev_t* h);

/* PostProcess_Atmosphere::Atmosphere_PlotData (atmosphere/PostProcess_Atm.cpp:624-652) and
 * PostProcess_Hydrosphere::Hydrosphere_PlotData (hydrosphere/PostProcess_Hyd.cpp:676-705): the surface table of a
 * time slice as text, fixed notation with 4 decimals, one row per (k, j).  path is the whole file name (the reference
 * builds it as output_path + "/[" + name + "]_PlotData_Atm" + ("_" + iter_cnt if iter_cnt > 0) + ".xyz"). */
int atomdev_atm_write_plot_data(atomdev_t* h, const char* path);
int atomdev_hyd_write_plot_data(atomdev_t* h, const char* path);

/* The ASCII VTK sections the models write for ParaView: PostProcess_Atmosphere::paraview_vtk_radial / _longal /
 * _zonal (atmosphere/PostProcess_Atm.cpp:240-621) and PostProcess_Hydrosphere::paraview_vtk_longal / _radial /
 * _zonal (hydrosphere/PostProcess_Hyd.cpp:373-636), chosen by the handle's model and `section`: the level i, the
 * row j or the column k = index.  path is the whole file name (the reference builds it as output_path + "/[" +
 * name + "]_Atm_radial_" + index + "_" + n + ".vtk" and alike).  The reference's writers use aux_u / aux_v / aux_w
 * (and, radial atmosphere, the 2-D Celsius copy of temperature_NASA) as scratch for derived columns; the same
 * values are left in the same places here.  Ma is read by the atmosphere's radial section only (:287). */
enum atomdev_section { ATOMDEV_RADIAL = 0, ATOMDEV_LONGAL = 1, ATOMDEV_ZONAL = 2 };
int atomdev_write_vtk(atomdev_t* h, int section, int index, int Ma, const char* path);

/* PostProcess_Atmosphere::paraview_panorama_vts (atmosphere/PostProcess_Atm.cpp:120-238) and
 * PostProcess_Hydrosphere::paraview_panorama_vts (hydrosphere/PostProcess_Hyd.cpp:269-366), chosen by the handle's
 * model: the whole grid as an ASCII .vts file (the models write it only when their paraview_panorama_vts flag is set;
 * about 150 bytes per grid point).  path is the whole file name (the reference builds it as output_path + "/[" + name +
 * "]_Atm_panorama_" + n + ".vts"). */
int atomdev_write_panorama_vts(atomdev_t* h, const char* path);

/* BC_Bathymetry_Atmosphere::BC_MountainSurface (atmosphere/BC_Bath_Atm.cpp:39-90): reads the topography file (lon lat
 * height per line, jm x km lines, longitudes -180..180), builds h (1 = solid up to the level of the height) and
 * Topography (metres, 0 over the sea), and moves both to 0..360.  A file with fewer than jm x km entries is an error
 * (the reference aborts); heights above L_atm fill the column (the reference writes beyond its array there).  Host
 * work; the result is uploaded as ATOMDEV_H and ATOMDEV_TOPOGRAPHY. */
int atomdev_atm_read_mountain_surface(atomdev_t* h, const char* path);
/* cAtmosphereModel::run_2D_loop / run_3D_loop (atmosphere/cAtmosphereModel.cpp:660-735, 743-898) without their printing,
 * as one call each: the entry points above in the reference's order, nothing uploaded or downloaded in between.
 * column_physics != 0 adds BC_Thermo's Ice_Water_Saturation_Adjustment and Two_Category_Ice_Scheme (every second
 * velocity iteration), BC_Radiation_multi_layer (when RadiationModel == 1, :828-831), Latent_Heat and the vegetation
 * distribution where run_3D_loop calls them - with it the loop is the reference's; without it, the flow solver alone.
 * The class members these carry from call to call travel in atomdev_atm_loop_state, zero at the start of a time slice.
 * Both loops end after the pressure iteration that takes the count of velocity iterations beyond nm (:731-735, 891-895).  queries != 0 runs the residuum and steady-state queries where the reference does; the state receives the
 * last pair of residuum values (one Accuracy_Atm object: the second query starts from the first one's result). */
typedef struct atomdev_atm_loop {
    int velocity_iter_max_2D, pressure_iter_max_2D, velocity_iter_max, pressure_iter_max;  /* param.py:33-36: 2, 10, 2, 2 */
    int column_physics;
    int RadiationModel;            /* argument of run_MSL_data (1) */
    atomdev_radiation radiation;   /* constants of BC_Radiation_multi_layer */
    const int* i_trop;             /* jm tropopause levels (cAtmosphereModel::get_tropopause), needed with column_physics */
    int queries;
    int nm;                        /* most velocity iterations of a loop (cAtmosphereModel.h:123: 200) */
} atomdev_atm_loop;
typedef struct atomdev_atm_loop_state {
    double q_v_hyp, S_nuc, S_i_dep;   /* members of BC_Thermo carried between its calls */
    double residuum[2];               /* continuity residuum before and after the last velocity iteration */
    int residuum_loc[3];
} atomdev_atm_loop_state;
void atomdev_default_atm_loop(atomdev_atm_loop* L);
int atomdev_atm_run_2d_loop(atomdev_t* h, const atomdev_atm_loop* L);
/* The surface loops of n independent time slices (one atmosphere handle each, same surface size) in lockstep, their
 * sweeps sharing one launch (atomdev_rk2d_batch).  No counterpart in the reference; same bits as calls of their own. */
int atomdev_atm_run_2d_loops(atomdev_t* const* handles, int n, const atomdev_atm_loop* L);
int atomdev_atm_run_3d_loop(atomdev_t* h, const atomdev_atm_loop* L, atomdev_atm_loop_state* state);

/* BC_Bathymetry_Atmosphere::vegetationDistribution (atmosphere/BC_Bath_Atm.cpp:129-142): Vegetation =
 * Precipitation / max_Precipitation on land surface cells that are not colder than t = 1, else 0. */
int atomdev_atm_vegetation_distribution(atomdev_t* h, double max_Precipitation);

/* ---- the atmosphere's start of a time slice (atmosphere/cAtmosphereModel.cpp:220-302) ---- */

/* What BC_Thermo's constructor takes from the model beyond the scalars of struct atomdev_params - atmosphere/BC_Thermo.cpp:38-94,
 * defaults param.py:45-58, 108-130 - and what it derives from the model's time list and temperature curve. */
typedef struct atomdev_atm_start {
    int Ma;                          /* time slice in million years */
    int first_time_slice;            /* cAtmosphereModel::is_first_time_slice() */
    int Ma_previous;                 /* the time slice before this one (read unless first_time_slice) */
    int sun, RadiationModel, NASATemperature, declination, sun_position_lat, sun_position_lon;
    double t_cretaceous;             /* mean temperature (Ma) - mean temperature (0) of the curve, in C (:93-94) */
    double t_cretaceous_add;         /* (mean temperature (Ma) - mean temperature (previous Ma)) / t_0; 0 on the first slice (:446-451) */
    double t_average, t_equator, t_pole, t_land;
    double c_tropopause, c_ocean, c_land;
    double co2_average, co2_equator, co2_tropopause, co2_pole, co2_cretaceous, co2_vegetation, co2_ocean, co2_land;
    const int* im_tropopause;        /* jm tropopause levels (BC_Thermo::TropopauseLocation, :884-932: tropopause_equator everywhere) */
} atomdev_atm_start;
/* Defaults of param.py for the first slice at Ma = 0; im_tropopause is left NULL. */
void atomdev_default_atm_start(atomdev_atm_start* s);

/* BC_Thermo::IC_CellStructure (atmosphere/BC_Thermo.cpp:934-2307): u, v, w of the Hadley, Ferrel and polar cells - vertical
 * profiles on thirteen rows of the 1-degree grid (jm must be 181), linear interpolation between them in the reference's
 * order (the interpolation loops rewrite their own end rows, later loops read the rewritten values), seven-row smoothing
 * around the cell borders, division by u_0, zero on land.  Works in place on whatever u, v, w hold, like the reference
 * (its caller passes the reset values ua, va, wa). */
int atomdev_atm_ic_cell_structure(atomdev_t* h, const int* im_tropopause);
/* BC_Thermo::BC_Surface_Temperature_NASA (:2310-2350): surface temperature in C (lat lon value per line, km x jm lines,
 * j fastest) into t at level 0 and into temperature_NASA as (value + t_0) / t_0; the column k = (km-1)/2 is replaced by the
 * mean of its neighbours.  BC_Surface_Precipitation_NASA (:2352-2385): the same file layout into precipitation_NASA. */
int atomdev_atm_read_surface_temperature(atomdev_t* h, const char* path);
int atomdev_atm_read_surface_precipitation(atomdev_t* h, const char* path);
/* BC_Thermo::BC_Temperature (:427-676): surface temperature on the topography (parabola from pole to pole on land, the
 * file's values over the ocean, both with the increments of a paleo time slice), linear decay up to the tropopause,
 * constant above. */
int atomdev_atm_bc_temperature(atomdev_t* h, const atomdev_atm_start* s);
/* BC_Thermo::IC_Temperature_WestEastCoast (:3151-3233): surface temperature held constant over seven cells off every
 * coast, four order-dependent scans.  The model calls it when NASATemperature == 1, Ma > 0 and the plate reconstruction
 * is off (cAtmosphereModel.cpp:276-279). */
int atomdev_atm_ic_temperature_west_east_coast(atomdev_t* h);
/* BC_Thermo::BC_Pressure (:2387-2407): p_stat from the surface temperature by the polytropic formula. */
int atomdev_atm_bc_pressure(atomdev_t* h);
/* BC_Thermo::BC_WaterVapour (:679-780): saturation vapour on the topography reduced by c_ocean / c_land plus Dalton's
 * evaporation over the ocean (on land the value the last ocean point visited before, in (k, j) order, left behind),
 * parabolic decrease up to the tropopause. */
int atomdev_atm_bc_water_vapour(atomdev_t* h, const atomdev_atm_start* s);
/* BC_Thermo::BC_CO2 (:788-881): CO2 on the topography from the pole-to-pole parabola, vegetation and the paleo increment,
 * parabolic decrease up to the tropopause.  The reference divides its members co2_equator, co2_tropopause, ... by co2_0 in
 * this call and its radiation model later works with the divided values: see atomdev_atm_co2_coeff. */
int atomdev_atm_bc_co2(atomdev_t* h, const atomdev_atm_start* s);
/* co2_coeff of atomdev_radiation as BC_Radiation_multi_layer derives it (:251-256) after BC_CO2 has run on the same
 * BC_Thermo object: co2_factor * ((co2_equator / co2_0) / (co2_tropopause / co2_0)) if CO2 == 1, else 1. */
double atomdev_atm_co2_coeff(double CO2, double co2_factor, double co2_equator, double co2_tropopause, double co2_0);

/* cAtmosphereModel::restrain_temperature (atmosphere/cAtmosphereModel.cpp:952-1006): the sea-level temperature shifted so
 * that its mean, weighted with the cosine of the latitude of the 1-degree rows and rounded to float like the reference's,
 * meets mean_temperature (get_mean_temperature_from_curve of the time slice, in C), then capped at 40 C.  The mean is
 * summed on the host in the reference's order. */
int atomdev_atm_restrain_temperature(atomdev_t* h, double mean_temperature);
/* cAtmosphereModel::calculate_mean_temperature (atmosphere/cAtmosphereModel.cpp:952-988) of the handle's temperature
 * field: that weighted mean of the sea-level row, in C, as the float the reference returns. */
int atomdev_atm_calculate_mean_temperature(atomdev_t* h, float* mean);

/* cAtmosphereModel::RunTimeSlice (atmosphere/cAtmosphereModel.cpp:140-341) without its printing and without the plate
 * reconstruction script (:177-189; the caller names the surface files): reset_arrays, the topography, the initial
 * conditions above in the reference's order, BC_Radiation_multi_layer when RadiationModel == 1, the surface loop, the
 * volume loop with BC_Thermo's column physics, restrain_temperature, the output files - every step the entry point that
 * stands for it, nothing uploaded or downloaded in between. */
typedef struct atomdev_atm_slice {
    atomdev_atm_start start;         /* im_tropopause NULL: tropopause_equator on every row (BC_Thermo::TropopauseLocation) */
    atomdev_atm_loop loop;           /* i_trop, radiation.co2_coeff and column_physics are not read: the slice derives the
                                      * first two (levels + the paleo shift of :226; atomdev_atm_co2_coeff) and runs the physics */
    int tropopause_equator;          /* param.py:58 */
    int coast_correction;            /* IC_Temperature_WestEastCoast; the model: NASATemperature == 1 && Ma > 0 && no plate reconstruction (:276) */
    int checkpoint;                  /* output files also after every checkpoint-th pressure iteration (:886-888, param.py:37); 0: never */
    double CO2, co2_factor;          /* param.py:41, 130 */
    double ua, va, wa, coa;          /* reset values beyond ta, pa, ca of struct atomdev_params - param.py:97-103 */
    double mean_temperature;         /* mean temperature (Ma) of the model's curve in C, for restrain_temperature */
    const char* bathymetry_file;     /* bathymetry_path + "/" + bathymetry_name */
    const char* temperature_file;
    const char* precipitation_file;
    const char* output_path;         /* NULL: no output files */
    const char* bathymetry_name;     /* "<Ma>Ma_Golonka.xyz": part of every output file's name (write_file, :599-652) */
} atomdev_atm_slice;
/* Defaults of param.py for the first slice at Ma = 0; the five strings are left NULL. */
void atomdev_default_atm_slice(atomdev_atm_slice* s);
int atomdev_atm_run_time_slice(atomdev_t* h, const atomdev_atm_slice* s, atomdev_atm_loop_state* state);

/* nk columns from k0 of several 3-D fields as one dense message [nf][im][jm][nk] (halo exchange between slabs).  dst / src
 * may be host memory (copied through a staging buffer of the handle; the call returns when the buffer is complete) or
 * device memory (packed by a kernel on the handle's stream; the call returns at once and atomdev_sync orders it).  No
 * counterpart in the reference. */
int atomdev_pack_columns(atomdev_t* h, const int* fields3d, int nf, int k0, int nk, void* dst);
int atomdev_unpack_columns(atomdev_t* h, const int* fields3d, int nf, int k0, int nk, const void* src);

/* The accuracy queries of a model that holds a slab of longitudes - the columns [kg0, kg0 + nk) of a grid with km_global
 * columns, stored at its local columns [k0, k0 + nk) - for runs that split one model over several GPUs.  No counterpart in the
 * reference, which has one domain.
 * atomdev_residuum3d_field: the continuity residuum at every interior local point into ATOMDEV_RESIDUUM (w one column
 *   beyond each end of the slab must be valid).
 * atomdev_residuum3d_suffix: what the slab contributes to one step of the reference's order-dependent scan over the whole
 *   grid's lexicographic sequence (Accuracy_Atm.cpp:82-116): over its points at global sequence position >= from,
 *   out = { largest |res| (-1: none), its first position, res there, first position with res >= 0 (-1: none), res there,
 *   last position whose res is no NaN (-1: none), res there }.
 * atomdev_steady3d_part: max |x - xn| of every field of the steady query (9 for the atmosphere, 6 for the hydrosphere, in the
 *   order of atomdev_*_steady3d) over the slab's columns, and its position (i jm + j) km_global + k - the first one for the
 *   atmosphere, the last one for the hydrosphere, as the two reference classes compare.  A value of -1: no point. */
int atomdev_residuum3d_field(atomdev_t* h);
int atomdev_residuum3d_suffix(atomdev_t* h, int k0, int nk, int kg0, int km_global, double from, double out[7]);
int atomdev_steady3d_part(atomdev_t* h, int k0, int nk, int kg0, int km_global, double* values, double* positions);

/* ---- one model split into phi-slabs over several handles of ONE process (one handle per GPU) ----
 * No counterpart in the reference, which has one domain.  Slab r of n holds, as an ordinary smaller model of
 * km = atomdev_slabs_local_km(km_global, n, r) columns: [5 copies of the columns behind the phi seam: last slab only]
 * [3 ghost columns of its left neighbour] [its owned columns] [3 ghost columns of its right neighbour] [5 seam copies:
 * slab 0 only], and runs the unmodified operators on it; atomdev_slabs_columns names the global column of every local
 * one (for scattering the fields and the mask) and the owned range.  The 3-D Runge-Kutta sweeps, the limiters and the
 * r / theta / phi fills need no exchange (no phi coupling: RHS_Atm.cpp:146-149, RHS_Hyd.cpp:136-139); the pressure step
 * reads p_dynn and aux_u / aux_v / aux_w at the ghost and seam columns (Pressure_Atm.cpp:57-66, 140-155, 170-176), the
 * continuity residuum reads w (Accuracy_Atm.cpp:82-116).
 *   atomdev_slabs_begin_exchange: the owners' columns of the named fields set out for the slabs that hold copies of them -
 *     on a second stream per handle, behind an event on its compute stream (so: call it right after the operator that
 *     makes the fields final), packed by one kernel per message and carried by hipMemcpyPeerAsync between persistent
 *     device buffers.  Returns at once; up to four exchanges may be in flight.
 *   atomdev_slabs_finish_exchange: every holder's compute stream waits (on the device: hipStreamWaitEvent, the host goes
 *     on) for its messages and stores them into its ghost and seam columns.  Call it just before the operator that reads.
 *   atomdev_slabs_residuum3d / _steady3d: the accuracy queries of the whole grid (Accuracy_Atm.cpp:82-116, 192-337,
 *     Accuracy_Hyd.cpp:88-184) with the single domain's values and locations; *min carries the atmosphere's scan state
 *     from call to call as atomdev_atm_residuum3d does; values: 9 (atmosphere) or 6 numbers, loc: 3 per value.
 * A C++ caller with one process per node drives all its GPUs this way; one process per GPU: atom_amd/slabs.py. */
typedef struct atomdev_slabs atomdev_slabs_t;
int atomdev_slabs_local_km(int km_global, int world, int rank);   /* -1: no such plan */
int atomdev_slabs_columns(int km_global, int world, int rank, int* global_columns, int* owned_first_local, int* owned_count);
int atomdev_slabs_create(atomdev_t* const* handles, int n, int km_global, atomdev_slabs_t** out);
int atomdev_slabs_destroy(atomdev_slabs_t* g);
int atomdev_slabs_begin_exchange(atomdev_slabs_t* g, const int* fields3d, int nf);
int atomdev_slabs_finish_exchange(atomdev_slabs_t* g);
int atomdev_slabs_residuum3d(atomdev_slabs_t* g, double* min, int loc[3]);
int atomdev_slabs_steady3d(atomdev_slabs_t* g, double* values, int* loc);

/* The library's arithmetic helpers evaluated on the device for n caller-supplied operands (host pointers): the
 * double-double exp / pow / x^4 / fourth root the column physics uses in place of the device library's functions, and the
 * sweeps' division by a known reciprocal beside the hardware's IEEE division.  For property tests; no counterpart in the
 * reference. */
enum atomdev_selftest {
    ATOMDEV_SELFTEST_EXP = 0, ATOMDEV_SELFTEST_POW, ATOMDEV_SELFTEST_POW4, ATOMDEV_SELFTEST_ROOT4,
    ATOMDEV_SELFTEST_DIV_BY,   /* x / y as the sweeps divide: x * (1/y) and two fused corrections */
    ATOMDEV_SELFTEST_DIVIDE    /* x / y */
};
int atomdev_selftest_math(atomdev_t* h, int kind, size_t n, const double* x, const double* y, double* out);
/* Enqueues a kernel that sets the word by which a kernel reports a failure it detects itself (the hydrosphere sweep's
 * bounded wait between workgroups): the next call that waits for the handle's stream - downloads, queries, file
 * writers, atomdev_sync - must fail once, and the call after it succeed.  For tests of that path. */
int atomdev_selftest_raise_kernel_error(atomdev_t* h);
/* Hands the library's deferred file writer (the threads that finish a slice's text files while the slice goes on) two
 * jobs for one path, the first of them slow: jobs for one path must run in the order given, so the file holds the second
 * job's text afterwards ("second\n").  For tests of that path; needs no GPU work. */
int atomdev_selftest_deferred_file_order(atomdev_t* h, const char* path);

/* ---- hydrosphere ---- */
/* BC_Hydrosphere::RB_radius / RB_theta / RB_phi (hydrosphere/BC_Hyd.cpp:38-158) */
int atomdev_hyd_bc_radius(atomdev_t* h);
int atomdev_hyd_bc_theta(atomdev_t* h);
int atomdev_hyd_bc_phi(atomdev_t* h);
/* The three of them in the order the 3-D loop calls them (hydrosphere/cHydrosphereModel.cpp:443-445) as one launch, like
 * atomdev_atm_bc_radius_theta_phi.  Same bits as the three calls. */
int atomdev_hyd_bc_radius_theta_phi(atomdev_t* h);
/* BC_Thermohalin::Value_Limitation_Hyd (hydrosphere/BC_Thermohalin.cpp:699-722) */
int atomdev_hyd_value_limitation(atomdev_t* h);
/* BC_Bathymetry_Hydrosphere::BC_SolidGround (hydrosphere/BC_Bath_Hyd.cpp:120-139) */
int atomdev_hyd_solid_ground(atomdev_t* h);
/* BC_Thermohalin::BC_Pressure_Density (hydrosphere/BC_Thermohalin.cpp:518-583): hydrostatic
 * pressure and the densities of water and salt water, integrated down each column from the
 * surface level im-1.  Reads t, c; writes p_stat, r_water, r_salt_water at every point. */
int atomdev_hyd_pressure_density(atomdev_t* h);

/* ---- the hydrosphere's start of a time slice (cHydrosphereModel.cpp:244-292) ---- */

/* What BC_Thermohalin's constructor receives beyond the scalars of atomdev_params: hydrosphere/BC_Thermohalin.cpp:25-82,
 * cHydrosphereModel.cpp:182-183, 268-270; defaults param.py:53-54, 168-176. */
typedef struct atomdev_hyd_start {
    int i_beg, i_max;            /* first level of the thermocline (0) and the sea level (im - 1) */
    int Ma, Ma_max, Ma_max_half; /* time slice in million years and the two constants of the Cretaceous parabola */
    double t_average, t_cretaceous_max;
} atomdev_hyd_start;
void atomdev_default_hyd_start(int im, atomdev_hyd_start* s);

/* BC_Bathymetry_Hydrosphere::BC_SeaGround (hydrosphere/BC_Bath_Hyd.cpp:31-114): reads the bathymetry file
 * (lon lat depth per line, jm x km lines, longitudes -180..180), builds h (1 = solid) and Bathymetry (metres,
 * positive down), moves both to 0..360 and removes the needles, in the reference's visiting order (the smoothing
 * is order dependent and, once the first needle was met, keeps clearing diagonal neighbours for the rest of the
 * grid).  A file with fewer than jm x km entries is an error (the reference aborts).  Host work; the result is
 * uploaded as ATOMDEV_H and ATOMDEV_TOPOGRAPHY. */
int atomdev_hyd_read_sea_ground(atomdev_t* h, const char* path);
/* BC_Thermohalin::BC_Surface_Salinity_NASA (hydrosphere/BC_Thermohalin.cpp:636-670): surface salinity in psu
 * (lon lat value per line, km x jm lines, j fastest) into c at the sea level, divided by c_0; negative values
 * leave c as it is. */
int atomdev_hyd_read_surface_salinity(atomdev_t* h, const char* path);
/* BC_Thermohalin::IC_v_w_EkmanSpiral (hydrosphere/BC_Thermohalin.cpp:88-360): the surface wind of the transfer
 * file becomes the surface current (3 %, turned by 45 degrees, band by band in latitude), then every column is
 * filled downwards from it.  The latitude bands are fixed rows of the 1-degree grid (30, 60, ... 150): jm must
 * exceed 150, as in the reference, which reads beyond its arrays otherwise.  asin / sin / cos of the surface
 * step come from the device's math library: v and w agree with the reference within 1e-10 relative. */
int atomdev_hyd_ic_ekman_spiral(atomdev_t* h, int i_beg);
/* BC_Thermohalin::IC_CircumPolar_Current (hydrosphere/BC_Thermohalin.cpp:676-690): w = 1 cm/s in the water
 * cells of rows 147..152 (jm must exceed 152) from level i_beg upwards. */
int atomdev_hyd_ic_circumpolar_current(atomdev_t* h, int i_beg);
/* BC_Thermohalin::BC_Temperature_Salinity (hydrosphere/BC_Thermohalin.cpp:367-513): surface temperature not
 * below 4 C, parabolic decrease of t and c with depth down to level i_beg, (ta, ca) below and on land. */
int atomdev_hyd_bc_temperature_salinity(atomdev_t* h, const atomdev_hyd_start* s);

/* cHydrosphereModel::RunTimeSlice (hydrosphere/cHydrosphereModel.cpp:234-610) without its printing and its output
 * files: transfer file, sea ground, initial conditions, pressure_iter_max_2D x velocity_iter_max_2D surface iterations,
 * pressure_iter_max x velocity_iter_max volume iterations with Results_Hyd::run_data, each operator the entry point
 * above that stands for it, in the reference's order, nothing uploaded or downloaded in between.  transfer_file and
 * salinity_file may be NULL (the surface fields are taken as they are / the salinity is not read; the reference reads
 * it when Ma == 0).  queries != 0 runs the residuum and steady-state queries where the reference does (their results
 * are printed only there); residuum[0], residuum[1] (may be NULL) receive the continuity residuum before and after
 * the last velocity iteration, residuum_loc (may be NULL) the location of the latter.  Output files: with output_path
 * set, the three VTK sections (and the panorama file) and the PlotData table at the checkpoints and at the end, under
 * the reference's names; otherwise atomdev_write_vtk, atomdev_write_panorama_vts, atomdev_hyd_write_plot_data afterwards. */
typedef struct atomdev_hyd_slice {
    atomdev_hyd_start start;
    int velocity_iter_max_2D, pressure_iter_max_2D, velocity_iter_max, pressure_iter_max;  /* param.py:135-138: 2, 10, 2, 2 */
    const char* transfer_file;    /* output_path/[<Ma>Ma_Golonka.xyz]_Transfer_Atm.vw */
    const char* bathymetry_file;  /* bathymetry_path/<Ma>Ma_Golonka.xyz */
    const char* salinity_file;    /* Name_SurfaceSalinity_File */
    int queries;
    /* the output files of cHydrosphereModel::write_file (:739-772): after every checkpoint-th pressure iteration
     * (:612-614; param.py:139) and at the end of the slice; output_path NULL: none */
    int checkpoint;
    int paraview_panorama_vts;    /* the whole-grid .vts file as well */
    const char* output_path;
    const char* bathymetry_name;  /* "<Ma>Ma_Golonka.xyz": part of every output file's name */
} atomdev_hyd_slice;
void atomdev_default_hyd_slice(int im, atomdev_hyd_slice* s);
int atomdev_hyd_run_time_slice(atomdev_t* h, const atomdev_hyd_slice* s, double residuum[2], int residuum_loc[3]);
/* The same for n independent time slices (the slices of a paleo series, each its own handle on one device, same grid
 * and iteration counts) advanced in lockstep: every operator is enqueued for all slices, each on its handle's stream,
 * and the surface sweeps of all slices share one launch (atomdev_rk2d_batch).  residuum: 2 n values, residuum_loc:
 * 3 n (both may be NULL).  No counterpart in the reference, which runs one slice after the other; every slice ends with
 * the bits of a call of its own. */
int atomdev_hyd_run_time_slices(atomdev_t* const* handles, const atomdev_hyd_slice* slices, int n, double* residuum,
                                int* residuum_loc);

/* solveRungeKutta_3D_Hydrosphere + RK_RHS_3D_Hydrosphere
 * (hydrosphere/RungeKutta_Hyd.cpp:30-124, RHS_Hyd.cpp:60-447) */
int atomdev_hyd_rk3d(atomdev_t* h);
/* solveRungeKutta_2D_Hydrosphere + RK_RHS_2D_Hydrosphere
 * (hydrosphere/RungeKutta_Hyd.cpp:130-186, RHS_Hyd.cpp:453-653) */
int atomdev_hyd_rk2d(atomdev_t* h);
/* atomdev_atm_rk2d / atomdev_hyd_rk2d for several independent models (handles on one GPU with the same jm, km;
 * atmosphere and hydrosphere may be mixed) in one launch.  No counterpart in the reference, which computes one
 * time slice at a time: the surface sweep is a chain of dependent anti-diagonals that keeps a single workgroup
 * busy, so the sweeps of the slices of a paleo series run side by side instead of one after the other. */
int atomdev_rk2d_batch(atomdev_t* const* handles, int n);
/* Pressure_Hyd::computePressure_3D / _2D (hydrosphere/Pressure_Hyd.cpp:42-317) */
int atomdev_hyd_pressure3d(atomdev_t* h);
int atomdev_hyd_pressure2d(atomdev_t* h);
/* Accuracy_Hyd::residuumQuery_3D (hydrosphere/Accuracy_Hyd.cpp:88-116): state is
 * reset on every call and the comparison is >=. */
int atomdev_hyd_residuum3d(atomdev_t* h, double* min, int loc[3]);
/* Accuracy_Hyd::steadyQuery_3D (:121-184): order u v w t c p_dyn, >= comparison
 * (last location of the maximum); loc = 6 x {i,j,k}. */
int atomdev_hyd_steady3d(atomdev_t* h, double out[6], int loc[18]);

/* ---- extreme values (both models) ---- */
/* MinMax_Atm::searchMinMax_3D / _2D (atmosphere/MinMax_Atm.cpp:74-196) on an atmosphere handle,
 * MinMax_Hyd::searchMinMax_3D / _2D (hydrosphere/MinMax_Hyd.cpp:51-260) on a hydrosphere handle: the search
 * itself, without the unit conversion and the printout that follow it.  out = {maxValue, minValue},
 * loc = {imax, jmax, kmax, imin, jmin, kmin}, found as the reference's (j, k, i) scans with strict comparisons
 * find them: MinMax_Atm seeds both from the first element (the 2-D scan covers interior points only),
 * MinMax_Hyd seeds the maximum with 0 and the minimum with the maximum, and its 2-D minimum skips zeros and
 * is raised to 0 if negative.  The atmosphere's max_Precipitation (cAtmosphereModel.cpp:555) is
 * atomdev_minmax2d(h, ATOMDEV_PRECIPITATION, ...)'s out[0]. */
int atomdev_minmax3d(atomdev_t* h, int field3d, double out[2], int loc[6]);
int atomdev_minmax2d(atomdev_t* h, int field2d, double out[2], int loc[6]);

/* ---- atmosphere -> hydrosphere coupling ---- */
/* PostProcess_Atmosphere::Atmosphere_v_w_Transfer (atmosphere/PostProcess_Atm.cpp:98-118) followed by
 * PostProcess_Hydrosphere::Atmosphere_TransferFile_read (hydrosphere/PostProcess_Hyd.cpp:642-668), without the
 * text file between them: level 0 of v * u_0, w * u_0, t of the atmosphere and its Evaporation_Dalton and
 * Precipitation arrive in level im-1 of v, w, t and the two 2-D fields of the hydrosphere, each value as the
 * file's fixed notation with 4 decimals would have carried it; p_dyn of that level is set to 0.  The handles
 * may live on different GPUs; both streams are waited for. */
int atomdev_transfer_atm_to_hyd(atomdev_t* atm, atomdev_t* hyd);
/* The two halves with the file: the same bytes the reference writes, the same values it reads. */
int atomdev_atm_write_transfer_file(atomdev_t* atm, const char* path);
int atomdev_hyd_read_transfer_file(atomdev_t* hyd, const char* path);

/* HIP-event timing on the handle's stream (for benchmarks): everything enqueued
 * between start and stop is measured; stop waits for it. */
int atomdev_timer_start(atomdev_t* h);
int atomdev_timer_stop(atomdev_t* h, float* milliseconds);

/* Tuning knob without a reference counterpart: which kernel form the 3-D RK sweeps
 * use.  0: wave-specialised, one prognostic variable per wave, 64 lines per workgroup, every
 * operation rounded as the reference's.  1: one thread per theta line.  Both give the same
 * bits.  2: the kernel of 0 in its tolerance form - a division by a loop-invariant divisor is
 * one multiplication by its reciprocal and products feeding sums are fused - which costs up
 * to one unit in the last place per operation (measured on the 1-degree grid: every field
 * of a sweep within 3e-14 of variant 0, a whole time slice with the column physics as far from
 * the reference as with variant 0) and about a fifth less time.  3: the pipelined sweep -
 * the arithmetic of 2 restated as affine recurrences along the theta line, two waves of a
 * workgroup carrying the recurrences through LDS while eight others do all loads, stores and
 * per-row coefficients rows ahead (within 1e-12 of variant 0 per sweep; the 1-degree
 * hydrosphere time slice ends 1.5e-13 from the reference; the atmosphere's, whose column
 * physics amplifies last-place differences, 4e-10 in c, against 2e-11 with variant 2).
 * Initial value: 3 for a hydrosphere handle, 2 for an atmosphere handle (the stated bound is
 * 1e-10 per field after a time slice with identical iteration counts); the hydrosphere's
 * variants 0 and 1 carry the reference's bits, and the models' outer interfaces
 * (atom_models.hpp, atomdev_mirror.hpp) select 0 so that their results and files carry the
 * reference's bits and bytes.  The surface sweeps (atomdev_*_rk2d, atomdev_rk2d_batch) follow the
 * same knob: exact form for 0 (1: the plain one-workgroup form, same bits), tolerance form for 2
 * and 3 (1e-16 per field and sweep, 0.24 ms instead of 0.55 ms on the 1-degree surface); the handles
 * of one batch must agree on the form.  The environment variable ATOMDEV_RK_VARIANT overrides the initial
 * value and the outer interface's choice.  atomdev_get_rk_variant returns the current value. */
int atomdev_set_rk_variant(atomdev_t* h, int variant);
int atomdev_get_rk_variant(const atomdev_t* h);

/* Per-launch timing of the 3-D RK sweep kernels, without host synchronisation in
 * between: while enabled, every atomdev_atm_rk3d / atomdev_hyd_rk3d call brackets
 * its sweep kernel(s) with HIP events on the handle's stream (enable = 1), or the sweep
 * together with the kernel that prepares its per-point inputs (enable = 2; the atmosphere's
 * pressure-height term).  atomdev_profile_read waits for the stream, returns the summed
 * time and the number of sweeps since the last read, and resets both. */
int atomdev_profile_rk(atomdev_t* h, int enable);
int atomdev_profile_read(atomdev_t* h, float* total_ms, int* sweeps);

#ifdef __cplusplus
}
#endif
#endif /* ATOMDEV_H_ */
