/*
 * ref_shim.cpp — flat-array C entry points around the *compiled reference classes*.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/atom_oracle.c).  This translation unit
 * is our own code; it is linked against the reference's hot-path sources
 * compiled where they lie under /root/reference (oracle/Makefile, target _ref).
 * The result, oracle/_ref/libatomref.so, is used to pin atom_oracle.c bit for
 * bit, to generate tests/golden/*.npz, and optionally as the CPU baseline.
 *
 * Every op copies the bound flat arrays into the reference's jagged Array
 * objects, calls the reference method, and copies everything back.
 */
#include "ref_shim.h"

extern "C" void ref_read_mask(void* v, const char* xyz_path);

extern "C" {

void* ref_create(int model, int im, int jm, int km, const atomdev_params* p) {
    RefCtx* c = new RefCtx();
    c->model = model; c->im = im; c->jm = jm; c->km = km; c->p = *p;
    for (int n = 0; n < ATOMDEV_N3D; n++) { c->f[n] = nullptr; c->a[n] = new Array(im, jm, km, 0.); }
    for (int n = 0; n < ATOMDEV_N2D; n++) { c->g[n] = nullptr; c->b[n] = new Array_2D(jm, km, 0.); }
    for (int n = 0; n < 8; n++) c->rhs[n].initArray(im, jm, km, 0.);
    // RK_RHS_2D_Hydrosphere indexes rad at 40 whatever the grid is
    int nrad = im > 41 ? im : 41;
    c->rad.initArray_1D(nrad, 0.); c->the.initArray_1D(jm, 0.); c->phi.initArray_1D(km, 0.);
    c->rad.Coordinates(nrad, p->r0, p->dr);
    c->the.Coordinates(jm, p->the0, p->dthe);
    c->phi.Coordinates(km, p->phi0, p->dphi);
    return c;
}

void ref_destroy(void* v) {
    RefCtx* c = (RefCtx*)v;
    if (!c) return;
    for (int n = 0; n < ATOMDEV_N3D; n++) delete c->a[n];
    for (int n = 0; n < ATOMDEV_N2D; n++) delete c->b[n];
    delete c;
}

void ref_bind3d(void* v, int field, double* ptr) { ((RefCtx*)v)->f[field] = ptr; }
void ref_bind2d(void* v, int field, double* ptr) { ((RefCtx*)v)->g[field] = ptr; }

void ref_atm_rk3d(void* v) {
    RefCtx* c = (RefCtx*)v; CoutSilencer s; copy_in(c);
    const atomdev_params& P = c->p;
    RHS_Atmosphere prepare(c->im, c->jm, c->km, P.dt, P.dr, P.dthe, P.dphi, P.re, P.sc_WaterVapour, P.sc_CO2,
                           P.g, P.pr, 1., P.Buoyancy, 1., P.gam, 5.670280e-8, 0.0262);
    RungeKutta_Atmosphere result(c->im, c->jm, c->km, P.dt, P.dr, P.dphi, P.dthe);
    int n = 1;
    // arguments the RHS never reads (lv, ls, ep, hp, c_0, co2_0, p_0, densities, R_co2,
    // p_stat, Latency, Q_Sensible, P_rain, P_snow, S_c_c) get reference defaults / spare arrays
    Array spare(c->im, c->jm, c->km, 0.);
    result.solveRungeKutta_3D_Atmosphere(prepare, n, 2.52e6, 2.83e6, 0.623, 6.1078, P.u_0, P.t_0, P.c_0, 280.0,
        1013.25, P.r_air, 0.0094, 0.0019767, P.L_atm, P.cp_l, P.R_Air, P.R_WaterVapour, 188.91,
        c->rad, c->the, c->phi, c->rhs[0], c->rhs[1], c->rhs[2], c->rhs[3], c->rhs[4], c->rhs[5], c->rhs[6], c->rhs[7],
        A(H), A(T), A(U), A(V), A(W), A(P_DYN), spare, A(C), A(CLOUD), A(ICE), A(CO2),
        A(TN), A(UN), A(VN), A(WN), A(P_DYNN), A(CN), A(CLOUDN), A(ICEN), A(CO2N),
        A(AUX_U), A(AUX_V), A(AUX_W), spare, A(BUOYANCY), spare, spare, spare,
        A(S_V), A(S_C), A(S_I), A(S_R), A(S_S), spare, B(TOPOGRAPHY), B(EVAPORATION_DALTON), B(PRECIPITATION));
    copy_out(c);
}

void ref_atm_rk2d(void* v) {
    RefCtx* c = (RefCtx*)v; CoutSilencer s; copy_in(c);
    const atomdev_params& P = c->p;
    RHS_Atmosphere prepare_2D(c->jm, c->km, P.dthe, P.dphi, P.re);
    RungeKutta_Atmosphere result(c->im, c->jm, c->km, P.dt, P.dr, P.dphi, P.dthe);
    int n = 1;
    result.solveRungeKutta_2D_Atmosphere(prepare_2D, n, P.r_air, P.u_0, 1013.25, P.L_atm, c->rad, c->the, c->phi,
        c->rhs[2], c->rhs[3], A(H), A(V), A(W), A(P_DYN), A(VN), A(WN), A(P_DYNN), A(AUX_V), A(AUX_W));
    copy_out(c);
}

void ref_atm_pressure3d(void* v) {
    RefCtx* c = (RefCtx*)v; CoutSilencer s; copy_in(c);
    Pressure_Atm startPressure(c->im, c->jm, c->km, c->p.dr, c->p.dthe, c->p.dphi);
    startPressure.computePressure_3D(c->p.u_0, c->p.r_air, c->rad, c->the, A(P_DYN), A(P_DYNN), A(H), A(AUX_U), A(AUX_V), A(AUX_W));
    copy_out(c);
}

void ref_atm_pressure2d(void* v) {
    RefCtx* c = (RefCtx*)v; CoutSilencer s; copy_in(c);
    Pressure_Atm startPressure(c->im, c->jm, c->km, c->p.dr, c->p.dthe, c->p.dphi);
    startPressure.computePressure_2D(c->p.u_0, c->p.r_air, c->rad, c->the, A(P_DYN), A(P_DYNN), A(H), A(AUX_V), A(AUX_W));
    copy_out(c);
}

static void atm_bc(void* v, int which) {
    RefCtx* c = (RefCtx*)v; CoutSilencer s; copy_in(c);
    BC_Atmosphere boundary(c->im, c->jm, c->km, c->p.t_tropopause);
    if (which == 0) boundary.BC_radius(A(T), A(U), A(V), A(W), A(P_DYN), A(C), A(CLOUD), A(ICE), A(CO2));
    if (which == 1) boundary.BC_theta(A(T), A(U), A(V), A(W), A(P_DYN), A(C), A(CLOUD), A(ICE), A(CO2));
    if (which == 2) boundary.BC_phi(A(T), A(U), A(V), A(W), A(P_DYN), A(C), A(CLOUD), A(ICE), A(CO2));
    copy_out(c);
}
void ref_atm_bc_radius(void* v) { atm_bc(v, 0); }
void ref_atm_bc_theta(void* v) { atm_bc(v, 1); }
void ref_atm_bc_phi(void* v) { atm_bc(v, 2); }

void ref_atm_solid_ground(void* v) {
    RefCtx* c = (RefCtx*)v; CoutSilencer s; copy_in(c);
    BC_Bathymetry_Atmosphere LandArea(1, c->im, c->jm, c->km, 3.0, 0., 0.);
    Array spare(c->im, c->jm, c->km, 0.);
    Array_2D spare2(c->jm, c->km, 0.);
    // only h and the seven reset fields are touched (BC_Bath_Atm.cpp:101-118)
    LandArea.BC_SolidGround(1, 0, c->p.g, 6.1078, 0.623, c->p.r_air, c->p.R_Air, c->p.t_0, c->p.c_0, 0., 0., 1.0842,
        0.945, c->p.t_tropopause, 0.64, 0.001, 280.0, 330.0, 320.0, 320.0, 0., c->p.gam, 5.670280e-8,
        A(H), A(U), A(V), A(W), A(T), A(P_DYN), A(C), A(CLOUD), A(ICE), A(CO2), spare, spare2);
    copy_out(c);
}

void ref_move_data_to_new_arrays(void* v, int level) {
    RefCtx* c = (RefCtx*)v; copy_in(c);
    std::vector<Array*> o3 = { &A(U), &A(V), &A(W), &A(T), &A(P_DYN), &A(C) };
    std::vector<Array*> n3 = { &A(UN), &A(VN), &A(WN), &A(TN), &A(P_DYNN), &A(CN) };
    if (c->model == ATOMDEV_ATM) {
        o3.push_back(&A(CLOUD)); o3.push_back(&A(ICE)); o3.push_back(&A(CO2));
        n3.push_back(&A(CLOUDN)); n3.push_back(&A(ICEN)); n3.push_back(&A(CO2N));
    }
    std::vector<Array*> o2 = { &A(V), &A(W), &A(P_DYN) };
    std::vector<Array*> n2 = { &A(VN), &A(WN), &A(P_DYNN) };
    if (level < 0) AtomUtils::move_data_to_new_arrays(c->im, c->jm, c->km, 1., o3, n3);
    else AtomUtils::move_data_to_new_arrays(c->jm, c->km, 1., o2, n2, level);
    copy_out(c);
}

void ref_atm_residuum3d(void* v, double* min, int loc[3]) {
    RefCtx* c = (RefCtx*)v; CoutSilencer s; copy_in(c);
    // The object's private state (min, i_res, j_res, k_res) starts at zero; a call
    // with carried-in state is reproduced by the caller (tests use a fresh object
    // for the first call and two consecutive calls on one object for the second).
    Accuracy_Atm acc(c->im, c->jm, c->km, c->p.dr, c->p.dthe, c->p.dphi);
    Vector3D<> res(c->im, c->jm, c->km, 0.);
    auto r = acc.residuumQuery_3D(c->rad, c->the, A(U), A(V), A(W), res);
    *min = std::get<0>(r); loc[0] = std::get<1>(r); loc[1] = std::get<2>(r); loc[2] = std::get<3>(r);
    if (c->f[ATOMDEV_RESIDUUM])
        for (int i = 0; i < c->im; i++) for (int j = 0; j < c->jm; j++) for (int k = 0; k < c->km; k++)
            c->f[ATOMDEV_RESIDUUM][((size_t)i * c->jm + j) * c->km + k] = res(i, j, k);
}

/* Two consecutive queries on ONE Accuracy_Atm object, with the fields of the
 * second call taken from a second set of u, v, w (bound as UN, VN, WN). */
void ref_atm_residuum3d_twice(void* v, double min[2], int loc[6]) {
    RefCtx* c = (RefCtx*)v; CoutSilencer s; copy_in(c);
    Accuracy_Atm acc(c->im, c->jm, c->km, c->p.dr, c->p.dthe, c->p.dphi);
    Vector3D<> res(c->im, c->jm, c->km, 0.);
    auto r = acc.residuumQuery_3D(c->rad, c->the, A(U), A(V), A(W), res);
    min[0] = std::get<0>(r); loc[0] = std::get<1>(r); loc[1] = std::get<2>(r); loc[2] = std::get<3>(r);
    r = acc.residuumQuery_3D(c->rad, c->the, A(UN), A(VN), A(WN), res);
    min[1] = std::get<0>(r); loc[3] = std::get<1>(r); loc[4] = std::get<2>(r); loc[5] = std::get<3>(r);
}

void ref_atm_residuum2d(void* v, double* min, int loc[3]) {
    RefCtx* c = (RefCtx*)v; CoutSilencer s; copy_in(c);
    Accuracy_Atm acc(c->im, c->jm, c->km, c->p.dthe, c->p.dphi);
    Vector3D<> res(c->im, c->jm, c->km, 0.);
    auto r = acc.residuumQuery_2D(c->rad, c->the, A(V), A(W), res);
    *min = std::get<0>(r); loc[0] = 0; loc[1] = std::get<1>(r); loc[2] = std::get<2>(r);
    if (c->f[ATOMDEV_RESIDUUM])
        for (int j = 0; j < c->jm; j++) for (int k = 0; k < c->km; k++)
            c->f[ATOMDEV_RESIDUUM][(size_t)j * c->km + k] = res(0, j, k);
}

/* Results_MSL_Atm::run_MSL_data prints values at (j, k) = (39, 346), (123, 151), (90, 180)
 * (Results_Atm.cpp:319-326): the grid must contain them (jm > 123, km > 346). */
int ref_atm_msl_data(void* v, int RadiationModel) {
    RefCtx* c = (RefCtx*)v; CoutSilencer s;
    if (c->jm <= 123 || c->km <= 346) return 1;
    copy_in(c);
    const atomdev_params& P = c->p;
    Results_MSL_Atm calculate_MSL(c->im, c->jm, c->km, 0, P.g, P.ep, P.hp, P.u_0, P.p_0, P.t_0, P.c_0, 280.0, 5.670280e-8, 0.15,
                                  P.lv, P.ls, P.cp_l, P.L_atm, P.dt, P.dr, P.dthe, P.dphi, P.r_air, P.R_Air, 0.0094,
                                  P.R_WaterVapour, 3.0, 0., 3., P.gam, -4.0, 0., 15.0);
    double t_cretaceous = 0.;
    Array spare(c->im, c->jm, c->km, 0.);
    Array_2D spare2(c->jm, c->km, 0.), nasa_t(c->jm, c->km, 1.), nasa_p(c->jm, c->km, 0.);
    calculate_MSL.run_MSL_data(1, 2, RadiationModel, t_cretaceous, c->rad, c->the, c->phi, A(H), A(C), A(CN), A(CO2), A(CO2N),
        A(T), A(TN), A(P_DYN), A(P_STAT), A(BUOYANCY), A(U), A(V), A(W), A(Q_LATENT), A(Q_SENSIBLE), A(RADIATION_3D),
        A(CLOUD), A(CLOUDN), A(ICE), A(ICEN), A(P_RAIN), A(P_SNOW), A(AUX_U), A(AUX_V), A(AUX_W), nasa_t, nasa_p,
        B(PRECIPITABLE_WATER), B(Q_RADIATION), B(Q_EVAPORATION), B(Q_LATENT_2D), B(Q_SENSIBLE_2D), B(Q_BOTTOM),
        B(EVAPORATION_PENMAN), B(EVAPORATION_DALTON), spare2, spare2, B(CO2_TOTAL), B(PRECIPITATION),
        A(S_V), A(S_C), A(S_I), A(S_R), A(S_S), spare);
    copy_out(c);
    return 0;
}

void ref_atm_vegetation_distribution(void* v, double max_Precipitation) {
    RefCtx* c = (RefCtx*)v; CoutSilencer s; copy_in(c);
    BC_Bathymetry_Atmosphere LandArea(1, c->im, c->jm, c->km, 3.0, 0., 0.);
    LandArea.vegetationDistribution(max_Precipitation, B(PRECIPITATION), B(VEGETATION), A(T), A(H));
    copy_out(c);
}

/* ---- hydrosphere ---- */

void ref_hyd_rk3d(void* v) {
    RefCtx* c = (RefCtx*)v; CoutSilencer s; copy_in(c);
    const atomdev_params& P = c->p;
    RHS_Hydrosphere prepare(c->im, c->jm, c->km, P.dt, P.dr, P.dthe, P.dphi, P.re, P.sc, P.g, P.pr, P.Buoyancy);
    RungeKutta_Hydrosphere result(c->im, c->jm, c->km, P.dt);
    int n = 1;
    Array spare(c->im, c->jm, c->km, 0.);
    result.solveRungeKutta_3D_Hydrosphere(prepare, n, P.L_hyd, P.g, 4182.0, P.u_0, P.t_0, P.c_0, P.r_0_water, P.ta, P.pa, P.ca,
        c->rad, c->the, c->phi, B(EVAPORATION_DALTON), B(PRECIPITATION), A(H),
        c->rhs[0], c->rhs[1], c->rhs[2], c->rhs[3], c->rhs[4],
        A(T), A(U), A(V), A(W), A(P_DYN), A(C), A(TN), A(UN), A(VN), A(WN), A(P_DYNN), A(CN),
        A(AUX_U), A(AUX_V), A(AUX_W), A(SALT_FINGER), A(SALT_DIFFUSION), A(BUOYANCY), A(SALT_BALANCE), spare,
        A(R_WATER), A(R_SALT_WATER), B(TOPOGRAPHY));
    copy_out(c);
}

void ref_hyd_rk2d(void* v) {
    RefCtx* c = (RefCtx*)v; CoutSilencer s; copy_in(c);
    const atomdev_params& P = c->p;
    RHS_Hydrosphere prepare_2D(c->jm, c->km, P.dthe, P.dphi, P.re);
    RungeKutta_Hydrosphere result(c->im, c->jm, c->km, P.dt);
    int n = 1;
    result.solveRungeKutta_2D_Hydrosphere(prepare_2D, n, P.r_0_water, c->rad, c->the, c->phi, c->rhs[2], c->rhs[3],
        A(H), A(V), A(W), A(P_DYN), A(VN), A(WN), A(P_DYNN), A(AUX_V), A(AUX_W));
    copy_out(c);
}

void ref_hyd_pressure3d(void* v) {
    RefCtx* c = (RefCtx*)v; CoutSilencer s; copy_in(c);
    Pressure_Hyd startPressure(c->im, c->jm, c->km, c->p.dr, c->p.dthe, c->p.dphi);
    startPressure.computePressure_3D(c->p.u_0, c->p.r_0_water, c->rad, c->the, A(P_DYN), A(P_DYNN), A(H), A(AUX_U), A(AUX_V), A(AUX_W));
    copy_out(c);
}

void ref_hyd_pressure2d(void* v) {
    RefCtx* c = (RefCtx*)v; CoutSilencer s; copy_in(c);
    Pressure_Hyd startPressure(c->im, c->jm, c->km, c->p.dr, c->p.dthe, c->p.dphi);
    startPressure.computePressure_2D(c->p.u_0, c->p.r_0_water, c->rad, c->the, A(P_DYN), A(P_DYNN), A(H), A(AUX_V), A(AUX_W));
    copy_out(c);
}

static void hyd_bc(void* v, int which) {
    RefCtx* c = (RefCtx*)v; CoutSilencer s; copy_in(c);
    BC_Hydrosphere boundary(c->im, c->jm, c->km);
    const atomdev_params& P = c->p;
    if (which == 0) boundary.RB_radius(P.ca, P.ta, P.pa, P.dr, c->rad, A(T), A(U), A(V), A(W), A(P_DYN), A(C));
    if (which == 1) boundary.RB_theta(P.ca, P.ta, P.pa, A(T), A(U), A(V), A(W), A(P_DYN), A(C));
    if (which == 2) boundary.RB_phi(A(T), A(U), A(V), A(W), A(P_DYN), A(C));
    copy_out(c);
}
void ref_hyd_bc_radius(void* v) { hyd_bc(v, 0); }
void ref_hyd_bc_theta(void* v) { hyd_bc(v, 1); }
void ref_hyd_bc_phi(void* v) { hyd_bc(v, 2); }

void ref_hyd_value_limitation(void* v) {
    RefCtx* c = (RefCtx*)v; CoutSilencer s; copy_in(c);
    const atomdev_params& P = c->p;
    BC_Thermohalin oceanflow(c->im, c->jm, c->km, 0, c->im - 1, 0, 300, 150, P.dr, P.g, P.r_0_water, 0., 0., 0., P.ta,
                             P.ca, P.pa, P.u_0, 1013.25, P.t_0, P.c_0, 4182.0, P.L_hyd, 15.0, 10.0, 1.1355, 1.0146, "");
    oceanflow.Value_Limitation_Hyd(A(H), A(U), A(V), A(W), A(P_DYN), A(T), A(C));
    copy_out(c);
}

void ref_hyd_pressure_density(void* v) {
    RefCtx* c = (RefCtx*)v; CoutSilencer s; copy_in(c);
    const atomdev_params& P = c->p;
    BC_Thermohalin oceanflow(c->im, c->jm, c->km, 0, c->im - 1, 0, 300, 150, P.dr, P.g, P.r_0_water, 0., 0., 0., P.ta,
                             P.ca, P.pa, P.u_0, P.p_0, P.t_0, P.c_0, 4182.0, P.L_hyd, 15.0, 10.0, 1.1355, 1.0146, "");
    oceanflow.BC_Pressure_Density(A(P_STAT), A(R_WATER), A(R_SALT_WATER), A(T), A(C), A(H));
    copy_out(c);
}

/* the hydrosphere's start of a time slice: BC_Thermohalin's initial conditions and the two readers */
static BC_Thermohalin make_oceanflow(RefCtx* c, const atomdev_hyd_start* s) {
    const atomdev_params& P = c->p;
    return BC_Thermohalin(c->im, c->jm, c->km, s->i_beg, s->i_max, s->Ma, s->Ma_max, s->Ma_max_half, P.dr, P.g, P.r_0_water, 0., 0., 0.,
                          P.ta, P.ca, P.pa, P.u_0, P.p_0, P.t_0, P.c_0, 4182.0, P.L_hyd, s->t_average, s->t_cretaceous_max, 1.1355,
                          1.0146, "");
}
void ref_hyd_ic_ekman_spiral(void* v, int i_beg) {
    RefCtx* c = (RefCtx*)v; CoutSilencer q; copy_in(c);
    atomdev_hyd_start s = { i_beg, c->im - 1, 0, 300, 150, 15., 10. };
    BC_Thermohalin oceanflow = make_oceanflow(c, &s);
    oceanflow.IC_v_w_EkmanSpiral(c->rad, c->the, A(H), A(V), A(W));
    copy_out(c);
}
void ref_hyd_ic_circumpolar_current(void* v, int i_beg) {
    RefCtx* c = (RefCtx*)v; CoutSilencer q; copy_in(c);
    atomdev_hyd_start s = { i_beg, c->im - 1, 0, 300, 150, 15., 10. };
    BC_Thermohalin oceanflow = make_oceanflow(c, &s);
    oceanflow.IC_CircumPolar_Current(A(H), A(U), A(V), A(W), A(C));
    copy_out(c);
}
void ref_hyd_bc_temperature_salinity(void* v, const atomdev_hyd_start* s) {
    RefCtx* c = (RefCtx*)v; CoutSilencer q; copy_in(c);
    BC_Thermohalin oceanflow = make_oceanflow(c, s);
    oceanflow.BC_Temperature_Salinity(A(H), A(T), A(C), A(P_DYN));
    copy_out(c);
}
int ref_hyd_read_surface_salinity(void* v, const char* path) {
    RefCtx* c = (RefCtx*)v; CoutSilencer q; copy_in(c);
    atomdev_hyd_start s = { 0, c->im - 1, 0, 300, 150, 15., 10. };
    BC_Thermohalin oceanflow = make_oceanflow(c, &s);
    oceanflow.BC_Surface_Salinity_NASA(path, A(C));
    copy_out(c);
    return 0;
}
int ref_atm_read_mountain_surface(void* v, const char* path) {
    copy_in((RefCtx*)v);
    ref_read_mask(v, path);
    return 0;
}
int ref_hyd_read_sea_ground(void* v, const char* path) {
    copy_in((RefCtx*)v);  // the other fields pass through unchanged
    ref_read_mask(v, path);
    return 0;
}

void ref_hyd_solid_ground(void* v) {
    RefCtx* c = (RefCtx*)v; CoutSilencer s; copy_in(c);
    BC_Bathymetry_Hydrosphere depth(c->im, c->jm, c->km);
    depth.BC_SolidGround(c->p.ca, c->p.ta, c->p.pa, A(H), A(T), A(U), A(V), A(W), A(P_DYN), A(C),
                         A(TN), A(UN), A(VN), A(WN), A(P_DYNN), A(CN));
    copy_out(c);
}

void ref_hyd_residuum3d(void* v, double* min, int loc[3]) {
    RefCtx* c = (RefCtx*)v; CoutSilencer s; copy_in(c);
    Accuracy_Hyd acc(c->im, c->jm, c->km, c->p.dr, c->p.dthe, c->p.dphi);
    acc.residuumQuery_3D(c->rad, c->the, A(U), A(V), A(W));
    *min = acc.out_min(); loc[0] = acc.out_i_res(); loc[1] = acc.out_j_res(); loc[2] = acc.out_k_res();
}

/* MinMax_*::searchMinMax_*: the classes expose the two values only (the locations are printed);
 * names that trigger no unit conversion (MinMax_Hyd.cpp:168-205). */
void ref_minmax(void* v, int field, int two_d, double out[2]) {
    RefCtx* c = (RefCtx*)v; CoutSilencer s; copy_in(c);
    std::string a(" max x "), b(" min x "), u("-");
    if (c->model == ATOMDEV_ATM) {
        MinMax_Atm m(c->im, c->jm, c->km);
        if (two_d) m.searchMinMax_2D(a, b, u, *c->b[field], A(H));
        else m.searchMinMax_3D(a, b, u, *c->a[field], A(H));
        out[0] = m.out_maxValue(); out[1] = m.out_minValue();
    } else {
        MinMax_Hyd m(c->im, c->jm, c->km, c->p.u_0, c->p.c_0, c->p.L_hyd);
        if (two_d) m.searchMinMax_2D(a, b, u, *c->b[field], A(H));
        else m.searchMinMax_3D(a, b, u, *c->a[field], A(H));
        out[0] = m.out_maxValue(); out[1] = m.out_minValue();
    }
}

/* The reference's own writer and reader around the file dir/[name]_Transfer_Atm.vw */
int ref_transfer_atm_to_hyd(void* va, void* vh, const char* dir) {
    RefCtx* c = (RefCtx*)va; CoutSilencer s;
    copy_in(c);
    std::string out(dir), name("bridge");
    PostProcess_Atmosphere ppa(c->im, c->jm, c->km, out);
    ppa.Atmosphere_v_w_Transfer(name, c->p.u_0, A(V), A(W), A(T), A(P_DYN), B(EVAPORATION_DALTON), B(PRECIPITATION));
    c = (RefCtx*)vh;
    copy_in(c);
    PostProcess_Hydrosphere read_Transfer(c->im, c->jm, c->km, out);
    read_Transfer.Atmosphere_TransferFile_read(name, A(V), A(W), A(T), A(P_DYN), B(EVAPORATION_DALTON), B(PRECIPITATION));
    copy_out(c);
    return 0;
}

/* Results_Hyd::run_data.  The class divides by a member (dphi) that no constructor sets (Results_Hyd.h:26); the
 * numerator is x - x, so any finite positive value gives the same +0.  The object is built in storage filled with
 * the byte 0x3f, which makes that member 4.8e-4 instead of whatever the stack held. */
void ref_hyd_run_data(void* v) {
    RefCtx* c = (RefCtx*)v; CoutSilencer s; copy_in(c);
    const atomdev_params& P = c->p;
    alignas(16) static unsigned char storage[sizeof(Results_Hyd)];
    std::memset(storage, 0x3f, sizeof(storage));
    Results_Hyd* calculate_MSL = new (storage) Results_Hyd(c->im, c->jm, c->km);
    double dr = P.dr, dthe = P.dthe, L_hyd = P.L_hyd, u_0 = P.u_0, c_0 = P.c_0;
    calculate_MSL->run_data(0, dr, dthe, L_hyd, u_0, c_0, c->rad, c->the, A(H), A(U), A(V), A(W), A(C), A(SALT_BALANCE),
                            A(SALT_FINGER), A(SALT_DIFFUSION), A(BUOYANCY), B(UPWELLING), B(DOWNWELLING), B(SALT_FINGER_2D),
                            B(SALT_DIFFUSION_2D), B(BUOYANCY_FORCE_2D), B(SALT_TOTAL), B(BOTTOM_WATER));
    calculate_MSL->~Results_Hyd();
    copy_out(c);
}

/* the reference's PlotData writers; the file appears as dir/[bridge]_PlotData_{Atm,Hyd}.xyz */
int ref_write_plot_data(void* v, const char* dir) {
    RefCtx* c = (RefCtx*)v; CoutSilencer s; copy_in(c);
    std::string out(dir), name("bridge");
    double u_0 = c->p.u_0;
    if (c->model == ATOMDEV_ATM) {
        PostProcess_Atmosphere ppa(c->im, c->jm, c->km, out);
        ppa.Atmosphere_PlotData(name, -1, u_0, c->p.t_0, A(H), A(V), A(W), A(T), A(C), B(PRECIPITATION), B(PRECIPITABLE_WATER));
    } else {
        PostProcess_Hydrosphere ppa(c->im, c->jm, c->km, out);
        ppa.Hydrosphere_PlotData(name, -1, u_0, A(H), A(V), A(W), A(T), A(C), B(BOTTOM_WATER), B(UPWELLING), B(DOWNWELLING));
    }
    return 0;
}

/* the reference's panorama writers; the file appears as dir/[bridge]_{Atm,Hyd}_panorama_0.vts */
int ref_write_panorama_vts(void* v, const char* dir) {
    RefCtx* c = (RefCtx*)v; CoutSilencer s; copy_in(c);
    std::string out(dir), name("bridge");
    atomdev_params& P = c->p;
    double p_0 = P.p_0, c_0 = P.c_0;
    if (c->model == ATOMDEV_ATM) {
        PostProcess_Atmosphere w(c->im, c->jm, c->km, out);
        w.paraview_panorama_vts(name, 0, P.u_0, P.t_0, p_0, P.r_air, c_0, P.co2_0, A(H), A(T), A(P_DYN), A(P_STAT), A(BUOYANCY), A(U), A(V),
                                A(W), A(C), A(CO2), A(CLOUD), A(ICE), A(AUX_U), A(AUX_V), A(AUX_W), A(Q_LATENT), A(Q_SENSIBLE),
                                A(EPSILON_3D), A(P_RAIN), A(P_SNOW));
    } else {
        PostProcess_Hydrosphere w(c->im, c->jm, c->km, out);
        w.paraview_panorama_vts(name, 0, P.u_0, P.r_0_water, A(H), A(T), A(P_DYN), A(P_STAT), A(R_WATER), A(R_SALT_WATER), A(U), A(V), A(W),
                                A(C), A(AUX_U), A(AUX_V), A(AUX_W), A(SALT_FINGER), A(SALT_DIFFUSION), A(BUOYANCY), A(SALT_BALANCE));
    }
    return 0;
}

/* the reference's ParaView section writers; the file appears as dir/[bridge]_{Atm,Hyd}_{radial,longal,zonal}_<index>_0.vtk */
int ref_write_vtk(void* v, int section, int index, int Ma, const char* dir) {
    RefCtx* c = (RefCtx*)v; CoutSilencer s; copy_in(c);
    std::string out(dir), name("bridge");
    atomdev_params& P = c->p;
    double p_0 = P.p_0, c_0 = P.c_0;
    if (c->model == ATOMDEV_ATM) {
        PostProcess_Atmosphere w(c->im, c->jm, c->km, out);
        if (section == ATOMDEV_RADIAL)
            w.paraview_vtk_radial(name, Ma, index, 0, P.u_0, P.t_0, p_0, P.r_air, c_0, P.co2_0, A(H), A(P_DYN), A(P_STAT), A(BUOYANCY),
                                  A(T), A(U), A(V), A(W), A(C), A(CO2), A(CLOUD), A(ICE), A(AUX_U), A(AUX_V), A(AUX_W), A(RADIATION_3D),
                                  A(Q_LATENT), A(Q_SENSIBLE), A(EPSILON_3D), A(P_RAIN), A(P_SNOW), B(PRECIPITABLE_WATER), B(Q_BOTTOM),
                                  B(Q_RADIATION), B(Q_LATENT_2D), B(Q_SENSIBLE_2D), B(EVAPORATION_PENMAN), B(EVAPORATION_DALTON),
                                  B(Q_EVAPORATION), B(TEMPERATURE_NASA), B(PRECIPITATION_NASA), B(VEGETATION), B(ALBEDO), B(EPSILON),
                                  B(PRECIPITATION), B(TOPOGRAPHY), B(TEMP_NASA_CELSIUS));
        else if (section == ATOMDEV_ZONAL)
            w.paraview_vtk_zonal(name, index, 0, P.hp, P.ep, P.R_Air, P.g, P.L_atm, P.u_0, P.t_0, p_0, P.r_air, c_0, P.co2_0, A(H), A(P_DYN),
                                 A(P_STAT), A(BUOYANCY), A(T), A(U), A(V), A(W), A(C), A(CO2), A(CLOUD), A(ICE), A(AUX_U), A(AUX_V), A(AUX_W),
                                 A(Q_LATENT), A(Q_SENSIBLE), A(RADIATION_3D), A(EPSILON_3D), A(P_RAIN), A(P_SNOW), A(S_V), A(S_C), A(S_I),
                                 A(S_R), A(S_S), A(S_C_C));
        else
            w.paraview_vtk_longal(name, index, 0, P.u_0, P.t_0, p_0, P.r_air, c_0, P.co2_0, A(H), A(P_DYN), A(P_STAT), A(BUOYANCY), A(T),
                                  A(U), A(V), A(W), A(C), A(CO2), A(CLOUD), A(ICE), A(AUX_U), A(AUX_V), A(AUX_W), A(Q_LATENT), A(Q_SENSIBLE),
                                  A(EPSILON_3D), A(P_RAIN), A(P_SNOW));
    } else {
        PostProcess_Hydrosphere w(c->im, c->jm, c->km, out);
        if (section == ATOMDEV_RADIAL)
            w.paraview_vtk_radial(name, index, 0, P.u_0, P.t_0, P.r_0_water, A(H), A(P_DYN), A(P_STAT), A(R_WATER), A(R_SALT_WATER), A(T),
                                  A(U), A(V), A(W), A(C), A(AUX_U), A(AUX_V), A(SALT_FINGER), A(SALT_DIFFUSION), A(BUOYANCY), A(SALT_BALANCE),
                                  B(UPWELLING), B(DOWNWELLING), B(SALT_FINGER_2D), B(SALT_DIFFUSION_2D), B(BUOYANCY_FORCE_2D),
                                  B(BOTTOM_WATER), B(EVAPORATION_DALTON), B(PRECIPITATION), B(TOPOGRAPHY));
        else if (section == ATOMDEV_ZONAL)
            w.paraview_vtk_zonal(name, index, 0, P.u_0, P.r_0_water, A(H), A(P_DYN), A(P_STAT), A(R_WATER), A(R_SALT_WATER), A(T), A(U), A(V),
                                 A(W), A(C), A(SALT_FINGER), A(SALT_DIFFUSION), A(BUOYANCY), A(SALT_BALANCE));
        else
            w.paraview_vtk_longal(name, index, 0, P.u_0, P.r_0_water, A(H), A(P_DYN), A(P_STAT), A(R_WATER), A(R_SALT_WATER), A(T), A(U), A(V),
                                  A(W), A(C), A(AUX_U), A(AUX_V), A(SALT_FINGER), A(SALT_DIFFUSION), A(BUOYANCY), A(SALT_BALANCE));
    }
    copy_out(c);
    return 0;
}

}  // extern "C"

/* Land masks exactly as the reference's readers build them:
 * BC_Bathymetry_Atmosphere::BC_MountainSurface (atmosphere/BC_Bath_Atm.cpp:39-90) and
 * BC_Bathymetry_Hydrosphere::BC_SeaGround (hydrosphere/BC_Bath_Hyd.cpp:31-114).
 * Results land in the bound H and TOPOGRAPHY buffers. */
extern "C" void ref_read_mask(void* v, const char* xyz_path) {
    RefCtx* c = (RefCtx*)v; CoutSilencer s;
    std::string path(xyz_path);
    if (c->model == ATOMDEV_ATM) {
        BC_Bathymetry_Atmosphere LandArea(1, c->im, c->jm, c->km, 3.0, 0., 0.);
        LandArea.BC_MountainSurface(path, c->p.L_atm, B(TOPOGRAPHY), A(H));
    } else {
        BC_Bathymetry_Hydrosphere depth(c->im, c->jm, c->km);
        depth.BC_SeaGround(path, c->p.L_hyd, A(H), B(TOPOGRAPHY));
    }
    copy_out(c);
    if (c->g[ATOMDEV_TOPOGRAPHY])
        for (int j = 0; j < c->jm; j++)
            std::memcpy(c->g[ATOMDEV_TOPOGRAPHY] + (size_t)j * c->km, c->b[ATOMDEV_TOPOGRAPHY]->y[j], sizeof(double) * c->km);
}
