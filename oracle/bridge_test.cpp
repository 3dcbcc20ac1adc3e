/*
 * bridge_test.cpp — the reference's call sites, once with the reference's own classes and once with
 * the same-named classes of include/atomdev_mirror.hpp, on the same seeded fields.
 *
 * TEST INFRASTRUCTURE ONLY.  Our own code; built here (oracle/Makefile, target bridge) against the
 * reference's sources where they lie under /root/reference and against atom_amd/libatomdev.so; the
 * binary oracle/_ref/bridge_test travels to the GPU box with the tree and is run by
 * tests/test_bridge.py.  One function template holds the call text; it is instantiated with the two
 * sets of class names, so what is compared is exactly "same source, other classes".
 *
 * Bars: hydrosphere and every atmosphere operator except the 3-D RK sweep bit for bit; the
 * atmosphere sweep and what follows it 1e-10 relative per field (host and device pow differ in the
 * last place).
 */
#include <cmath>
#include <cstdio>
#include <cstring>
#include <iostream>
#include <random>
#include <sstream>
#include <string>
#include <vector>

#include "Array.h"
#include "Array_1D.h"
#include "Array_2D.h"
#include "RHS_Atm.h"
#include "RungeKutta_Atm.h"
#include "Pressure_Atm.h"
#include "BC_Atm.h"
#include "Results_Atm.h"
#include "BC_Bath_Atm.h"
#undef _PRESSURE_
#include "RHS_Hyd.h"
#include "RungeKutta_Hyd.h"
#include "Pressure_Hyd.h"
#include "BC_Hyd.h"
#include "BC_Bath_Hyd.h"
#include "BC_Thermohalin.h"
#include "Results_Hyd.h"
#include "PostProcess_Atm.h"
#include "PostProcess_Hyd.h"
#include <fstream>
#include <iterator>
#include <new>

#include "../include/atomdev_mirror.hpp"

namespace {

struct RefClasses {
    typedef ::RHS_Atmosphere RHS_A; typedef ::RungeKutta_Atmosphere RK_A; typedef ::Pressure_Atm P_A; typedef ::BC_Atmosphere BC_A;
    typedef ::Results_MSL_Atm MSL_A; typedef ::BC_Bathymetry_Atmosphere BATH_A;
    typedef ::RHS_Hydrosphere RHS_H; typedef ::RungeKutta_Hydrosphere RK_H; typedef ::Pressure_Hyd P_H; typedef ::BC_Hydrosphere BC_H;
    typedef ::BC_Thermohalin TH_H; typedef ::Results_Hyd RES_H;
    typedef ::PostProcess_Atmosphere PP_A; typedef ::PostProcess_Hydrosphere PP_H;
    typedef ::BC_Bathymetry_Hydrosphere BATH_H;
};
struct DevClasses {
    typedef atomdev_mirror::RHS_Atmosphere RHS_A; typedef atomdev_mirror::RungeKutta_Atmosphere RK_A;
    typedef atomdev_mirror::Pressure_Atm P_A; typedef atomdev_mirror::BC_Atmosphere BC_A;
    typedef atomdev_mirror::Results_MSL_Atm MSL_A; typedef atomdev_mirror::BC_Bathymetry_Atmosphere BATH_A;
    typedef atomdev_mirror::RHS_Hydrosphere RHS_H; typedef atomdev_mirror::RungeKutta_Hydrosphere RK_H;
    typedef atomdev_mirror::Pressure_Hyd P_H; typedef atomdev_mirror::BC_Hydrosphere BC_H;
    typedef atomdev_mirror::BC_Thermohalin TH_H; typedef atomdev_mirror::Results_Hyd RES_H;
    typedef atomdev_mirror::PostProcess_Atmosphere PP_A; typedef atomdev_mirror::PostProcess_Hydrosphere PP_H;
    typedef atomdev_mirror::BC_Bathymetry_Hydrosphere BATH_H;
};

struct Scalars {
    double dt = 0.00001, dr, dthe, dphi, re, pr, g = 9.8066, Buoyancy = 1., u_0, t_0 = 273.15, c_0;
    double sc_WaterVapour = .61, sc_CO2 = .96, gam = .65, r_air = 1.2041, R_Air = 287.1, R_WaterVapour = 461.6,
           L_atm = 16000., cp_l = 1005., t_tropopause = .798;
    double sc = 1.7329, L_hyd = 1000., r_0_water = 1000., ta = 1.01464, pa = 0., ca = 1.;
};

struct Fields {
    int im, jm, km;
    std::vector<std::string> names;
    std::vector<Array*> arr;
    Array_2D topo, evap, precip;
    Array_1D rad, the, phi;
    Array& operator[](const char* n) {
        for (size_t i = 0; i < names.size(); i++) if (names[i] == n) return *arr[i];
        names.push_back(n); arr.push_back(new Array(im, jm, km, 0.));
        return *arr.back();
    }
    Fields(int im_, int jm_, int km_, const Scalars& s) : im(im_), jm(jm_), km(km_), topo(jm_, km_, 0.), evap(jm_, km_, 1.), precip(jm_, km_, .5) {
        const int nrad = im > 41 ? im : 41;
        rad.initArray_1D(nrad, 0.); the.initArray_1D(jm, 0.); phi.initArray_1D(km, 0.);
        rad.Coordinates(nrad, 1., s.dr); the.Coordinates(jm, 0., s.dthe); phi.Coordinates(km, 0., s.dphi);
    }
    std::vector<std::string> names2;
    std::vector<Array_2D*> arr2;
    Array_2D& g(const char* n) {  // surface fields by name, created on first use
        for (size_t i = 0; i < names2.size(); i++) if (names2[i] == n) return *arr2[i];
        names2.push_back(n); arr2.push_back(new Array_2D(jm, km, 0.));
        return *arr2.back();
    }
    ~Fields() { for (Array* a : arr) delete a; for (Array_2D* a : arr2) delete a; }
};

void fill(Array& a, int im, int jm, int km, std::mt19937_64& rng, double lo, double hi) {
    std::uniform_real_distribution<double> d(lo, hi);
    for (int i = 0; i < im; i++) for (int j = 0; j < jm; j++) for (int k = 0; k < km; k++) a.x[i][j][k] = d(rng);
}

// columns of solid cells rising from level 0, a third of the columns, up to half the levels
void make_mask(Fields& F, std::mt19937_64& rng, double L) {
    std::uniform_real_distribution<double> u01(0., 1.);
    Array& h = F["h"];
    for (int j = 0; j < F.jm; j++)
        for (int k = 0; k < F.km; k++) {
            int top = u01(rng) < .33 ? 1 + (int)(u01(rng) * (F.im / 2)) : 0;
            F.topo.y[j][k] = top * (L / (double)(F.im - 1));
            for (int i = 0; i < F.im; i++) h.x[i][j][k] = i < top ? 1. : 0.;
        }
}

void init_atm(Fields& F, unsigned seed, const Scalars& s) {
    std::mt19937_64 rng(seed);
    const int im = F.im, jm = F.jm, km = F.km;
    make_mask(F, rng, s.L_atm);
    const char* st[9] = { "t", "u", "v", "w", "p_dyn", "c", "cloud", "ice", "co2" };
    const double lo[9] = { .8, -.1, -.1, -.4, -.1, 0., 0., 0., 1. }, hi[9] = { 1.1, .1, .1, .4, .1, .03, .01, .005, 1.2 };
    for (int n = 0; n < 9; n++) {
        fill(F[st[n]], im, jm, km, rng, lo[n], hi[n]);
        fill(F[(std::string(st[n]) + "n").c_str()], im, jm, km, rng, lo[n], hi[n]);
    }
    for (const char* n : { "S_v", "S_c", "S_i", "S_r", "S_s" }) fill(F[n], im, jm, km, rng, -1e-6, 1e-6);
    for (const char* n : { "aux_u", "aux_v", "aux_w" }) fill(F[n], im, jm, km, rng, -1., 1.);
    for (const char* n : { "BuoyancyForce", "spare" }) (void)F[n];
    for (int n = 0; n < 8; n++) (void)F[("rhs" + std::to_string(n)).c_str()];
}

void init_hyd(Fields& F, unsigned seed, const Scalars& s) {
    std::mt19937_64 rng(seed);
    const int im = F.im, jm = F.jm, km = F.km;
    make_mask(F, rng, s.L_hyd);
    const char* st[6] = { "t", "u", "v", "w", "p_dyn", "c" };
    const double lo[6] = { .9, -.1, -.1, -.1, -.1, .95 }, hi[6] = { 1.1, .1, .1, .1, .1, 1.1 };
    for (int n = 0; n < 6; n++) {
        fill(F[st[n]], im, jm, km, rng, lo[n], hi[n]);
        fill(F[(std::string(st[n]) + "n").c_str()], im, jm, km, rng, lo[n], hi[n]);
    }
    fill(F["r_water"], im, jm, km, rng, 995., 1005.);
    fill(F["r_salt_water"], im, jm, km, rng, 1020., 1035.);
    for (const char* n : { "aux_u", "aux_v", "aux_w" }) fill(F[n], im, jm, km, rng, -1., 1.);
    for (const char* n : { "Salt_Finger", "Salt_Diffusion", "Buoyancy_Force", "Salt_Balance", "spare" }) (void)F[n];
    for (int n = 0; n < 5; n++) (void)F[("rhs" + std::to_string(n)).c_str()];
}

// ---- the reference's call text (cAtmosphereModel.cpp:777-874), for either set of classes ----
template <class NS>
void atmosphere_calls(Fields& F, const Scalars& s, bool with_rk3d) {
    const int im = F.im, jm = F.jm, km = F.km;
    typename NS::BC_A boundary(im, jm, km, s.t_tropopause);
    typename NS::RHS_A prepare(im, jm, km, s.dt, s.dr, s.dthe, s.dphi, s.re, s.sc_WaterVapour, s.sc_CO2, s.g, s.pr, 1., s.Buoyancy,
                               1., s.gam, 5.670280e-8, 0.0262);
    typename NS::RHS_A prepare_2D(jm, km, s.dthe, s.dphi, s.re);
    typename NS::RK_A result(im, jm, km, s.dt, s.dr, s.dphi, s.dthe);
    typename NS::P_A startPressure(im, jm, km, s.dr, s.dthe, s.dphi);
    int n = 1;
    boundary.BC_radius(F["t"], F["u"], F["v"], F["w"], F["p_dyn"], F["c"], F["cloud"], F["ice"], F["co2"]);
    boundary.BC_theta(F["t"], F["u"], F["v"], F["w"], F["p_dyn"], F["c"], F["cloud"], F["ice"], F["co2"]);
    boundary.BC_phi(F["t"], F["u"], F["v"], F["w"], F["p_dyn"], F["c"], F["cloud"], F["ice"], F["co2"]);
    if (with_rk3d)
        result.solveRungeKutta_3D_Atmosphere(prepare, n, 2.52e6, 2.83e6, 0.623, 6.1078, s.u_0, s.t_0, s.c_0, 280.0, 1013.25,
            s.r_air, 0.0094, 0.0019767, s.L_atm, s.cp_l, s.R_Air, s.R_WaterVapour, 188.91, F.rad, F.the, F.phi,
            F["rhs0"], F["rhs1"], F["rhs2"], F["rhs3"], F["rhs4"], F["rhs5"], F["rhs6"], F["rhs7"],
            F["h"], F["t"], F["u"], F["v"], F["w"], F["p_dyn"], F["spare"], F["c"], F["cloud"], F["ice"], F["co2"],
            F["tn"], F["un"], F["vn"], F["wn"], F["p_dynn"], F["cn"], F["cloudn"], F["icen"], F["co2n"],
            F["aux_u"], F["aux_v"], F["aux_w"], F["spare"], F["BuoyancyForce"], F["spare"], F["spare"], F["spare"],
            F["S_v"], F["S_c"], F["S_i"], F["S_r"], F["S_s"], F["spare"], F.topo, F.evap, F.precip);
    startPressure.computePressure_3D(s.u_0, s.r_air, F.rad, F.the, F["p_dyn"], F["p_dynn"], F["h"], F["aux_u"], F["aux_v"], F["aux_w"]);
    result.solveRungeKutta_2D_Atmosphere(prepare_2D, n, s.r_air, s.u_0, 1013.25, s.L_atm, F.rad, F.the, F.phi, F["rhs2"], F["rhs3"],
        F["h"], F["v"], F["w"], F["p_dyn"], F["vn"], F["wn"], F["p_dynn"], F["aux_v"], F["aux_w"]);
    startPressure.computePressure_2D(s.u_0, s.r_air, F.rad, F.the, F["p_dyn"], F["p_dynn"], F["h"], F["aux_v"], F["aux_w"]);
}

// ---- cAtmosphereModel.cpp:854-859 ----
template <class NS>
void msl_calls(Fields& F, const Scalars& s) {
    typename NS::MSL_A calculate_MSL(F.im, F.jm, F.km, 0, s.g, 0.623, 6.1078, s.u_0, 1013.25, s.t_0, s.c_0, 280.0, 5.670280e-8,
                                     0.15, 2.52e6, 2.83e6, s.cp_l, s.L_atm, s.dt, s.dr, s.dthe, s.dphi, s.r_air, s.R_Air, 0.0094,
                                     s.R_WaterVapour, 3.0, 0., 3., s.gam, -4.0, 0., 15.0);
    double t_cretaceous = 0.;
    calculate_MSL.run_MSL_data(1, 2, 1, t_cretaceous, F.rad, F.the, F.phi, F["h"], F["c"], F["cn"], F["co2"], F["co2n"], F["t"],
        F["tn"], F["p_dyn"], F["p_stat"], F["BuoyancyForce"], F["u"], F["v"], F["w"], F["Q_Latent"], F["Q_Sensible"],
        F["radiation_3D"], F["cloud"], F["cloudn"], F["ice"], F["icen"], F["P_rain"], F["P_snow"], F["aux_u"], F["aux_v"],
        F["aux_w"], F.g("temperature_NASA"), F.g("precipitation_NASA"), F.g("precipitable_water"), F.g("Q_radiation"),
        F.g("Q_Evaporation"), F.g("Q_latent"), F.g("Q_sensible"), F.g("Q_bottom"), F.g("Evaporation_Penman"), F.evap,
        F.g("Vegetation"), F.g("albedo"), F.g("co2_total"), F.precip, F["S_v"], F["S_c"], F["S_i"], F["S_r"], F["S_s"],
        F["spare"]);
}

void init_msl(Fields& F, unsigned seed, const Scalars& s) {
    init_atm(F, seed, s);
    std::mt19937_64 rng(seed + 1000);
    const int im = F.im, jm = F.jm, km = F.km;
    fill(F["p_stat"], im, jm, km, rng, 200., 1013.); fill(F["Q_Latent"], im, jm, km, rng, -50., 150.);
    fill(F["Q_Sensible"], im, jm, km, rng, -50., 50.); fill(F["radiation_3D"], im, jm, km, rng, 100., 400.);
    fill(F["P_rain"], im, jm, km, rng, -1e-5, 4e-4); fill(F["P_snow"], im, jm, km, rng, -1e-5, 1e-4);
    fill(F["BuoyancyForce"], im, jm, km, rng, -1., 1.); fill(F["cloud"], im, jm, km, rng, -1e-3, .01);
    std::uniform_real_distribution<double> d(0., 5.);
    for (const char* n : { "Q_radiation", "Q_Evaporation", "Q_latent", "Q_sensible", "Q_bottom" })
        for (int j = 0; j < jm; j++) for (int k = 0; k < km; k++) F.g(n).y[j][k] = d(rng);
    for (int j = 0; j < jm; j++) for (int k = 0; k < km; k++) F.g("temperature_NASA").y[j][k] = 1.;
}

// ---- cAtmosphereModel.cpp:793-796, 850 ----
template <class NS>
void land_calls(Fields& F, const Scalars& s) {
    typename NS::BATH_A LandArea(1, F.im, F.jm, F.km, 3.0, 0., 0.);
    LandArea.BC_SolidGround(1, 0, s.g, 6.1078, 0.623, s.r_air, s.R_Air, s.t_0, s.c_0, 0., 0., 1.0842, 0.945, s.t_tropopause,
        0.64, 0.001, 280.0, 330.0, 320.0, 320.0, 0., s.gam, 5.670280e-8, F["h"], F["u"], F["v"], F["w"], F["t"], F["p_dyn"],
        F["c"], F["cloud"], F["ice"], F["co2"], F["radiation_3D"], F.g("Vegetation"));
    LandArea.vegetationDistribution(.8, F.precip, F.g("Vegetation"), F["t"], F["h"]);
}

// ---- the BC_Thermo members of the iteration loop (cAtmosphereModel.cpp:788-865) ----
// This program links the reference's operator classes without its model classes (BC_Thermo needs a cAtmosphereModel
// object), so the mirror of BC_Thermo is checked for its plumbing: the same calls through the C-ABI with all fields resident must give the same bits.
struct LoopModel {  // the public members BC_Thermo's constructor reads (BC_Thermo.cpp:38-94)
    double L_atm = 16000., dt = 0.00001, dr, dthe, dphi, g = 9.8066, ep = .623, hp = 6.1078, u_0 = 8., p_0 = 1013.25,
           t_0 = 273.15, sigma = 5.670280e-8, albedo_equator = .15, albedo_pole = .7, gam = .65, lv = 2.52e6, ls = 2.83e6,
           cp_l = 1005., r_air = 1.2041, R_Air = 287.1, r_water_vapour = .0094, R_WaterVapour = 461.6, co2_factor = .98,
           rad_equator = 230., rad_pole = 40., epsilon_pole = .59, epsilon_tropopause = .001, epsilon_equator = .594,
           co2_equator = 330., co2_tropopause = 320., t_tropopause = .798, CO2 = 1.;
    std::vector<int> trop;
    double now = 0.;
    int* get_tropopause() { return trop.data(); }
    double* get_current_time() { return &now; }
    double get_mean_temperature_from_curve(int) const { return 0.; }
};

void init_thermo(Fields& F, unsigned seed, const Scalars& s) {
    init_msl(F, seed, s);
    std::mt19937_64 rng(seed + 2000);
    const int im = F.im, jm = F.jm, km = F.km;
    fill(F["cloud"], im, jm, km, rng, -1e-5, 1e-3); fill(F["ice"], im, jm, km, rng, -1e-5, 1e-3);
    fill(F["c"], im, jm, km, rng, -1e-5, 4e-3); fill(F["cn"], im, jm, km, rng, 0., 4e-3);
    fill(F["S_c_c"], im, jm, km, rng, -1e-7, 1e-7); fill(F["epsilon_3D"], im, jm, km, rng, .3, 1.2);
    for (int i = 0; i < im; i++) for (int j = 0; j < jm; j++) for (int k = 0; k < km; k++)
        F["t"].x[i][j][k] = 1.07 - .25 * i / (im - 1.) + .02 * std::sin(.37 * j + .11 * k);
    for (int j = 0; j < jm; j++) for (int k = 0; k < km; k++) F["h"].x[im - 1][j][k] = 0.;
    for (const char* n : { "albedo", "epsilon", "radiation_surface" }) (void)F.g(n);
}

void thermo_mirror_calls(Fields& F, LoopModel& m) {
    atomdev_mirror::BC_Thermo circulation(&m, F.im, F.jm, F.km, F["h"]);
    for (int velocity_iter = 1; velocity_iter <= 2; velocity_iter++) {  // two rounds: the carried members travel
        circulation.Ice_Water_Saturation_Adjustment(F["h"], F["c"], F["cn"], F["cloud"], F["cloudn"], F["ice"], F["icen"], F["t"],
            F["p_stat"], F["S_c_c"]);
        circulation.Value_Limitation_Atm(F["h"], F["u"], F["v"], F["w"], F["p_dyn"], F["t"], F["c"], F["cloud"], F["ice"], F["co2"]);
        circulation.BC_Radiation_multi_layer(F.g("albedo"), F.g("epsilon"), F.g("radiation_surface"), F["p_stat"], F["t"], F["c"],
            F["h"], F["epsilon_3D"], F["radiation_3D"], F["cloud"], F["ice"], F["co2"]);
        circulation.Latent_Heat(F.rad, F.the, F.phi, F["h"], F["t"], F["tn"], F["u"], F["v"], F["w"], F["p_dyn"], F["p_stat"], F["c"],
            F["ice"], F["Q_Latent"], F["Q_Sensible"], F["radiation_3D"], F.g("Q_radiation"), F.g("Q_latent"), F.g("Q_sensible"),
            F.g("Q_bottom"));
        circulation.Two_Category_Ice_Scheme(F["h"], F["c"], F["t"], F["p_stat"], F["cloud"], F["ice"], F["P_rain"], F["P_snow"],
            F["S_v"], F["S_c"], F["S_i"], F["S_r"], F["S_s"], F["S_c_c"]);
    }
}

#define CHECK(call) do { if (call) { std::printf("  %s: %s\n", #call, atomdev_last_error()); std::abort(); } } while (0)
void thermo_resident_calls(Fields& F, LoopModel& m, const Scalars& s) {
    static const struct { const char* name; int id; } map3[] = {
        { "h", ATOMDEV_H }, { "t", ATOMDEV_T }, { "u", ATOMDEV_U }, { "v", ATOMDEV_V }, { "w", ATOMDEV_W }, { "p_dyn", ATOMDEV_P_DYN },
        { "c", ATOMDEV_C }, { "cloud", ATOMDEV_CLOUD }, { "ice", ATOMDEV_ICE }, { "co2", ATOMDEV_CO2 }, { "cn", ATOMDEV_CN },
        { "cloudn", ATOMDEV_CLOUDN }, { "icen", ATOMDEV_ICEN }, { "p_stat", ATOMDEV_P_STAT }, { "S_c_c", ATOMDEV_S_C_C },
        { "epsilon_3D", ATOMDEV_EPSILON_3D }, { "radiation_3D", ATOMDEV_RADIATION_3D }, { "Q_Latent", ATOMDEV_Q_LATENT },
        { "Q_Sensible", ATOMDEV_Q_SENSIBLE }, { "P_rain", ATOMDEV_P_RAIN }, { "P_snow", ATOMDEV_P_SNOW }, { "S_v", ATOMDEV_S_V },
        { "S_c", ATOMDEV_S_C }, { "S_i", ATOMDEV_S_I }, { "S_r", ATOMDEV_S_R }, { "S_s", ATOMDEV_S_S } };
    static const struct { const char* name; int id; } map2[] = {
        { "albedo", ATOMDEV_ALBEDO }, { "epsilon", ATOMDEV_EPSILON }, { "radiation_surface", ATOMDEV_RADIATION_SURFACE } };
    const int im = F.im, jm = F.jm, km = F.km;
    atomdev_params p; atomdev_default_params(ATOMDEV_ATM, &p);
    p.dr = m.dr; p.dthe = m.dthe; p.dphi = m.dphi; p.dt = m.dt; p.u_0 = m.u_0; p.t_tropopause = m.t_tropopause;
    p.r0 = F.rad.z[0]; p.the0 = F.the.z[0]; (void)s;
    atomdev_t* h = nullptr;
    CHECK(atomdev_create(ATOMDEV_ATM, im, jm, km, &p, -1, &h));
    std::vector<double> buf((size_t)im * jm * km);
    for (const auto& e : map3) {
        Array& a = F[e.name];
        for (int i = 0; i < im; i++) for (int j = 0; j < jm; j++) std::memcpy(&buf[((size_t)i * jm + j) * km], a.x[i][j], 8 * km);
        CHECK(atomdev_upload(h, e.id, buf.data()));
    }
    for (const auto& e : map2) {
        Array_2D& a = F.g(e.name);
        for (int j = 0; j < jm; j++) std::memcpy(&buf[(size_t)j * km], a.y[j], 8 * km);
        CHECK(atomdev_upload2d(h, e.id, buf.data()));
    }
    atomdev_radiation r; atomdev_default_radiation(&r);
    double q_v_hyp = 0., ice_state[2] = { 0., 0. };
    for (int velocity_iter = 1; velocity_iter <= 2; velocity_iter++) {
        CHECK(atomdev_atm_ice_water_saturation_adjustment(h, &q_v_hyp));
        CHECK(atomdev_atm_value_limitation(h));
        CHECK(atomdev_atm_radiation_multi_layer(h, &r, m.trop.data()));
        CHECK(atomdev_atm_latent_heat(h));
        CHECK(atomdev_atm_two_category_ice_scheme(h, ice_state));
    }
    for (const auto& e : map3) {
        Array& a = F[e.name];
        CHECK(atomdev_download(h, e.id, buf.data()));
        for (int i = 0; i < im; i++) for (int j = 0; j < jm; j++) std::memcpy(a.x[i][j], &buf[((size_t)i * jm + j) * km], 8 * km);
    }
    for (const auto& e : map2) {
        Array_2D& a = F.g(e.name);
        CHECK(atomdev_download2d(h, e.id, buf.data()));
        for (int j = 0; j < jm; j++) std::memcpy(a.y[j], &buf[(size_t)j * km], 8 * km);
    }
    atomdev_destroy(h);
}

// ---- cHydrosphereModel.cpp:437-610 ----
template <class NS>
void hydrosphere_calls(Fields& F, const Scalars& s) {
    const int im = F.im, jm = F.jm, km = F.km;
    typename NS::BC_H boundary(im, jm, km);
    typename NS::RHS_H prepare(im, jm, km, s.dt, s.dr, s.dthe, s.dphi, s.re, s.sc, s.g, s.pr, s.Buoyancy);
    typename NS::RHS_H prepare_2D(jm, km, s.dthe, s.dphi, s.re);
    typename NS::RK_H result(im, jm, km, s.dt);
    typename NS::P_H startPressure(im, jm, km, s.dr, s.dthe, s.dphi);
    int n = 1;
    boundary.RB_radius(s.ca, s.ta, s.pa, s.dr, F.rad, F["t"], F["u"], F["v"], F["w"], F["p_dyn"], F["c"]);
    boundary.RB_theta(s.ca, s.ta, s.pa, F["t"], F["u"], F["v"], F["w"], F["p_dyn"], F["c"]);
    boundary.RB_phi(F["t"], F["u"], F["v"], F["w"], F["p_dyn"], F["c"]);
    typename NS::TH_H oceanflow(im, jm, km, 0, im - 1, 0, 300, 150, s.dr, s.g, s.r_0_water, 0., 0., 0., s.ta, s.ca, s.pa, s.u_0,
                                1013.25, s.t_0, s.c_0, 4182.0, s.L_hyd, 15.0, 10.0, 1.1355, 1.0146, "");
    oceanflow.BC_Pressure_Density(F["p_stat"], F["r_water"], F["r_salt_water"], F["t"], F["c"], F["h"]);
    oceanflow.Value_Limitation_Hyd(F["h"], F["u"], F["v"], F["w"], F["p_dyn"], F["t"], F["c"]);
    result.solveRungeKutta_3D_Hydrosphere(prepare, n, s.L_hyd, s.g, 4182.0, s.u_0, s.t_0, s.c_0, s.r_0_water, s.ta, s.pa, s.ca,
        F.rad, F.the, F.phi, F.evap, F.precip, F["h"], F["rhs0"], F["rhs1"], F["rhs2"], F["rhs3"], F["rhs4"],
        F["t"], F["u"], F["v"], F["w"], F["p_dyn"], F["c"], F["tn"], F["un"], F["vn"], F["wn"], F["p_dynn"], F["cn"],
        F["aux_u"], F["aux_v"], F["aux_w"], F["Salt_Finger"], F["Salt_Diffusion"], F["Buoyancy_Force"], F["Salt_Balance"],
        F["p_stat"], F["r_water"], F["r_salt_water"], F.topo);
    startPressure.computePressure_3D(s.u_0, s.r_0_water, F.rad, F.the, F["p_dyn"], F["p_dynn"], F["h"], F["aux_u"], F["aux_v"], F["aux_w"]);
    result.solveRungeKutta_2D_Hydrosphere(prepare_2D, n, s.r_0_water, F.rad, F.the, F.phi, F["rhs2"], F["rhs3"],
        F["h"], F["v"], F["w"], F["p_dyn"], F["vn"], F["wn"], F["p_dynn"], F["aux_v"], F["aux_w"]);
    startPressure.computePressure_2D(s.u_0, s.r_0_water, F.rad, F.the, F["p_dyn"], F["p_dynn"], F["h"], F["aux_v"], F["aux_w"]);
}

// ---- cHydrosphereModel.cpp:597-599, :771 and cAtmosphereModel.cpp's Atmosphere_PlotData call ----
// Results_Hyd divides x - x by a member that no constructor sets (Results_Hyd.h:26); both sets of classes are built in
// storage filled with the byte 0x3f, which makes that member a small positive number for the reference's class.
template <class NS>
void results_calls(Fields& F, const Scalars& s, const std::string& dir) {
    alignas(16) static unsigned char storage[sizeof(typename NS::RES_H)];
    std::memset(storage, 0x3f, sizeof(storage));
    typedef typename NS::RES_H Results;
    Results* calculate_MSL = new (storage) Results(F.im, F.jm, F.km);
    calculate_MSL->run_data(0, s.dr, s.dthe, s.L_hyd, s.u_0, s.c_0, F.rad, F.the, F["h"], F["u"], F["v"], F["w"], F["c"],
        F["Salt_Balance"], F["Salt_Finger"], F["Salt_Diffusion"], F["Buoyancy_Force"], F.g("Upwelling"), F.g("Downwelling"),
        F.g("SaltFinger"), F.g("SaltDiffusion"), F.g("BuoyancyForce_2D"), F.g("Salt_total"), F.g("BottomWater"));
    calculate_MSL->~Results();
    typename NS::PP_H ppa(F.im, F.jm, F.km, dir);
    double u_0 = s.u_0;
    ppa.Hydrosphere_PlotData("bridge", -1, u_0, F["h"], F["v"], F["w"], F["t"], F["c"], F.g("BottomWater"), F.g("Upwelling"),
                             F.g("Downwelling"));
}

template <class NS>
void atm_plot_calls(Fields& F, const Scalars& s, std::string dir) {
    typename NS::PP_A ppa(F.im, F.jm, F.km, dir);
    std::string name("bridge");
    ppa.Atmosphere_PlotData(name, 3, s.u_0, s.t_0, F["h"], F["v"], F["w"], F["t"], F["c"], F.precip, F.g("precipitable_water"));
}

// ---- cAtmosphereModel.cpp:612-629 and cHydrosphereModel.cpp:747-765: the ParaView sections of a time slice ----
template <class NS>
void atm_vtk_calls(Fields& F, const Scalars& s, std::string dir, int i_radial, int j_longal, int k_zonal) {
    typename NS::PP_A write_File(F.im, F.jm, F.km, dir);
    std::string bathymetry_name("bridge");
    int Ma = 0, iter_cnt = 1;
    double u_0 = s.u_0, t_0 = s.t_0, p_0 = 1013.25, r_air = s.r_air, c_0 = s.c_0, co2_0 = 280., hp = 6.1078, ep = .623,
           R_Air = s.R_Air, g = s.g, L_atm = s.L_atm;
    write_File.paraview_panorama_vts(bathymetry_name, iter_cnt - 1, u_0, t_0, p_0, r_air, c_0, co2_0, F["h"], F["t"], F["p_dyn"],
        F["p_stat"], F["BuoyancyForce"], F["u"], F["v"], F["w"], F["c"], F["co2"], F["cloud"], F["ice"], F["aux_u"], F["aux_v"],
        F["aux_w"], F["Q_Latent"], F["Q_Sensible"], F["epsilon_3D"], F["P_rain"], F["P_snow"]);
    write_File.paraview_vtk_radial(bathymetry_name, Ma, i_radial, iter_cnt - 1, u_0, t_0, p_0, r_air, c_0, co2_0, F["h"], F["p_dyn"],
        F["p_stat"], F["BuoyancyForce"], F["t"], F["u"], F["v"], F["w"], F["c"], F["co2"], F["cloud"], F["ice"], F["aux_u"], F["aux_v"],
        F["aux_w"], F["radiation_3D"], F["Q_Latent"], F["Q_Sensible"], F["epsilon_3D"], F["P_rain"], F["P_snow"],
        F.g("precipitable_water"), F.g("Q_bottom"), F.g("Q_radiation"), F.g("Q_latent"), F.g("Q_sensible"), F.g("Evaporation_Penman"),
        F.evap, F.g("Q_Evaporation"), F.g("temperature_NASA"), F.g("precipitation_NASA"), F.g("Vegetation"), F.g("albedo"),
        F.g("epsilon"), F.precip, F.topo, F.g("temp_NASA"));
    write_File.paraview_vtk_longal(bathymetry_name, j_longal, iter_cnt - 1, u_0, t_0, p_0, r_air, c_0, co2_0, F["h"], F["p_dyn"],
        F["p_stat"], F["BuoyancyForce"], F["t"], F["u"], F["v"], F["w"], F["c"], F["co2"], F["cloud"], F["ice"], F["aux_u"], F["aux_v"],
        F["aux_w"], F["Q_Latent"], F["Q_Sensible"], F["epsilon_3D"], F["P_rain"], F["P_snow"]);
    write_File.paraview_vtk_zonal(bathymetry_name, k_zonal, iter_cnt - 1, hp, ep, R_Air, g, L_atm, u_0, t_0, p_0, r_air, c_0, co2_0,
        F["h"], F["p_dyn"], F["p_stat"], F["BuoyancyForce"], F["t"], F["u"], F["v"], F["w"], F["c"], F["co2"], F["cloud"], F["ice"],
        F["aux_u"], F["aux_v"], F["aux_w"], F["Q_Latent"], F["Q_Sensible"], F["radiation_3D"], F["epsilon_3D"], F["P_rain"],
        F["P_snow"], F["S_v"], F["S_c"], F["S_i"], F["S_r"], F["S_s"], F["S_c_c"]);
}

template <class NS>
void hyd_vtk_calls(Fields& F, const Scalars& s, const std::string& dir, int i_radial, int j_longal, int k_zonal) {
    typename NS::PP_H write_File(F.im, F.jm, F.km, dir);
    const std::string bathymetry_name("bridge");
    int iter_cnt = 1;
    double u_0 = s.u_0, t_0 = s.t_0, r_0_water = s.r_0_water;
    write_File.paraview_vtk_longal(bathymetry_name, j_longal, iter_cnt - 1, u_0, r_0_water, F["h"], F["p_dyn"], F["p_stat"], F["r_water"],
        F["r_salt_water"], F["t"], F["u"], F["v"], F["w"], F["c"], F["aux_u"], F["aux_v"], F["Salt_Finger"], F["Salt_Diffusion"],
        F["Buoyancy_Force"], F["Salt_Balance"]);
    write_File.paraview_panorama_vts(bathymetry_name, iter_cnt - 1, u_0, r_0_water, F["h"], F["t"], F["p_dyn"], F["p_stat"], F["r_water"],
        F["r_salt_water"], F["u"], F["v"], F["w"], F["c"], F["aux_u"], F["aux_v"], F["aux_w"], F["Salt_Finger"], F["Salt_Diffusion"],
        F["Buoyancy_Force"], F["Salt_Balance"]);
    write_File.paraview_vtk_zonal(bathymetry_name, k_zonal, iter_cnt - 1, u_0, r_0_water, F["h"], F["p_dyn"], F["p_stat"], F["r_water"],
        F["r_salt_water"], F["t"], F["u"], F["v"], F["w"], F["c"], F["Salt_Finger"], F["Salt_Diffusion"], F["Buoyancy_Force"],
        F["Salt_Balance"]);
    write_File.paraview_vtk_radial(bathymetry_name, i_radial, iter_cnt - 1, u_0, t_0, r_0_water, F["h"], F["p_dyn"], F["p_stat"],
        F["r_water"], F["r_salt_water"], F["t"], F["u"], F["v"], F["w"], F["c"], F["aux_u"], F["aux_v"], F["Salt_Finger"],
        F["Salt_Diffusion"], F["Buoyancy_Force"], F["Salt_Balance"], F.g("Upwelling"), F.g("Downwelling"), F.g("SaltFinger"),
        F.g("SaltDiffusion"), F.g("BuoyancyForce_2D"), F.g("BottomWater"), F.evap, F.precip, F.topo);
}

// ---- cHydrosphereModel.cpp:244-292: the hydrosphere's start of a time slice ----
template <class NS>
void hyd_start_calls(Fields& F, const Scalars& s, const std::string& bathymetry_filepath, const std::string& Name_SurfaceSalinity_File) {
    const int im = F.im, jm = F.jm, km = F.km;
    int i_max = im - 1, i_beg = 0, Ma = 0, Ma_max = 300, Ma_max_half = 150;
    double ua = 0., va = 0., wa = 0., p_0 = 1013.25, cp_w = 4182., t_average = 15., t_cretaceous_max = 10., t_equator = 1.1355,
           t_pole = 1.0146;
    std::string input_path;
    typename NS::BATH_H depth(im, jm, km);
    depth.BC_SeaGround(bathymetry_filepath, s.L_hyd, F["h"], F.topo);
    typename NS::TH_H oceanflow(im, jm, km, i_beg, i_max, Ma, Ma_max, Ma_max_half, s.dr, s.g, s.r_0_water, ua, va, wa, s.ta, s.ca, s.pa,
                                s.u_0, p_0, s.t_0, s.c_0, cp_w, s.L_hyd, t_average, t_cretaceous_max, t_equator, t_pole, input_path);
    oceanflow.BC_Surface_Salinity_NASA(Name_SurfaceSalinity_File, F["c"]);
    oceanflow.IC_v_w_EkmanSpiral(F.rad, F.the, F["h"], F["v"], F["w"]);
    if (Ma <= 41) oceanflow.IC_CircumPolar_Current(F["h"], F["u"], F["v"], F["w"], F["c"]);
    oceanflow.BC_Temperature_Salinity(F["h"], F["t"], F["c"], F["p_dyn"]);
    oceanflow.BC_Pressure_Density(F["p_stat"], F["r_water"], F["r_salt_water"], F["t"], F["c"], F["h"]);
}

std::string slurp(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    return std::string(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
}

int same_file(const std::string& a, const std::string& b, const char* what) {
    const std::string x = slurp(a), y = slurp(b);
    const bool ok = !x.empty() && x == y;
    std::printf("%s: %zu bytes, %s\n", what, x.size(), ok ? "ok (byte-identical)" : "FAIL");
    return ok ? 0 : 1;
}

int compare(Fields& A, Fields& B, double tol, const char* what) {
    int bad = 0;
    for (size_t n = 0; n < A.names.size(); n++) {
        if (A.names[n].compare(0, 3, "rhs") == 0 || A.names[n] == "spare") continue;  // scratch of the reference's RHS
        Array& a = *A.arr[n]; Array& b = B[A.names[n].c_str()];
        double maxd = 0., maxa = 0.; bool bits = true;
        for (int i = 0; i < A.im; i++) for (int j = 0; j < A.jm; j++) for (int k = 0; k < A.km; k++) {
            const double x = a.x[i][j][k], y = b.x[i][j][k];
            if (std::memcmp(&x, &y, 8) != 0 && !(x == 0. && y == 0.)) bits = false;
            maxd = std::fmax(maxd, std::fabs(x - y)); maxa = std::fmax(maxa, std::fabs(x));
        }
        const double rel = maxa > 0. ? maxd / maxa : maxd;
        const bool ok = tol == 0. ? bits : rel <= tol;
        if (!ok) { bad++; std::printf("  %s %-16s MISMATCH rel %.3e bits %d\n", what, A.names[n].c_str(), rel, (int)bits); }
    }
    B.g("evap") ; A.g("evap");  // compare the constructor-made surface fields as well
    for (int j = 0; j < A.jm; j++) for (int k = 0; k < A.km; k++) { A.g("evap").y[j][k] = A.evap.y[j][k]; B.g("evap").y[j][k] = B.evap.y[j][k]; }
    A.g("topo"); B.g("topo");
    for (int j = 0; j < A.jm; j++) for (int k = 0; k < A.km; k++) { A.g("topo").y[j][k] = A.topo.y[j][k]; B.g("topo").y[j][k] = B.topo.y[j][k]; }
    A.g("precip"); B.g("precip");
    for (int j = 0; j < A.jm; j++) for (int k = 0; k < A.km; k++) { A.g("precip").y[j][k] = A.precip.y[j][k]; B.g("precip").y[j][k] = B.precip.y[j][k]; }
    for (size_t n = 0; n < A.names2.size(); n++) {
        Array_2D& a = *A.arr2[n]; Array_2D& b = B.g(A.names2[n].c_str());
        double maxd = 0., maxa = 0.; bool bits = true;
        for (int j = 0; j < A.jm; j++) for (int k = 0; k < A.km; k++) {
            const double x = a.y[j][k], y = b.y[j][k];
            if (std::memcmp(&x, &y, 8) != 0 && !(x == 0. && y == 0.)) bits = false;
            maxd = std::fmax(maxd, std::fabs(x - y)); maxa = std::fmax(maxa, std::fabs(x));
        }
        const double rel = maxa > 0. ? maxd / maxa : maxd;
        const bool ok = tol == 0. ? bits : rel <= tol;
        if (!ok) { bad++; std::printf("  %s %-16s MISMATCH rel %.3e bits %d\n", what, A.names2[n].c_str(), rel, (int)bits); }
    }
    std::printf("%s: %zu + %zu fields, %s (%s)\n", what, A.names.size(), A.names2.size(), bad ? "FAIL" : "ok", tol == 0. ? "bit-exact" : "<= 1e-10 relative");
    return bad;
}

struct Quiet {  // the reference prints progress on std::cout
    std::streambuf* old; std::ostringstream sink;
    Quiet() : old(std::cout.rdbuf(sink.rdbuf())) {}
    ~Quiet() { std::cout.rdbuf(old); }
};

}  // namespace

int main(int argc, char** argv) {
    int bad = 0;
    const std::string tmp = argc > 1 ? argv[1] : "/tmp";
    {   // atmosphere without the pow-dependent sweep: bit-exact
        Scalars s; s.re = 1000.; s.pr = .7179; s.u_0 = 8.; s.c_0 = .035;
        const int im = 9, jm = 21, km = 70;
        s.dr = 1. / (im - 1); s.dthe = M_PI / (jm - 1); s.dphi = s.dthe;
        Fields R(im, jm, km, s), D(im, jm, km, s);
        init_atm(R, 7, s); init_atm(D, 7, s);
        { Quiet q; atmosphere_calls<RefClasses>(R, s, false); }
        atmosphere_calls<DevClasses>(D, s, false);
        bad += compare(R, D, 0., "atmosphere, fills + pressure + surface loop");
        Fields R2(im, jm, km, s), D2(im, jm, km, s);
        init_atm(R2, 8, s); init_atm(D2, 8, s);
        { Quiet q; atmosphere_calls<RefClasses>(R2, s, true); }
        atmosphere_calls<DevClasses>(D2, s, true);
        bad += compare(R2, D2, 1e-10, "atmosphere, with the 3-D RK sweep");
    }
    {   // surface fluxes: run_MSL_data prints at fixed 1-degree locations, so the surface stays 181 x 361
        Scalars s; s.re = 1000.; s.pr = .7179; s.u_0 = 8.; s.c_0 = .035;
        const int im = 5, jm = 181, km = 361;
        s.dr = 1. / (im - 1); s.dthe = M_PI / (jm - 1); s.dphi = s.dthe;
        Fields R(im, jm, km, s), D(im, jm, km, s);
        init_msl(R, 10, s); init_msl(D, 10, s);
        { Quiet q; msl_calls<RefClasses>(R, s); }
        msl_calls<DevClasses>(D, s);
        bad += compare(R, D, 1e-10, "atmosphere, run_MSL_data");
    }
    {   // land reset and vegetation: the reference's class against the mirror, bit-exact
        Scalars s; s.re = 1000.; s.pr = .7179; s.u_0 = 8.; s.c_0 = .035;
        const int im = 9, jm = 21, km = 70;
        s.dr = 1. / (im - 1); s.dthe = M_PI / (jm - 1); s.dphi = s.dthe;
        Fields R(im, jm, km, s), D(im, jm, km, s);
        init_msl(R, 11, s); init_msl(D, 11, s);
        { Quiet q; land_calls<RefClasses>(R, s); }
        land_calls<DevClasses>(D, s);
        bad += compare(R, D, 0., "atmosphere, BC_SolidGround + vegetationDistribution");
    }
    {   // BC_Thermo's loop members: mirror class (fields travel with every call) against resident fields, bit-exact
        Scalars s; s.re = 1000.; s.pr = .7179; s.u_0 = 8.; s.c_0 = .035;
        const int im = 12, jm = 21, km = 34;
        s.dr = 1. / (im - 1); s.dthe = M_PI / (jm - 1); s.dphi = s.dthe;
        LoopModel m; m.dr = s.dr; m.dthe = s.dthe; m.dphi = s.dphi; m.trop.assign(jm, im - 3);
        Fields A(im, jm, km, s), B(im, jm, km, s);
        init_thermo(A, 12, s); init_thermo(B, 12, s);
        thermo_mirror_calls(A, m);
        thermo_resident_calls(B, m, s);
        bad += compare(A, B, 0., "atmosphere, BC_Thermo loop members: mirror against resident C-ABI");
    }
    {   // hydrosphere (the surface RHS hard-codes im = 41, RHS_Hyd.cpp:466)
        Scalars s; s.re = 10.; s.pr = 6.957; s.u_0 = .25; s.c_0 = 34.6;
        const int im = 41, jm = 43, km = 24;
        s.dr = 1. / (im - 1); s.dthe = M_PI / (jm - 1); s.dphi = s.dthe;
        Fields R(im, jm, km, s), D(im, jm, km, s);
        init_hyd(R, 9, s); init_hyd(D, 9, s);
        { Quiet q; hydrosphere_calls<RefClasses>(R, s); }
        hydrosphere_calls<DevClasses>(D, s);
        bad += compare(R, D, 0., "hydrosphere");
    }
    {   // Results_Hyd::run_data and the hydrosphere's PlotData table; the class prints at (j, k) = (90, 180)
        Scalars s; s.re = 10.; s.pr = 6.957; s.u_0 = .25; s.c_0 = 34.6;
        const int im = 9, jm = 91, km = 181;
        s.dr = 1. / (im - 1); s.dthe = M_PI / (jm - 1); s.dphi = s.dthe;
        Fields R(im, jm, km, s), D(im, jm, km, s);
        init_hyd(R, 13, s); init_hyd(D, 13, s);
        for (Fields* F : { &R, &D }) {
            std::mt19937_64 rng(14);
            for (const char* n : { "Salt_Balance", "Salt_Finger", "Salt_Diffusion", "Buoyancy_Force" }) fill((*F)[n], im, jm, km, rng, -1., 1.);
        }
        const std::string dr_ = tmp + "/bridge_ref", dd_ = tmp + "/bridge_dev";
        (void)std::system(("mkdir -p " + dr_ + " " + dd_).c_str());
        { Quiet q; results_calls<RefClasses>(R, s, dr_); }
        results_calls<DevClasses>(D, s, dd_);
        bad += compare(R, D, 0., "hydrosphere, Results_Hyd::run_data");
        bad += same_file(dr_ + "/[bridge]_PlotData_Hyd.xyz", dd_ + "/[bridge]_PlotData_Hyd.xyz", "hydrosphere, PlotData file");
        Scalars sa; sa.re = 1000.; sa.pr = .7179; sa.u_0 = 8.; sa.c_0 = .035;
        sa.dr = 1. / (im - 1); sa.dthe = M_PI / (jm - 1); sa.dphi = sa.dthe;
        Fields RA(im, jm, km, sa), DA(im, jm, km, sa);
        init_msl(RA, 15, sa); init_msl(DA, 15, sa);
        { Quiet q; atm_plot_calls<RefClasses>(RA, sa, dr_); }
        atm_plot_calls<DevClasses>(DA, sa, dd_);
        bad += same_file(dr_ + "/[bridge]_PlotData_Atm_3.xyz", dd_ + "/[bridge]_PlotData_Atm_3.xyz", "atmosphere, PlotData file");
    }
    {   // the ParaView section files of both models, and what their writers leave in aux_u / aux_v / aux_w
        const int im = 9, jm = 21, km = 31;
        const std::string dr_ = tmp + "/bridge_ref", dd_ = tmp + "/bridge_dev";
        (void)std::system(("mkdir -p " + dr_ + " " + dd_).c_str());
        Scalars sa; sa.re = 1000.; sa.pr = .7179; sa.u_0 = 8.; sa.c_0 = .035;
        sa.dr = 1. / (im - 1); sa.dthe = M_PI / (jm - 1); sa.dphi = sa.dthe;
        Fields RA(im, jm, km, sa), DA(im, jm, km, sa);
        init_msl(RA, 17, sa); init_msl(DA, 17, sa);
        for (Fields* F : { &RA, &DA }) {
            std::mt19937_64 rng(18);
            fill((*F)["epsilon_3D"], im, jm, km, rng, 0., 1.); fill((*F)["S_c_c"], im, jm, km, rng, -1e-6, 1e-6);
            std::uniform_real_distribution<double> d(0., 1.2);
            for (const char* n : { "temperature_NASA", "precipitation_NASA", "Vegetation", "albedo", "epsilon", "precipitable_water",
                                   "Evaporation_Penman", "temp_NASA" })
                for (int j = 0; j < jm; j++) for (int k = 0; k < km; k++) F->g(n).y[j][k] = d(rng);
        }
        { Quiet q; atm_vtk_calls<RefClasses>(RA, sa, dr_, 0, 7, 11); }
        atm_vtk_calls<DevClasses>(DA, sa, dd_, 0, 7, 11);
        bad += compare(RA, DA, 0., "atmosphere, ParaView sections");
        bad += same_file(dr_ + "/[bridge]_Atm_panorama_0.vts", dd_ + "/[bridge]_Atm_panorama_0.vts", "atmosphere, panorama file");
        bad += same_file(dr_ + "/[bridge]_Atm_radial_0_0.vtk", dd_ + "/[bridge]_Atm_radial_0_0.vtk", "atmosphere, radial section file");
        bad += same_file(dr_ + "/[bridge]_Atm_longal_7_0.vtk", dd_ + "/[bridge]_Atm_longal_7_0.vtk", "atmosphere, longitudinal section file");
        bad += same_file(dr_ + "/[bridge]_Atm_zonal_11_0.vtk", dd_ + "/[bridge]_Atm_zonal_11_0.vtk", "atmosphere, zonal section file");
        Scalars s; s.re = 10.; s.pr = 6.957; s.u_0 = .25; s.c_0 = 34.6;
        s.dr = sa.dr; s.dthe = sa.dthe; s.dphi = sa.dphi;
        Fields R(im, jm, km, s), D(im, jm, km, s);
        init_hyd(R, 19, s); init_hyd(D, 19, s);
        for (Fields* F : { &R, &D }) {
            std::mt19937_64 rng(20);
            for (const char* n : { "Salt_Balance", "Salt_Finger", "Salt_Diffusion", "Buoyancy_Force", "p_stat" }) fill((*F)[n], im, jm, km, rng, -1., 1.);
            std::uniform_real_distribution<double> d(-2., 2.);
            for (const char* n : { "Upwelling", "Downwelling", "BottomWater" })
                for (int j = 0; j < jm; j++) for (int k = 0; k < km; k++) F->g(n).y[j][k] = d(rng);
        }
        { Quiet q; hyd_vtk_calls<RefClasses>(R, s, dr_, im - 1, 7, 11); }
        hyd_vtk_calls<DevClasses>(D, s, dd_, im - 1, 7, 11);
        bad += compare(R, D, 0., "hydrosphere, ParaView sections");
        bad += same_file(dr_ + "/[bridge]_Hyd_panorama_0.vts", dd_ + "/[bridge]_Hyd_panorama_0.vts", "hydrosphere, panorama file");
        bad += same_file(dr_ + "/[bridge]_Hyd_radial_8_0.vtk", dd_ + "/[bridge]_Hyd_radial_8_0.vtk", "hydrosphere, radial section file");
        bad += same_file(dr_ + "/[bridge]_Hyd_longal_7_0.vtk", dd_ + "/[bridge]_Hyd_longal_7_0.vtk", "hydrosphere, longitudinal section file");
        bad += same_file(dr_ + "/[bridge]_Hyd_zonal_11_0.vtk", dd_ + "/[bridge]_Hyd_zonal_11_0.vtk", "hydrosphere, zonal section file");
    }
    {   // the hydrosphere's start of a time slice; the Ekman bands are rows of the 1-degree grid
        Scalars s; s.re = 10.; s.pr = 6.957; s.u_0 = .25; s.c_0 = 34.6;
        const int im = 9, jm = 181, km = 24;
        s.dr = 1. / (im - 1); s.dthe = M_PI / (jm - 1); s.dphi = s.dthe;
        const std::string bath = tmp + "/bridge_bathymetry.xyz", salt = tmp + "/bridge_salinity.xyz";
        {
            std::mt19937_64 rng(21);
            std::uniform_real_distribution<double> z(-1400., 300.), c(-5., 38.);
            std::ofstream fb(bath), fs(salt);
            for (int j = 0; j < jm; j++) for (int k = 0; k < km; k++) fb << k - 180 << " " << 90 - j << " " << z(rng) << "\n";
            for (int k = 0; k < km; k++) for (int j = 0; j < jm; j++) fs << k << " " << 90 - j << " " << c(rng) << "\n";
        }
        Fields R(im, jm, km, s), D(im, jm, km, s);
        init_hyd(R, 22, s); init_hyd(D, 22, s);
        for (Fields* F : { &R, &D }) {
            std::mt19937_64 rng(23);
            std::uniform_real_distribution<double> wind(-10., 10.);
            for (int j = 0; j < jm; j++) for (int k = 0; k < km; k++) {
                (*F)["v"].x[im - 1][j][k] = wind(rng); (*F)["w"].x[im - 1][j][k] = wind(rng);
            }
            (void)(*F)["p_stat"];
        }
        { Quiet q; hyd_start_calls<RefClasses>(R, s, bath, salt); }
        hyd_start_calls<DevClasses>(D, s, bath, salt);
        bad += compare(R, D, 1e-10, "hydrosphere, start of a time slice");
    }
    std::printf(bad ? "BRIDGE FAIL\n" : "BRIDGE PASS\n");
    return bad ? 1 : 0;
}
