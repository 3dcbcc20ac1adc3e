// atomdev_mirror — the reference's dynamical-core operator classes, same names, same
// constructor and method signatures, executing on the GPU through libatomdev.so.
//
// For a maintainer of the reference: include this header next to lib/Array.h,
// lib/Array_1D.h and lib/Array_2D.h (it only uses their public members x, y, z) and
// replace e.g. `RungeKutta_Atmosphere result(...)` by
// `atomdev_mirror::RungeKutta_Atmosphere result(...)`; the call sites in
// cAtmosphereModel::run_3D_loop (atmosphere/cAtmosphereModel.cpp:777-874) and
// cHydrosphereModel::RunTimeSlice (hydrosphere/cHydrosphereModel.cpp:437-610) stay as
// they are.  Every call uploads the arrays the operator reads, runs it, and downloads the
// arrays it writes, so the classes are drop-in but pay two PCIe transfers per call; a port
// that keeps the fields resident uses the C-ABI (atomdev.h) directly.
//
// Scalars.  The reference hands each operator only the scalars it needs, with every call.
// The mirror keeps one parameter block per model, updates it from whatever a constructor
// or method receives, and pushes it to the device handle (atomdev_set_params) before the
// operator runs.  Scalars an operator never reads (lv, ls, ep, hp, co2_0, p_0, ...) are
// accepted and ignored, like the reference's own bodies ignore them.
//
// Errors.  The reference prints and aborts (e.g. atmosphere/cAtmosphereModel.cpp:132-135);
// so does the mirror, with the library's error text.
#ifndef ATOMDEV_MIRROR_HPP_
#define ATOMDEV_MIRROR_HPP_

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>
#include <vector>

#include "Array.h"
#include "Array_1D.h"
#include "Array_2D.h"
#include "atomdev.h"

namespace atomdev_mirror {

// One device handle per model, created on first use and re-created when the grid changes.
class Device {
public:
    static Device& of(int model) {
        static Device dev[2];
        dev[model].model_ = model;
        return dev[model];
    }
    atomdev_params& params() {
        if (!have_params_) { atomdev_default_params(model_, &p_); have_params_ = true; }
        return p_;
    }
    // coordinates as the reference builds them: z[0] and the step (lib/Array_1D.cpp:54-65)
    void grid(int im, int jm, int km) {
        if (h_ && (im != im_ || jm != jm_ || km != km_)) { atomdev_destroy(h_); h_ = nullptr; }
        im_ = im; jm_ = jm; km_ = km;
    }
    void origin(Array_1D& rad, Array_1D& the) { params().r0 = rad.z[0]; params().the0 = the.z[0]; }
    atomdev_t* handle() {
        if (!h_) {
            if (atomdev_create(model_, im_, jm_, km_, &params(), -1, &h_)) die("atomdev_create");
            // drop-in classes: the sweeps in their exact form (the hydrosphere's results carry the reference's bits)
            // unless the caller asks for a faster one
            if (!std::getenv("ATOMDEV_RK_VARIANT") && atomdev_set_rk_variant(h_, 0)) die("atomdev_set_rk_variant");
            stage_.resize((size_t)im_ * jm_ * km_);
            std::memcpy(&sent_, &p_, sizeof(p_));
        } else if (std::memcmp(&sent_, &p_, sizeof(p_)) != 0) {
            if (atomdev_set_params(h_, &p_)) die("atomdev_set_params");
            std::memcpy(&sent_, &p_, sizeof(p_));
        }
        return h_;
    }
    void push(int field, Array& a) {
        atomdev_t* h = handle();
        for (int i = 0; i < im_; i++)
            for (int j = 0; j < jm_; j++)
                std::memcpy(&stage_[((size_t)i * jm_ + j) * km_], a.x[i][j], sizeof(double) * km_);
        if (atomdev_upload(h, field, stage_.data())) die("atomdev_upload");
    }
    void pull(int field, Array& a) {
        if (atomdev_download(handle(), field, stage_.data())) die("atomdev_download");
        for (int i = 0; i < im_; i++)
            for (int j = 0; j < jm_; j++)
                std::memcpy(a.x[i][j], &stage_[((size_t)i * jm_ + j) * km_], sizeof(double) * km_);
    }
    void push2d(int field, Array_2D& a) {
        atomdev_t* h = handle();
        for (int j = 0; j < jm_; j++) std::memcpy(&stage_[(size_t)j * km_], a.y[j], sizeof(double) * km_);
        if (atomdev_upload2d(h, field, stage_.data())) die("atomdev_upload2d");
    }
    void pull2d(int field, Array_2D& a) {
        if (atomdev_download2d(handle(), field, stage_.data())) die("atomdev_download2d");
        for (int j = 0; j < jm_; j++) std::memcpy(a.y[j], &stage_[(size_t)j * km_], sizeof(double) * km_);
    }
    void run(int rc, const char* what) { if (rc) die(what); }
    static void die(const char* where) {
        std::cerr << "ERROR: " << where << ": " << atomdev_last_error() << std::endl;
        std::abort();
    }
    int im() const { return im_; }

private:
    int model_ = 0, im_ = 0, jm_ = 0, km_ = 0;
    bool have_params_ = false;
    atomdev_params p_, sent_;
    atomdev_t* h_ = nullptr;
    std::vector<double> stage_;
};

struct FieldRef { int id; Array* a; };

inline void push_all(Device& d, std::initializer_list<FieldRef> l) { for (const FieldRef& f : l) d.push(f.id, *f.a); }
inline void pull_all(Device& d, std::initializer_list<FieldRef> l) { for (const FieldRef& f : l) d.pull(f.id, *f.a); }

// ---------------------------------------------------------------------------------------------
// atmosphere
// ---------------------------------------------------------------------------------------------

// atmosphere/RHS_Atm.h:23-60.  Holds the scalars of the right-hand sides; the evaluation itself is
// fused into the sweeps of RungeKutta_Atmosphere on the device.
class RHS_Atmosphere {
public:
    RHS_Atmosphere(int jm, int km, double dthe, double dphi, double re) {
        atomdev_params& p = Device::of(ATOMDEV_ATM).params();
        (void)jm; (void)km; p.dthe = dthe; p.dphi = dphi; p.re = re;
    }
    RHS_Atmosphere(int im, int jm, int km, double dt, double dr, double dthe, double dphi, double re,
                   double sc_WaterVapour, double sc_CO2, double g, double pr, double WaterVapour, double Buoyancy,
                   double CO2, double gam, double sigma, double lamda) {
        atomdev_params& p = Device::of(ATOMDEV_ATM).params();
        (void)im; (void)jm; (void)km; (void)WaterVapour; (void)CO2; (void)sigma; (void)lamda;
        p.dt = dt; p.dr = dr; p.dthe = dthe; p.dphi = dphi; p.re = re; p.sc_WaterVapour = sc_WaterVapour;
        p.sc_CO2 = sc_CO2; p.g = g; p.pr = pr; p.Buoyancy = Buoyancy; p.gam = gam;
    }
};

// atmosphere/RungeKutta_Atm.h:22-52
class RungeKutta_Atmosphere {
public:
    RungeKutta_Atmosphere(int im, int jm, int km, double dt, double dr, double dphi, double dthe)
        : im_(im), jm_(jm), km_(km) {
        atomdev_params& p = Device::of(ATOMDEV_ATM).params();
        p.dt = dt; p.dr = dr; p.dphi = dphi; p.dthe = dthe;
    }

    // atmosphere/RungeKutta_Atm.cpp:32-156
    void solveRungeKutta_3D_Atmosphere(RHS_Atmosphere&, int&, double, double, double, double, double u_0, double t_0,
                                       double c_0, double, double, double r_air, double, double, double L_atm, double cp_l,
                                       double R_Air, double R_WaterVapour, double, Array_1D& rad, Array_1D& the, Array_1D&,
                                       Array&, Array&, Array&, Array&, Array&, Array&, Array&, Array&, Array& h, Array& t,
                                       Array& u, Array& v, Array& w, Array& p_dyn, Array&, Array& c, Array& cloud, Array& ice,
                                       Array& co2, Array& tn, Array& un, Array& vn, Array& wn, Array& p_dynn, Array& cn,
                                       Array& cloudn, Array& icen, Array& co2n, Array& aux_u, Array& aux_v, Array& aux_w,
                                       Array&, Array& BuoyancyForce, Array&, Array&, Array&, Array& S_v, Array& S_c,
                                       Array& S_i, Array& S_r, Array& S_s, Array&, Array_2D& Topography,
                                       Array_2D& Evaporation_Dalton, Array_2D& Precipitation) {
        Device& d = Device::of(ATOMDEV_ATM);
        d.grid(im_, jm_, km_); d.origin(rad, the);
        atomdev_params& p = d.params();
        p.u_0 = u_0; p.t_0 = t_0; p.c_0 = c_0; p.r_air = r_air; p.L_atm = L_atm; p.cp_l = cp_l; p.R_Air = R_Air;
        p.R_WaterVapour = R_WaterVapour;
        push_all(d, { { ATOMDEV_H, &h }, { ATOMDEV_T, &t }, { ATOMDEV_U, &u }, { ATOMDEV_V, &v }, { ATOMDEV_W, &w },
                      { ATOMDEV_P_DYN, &p_dyn }, { ATOMDEV_C, &c }, { ATOMDEV_CLOUD, &cloud }, { ATOMDEV_ICE, &ice },
                      { ATOMDEV_CO2, &co2 }, { ATOMDEV_TN, &tn }, { ATOMDEV_UN, &un }, { ATOMDEV_VN, &vn },
                      { ATOMDEV_WN, &wn }, { ATOMDEV_P_DYNN, &p_dynn }, { ATOMDEV_CN, &cn }, { ATOMDEV_CLOUDN, &cloudn },
                      { ATOMDEV_ICEN, &icen }, { ATOMDEV_CO2N, &co2n }, { ATOMDEV_S_V, &S_v }, { ATOMDEV_S_C, &S_c },
                      { ATOMDEV_S_I, &S_i }, { ATOMDEV_S_R, &S_r }, { ATOMDEV_S_S, &S_s } });
        d.push2d(ATOMDEV_TOPOGRAPHY, Topography);
        d.push2d(ATOMDEV_EVAPORATION_DALTON, Evaporation_Dalton);
        d.push2d(ATOMDEV_PRECIPITATION, Precipitation);
        d.run(atomdev_atm_rk3d(d.handle()), "atomdev_atm_rk3d");
        pull_all(d, { { ATOMDEV_T, &t }, { ATOMDEV_U, &u }, { ATOMDEV_V, &v }, { ATOMDEV_W, &w }, { ATOMDEV_C, &c },
                      { ATOMDEV_CLOUD, &cloud }, { ATOMDEV_ICE, &ice }, { ATOMDEV_CO2, &co2 }, { ATOMDEV_AUX_U, &aux_u },
                      { ATOMDEV_AUX_V, &aux_v }, { ATOMDEV_AUX_W, &aux_w }, { ATOMDEV_BUOYANCY, &BuoyancyForce } });
    }

    // atmosphere/RungeKutta_Atm.cpp:163-217
    void solveRungeKutta_2D_Atmosphere(RHS_Atmosphere&, int&, double r_air, double u_0, double, double L_atm, Array_1D& rad,
                                       Array_1D& the, Array_1D&, Array&, Array&, Array& h, Array& v, Array& w, Array& p_dyn,
                                       Array& vn, Array& wn, Array& p_dynn, Array& aux_v, Array& aux_w) {
        Device& d = Device::of(ATOMDEV_ATM);
        d.grid(im_, jm_, km_); d.origin(rad, the);
        atomdev_params& p = d.params();
        p.r_air = r_air; p.u_0 = u_0; p.L_atm = L_atm;
        push_all(d, { { ATOMDEV_H, &h }, { ATOMDEV_V, &v }, { ATOMDEV_W, &w }, { ATOMDEV_P_DYN, &p_dyn }, { ATOMDEV_VN, &vn },
                      { ATOMDEV_WN, &wn }, { ATOMDEV_P_DYNN, &p_dynn }, { ATOMDEV_AUX_V, &aux_v }, { ATOMDEV_AUX_W, &aux_w } });
        d.run(atomdev_atm_rk2d(d.handle()), "atomdev_atm_rk2d");
        pull_all(d, { { ATOMDEV_V, &v }, { ATOMDEV_W, &w }, { ATOMDEV_AUX_V, &aux_v }, { ATOMDEV_AUX_W, &aux_w } });
    }

private:
    int im_, jm_, km_;
};

// atmosphere/Pressure_Atm.h
class Pressure_Atm {
public:
    Pressure_Atm(int im, int jm, int km, double dr, double dthe, double dphi) : im_(im), jm_(jm), km_(km) {
        atomdev_params& p = Device::of(ATOMDEV_ATM).params();
        p.dr = dr; p.dthe = dthe; p.dphi = dphi;
    }
    // atmosphere/Pressure_Atm.cpp:39-190
    void computePressure_3D(double u_0, double r_air, Array_1D& rad, Array_1D& the, Array& p_dyn, Array& p_dynn, Array& h,
                            Array& aux_u, Array& aux_v, Array& aux_w) {
        Device& d = begin(u_0, r_air, rad, the);
        push_all(d, { { ATOMDEV_P_DYN, &p_dyn }, { ATOMDEV_P_DYNN, &p_dynn }, { ATOMDEV_H, &h }, { ATOMDEV_AUX_U, &aux_u },
                      { ATOMDEV_AUX_V, &aux_v }, { ATOMDEV_AUX_W, &aux_w } });
        d.run(atomdev_atm_pressure3d(d.handle()), "atomdev_atm_pressure3d");
        pull_all(d, { { ATOMDEV_P_DYN, &p_dyn }, { ATOMDEV_AUX_U, &aux_u }, { ATOMDEV_AUX_V, &aux_v }, { ATOMDEV_AUX_W, &aux_w } });
    }
    // atmosphere/Pressure_Atm.cpp:195-301
    void computePressure_2D(double u_0, double r_air, Array_1D& rad, Array_1D& the, Array& p_dyn, Array& p_dynn, Array& h,
                            Array& aux_v, Array& aux_w) {
        Device& d = begin(u_0, r_air, rad, the);
        push_all(d, { { ATOMDEV_P_DYN, &p_dyn }, { ATOMDEV_P_DYNN, &p_dynn }, { ATOMDEV_H, &h }, { ATOMDEV_AUX_V, &aux_v },
                      { ATOMDEV_AUX_W, &aux_w } });
        d.run(atomdev_atm_pressure2d(d.handle()), "atomdev_atm_pressure2d");
        pull_all(d, { { ATOMDEV_P_DYN, &p_dyn }, { ATOMDEV_AUX_V, &aux_v } });
    }

private:
    Device& begin(double u_0, double r_air, Array_1D& rad, Array_1D& the) {
        Device& d = Device::of(ATOMDEV_ATM);
        d.grid(im_, jm_, km_); d.origin(rad, the);
        d.params().u_0 = u_0; d.params().r_air = r_air;
        return d;
    }
    int im_, jm_, km_;
};

// atmosphere/BC_Atm.h
class BC_Atmosphere {
public:
    BC_Atmosphere(int im, int jm, int km, double t_tropopause) : im_(im), jm_(jm), km_(km) {
        Device::of(ATOMDEV_ATM).params().t_tropopause = t_tropopause;
    }
#define ATOMDEV_MIRROR_ATM_BC(NAME, ENTRY)                                                                             \
    void NAME(Array& t, Array& u, Array& v, Array& w, Array& p_dyn, Array& c, Array& cloud, Array& ice, Array& co2) { \
        Device& d = Device::of(ATOMDEV_ATM);                                                                           \
        d.grid(im_, jm_, km_);                                                                                         \
        push_all(d, { { ATOMDEV_T, &t }, { ATOMDEV_U, &u }, { ATOMDEV_V, &v }, { ATOMDEV_W, &w }, { ATOMDEV_P_DYN, &p_dyn }, \
                      { ATOMDEV_C, &c }, { ATOMDEV_CLOUD, &cloud }, { ATOMDEV_ICE, &ice }, { ATOMDEV_CO2, &co2 } });    \
        d.run(ENTRY(d.handle()), #ENTRY);                                                                              \
        pull_all(d, { { ATOMDEV_T, &t }, { ATOMDEV_U, &u }, { ATOMDEV_V, &v }, { ATOMDEV_W, &w }, { ATOMDEV_P_DYN, &p_dyn }, \
                      { ATOMDEV_C, &c }, { ATOMDEV_CLOUD, &cloud }, { ATOMDEV_ICE, &ice }, { ATOMDEV_CO2, &co2 } });    \
    }
    ATOMDEV_MIRROR_ATM_BC(BC_radius, atomdev_atm_bc_radius)  // atmosphere/BC_Atm.cpp:39-73
    ATOMDEV_MIRROR_ATM_BC(BC_theta, atomdev_atm_bc_theta)    // atmosphere/BC_Atm.cpp:80-107
    ATOMDEV_MIRROR_ATM_BC(BC_phi, atomdev_atm_bc_phi)        // atmosphere/BC_Atm.cpp:115-150
#undef ATOMDEV_MIRROR_ATM_BC

private:
    int im_, jm_, km_;
};

// atmosphere/Results_Atm.h.  run_MSL_data without its printout (Results_Atm.cpp:355-625 prints averages and
// values at three fixed 1-degree locations; nothing of that is fed back into the model).
class Results_MSL_Atm {
public:
    Results_MSL_Atm(int im, int jm, int km, int, double g, double ep, double hp, double u_0, double p_0, double t_0, double,
                    double, double, double, double lv, double ls, double cp_l, double L_atm, double, double, double, double,
                    double r_air, double R_Air, double, double R_WaterVapour, double, double, double, double, double, double,
                    double)
        : im_(im), jm_(jm), km_(km) {
        atomdev_params& p = Device::of(ATOMDEV_ATM).params();
        p.g = g; p.ep = ep; p.hp = hp; p.u_0 = u_0; p.p_0 = p_0; p.t_0 = t_0; p.lv = lv; p.ls = ls; p.cp_l = cp_l;
        p.L_atm = L_atm; p.r_air = r_air; p.R_Air = R_Air; p.R_WaterVapour = R_WaterVapour;
    }
    // atmosphere/Results_Atm.cpp:118-353
    void run_MSL_data(int, int, int RadiationModel, double&, Array_1D&, Array_1D&, Array_1D&, Array& h, Array& c, Array&,
                      Array& co2, Array&, Array& t, Array&, Array&, Array& p_stat, Array& BuoyancyForce, Array&, Array& v,
                      Array& w, Array& Q_Latent, Array& Q_Sensible, Array& radiation_3D, Array& cloud, Array&, Array& ice,
                      Array&, Array& P_rain, Array& P_snow, Array&, Array&, Array&, Array_2D&, Array_2D&,
                      Array_2D& precipitable_water, Array_2D& Q_radiation, Array_2D& Q_Evaporation, Array_2D& Q_latent,
                      Array_2D& Q_sensible, Array_2D& Q_bottom, Array_2D& Evaporation_Penman, Array_2D& Evaporation_Dalton,
                      Array_2D&, Array_2D&, Array_2D& co2_total, Array_2D& Precipitation, Array&, Array&, Array&, Array&,
                      Array&, Array&) {
        Device& d = Device::of(ATOMDEV_ATM);
        d.grid(im_, jm_, km_);
        push_all(d, { { ATOMDEV_H, &h }, { ATOMDEV_T, &t }, { ATOMDEV_C, &c }, { ATOMDEV_V, &v }, { ATOMDEV_W, &w },
                      { ATOMDEV_CO2, &co2 }, { ATOMDEV_CLOUD, &cloud }, { ATOMDEV_ICE, &ice }, { ATOMDEV_P_STAT, &p_stat },
                      { ATOMDEV_BUOYANCY, &BuoyancyForce }, { ATOMDEV_Q_LATENT, &Q_Latent }, { ATOMDEV_Q_SENSIBLE, &Q_Sensible },
                      { ATOMDEV_RADIATION_3D, &radiation_3D }, { ATOMDEV_P_RAIN, &P_rain }, { ATOMDEV_P_SNOW, &P_snow } });
        // columns without a sea surface or a summit keep what these held
        d.push2d(ATOMDEV_Q_RADIATION, Q_radiation); d.push2d(ATOMDEV_Q_EVAPORATION, Q_Evaporation);
        d.push2d(ATOMDEV_Q_LATENT_2D, Q_latent); d.push2d(ATOMDEV_Q_SENSIBLE_2D, Q_sensible); d.push2d(ATOMDEV_Q_BOTTOM, Q_bottom);
        d.run(atomdev_atm_msl_data(d.handle(), RadiationModel), "atomdev_atm_msl_data");
        pull_all(d, { { ATOMDEV_C, &c }, { ATOMDEV_CLOUD, &cloud }, { ATOMDEV_ICE, &ice }, { ATOMDEV_P_STAT, &p_stat },
                      { ATOMDEV_BUOYANCY, &BuoyancyForce }, { ATOMDEV_P_RAIN, &P_rain }, { ATOMDEV_P_SNOW, &P_snow } });
        d.pull2d(ATOMDEV_Q_RADIATION, Q_radiation); d.pull2d(ATOMDEV_Q_EVAPORATION, Q_Evaporation);
        d.pull2d(ATOMDEV_Q_LATENT_2D, Q_latent); d.pull2d(ATOMDEV_Q_SENSIBLE_2D, Q_sensible); d.pull2d(ATOMDEV_Q_BOTTOM, Q_bottom);
        d.pull2d(ATOMDEV_PRECIPITABLE_WATER, precipitable_water); d.pull2d(ATOMDEV_EVAPORATION_PENMAN, Evaporation_Penman);
        d.pull2d(ATOMDEV_EVAPORATION_DALTON, Evaporation_Dalton); d.pull2d(ATOMDEV_CO2_TOTAL, co2_total);
        d.pull2d(ATOMDEV_PRECIPITATION, Precipitation);
    }

private:
    int im_, jm_, km_;
};

// atmosphere/BC_Bath_Atm.h: the two members the iteration loop calls; the mask reader stays on the host.
class BC_Bathymetry_Atmosphere {
public:
    BC_Bathymetry_Atmosphere(int, int im, int jm, int km, double, double, double) : im_(im), jm_(jm), km_(km) {}
    // atmosphere/BC_Bath_Atm.cpp:39-90
    void BC_MountainSurface(std::string& topo_filename, double L_atm, Array_2D& Topography, Array& h) {
        Device& d = Device::of(ATOMDEV_ATM);
        d.grid(im_, jm_, km_);
        d.params().L_atm = L_atm;
        d.run(atomdev_atm_read_mountain_surface(d.handle(), topo_filename.c_str()), "atomdev_atm_read_mountain_surface");
        d.pull(ATOMDEV_H, h);
        d.pull2d(ATOMDEV_TOPOGRAPHY, Topography);
    }
    // atmosphere/BC_Bath_Atm.cpp:94-119: only h and the seven reset fields are touched, most scalars are unused
    void BC_SolidGround(int, int, double g, double, double, double r_air, double R_Air, double t_0, double c_0, double, double,
                        double, double, double t_tropopause, double, double, double, double, double, double, double,
                        double gam, double, Array& h, Array& u, Array& v, Array& w, Array& t, Array& p_dyn, Array& c,
                        Array& cloud, Array& ice, Array& co2, Array&, Array_2D&) {
        Device& d = Device::of(ATOMDEV_ATM);
        d.grid(im_, jm_, km_);
        atomdev_params& p = d.params();
        p.g = g; p.r_air = r_air; p.R_Air = R_Air; p.t_0 = t_0; p.c_0 = c_0; p.t_tropopause = t_tropopause; p.gam = gam;
        push_all(d, { { ATOMDEV_H, &h }, { ATOMDEV_U, &u }, { ATOMDEV_V, &v }, { ATOMDEV_W, &w }, { ATOMDEV_T, &t },
                      { ATOMDEV_P_DYN, &p_dyn }, { ATOMDEV_C, &c }, { ATOMDEV_CLOUD, &cloud }, { ATOMDEV_ICE, &ice },
                      { ATOMDEV_CO2, &co2 } });
        d.run(atomdev_atm_solid_ground(d.handle()), "atomdev_atm_solid_ground");
        pull_all(d, { { ATOMDEV_U, &u }, { ATOMDEV_V, &v }, { ATOMDEV_W, &w }, { ATOMDEV_T, &t }, { ATOMDEV_P_DYN, &p_dyn },
                      { ATOMDEV_C, &c }, { ATOMDEV_CLOUD, &cloud }, { ATOMDEV_ICE, &ice }, { ATOMDEV_CO2, &co2 } });
    }
    // atmosphere/BC_Bath_Atm.cpp:129-142
    void vegetationDistribution(double max_Precipitation, Array_2D& Precipitation, Array_2D& Vegetation, Array& t, Array& h) {
        Device& d = Device::of(ATOMDEV_ATM);
        d.grid(im_, jm_, km_);
        push_all(d, { { ATOMDEV_H, &h }, { ATOMDEV_T, &t } });
        d.push2d(ATOMDEV_PRECIPITATION, Precipitation); d.push2d(ATOMDEV_VEGETATION, Vegetation);
        d.run(atomdev_atm_vegetation_distribution(d.handle(), max_Precipitation), "atomdev_atm_vegetation_distribution");
        d.pull2d(ATOMDEV_VEGETATION, Vegetation);
    }

private:
    int im_, jm_, km_;
};

// atmosphere/BC_Thermo.h: the members cAtmosphereModel::run_3D_loop calls (cAtmosphereModel.cpp:788-865); the
// initial-condition members (BC_Temperature, BC_WaterVapour, IC_CellStructure, ...) run once per time slice and stay
// on the host.  The constructor reads the same public members of the model object as BC_Thermo.cpp:30-94 does; it is a
// template so that this header needs no model class of its own.
class BC_Thermo {
public:
    template <class Model>
    BC_Thermo(Model* model, int im, int jm, int km, Array& h) : im_(im), jm_(jm), km_(km), h_(&h) {
        Device& d = Device::of(ATOMDEV_ATM);
        d.grid(im, jm, km);
        atomdev_params& p = d.params();
        p.L_atm = model->L_atm; p.dt = model->dt; p.dr = model->dr; p.dthe = model->dthe; p.dphi = model->dphi;
        p.g = model->g; p.ep = model->ep; p.hp = model->hp; p.u_0 = model->u_0; p.p_0 = model->p_0; p.t_0 = model->t_0;
        p.gam = model->gam; p.lv = model->lv; p.ls = model->ls; p.cp_l = model->cp_l; p.r_air = model->r_air;
        p.R_Air = model->R_Air; p.r_water_vapour = model->r_water_vapour; p.R_WaterVapour = model->R_WaterVapour;
        p.t_tropopause = model->t_tropopause;
        atomdev_default_radiation(&rad_);
        rad_.sigma = model->sigma; rad_.albedo_equator = model->albedo_equator; rad_.albedo_pole = model->albedo_pole;
        rad_.rad_equator = model->rad_equator; rad_.rad_pole = model->rad_pole; rad_.epsilon_pole = model->epsilon_pole;
        rad_.epsilon_tropopause = model->epsilon_tropopause; rad_.epsilon_equator = model->epsilon_equator;
        rad_.co2_coeff = std::fabs(model->CO2 - 1) < 2.220446049250313e-16   // BC_Thermo.cpp:252-257
                             ? model->co2_factor * (model->co2_equator / model->co2_tropopause) : 1.;
        im_tropopause_ = model->get_tropopause();
        const int Ma = int(round(*model->get_current_time()));
        t_cretaceous_ = model->get_mean_temperature_from_curve(Ma) - model->get_mean_temperature_from_curve(0);
        t_0_ = model->t_0;
    }
    // atmosphere/BC_Thermo.cpp:3235-3275
    void Value_Limitation_Atm(Array& h, Array& u, Array& v, Array& w, Array& p_dyn, Array& t, Array& c, Array& cloud,
                              Array& ice, Array& co2) {
        Device& d = start();
        push_all(d, { { ATOMDEV_H, &h }, { ATOMDEV_U, &u }, { ATOMDEV_V, &v }, { ATOMDEV_W, &w }, { ATOMDEV_P_DYN, &p_dyn },
                      { ATOMDEV_T, &t }, { ATOMDEV_C, &c }, { ATOMDEV_CLOUD, &cloud }, { ATOMDEV_ICE, &ice }, { ATOMDEV_CO2, &co2 } });
        d.run(atomdev_atm_value_limitation(d.handle()), "atomdev_atm_value_limitation");
        pull_all(d, { { ATOMDEV_U, &u }, { ATOMDEV_V, &v }, { ATOMDEV_W, &w }, { ATOMDEV_P_DYN, &p_dyn }, { ATOMDEV_T, &t },
                      { ATOMDEV_C, &c }, { ATOMDEV_CLOUD, &cloud }, { ATOMDEV_ICE, &ice }, { ATOMDEV_CO2, &co2 } });
    }
    // atmosphere/BC_Thermo.cpp:166-424
    void BC_Radiation_multi_layer(Array_2D& albedo, Array_2D& epsilon, Array_2D& radiation_surface, Array& p_stat, Array& t,
                                  Array& c, Array& h, Array& epsilon_3D, Array& radiation_3D, Array& cloud, Array& ice, Array&) {
        Device& d = start();
        push_all(d, { { ATOMDEV_H, &h }, { ATOMDEV_P_STAT, &p_stat }, { ATOMDEV_T, &t }, { ATOMDEV_C, &c },
                      { ATOMDEV_CLOUD, &cloud }, { ATOMDEV_ICE, &ice }, { ATOMDEV_EPSILON_3D, &epsilon_3D },
                      { ATOMDEV_RADIATION_3D, &radiation_3D } });
        d.push2d(ATOMDEV_ALBEDO, albedo); d.push2d(ATOMDEV_EPSILON, epsilon); d.push2d(ATOMDEV_RADIATION_SURFACE, radiation_surface);
        std::vector<int> i_trop(jm_);
        for (int j = 0; j < jm_; j++) i_trop[j] = im_tropopause_[j] + GetTropopauseHightAdd(t_cretaceous_ / t_0_);  // :226
        d.run(atomdev_atm_radiation_multi_layer(d.handle(), &rad_, i_trop.data()), "atomdev_atm_radiation_multi_layer");
        pull_all(d, { { ATOMDEV_T, &t }, { ATOMDEV_C, &c }, { ATOMDEV_CLOUD, &cloud }, { ATOMDEV_ICE, &ice },
                      { ATOMDEV_EPSILON_3D, &epsilon_3D }, { ATOMDEV_RADIATION_3D, &radiation_3D } });
        d.pull2d(ATOMDEV_ALBEDO, albedo); d.pull2d(ATOMDEV_EPSILON, epsilon); d.pull2d(ATOMDEV_RADIATION_SURFACE, radiation_surface);
    }
    // atmosphere/BC_Thermo.cpp:2509-2796; q_v_hyp is a member there too
    void Ice_Water_Saturation_Adjustment(Array& h, Array& c, Array& cn, Array& cloud, Array& cloudn, Array& ice, Array& icen,
                                         Array& t, Array&, Array& S_c_c) {
        Device& d = start();
        push_all(d, { { ATOMDEV_H, &h }, { ATOMDEV_C, &c }, { ATOMDEV_CN, &cn }, { ATOMDEV_CLOUD, &cloud },
                      { ATOMDEV_CLOUDN, &cloudn }, { ATOMDEV_ICE, &ice }, { ATOMDEV_ICEN, &icen }, { ATOMDEV_T, &t },
                      { ATOMDEV_S_C_C, &S_c_c } });
        d.run(atomdev_atm_ice_water_saturation_adjustment(d.handle(), &q_v_hyp_), "atomdev_atm_ice_water_saturation_adjustment");
        pull_all(d, { { ATOMDEV_C, &c }, { ATOMDEV_CN, &cn }, { ATOMDEV_CLOUD, &cloud }, { ATOMDEV_CLOUDN, &cloudn },
                      { ATOMDEV_ICE, &ice }, { ATOMDEV_ICEN, &icen }, { ATOMDEV_T, &t }, { ATOMDEV_S_C_C, &S_c_c } });
    }
    // atmosphere/BC_Thermo.cpp:2803-3143; S_nuc and S_i_dep are members there too
    void Two_Category_Ice_Scheme(Array& h, Array& c, Array& t, Array&, Array& cloud, Array& ice, Array& P_rain, Array& P_snow,
                                 Array& S_v, Array& S_c, Array& S_i, Array& S_r, Array& S_s, Array& S_c_c) {
        Device& d = start();
        push_all(d, { { ATOMDEV_H, &h }, { ATOMDEV_C, &c }, { ATOMDEV_T, &t }, { ATOMDEV_CLOUD, &cloud }, { ATOMDEV_ICE, &ice },
                      { ATOMDEV_P_RAIN, &P_rain }, { ATOMDEV_P_SNOW, &P_snow }, { ATOMDEV_S_V, &S_v }, { ATOMDEV_S_C, &S_c },
                      { ATOMDEV_S_I, &S_i }, { ATOMDEV_S_R, &S_r }, { ATOMDEV_S_S, &S_s }, { ATOMDEV_S_C_C, &S_c_c } });
        d.run(atomdev_atm_two_category_ice_scheme(d.handle(), ice_state_), "atomdev_atm_two_category_ice_scheme");
        pull_all(d, { { ATOMDEV_C, &c }, { ATOMDEV_CLOUD, &cloud }, { ATOMDEV_ICE, &ice }, { ATOMDEV_P_RAIN, &P_rain },
                      { ATOMDEV_P_SNOW, &P_snow }, { ATOMDEV_S_V, &S_v }, { ATOMDEV_S_C, &S_c }, { ATOMDEV_S_I, &S_i },
                      { ATOMDEV_S_R, &S_r }, { ATOMDEV_S_S, &S_s }, { ATOMDEV_S_C_C, &S_c_c } });
    }
    // atmosphere/BC_Thermo.cpp:2418-2502
    void Latent_Heat(Array_1D&, Array_1D&, Array_1D&, Array& h, Array& t, Array&, Array&, Array&, Array&, Array&, Array& p_stat,
                     Array& c, Array& ice, Array& Q_Latent, Array& Q_Sensible, Array&, Array_2D&, Array_2D&, Array_2D&, Array_2D&) {
        Device& d = start();
        push_all(d, { { ATOMDEV_H, &h }, { ATOMDEV_T, &t }, { ATOMDEV_P_STAT, &p_stat }, { ATOMDEV_C, &c }, { ATOMDEV_ICE, &ice },
                      { ATOMDEV_Q_LATENT, &Q_Latent }, { ATOMDEV_Q_SENSIBLE, &Q_Sensible } });
        d.run(atomdev_atm_latent_heat(d.handle()), "atomdev_atm_latent_heat");
        pull_all(d, { { ATOMDEV_Q_LATENT, &Q_Latent }, { ATOMDEV_Q_SENSIBLE, &Q_Sensible } });
    }
    // atmosphere/BC_Thermo.cpp:3299-3303
    int GetTropopauseHightAdd(double t_cret) { return (int)round((t_cret * t_0_) / 2.6); }

private:
    Device& start() { Device& d = Device::of(ATOMDEV_ATM); d.grid(im_, jm_, km_); return d; }
    int im_, jm_, km_;
    Array* h_;
    atomdev_radiation rad_;
    int* im_tropopause_ = nullptr;
    double t_cretaceous_ = 0., t_0_ = 273.15;
    double q_v_hyp_ = 0., ice_state_[2] = { 0., 0. };
};

// atmosphere/PostProcess_Atm.h: the transfer file of the coupling; the plot and VTK writers stay on the host.
class PostProcess_Atmosphere {
public:
    PostProcess_Atmosphere(int im, int jm, int km, std::string& output_path) : im_(im), jm_(jm), km_(km), out_(output_path) {}
    // atmosphere/PostProcess_Atm.cpp:98-118
    void Atmosphere_v_w_Transfer(std::string& Name_Bathymetry_File, double u_0, Array& v, Array& w, Array& t, Array& p_dyn,
                                 Array_2D& Evaporation_Dalton, Array_2D& Precipitation) {
        Device& d = Device::of(ATOMDEV_ATM);
        d.grid(im_, jm_, km_);
        d.params().u_0 = u_0;
        push_all(d, { { ATOMDEV_V, &v }, { ATOMDEV_W, &w }, { ATOMDEV_T, &t }, { ATOMDEV_P_DYN, &p_dyn } });
        d.push2d(ATOMDEV_EVAPORATION_DALTON, Evaporation_Dalton); d.push2d(ATOMDEV_PRECIPITATION, Precipitation);
        const std::string path = out_ + "/[" + Name_Bathymetry_File + "]_Transfer_Atm.vw";
        d.run(atomdev_atm_write_transfer_file(d.handle(), path.c_str()), "atomdev_atm_write_transfer_file");
    }
    // atmosphere/PostProcess_Atm.cpp:120-238
    void paraview_panorama_vts(std::string& Name_Bathymetry_File, int n, double& u_0, double& t_0, double&, double& r_air, double&,
                               double&, Array& h, Array& t, Array& p_dyn, Array& p_stat, Array& BuoyancyForce, Array& u, Array& v,
                               Array& w, Array& c, Array& co2, Array& cloud, Array& ice, Array&, Array&, Array&, Array& Q_Latent,
                               Array& Q_Sensible, Array& epsilon_3D, Array& P_rain, Array& P_snow) {
        Device& d = Device::of(ATOMDEV_ATM);
        d.grid(im_, jm_, km_);
        atomdev_params& p = d.params();
        p.u_0 = u_0; p.t_0 = t_0; p.r_air = r_air;
        push_all(d, { { ATOMDEV_H, &h }, { ATOMDEV_T, &t }, { ATOMDEV_P_DYN, &p_dyn }, { ATOMDEV_P_STAT, &p_stat },
                      { ATOMDEV_BUOYANCY, &BuoyancyForce }, { ATOMDEV_U, &u }, { ATOMDEV_V, &v }, { ATOMDEV_W, &w }, { ATOMDEV_C, &c },
                      { ATOMDEV_CO2, &co2 }, { ATOMDEV_CLOUD, &cloud }, { ATOMDEV_ICE, &ice }, { ATOMDEV_Q_LATENT, &Q_Latent },
                      { ATOMDEV_Q_SENSIBLE, &Q_Sensible }, { ATOMDEV_EPSILON_3D, &epsilon_3D }, { ATOMDEV_P_RAIN, &P_rain },
                      { ATOMDEV_P_SNOW, &P_snow } });
        const std::string path = out_ + "/[" + Name_Bathymetry_File + "]_Atm_panorama_" + std::to_string(n) + ".vts";
        d.run(atomdev_write_panorama_vts(d.handle(), path.c_str()), "atomdev_write_panorama_vts");
    }
    // atmosphere/PostProcess_Atm.cpp:240-389
    void paraview_vtk_radial(std::string& Name_Bathymetry_File, int Ma, int i_radial, int n, double& u_0, double& t_0, double&,
                             double& r_air, double&, double& co2_0, Array& h, Array& p_dyn, Array& p_stat, Array& BuoyancyForce,
                             Array& t, Array& u, Array& v, Array& w, Array& c, Array& co2, Array& cloud, Array& ice, Array&,
                             Array& aux_v, Array& aux_w, Array& radiation_3D, Array& Q_Latent, Array& Q_Sensible, Array& epsilon_3D,
                             Array& P_rain, Array& P_snow, Array_2D& precipitable_water, Array_2D& Q_bottom, Array_2D& Q_radiation,
                             Array_2D& Q_latent, Array_2D& Q_sensible, Array_2D& Evaporation_Penman, Array_2D& Evaporation_Dalton,
                             Array_2D& Q_Evaporation, Array_2D& temperature_NASA, Array_2D& precipitation_NASA, Array_2D& Vegetation,
                             Array_2D& albedo, Array_2D& epsilon, Array_2D& Precipitation, Array_2D& Topography, Array_2D& temp_NASA) {
        Device& d = Device::of(ATOMDEV_ATM);
        d.grid(im_, jm_, km_);
        atomdev_params& p = d.params();
        p.u_0 = u_0; p.t_0 = t_0; p.r_air = r_air; p.co2_0 = co2_0;
        push_all(d, { { ATOMDEV_H, &h }, { ATOMDEV_P_DYN, &p_dyn }, { ATOMDEV_P_STAT, &p_stat }, { ATOMDEV_BUOYANCY, &BuoyancyForce },
                      { ATOMDEV_T, &t }, { ATOMDEV_U, &u }, { ATOMDEV_V, &v }, { ATOMDEV_W, &w }, { ATOMDEV_C, &c }, { ATOMDEV_CO2, &co2 },
                      { ATOMDEV_CLOUD, &cloud }, { ATOMDEV_ICE, &ice }, { ATOMDEV_AUX_V, &aux_v }, { ATOMDEV_AUX_W, &aux_w },
                      { ATOMDEV_RADIATION_3D, &radiation_3D }, { ATOMDEV_Q_LATENT, &Q_Latent }, { ATOMDEV_Q_SENSIBLE, &Q_Sensible },
                      { ATOMDEV_EPSILON_3D, &epsilon_3D }, { ATOMDEV_P_RAIN, &P_rain }, { ATOMDEV_P_SNOW, &P_snow } });
        d.push2d(ATOMDEV_PRECIPITABLE_WATER, precipitable_water); d.push2d(ATOMDEV_Q_BOTTOM, Q_bottom);
        d.push2d(ATOMDEV_Q_RADIATION, Q_radiation); d.push2d(ATOMDEV_Q_LATENT_2D, Q_latent); d.push2d(ATOMDEV_Q_SENSIBLE_2D, Q_sensible);
        d.push2d(ATOMDEV_EVAPORATION_PENMAN, Evaporation_Penman); d.push2d(ATOMDEV_EVAPORATION_DALTON, Evaporation_Dalton);
        d.push2d(ATOMDEV_Q_EVAPORATION, Q_Evaporation); d.push2d(ATOMDEV_TEMPERATURE_NASA, temperature_NASA);
        d.push2d(ATOMDEV_PRECIPITATION_NASA, precipitation_NASA); d.push2d(ATOMDEV_VEGETATION, Vegetation);
        d.push2d(ATOMDEV_ALBEDO, albedo); d.push2d(ATOMDEV_EPSILON, epsilon); d.push2d(ATOMDEV_PRECIPITATION, Precipitation);
        d.push2d(ATOMDEV_TOPOGRAPHY, Topography); d.push2d(ATOMDEV_TEMP_NASA_CELSIUS, temp_NASA);
        const std::string path = out_ + "/[" + Name_Bathymetry_File + "]_Atm_radial_" + std::to_string(i_radial) + "_" + std::to_string(n) + ".vtk";
        d.run(atomdev_write_vtk(d.handle(), ATOMDEV_RADIAL, i_radial, Ma, path.c_str()), "atomdev_write_vtk");
        pull_all(d, { { ATOMDEV_AUX_V, &aux_v }, { ATOMDEV_AUX_W, &aux_w } });
        d.pull2d(ATOMDEV_TEMP_NASA_CELSIUS, temp_NASA);
    }
    // atmosphere/PostProcess_Atm.cpp:391-520
    void paraview_vtk_zonal(std::string& Name_Bathymetry_File, int k_zonal, int n, double& hp, double& ep, double& R_Air, double& g,
                            double& L_atm, double& u_0, double& t_0, double&, double& r_air, double&, double& co2_0, Array& h,
                            Array& p_dyn, Array& p_stat, Array& BuoyancyForce, Array& t, Array& u, Array& v, Array& w, Array& c,
                            Array& co2, Array& cloud, Array& ice, Array& aux_u, Array& aux_v, Array& aux_w, Array& Q_Latent,
                            Array& Q_Sensible, Array& radiation_3D, Array& epsilon_3D, Array& P_rain, Array& P_snow, Array& S_v,
                            Array& S_c, Array& S_i, Array& S_r, Array& S_s, Array& S_c_c) {
        Device& d = Device::of(ATOMDEV_ATM);
        d.grid(im_, jm_, km_);
        atomdev_params& p = d.params();
        p.hp = hp; p.ep = ep; p.R_Air = R_Air; p.g = g; p.L_atm = L_atm; p.u_0 = u_0; p.t_0 = t_0; p.r_air = r_air; p.co2_0 = co2_0;
        push_all(d, { { ATOMDEV_H, &h }, { ATOMDEV_P_DYN, &p_dyn }, { ATOMDEV_P_STAT, &p_stat }, { ATOMDEV_BUOYANCY, &BuoyancyForce },
                      { ATOMDEV_T, &t }, { ATOMDEV_U, &u }, { ATOMDEV_V, &v }, { ATOMDEV_W, &w }, { ATOMDEV_C, &c }, { ATOMDEV_CO2, &co2 },
                      { ATOMDEV_CLOUD, &cloud }, { ATOMDEV_ICE, &ice }, { ATOMDEV_AUX_U, &aux_u }, { ATOMDEV_AUX_V, &aux_v },
                      { ATOMDEV_AUX_W, &aux_w }, { ATOMDEV_Q_LATENT, &Q_Latent }, { ATOMDEV_Q_SENSIBLE, &Q_Sensible },
                      { ATOMDEV_RADIATION_3D, &radiation_3D }, { ATOMDEV_EPSILON_3D, &epsilon_3D }, { ATOMDEV_P_RAIN, &P_rain },
                      { ATOMDEV_P_SNOW, &P_snow }, { ATOMDEV_S_V, &S_v }, { ATOMDEV_S_C, &S_c }, { ATOMDEV_S_I, &S_i },
                      { ATOMDEV_S_R, &S_r }, { ATOMDEV_S_S, &S_s }, { ATOMDEV_S_C_C, &S_c_c } });
        const std::string path = out_ + "/[" + Name_Bathymetry_File + "]_Atm_zonal_" + std::to_string(k_zonal) + "_" + std::to_string(n) + ".vtk";
        d.run(atomdev_write_vtk(d.handle(), ATOMDEV_ZONAL, k_zonal, 0, path.c_str()), "atomdev_write_vtk");
        pull_all(d, { { ATOMDEV_AUX_U, &aux_u }, { ATOMDEV_AUX_V, &aux_v }, { ATOMDEV_AUX_W, &aux_w } });
    }
    // atmosphere/PostProcess_Atm.cpp:522-621
    void paraview_vtk_longal(std::string& Name_Bathymetry_File, int j_longal, int n, double& u_0, double& t_0, double&, double& r_air,
                             double&, double& co2_0, Array& h, Array& p_dyn, Array& p_stat, Array& BuoyancyForce, Array& t, Array& u,
                             Array& v, Array& w, Array& c, Array& co2, Array& cloud, Array& ice, Array&, Array&, Array& aux_w,
                             Array& Q_Latent, Array& Q_Sensible, Array& epsilon_3D, Array& P_rain, Array& P_snow) {
        Device& d = Device::of(ATOMDEV_ATM);
        d.grid(im_, jm_, km_);
        atomdev_params& p = d.params();
        p.u_0 = u_0; p.t_0 = t_0; p.r_air = r_air; p.co2_0 = co2_0;
        push_all(d, { { ATOMDEV_H, &h }, { ATOMDEV_P_DYN, &p_dyn }, { ATOMDEV_P_STAT, &p_stat }, { ATOMDEV_BUOYANCY, &BuoyancyForce },
                      { ATOMDEV_T, &t }, { ATOMDEV_U, &u }, { ATOMDEV_V, &v }, { ATOMDEV_W, &w }, { ATOMDEV_C, &c }, { ATOMDEV_CO2, &co2 },
                      { ATOMDEV_CLOUD, &cloud }, { ATOMDEV_ICE, &ice }, { ATOMDEV_AUX_W, &aux_w }, { ATOMDEV_Q_LATENT, &Q_Latent },
                      { ATOMDEV_Q_SENSIBLE, &Q_Sensible }, { ATOMDEV_EPSILON_3D, &epsilon_3D }, { ATOMDEV_P_RAIN, &P_rain },
                      { ATOMDEV_P_SNOW, &P_snow } });
        const std::string path = out_ + "/[" + Name_Bathymetry_File + "]_Atm_longal_" + std::to_string(j_longal) + "_" + std::to_string(n) + ".vtk";
        d.run(atomdev_write_vtk(d.handle(), ATOMDEV_LONGAL, j_longal, 0, path.c_str()), "atomdev_write_vtk");
        pull_all(d, { { ATOMDEV_AUX_W, &aux_w } });
    }
    // atmosphere/PostProcess_Atm.cpp:624-652
    void Atmosphere_PlotData(std::string& Name_Bathymetry_File, int iter_cnt, double u_0, double t_0, Array& h, Array& v, Array& w,
                             Array& t, Array& c, Array_2D& Precipitation, Array_2D& precipitable_water) {
        Device& d = Device::of(ATOMDEV_ATM);
        d.grid(im_, jm_, km_);
        d.params().u_0 = u_0; d.params().t_0 = t_0;
        push_all(d, { { ATOMDEV_H, &h }, { ATOMDEV_V, &v }, { ATOMDEV_W, &w }, { ATOMDEV_T, &t }, { ATOMDEV_C, &c } });
        d.push2d(ATOMDEV_PRECIPITATION, Precipitation); d.push2d(ATOMDEV_PRECIPITABLE_WATER, precipitable_water);
        const std::string path = out_ + "/[" + Name_Bathymetry_File + "]_PlotData_Atm" +
                                 (iter_cnt > 0 ? "_" + std::to_string(iter_cnt) : "") + ".xyz";
        d.run(atomdev_atm_write_plot_data(d.handle(), path.c_str()), "atomdev_atm_write_plot_data");
    }

private:
    int im_, jm_, km_;
    std::string out_;
};

// ---------------------------------------------------------------------------------------------
// hydrosphere
// ---------------------------------------------------------------------------------------------

// hydrosphere/RHS_Hyd.h
class RHS_Hydrosphere {
public:
    RHS_Hydrosphere(int jm, int km, double dthe, double dphi, double re) {
        atomdev_params& p = Device::of(ATOMDEV_HYD).params();
        (void)jm; (void)km; p.dthe = dthe; p.dphi = dphi; p.re = re;
    }
    RHS_Hydrosphere(int im, int jm, int km, double dt, double dr, double dthe, double dphi, double re, double sc, double g,
                    double pr, double buoyancy) {
        atomdev_params& p = Device::of(ATOMDEV_HYD).params();
        (void)im; (void)jm; (void)km;
        p.dt = dt; p.dr = dr; p.dthe = dthe; p.dphi = dphi; p.re = re; p.sc = sc; p.g = g; p.pr = pr; p.Buoyancy = buoyancy;
    }
};

// hydrosphere/RungeKutta_Hyd.h
class RungeKutta_Hydrosphere {
public:
    RungeKutta_Hydrosphere(int im, int jm, int km, double dt) : im_(im), jm_(jm), km_(km) {
        Device::of(ATOMDEV_HYD).params().dt = dt;
    }

    // hydrosphere/RungeKutta_Hyd.cpp:30-124
    void solveRungeKutta_3D_Hydrosphere(RHS_Hydrosphere&, int&, double L_hyd, double g, double, double u_0, double t_0,
                                        double c_0, double r_0_water, double ta, double pa, double ca, Array_1D& rad,
                                        Array_1D& the, Array_1D&, Array_2D& Evaporation_Dalton, Array_2D& Precipitation,
                                        Array& h, Array&, Array&, Array&, Array&, Array&, Array& t, Array& u, Array& v,
                                        Array& w, Array& p_dyn, Array& c, Array& tn, Array& un, Array& vn, Array& wn,
                                        Array& p_dynn, Array& cn, Array& aux_u, Array& aux_v, Array& aux_w, Array& Salt_Finger,
                                        Array& Salt_Diffusion, Array& Buoyancy_Force, Array& Salt_Balance, Array&,
                                        Array& r_water, Array& r_salt_water, Array_2D& Bathymetry) {
        Device& d = Device::of(ATOMDEV_HYD);
        d.grid(im_, jm_, km_); d.origin(rad, the);
        atomdev_params& p = d.params();
        p.L_hyd = L_hyd; p.g = g; p.u_0 = u_0; p.t_0 = t_0; p.c_0 = c_0; p.r_0_water = r_0_water; p.ta = ta; p.pa = pa; p.ca = ca;
        push_all(d, { { ATOMDEV_H, &h }, { ATOMDEV_T, &t }, { ATOMDEV_U, &u }, { ATOMDEV_V, &v }, { ATOMDEV_W, &w },
                      { ATOMDEV_P_DYN, &p_dyn }, { ATOMDEV_C, &c }, { ATOMDEV_TN, &tn }, { ATOMDEV_UN, &un }, { ATOMDEV_VN, &vn },
                      { ATOMDEV_WN, &wn }, { ATOMDEV_P_DYNN, &p_dynn }, { ATOMDEV_CN, &cn }, { ATOMDEV_R_WATER, &r_water },
                      { ATOMDEV_R_SALT_WATER, &r_salt_water } });
        d.push2d(ATOMDEV_TOPOGRAPHY, Bathymetry);
        d.push2d(ATOMDEV_EVAPORATION_DALTON, Evaporation_Dalton);
        d.push2d(ATOMDEV_PRECIPITATION, Precipitation);
        d.run(atomdev_hyd_rk3d(d.handle()), "atomdev_hyd_rk3d");
        pull_all(d, { { ATOMDEV_T, &t }, { ATOMDEV_U, &u }, { ATOMDEV_V, &v }, { ATOMDEV_W, &w }, { ATOMDEV_C, &c },
                      { ATOMDEV_AUX_U, &aux_u }, { ATOMDEV_AUX_V, &aux_v }, { ATOMDEV_AUX_W, &aux_w },
                      { ATOMDEV_SALT_FINGER, &Salt_Finger }, { ATOMDEV_SALT_DIFFUSION, &Salt_Diffusion },
                      { ATOMDEV_BUOYANCY, &Buoyancy_Force }, { ATOMDEV_SALT_BALANCE, &Salt_Balance },
                      { ATOMDEV_R_WATER, &r_water }, { ATOMDEV_R_SALT_WATER, &r_salt_water } });
    }

    // hydrosphere/RungeKutta_Hyd.cpp:130-186
    void solveRungeKutta_2D_Hydrosphere(RHS_Hydrosphere&, int&, double r_0_water, Array_1D& rad, Array_1D& the, Array_1D&,
                                        Array&, Array&, Array& h, Array& v, Array& w, Array& p_dyn, Array& vn, Array& wn,
                                        Array& p_dynn, Array& aux_v, Array& aux_w) {
        Device& d = Device::of(ATOMDEV_HYD);
        d.grid(im_, jm_, km_); d.origin(rad, the);
        d.params().r_0_water = r_0_water;
        push_all(d, { { ATOMDEV_H, &h }, { ATOMDEV_V, &v }, { ATOMDEV_W, &w }, { ATOMDEV_P_DYN, &p_dyn }, { ATOMDEV_VN, &vn },
                      { ATOMDEV_WN, &wn }, { ATOMDEV_P_DYNN, &p_dynn }, { ATOMDEV_AUX_V, &aux_v }, { ATOMDEV_AUX_W, &aux_w } });
        d.run(atomdev_hyd_rk2d(d.handle()), "atomdev_hyd_rk2d");
        pull_all(d, { { ATOMDEV_V, &v }, { ATOMDEV_W, &w }, { ATOMDEV_AUX_V, &aux_v }, { ATOMDEV_AUX_W, &aux_w } });
    }

private:
    int im_, jm_, km_;
};

// hydrosphere/Pressure_Hyd.h
class Pressure_Hyd {
public:
    Pressure_Hyd(int im, int jm, int km, double dr, double dthe, double dphi) : im_(im), jm_(jm), km_(km) {
        atomdev_params& p = Device::of(ATOMDEV_HYD).params();
        p.dr = dr; p.dthe = dthe; p.dphi = dphi;
    }
    // hydrosphere/Pressure_Hyd.cpp:42-200
    void computePressure_3D(double u_0, double r_0_water, Array_1D& rad, Array_1D& the, Array& p_dyn, Array& p_dynn, Array& h,
                            Array& aux_u, Array& aux_v, Array& aux_w) {
        Device& d = begin(u_0, r_0_water, rad, the);
        push_all(d, { { ATOMDEV_P_DYN, &p_dyn }, { ATOMDEV_P_DYNN, &p_dynn }, { ATOMDEV_H, &h }, { ATOMDEV_AUX_U, &aux_u },
                      { ATOMDEV_AUX_V, &aux_v }, { ATOMDEV_AUX_W, &aux_w } });
        d.run(atomdev_hyd_pressure3d(d.handle()), "atomdev_hyd_pressure3d");
        pull_all(d, { { ATOMDEV_P_DYN, &p_dyn }, { ATOMDEV_AUX_U, &aux_u }, { ATOMDEV_AUX_V, &aux_v }, { ATOMDEV_AUX_W, &aux_w } });
    }
    // hydrosphere/Pressure_Hyd.cpp:206-317
    void computePressure_2D(double u_0, double r_0_water, Array_1D& rad, Array_1D& the, Array& p_dyn, Array& p_dynn, Array& h,
                            Array& aux_v, Array& aux_w) {
        Device& d = begin(u_0, r_0_water, rad, the);
        push_all(d, { { ATOMDEV_P_DYN, &p_dyn }, { ATOMDEV_P_DYNN, &p_dynn }, { ATOMDEV_H, &h }, { ATOMDEV_AUX_V, &aux_v },
                      { ATOMDEV_AUX_W, &aux_w } });
        d.run(atomdev_hyd_pressure2d(d.handle()), "atomdev_hyd_pressure2d");
        pull_all(d, { { ATOMDEV_P_DYN, &p_dyn }, { ATOMDEV_AUX_V, &aux_v } });
    }

private:
    Device& begin(double u_0, double r_0_water, Array_1D& rad, Array_1D& the) {
        Device& d = Device::of(ATOMDEV_HYD);
        d.grid(im_, jm_, km_); d.origin(rad, the);
        d.params().u_0 = u_0; d.params().r_0_water = r_0_water;
        return d;
    }
    int im_, jm_, km_;
};

// hydrosphere/BC_Hyd.h
class BC_Hydrosphere {
public:
    BC_Hydrosphere(int im, int jm, int km) : im_(im), jm_(jm), km_(km) {}
    // hydrosphere/BC_Hyd.cpp:38-83
    void RB_radius(double ca, double ta, double pa, double dr, Array_1D& rad, Array& t, Array& u, Array& v, Array& w,
                   Array& p_dyn, Array& c) {
        Device& d = Device::of(ATOMDEV_HYD);
        d.params().ca = ca; d.params().ta = ta; d.params().pa = pa; d.params().dr = dr; d.params().r0 = rad.z[0];
        apply(atomdev_hyd_bc_radius, "atomdev_hyd_bc_radius", t, u, v, w, p_dyn, c);
    }
    // hydrosphere/BC_Hyd.cpp:90-118
    void RB_theta(double ca, double ta, double pa, Array& t, Array& u, Array& v, Array& w, Array& p_dyn, Array& c) {
        Device& d = Device::of(ATOMDEV_HYD);
        d.params().ca = ca; d.params().ta = ta; d.params().pa = pa;
        apply(atomdev_hyd_bc_theta, "atomdev_hyd_bc_theta", t, u, v, w, p_dyn, c);
    }
    // hydrosphere/BC_Hyd.cpp:125-158
    void RB_phi(Array& t, Array& u, Array& v, Array& w, Array& p_dyn, Array& c) {
        apply(atomdev_hyd_bc_phi, "atomdev_hyd_bc_phi", t, u, v, w, p_dyn, c);
    }

private:
    void apply(int (*entry)(atomdev_t*), const char* name, Array& t, Array& u, Array& v, Array& w, Array& p_dyn, Array& c) {
        Device& d = Device::of(ATOMDEV_HYD);
        d.grid(im_, jm_, km_);
        push_all(d, { { ATOMDEV_T, &t }, { ATOMDEV_U, &u }, { ATOMDEV_V, &v }, { ATOMDEV_W, &w }, { ATOMDEV_P_DYN, &p_dyn },
                      { ATOMDEV_C, &c } });
        d.run(entry(d.handle()), name);
        pull_all(d, { { ATOMDEV_T, &t }, { ATOMDEV_U, &u }, { ATOMDEV_V, &v }, { ATOMDEV_W, &w }, { ATOMDEV_P_DYN, &p_dyn },
                      { ATOMDEV_C, &c } });
    }
    int im_, jm_, km_;
};

// hydrosphere/BC_Thermohalin.h: the two members the 3-D loop calls every iteration
// (hydrosphere/cHydrosphereModel.cpp:448-451); the initial-condition builders stay on the host.
class BC_Thermohalin {
public:
    BC_Thermohalin(int im, int jm, int km, int i_beg, int i_max, int Ma, int Ma_max, int Ma_max_half, double dr, double g,
                   double r_0_water, double, double, double, double ta, double ca, double pa, double u_0, double p_0, double t_0,
                   double c_0, double, double L_hyd, double t_average, double t_cretaceous_max, double, double, const std::string&)
        : im_(im), jm_(jm), km_(km) {
        atomdev_params& p = Device::of(ATOMDEV_HYD).params();
        p.dr = dr; p.g = g; p.r_0_water = r_0_water; p.ta = ta; p.ca = ca; p.pa = pa; p.u_0 = u_0; p.p_0 = p_0; p.t_0 = t_0;
        p.c_0 = c_0; p.L_hyd = L_hyd;
        start_.i_beg = i_beg; start_.i_max = i_max; start_.Ma = Ma; start_.Ma_max = Ma_max; start_.Ma_max_half = Ma_max_half;
        start_.t_average = t_average; start_.t_cretaceous_max = t_cretaceous_max;
    }
    // hydrosphere/BC_Thermohalin.cpp:88-360 (rad and the are arguments the reference never reads)
    void IC_v_w_EkmanSpiral(Array_1D&, Array_1D&, Array& h, Array& v, Array& w) {
        Device& d = Device::of(ATOMDEV_HYD);
        d.grid(im_, jm_, km_);
        push_all(d, { { ATOMDEV_H, &h }, { ATOMDEV_V, &v }, { ATOMDEV_W, &w } });
        d.run(atomdev_hyd_ic_ekman_spiral(d.handle(), start_.i_beg), "atomdev_hyd_ic_ekman_spiral");
        pull_all(d, { { ATOMDEV_V, &v }, { ATOMDEV_W, &w } });
    }
    // hydrosphere/BC_Thermohalin.cpp:676-690 (only w changes)
    void IC_CircumPolar_Current(Array& h, Array&, Array&, Array& w, Array&) {
        Device& d = Device::of(ATOMDEV_HYD);
        d.grid(im_, jm_, km_);
        push_all(d, { { ATOMDEV_H, &h }, { ATOMDEV_W, &w } });
        d.run(atomdev_hyd_ic_circumpolar_current(d.handle(), start_.i_beg), "atomdev_hyd_ic_circumpolar_current");
        pull_all(d, { { ATOMDEV_W, &w } });
    }
    // hydrosphere/BC_Thermohalin.cpp:367-513 (p_dyn is not touched; the banner is not printed)
    void BC_Temperature_Salinity(Array& h, Array& t, Array& c, Array&) {
        Device& d = Device::of(ATOMDEV_HYD);
        d.grid(im_, jm_, km_);
        push_all(d, { { ATOMDEV_H, &h }, { ATOMDEV_T, &t }, { ATOMDEV_C, &c } });
        d.run(atomdev_hyd_bc_temperature_salinity(d.handle(), &start_), "atomdev_hyd_bc_temperature_salinity");
        pull_all(d, { { ATOMDEV_T, &t }, { ATOMDEV_C, &c } });
    }
    // hydrosphere/BC_Thermohalin.cpp:636-670
    void BC_Surface_Salinity_NASA(const std::string& Name_SurfaceSalinity_File, Array& c) {
        Device& d = Device::of(ATOMDEV_HYD);
        d.grid(im_, jm_, km_);
        push_all(d, { { ATOMDEV_C, &c } });
        d.run(atomdev_hyd_read_surface_salinity(d.handle(), Name_SurfaceSalinity_File.c_str()), "atomdev_hyd_read_surface_salinity");
        pull_all(d, { { ATOMDEV_C, &c } });
    }
    // hydrosphere/BC_Thermohalin.cpp:518-583 (h is an argument the reference never reads)
    void BC_Pressure_Density(Array& p_stat, Array& r_water, Array& r_salt_water, Array& t, Array& c, Array&) {
        Device& d = Device::of(ATOMDEV_HYD);
        d.grid(im_, jm_, km_);
        push_all(d, { { ATOMDEV_T, &t }, { ATOMDEV_C, &c } });
        d.run(atomdev_hyd_pressure_density(d.handle()), "atomdev_hyd_pressure_density");
        pull_all(d, { { ATOMDEV_P_STAT, &p_stat }, { ATOMDEV_R_WATER, &r_water }, { ATOMDEV_R_SALT_WATER, &r_salt_water } });
    }
    // hydrosphere/BC_Thermohalin.cpp:699-722 (h and p_dyn are not touched)
    void Value_Limitation_Hyd(Array&, Array& u, Array& v, Array& w, Array&, Array& t, Array& c) {
        Device& d = Device::of(ATOMDEV_HYD);
        d.grid(im_, jm_, km_);
        push_all(d, { { ATOMDEV_U, &u }, { ATOMDEV_V, &v }, { ATOMDEV_W, &w }, { ATOMDEV_T, &t }, { ATOMDEV_C, &c } });
        d.run(atomdev_hyd_value_limitation(d.handle()), "atomdev_hyd_value_limitation");
        pull_all(d, { { ATOMDEV_U, &u }, { ATOMDEV_V, &v }, { ATOMDEV_W, &w }, { ATOMDEV_T, &t }, { ATOMDEV_C, &c } });
    }

private:
    int im_, jm_, km_;
    atomdev_hyd_start start_;
};

// hydrosphere/BC_Bath_Hyd.h: the bathymetry reader.
class BC_Bathymetry_Hydrosphere {
public:
    BC_Bathymetry_Hydrosphere(int im, int jm, int km) : im_(im), jm_(jm), km_(km) {}
    // hydrosphere/BC_Bath_Hyd.cpp:31-114
    void BC_SeaGround(const std::string& bathymetry_file, double L_hyd, Array& h, Array_2D& Bathymetry) {
        Device& d = Device::of(ATOMDEV_HYD);
        d.grid(im_, jm_, km_);
        d.params().L_hyd = L_hyd;
        d.run(atomdev_hyd_read_sea_ground(d.handle(), bathymetry_file.c_str()), "atomdev_hyd_read_sea_ground");
        d.pull(ATOMDEV_H, h);
        d.pull2d(ATOMDEV_TOPOGRAPHY, Bathymetry);
    }

private:
    int im_, jm_, km_;
};

// hydrosphere/PostProcess_Hyd.h: the reading half of the coupling.
class PostProcess_Hydrosphere {
public:
    PostProcess_Hydrosphere(int im, int jm, int km, const std::string& output_path) : im_(im), jm_(jm), km_(km), out_(output_path) {}
    // hydrosphere/PostProcess_Hyd.cpp:642-668: only level im-1 of v, w, t, p and the two 2-D fields change
    void Atmosphere_TransferFile_read(const std::string& Name_Bathymetry_File, Array& v, Array& w, Array& t, Array& p,
                                      Array_2D& Evaporation_Dalton, Array_2D& Precipitation) {
        Device& d = Device::of(ATOMDEV_HYD);
        d.grid(im_, jm_, km_);
        push_all(d, { { ATOMDEV_V, &v }, { ATOMDEV_W, &w }, { ATOMDEV_T, &t }, { ATOMDEV_P_DYN, &p } });
        const std::string path = out_ + "/[" + Name_Bathymetry_File + "]_Transfer_Atm.vw";
        d.run(atomdev_hyd_read_transfer_file(d.handle(), path.c_str()), "atomdev_hyd_read_transfer_file");
        pull_all(d, { { ATOMDEV_V, &v }, { ATOMDEV_W, &w }, { ATOMDEV_T, &t }, { ATOMDEV_P_DYN, &p } });
        d.pull2d(ATOMDEV_EVAPORATION_DALTON, Evaporation_Dalton); d.pull2d(ATOMDEV_PRECIPITATION, Precipitation);
    }

    // hydrosphere/PostProcess_Hyd.cpp:269-366
    void paraview_panorama_vts(const std::string& Name_Bathymetry_File, int n, double& u_0, double& r_0_water, Array& h, Array& t,
                               Array& p_dyn, Array& p_stat, Array& r_water, Array& r_salt_water, Array& u, Array& v, Array& w, Array& c,
                               Array&, Array&, Array&, Array& Salt_Finger, Array& Salt_Diffusion, Array& Buoyancy_Force,
                               Array& Salt_Balance) {
        Device& d = Device::of(ATOMDEV_HYD);
        d.grid(im_, jm_, km_);
        d.params().u_0 = u_0; d.params().r_0_water = r_0_water;
        push_all(d, { { ATOMDEV_H, &h }, { ATOMDEV_T, &t }, { ATOMDEV_P_DYN, &p_dyn }, { ATOMDEV_P_STAT, &p_stat },
                      { ATOMDEV_R_WATER, &r_water }, { ATOMDEV_R_SALT_WATER, &r_salt_water }, { ATOMDEV_U, &u }, { ATOMDEV_V, &v },
                      { ATOMDEV_W, &w }, { ATOMDEV_C, &c }, { ATOMDEV_SALT_FINGER, &Salt_Finger },
                      { ATOMDEV_SALT_DIFFUSION, &Salt_Diffusion }, { ATOMDEV_BUOYANCY, &Buoyancy_Force },
                      { ATOMDEV_SALT_BALANCE, &Salt_Balance } });
        const std::string path = out_ + "/[" + Name_Bathymetry_File + "]_Hyd_panorama_" + std::to_string(n) + ".vts";
        d.run(atomdev_write_panorama_vts(d.handle(), path.c_str()), "atomdev_write_panorama_vts");
    }
    // hydrosphere/PostProcess_Hyd.cpp:373-449, 559-636: the two sections that need no surface fields
    void paraview_vtk_longal(const std::string& Name_Bathymetry_File, int j_longal, int n, double& u_0, double& r_0_water, Array& h,
                             Array& p_dyn, Array& p_stat, Array& r_water, Array& r_salt_water, Array& t, Array& u, Array& v, Array& w,
                             Array& c, Array&, Array&, Array& Salt_Finger, Array& Salt_Diffusion, Array& Buoyancy_Force,
                             Array& Salt_Balance) {
        section(ATOMDEV_LONGAL, j_longal, "_Hyd_longal_", Name_Bathymetry_File, n, u_0, r_0_water, h, p_dyn, p_stat, r_water,
                r_salt_water, t, u, v, w, c, Salt_Finger, Salt_Diffusion, Buoyancy_Force, Salt_Balance);
    }
    void paraview_vtk_zonal(const std::string& Name_Bathymetry_File, int k_zonal, int n, double& u_0, double& r_0_water, Array& h,
                            Array& p_dyn, Array& p_stat, Array& r_water, Array& r_salt_water, Array& t, Array& u, Array& v, Array& w,
                            Array& c, Array& Salt_Finger, Array& Salt_Diffusion, Array& Buoyancy_Force, Array& Salt_Balance) {
        section(ATOMDEV_ZONAL, k_zonal, "_Hyd_zonal_", Name_Bathymetry_File, n, u_0, r_0_water, h, p_dyn, p_stat, r_water,
                r_salt_water, t, u, v, w, c, Salt_Finger, Salt_Diffusion, Buoyancy_Force, Salt_Balance);
    }
    // hydrosphere/PostProcess_Hyd.cpp:456-552
    void paraview_vtk_radial(const std::string& Name_Bathymetry_File, int i_radial, int n, double& u_0, double& t_0, double& r_0_water,
                             Array& h, Array& p_dyn, Array& p_stat, Array& r_water, Array& r_salt_water, Array& t, Array& u, Array& v,
                             Array& w, Array& c, Array&, Array& aux_v, Array& Salt_Finger, Array& Salt_Diffusion, Array& Buoyancy_Force,
                             Array& Salt_Balance, Array_2D& Upwelling, Array_2D& Downwelling, Array_2D&, Array_2D&, Array_2D&,
                             Array_2D& BottomWater, Array_2D& Evaporation_Dalton, Array_2D& Precipitation, Array_2D& Bathymetry) {
        Device& d = Device::of(ATOMDEV_HYD);
        d.grid(im_, jm_, km_);
        d.params().t_0 = t_0;
        d.push(ATOMDEV_AUX_V, aux_v);
        d.push2d(ATOMDEV_UPWELLING, Upwelling); d.push2d(ATOMDEV_DOWNWELLING, Downwelling); d.push2d(ATOMDEV_BOTTOM_WATER, BottomWater);
        d.push2d(ATOMDEV_EVAPORATION_DALTON, Evaporation_Dalton); d.push2d(ATOMDEV_PRECIPITATION, Precipitation);
        d.push2d(ATOMDEV_TOPOGRAPHY, Bathymetry);
        section(ATOMDEV_RADIAL, i_radial, "_Hyd_radial_", Name_Bathymetry_File, n, u_0, r_0_water, h, p_dyn, p_stat, r_water,
                r_salt_water, t, u, v, w, c, Salt_Finger, Salt_Diffusion, Buoyancy_Force, Salt_Balance);
        d.pull(ATOMDEV_AUX_V, aux_v);
    }
    // hydrosphere/PostProcess_Hyd.cpp:676-705
    void Hydrosphere_PlotData(const std::string& Name_Bathymetry_File, int iter_cnt, double& u_0, Array& h, Array& v, Array& w,
                              Array& t, Array& c, Array_2D& BottomWater, Array_2D& Upwelling, Array_2D& Downwelling) {
        Device& d = Device::of(ATOMDEV_HYD);
        d.grid(im_, jm_, km_);
        d.params().u_0 = u_0;
        push_all(d, { { ATOMDEV_H, &h }, { ATOMDEV_V, &v }, { ATOMDEV_W, &w }, { ATOMDEV_T, &t }, { ATOMDEV_C, &c } });
        d.push2d(ATOMDEV_BOTTOM_WATER, BottomWater); d.push2d(ATOMDEV_UPWELLING, Upwelling); d.push2d(ATOMDEV_DOWNWELLING, Downwelling);
        const std::string path = out_ + "/[" + Name_Bathymetry_File + "]_PlotData_Hyd" +
                                 (iter_cnt > 0 ? "_" + std::to_string(iter_cnt) : "") + ".xyz";
        d.run(atomdev_hyd_write_plot_data(d.handle(), path.c_str()), "atomdev_hyd_write_plot_data");
    }

private:
    void section(int kind, int index, const char* tag, const std::string& name, int n, double u_0, double r_0_water, Array& h,
                 Array& p_dyn, Array& p_stat, Array& r_water, Array& r_salt_water, Array& t, Array& u, Array& v, Array& w, Array& c,
                 Array& Salt_Finger, Array& Salt_Diffusion, Array& Buoyancy_Force, Array& Salt_Balance) {
        Device& d = Device::of(ATOMDEV_HYD);
        d.grid(im_, jm_, km_);
        d.params().u_0 = u_0; d.params().r_0_water = r_0_water;
        push_all(d, { { ATOMDEV_H, &h }, { ATOMDEV_P_DYN, &p_dyn }, { ATOMDEV_P_STAT, &p_stat }, { ATOMDEV_R_WATER, &r_water },
                      { ATOMDEV_R_SALT_WATER, &r_salt_water }, { ATOMDEV_T, &t }, { ATOMDEV_U, &u }, { ATOMDEV_V, &v }, { ATOMDEV_W, &w },
                      { ATOMDEV_C, &c }, { ATOMDEV_SALT_FINGER, &Salt_Finger }, { ATOMDEV_SALT_DIFFUSION, &Salt_Diffusion },
                      { ATOMDEV_BUOYANCY, &Buoyancy_Force }, { ATOMDEV_SALT_BALANCE, &Salt_Balance } });
        const std::string path = out_ + "/[" + name + "]" + tag + std::to_string(index) + "_" + std::to_string(n) + ".vtk";
        d.run(atomdev_write_vtk(d.handle(), kind, index, 0, path.c_str()), "atomdev_write_vtk");
    }
    int im_, jm_, km_;
    std::string out_;
};

// hydrosphere/Results_Hyd.h: run_data without its printout (Results_Hyd.cpp:322-395 prints six values at one fixed
// 1-degree location; nothing of that is fed back into the model)
class Results_Hyd {
public:
    Results_Hyd(int im, int jm, int km) : im_(im), jm_(jm), km_(km) {}
    // hydrosphere/Results_Hyd.cpp:102-320
    void run_data(int, double dr, double dthe, double L_hyd, double u_0, double c_0, Array_1D& rad, Array_1D& the, Array& h,
                  Array&, Array& v, Array& w, Array& c, Array& Salt_Balance, Array& Salt_Finger, Array& Salt_Diffusion,
                  Array& Buoyancy_Force_3D, Array_2D& Upwelling, Array_2D& Downwelling, Array_2D& SaltFinger,
                  Array_2D& SaltDiffusion, Array_2D& BuoyancyForce_2D, Array_2D& Salt_total, Array_2D& BottomWater) {
        Device& d = Device::of(ATOMDEV_HYD);
        d.grid(im_, jm_, km_);
        d.origin(rad, the);
        atomdev_params& p = d.params();
        p.dr = dr; p.dthe = dthe; p.L_hyd = L_hyd; p.u_0 = u_0; p.c_0 = c_0;
        push_all(d, { { ATOMDEV_H, &h }, { ATOMDEV_V, &v }, { ATOMDEV_W, &w }, { ATOMDEV_C, &c },
                      { ATOMDEV_SALT_BALANCE, &Salt_Balance }, { ATOMDEV_SALT_FINGER, &Salt_Finger },
                      { ATOMDEV_SALT_DIFFUSION, &Salt_Diffusion }, { ATOMDEV_BUOYANCY, &Buoyancy_Force_3D } });
        d.run(atomdev_hyd_run_data(d.handle()), "atomdev_hyd_run_data");
        pull_all(d, { { ATOMDEV_SALT_BALANCE, &Salt_Balance }, { ATOMDEV_SALT_FINGER, &Salt_Finger },
                      { ATOMDEV_SALT_DIFFUSION, &Salt_Diffusion }, { ATOMDEV_BUOYANCY, &Buoyancy_Force_3D } });
        d.pull2d(ATOMDEV_UPWELLING, Upwelling); d.pull2d(ATOMDEV_DOWNWELLING, Downwelling); d.pull2d(ATOMDEV_SALT_FINGER_2D, SaltFinger);
        d.pull2d(ATOMDEV_SALT_DIFFUSION_2D, SaltDiffusion); d.pull2d(ATOMDEV_BUOYANCY_FORCE_2D, BuoyancyForce_2D);
        d.pull2d(ATOMDEV_SALT_TOTAL, Salt_total); d.pull2d(ATOMDEV_BOTTOM_WATER, BottomWater);
    }

private:
    int im_, jm_, km_;
};

}  // namespace atomdev_mirror

#endif  // ATOMDEV_MIRROR_HPP_
