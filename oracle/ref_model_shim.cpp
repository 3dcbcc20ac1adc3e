/*
 * ref_model_shim.cpp - flat-array C entry points around the reference's BC_Thermo and model classes.
 *
 * TEST INFRASTRUCTURE ONLY (see ref_shim.cpp): our own code, linked against atmosphere/BC_Thermo.cpp,
 * atmosphere/cAtmosphereModel.cpp, hydrosphere/cHydrosphereModel.cpp, lib/Config.cpp, tinyxml2 and
 * cli/DefaultStream.cpp compiled where they lie, plus the pieces the reference's own param.py generates
 * (oracle/gen_ref_headers.py).  BC_Thermo takes its constants from a cAtmosphereModel object and from the
 * globals of namespace AtmParameters, which only cAtmosphereModel::LoadConfig fills
 * (AtmosphereLoadConfig.cpp.inc): one model object is created here, given a config file without
 * parameters (every element keeps its default) through LoadConfig, and then receives the scalars of
 * the caller's atomdev_params in both places.
 *
 * The reference keeps these classes' state private; this translation unit reads and writes it (the
 * members BC_Thermo carries between calls, the model's time list and tropopause table) by including
 * the class definitions with `private` spelled `public`.  Access specifiers do not change the layout
 * the compiler gives these classes, and the reference's own objects are compiled untouched.
 */
#include <algorithm>
#include <cmath>
#include <csignal>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <limits>
#include <map>
#include <set>
#include <sstream>
#include <string>
#include <tuple>
#include <vector>
#include <unistd.h>

#define private public
#include "cAtmosphereModel.h"
#include "BC_Thermo.h"
#undef private
#include "cHydrosphereModel.h"
#include "AtmParameters.h"

#include "ref_shim.h"

#ifndef ATOM_REF_DIR
#define ATOM_REF_DIR "/root/reference"
#endif

namespace {

cAtmosphereModel* g_model = nullptr;

// Sets a parameter where the reference keeps it twice: the model's member and the AtmParameters global.
#define SETP(name, value) do { g_model->name = (value); AtmParameters::name = (value); } while (0)

cAtmosphereModel* model() {
    if (g_model) return g_model;
    CoutSilencer quiet;
    void (*old_int)(int) = std::signal(SIGINT, SIG_DFL);   // the constructor installs exit() on SIGINT
    g_model = new cAtmosphereModel();
    std::signal(SIGINT, old_int);
    char path[] = "/tmp/atomref_cfg_XXXXXX";
    int fd = mkstemp(path);
    const char* xml = "<atom><common></common><atmosphere></atmosphere></atom>";
    if (fd < 0 || write(fd, xml, std::strlen(xml)) < 0) std::abort();
    close(fd);
    g_model->LoadConfig(path);
    unlink(path);
    // the constructor looked for the temperature curve relative to the working directory (param.py:24)
    g_model->temperature_curve_file = std::string(ATOM_REF_DIR) + "/data/Lenton_etal_COPSE_time_temp.txt";
    AtmParameters::temperature_curve_file = g_model->temperature_curve_file;
    g_model->m_temperature_curve.clear();
    g_model->load_temperature_curve();
    return g_model;
}

void set_scalars(const atomdev_params& p) {
    model();
    SETP(L_atm, p.L_atm); SETP(g, p.g); SETP(ep, p.ep); SETP(hp, p.hp); SETP(u_0, p.u_0); SETP(p_0, p.p_0);
    SETP(t_0, p.t_0); SETP(c_0, p.c_0); SETP(co2_0, p.co2_0); SETP(cp_l, p.cp_l); SETP(r_air, p.r_air);
    SETP(R_Air, p.R_Air); SETP(r_water_vapour, p.r_water_vapour); SETP(R_WaterVapour, p.R_WaterVapour);
    SETP(lv, p.lv); SETP(ls, p.ls); SETP(gam, p.gam); SETP(t_tropopause, p.t_tropopause);
    SETP(re, p.re); SETP(pr, p.pr); SETP(sc_WaterVapour, p.sc_WaterVapour); SETP(sc_CO2, p.sc_CO2);
    SETP(Buoyancy, p.Buoyancy);
}

void set_time(int Ma) {
    cAtmosphereModel* m = model();
    m->m_time_list.clear();
    m->m_current_time = m->m_time_list.insert(float(Ma)).first;
}

// The model's members and time list as a time slice described by atomdev_atm_start would have them.
void set_start(const atomdev_atm_start* s) {
    cAtmosphereModel* m = model();
    SETP(sun, s->sun); SETP(RadiationModel, s->RadiationModel); SETP(NASATemperature, s->NASATemperature);
    SETP(declination, s->declination); SETP(sun_position_lat, s->sun_position_lat); SETP(sun_position_lon, s->sun_position_lon);
    SETP(t_average, s->t_average); SETP(t_equator, s->t_equator); SETP(t_pole, s->t_pole); SETP(t_land, s->t_land);
    SETP(c_tropopause, s->c_tropopause); SETP(c_ocean, s->c_ocean); SETP(c_land, s->c_land);
    SETP(co2_average, s->co2_average); SETP(co2_equator, s->co2_equator); SETP(co2_tropopause, s->co2_tropopause);
    SETP(co2_pole, s->co2_pole); SETP(co2_cretaceous, s->co2_cretaceous); SETP(co2_vegetation, s->co2_vegetation);
    SETP(co2_ocean, s->co2_ocean); SETP(co2_land, s->co2_land);
    m->m_time_list.clear();
    if (!s->first_time_slice) m->m_time_list.insert(float(s->Ma_previous));
    m->m_current_time = m->m_time_list.insert(float(s->Ma)).first;
}

// A BC_Thermo object on the context's grid.  The constructor copies dt, dr, dthe, dphi from the model's
// compile-time constants of the 1-degree grid (BC_Thermo.cpp:41-44) and derives dt_dim and dr_dim from
// them (:121-123); on another grid they are replaced by the context's step sizes, as a model built for
// that grid would have passed them.
struct Thermo {
    RefCtx* c;
    std::vector<int> trop;
    int* old_trop;
    BC_Thermo* circ;
    Thermo(RefCtx* c_, int Ma = 0, const int* im_tropopause = nullptr, const atomdev_atm_start* start = nullptr)
        : c(c_), trop(c_->jm, 0) {
        set_scalars(c->p);
        if (start) { set_start(start); im_tropopause = start->im_tropopause; }
        else set_time(Ma);
        cAtmosphereModel* m = model();
        if (im_tropopause) std::copy(im_tropopause, im_tropopause + c->jm, trop.begin());
        old_trop = m->im_tropopause;
        m->im_tropopause = trop.data();
        copy_in(c);
        circ = new BC_Thermo(m, c->im, c->jm, c->km, *c->a[ATOMDEV_H]);
        circ->dt = c->p.dt; circ->dr = c->p.dr; circ->dthe = c->p.dthe; circ->dphi = c->p.dphi;
        circ->dt_dim = circ->L_atm / circ->u_0 * circ->dt;
        circ->dr_dim = circ->dr * circ->L_atm;
    }
    ~Thermo() {
        delete circ;
        model()->im_tropopause = old_trop;
        copy_out(c);
    }
};

}  // namespace

extern "C" {

/* BC_Thermo::Value_Limitation_Atm (atmosphere/BC_Thermo.cpp:3235-3275) */
void ref_atm_value_limitation(void* v) {
    RefCtx* c = (RefCtx*)v;
    CoutSilencer quiet;
    Thermo th(c);
    th.circ->Value_Limitation_Atm(A(H), A(U), A(V), A(W), A(P_DYN), A(T), A(C), A(CLOUD), A(ICE), A(CO2));
}

/* BC_Thermo::Latent_Heat (atmosphere/BC_Thermo.cpp:2418-2502), called as cAtmosphereModel.cpp:844-845 does */
void ref_atm_latent_heat(void* v) {
    RefCtx* c = (RefCtx*)v;
    CoutSilencer quiet;
    Thermo th(c);
    th.circ->Latent_Heat(c->rad, c->the, c->phi, A(H), A(T), A(TN), A(U), A(V), A(W), A(P_DYN), A(P_STAT), A(C), A(ICE),
                         A(Q_LATENT), A(Q_SENSIBLE), A(RADIATION_3D), B(Q_RADIATION), B(Q_LATENT_2D), B(Q_SENSIBLE_2D),
                         B(Q_BOTTOM));
}

/* BC_Thermo::Two_Category_Ice_Scheme (atmosphere/BC_Thermo.cpp:2803-3143); state = the members S_nuc, S_i_dep */
void ref_atm_two_category_ice_scheme(void* v, double state[2]) {
    RefCtx* c = (RefCtx*)v;
    CoutSilencer quiet;
    Thermo th(c);
    th.circ->S_nuc = state[0]; th.circ->S_i_dep = state[1];
    th.circ->Two_Category_Ice_Scheme(A(H), A(C), A(T), A(P_STAT), A(CLOUD), A(ICE), A(P_RAIN), A(P_SNOW), A(S_V), A(S_C),
                                     A(S_I), A(S_R), A(S_S), A(S_C_C));
    state[0] = th.circ->S_nuc; state[1] = th.circ->S_i_dep;
}

/* BC_Thermo::Ice_Water_Saturation_Adjustment (atmosphere/BC_Thermo.cpp:2509-2796); *q_v_hyp = the member of that name */
void ref_atm_ice_water_saturation_adjustment(void* v, double* q_v_hyp) {
    RefCtx* c = (RefCtx*)v;
    CoutSilencer quiet;
    Thermo th(c);
    th.circ->q_v_hyp = *q_v_hyp;
    th.circ->Ice_Water_Saturation_Adjustment(A(H), A(C), A(CN), A(CLOUD), A(CLOUDN), A(ICE), A(ICEN), A(T), A(P_STAT),
                                             A(S_C_C));
    *q_v_hyp = th.circ->q_v_hyp;
}

/* BC_Thermo::BC_Radiation_multi_layer (atmosphere/BC_Thermo.cpp:166-424) at Ma = 0, where the tropopause level of a
 * row is im_tropopause[j] itself (:226 adds GetTropopauseHightAdd(0) = 0).  Returns non-zero if co2_coeff is neither
 * of the two values the function can derive (:251-256). */
int ref_atm_radiation_multi_layer(void* v, const atomdev_radiation* R, const int* i_trop) {
    RefCtx* c = (RefCtx*)v;
    CoutSilencer quiet;
    cAtmosphereModel* m = model();
    SETP(albedo_pole, R->albedo_pole); SETP(albedo_equator, R->albedo_equator);
    SETP(epsilon_pole, R->epsilon_pole); SETP(epsilon_equator, R->epsilon_equator);
    SETP(epsilon_tropopause, R->epsilon_tropopause); SETP(rad_pole, R->rad_pole); SETP(rad_equator, R->rad_equator);
    SETP(sigma, R->sigma);
    if (R->co2_coeff == 1.) SETP(CO2, 0.);
    else if (R->co2_coeff == m->co2_factor * (m->co2_equator / m->co2_tropopause)) SETP(CO2, 1.);
    else return 1;
    Thermo th(c, 0, i_trop);
    th.circ->BC_Radiation_multi_layer(B(ALBEDO), B(EPSILON), B(RADIATION_SURFACE), A(P_STAT), A(T), A(C), A(H),
                                      A(EPSILON_3D), A(RADIATION_3D), A(CLOUD), A(ICE), A(CO2));
    return 0;
}

/* ---- the atmosphere's start of a time slice (cAtmosphereModel.cpp:260-292), member by member ---- */

/* cAtmosphereModel::get_mean_temperature_from_curve (cAtmosphereModel.cpp:924-946), in float like the reference */
float ref_atm_mean_temperature_from_curve(float time) { CoutSilencer quiet; return model()->get_mean_temperature_from_curve(time); }

int ref_atm_ic_cell_structure(void* v, const int* im_tropopause) {
    RefCtx* c = (RefCtx*)v;
    CoutSilencer quiet;
    Thermo th(c, 0, im_tropopause);
    th.circ->IC_CellStructure(A(H), A(U), A(V), A(W));
    return 0;
}
int ref_atm_read_surface_temperature(void* v, const char* path) {
    RefCtx* c = (RefCtx*)v;
    CoutSilencer quiet;
    Thermo th(c);
    th.circ->BC_Surface_Temperature_NASA(path, B(TEMPERATURE_NASA), A(T));
    return 0;
}
int ref_atm_read_surface_precipitation(void* v, const char* path) {
    RefCtx* c = (RefCtx*)v;
    CoutSilencer quiet;
    Thermo th(c);
    th.circ->BC_Surface_Precipitation_NASA(path, B(PRECIPITATION_NASA));
    return 0;
}
void ref_atm_bc_temperature(void* v, const atomdev_atm_start* s) {
    RefCtx* c = (RefCtx*)v;
    CoutSilencer quiet;
    Thermo th(c, 0, nullptr, s);
    th.circ->BC_Temperature(B(TEMPERATURE_NASA), A(H), A(T), A(TN), A(P_DYN), A(P_STAT));
}
void ref_atm_ic_temperature_west_east_coast(void* v) {
    RefCtx* c = (RefCtx*)v;
    CoutSilencer quiet;
    Thermo th(c);
    th.circ->IC_Temperature_WestEastCoast(A(H), A(T));
}
void ref_atm_bc_pressure(void* v) {
    RefCtx* c = (RefCtx*)v;
    CoutSilencer quiet;
    Thermo th(c);
    th.circ->BC_Pressure(A(P_STAT), A(P_DYN), A(T), A(H));
}
void ref_atm_bc_water_vapour(void* v, const atomdev_atm_start* s) {
    RefCtx* c = (RefCtx*)v;
    CoutSilencer quiet;
    Thermo th(c, 0, nullptr, s);
    th.circ->BC_WaterVapour(A(H), A(P_STAT), A(T), A(C), A(V), A(W));
}
void ref_atm_bc_co2(void* v, const atomdev_atm_start* s) {
    RefCtx* c = (RefCtx*)v;
    CoutSilencer quiet;
    Thermo th(c, 0, nullptr, s);
    th.circ->BC_CO2(B(VEGETATION), A(H), A(T), A(P_DYN), A(CO2));
}

/* ---- cAtmosphereModel::RunTimeSlice (cAtmosphereModel.cpp:140-341) as a whole ---- */

struct ref_atm_slice_files {
    const char* bathymetry_path;     /* directory of the <Ma>Ma_Golonka.xyz files */
    const char* temperature_file;
    const char* precipitation_file;
    const char* output_path;
};

/* Runs the model object's own RunTimeSlice on its own arrays (the context must have the model's compile-time grid,
 * 41 x 181 x 361) and copies every bound field out afterwards.  The plate reconstruction is switched off (its script
 * needs pygplates): Ma > 0 reads the same surface files and runs IC_Temperature_WestEastCoast.  Returns 1 on a grid
 * mismatch. */
int ref_atm_run_time_slice(void* v, const atomdev_atm_start* s, const atomdev_atm_loop* L, const ref_atm_slice_files* files) {
    RefCtx* c = (RefCtx*)v;
    if (c->im != 41 || c->jm != 181 || c->km != 361) return 1;
    CoutSilencer quiet;
    cAtmosphereModel* m = model();
    set_scalars(c->p);
    set_start(s);
    const atomdev_radiation* R = &L->radiation;
    SETP(albedo_pole, R->albedo_pole); SETP(albedo_equator, R->albedo_equator);
    SETP(epsilon_pole, R->epsilon_pole); SETP(epsilon_equator, R->epsilon_equator);
    SETP(epsilon_tropopause, R->epsilon_tropopause); SETP(rad_pole, R->rad_pole); SETP(rad_equator, R->rad_equator);
    SETP(sigma, R->sigma);
    SETP(CO2, 1.);
    SETP(velocity_iter_max_2D, L->velocity_iter_max_2D); SETP(pressure_iter_max_2D, L->pressure_iter_max_2D);
    SETP(velocity_iter_max, L->velocity_iter_max); SETP(pressure_iter_max, L->pressure_iter_max);
    SETP(use_earthbyte_reconstruction, false);
    SETP(verbose, false); SETP(paraview_panorama_vts, false); SETP(debug, false);
    SETP(bathymetry_path, std::string(files->bathymetry_path));
    SETP(temperature_file, std::string(files->temperature_file));
    SETP(precipitation_file, std::string(files->precipitation_file));
    SETP(output_path, std::string(files->output_path));
    // the time list as the slices before this one left it; RunTimeSlice inserts Ma itself (:149)
    m->m_time_list.clear();
    if (!s->first_time_slice) m->m_time_list.insert(float(s->Ma_previous));
    m->m_current_time = m->m_time_list.begin();
    m->RunTimeSlice(s->Ma);
    struct { int id; Array* a; } f3[] = {
        { ATOMDEV_T, &m->t }, { ATOMDEV_U, &m->u }, { ATOMDEV_V, &m->v }, { ATOMDEV_W, &m->w }, { ATOMDEV_P_DYN, &m->p_dyn },
        { ATOMDEV_C, &m->c }, { ATOMDEV_CLOUD, &m->cloud }, { ATOMDEV_ICE, &m->ice }, { ATOMDEV_CO2, &m->co2 },
        { ATOMDEV_TN, &m->tn }, { ATOMDEV_UN, &m->un }, { ATOMDEV_VN, &m->vn }, { ATOMDEV_WN, &m->wn },
        { ATOMDEV_P_DYNN, &m->p_dynn }, { ATOMDEV_CN, &m->cn }, { ATOMDEV_CLOUDN, &m->cloudn }, { ATOMDEV_ICEN, &m->icen },
        { ATOMDEV_CO2N, &m->co2n }, { ATOMDEV_H, &m->h }, { ATOMDEV_AUX_U, &m->aux_u }, { ATOMDEV_AUX_V, &m->aux_v },
        { ATOMDEV_AUX_W, &m->aux_w }, { ATOMDEV_BUOYANCY, &m->BuoyancyForce }, { ATOMDEV_S_V, &m->S_v }, { ATOMDEV_S_C, &m->S_c },
        { ATOMDEV_S_I, &m->S_i }, { ATOMDEV_S_R, &m->S_r }, { ATOMDEV_S_S, &m->S_s }, { ATOMDEV_P_STAT, &m->p_stat },
        { ATOMDEV_Q_LATENT, &m->Q_Latent }, { ATOMDEV_Q_SENSIBLE, &m->Q_Sensible }, { ATOMDEV_RADIATION_3D, &m->radiation_3D },
        { ATOMDEV_P_RAIN, &m->P_rain }, { ATOMDEV_P_SNOW, &m->P_snow }, { ATOMDEV_EPSILON_3D, &m->epsilon_3D },
        { ATOMDEV_S_C_C, &m->S_c_c } };
    for (auto& e : f3)
        if (c->f[e.id])
            for (int i = 0; i < c->im; i++)
                for (int j = 0; j < c->jm; j++)
                    std::memcpy(c->f[e.id] + ((size_t)i * c->jm + j) * c->km, e.a->x[i][j], sizeof(double) * c->km);
    struct { int id; Array_2D* a; } f2[] = {
        { ATOMDEV_TOPOGRAPHY, &m->Topography }, { ATOMDEV_EVAPORATION_DALTON, &m->Evaporation_Dalton },
        { ATOMDEV_PRECIPITATION, &m->Precipitation }, { ATOMDEV_Q_RADIATION, &m->Q_radiation },
        { ATOMDEV_PRECIPITABLE_WATER, &m->precipitable_water }, { ATOMDEV_EVAPORATION_PENMAN, &m->Evaporation_Penman },
        { ATOMDEV_Q_EVAPORATION, &m->Q_Evaporation }, { ATOMDEV_Q_LATENT_2D, &m->Q_latent }, { ATOMDEV_Q_SENSIBLE_2D, &m->Q_sensible },
        { ATOMDEV_Q_BOTTOM, &m->Q_bottom }, { ATOMDEV_CO2_TOTAL, &m->co2_total }, { ATOMDEV_VEGETATION, &m->Vegetation },
        { ATOMDEV_ALBEDO, &m->albedo }, { ATOMDEV_EPSILON, &m->epsilon }, { ATOMDEV_RADIATION_SURFACE, &m->radiation_surface },
        { ATOMDEV_TEMPERATURE_NASA, &m->temperature_NASA }, { ATOMDEV_PRECIPITATION_NASA, &m->precipitation_NASA },
        { ATOMDEV_TEMP_NASA_CELSIUS, &m->temp_NASA } };
    for (auto& e : f2)
        if (c->g[e.id])
            for (int j = 0; j < c->jm; j++) std::memcpy(c->g[e.id] + (size_t)j * c->km, e.a->y[j], sizeof(double) * c->km);
    return 0;
}

/* cHydrosphereModel::RunTimeSlice (hydrosphere/cHydrosphereModel.cpp:129-632) of a model object of its own with the
 * reference's default parameters: reads output_path/[<Ma>Ma_Golonka.xyz]_Transfer_Atm.vw, the bathymetry under
 * bathymetry_path and the surface salinity file, and writes its output files to output_path (at the checkpoints of its
 * loop and at the end).  The plate reconstruction is switched off. */
int ref_hyd_run_time_slice(const char* bathymetry_path, const char* salinity_file, const char* output_path, int Ma) {
    CoutSilencer quiet;
    static cHydrosphereModel* hm = nullptr;
    if (!hm) {
        void (*old_int)(int) = std::signal(SIGINT, SIG_DFL);   // the constructor installs exit() on SIGINT
        hm = new cHydrosphereModel();
        std::signal(SIGINT, old_int);
    }
    hm->bathymetry_path = bathymetry_path;
    hm->BathymetrySuffix = "Ma_Golonka.xyz";
    hm->salinity_file = salinity_file;
    hm->output_path = output_path;
    hm->use_earthbyte_reconstruction = false;
    hm->verbose = false; hm->debug = false; hm->paraview_panorama_vts = false;
    hm->RunTimeSlice(Ma);
    return 0;
}

}  // extern "C"
