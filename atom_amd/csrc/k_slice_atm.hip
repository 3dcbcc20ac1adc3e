// cAtmosphereModel::run_2D_loop and run_3D_loop (atmosphere/cAtmosphereModel.cpp:660-735, 743-898) as one call each:
// the operators of the loops in the reference's order, every one of them an entry point of this library, fields
// resident throughout.  What the reference prints on the way is returned (the last residuum queries) or left out.
#include <cmath>
#include <string>
#include <vector>

#include "atomdev_internal.h"

extern "C" void atomdev_default_atm_loop(atomdev_atm_loop* L) {
    L->velocity_iter_max_2D = 2; L->pressure_iter_max_2D = 10;  // param.py:33-36
    L->velocity_iter_max = 2; L->pressure_iter_max = 2;
    L->column_physics = 0; L->RadiationModel = 1;
    atomdev_default_radiation(&L->radiation);
    L->i_trop = nullptr;
    L->queries = 1;
    L->nm = 200;  // cAtmosphereModel.h:123
}

#include "step_clock.h"
#define STEP(call) do { StepClock clock__(h, #call); if (int rc__ = (call)) return rc__; } while (0)

// The surface loops of n independent time slices in lockstep: each operator enqueued for all slices on their own
// streams, the surface sweeps - a few waves' worth of dependent diagonals each - in one launch.
extern "C" int atomdev_atm_run_2d_loops(atomdev_t* const* handles, int n, const atomdev_atm_loop* L) {
    ATOMDEV_REQUIRE(handles && n >= 1 && L, "atomdev_atm_run_2d_loops: bad argument");
    for (int m = 0; m < n; m++)
        ATOMDEV_REQUIRE(handles[m] && handles[m]->model == ATOMDEV_ATM, "atomdev_atm_run_2d_loops: needs atmosphere handles");
    atomdev_t* const h = handles[0];  // the handle whose stream the step times are taken on
#define EACH(call) do { StepClock clock__(h, #call); for (int m = 0; m < n; m++) { atomdev_t* h = handles[m]; if (int rc__ = (call)) return rc__; } } while (0)
    std::vector<double> res(n, 0.);
    std::vector<int> loc(3 * (size_t)n, 0);
    double st[3]; int st_loc[9];
    for (int p = 1; p <= L->pressure_iter_max_2D; p++) {
        for (int v = 1; v <= L->velocity_iter_max_2D; v++) {
            EACH(atomdev_atm_bc_theta(h));
            EACH(atomdev_atm_bc_phi(h));
            if (L->queries) EACH((res[m] = 0., loc[3 * m] = loc[3 * m + 1] = loc[3 * m + 2] = 0, atomdev_atm_residuum2d(h, &res[m], &loc[3 * m])));
            EACH(atomdev_atm_value_limitation(h));
            EACH(atomdev_atm_solid_ground(h));
            if (n == 1) STEP(atomdev_atm_rk2d(handles[0])); else STEP(atomdev_rk2d_batch(handles, n));
            if (L->queries) {  // the same Accuracy_Atm object as before the sweep: its state is carried (:688-712)
                EACH(atomdev_atm_residuum2d(h, &res[m], &loc[3 * m]));
                EACH(atomdev_atm_steady2d(h, st, st_loc));
            }
            EACH(atomdev_move_data_to_new_arrays(h, 0));
        }
        EACH(atomdev_atm_pressure2d(h));
        if (p * L->velocity_iter_max_2D + 1 > L->nm) break;  // iter_cnt > nm (:731-735)
    }
#undef EACH
    return 0;
}

extern "C" int atomdev_atm_run_2d_loop(atomdev_t* h, const atomdev_atm_loop* L) {
    ATOMDEV_REQUIRE(h && L, "atomdev_atm_run_2d_loop: needs an atmosphere handle and the loop description");
    return atomdev_atm_run_2d_loops(&h, 1, L);
}

namespace {

// The output files of cAtmosphereModel::write_file (:599-652) for the slice `s`; n: velocity iterations done so far.
int write_files(atomdev_t* h, const atomdev_atm_slice* s, int n, bool is_final_result) {
    if (!s || !s->output_path) return 0;
    const std::string stem = std::string(s->output_path) + "/[" + s->bathymetry_name + "]";
    const std::string count = std::to_string(n);
    const int i_radial = 0, j_longal = 62, k_zonal = 87;  // Mount Everest / Himalaya
    STEP(atomdev_write_vtk(h, ATOMDEV_RADIAL, i_radial, s->start.Ma, (stem + "_Atm_radial_" + std::to_string(i_radial) + "_" + count + ".vtk").c_str()));
    STEP(atomdev_write_vtk(h, ATOMDEV_LONGAL, j_longal, s->start.Ma, (stem + "_Atm_longal_" + std::to_string(j_longal) + "_" + count + ".vtk").c_str()));
    STEP(atomdev_write_vtk(h, ATOMDEV_ZONAL, k_zonal, s->start.Ma, (stem + "_Atm_zonal_" + std::to_string(k_zonal) + "_" + count + ".vtk").c_str()));
    STEP(atomdev_atm_write_transfer_file(h, (stem + "_Transfer_Atm.vw").c_str()));
    const int iter_cnt = is_final_result ? -1 : n;
    return atomdev_atm_write_plot_data(h, (stem + "_PlotData_Atm" + (iter_cnt > 0 ? "_" + count : "") + ".xyz").c_str());
}

int run_3d_loop(atomdev_t* h, const atomdev_atm_loop* L, atomdev_atm_loop_state* S, const atomdev_atm_slice* slice, int* iterations);

}  // namespace

extern "C" int atomdev_atm_run_3d_loop(atomdev_t* h, const atomdev_atm_loop* L, atomdev_atm_loop_state* S) {
    int iterations = 0;
    return run_3d_loop(h, L, S, nullptr, &iterations);
}

namespace {

int run_3d_loop(atomdev_t* h, const atomdev_atm_loop* L, atomdev_atm_loop_state* S, const atomdev_atm_slice* slice, int* iterations) {
    ATOMDEV_REQUIRE(h && L && S && h->model == ATOMDEV_ATM, "atomdev_atm_run_3d_loop: needs an atmosphere handle, the loop description and the carried state");
    ATOMDEV_REQUIRE(!L->column_physics || L->i_trop, "atomdev_atm_run_3d_loop: the column physics need the tropopause levels");
    double res = 0., st[9]; int loc[3] = { 0, 0, 0 }, st_loc[27];
    int iter_cnt = 1;
    STEP(atomdev_move_data_to_new_arrays(h, -1));  // :748
    for (int p = 1; p <= L->pressure_iter_max; p++) {
        for (int v = 1; v <= L->velocity_iter_max; v++) {
            if (L->queries) { res = 0.; loc[0] = loc[1] = loc[2] = 0; STEP(atomdev_atm_residuum3d(h, &res, loc)); S->residuum[0] = res; }
            STEP(atomdev_atm_bc_radius_theta_phi(h));  // :782-784 in one launch
            if (L->column_physics && v % 2 == 0) STEP(atomdev_atm_ice_water_saturation_adjustment(h, &S->q_v_hyp));  // :787-789
            STEP(atomdev_atm_limit_ground_rk3d_limit(h));  // :791, :793, :809 and the limiter of :825 in one
            if (L->column_physics && L->RadiationModel == 1) STEP(atomdev_atm_radiation_multi_layer(h, &L->radiation, L->i_trop));  // :828-831
            if (L->queries) {  // the same Accuracy_Atm object as before the sweep: its state is carried (:776-777, 833)
                STEP(atomdev_atm_residuum3d(h, &res, loc));
                S->residuum[1] = res; S->residuum_loc[0] = loc[0]; S->residuum_loc[1] = loc[1]; S->residuum_loc[2] = loc[2];
                STEP(atomdev_atm_steady3d(h, st, st_loc));
            }
            if (L->column_physics) {  // :844-852
                STEP(atomdev_atm_latent_heat(h));
                double extremes[2]; int where[6];
                STEP(atomdev_minmax2d(h, ATOMDEV_PRECIPITATION, extremes, where));
                STEP(atomdev_atm_vegetation_distribution(h, extremes[0]));
            }
            STEP(atomdev_atm_msl_data(h, L->RadiationModel));  // :854-859
            if (L->column_physics && v % 2 == 0) {  // :861-866
                double members[2] = { S->S_nuc, S->S_i_dep };
                STEP(atomdev_atm_two_category_ice_scheme(h, members));
                S->S_nuc = members[0]; S->S_i_dep = members[1];
            }
            STEP(atomdev_move_data_to_new_arrays(h, -1));
            iter_cnt++;
        }
        STEP(atomdev_atm_pressure3d(h));
        if (slice && slice->checkpoint > 0 && p % slice->checkpoint == 0) STEP(write_files(h, slice, iter_cnt - 1, false));  // :886-888
        if (iter_cnt > L->nm) break;  // :891-895
    }
    *iterations = iter_cnt - 1;
    return atomdev_sync(h);
}

}  // namespace

extern "C" void atomdev_default_atm_slice(atomdev_atm_slice* s) {
    atomdev_default_atm_start(&s->start);
    atomdev_default_atm_loop(&s->loop);
    s->loop.column_physics = 1;
    s->tropopause_equator = 30; s->coast_correction = 0; s->checkpoint = 2;  // param.py:58, 37
    s->CO2 = 1.; s->co2_factor = .98;
    s->ua = s->va = s->wa = 0.; s->coa = 1.;
    s->mean_temperature = 0.;
    s->bathymetry_file = s->temperature_file = s->precipitation_file = s->output_path = s->bathymetry_name = nullptr;
}

extern "C" int atomdev_atm_run_time_slice(atomdev_t* h, const atomdev_atm_slice* s, atomdev_atm_loop_state* S) {
    ATOMDEV_REQUIRE(h && s && S && h->model == ATOMDEV_ATM, "atomdev_atm_run_time_slice: needs an atmosphere handle, the slice and the carried state");
    ATOMDEV_REQUIRE(s->bathymetry_file && s->temperature_file && s->precipitation_file, "atomdev_atm_run_time_slice: needs the three input files");
    ATOMDEV_REQUIRE(!s->output_path || s->bathymetry_name, "atomdev_atm_run_time_slice: output files need bathymetry_name");
    const atomdev_params& P = h->p;
    // the text of the output files is made by threads of their own while the slice goes on; all are waited for before
    // the call returns, whichever way it returns
    struct Files {
        atomdev_t* h;
        explicit Files(atomdev_t* h_) : h(h_) { h->defer_files = true; }
        ~Files() { h->defer_files = false; (void)atomdev_join_files(h); }
    } files(h);
    // reset_arrays (:384-472)
    for (int g = 0; g < ATOMDEV_N2D; g++) STEP(atomdev_fill2d(h, g, 0.));
    for (int f = 0; f < ATOMDEV_N3D; f++)
        if (h->f[f] && f != ATOMDEV_RESIDUUM) STEP(atomdev_fill(h, f, 0.));
    const struct { int field; double value; } init[] = {
        { ATOMDEV_T, P.ta }, { ATOMDEV_U, s->ua }, { ATOMDEV_V, s->va }, { ATOMDEV_W, s->wa }, { ATOMDEV_C, P.ca }, { ATOMDEV_CO2, s->coa },
        { ATOMDEV_TN, P.ta }, { ATOMDEV_UN, s->ua }, { ATOMDEV_VN, s->va }, { ATOMDEV_WN, s->wa }, { ATOMDEV_CN, P.ca }, { ATOMDEV_CO2N, s->coa },
        { ATOMDEV_P_DYN, P.pa }, { ATOMDEV_P_DYNN, P.pa }, { ATOMDEV_P_STAT, P.pa } };
    for (const auto& e : init)
        if (e.value != 0.) STEP(atomdev_fill(h, e.field, e.value));
    S->q_v_hyp = S->S_nuc = S->S_i_dep = 0.;
    // :220-302
    STEP(atomdev_atm_read_mountain_surface(h, s->bathymetry_file));
    atomdev_atm_start start = s->start;
    std::vector<int> levels(h->jm, s->tropopause_equator);
    if (start.im_tropopause) levels.assign(start.im_tropopause, start.im_tropopause + h->jm);
    start.im_tropopause = levels.data();
    STEP(atomdev_atm_ic_cell_structure(h, levels.data()));
    STEP(atomdev_atm_read_surface_temperature(h, s->temperature_file));
    STEP(atomdev_atm_read_surface_precipitation(h, s->precipitation_file));
    STEP(atomdev_atm_bc_temperature(h, &start));
    if (s->coast_correction) STEP(atomdev_atm_ic_temperature_west_east_coast(h));
    STEP(atomdev_atm_bc_pressure(h));
    STEP(atomdev_atm_bc_water_vapour(h, &start));
    STEP(atomdev_atm_bc_co2(h, &start));
    atomdev_atm_loop L = s->loop;
    L.column_physics = 1;
    L.RadiationModel = start.RadiationModel;
    L.radiation.co2_coeff = atomdev_atm_co2_coeff(s->CO2, s->co2_factor, start.co2_equator, start.co2_tropopause, P.co2_0);
    const int add = (int)std::round(((start.t_cretaceous / P.t_0) * P.t_0) / 2.6);  // BC_Thermo.cpp:226, 3298-3302
    std::vector<int> shifted(levels);
    for (int& l : shifted) l += add;
    L.i_trop = shifted.data();
    if (L.RadiationModel == 1) STEP(atomdev_atm_radiation_multi_layer(h, &L.radiation, L.i_trop));
    STEP(atomdev_move_data_to_new_arrays(h, -1));
    STEP(atomdev_move_data_to_new_arrays(h, 0));
    STEP(atomdev_atm_run_2d_loop(h, &L));
    int iterations = 0;
    STEP(run_3d_loop(h, &L, S, s, &iterations));
    STEP(atomdev_atm_restrain_temperature(h, s->mean_temperature));
    STEP(write_files(h, s, iterations, true));
    h->defer_files = false;
    STEP(atomdev_join_files(h));
    return atomdev_sync(h);
}
