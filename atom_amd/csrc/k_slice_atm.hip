// cAtmosphereModel::run_2D_loop and run_3D_loop (atmosphere/cAtmosphereModel.cpp:660-735, 743-898) as one call each:
// the operators of the loops in the reference's order, every one of them an entry point of this library, fields
// resident throughout.  What the reference prints on the way is returned (the last residuum queries) or left out.
#include <vector>

#include "atomdev_internal.h"

extern "C" void atomdev_default_atm_loop(atomdev_atm_loop* L) {
    L->velocity_iter_max_2D = 2; L->pressure_iter_max_2D = 10;  // param.py:33-36
    L->velocity_iter_max = 2; L->pressure_iter_max = 2;
    L->column_physics = 0; L->RadiationModel = 1;
    atomdev_default_radiation(&L->radiation);
    L->i_trop = nullptr;
    L->queries = 1;
}

#define STEP(call) do { if (int rc__ = (call)) return rc__; } while (0)

// The surface loops of n independent time slices in lockstep: each operator enqueued for all slices on their own
// streams, the surface sweeps - one workgroup's worth of dependent diagonals each - in one launch.
extern "C" int atomdev_atm_run_2d_loops(atomdev_t* const* handles, int n, const atomdev_atm_loop* L) {
    ATOMDEV_REQUIRE(handles && n >= 1 && L, "atomdev_atm_run_2d_loops: bad argument");
    for (int m = 0; m < n; m++)
        ATOMDEV_REQUIRE(handles[m] && handles[m]->model == ATOMDEV_ATM, "atomdev_atm_run_2d_loops: needs atmosphere handles");
#define EACH(call) do { for (int m = 0; m < n; m++) { atomdev_t* h = handles[m]; if (int rc__ = (call)) return rc__; } } while (0)
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
    }
#undef EACH
    return 0;
}

extern "C" int atomdev_atm_run_2d_loop(atomdev_t* h, const atomdev_atm_loop* L) {
    ATOMDEV_REQUIRE(h && L, "atomdev_atm_run_2d_loop: needs an atmosphere handle and the loop description");
    return atomdev_atm_run_2d_loops(&h, 1, L);
}

extern "C" int atomdev_atm_run_3d_loop(atomdev_t* h, const atomdev_atm_loop* L, atomdev_atm_loop_state* S) {
    ATOMDEV_REQUIRE(h && L && S && h->model == ATOMDEV_ATM, "atomdev_atm_run_3d_loop: needs an atmosphere handle, the loop description and the carried state");
    ATOMDEV_REQUIRE(!L->column_physics || L->i_trop, "atomdev_atm_run_3d_loop: the column physics need the tropopause levels");
    double res = 0., st[9]; int loc[3] = { 0, 0, 0 }, st_loc[27];
    STEP(atomdev_move_data_to_new_arrays(h, -1));  // :748
    for (int p = 1; p <= L->pressure_iter_max; p++) {
        for (int v = 1; v <= L->velocity_iter_max; v++) {
            if (L->queries) { res = 0.; loc[0] = loc[1] = loc[2] = 0; STEP(atomdev_atm_residuum3d(h, &res, loc)); S->residuum[0] = res; }
            STEP(atomdev_atm_bc_radius(h));
            STEP(atomdev_atm_bc_theta(h));
            STEP(atomdev_atm_bc_phi(h));
            if (L->column_physics && v % 2 == 0) STEP(atomdev_atm_ice_water_saturation_adjustment(h, &S->q_v_hyp));  // :787-789
            STEP(atomdev_atm_value_limitation(h));
            STEP(atomdev_atm_solid_ground(h));
            STEP(atomdev_atm_rk3d(h));
            STEP(atomdev_atm_value_limitation(h));
            if (L->column_physics) STEP(atomdev_atm_radiation_multi_layer(h, &L->radiation, L->i_trop));  // :828-831
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
        }
        STEP(atomdev_atm_pressure3d(h));
    }
    return atomdev_sync(h);
}
