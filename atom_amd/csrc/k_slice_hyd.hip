// cHydrosphereModel::RunTimeSlice (hydrosphere/cHydrosphereModel.cpp:234-610) as one call: the operators of the time
// slice in the reference's order, every one of them an entry point of this library, fields resident throughout.
// What the reference prints on the way (the residuum and steady-state queries, the min/max searches) is returned or
// left to the caller; nothing here computes on the host.
#include "atomdev_internal.h"

extern "C" void atomdev_default_hyd_slice(int im, atomdev_hyd_slice* s) {
    atomdev_default_hyd_start(im, &s->start);
    s->velocity_iter_max_2D = 2; s->pressure_iter_max_2D = 10;  // param.py:135-138
    s->velocity_iter_max = 2; s->pressure_iter_max = 2;
    s->transfer_file = nullptr; s->bathymetry_file = nullptr; s->salinity_file = nullptr;
    s->queries = 1;
}

extern "C" int atomdev_hyd_run_time_slice(atomdev_t* h, const atomdev_hyd_slice* s, double residuum[2], int residuum_loc[3]) {
    ATOMDEV_REQUIRE(h && s && h->model == ATOMDEV_HYD, "atomdev_hyd_run_time_slice: needs a hydrosphere handle and the slice description");
    ATOMDEV_REQUIRE(s->bathymetry_file, "atomdev_hyd_run_time_slice: no bathymetry file");
    const int top = h->im - 1;
#define STEP(call) do { if (int rc__ = (call)) return rc__; } while (0)
    // :234-297: the atmosphere's surface, the sea ground, the initial conditions, the first old -> new copies
    if (s->transfer_file) STEP(atomdev_hyd_read_transfer_file(h, s->transfer_file));
    STEP(atomdev_hyd_read_sea_ground(h, s->bathymetry_file));
    if (s->start.Ma == 0 && s->salinity_file) STEP(atomdev_hyd_read_surface_salinity(h, s->salinity_file));
    STEP(atomdev_hyd_ic_ekman_spiral(h, s->start.i_beg));
    if (s->start.Ma <= 41) STEP(atomdev_hyd_ic_circumpolar_current(h, s->start.i_beg));  // Drake passage
    STEP(atomdev_hyd_bc_temperature_salinity(h, &s->start));
    STEP(atomdev_hyd_pressure_density(h));
    STEP(atomdev_move_data_to_new_arrays(h, -1));
    STEP(atomdev_move_data_to_new_arrays(h, top));
    // :314-404: surface iterations (their two queries are printed only)
    for (int p = 1; p <= s->pressure_iter_max_2D; p++) {
        for (int v = 1; v <= s->velocity_iter_max_2D; v++) {
            STEP(atomdev_hyd_bc_theta(h));
            STEP(atomdev_hyd_bc_phi(h));
            STEP(atomdev_hyd_value_limitation(h));
            STEP(atomdev_hyd_rk2d(h));
            STEP(atomdev_move_data_to_new_arrays(h, top));
        }
        STEP(atomdev_hyd_pressure2d(h));
    }
    // :420-610: volume iterations
    double before = 0., after = 0.;
    int loc[3] = { 0, 0, 0 };
    double steady[6]; int steady_loc[18];
    for (int p = 1; p <= s->pressure_iter_max; p++) {
        for (int v = 1; v <= s->velocity_iter_max; v++) {
            if (s->queries) { before = 0.; STEP(atomdev_hyd_residuum3d(h, &before, loc)); }
            STEP(atomdev_hyd_bc_radius(h));
            STEP(atomdev_hyd_bc_theta(h));
            STEP(atomdev_hyd_bc_phi(h));
            STEP(atomdev_hyd_pressure_density(h));
            STEP(atomdev_hyd_value_limitation(h));
            STEP(atomdev_hyd_rk3d(h));
            STEP(atomdev_hyd_solid_ground(h));
            if (s->queries) {
                after = 0.; loc[0] = loc[1] = loc[2] = 0;
                STEP(atomdev_hyd_residuum3d(h, &after, loc));
                STEP(atomdev_hyd_steady3d(h, steady, steady_loc));
            }
            STEP(atomdev_hyd_run_data(h));
            STEP(atomdev_move_data_to_new_arrays(h, -1));
        }
        STEP(atomdev_hyd_pressure3d(h));
    }
#undef STEP
    if (residuum) { residuum[0] = before; residuum[1] = after; }
    if (residuum_loc) { residuum_loc[0] = loc[0]; residuum_loc[1] = loc[1]; residuum_loc[2] = loc[2]; }
    return atomdev_sync(h);
}
