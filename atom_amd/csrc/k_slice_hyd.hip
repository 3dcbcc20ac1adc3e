// cHydrosphereModel::RunTimeSlice (hydrosphere/cHydrosphereModel.cpp:234-610) as one call: the operators of the time
// slice in the reference's order, every one of them an entry point of this library, fields resident throughout.
// What the reference prints on the way (the residuum and steady-state queries, the min/max searches) is returned or
// left to the caller; nothing here computes on the host.
#include <string>
#include <thread>
#include <vector>

#include "atomdev_internal.h"
#include "step_clock.h"

extern "C" void atomdev_default_hyd_slice(int im, atomdev_hyd_slice* s) {
    atomdev_default_hyd_start(im, &s->start);
    s->velocity_iter_max_2D = 2; s->pressure_iter_max_2D = 10;  // param.py:135-138
    s->velocity_iter_max = 2; s->pressure_iter_max = 2;
    s->transfer_file = nullptr; s->bathymetry_file = nullptr; s->salinity_file = nullptr;
    s->queries = 1;
    s->checkpoint = 1;  // param.py:139
    s->paraview_panorama_vts = 0;
    s->output_path = s->bathymetry_name = nullptr;
}

namespace {
// The output files of cHydrosphereModel::write_file (:739-772) for the slice `s`; n: velocity iterations done so far.
int write_files(atomdev_t* h, const atomdev_hyd_slice& s, int n, bool is_final_result) {
    if (!s.output_path) return 0;
    const std::string stem = std::string(s.output_path) + "/[" + s.bathymetry_name + "]";
    const std::string count = std::to_string(n);
    const int j_longal = 75, k_zonal = 185, i_radial = 40;
    if (int rc = atomdev_write_vtk(h, ATOMDEV_LONGAL, j_longal, s.start.Ma, (stem + "_Hyd_longal_" + std::to_string(j_longal) + "_" + count + ".vtk").c_str())) return rc;
    if (int rc = atomdev_write_vtk(h, ATOMDEV_ZONAL, k_zonal, s.start.Ma, (stem + "_Hyd_zonal_" + std::to_string(k_zonal) + "_" + count + ".vtk").c_str())) return rc;
    if (int rc = atomdev_write_vtk(h, ATOMDEV_RADIAL, i_radial, s.start.Ma, (stem + "_Hyd_radial_" + std::to_string(i_radial) + "_" + count + ".vtk").c_str())) return rc;
    if (s.paraview_panorama_vts)
        if (int rc = atomdev_write_panorama_vts(h, (stem + "_Hyd_panorama_" + count + ".vts").c_str())) return rc;
    const int iter_cnt = is_final_result ? -1 : n;  // PostProcess_Hyd.cpp:676-677
    return atomdev_hyd_write_plot_data(h, (stem + "_PlotData_Hyd" + (iter_cnt > 0 ? "_" + count : "") + ".xyz").c_str());
}
}  // namespace

// n slices advanced in lockstep: every operator is enqueued for all of them (each on its handle's stream, so the
// slices overlap on the GPU) before the next one; the surface sweeps - one workgroup's worth of dependent diagonals
// each - go into one launch for all slices.
extern "C" int atomdev_hyd_run_time_slices(atomdev_t* const* handles, const atomdev_hyd_slice* slices, int n, double* residuum,
                                           int* residuum_loc) {
    ATOMDEV_REQUIRE(handles && slices && n >= 1, "atomdev_hyd_run_time_slices: bad argument");
    const atomdev_hyd_slice& s0 = slices[0];
    for (int m = 0; m < n; m++) {
        ATOMDEV_REQUIRE(handles[m] && handles[m]->model == ATOMDEV_HYD, "atomdev_hyd_run_time_slices: needs hydrosphere handles");
        ATOMDEV_REQUIRE(slices[m].bathymetry_file, "atomdev_hyd_run_time_slices: no bathymetry file");
        ATOMDEV_REQUIRE(!slices[m].output_path || slices[m].bathymetry_name, "atomdev_hyd_run_time_slices: output files need bathymetry_name");
        ATOMDEV_REQUIRE(slices[m].velocity_iter_max_2D == s0.velocity_iter_max_2D && slices[m].pressure_iter_max_2D == s0.pressure_iter_max_2D
                        && slices[m].velocity_iter_max == s0.velocity_iter_max && slices[m].pressure_iter_max == s0.pressure_iter_max,
                        "atomdev_hyd_run_time_slices: the slices of one call share their iteration counts");
    }
#define EACH(call) do { StepClock clock__(handles[0], #call); for (int m = 0; m < n; m++) { atomdev_t* h = handles[m]; const atomdev_hyd_slice& s = slices[m]; (void)s; \
                        if (int rc__ = (call)) return rc__; } } while (0)
    // :234-297: the atmosphere's surface, the sea ground, the initial conditions, the first old -> new copies
    // the readers are host work (text files, the needle smoothing): one host thread per slice, at most 16 at a time
    auto read_files = [&](int m) -> int {
        atomdev_t* h = handles[m];
        const atomdev_hyd_slice& s = slices[m];
        if (s.transfer_file) { StepClock c__(h, "atomdev_hyd_read_transfer_file"); if (int rc = atomdev_hyd_read_transfer_file(h, s.transfer_file)) return rc; }
        { StepClock c__(h, "atomdev_hyd_read_sea_ground"); if (int rc = atomdev_hyd_read_sea_ground(h, s.bathymetry_file)) return rc; }
        if (s.start.Ma == 0 && s.salinity_file) { StepClock c__(h, "atomdev_hyd_read_surface_salinity"); if (int rc = atomdev_hyd_read_surface_salinity(h, s.salinity_file)) return rc; }
        return 0;
    };
    if (n == 1) {
        if (int rc = read_files(0)) return rc;
    } else {
        std::vector<int> status(n, 0);
        std::vector<std::string> message(n);
        for (int first = 0; first < n; first += 16) {
            std::vector<std::thread> readers;
            for (int m = first; m < n && m < first + 16; m++)
                readers.emplace_back([&, m]() {
                    status[m] = read_files(m);
                    if (status[m]) message[m] = atomdev_last_error();  // the error text is kept per thread
                });
            for (std::thread& t : readers) t.join();
        }
        for (int m = 0; m < n; m++)
            if (status[m]) { atomdev_set_error(message[m]); return status[m]; }
    }
    EACH(atomdev_hyd_ic_ekman_spiral(h, s.start.i_beg));
    EACH(s.start.Ma <= 41 ? atomdev_hyd_ic_circumpolar_current(h, s.start.i_beg) : 0);  // Drake passage
    EACH(atomdev_hyd_bc_temperature_salinity(h, &s.start));
    EACH(atomdev_hyd_pressure_density(h));
    EACH(atomdev_move_data_to_new_arrays(h, -1));
    EACH(atomdev_move_data_to_new_arrays(h, h->im - 1));
    // :314-404: surface iterations (their two queries are printed only)
    for (int p = 1; p <= s0.pressure_iter_max_2D; p++) {
        for (int v = 1; v <= s0.velocity_iter_max_2D; v++) {
            EACH(atomdev_hyd_bc_theta(h));
            EACH(atomdev_hyd_bc_phi(h));
            EACH(atomdev_hyd_value_limitation(h));
            {
                StepClock clock__(handles[0], "atomdev_hyd_rk2d");
                if (n == 1) { if (int rc = atomdev_hyd_rk2d(handles[0])) return rc; }
                else if (int rc = atomdev_rk2d_batch(handles, n)) return rc;
            }
            EACH(atomdev_move_data_to_new_arrays(h, h->im - 1));
        }
        EACH(atomdev_hyd_pressure2d(h));
    }
    // :420-610: volume iterations
    std::vector<double> before(n, 0.), after(n, 0.);
    std::vector<int> loc(3 * (size_t)n, 0);
    double steady[6]; int steady_loc[18];
    int iter_cnt = 1;  // :411
    for (int p = 1; p <= s0.pressure_iter_max; p++) {
        for (int v = 1; v <= s0.velocity_iter_max; v++) {
            EACH(s.queries ? (before[m] = 0., atomdev_hyd_residuum3d(h, &before[m], &loc[3 * m])) : 0);
            EACH(atomdev_hyd_bc_radius_theta_phi(h));  // RB_radius, RB_theta, RB_phi in one launch
            EACH(atomdev_hyd_pressure_density(h));
            EACH(atomdev_hyd_value_limitation(h));
            EACH(atomdev_hyd_rk3d(h));
            EACH(atomdev_hyd_solid_ground(h));
            EACH(s.queries ? (after[m] = 0., loc[3 * m] = loc[3 * m + 1] = loc[3 * m + 2] = 0,
                              atomdev_hyd_residuum3d(h, &after[m], &loc[3 * m])) : 0);
            EACH(s.queries ? atomdev_hyd_steady3d(h, steady, steady_loc) : 0);
            EACH(atomdev_hyd_run_data(h));
            EACH(atomdev_move_data_to_new_arrays(h, -1));
            iter_cnt++;
        }
        EACH(atomdev_hyd_pressure3d(h));
        EACH(s.checkpoint > 0 && p % s.checkpoint == 0 ? write_files(h, s, iter_cnt - 1, false) : 0);  // :612-614
    }
    EACH(write_files(h, s, iter_cnt - 1, true));  // :626
    for (int m = 0; m < n; m++) {
        if (residuum) { residuum[2 * m] = before[m]; residuum[2 * m + 1] = after[m]; }
        if (residuum_loc) for (int q = 0; q < 3; q++) residuum_loc[3 * m + q] = loc[3 * m + q];
    }
    EACH(atomdev_sync(h));
#undef EACH
    return 0;
}

extern "C" int atomdev_hyd_run_time_slice(atomdev_t* h, const atomdev_hyd_slice* s, double residuum[2], int residuum_loc[3]) {
    ATOMDEV_REQUIRE(h && s, "atomdev_hyd_run_time_slice: needs a hydrosphere handle and the slice description");
    return atomdev_hyd_run_time_slices(&h, s, 1, residuum, residuum_loc);
}
