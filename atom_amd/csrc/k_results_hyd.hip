// Results_Hyd::run_data (hydrosphere/Results_Hyd.cpp:102-320) without its printout, and the PlotData tables of both
// models (atmosphere/PostProcess_Atm.cpp:624-652, hydrosphere/PostProcess_Hyd.cpp:676-705).
#include <cstdio>
#include <string>

#include "atomdev_internal.h"
#include "fixed4.h"
#include <memory>

#include "host_parallel.h"

namespace {

constexpr int TPB = 256;
constexpr double c43 = 4. / 3., c13 = 1. / 3.;

struct RunArgs {
    const double *v, *w, *c, *salt_finger, *salt_diffusion, *buoyancy;
    const uint8_t* flags;
    double *aux_v, *aux_w;  // Ekman-layer integrals, [jm][km]
    double *salt_finger_2d, *salt_diffusion_2d, *buoyancy_2d, *salt_total;
    int im, jm, km, i_Ekman, i_diff, i_half;
    double dr, c_0;
};

// Simpson's rule as lib/Utils.h:58-65 applies it: over entries 0..n of a buffer that the caller filled at
// i_Ekman..im-2 (water: the velocity, land: 0) and that holds 0 everywhere else
__device__ __forceinline__ double ekman_integral(const RunArgs& a, const double* x, size_t q, size_t n2) {
    const int n = a.i_diff;
    if (n % 2 != 0) return 0.;  // the reference refuses and leaves the 0 it started from
    auto value = [&](int l) {
        if (l < a.i_Ekman || l > a.im - 2) return 0.;
        const size_t id = (size_t)l * n2 + q;
        return (a.flags[id] & FL_LAND) ? 0. : x[id];
    };
    double sum_even = 0, sum_odd = 0;
    for (int i = 1; i < n; i += 2) sum_odd += 4 * value(i);
    for (int i = 2; i < n; i += 2) sum_even += 2 * value(i);
    return a.dr / 3 * (value(0) + sum_odd + sum_even + value(n));
}

// per column: the two integrals (:136-160) and the sums over the upper half (:227-238)
__global__ void k_hyd_run_columns(RunArgs a) {
    const size_t n2 = (size_t)a.jm * a.km;
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n2) return;
    a.aux_v[q] = ekman_integral(a, a.v, q, n2);
    a.aux_w[q] = ekman_integral(a, a.w, q, n2);
    double sf = 0., sd = 0., bf = 0., st = 0.;
    for (int i = a.i_half; i < a.im; i++) {
        const size_t id = (size_t)i * n2 + q;
        if (!(a.flags[id] & FL_LAND)) {
            sf += a.salt_finger[id]; sd += a.salt_diffusion[id]; bf += a.buoyancy[id]; st += a.c[id] * a.c_0;
        }
    }
    a.salt_finger_2d[q] = sf; a.salt_diffusion_2d[q] = sd; a.buoyancy_2d[q] = bf; a.salt_total[q] = st;
}

// BottomWater, Upwelling, Downwelling at every (j, k) (:112-195)
__global__ void k_hyd_bottom_water(const double* __restrict__ aux_v, const double* __restrict__ aux_w,
                                   const double* __restrict__ sinthe, double rad_top, double dthe, double dphi, double u_0,
                                   int jm, int km, double* __restrict__ bottom, double* __restrict__ up,
                                   double* __restrict__ down) {
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= (size_t)jm * km) return;
    const int j = (int)(q / km), k = (int)(q - (size_t)j * km);
    const int j_half = (jm - 1) / 2;
    double b = 0., u = 0., d = 0.;
    if (k >= 1 && k < km - 1 && j >= 1 && j < jm - 1) {
        if (j < j_half - 1 || j >= j_half + 2) {
            const double rmsinthe = rad_top * sinthe[j];
            const double aw = aux_w[q + 1];
            b = (aux_v[q + km] - aux_v[q - km]) / (2. * rad_top * dthe) + (aw - aw) / (2. * rmsinthe * dphi);
            b = b * u_0;
        }
        u = b >= 0. ? b : 0.;
        d = b < 0. ? b : 0.;
    }
    bottom[q] = b; up[q] = u; down[q] = fabs(d);
}

// rows j = 0 and jm-1 of the two surface fields (:197-209), then columns k = 0 and km-1 with their mean (:211-225)
__global__ void k_surface_fill_theta(double* up, double* down, int jm, int km) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= km) return;
    double* x = blockIdx.y ? down : up;
    x[k] = c43 * x[km + k] - c13 * x[2 * km + k];
    x[(size_t)(jm - 1) * km + k] = c43 * x[(size_t)(jm - 2) * km + k] - c13 * x[(size_t)(jm - 3) * km + k];
}
__global__ void k_surface_fill_phi(double* up, double* down, int jm, int km) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= jm) return;
    double* x = (blockIdx.y ? down : up) + (size_t)j * km;
    const double lo = c43 * x[1] - c13 * x[2], hi = c43 * x[km - 2] - c13 * x[km - 3];
    const double m = (lo + hi) / 2.;
    x[0] = m; x[km - 1] = m;
}

// bottom and top level of the four diagnostic 3-D fields (:242-264)
struct Four { double* p[4]; };
__global__ void k_fill_r_neumann(Four f, int im, size_t n2) {
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n2) return;
    double* x = f.p[blockIdx.y];
    const size_t top = (size_t)(im - 1) * n2 + q;
    x[q] = c43 * x[n2 + q] - c13 * x[2 * n2 + q];
    x[top] = c43 * x[top - n2] - c13 * x[top - 2 * n2];
}

int write_plot_data(atomdev* h, const char* path, bool atm) {
    ATOMDEV_TRY(hipSetDevice(h->device));
    const int jm = h->jm, km = h->km, L = atm ? 0 : h->im - 1;
    const size_t n2 = h->n2;
    auto values = std::make_shared<std::vector<double>>(8 * n2);
    std::vector<double>& s = *values;
    const int f3[5] = { ATOMDEV_H, ATOMDEV_V, ATOMDEV_W, ATOMDEV_T, ATOMDEV_C };
    const int f2a[2] = { ATOMDEV_PRECIPITATION, ATOMDEV_PRECIPITABLE_WATER };
    const int f2h[3] = { ATOMDEV_BOTTOM_WATER, ATOMDEV_UPWELLING, ATOMDEV_DOWNWELLING };
    for (int n = 0; n < 5; n++)
        ATOMDEV_TRY(hipMemcpyAsync(&s[n * n2], h->f[f3[n]] + (size_t)L * n2, n2 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    const int n2d = atm ? 2 : 3;
    for (int n = 0; n < n2d; n++)
        ATOMDEV_TRY(hipMemcpyAsync(&s[(5 + n) * n2], h->g[atm ? f2a[n] : f2h[n]], n2 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    if (int rc__ = atomdev_sync_stream(h)) return rc__;
    const double u_0 = h->p.u_0, t_0 = h->p.t_0;
    const std::string file(path);
    // The values are on the host: inside a one-call slice the text is made and written by a thread of its own while the
    // slice goes on (the call that ends the slice waits for it).
    return atomdev_defer_file(h, file, [values, file, atm, jm, km, n2, u_0, t_0]() -> std::string {
        const std::vector<double>& s = *values;
        FILE* f = std::fopen(file.c_str(), "w");
        if (!f) return "could not open PlotData file " + file;
        std::string out = atm ? "lons(deg), lats(deg), topography, v-velocity(m/s), w-velocity(m/s), velocity-mag(m/s), "
                                "temperature(Celsius), water_vapour(g/kg), precipitation(mm), precipitable water(mm)\n"
                              : "lons(deg), lats(deg), topography, v-velocity(m/s), w-velocity(m/s), velocity-mag(m/s), "
                                "temperature(Celsius), salinity(psu), bottom_water(m/s), upwelling(m/s), downwelling(m/s)\n";
        std::fwrite(out.data(), 1, out.size(), f);
        // the rows of a longitude are a piece of text of their own; a few host threads make the pieces side by side
        std::vector<std::string> piece(km);
        host_parallel_run((size_t)km, [&](size_t kk) {
            const int k = (int)kk;
            std::string& o = piece[kk];
            o.reserve((size_t)jm * 96);
            char b[400 * 9 + 64];
            for (int j = 0; j < jm; j++) {
                const size_t q = (size_t)j * km + k;
                const double v = s[n2 + q] * u_0, w = s[2 * n2 + q] * u_0;
                const double vel_mag = std::sqrt(v * v + w * w);  // pow(x, 2) is x * x, correctly rounded either way
                char* e = b + std::snprintf(b, 64, "%d %d ", k, 90 - j);
                auto num = [&e](double x, const char* after) { e = put_fixed4(e, x); while (*after) *e++ = *after++; };
                num(s[q], " "); num(v, " "); num(w, " "); num(vel_mag, " ");
                if (atm) {
                    num(s[3 * n2 + q] * t_0 - t_0, " "); num(s[4 * n2 + q] * 1000., " "); num(s[5 * n2 + q], " "); num(s[6 * n2 + q], " \n");
                } else {
                    num(s[3 * n2 + q] * 273.15 - 273.15, " "); num(s[4 * n2 + q] * 35., " "); num(s[5 * n2 + q], " ");
                    num(s[6 * n2 + q], "   "); num(s[7 * n2 + q], " \n");
                }
                o.append(b, (size_t)(e - b));
            }
        });
        for (const std::string& o : piece) std::fwrite(o.data(), 1, o.size(), f);
        const bool bad = std::ferror(f) != 0;
        if (std::fclose(f) != 0 || bad) return "write failed: " + file;
        return std::string();
    });
}

}  // namespace

extern "C" int atomdev_hyd_run_data(atomdev_t* h) {
    ATOMDEV_REQUIRE(h && h->model == ATOMDEV_HYD, "atomdev_hyd_run_data: not a hydrosphere handle");
    ATOMDEV_TRY(hipSetDevice(h->device));
    if (int rc = atomdev_refresh_flags(h)) return rc;
    const atomdev_params& P = h->p;
    const int im = h->im, jm = h->jm, km = h->km;
    RunArgs a;
    a.v = h->f[ATOMDEV_V]; a.w = h->f[ATOMDEV_W]; a.c = h->f[ATOMDEV_C];
    a.salt_finger = h->f[ATOMDEV_SALT_FINGER]; a.salt_diffusion = h->f[ATOMDEV_SALT_DIFFUSION]; a.buoyancy = h->f[ATOMDEV_BUOYANCY];
    a.flags = h->flags;
    a.aux_v = h->scr[0]; a.aux_w = h->scr[0] + h->n2;
    a.salt_finger_2d = h->g[ATOMDEV_SALT_FINGER_2D]; a.salt_diffusion_2d = h->g[ATOMDEV_SALT_DIFFUSION_2D];
    a.buoyancy_2d = h->g[ATOMDEV_BUOYANCY_FORCE_2D]; a.salt_total = h->g[ATOMDEV_SALT_TOTAL];
    a.im = im; a.jm = jm; a.km = km;
    const double coeff = 500. / P.L_hyd;  // assumed Ekman-layer depth of 500 m (:126-127)
    a.i_Ekman = (int)((im - 1) * (1. - coeff));
    a.i_diff = (im - 1) - a.i_Ekman;
    a.i_half = (im - 1) / 2;
    a.dr = P.dr; a.c_0 = P.c_0;
    ATOMDEV_REQUIRE(a.i_Ekman >= 0 && a.i_diff >= 0, "atomdev_hyd_run_data: the Ekman layer (500 m) is deeper than the model (L_hyd)");
    const unsigned nb2 = (unsigned)((h->n2 + TPB - 1) / TPB);
    hipLaunchKernelGGL(k_hyd_run_columns, dim3(nb2), dim3(TPB), 0, h->stream, a);
    double rad_top = 0.;
    ATOMDEV_TRY(hipMemcpyAsync(&rad_top, h->rad + (im - 1), sizeof(double), hipMemcpyDeviceToHost, h->stream));
    if (int rc__ = atomdev_sync_stream(h)) return rc__;
    double *up = h->g[ATOMDEV_UPWELLING], *down = h->g[ATOMDEV_DOWNWELLING];
    hipLaunchKernelGGL(k_hyd_bottom_water, dim3(nb2), dim3(TPB), 0, h->stream, a.aux_v, a.aux_w, h->sinthe, rad_top, P.dthe, P.dphi,
                       P.u_0, jm, km, h->g[ATOMDEV_BOTTOM_WATER], up, down);
    hipLaunchKernelGGL(k_surface_fill_theta, dim3((km + TPB - 1) / TPB, 2), dim3(TPB), 0, h->stream, up, down, jm, km);
    hipLaunchKernelGGL(k_surface_fill_phi, dim3((jm + TPB - 1) / TPB, 2), dim3(TPB), 0, h->stream, up, down, jm, km);
    Four four = { { h->f[ATOMDEV_BUOYANCY], h->f[ATOMDEV_SALT_FINGER], h->f[ATOMDEV_SALT_DIFFUSION], h->f[ATOMDEV_SALT_BALANCE] } };
    hipLaunchKernelGGL(k_fill_r_neumann, dim3(nb2, 4), dim3(TPB), 0, h->stream, four, im, h->n2);
    ATOMDEV_TRY(hipGetLastError());
    static const int ids[4] = { ATOMDEV_BUOYANCY, ATOMDEV_SALT_FINGER, ATOMDEV_SALT_DIFFUSION, ATOMDEV_SALT_BALANCE };
    for (int n = 0; n < 4; n++) {  // theta then phi, every level and every row (:266-320)
        if (int rc = atomdev_fill_theta_neumann(h, ids[n])) return rc;
        if (int rc = atomdev_fill_phi_seam(h, ids[n], 0, jm)) return rc;
    }
    return 0;
}

extern "C" int atomdev_atm_write_plot_data(atomdev_t* h, const char* path) {
    ATOMDEV_REQUIRE(h && h->model == ATOMDEV_ATM && path, "atomdev_atm_write_plot_data: bad argument");
    return write_plot_data(h, path, true);
}

extern "C" int atomdev_hyd_write_plot_data(atomdev_t* h, const char* path) {
    ATOMDEV_REQUIRE(h && h->model == ATOMDEV_HYD && path, "atomdev_hyd_write_plot_data: bad argument");
    return write_plot_data(h, path, false);
}
