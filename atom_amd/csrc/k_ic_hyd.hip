// The hydrosphere's start of a time slice (cHydrosphereModel.cpp:244-292): the bathymetry and surface-salinity
// readers (host work: text files, an order-dependent smoothing) and the initial conditions of
// hydrosphere/BC_Thermohalin.cpp as column kernels - one thread per (theta, phi) column, 64 neighbouring
// longitudes per wave so that every level is read and written as coalesced rows.
#include <algorithm>
#include <cmath>
#include <fstream>
#include <string>
#include <vector>

#include "atomdev_internal.h"
#include "text_numbers.h"

namespace {

// ---- BC_Thermohalin::IC_v_w_EkmanSpiral (BC_Thermohalin.cpp:88-360) ----

// One latitude band of the surface step: which component picks the sign of the turning angle, and the three
// assignments the reference makes one after the other (each tests the values the one before left behind).
struct Band {
    int sel;         // angle = +alfa - E when: 0: w >= 0   1: v >= 0   2: v <= 0
    int third;       // condition of the third assignment:  0: w <= 0   1: v <= 0   2: v >= 0
    double sv[3], sw[3];  // signs of v = s * m cos(angle), w = s * m sin(angle) in the three assignments
};
__constant__ Band c_bands[6] = {
    { 0, 0, { +1., +1., +1. }, { +1., +1., +1. } },   // rows  0 ..  29 (:106-140)
    { 1, 1, { +1., +1., -1. }, { +1., -1., -1. } },   // rows 30 ..  59 (:143-177)
    { 0, 0, { +1., +1., +1. }, { +1., +1., +1. } },   // rows 60 ..  89 (:180-214)
    { 0, 0, { -1., -1., -1. }, { +1., +1., +1. } },   // rows 90 .. 119 (:217-251)
    { 2, 2, { -1., -1., +1. }, { +1., -1., -1. } },   // rows 120 .. 149 (:254-288)
    { 0, 0, { -1., -1., -1. }, { +1., +1., +1. } },   // rows 150 .. jm-1 (:291-324)
};

// tab: [0][i] exp(-gam_z), [1][i] sin(gam_z), [2][i] cos(gam_z) of level i, from the host's libm
__global__ __launch_bounds__(256) void k_hyd_ekman(double* __restrict__ v, double* __restrict__ w, const uint8_t* __restrict__ flags,
                                                   const double* __restrict__ tab, int im, int jm, int km, int i_beg,
                                                   double water_wind, double u_0, double Ekman_angle) {
    const size_t n2 = (size_t)jm * km;
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n2) return;
    const int j = (int)(q / km);
    size_t p = (size_t)(im - 1) * n2 + q;
    double vs = v[p], ws = w[p];
    if (!(flags[p] & FL_LAND)) {
        vs = water_wind * vs / u_0;  // :99-100
        ws = water_wind * ws / u_0;
        double m = sqrt(vs * vs + ws * ws);
        if (ws == 0.) ws = 1.e-6;
        if (vs == 0.) vs = 1.e-6;
        if (m == 0.) m = 1.e-6;
        const Band& B = c_bands[j < 150 ? j / 30 : 5];
        const double alfa = asin(fabs(ws) / m);
        const bool plus = B.sel == 0 ? ws >= 0. : (B.sel == 1 ? vs >= 0. : vs <= 0.);
        const double angle = plus ? alfa - Ekman_angle : -alfa - Ekman_angle;
        const double mc = m * cos(angle), ms = m * sin(angle);
        if (ws >= 0. && angle >= 0.) { vs = B.sv[0] * mc; ws = B.sw[0] * ms; }
        if (ws >= 0. && angle <= 0.) { vs = B.sv[1] * mc; ws = B.sw[1] * ms; }
        if (B.third == 0 ? ws <= 0. : (B.third == 1 ? vs <= 0. : vs >= 0.)) { vs = B.sv[2] * mc; ws = B.sw[2] * ms; }
        v[p] = vs; w[p] = ws;
    }
    // :335-359: every water cell below takes the pair of the level above it (whatever that level holds)
    const bool north = j <= (jm - 1) / 2;
    double v_g = vs, w_g = ws;
    for (int i = im - 2; i >= i_beg; i--) {
        p -= n2;
        if (!(flags[p] & FL_LAND)) {
            const double e = tab[i], s = north ? tab[im + i] : -tab[im + i], c = tab[2 * im + i];
            const double vn = w_g * e * s + v_g * (1. - e * c);
            const double wn = w_g * (1. - e * c) - v_g * e * s;
            v[p] = vn; w[p] = wn;
            v_g = vn; w_g = wn;
        } else {
            v_g = v[p]; w_g = w[p];
        }
    }
}

// ---- BC_Thermohalin::IC_CircumPolar_Current (:676-690) ----
__global__ void k_hyd_circumpolar(double* __restrict__ w, const uint8_t* __restrict__ flags, int im, int jm, int km, int i_beg, double w_acc) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= km) return;
    const int j = 147 + blockIdx.y, i = i_beg + blockIdx.z;
    const size_t p = idx3(jm, km, i, j, k);
    if (!(flags[p] & FL_LAND)) w[p] = w_acc;
}

// ---- BC_Thermohalin::BC_Temperature_Salinity (:444-512) ----
__global__ __launch_bounds__(256) void k_hyd_temperature_salinity(double* __restrict__ t, double* __restrict__ c, const uint8_t* __restrict__ flags,
                                                                  int im, int jm, int km, int i_beg, int i_max, int Ma, double t_cretaceous,
                                                                  double c_cretaceous, double ta, double ca, double t_0, double c_0) {
    const size_t n2 = (size_t)jm * km;
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n2) return;
    size_t p = (size_t)(im - 1) * n2 + q;
    double ts, cs;
    if (!(flags[p] & FL_LAND)) {
        ts = t[p] + t_cretaceous;
        cs = c[p];
        if (ts * t_0 - t_0 < 4.) ts = ta + t_cretaceous;
        if (Ma == 0) cs = cs + c_cretaceous / c_0;
    } else {
        ts = ta + t_cretaceous;
        cs = ca + c_cretaceous / c_0;
    }
    t[p] = ts; c[p] = cs;
    const double span = (double)(i_max * i_max - i_beg * i_beg);
    const double tm_tbeg = (ts - ta) / span, cm_cbeg = (cs - ca) / span;
    for (int i = im - 2; i >= 0; i--) {  // levels are independent of each other: any order
        p -= n2;
        const bool water = !(flags[p] & FL_LAND);
        if (i >= i_beg) {
            double ti = ta, ci = ca;
            if (water) {
                const double d = (double)(i * i - i_beg * i_beg);
                ti = ta + tm_tbeg * d;
                ci = ca + cm_cbeg * d;
                if (ti * t_0 - t_0 < 4.) { ti = ta; ci = ca; }
            }
            t[p] = ti; c[p] = ci;
        } else if (water) {
            t[p] = ta; c[p] = ca;
        }
    }
}

}  // namespace

extern "C" void atomdev_default_hyd_start(int im, atomdev_hyd_start* s) {
    s->i_beg = 0; s->i_max = im - 1;
    s->Ma = 0; s->Ma_max = 300; s->Ma_max_half = 150;
    s->t_average = 15.; s->t_cretaceous_max = 10.;
}

extern "C" int atomdev_hyd_ic_ekman_spiral(atomdev_t* h, int i_beg) {
    ATOMDEV_REQUIRE(h && h->model == ATOMDEV_HYD, "atomdev_hyd_ic_ekman_spiral: needs a hydrosphere handle");
    ATOMDEV_REQUIRE(h->jm > 150, "atomdev_hyd_ic_ekman_spiral: the latitude bands are rows of the 1-degree grid, jm must exceed 150");
    ATOMDEV_REQUIRE(i_beg >= 0 && i_beg < h->im - 1, "atomdev_hyd_ic_ekman_spiral: i_beg outside 0 .. im-2");
    ATOMDEV_TRY(hipSetDevice(h->device));
    if (int rc = atomdev_refresh_flags(h)) return rc;
    const int im = h->im;
    ATOMDEV_REQUIRE((size_t)3 * im <= h->n2, "atomdev_hyd_ic_ekman_spiral: more levels than a third of a surface");
    std::vector<double> tab((size_t)3 * im, 0.);
    for (int i = im - 2; i >= i_beg; i--) {  // :337-340, with the host's libm like the reference
        const double gam_z = M_PI * (double)(i - i_beg) / (double)(im - 1 - i_beg);
        tab[i] = std::exp(-gam_z);
        tab[im + i] = std::sin(gam_z);
    }
    for (int i = im - 2; i >= i_beg; i--) {  // a loop of its own: no sin/cos pair for the compiler to merge
        const double gam_z = M_PI * (double)(i - i_beg) / (double)(im - 1 - i_beg);
        tab[2 * im + i] = std::cos(gam_z);
    }
    ATOMDEV_TRY(hipMemcpyAsync(h->plane, tab.data(), tab.size() * 8, hipMemcpyHostToDevice, h->stream));
    if (int rc__ = atomdev_sync_stream(h)) return rc__;  // tab leaves scope
    const double pi180 = 180. / M_PI;
    const double Ekman_angle = 45.0 / pi180;  // :77-81
    const unsigned nb = (unsigned)((h->n2 + 255) / 256);
    hipLaunchKernelGGL(k_hyd_ekman, dim3(nb), dim3(256), 0, h->stream, h->f[ATOMDEV_V], h->f[ATOMDEV_W], h->flags, h->plane,
                       im, h->jm, h->km, i_beg, .03, h->p.u_0, Ekman_angle);
    ATOMDEV_TRY(hipGetLastError());
    return 0;
}

extern "C" int atomdev_hyd_ic_circumpolar_current(atomdev_t* h, int i_beg) {
    ATOMDEV_REQUIRE(h && h->model == ATOMDEV_HYD, "atomdev_hyd_ic_circumpolar_current: needs a hydrosphere handle");
    ATOMDEV_REQUIRE(h->jm > 152, "atomdev_hyd_ic_circumpolar_current: rows 147..152 must exist");
    ATOMDEV_REQUIRE(i_beg >= 0 && i_beg < h->im, "atomdev_hyd_ic_circumpolar_current: i_beg outside the grid");
    ATOMDEV_TRY(hipSetDevice(h->device));
    if (int rc = atomdev_refresh_flags(h)) return rc;
    hipLaunchKernelGGL(k_hyd_circumpolar, dim3((h->km + 255) / 256, 6, h->im - i_beg), dim3(256), 0, h->stream, h->f[ATOMDEV_W], h->flags,
                       h->im, h->jm, h->km, i_beg, .01 / h->p.u_0);
    ATOMDEV_TRY(hipGetLastError());
    return 0;
}

extern "C" int atomdev_hyd_bc_temperature_salinity(atomdev_t* h, const atomdev_hyd_start* s) {
    ATOMDEV_REQUIRE(h && s && h->model == ATOMDEV_HYD, "atomdev_hyd_bc_temperature_salinity: needs a hydrosphere handle and the start values");
    ATOMDEV_REQUIRE(s->i_beg >= 0 && s->i_beg < h->im - 1 && s->i_max > s->i_beg, "atomdev_hyd_bc_temperature_salinity: i_beg / i_max outside the grid");
    ATOMDEV_REQUIRE(s->Ma_max != 0, "atomdev_hyd_bc_temperature_salinity: Ma_max is zero");
    ATOMDEV_TRY(hipSetDevice(h->device));
    if (int rc = atomdev_refresh_flags(h)) return rc;
    // :385-390, 421-424: the reference computes the Cretaceous offsets and then sets the temperature offset to zero
    double t_cretaceous_coeff = s->t_cretaceous_max / ((double)s->Ma_max_half - (double)(s->Ma_max_half * s->Ma_max_half / s->Ma_max));
    double t_cretaceous = t_cretaceous_coeff * (double)(-(s->Ma * s->Ma) / s->Ma_max + s->Ma);
    if (s->Ma == 0) t_cretaceous = 0.;
    t_cretaceous = 0.;
    const double c_average = (s->t_average + 346.) / 10.;
    double c_cretaceous = (s->t_average + t_cretaceous + 346.) / 10.;
    c_cretaceous = c_cretaceous - c_average;
    if (s->Ma == 0) c_cretaceous = 0.;
    const unsigned nb = (unsigned)((h->n2 + 255) / 256);
    hipLaunchKernelGGL(k_hyd_temperature_salinity, dim3(nb), dim3(256), 0, h->stream, h->f[ATOMDEV_T], h->f[ATOMDEV_C], h->flags,
                       h->im, h->jm, h->km, s->i_beg, s->i_max, s->Ma, t_cretaceous, c_cretaceous, h->p.ta, h->p.ca, h->p.t_0, h->p.c_0);
    ATOMDEV_TRY(hipGetLastError());
    return 0;
}

extern "C" int atomdev_hyd_read_surface_salinity(atomdev_t* h, const char* path) {
    ATOMDEV_REQUIRE(h && path && h->model == ATOMDEV_HYD, "atomdev_hyd_read_surface_salinity: needs a hydrosphere handle and a path");
    ATOMDEV_TRY(hipSetDevice(h->device));
    const int jm = h->jm, km = h->km;
    std::vector<double> top(h->n2), values;
    double* level = h->f[ATOMDEV_C] + (size_t)(h->im - 1) * h->n2;
    ATOMDEV_TRY(hipMemcpyAsync(top.data(), level, h->n2 * 8, hipMemcpyDeviceToHost, h->stream));
    bool opened = false;
    const bool fast = text_third_of_triples(path, h->n2, values, &opened);  // overlaps the download
    if (int rc__ = atomdev_sync_stream(h)) return rc__;
    ATOMDEV_REQUIRE(opened, std::string("could not open SurfaceSalinity_File file at ") + path);
    if (fast) {
        for (int k = 0; k < km; k++)
            for (int j = 0; j < jm; j++) {
                const double value = values[(size_t)k * jm + j];
                if (!(value < 0.)) top[(size_t)j * km + k] = value / h->p.c_0;
            }
    } else {  // short or malformed file: the stream's own treatment of it, as in the reference
        std::ifstream file(path);
        double lon = 0., lat = 0., value = 0.;
        for (int k = 0; k < km && !file.eof(); k++)
            for (int j = 0; j < jm; j++) {
                file >> lon;
                file >> lat;
                file >> value;
                if (!(value < 0.)) top[(size_t)j * km + k] = value / h->p.c_0;
            }
    }
    ATOMDEV_TRY(hipMemcpyAsync(level, top.data(), h->n2 * 8, hipMemcpyHostToDevice, h->stream));
    if (int rc__ = atomdev_sync_stream(h)) return rc__;
    return 0;
}

extern "C" int atomdev_hyd_read_sea_ground(atomdev_t* h, const char* path) {
    ATOMDEV_REQUIRE(h && path && h->model == ATOMDEV_HYD, "atomdev_hyd_read_sea_ground: needs a hydrosphere handle and a path");
    const int im = h->im, jm = h->jm, km = h->km;
    const double L_hyd = h->p.L_hyd;
    std::vector<uint8_t> solid(h->n3, 0);  // h as bytes: 1 = solid
    std::vector<double> bathymetry(h->n2, 0.), values;
    auto H = [&](int i, int j, int k) -> uint8_t& { return solid[idx3(jm, km, i, j, k)]; };
    auto entry = [&](int j, int k, double depth) {
        if (depth > 0.) depth = 0;
        const int i_boden = (im - 1) + int(std::floor(depth / L_hyd * (im - 1)));
        bathymetry[(size_t)j * km + k] = -depth;
        for (int i = 0; i <= i_boden; i++) H(i, j, k) = 1;
    };
    bool opened = false;
    if (text_third_of_triples(path, h->n2, values, &opened)) {
        for (int j = 0; j < jm; j++)
            for (int k = 0; k < km; k++) entry(j, k, values[(size_t)j * km + k]);
    } else {  // short or malformed file: the stream's own treatment of it, as in the reference
        ATOMDEV_REQUIRE(opened, std::string("could not open bathymetry file ") + path);
        std::ifstream file(path);
        int j = 0, k = 0;
        for (j = 0; j < jm && !file.eof(); j++)
            for (k = 0; k < km && !file.eof(); k++) {
                double lon, lat, depth = 999;
                file >> lon >> lat >> depth;
                entry(j, k, depth);
                if (file.fail()) {  // an entry that is no number: skip the rest of its line
                    file.clear();
                    std::string rest;
                    std::getline(file, rest);
                }
            }
        ATOMDEV_REQUIRE(j == jm && k == km, "wrong topography file size");
    }
    // -180..180 -> 0..360 (lib/Utils.cpp:31-37): the halves of a row change places, the last entry repeats the first
    auto rotate = [km](auto* row) {
        for (int n = 0; n < km / 2; n++) std::swap(row[n], row[km / 2 + n]);
        row[km - 1] = row[0];
    };
    for (int jj = 0; jj < jm; jj++) {
        rotate(&bathymetry[(size_t)jj * km]);
        for (int i = 0; i < im; i++) rotate(&H(i, jj, 0));
    }
    // Needles (:88-113): a solid cell with water on its four sides becomes water, and from the first such cell on
    // (`h_center` is never reset) every visited point clears the diagonal neighbours that have water two cells away
    // on both of their sides.  The reference visits the points k-outer, j, i-inner; a point reads and writes its own
    // level only, and the one thing levels share is whether the first needle has been met yet.  So: the first needle
    // of every level on the unsmoothed mask (nothing is modified before the first needle of all), the earliest of
    // them in visiting order, then level by level - a plane of bytes that stays in cache - with `met` switched on
    // from that moment of the visiting order.
    auto needle = [&](int i, int jj, int kk) {
        return H(i, jj, kk) && !H(i, jj + 1, kk) && !H(i, jj - 1, kk) && !H(i, jj, kk + 1) && !H(i, jj, kk - 1);
    };
    long first = -1;  // (k * jm + j) * im + i of the first needle
    for (int i = 2; i < im - 2; i++)
        for (int kk = 2; kk < km - 2; kk++) {
            bool found = false;
            for (int jj = 2; jj < jm - 2 && !found; jj++)
                if (needle(i, jj, kk)) {
                    const long at = ((long)kk * jm + jj) * im + i;
                    if (first < 0 || at < first) first = at;
                    found = true;
                }
            if (found || (first >= 0 && ((long)(kk + 1) * jm) * im > first)) break;  // later columns cannot be earlier
        }
    if (first >= 0)
        for (int i = 2; i < im - 2; i++)
            for (int kk = 2; kk < km - 2; kk++)
                for (int jj = 2; jj < jm - 2; jj++) {
                    if (needle(i, jj, kk)) H(i, jj, kk) = 0;
                    if (((long)kk * jm + jj) * im + i >= first) {
                        const bool n2 = !H(i, jj - 2, kk), s2 = !H(i, jj + 2, kk);
                        if (n2 && !H(i, jj, kk + 2)) H(i, jj - 1, kk + 1) = 0;
                        if (n2 && !H(i, jj, kk - 2)) H(i, jj - 1, kk - 1) = 0;
                        if (s2 && !H(i, jj, kk - 2)) H(i, jj + 1, kk - 1) = 0;
                        if (s2 && !H(i, jj, kk + 2)) H(i, jj + 1, kk + 1) = 0;
                    }
                }
    if (int rc = atomdev_upload_mask_bytes(h, solid.data())) return rc;
    return atomdev_upload2d(h, ATOMDEV_TOPOGRAPHY, bathymetry.data());
}
