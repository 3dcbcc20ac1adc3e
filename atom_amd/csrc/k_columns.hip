// Operators that walk each (theta, phi) column level by level: a thread owns one column, a
// wavefront covers 64 neighbouring longitudes, so every level is one coalesced row segment.
#include "atomdev_internal.h"

namespace {

constexpr int TPB = 256;

// BC_Thermohalin::BC_Pressure_Density (hydrosphere/BC_Thermohalin.cpp:518-583).  The surface
// level im-1 is set from the air above it (:537-553); below, the water density of level i follows
// from the density of level i+1 and the pressure step between the two (:556-582), so the walk
// is sequential from the surface down.  r_air and R_Air are the function's own locals (:533-534),
// not the model parameters.  Every operation is written in the reference's association order.
__global__ void k_hyd_pressure_density(const double* __restrict__ t, const double* __restrict__ c,
                                       double* __restrict__ p_stat, double* __restrict__ r_water,
                                       double* __restrict__ r_salt_water, int im, size_t n2, double t_0, double c_0,
                                       double g, double L_hyd, double p_0, double r_0_water) {
    const size_t col = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= n2) return;
    const double E_water = 2.15e9, beta_water = 8.8e-5, r_air = 1.2041, R_Air = 287.1;
    const double step = L_hyd / (double)(im - 1);
    const double p_ref = p_0 / 1000.;

    size_t id = (size_t)(im - 1) * n2 + col;
    double t_above = t[id];
    double t_Celsius_0 = t_above * t_0 - t_0;
    double ps_above = .01 * (r_air * R_Air * t_above * t_0) / 1000.;
    double rw_above = r_0_water;
    {
        const double cc = c[id], t_Celsius_1 = t_Celsius_0;
        const double alfa_t_p = .0708 * (1. + .068 * t_Celsius_1);
        const double gamma_t_p = .003 * (1. - .012 * t_Celsius_1);
        p_stat[id] = ps_above;
        r_water[id] = rw_above;
        r_salt_water[id] = 999.83 + .808 * cc * c_0 - alfa_t_p * t_Celsius_1 - gamma_t_p * (c_0 - cc * c_0) * t_Celsius_1;
    }
    double t_here = t[id - n2], c_here = c[id - n2];
    for (int i = im - 2; i >= 0; i--) {
        id -= n2;
        const double tt = t_here, cc = c_here;
        if (i > 0) { t_here = t[id - n2]; c_here = c[id - n2]; }  // next level's loads fly during this level's divisions
        const double d_i = (double)(im - 1 - i);
        const double t_Celsius_1 = tt * t_0 - t_0;
        const double ps = rw_above * g * d_i * step / 100000. + p_ref;
        const double rw = rw_above / (1. + beta_water * (t_Celsius_1 - t_Celsius_0)) / (1. - (ps - ps_above) / E_water * 1e5);
        const double p_km = d_i * step / 1000.;
        const double C_p = 999.83 + 5.053 * p_km - .048 * p_km * p_km;
        const double beta_p = .808 - .0085 * p_km;
        const double alfa_t_p = .0708 * (1. + .351 * p_km + .068 * (1. - .0683 * p_km) * t_Celsius_1);
        const double gamma_t_p = .003 * (1. - .059 * p_km - .012 * (1. - .064 * p_km) * t_Celsius_1);
        p_stat[id] = ps;
        r_water[id] = rw;
        r_salt_water[id] = C_p + beta_p * cc * c_0 - alfa_t_p * t_Celsius_1 - gamma_t_p * (c_0 - cc * c_0) * t_Celsius_1;
        ps_above = ps; rw_above = rw; t_Celsius_0 = t_Celsius_1;
    }
}

}  // namespace

extern "C" int atomdev_hyd_pressure_density(atomdev_t* h) {
    ATOMDEV_REQUIRE(h && h->model == ATOMDEV_HYD, "atomdev_hyd_pressure_density: not a hydrosphere handle");
    ATOMDEV_TRY(hipSetDevice(h->device));
    const atomdev_params& P = h->p;
    hipLaunchKernelGGL(k_hyd_pressure_density, dim3((unsigned)((h->n2 + TPB - 1) / TPB)), dim3(TPB), 0, h->stream,
                       h->f[ATOMDEV_T], h->f[ATOMDEV_C], h->f[ATOMDEV_P_STAT], h->f[ATOMDEV_R_WATER],
                       h->f[ATOMDEV_R_SALT_WATER], h->im, h->n2, P.t_0, P.c_0, P.g, P.L_hyd, P.p_0, P.r_0_water);
    ATOMDEV_TRY(hipGetLastError());
    return 0;
}
