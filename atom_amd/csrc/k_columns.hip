// Operators that walk each (theta, phi) column level by level: a thread owns one column, a
// wavefront covers 64 neighbouring longitudes, so every level is one coalesced row segment.
#include "atomdev_internal.h"
#include "cr_math.h"

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
        __builtin_nontemporal_store(ps_above, p_stat + id);   // (nothing in the loop reads p_stat: kept out of the last-level cache)
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
        __builtin_nontemporal_store(ps, p_stat + id);
        r_water[id] = rw;
        r_salt_water[id] = C_p + beta_p * cc * c_0 - alfa_t_p * t_Celsius_1 - gamma_t_p * (c_0 - cc * c_0) * t_Celsius_1;
        ps_above = ps; rw_above = rw; t_Celsius_0 = t_Celsius_1;
    }
}

// Results_MSL_Atm::run_MSL_data (atmosphere/Results_Atm.cpp:139-353) without its printout.
struct MslArgs {
    const double *t, *v, *w, *q_latent, *q_sensible, *radiation_3d, *co2;
    double *c, *cloud, *ice, *p_rain, *p_snow, *p_stat;
    double *q_radiation, *precipitable_water, *evaporation_penman, *evaporation_dalton, *q_evaporation, *q_latent_2d,
           *q_sensible_2d, *q_bottom, *co2_total, *precipitation;
    const uint8_t* flags;
    int im, radiation_model;
    size_t n2;
    double r_air, R_Air, t_0, g, L_atm, ep, hp, cp_l, lv, u_0, R_WaterVapour;
};

// Results_MSL_Atm::C_Dalton (Results_Atm.cpp:685-692)
__device__ __forceinline__ double c_dalton(double u_0, double v, double w) {
    const double C_max = -.053, v_max = 10.;
    const double vel_magnitude = sqrt(v * v + w * w) * u_0;
    return sqrt(C_max * C_max / v_max * vel_magnitude);
}

// One thread per column.  The surface branches (:149-243) are evaluated at every top of the topography
// (land cell with air above; the last one met wins the 2-D fields) and on the sea surface: the land bits of
// the column are gathered first (loads only, so they overlap), then only the levels that qualify are
// visited, in level order.  The precipitable-water sum (:328-338) follows in level order as well; it reads
// the p_stat values just written and c as the zero clamp (:280-290, done by k_clamp_at_zero afterwards)
// leaves it.  The reference's test of the cell above level im-1 (one level outside the array) counts as
// "not air".
__global__ void k_atm_msl_columns(MslArgs a) {
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= a.n2) return;
    const size_t n2 = a.n2;
    const int im = a.im;
    const double f_Penman = 2.;
    const double step = a.L_atm / (double)(im - 1);
    double q_rad = a.q_radiation[q];
    if (a.radiation_model >= 1) { q_rad = a.radiation_3d[q]; a.q_radiation[q] = q_rad; }
    double ev_penman = 0., ev_dalton = 0.;
    double ps0 = a.p_stat[q];
    // 64 levels per pass: bit l of `land` = level base + l is land; level base + 64 (or im) closes the window
    for (int base = 0; base < im; base += 64) {
        const int n = min(64, im - base);
        unsigned long long land = 0ull;
#pragma unroll 8
        for (int l = 0; l < n; l++)
            land |= (unsigned long long)(a.flags[(size_t)(base + l) * n2 + q] & FL_LAND) << l;
        const bool land_next = (base + n < im) ? (a.flags[(size_t)(base + n) * n2 + q] & FL_LAND) != 0 : true;
        const unsigned long long above = (land >> 1) | ((unsigned long long)land_next << (n - 1));
        unsigned long long visit = land & ~above;                 // summits
        if (base == 0 && !(land & 1ull)) visit |= 1ull;           // sea surface
        if (n < 64) visit &= (1ull << n) - 1ull;
        while (visit) {
            const int l = __ffsll((long long)visit) - 1;
            visit &= visit - 1ull;
            const int i = base + l;
            const bool summit = (land >> l) & 1ull;
            const size_t id = (size_t)i * n2 + q;
            const double t = a.t[id], cc = a.c[id];
            double ps;
            if (i == 0) { ps = (a.r_air * a.R_Air * t * a.t_0) * .01; ps0 = ps; }
            else ps = cr_exp(-a.g * (double)i * step / (a.R_Air * t * a.t_0)) * ps0;
            a.p_stat[id] = ps;
            const double t_Celsius = t * a.t_0 - a.t_0;
            const double e = cc * ps / a.ep;
            const double t_denom = t_Celsius + 234.175;
            const double E = a.hp * cr_exp(17.0809 * t_Celsius / t_denom);
            const double Delta = 4000. * E / (t_denom * t_denom);
            const double sat_deficit = (E - e);
            const double gamma = ps0 * a.cp_l / (a.ep * a.lv);
            const double v1 = a.v[id + n2], w1 = a.w[id + n2];  // level i + 1 exists in both branches
            const double E_a = .35 * (1. + .15 * sqrt((v1 * v1 + w1 * w1) / 2.) * a.u_0 * 3.6) * sat_deficit;
            double q_evap;
            if (summit) q_evap = 2300.;
            else if (t_Celsius >= -2.) q_evap = (2500.8 - 2.372 * (t * a.t_0 - a.t_0));
            else q_evap = (2500.8 - 2.372 * (t * a.t_0 - a.t_0)) + 300.;
            a.q_evaporation[q] = q_evap;
            const double q_lat = a.q_latent[id], q_sen = a.q_sensible[id];
            const double q_bot = -(q_rad - q_lat - q_sen);
            a.q_latent_2d[q] = q_lat; a.q_sensible_2d[q] = q_sen; a.q_bottom[q] = q_bot;
            double ed = c_dalton(a.u_0, a.v[id], a.w[id]) * sat_deficit * 24.;
            if (ed <= 0.) ed = 0.;
            double epm = f_Penman * .0346 * ((q_rad + q_bot) * Delta + gamma * E_a) / (Delta + gamma);
            if (epm <= 0.) epm = 0.;
            ev_penman = epm;
            ev_dalton = summit ? .5 * epm : ed;
        }
    }
    a.evaporation_penman[q] = ev_penman;
    a.evaporation_dalton[q] = ev_dalton;
    a.co2_total[q] = a.co2[q];

    double pw = 0.;
#pragma unroll 8
    for (int i = 0; i < im; i++) {
        const size_t id = (size_t)i * n2 + q;
        double cc = a.c[id];
        if (cc < 0.) cc = 0.;
        const double e = 100. * cc * a.p_stat[id] / a.ep;
        const double ah = e / (a.R_WaterVapour * a.t[id] * a.t_0);
        pw = pw + ah * a.L_atm / (double)(im - 1);
    }
    a.precipitable_water[q] = pw;
    double rain = a.p_rain[q], snow = a.p_snow[q];
    if (rain < 0.) rain = 0.;
    if (snow < 0.) snow = 0.;
    double pr = 86400. * (rain + snow);
    if (pr >= 25.) pr = 25.;
    if (pr <= 0) pr = 0.;
    a.precipitation[q] = pr;
}

// c, cloud, ice, P_rain, P_snow: negative values to zero (Results_Atm.cpp:280-290); blockIdx.y = field
struct Clamp5 { double* p[5]; };
__global__ void k_clamp_at_zero(Clamp5 f, size_t n) {
    const size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= n) return;
    double* x = f.p[blockIdx.y];
    if (x[id] < 0.) x[id] = 0.;
}

// BuoyancyForce on the bottom and top levels (Results_Atm.cpp:248-256): interior (j, k) only
__global__ void k_buoyancy_fill_r(double* b, const uint8_t* flags, int im, int jm, int km) {
    const size_t n2 = (size_t)jm * km;
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n2) return;
    const int j = (int)(q / km), k = (int)(q - (size_t)j * km);
    if (j < 1 || j > jm - 2 || k < 1 || k > km - 2) return;
    const double c43 = 4. / 3., c13 = 1. / 3.;
    const size_t top = (size_t)(im - 1) * n2 + q;
    b[q] = (flags[q] & FL_LAND) ? 0. : c43 * b[n2 + q] - c13 * b[2 * n2 + q];
    b[top] = c43 * b[top - n2] - c13 * b[top - 2 * n2];
}

// BC_Thermo::Latent_Heat (atmosphere/BC_Thermo.cpp:2418-2502), one thread per column.  i_topography as
// BC_Thermo's constructor builds it (:128-139): the highest land level at or below im-2, else 0.  The bottom
// formula (:2447-2462, one-sided over i_mount .. i_mount+2) divides by R_WaterVapour, the column formula
// (:2481-2499, central, i_mount+1 .. im-3) by r_water_vapour; the saturation pressures evaluated next to them are
// never used.  A column whose topography reaches level im-2 would read level im and gets no bottom value.
__global__ void k_atm_latent_heat(const double* __restrict__ t, const double* __restrict__ c, const double* __restrict__ ice,
                                  const double* __restrict__ p_stat, const uint8_t* __restrict__ flags,
                                  double* __restrict__ q_latent, double* __restrict__ q_sensible, int im, size_t n2,
                                  double L_atm, double dr, double ep, double t_0, double lv, double ls, double cp_l,
                                  double r_air, double R_WaterVapour, double r_water_vapour) {
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n2) return;
    const double step = L_atm / (double)(im - 1);
    const double coeff_Lv = lv / step, coeff_Ls = ls / step, coeff_Q = cp_l * r_air * t_0 / step;
    const double coeff_lat = .079, coeff_sen = .15;
    int i_mount = 0;
    for (int i = im - 2; i >= 0; i--)
        if (flags[(size_t)i * n2 + q] & FL_LAND) { i_mount = i; break; }
    if (i_mount + 2 <= im - 1) {
        const size_t id = (size_t)i_mount * n2 + q;
        const double e = .01 * c[id] * p_stat[id] / ep;
        const double a = e / (R_WaterVapour * t[id] * t_0);
        const double ql = -coeff_Lv * a * (-3. * c[id] + 4. * c[id + n2] - c[id + 2 * n2]) / (2. * dr);
        const double qi = -coeff_Ls * a * (-3. * ice[id] + 4. * ice[id + n2] - ice[id + 2 * n2]) / (2. * dr);
        q_latent[id] = coeff_lat * (ql + qi);
        q_sensible[id] = -coeff_sen * coeff_Q * (-3. * t[id] + 4. * t[id + n2] - t[id + 2 * n2]) / (2. * dr);
    }
    for (int i = i_mount + 1; i < im - 2; i++) {
        const size_t id = (size_t)i * n2 + q;
        const double e = .01 * c[id] * p_stat[id] / ep;
        const double a = e / (r_water_vapour * t[id] * t_0);
        const double ql = -coeff_Lv * a * (c[id + n2] - c[id - n2]) / (2. * dr);
        const double qi = -coeff_Ls * a * (ice[id + n2] - ice[id - n2]) / (2. * dr);
        q_latent[id] = coeff_lat * (ql + qi);
        q_sensible[id] = -coeff_sen * coeff_Q * (t[id + n2] - t[id - n2]) / (2. * dr);
    }
}

// BC_Thermo::BC_Radiation_multi_layer (atmosphere/BC_Thermo.cpp:166-424), one thread per column; the column's
// levels of epsilon_3D, radiation_3D and t are the working arrays, as in the reference, and alfa / beta of the
// Thomas recurrence live in two scratch fields.  The reference's transmission matrix CC is built by a
// recurrence whose first element is never set (:341-344), so every entry is 0 and so are the sums CCC and DDD
// taken from it (:357-362); they appear here as the zeros they are.
struct RadArgs {
    double *t, *c, *cloud, *ice, *epsilon_3d, *radiation_3d, *alfa, *beta;
    const double* p_stat;
    double *albedo, *epsilon, *radiation_surface;
    const uint8_t* flags;
    const int* i_trop;  // [jm]
    int im, jm, km;
    atomdev_radiation R;
    double t_0, ep, t_tropopause;
};

__device__ __forceinline__ double parabola(double x) { return x * x - 2 * x; }  // lib/Utils.h:41-43
// The reference writes pow(x, 4.) and pow(y, 1. / 4.) (:262, :380, :387 ...): the correctly rounded values from a few
// fused operations (cr_math.h), a hundredth of the cost of the general function.
__device__ __forceinline__ double pow4(double x) { return cr_pow4(x); }
__device__ __forceinline__ double root4(double y) { return cr_root4(y); }

__global__ void k_atm_radiation(RadArgs a) {
    const size_t n2 = (size_t)a.jm * a.km;
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n2) return;
    const int im = a.im, j = (int)(q / a.km);
    const double sigma = a.R.sigma, t_0 = a.t_0;
    const double j_max_half = (a.jm - 1) / 2;
    const double par_j = parabola(j / j_max_half);
    const int i_trop = a.i_trop[j];
#define AT(F, i) a.F[(size_t)(i) * n2 + q]
    // albedo where the column has a sea or a land surface (:210-218); i_topography (:128-139)
    bool surface = false;
    int i_mount = 0;
    {
        bool land = (AT(flags, 0) & FL_LAND) != 0;
        if (!land) surface = true;
        for (int i = 0; i < im - 1; i++) {
            const bool land_above = (AT(flags, i + 1) & FL_LAND) != 0;
            if (land && !land_above) surface = true;
            if (land) i_mount = i;  // the last land level at or below im-2
            land = land_above;
        }
    }
    if (surface) a.albedo[q] = (a.R.albedo_pole - a.R.albedo_equator) * par_j + a.R.albedo_pole;
    const double epsilon_eff_max = (a.R.epsilon_pole - a.R.epsilon_equator) * par_j + a.R.epsilon_pole;
    const double radiation_surface = (a.R.rad_pole - a.R.rad_equator) * par_j + a.R.rad_pole;
    a.radiation_surface[q] = radiation_surface;
    for (int i = 0; i <= i_trop; i++) {  // :238-266
        double cc = AT(c, i), cl = AT(cloud, i), ic = AT(ice, i);
        if (cc < 0.) { cc = 0.; AT(c, i) = 0.; }
        if (cl < 0.) { cl = 0.; AT(cloud, i) = 0.; }
        if (ic < 0.) { ic = 0.; AT(ice, i) = 0.; }
        const double e = (cc + cl + ic) * AT(p_stat, i) / a.ep;
        const double epsilon_eff = epsilon_eff_max - (a.R.epsilon_tropopause - epsilon_eff_max) * parabola((double)i / (im - 1));
        double eps;
        if (i >= i_mount) {
            eps = a.R.co2_coeff * epsilon_eff + .0416 * sqrt(e);
            AT(radiation_3d, i) = (1. - eps) * sigma * pow4(AT(t, i) * t_0);
        } else {
            eps = AT(epsilon_3d, i);
        }
        if (eps > 1.) eps = 1.;
        AT(epsilon_3d, i) = eps;
    }
    const double eps_trop = AT(epsilon_3d, i_trop);
    a.epsilon[q] = eps_trop;
    for (int i = i_mount - 1; i >= 0; i--) { AT(epsilon_3d, i) = AT(epsilon_3d, i_mount); AT(radiation_3d, i) = AT(radiation_3d, i_mount); }
    const double eps_above = AT(epsilon_3d, i_trop);  // read again: a mountain above the tropopause has just overwritten it
    const double rad_above = (1. - eps_above) * sigma * pow4(a.t_tropopause * t_0);
    for (int i = i_trop; i < im; i++) { AT(epsilon_3d, i) = eps_above; AT(t, i) = a.t_tropopause; AT(radiation_3d, i) = rad_above; }

    for (int iter_rad = 1; iter_rad <= 4; iter_rad++) {  // :293-399
        const double e_trop = AT(epsilon_3d, i_trop);
        const double rad_trop = (1. - e_trop) * sigma * pow4(AT(t, i_trop) * t_0);
        AT(radiation_3d, i_trop) = rad_trop;
        const double radiation_back = AT(epsilon_3d, i_mount + 1) * sigma * pow4(AT(t, i_mount + 1) * t_0);
        const double atmospheric_window = .1007 * radiation_surface;
        double rad_surf_diff = radiation_back + radiation_surface - atmospheric_window;
        const double fac_rad = (double)i_mount * .07 + 1.;
        rad_surf_diff = fac_rad * rad_surf_diff;
        // the column's AA: only AA[i - 1] and AA[i] are ever read together (:336, :353, :366)
        double AA_i = rad_surf_diff / AT(radiation_3d, i_trop);
        AT(radiation_3d, i_mount) = (1. - AT(epsilon_3d, i_mount)) * sigma * pow4(AT(t, i_mount) * t_0) / AT(radiation_3d, i_trop);
        // AA[i] for i_mount < i <= i_trop goes to the beta scratch until the Thomas loop has used it
        AT(beta, i_mount) = AA_i;
        for (int i = i_mount + 1; i <= i_trop; i++) {
            const double eps = AT(epsilon_3d, i);
            AA_i = AA_i * (1. - eps);
            AT(beta, i) = AA_i;
            const double tmp = sigma * pow4(AT(t, i) * t_0) / AT(radiation_3d, i_trop);
            AT(radiation_3d, i) = (1. - eps) * tmp;
        }
        const double CCC = 0., DDD = 0.;
        double alfa_prev = 0., beta_prev = 0., AA_prev = 0.;
        for (int i = i_mount; i < i_trop; i++) {
            const double AA_cur = AT(beta, i);
            double al, be;
            if (i == i_mount) {
                const double bb = -AT(radiation_3d, i), cc = AT(radiation_3d, i + 1), dd = -AA_cur;
                al = cc / bb; be = dd / bb;
            } else {
                const double aa = AT(radiation_3d, i - 1), bb = -2. * AT(radiation_3d, i), cc = AT(radiation_3d, i + 1);
                const double dd = -AA_prev + AA_cur + CCC - DDD;
                al = cc / (bb - aa * alfa_prev);
                be = (dd - aa * beta_prev) / (bb - aa * alfa_prev);
            }
            AT(alfa, i) = al; AT(beta, i) = be;
            alfa_prev = al; beta_prev = be; AA_prev = AA_cur;
        }
        AT(t, i_trop) = a.t_tropopause;
        double rad_next = (1. - AT(epsilon_3d, i_trop)) * sigma * pow4(a.t_tropopause * t_0);
        AT(radiation_3d, i_trop) = rad_next;
        for (int i = i_trop - 1; i >= i_mount; i--) {
            const double rad = -AT(alfa, i) * rad_next + AT(beta, i);
            AT(radiation_3d, i) = rad;
            AT(t, i) = .5 * (AT(t, i) + root4(rad / sigma) / t_0);
            rad_next = rad;
        }
        const double rad_top = (1. - AT(epsilon_3d, i_trop)) * sigma * pow4(a.t_tropopause * t_0);
        for (int i = i_trop; i < im; i++) { AT(t, i) = a.t_tropopause; AT(radiation_3d, i) = rad_top; }
    }
#undef AT
}

// The same function with the levels of a column spread over threads: a workgroup takes blockDim.x neighbouring
// columns, blockDim.y threads share a column's levels, and the column's t, epsilon_3D, radiation_3D, alfa and beta
// live in LDS.  What costs time - pow() at every level, twice per round - runs on all threads; the recurrences
// along the column (AA, the Thomas sweep and its back-substitution: a few multiplications and divisions per level)
// are done by the column's first thread between two barriers.
__global__ __launch_bounds__(1024) void k_atm_radiation_levels(RadArgs a) {
    extern __shared__ double lds[];  // [5][im][COLS] doubles, [COLS] ints, [im][COLS] bytes
    const int COLS = (int)blockDim.x, LT = (int)blockDim.y, tx = threadIdx.x, ty = threadIdx.y;
    const int im = a.im;
    const int plane = im * COLS;
    double *s_t = lds, *s_eps = lds + plane, *s_rad = lds + 2 * plane, *s_al = lds + 3 * plane, *s_be = lds + 4 * plane;
    int* s_mount = (int*)(lds + 5 * plane);
    uint8_t* s_land = (uint8_t*)(s_mount + COLS);
    const size_t n2 = (size_t)a.jm * a.km;
    size_t q = (size_t)blockIdx.x * COLS + tx;
    const bool valid = q < n2;
    if (!valid) q = n2 - 1;
    const int j = (int)(q / a.km);
    const double sigma = a.R.sigma, t_0 = a.t_0;
    const double j_max_half = (a.jm - 1) / 2;
    const double par_j = parabola(j / j_max_half);
    const int i_trop = a.i_trop[j];
    const double epsilon_eff_max = (a.R.epsilon_pole - a.R.epsilon_equator) * par_j + a.R.epsilon_pole;
    const double radiation_surface = (a.R.rad_pole - a.R.rad_equator) * par_j + a.R.rad_pole;
#define AT(F, i) a.F[(size_t)(i) * n2 + q]
#define S(P, i) P[(i) * COLS + tx]
    for (int i = ty; i < im; i += LT) S(s_land, i) = (AT(flags, i) & FL_LAND) ? 1 : 0;
    __syncthreads();
    if (ty == 0) {  // albedo where the column has a sea or a land surface (:210-218); i_topography (:128-139)
        bool surface = !S(s_land, 0);
        int i_mount = 0;
        for (int i = 0; i < im - 1; i++) {
            if (S(s_land, i) && !S(s_land, i + 1)) surface = true;
            if (S(s_land, i)) i_mount = i;
        }
        s_mount[tx] = i_mount;
        if (valid) {
            if (surface) a.albedo[q] = (a.R.albedo_pole - a.R.albedo_equator) * par_j + a.R.albedo_pole;
            a.radiation_surface[q] = radiation_surface;
        }
    }
    __syncthreads();
    const int i_mount = s_mount[tx];
    for (int i = ty; i < im; i += LT) {  // :238-266
        double eps = AT(epsilon_3d, i), rad = AT(radiation_3d, i);
        const double t = AT(t, i);
        if (i <= i_trop) {
            double cc = AT(c, i), cl = AT(cloud, i), ic = AT(ice, i);
            if (cc < 0.) { cc = 0.; if (valid) AT(c, i) = 0.; }
            if (cl < 0.) { cl = 0.; if (valid) AT(cloud, i) = 0.; }
            if (ic < 0.) { ic = 0.; if (valid) AT(ice, i) = 0.; }
            const double e = (cc + cl + ic) * AT(p_stat, i) / a.ep;
            const double epsilon_eff = epsilon_eff_max - (a.R.epsilon_tropopause - epsilon_eff_max) * parabola((double)i / (im - 1));
            if (i >= i_mount) {
                eps = a.R.co2_coeff * epsilon_eff + .0416 * sqrt(e);
                rad = (1. - eps) * sigma * pow4(t * t_0);
            }
            if (eps > 1.) eps = 1.;
        }
        S(s_t, i) = t; S(s_eps, i) = eps; S(s_rad, i) = rad;
    }
    __syncthreads();
    if (ty == 0 && valid) a.epsilon[q] = S(s_eps, i_trop);
    const double eps_mount = S(s_eps, i_mount), rad_mount = S(s_rad, i_mount);
    __syncthreads();
    for (int i = ty; i < i_mount; i += LT) { S(s_eps, i) = eps_mount; S(s_rad, i) = rad_mount; }
    __syncthreads();
    const double eps_above = S(s_eps, i_trop);  // read after the fill: a mountain above the tropopause has just overwritten it
    const double rad_above = (1. - eps_above) * sigma * pow4(a.t_tropopause * t_0);
    __syncthreads();
    for (int i = ty; i < im; i += LT)
        if (i >= i_trop) { S(s_eps, i) = eps_above; S(s_t, i) = a.t_tropopause; S(s_rad, i) = rad_above; }
    __syncthreads();
    for (int iter_rad = 1; iter_rad <= 4; iter_rad++) {  // :293-399
        // t(i_trop) is t_tropopause and epsilon_3D(i_trop) is eps_above from here on, so radiation_3D(i_trop) as the
        // round first sets it (:308) is rad_above, operation for operation
        const double rad_trop = rad_above;
        for (int i = ty; i < im; i += LT) {
            const double eps = S(s_eps, i), t = S(s_t, i);
            if (i == i_mount) S(s_rad, i) = (1. - eps) * sigma * pow4(t * t_0) / rad_trop;
            else if (i > i_mount && i <= i_trop) {
                const double tmp = sigma * pow4(t * t_0) / rad_trop;
                S(s_rad, i) = (1. - eps) * tmp;
            } else if (i == i_trop) S(s_rad, i) = rad_trop;
        }
        __syncthreads();
        if (ty == 0) {
            const double radiation_back = S(s_eps, i_mount + 1) * sigma * pow4(S(s_t, i_mount + 1) * t_0);
            const double atmospheric_window = .1007 * radiation_surface;
            double rad_surf_diff = radiation_back + radiation_surface - atmospheric_window;
            const double fac_rad = (double)i_mount * .07 + 1.;
            rad_surf_diff = fac_rad * rad_surf_diff;
            double AA_cur = rad_surf_diff / rad_trop;
            const double CCC = 0., DDD = 0.;
            double alfa_prev = 0., beta_prev = 0., AA_prev = 0.;
            for (int i = i_mount; i < i_trop; i++) {
                if (i > i_mount) AA_cur = AA_prev * (1. - S(s_eps, i));
                double al, be;
                if (i == i_mount) {
                    const double bb = -S(s_rad, i), cc = S(s_rad, i + 1), dd = -AA_cur;
                    al = cc / bb; be = dd / bb;
                } else {
                    const double aa = S(s_rad, i - 1), bb = -2. * S(s_rad, i), cc = S(s_rad, i + 1);
                    const double dd = -AA_prev + AA_cur + CCC - DDD;
                    al = cc / (bb - aa * alfa_prev);
                    be = (dd - aa * beta_prev) / (bb - aa * alfa_prev);
                }
                S(s_al, i) = al; S(s_be, i) = be;
                alfa_prev = al; beta_prev = be; AA_prev = AA_cur;
            }
            double rad_next = rad_above;  // (1 - epsilon_3D(i_trop)) sigma (t_tropopause t_0)^4, :379-380
            S(s_rad, i_trop) = rad_next;
            for (int i = i_trop - 1; i >= i_mount; i--) {
                const double rad = -S(s_al, i) * rad_next + S(s_be, i);
                S(s_rad, i) = rad;
                rad_next = rad;
            }
        }
        __syncthreads();
        for (int i = ty; i < im; i += LT) {
            if (i >= i_mount && i < i_trop) S(s_t, i) = .5 * (S(s_t, i) + root4(S(s_rad, i) / sigma) / t_0);
            else if (i >= i_trop) { S(s_t, i) = a.t_tropopause; S(s_rad, i) = rad_above; }
        }
        __syncthreads();
    }
    if (valid)
        for (int i = ty; i < im; i += LT) { AT(t, i) = S(s_t, i); AT(epsilon_3d, i) = S(s_eps, i); AT(radiation_3d, i) = S(s_rad, i); }
#undef S
#undef AT
}

// BC_Bathymetry_Atmosphere::vegetationDistribution (atmosphere/BC_Bath_Atm.cpp:129-142)
__global__ void k_vegetation(const double* __restrict__ precipitation, const double* __restrict__ t,
                             const uint8_t* __restrict__ flags, double max_precipitation, size_t n2,
                             double* __restrict__ vegetation) {
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n2) return;
    const bool grows = max_precipitation > 0 && (flags[q] & FL_LAND) && !(t[q] < 1.);
    vegetation[q] = grows ? precipitation[q] / max_precipitation : 0.;
}

}  // namespace

extern "C" void atomdev_default_radiation(atomdev_radiation* r) {  // param.py:60-70, 123-130
    r->albedo_pole = .7; r->albedo_equator = .15; r->epsilon_pole = .59; r->epsilon_equator = .594; r->epsilon_tropopause = .001;
    r->rad_pole = 40.; r->rad_equator = 230.; r->co2_coeff = .98 * (330. / 320.); r->sigma = 5.670280e-8;
}

extern "C" int atomdev_atm_radiation_multi_layer(atomdev_t* h, const atomdev_radiation* r, const int* i_trop) {
    ATOMDEV_REQUIRE(h && h->model == ATOMDEV_ATM && r && i_trop, "atomdev_atm_radiation_multi_layer: bad argument");
    for (int j = 0; j < h->jm; j++)
        ATOMDEV_REQUIRE(i_trop[j] >= 0 && i_trop[j] <= h->im - 1, "atomdev_atm_radiation_multi_layer: tropopause level outside the grid");
    ATOMDEV_TRY(hipSetDevice(h->device));
    if (int rc = atomdev_refresh_flags(h)) return rc;
    const size_t need = (size_t)h->jm * sizeof(int);
    ATOMDEV_REQUIRE(need <= h->red_bytes, "atomdev_atm_radiation_multi_layer: scratch too small");
    ATOMDEV_TRY(hipMemcpyAsync(h->red, i_trop, need, hipMemcpyHostToDevice, h->stream));
    if (int rc__ = atomdev_sync_stream(h)) return rc__;  // the caller's array may change after the call
    RadArgs a;
    a.t = h->f[ATOMDEV_T]; a.c = h->f[ATOMDEV_C]; a.cloud = h->f[ATOMDEV_CLOUD]; a.ice = h->f[ATOMDEV_ICE];
    a.epsilon_3d = h->f[ATOMDEV_EPSILON_3D]; a.radiation_3d = h->f[ATOMDEV_RADIATION_3D];
    a.alfa = h->scr[0]; a.beta = h->scr[1];
    h->h0i_valid = false;  // scr[1] held the sweep's cached damping coefficients
    a.p_stat = h->f[ATOMDEV_P_STAT];
    a.albedo = h->g[ATOMDEV_ALBEDO]; a.epsilon = h->g[ATOMDEV_EPSILON]; a.radiation_surface = h->g[ATOMDEV_RADIATION_SURFACE];
    a.flags = h->flags; a.i_trop = (const int*)h->red;
    a.im = h->im; a.jm = h->jm; a.km = h->km;
    a.R = *r; a.t_0 = h->p.t_0; a.ep = h->p.ep; a.t_tropopause = h->p.t_tropopause;
    {
#if defined(RAD_COLS) && defined(RAD_LT)
        const int cols = RAD_COLS, lt = h->im < RAD_LT ? h->im : RAD_LT;
#else
        // 16 columns and about 11 threads per column: small workgroups, so that several share a CU and one's
        // recurrences (16 lanes busy) overlap the others' level-parallel phases
        const int cols = 16, passes = (h->im + 10) / 11, lt = (h->im + passes - 1) / passes;
#endif
        const size_t lds = (size_t)5 * h->im * cols * sizeof(double) + cols * sizeof(int) + (size_t)h->im * cols;
#ifdef ATOMDEV_RAD_COLUMNS
        if (false)
#else
        if (lds <= 64 * 1024)
#endif
            hipLaunchKernelGGL(k_atm_radiation_levels, dim3((unsigned)((h->n2 + cols - 1) / cols)), dim3(cols, lt), lds, h->stream, a);
        else  // very tall grids: one thread per column, work arrays in global memory
            hipLaunchKernelGGL(k_atm_radiation, dim3((unsigned)((h->n2 + 127) / 128)), dim3(128), 0, h->stream, a);
    }
    ATOMDEV_TRY(hipGetLastError());
    return 0;
}

extern "C" int atomdev_atm_latent_heat(atomdev_t* h) {
    ATOMDEV_REQUIRE(h && h->model == ATOMDEV_ATM, "atomdev_atm_latent_heat: not an atmosphere handle");
    ATOMDEV_TRY(hipSetDevice(h->device));
    if (int rc = atomdev_refresh_flags(h)) return rc;
    const atomdev_params& P = h->p;
    hipLaunchKernelGGL(k_atm_latent_heat, dim3((unsigned)((h->n2 + TPB - 1) / TPB)), dim3(TPB), 0, h->stream, h->f[ATOMDEV_T],
                       h->f[ATOMDEV_C], h->f[ATOMDEV_ICE], h->f[ATOMDEV_P_STAT], h->flags, h->f[ATOMDEV_Q_LATENT],
                       h->f[ATOMDEV_Q_SENSIBLE], h->im, h->n2, P.L_atm, P.dr, P.ep, P.t_0, P.lv, P.ls, P.cp_l, P.r_air,
                       P.R_WaterVapour, P.r_water_vapour);
    ATOMDEV_TRY(hipGetLastError());
    return 0;
}

extern "C" int atomdev_atm_vegetation_distribution(atomdev_t* h, double max_Precipitation) {
    ATOMDEV_REQUIRE(h && h->model == ATOMDEV_ATM, "atomdev_atm_vegetation_distribution: not an atmosphere handle");
    ATOMDEV_TRY(hipSetDevice(h->device));
    if (int rc = atomdev_refresh_flags(h)) return rc;
    hipLaunchKernelGGL(k_vegetation, dim3((unsigned)((h->n2 + TPB - 1) / TPB)), dim3(TPB), 0, h->stream,
                       h->g[ATOMDEV_PRECIPITATION], h->f[ATOMDEV_T], h->flags, max_Precipitation, h->n2,
                       h->g[ATOMDEV_VEGETATION]);
    ATOMDEV_TRY(hipGetLastError());
    return 0;
}

extern "C" int atomdev_atm_msl_data(atomdev_t* h, int RadiationModel) {
    ATOMDEV_REQUIRE(h && h->model == ATOMDEV_ATM, "atomdev_atm_msl_data: not an atmosphere handle");
    ATOMDEV_TRY(hipSetDevice(h->device));
    if (int rc = atomdev_refresh_flags(h)) return rc;
    const atomdev_params& P = h->p;
    MslArgs a;
    a.t = h->f[ATOMDEV_T]; a.v = h->f[ATOMDEV_V]; a.w = h->f[ATOMDEV_W];
    a.q_latent = h->f[ATOMDEV_Q_LATENT]; a.q_sensible = h->f[ATOMDEV_Q_SENSIBLE]; a.radiation_3d = h->f[ATOMDEV_RADIATION_3D];
    a.co2 = h->f[ATOMDEV_CO2];
    a.c = h->f[ATOMDEV_C]; a.cloud = h->f[ATOMDEV_CLOUD]; a.ice = h->f[ATOMDEV_ICE];
    a.p_rain = h->f[ATOMDEV_P_RAIN]; a.p_snow = h->f[ATOMDEV_P_SNOW]; a.p_stat = h->f[ATOMDEV_P_STAT];
    a.q_radiation = h->g[ATOMDEV_Q_RADIATION]; a.precipitable_water = h->g[ATOMDEV_PRECIPITABLE_WATER];
    a.evaporation_penman = h->g[ATOMDEV_EVAPORATION_PENMAN]; a.evaporation_dalton = h->g[ATOMDEV_EVAPORATION_DALTON];
    a.q_evaporation = h->g[ATOMDEV_Q_EVAPORATION]; a.q_latent_2d = h->g[ATOMDEV_Q_LATENT_2D];
    a.q_sensible_2d = h->g[ATOMDEV_Q_SENSIBLE_2D]; a.q_bottom = h->g[ATOMDEV_Q_BOTTOM];
    a.co2_total = h->g[ATOMDEV_CO2_TOTAL]; a.precipitation = h->g[ATOMDEV_PRECIPITATION];
    a.flags = h->flags; a.im = h->im; a.radiation_model = RadiationModel; a.n2 = h->n2;
    a.r_air = P.r_air; a.R_Air = P.R_Air; a.t_0 = P.t_0; a.g = P.g; a.L_atm = P.L_atm; a.ep = P.ep; a.hp = P.hp;
    a.cp_l = P.cp_l; a.lv = P.lv; a.u_0 = P.u_0; a.R_WaterVapour = P.R_WaterVapour;
    const unsigned nb = (unsigned)((h->n2 + TPB - 1) / TPB);
    hipLaunchKernelGGL(k_atm_msl_columns, dim3(nb), dim3(TPB), 0, h->stream, a);
    ATOMDEV_TRY(hipGetLastError());
    Clamp5 cl = { { a.c, a.cloud, a.ice, a.p_rain, a.p_snow } };
    hipLaunchKernelGGL(k_clamp_at_zero, dim3((unsigned)((h->n3 + TPB - 1) / TPB), 5), dim3(TPB), 0, h->stream, cl, h->n3);
    ATOMDEV_TRY(hipGetLastError());
    // boundary values of BuoyancyForce, r then theta then phi as in the reference (:248-278)
    hipLaunchKernelGGL(k_buoyancy_fill_r, dim3(nb), dim3(TPB), 0, h->stream, h->f[ATOMDEV_BUOYANCY], h->flags, h->im, h->jm,
                       h->km);
    ATOMDEV_TRY(hipGetLastError());
    if (int rc = atomdev_fill_theta_neumann(h, ATOMDEV_BUOYANCY)) return rc;
    return atomdev_fill_phi_seam(h, ATOMDEV_BUOYANCY, 1, h->jm - 1);
}

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
