// The atmosphere's start of a time slice (cAtmosphereModel.cpp:260-292): the initial conditions of
// atmosphere/BC_Thermo.cpp.  The surface files are host work (text); everything else runs on the device with one
// thread per (level, longitude) pair or per (theta, phi) column, 64 neighbouring longitudes per wave so that every
// row is read and written coalesced.  Arithmetic is +, -, *, / in the reference's order, so the results carry the
// reference's bits; only BC_Pressure (pow) and BC_WaterVapour (exp, sqrt) go through the device's libm.
#include <cmath>
#include <fstream>
#include <string>
#include <vector>

#include "atomdev_internal.h"
#include "cr_math.h"
#include "text_numbers.h"

namespace {

__host__ __device__ __forceinline__ double parabola(double x) { return x * x - 2 * x; }  // lib/Utils.h:41-43

// BC_Thermo's i_topography (BC_Thermo.cpp:128-139): the highest land level at or below im-2, else 0
__device__ __forceinline__ int topography_level(const uint8_t* __restrict__ flags, size_t n2, size_t q, int im) {
    for (int i = im - 2; i >= 0; i--)
        if (flags[(size_t)i * n2 + q] & FL_LAND) return i;
    return 0;
}

// ---- BC_Thermo::IC_CellStructure (BC_Thermo.cpp:934-2307) ----
//
// The routine is a list of statements over rows of one longitude; no statement combines different longitudes or
// different levels.  The list is kept as data, in the reference's order, and every thread runs it for its own
// (level, longitude) pair.
enum CellOpKind : int {
    CELL_U,       // u below and at the tropopause: -a (i^2 / i_half^2 - 2 i / i_half), rows j0..j1 (e.g. :1014-1024)
    CELL_LOW,     // below the tropopause: (a - b) i / i_trop + b (e.g. :1082-1094)
    CELL_HIGH,    // from the tropopause up: a (i - (im-1)) / (i_trop - (im-1)); guard: 0 none (:1330-1340),
                  // 1 a zero divisor becomes 1e-6 (:1098), 2 the divisor becomes 1e-6 where i_trop == 40 (:1044)
    CELL_LINE,    // x[j] = (x[ja] - x[jb]) (j - jb) / (ja - jb) + x[jb] for j = j0 .. j1 in steps of dir, in place (e.g. :1472-1484)
    CELL_FLIP,    // sign of v south of the equator (:2126-2133)
    CELL_SMOOTH,  // seven rows around j0, u v w (e.g. :2144-2157)
};
struct CellOp {
    int kind, field, j0, j1, dir, ja, jb, guard;
    double a, b;
};
constexpr int CELL_MAX_OPS = 96;
struct CellOps {
    int n;
    CellOp op[CELL_MAX_OPS];
};

struct CellBuilder {
    CellOps* t;
    void push(CellOp o) { t->op[t->n++] = o; }
    void u(int j0, int j1, double ua) { push({ CELL_U, ATOMDEV_U, j0, j1, 1, 0, 0, 0, ua, 0. }); }
    void low(int f, int j0, int j1, double x_trop, double x_sl) { push({ CELL_LOW, f, j0, j1, 1, 0, 0, 0, x_trop, x_sl }); }
    void high(int f, int j0, int j1, double x_trop, int guard) { push({ CELL_HIGH, f, j0, j1, 1, 0, 0, guard, x_trop, 0. }); }
    void line(int f, int j0, int j1, int dir, int ja, int jb) { push({ CELL_LINE, f, j0, j1, dir, ja, jb, 0, 0., 0. }); }
    // one anchor row with u, v, w profiles and the row of the v extremum beside it
    void cell(int j, double ua, double v_sl, double v_trop, double w_sl, double w_trop, int guard, int jv, double vv_sl, double vv_trop) {
        u(j, j, ua);
        low(ATOMDEV_V, j, j, v_trop, v_sl); low(ATOMDEV_W, j, j, w_trop, w_sl);
        high(ATOMDEV_V, j, j, v_trop, guard); high(ATOMDEV_W, j, j, w_trop, guard);
        low(ATOMDEV_V, jv, jv, vv_trop, vv_sl); high(ATOMDEV_V, jv, jv, vv_trop, 1);
    }
};

void build_cell_ops(CellOps* t, int jm) {
    t->n = 0;
    CellBuilder b{ t };
    const int U = ATOMDEV_U, V = ATOMDEV_V, W = ATOMDEV_W;
    // rows of the 1-degree grid (:987-999) and the profile values (:944-984)
    const int aeq = 90, pol_n = 0, pol_s = jm - 1, pol_v_n = 15, pol_v_s = 165, fer_n = 30, fer_s = 150, fer_v_n = 45, fer_v_s = 135,
              had_n = 60, had_s = 120, had_v_n = 75, had_v_s = 105;
    // equator (:1014-1052)
    b.u(aeq, aeq, 1.);
    b.low(V, aeq, aeq, 0., 0.); b.low(W, aeq, aeq, -7.5, -1.);
    b.high(V, aeq, aeq, 0., 2); b.high(W, aeq, aeq, -7.5, 2);
    // northern polar cell (:1068-1133): v and w on every row from the pole to 60 N
    b.u(0, pol_n, -.5);
    b.low(V, pol_n, fer_n, 0., 0.); b.low(W, pol_n, fer_n, 0., -.01);
    b.high(V, pol_n, fer_n, 0., 1); b.high(W, pol_n, fer_n, 0., 1);
    b.low(V, pol_v_n, pol_v_n, -1., .5); b.high(V, pol_v_n, pol_v_n, -1., 1);
    // northern Ferrel and Hadley cells (:1143-1283), southern Hadley and Ferrel cells (:1302-1440, no guard at 30 S and 60 S)
    b.cell(fer_n, .5, .5, 1., -.2, 10., 1, fer_v_n, -.1, 1.);
    b.cell(had_n, -1., .25, -1., 1., 30., 1, had_v_n, 1., -1.);
    b.cell(had_s, -1., .25, -1., 1., 30., 0, had_v_s, 1., -1.);
    b.cell(fer_s, .5, .5, 1., -.2, 10., 0, fer_v_s, -.1, 1.);
    // southern polar cell (no guard between 60 S and the pole)
    b.u(pol_s, jm - 1, -.5);
    b.low(V, fer_s + 1, pol_s, 0., 0.); b.low(W, fer_s + 1, pol_s, 0., -.01);
    b.high(V, fer_s + 1, pol_s, 0., 0); b.high(W, fer_s + 1, pol_s, 0., 0);
    b.low(V, pol_v_s, pol_v_s, -1., .5); b.high(V, pol_v_s, pol_v_s, -1., 1);
    // linear connections between the rows (:1468-2120), in the reference's order: a loop rewrites its own end rows
    // and the loops after it read the rewritten values
    b.line(U, pol_n, fer_n, 1, fer_n, pol_n); b.line(W, pol_n, fer_n, 1, fer_n, pol_n);
    b.line(V, pol_n, pol_v_n, 1, pol_v_n, pol_n); b.line(V, pol_v_n, fer_n, 1, fer_n, pol_v_n);
    b.line(U, fer_n + 1, had_n, 1, had_n, fer_n); b.line(W, fer_n + 1, had_n, 1, had_n, fer_n);
    b.line(V, fer_n, fer_v_n, 1, fer_v_n, fer_n); b.line(V, fer_v_n, had_n, 1, had_n, fer_v_n);
    b.line(V, fer_v_s, had_s, -1, fer_v_s, had_s);
    b.line(U, had_n, aeq, 1, aeq, had_n); b.line(W, had_n, aeq, 1, aeq, had_n);
    b.line(V, had_v_n, aeq, 1, aeq, had_v_n); b.line(V, had_n, had_v_n, 1, had_v_n, had_n);
    b.line(U, fer_s, pol_s, 1, fer_s, pol_s); b.line(W, fer_s, pol_s, 1, fer_s, pol_s);
    b.line(V, pol_s, pol_v_s, -1, pol_s, pol_v_s); b.line(V, pol_v_s, fer_s, -1, pol_v_s, fer_s);
    b.line(U, had_s, fer_s, 1, had_s, fer_s); b.line(W, had_s, fer_s, 1, had_s, fer_s);
    b.line(V, fer_s, fer_v_s, -1, fer_s, fer_v_s);
    b.line(U, aeq, had_s, 1, aeq, had_s); b.line(W, aeq, had_s, 1, aeq, had_s);
    b.line(V, had_s, had_v_s, -1, had_s, had_v_s); b.line(V, had_v_s, aeq, -1, had_v_s, aeq);
    b.push({ CELL_FLIP, V, aeq + 1, jm - 1, 1, 0, 0, 0, 0., 0. });
    for (int jc : { had_n, fer_n, had_v_n, fer_v_n, aeq, had_s, fer_s, had_v_s, fer_v_s })  // :2144-2283
        b.push({ CELL_SMOOTH, U, jc, jc, 1, 0, 0, 0, 0., 0. });
}

struct FieldsUVW {
    double* f[3];  // u, v, w
};
__device__ __forceinline__ double* cell_field(const FieldsUVW& F, int field) {
    return field == ATOMDEV_U ? F.f[0] : (field == ATOMDEV_V ? F.f[1] : F.f[2]);
}

__global__ __launch_bounds__(64) void k_atm_cell_structure(FieldsUVW F, const uint8_t* __restrict__ flags, const int* __restrict__ trop,
                                                           const CellOps* __restrict__ ops, int im, int jm, int km, double u_0) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
    if (k >= km) return;
    const size_t base = (size_t)i * jm * km + k, row = (size_t)km;
    const double d_i = (double)i;
    const int n = ops->n;
    for (int m = 0; m < n; m++) {
        const CellOp o = ops->op[m];
        double* x = cell_field(F, o.field) + base;
        switch (o.kind) {
        case CELL_U:
            for (int j = o.j0; j <= o.j1; j++) {
                const int i_max = trop[j];
                const double d_i_half = (double)(i_max / 2);
                if (i <= i_max) x[j * row] = -o.a * (d_i * d_i / (d_i_half * d_i_half) - 2. * d_i / d_i_half);
            }
            break;
        case CELL_LOW:
            for (int j = o.j0; j <= o.j1; j++) {
                const int i_max = trop[j];
                if (i < i_max) x[j * row] = (o.a - o.b) * d_i / (double)i_max + o.b;
            }
            break;
        case CELL_HIGH:
            for (int j = o.j0; j <= o.j1; j++) {
                const int i_max = trop[j];
                double d_i_max = (double)(i_max - (im - 1));
                if (o.guard == 1 && d_i_max == 0.) d_i_max = 1.e-6;
                if (o.guard == 2 && i_max == 40) d_i_max = 1.e-6;
                if (i >= i_max) x[j * row] = o.a * (double)(i - (im - 1)) / d_i_max;
            }
            break;
        case CELL_LINE: {
            const double d_b = (double)o.jb, d_diff = (double)o.ja - (double)o.jb;
            double xa = x[o.ja * row], xb = x[o.jb * row];
            for (int j = o.j0; o.dir > 0 ? j <= o.j1 : j >= o.j1; j += o.dir) {
                const double v = (xa - xb) * ((double)j - d_b) / d_diff + xb;
                x[j * row] = v;
                if (j == o.ja) xa = v;
                if (j == o.jb) xb = v;
            }
            break;
        }
        case CELL_FLIP:
            for (int j = o.j0; j <= o.j1; j++) x[j * row] = -x[j * row];
            break;
        case CELL_SMOOTH: {
            const int j_s = o.j0 - 3, j_n = o.j0 + 3;
            for (int c = 0; c < 3; c++) {  // the three components do not read each other
                double* y = F.f[c] + base;
                double ys = y[j_s * row], yn = y[j_n * row];
                for (int j = j_s; j <= j_n; j++) {
                    const double v = (yn - ys) / (double)(j_n - j_s) * (double)(j - j_s) + ys;
                    y[j * row] = v;
                    if (j == j_s) ys = v;
                    if (j == j_n) yn = v;
                }
            }
            break;
        }
        }
    }
    // :2288-2303
    for (int j = 0; j < jm; j++) {
        const bool land = flags[base + j * row] & FL_LAND;
        for (int c = 0; c < 3; c++) {
            double* y = F.f[c] + base;
            y[j * row] = land ? 0. : y[j * row] / u_0;
        }
    }
}

// ---- BC_Thermo::BC_Temperature (:427-676) ----

struct SunProfile {  // :485-536
    double aa, bb, cc, j_d, dd, e, t_360, t_cretaceous_add;
    int k_par, k_pol;
};

// t on the sea level of the meridian k = 20 as the first loop of :505-515 leaves it: row j of `line`
__global__ void k_atm_sun_meridian(const double* __restrict__ t, double* __restrict__ line, SunProfile S, int jm, int km) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= jm) return;
    const double d_j = (double)j;
    double v = t[(size_t)j * km + 20];
    if (d_j <= S.j_d) v = S.dd * d_j + S.e + S.t_cretaceous_add;
    if (d_j > S.j_d) v = S.aa * d_j * d_j + S.bb * d_j + S.cc + S.t_cretaceous_add;
    line[j] = v;
}
// :517-535: the parabola along every row, anchored at the meridian's value
__global__ void k_atm_sun_rows(double* __restrict__ t, const double* __restrict__ line, SunProfile S, int jm, int km) {
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= (size_t)jm * km) return;
    const int j = (int)(q / km), k = (int)(q % km);
    const double jm_temp_asym = line[j];
    const double k_par_f = (double)S.k_par, k_pol_f = (double)S.k_pol, d_k = (double)k;
    const double aa = (jm_temp_asym - S.t_360) / (((k_par_f * k_par_f) - (k_pol_f * k_pol_f)) - 2. * k_par_f * (k_par_f - k_pol_f));
    const double bb = -2. * aa * k_par_f;
    const double cc = jm_temp_asym + aa * k_par_f * k_par_f;
    t[q] = aa * d_k * d_k + bb * d_k + cc;
}

struct TemperatureStart {
    int Ma, RadiationModel, NASATemperature;
    double t_eff, t_pole, t_land, t_cretaceous_add, t_pole_diff_ocean, t_pole_diff_land, t_0;
    double t_tropopause_pole, t_eff_tropo;
    int trop_add;
};

__global__ __launch_bounds__(256) void k_atm_bc_temperature(double* __restrict__ t, const uint8_t* __restrict__ flags, const int* __restrict__ trop,
                                                            TemperatureStart S, int im, int jm, int km) {
    const size_t n2 = (size_t)jm * km;
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n2) return;
    const int j = (int)(q / km);
    const int i_mount = topography_level(flags, n2, q, im);
    const bool land0 = flags[q] & FL_LAND;
    double* tm = t + (size_t)i_mount * n2 + q;
    const double d_j = (double)j, d_j_half = (double)(jm - 1) / 2.0;
    if (S.RadiationModel == 1) {  // :568-633
        if (S.NASATemperature == 0) {
            *tm = S.t_eff * parabola(d_j / d_j_half) + S.t_pole + S.t_cretaceous_add;
            if (land0) *tm = S.t_eff * parabola(d_j / d_j_half) + S.t_pole + S.t_cretaceous_add + S.t_land;
        } else if (land0) {
            double v = S.t_eff * parabola(d_j / d_j_half) + S.t_pole + S.t_cretaceous_add + S.t_land;
            v += S.t_pole_diff_land * fabs(parabola(d_j / d_j_half) + 1.) / S.t_0;
            *tm = v;
        } else if (S.Ma > 0) {
            t[q] += S.t_cretaceous_add + S.t_pole_diff_ocean * fabs(parabola(d_j / d_j_half) + 1.) / S.t_0;
        }
    }
    // :635-664: linear up to the tropopause, constant above
    const int i_trop = trop[j] + S.trop_add;
    const double temp_tropopause = S.t_eff_tropo * parabola(j / ((jm - 1) / 2.0)) + S.t_tropopause_pole + S.t_cretaceous_add;
    double tmp = *tm;
    tmp = (temp_tropopause - tmp) * ((double)i_mount / i_trop) + tmp;
    for (int i = 0; i < im; i++) {
        double v = temp_tropopause;
        if (i < i_trop + 1) v = i > i_mount ? (temp_tropopause - tmp) * ((double)(i - i_mount) / i_trop) + tmp : tmp;
        t[(size_t)i * n2 + q] = v;
    }
}

// ---- BC_Thermo::IC_Temperature_WestEastCoast (:3151-3233) ----
//
// Four scans over the sea level, each over lines that do not touch each other: one thread per line.  Within a line
// the statements run in the reference's order on the stored values.  The flags j_sequel / k_sequel of the second and
// third scan are not reset between lines; what a line inherits is whether the last cell of the line before it was
// fluid (a fluid cell always leaves the flag at 1, a land cell at 0), which the mask alone tells.
constexpr int COAST = 7;

__global__ void k_coast_north_south(double* __restrict__ t, const uint8_t* __restrict__ flags, int jm, int km, int south) {
    const int k = 1 + blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= km - 1) return;
    auto T0 = [&](int j) -> double& { return t[(size_t)j * km + k]; };
    auto land = [&](int j, int kk) { return (bool)(flags[(size_t)j * km + kk] & FL_LAND); };
    if (!south) {  // :3163-3180
        int flip = 0;
        for (int j = COAST; j < jm - 1; j++) {
            const int j_air = land(j, k) ? 1 : 0;
            if (!j_air) flip = 0;
            if (flip == 0 && j_air == 1) {
                const int ll = j - COAST;
                for (int l = ll; l < j; l++) T0(l) = T0(ll);
                flip = 1;
            }
        }
    } else {  // :3183-3200
        if (jm - COAST <= 0) return;
        int j_sequel = k == 1 ? 1 : (land(jm - COAST - 1, k - 1) ? 0 : 1);
        for (int j = 0; j < jm - COAST; j++) {
            if (land(j, k)) j_sequel = 0;
            else if (j_sequel == 0) {
                const int ll = j + COAST;
                for (int l = ll; l > j; l--) T0(l) = T0(ll);
                j_sequel = 1;
            }
        }
    }
}

__global__ void k_coast_east_west(double* __restrict__ t, const uint8_t* __restrict__ flags, int jm, int km, int west) {
    const int j = 1 + blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= jm - 1) return;
    auto T0 = [&](int k) -> double& { return t[(size_t)j * km + k]; };
    auto land = [&](int jj, int k) { return (bool)(flags[(size_t)jj * km + k] & FL_LAND); };
    if (!west) {  // :3203-3218
        if (km - COAST <= COAST) return;
        int k_sequel = j == 1 ? 1 : (land(j - 1, km - COAST - 1) ? 0 : 1);
        for (int k = COAST; k < km - COAST; k++) {
            if (land(j, k)) k_sequel = 0;
            else if (k_sequel == 0) {
                const int ll = k + COAST;
                for (int l = ll; l > k; l--) T0(l) = T0(ll);
                k_sequel = 1;
            }
        }
    } else {  // :3221-3237
        int flip = 0;
        for (int k = COAST; k < km - 1; k++) {
            const int k_air = land(j, k) ? 1 : 0;
            if (!k_air) flip = 0;
            if (flip == 0 && k_air == 1) {
                const int ll = k - COAST;
                for (int l = ll; l < k; l++) T0(l) = T0(ll);
                flip = 1;
            }
        }
    }
}

// ---- BC_Thermo::BC_Pressure (:2387-2407) ----
__global__ __launch_bounds__(256) void k_atm_bc_pressure(double* __restrict__ p_stat, const double* __restrict__ t, int im, size_t n2, double t_0,
                                                         double r_air, double R_Air, double gam, double L_atm, double exp_pressure) {
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n2) return;
    const double t0 = t[q];
    const double p0 = .01 * (r_air * R_Air * t0 * t_0);
    p_stat[q] = p0;
    for (int i = 1; i < im; i++) {
        const double hight = (double)i * (L_atm / (double)(im - 1));
        p_stat[(size_t)i * n2 + q] = cr_pow(((t0 * t_0 - gam * hight * 1.e-2) / (t0 * t_0)), exp_pressure) * p0;
    }
}

// ---- BC_Thermo::BC_WaterVapour (:679-780) ----

// Results_MSL_Atm::C_Dalton as BC_Thermo::C_Dalton repeats it (:3335-3344)
__device__ __forceinline__ double c_dalton(double u_0, double v, double w) {
    const double C_max = -.053, v_max = 10.;
    const double vel_magnitude = sqrt(v * v + w * w) * u_0;
    return sqrt(C_max * C_max / v_max * vel_magnitude);
}

struct VapourStart {
    double t_0, hp, ep, r_air, R_Air, R_WaterVapour, u_0, dt_dim, dr_dim, c_ocean, c_land, c_tropopause;
    int trop_add;
};

// The value on the topography without the evaporation term of a land column, and every ocean column's evaporation
// term.
__global__ __launch_bounds__(256) void k_vapour_surface(double* __restrict__ c, double* __restrict__ p_stat, const double* __restrict__ t,
                                                        const double* __restrict__ v, const double* __restrict__ w, double* __restrict__ dalton,
                                                        const uint8_t* __restrict__ flags, VapourStart S, int im, int jm, int km) {
    const size_t n2 = (size_t)jm * km;
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n2) return;
    const size_t p = (size_t)topography_level(flags, n2, q, im) * n2 + q;
    const double t_m = t[p], t_0 = S.t_0;
    double cm = S.hp * S.ep * cr_exp(17.0809 * (t_m * t_0 - t_0) / (234.175 + (t_m * t_0 - t_0))) / ((S.r_air * S.R_Air * t_m * t_0) * .01);
    if (!(flags[q] & FL_LAND)) {
        cm = S.c_ocean * cm;
        const double ps = (S.r_air * S.R_Air * t_m * t_0) * .01;
        p_stat[p] = ps;
        const double t_u = t_m * t_0;
        const double r_dry = 100. * ps / (S.R_Air * t_m * t_0);
        const double r_humid = r_dry / (1. + (S.R_WaterVapour / S.R_Air - 1.) * cm);
        const double e = cm * ps / S.ep;
        const double E = S.hp * cr_exp(17.2694 * (t_u - 273.15) / (t_u - 35.86));
        const double sat_difference = (E - e);
        const double ev = 8.46e-4 * c_dalton(S.u_0, v[p], w[p]) * sat_difference * S.dt_dim / (r_humid * S.dr_dim) * 24.;
        dalton[q] = ev;
        cm = ev + cm;
    } else {
        cm = S.c_land * cm;
    }
    c[p] = cm;
}

// A land column adds the evaporation term of the ocean column visited last before it in the reference's order
// (longitude outer, latitude inner; :744), 0 before the first one.  Then the parabola up to the tropopause (:757-779):
// the loop rewrites the topography's value on its way up and goes on with the rewritten value.
__global__ __launch_bounds__(256) void k_vapour_columns(double* __restrict__ c, const double* __restrict__ dalton, const uint8_t* __restrict__ flags,
                                                        const int* __restrict__ trop, VapourStart S, int im, int jm, int km) {
    const size_t n2 = (size_t)jm * km;
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n2) return;
    const int j = (int)(q / km), k = (int)(q % km);
    const int i_mount = topography_level(flags, n2, q, im);
    double cm = c[(size_t)i_mount * n2 + q];
    if (flags[q] & FL_LAND) {
        double ev = 0.;
        for (long s = (long)k * jm + j - 1; s >= 0; s--) {
            const size_t at = (size_t)(s % jm) * km + (size_t)(s / jm);
            if (!(flags[at] & FL_LAND)) { ev = dalton[at]; break; }
        }
        cm = ev + cm;
    }
    const int i_trop = trop[j] + S.trop_add;
    const double d_i_max = (double)i_trop;
    for (int i = 0; i < im; i++) {
        double val = S.c_tropopause;
        if (i < i_trop) {
            const double d_i = (double)i;
            val = cm - (S.c_tropopause - cm) * (d_i / d_i_max * (d_i / d_i_max - 2.));
        }
        c[(size_t)i * n2 + q] = val;
        if (i == i_mount) cm = val;
    }
}

// ---- BC_Thermo::BC_CO2 (:788-881) ----
struct Co2Start {
    double co2_eff, co2_pole, co2_cretaceous, co2_ocean, co2_land, co2_vegetation, co2_tropopause;
    int trop_add;
};

__global__ __launch_bounds__(256) void k_atm_bc_co2(double* __restrict__ co2, const double* __restrict__ vegetation, const uint8_t* __restrict__ flags,
                                                    const int* __restrict__ trop, Co2Start S, int im, int jm, int km) {
    const size_t n2 = (size_t)jm * km;
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n2) return;
    const int j = (int)(q / km);
    const int i_mount = topography_level(flags, n2, q, im);
    const double d_j = (double)j, d_j_half = (double)((jm - 1) / 2);
    const double lat = S.co2_eff * (d_j * d_j / (d_j_half * d_j_half) - 2. * d_j / d_j_half) + S.co2_pole + S.co2_cretaceous;
    double cm = (flags[(size_t)i_mount * n2 + q] & FL_LAND) ? lat + S.co2_land - S.co2_vegetation * vegetation[q] : lat + S.co2_ocean;
    if (i_mount == 0) co2[q] = cm;  // the loop below starts at level 1
    const int i_trop = trop[j] + S.trop_add;
    const double d_i_max = (double)i_trop;
    for (int i = 1; i <= im - 1; i++) {
        double val = S.co2_tropopause;
        if (i <= i_trop) {
            const double d_i = (double)i;
            val = cm - (S.co2_tropopause - cm) * (d_i / d_i_max * (d_i / d_i_max - 2.));
        }
        co2[(size_t)i * n2 + q] = val;
        if (i == i_mount) cm = val;
    }
    // :868-878: the inside of the topography takes the value of its top
    for (int i = i_trop - 1; i >= 0; i--) {
        if (i + 1 >= im) continue;
        if ((flags[(size_t)i * n2 + q] & FL_LAND) && (flags[(size_t)(i + 1) * n2 + q] & FL_LAND)) co2[(size_t)i * n2 + q] = cm;
    }
}

int tropopause_add(double t_cretaceous, double t_0) {  // BC_Thermo::GetTropopauseHightAdd (:3298-3302) as its callers use it
    return (int)std::round(((t_cretaceous / t_0) * t_0) / 2.6);
}

// BC_Thermo::GetPoleTemperature (:3305-3332) with the table of :544-561
double pole_temperature(int Ma) {
    static const int M[16] = { 0, 40, 45, 50, 55, 60, 70, 80, 90, 100, 120, 130, 140, 150, 160, 340 };
    static const double T[16] = { 0., 22., 23.5, 24.1, 24.3, 22.4, 24.2, 23.7, 22.8, 21.8, 19., 17.8, 16.9, 16.4, 16., 16. };
    if (Ma <= M[0]) return T[0];
    if (Ma > M[15]) return T[15];
    int up = 0, bottom = 1;
    for (int n = 0; n < 16; n++) {
        if (M[n] >= Ma) { bottom = n; break; }
        up = n;
    }
    return (T[bottom] - T[up]) / (double)(M[bottom] - M[up]) * (double)(Ma - M[up]) + T[up];
}

// The jm tropopause levels on the device, in the reduction scratch; every level together with `add` must stay on the grid.
int upload_tropopause(atomdev* h, const int* im_tropopause, int add, const char* who, int** dev) {
    ATOMDEV_REQUIRE(im_tropopause, std::string(who) + ": im_tropopause is NULL");
    ATOMDEV_REQUIRE((size_t)h->jm * sizeof(int) <= h->red_bytes, std::string(who) + ": more rows than the scratch buffer holds");
    for (int j = 0; j < h->jm; j++)
        ATOMDEV_REQUIRE(im_tropopause[j] + add >= 1 && im_tropopause[j] + add <= h->im - 1,
                        std::string(who) + ": a tropopause level (with the paleo shift) outside 1 .. im-1");
    ATOMDEV_TRY(hipMemcpyAsync(h->red, im_tropopause, (size_t)h->jm * sizeof(int), hipMemcpyHostToDevice, h->stream));
    if (int rc__ = atomdev_sync_stream(h)) return rc__;  // the caller's array may go out of scope
    *dev = (int*)h->red;
    return 0;
}

int read_lat_lon_value(const char* path, size_t n, std::vector<double>& values, const char* what) {
    bool opened = false;
    if (text_third_of_triples(path, n, values, &opened)) return 0;
    ATOMDEV_REQUIRE(opened, std::string("could not open ") + what + " file at " + path);
    // short or malformed file: the stream's own treatment of it; entries the stream cannot deliver are 0
    std::ifstream file(path);
    values.assign(n, 0.);
    for (size_t m = 0; m < n && !file.eof(); m++) {
        double lat = 0., lon = 0., value = 0.;
        file >> lat >> lon >> value;
        values[m] = value;
    }
    return 0;
}

}  // namespace

extern "C" void atomdev_default_atm_start(atomdev_atm_start* s) {
    s->Ma = 0; s->first_time_slice = 1; s->Ma_previous = 0;
    s->sun = 0; s->RadiationModel = 1; s->NASATemperature = 1; s->declination = 0; s->sun_position_lat = 60; s->sun_position_lon = 180;
    s->t_cretaceous = 0.; s->t_cretaceous_add = 0.;
    s->t_average = 15.; s->t_equator = 1.0842; s->t_pole = .945; s->t_land = 0.;
    s->c_tropopause = .001; s->c_ocean = .58; s->c_land = .64;
    s->co2_average = 280.; s->co2_equator = 330.; s->co2_tropopause = 320.; s->co2_pole = 320.; s->co2_cretaceous = 330.;
    s->co2_vegetation = 3.; s->co2_ocean = 0.; s->co2_land = 0.;
    s->im_tropopause = nullptr;
}

extern "C" double atomdev_atm_co2_coeff(double CO2, double co2_factor, double co2_equator, double co2_tropopause, double co2_0) {
    if (CO2 == 1.) return co2_factor * ((co2_equator / co2_0) / (co2_tropopause / co2_0));
    return 1.;
}

extern "C" int atomdev_atm_ic_cell_structure(atomdev_t* h, const int* im_tropopause) {
    ATOMDEV_REQUIRE(h && h->model == ATOMDEV_ATM, "atomdev_atm_ic_cell_structure: needs an atmosphere handle");
    ATOMDEV_REQUIRE(h->jm == 181, "atomdev_atm_ic_cell_structure: the cell borders are rows of the 1-degree grid, jm must be 181");
    ATOMDEV_TRY(hipSetDevice(h->device));
    if (int rc = atomdev_refresh_flags(h)) return rc;
    int* trop = nullptr;
    ATOMDEV_REQUIRE(im_tropopause, "atomdev_atm_ic_cell_structure: im_tropopause is NULL");
    for (int j = 0; j < h->jm; j++)
        ATOMDEV_REQUIRE(im_tropopause[j] >= 0 && im_tropopause[j] <= h->im - 1, "atomdev_atm_ic_cell_structure: a tropopause level outside 0 .. im-1");
    static_assert(sizeof(CellOps) % 8 == 0, "the tropopause levels follow the table");
    ATOMDEV_REQUIRE(sizeof(CellOps) + (size_t)h->jm * sizeof(int) <= h->red_bytes, "atomdev_atm_ic_cell_structure: scratch buffer too small");
    CellOps table;
    build_cell_ops(&table, h->jm);
    ATOMDEV_TRY(hipMemcpyAsync(h->red, &table, sizeof(table), hipMemcpyHostToDevice, h->stream));
    trop = (int*)((char*)h->red + sizeof(CellOps));
    ATOMDEV_TRY(hipMemcpyAsync(trop, im_tropopause, (size_t)h->jm * sizeof(int), hipMemcpyHostToDevice, h->stream));
    if (int rc__ = atomdev_sync_stream(h)) return rc__;  // both sources leave scope
    FieldsUVW F{ { h->f[ATOMDEV_U], h->f[ATOMDEV_V], h->f[ATOMDEV_W] } };
    hipLaunchKernelGGL(k_atm_cell_structure, dim3((h->km + 63) / 64, h->im), dim3(64), 0, h->stream, F, h->flags, trop, (const CellOps*)h->red,
                       h->im, h->jm, h->km, h->p.u_0);
    ATOMDEV_TRY(hipGetLastError());
    return 0;
}

extern "C" int atomdev_atm_read_surface_temperature(atomdev_t* h, const char* path) {
    ATOMDEV_REQUIRE(h && path && h->model == ATOMDEV_ATM, "atomdev_atm_read_surface_temperature: needs an atmosphere handle and a path");
    ATOMDEV_REQUIRE(h->km >= 3, "atomdev_atm_read_surface_temperature: needs three longitudes");
    ATOMDEV_TRY(hipSetDevice(h->device));
    const int jm = h->jm, km = h->km, k_half = (km - 1) / 2;
    std::vector<double> values, plane(h->n2);
    if (int rc = read_lat_lon_value(path, h->n2, values, "SurfaceTemperature_File")) return rc;
    for (int k = 0; k < km; k++)
        for (int j = 0; j < jm; j++) plane[(size_t)j * km + k] = (values[(size_t)k * jm + j] + h->p.t_0) / h->p.t_0;
    for (int j = 0; j < jm; j++)  // :2343-2348, the meridian at 180 E
        plane[(size_t)j * km + k_half] = (plane[(size_t)j * km + k_half + 1] + plane[(size_t)j * km + k_half - 1]) / 2.;
    ATOMDEV_TRY(hipMemcpyAsync(h->f[ATOMDEV_T], plane.data(), h->n2 * 8, hipMemcpyHostToDevice, h->stream));
    ATOMDEV_TRY(hipMemcpyAsync(h->g[ATOMDEV_TEMPERATURE_NASA], plane.data(), h->n2 * 8, hipMemcpyHostToDevice, h->stream));
    if (int rc__ = atomdev_sync_stream(h)) return rc__;
    return 0;
}

extern "C" int atomdev_atm_read_surface_precipitation(atomdev_t* h, const char* path) {
    ATOMDEV_REQUIRE(h && path && h->model == ATOMDEV_ATM, "atomdev_atm_read_surface_precipitation: needs an atmosphere handle and a path");
    ATOMDEV_TRY(hipSetDevice(h->device));
    const int jm = h->jm, km = h->km;
    std::vector<double> values, plane(h->n2);
    if (int rc = read_lat_lon_value(path, h->n2, values, "SurfacePrecipitation_File")) return rc;
    for (int k = 0; k < km; k++)
        for (int j = 0; j < jm; j++) plane[(size_t)j * km + k] = values[(size_t)k * jm + j];
    ATOMDEV_TRY(hipMemcpyAsync(h->g[ATOMDEV_PRECIPITATION_NASA], plane.data(), h->n2 * 8, hipMemcpyHostToDevice, h->stream));
    if (int rc__ = atomdev_sync_stream(h)) return rc__;
    return 0;
}

extern "C" int atomdev_atm_bc_temperature(atomdev_t* h, const atomdev_atm_start* s) {
    ATOMDEV_REQUIRE(h && s && h->model == ATOMDEV_ATM, "atomdev_atm_bc_temperature: needs an atmosphere handle and the start values");
    ATOMDEV_TRY(hipSetDevice(h->device));
    if (int rc = atomdev_refresh_flags(h)) return rc;
    const int im = h->im, jm = h->jm, km = h->km;
    const double t_0 = h->p.t_0;
    TemperatureStart S;
    S.trop_add = tropopause_add(s->t_cretaceous, t_0);
    int* trop = nullptr;
    if (int rc = upload_tropopause(h, s->im_tropopause, S.trop_add, "atomdev_atm_bc_temperature", &trop)) return rc;
    const unsigned nb = (unsigned)((h->n2 + 255) / 256);
    if (s->Ma > 0 && s->sun == 1) {  // :485-536
        ATOMDEV_REQUIRE(km > 20, "atomdev_atm_bc_temperature: the sun-position profile reads the meridian k = 20");
        SunProfile P;
        const double j_par_f = (double)(s->sun_position_lat + s->declination), j_pol_f = (double)(jm - 1);
        P.aa = (s->t_equator - s->t_pole) / (((j_par_f * j_par_f) - (j_pol_f * j_pol_f)) - 2. * j_par_f * (j_par_f - j_pol_f));
        P.bb = -2. * P.aa * j_par_f;
        P.cc = s->t_equator + P.aa * j_par_f * j_par_f;
        P.j_d = std::sqrt((P.cc - s->t_pole) / P.aa);
        P.dd = 2. * P.aa * P.j_d + P.bb;
        P.e = s->t_pole;
        P.t_360 = (t_0 + 5.) / t_0;
        P.t_cretaceous_add = s->t_cretaceous_add;
        P.k_par = s->sun_position_lon; P.k_pol = km - 1;
        hipLaunchKernelGGL(k_atm_sun_meridian, dim3((jm + 63) / 64), dim3(64), 0, h->stream, h->f[ATOMDEV_T], h->plane, P, jm, km);
        hipLaunchKernelGGL(k_atm_sun_rows, dim3(nb), dim3(256), 0, h->stream, h->f[ATOMDEV_T], h->plane, P, jm, km);
    }
    S.Ma = s->Ma; S.RadiationModel = s->RadiationModel; S.NASATemperature = s->NASATemperature;
    S.t_eff = s->t_pole - s->t_equator; S.t_pole = s->t_pole; S.t_land = s->t_land; S.t_cretaceous_add = s->t_cretaceous_add; S.t_0 = t_0;
    S.t_pole_diff_ocean = s->first_time_slice ? 0. : pole_temperature(s->Ma) - pole_temperature(s->Ma_previous);
    S.t_pole_diff_land = pole_temperature(s->Ma);
    double t_tropopause_pole = -4.;  // :635-641
    t_tropopause_pole = h->p.t_tropopause + t_tropopause_pole / t_0;
    S.t_tropopause_pole = t_tropopause_pole;
    S.t_eff_tropo = t_tropopause_pole - h->p.t_tropopause;
    hipLaunchKernelGGL(k_atm_bc_temperature, dim3(nb), dim3(256), 0, h->stream, h->f[ATOMDEV_T], h->flags, trop, S, im, jm, km);
    ATOMDEV_TRY(hipGetLastError());
    return 0;
}

extern "C" int atomdev_atm_ic_temperature_west_east_coast(atomdev_t* h) {
    ATOMDEV_REQUIRE(h && h->model == ATOMDEV_ATM, "atomdev_atm_ic_temperature_west_east_coast: needs an atmosphere handle");
    ATOMDEV_REQUIRE(h->jm > COAST + 1 && h->km > COAST + 1, "atomdev_atm_ic_temperature_west_east_coast: fewer rows or columns than the coast width");
    ATOMDEV_TRY(hipSetDevice(h->device));
    if (int rc = atomdev_refresh_flags(h)) return rc;
    const int jm = h->jm, km = h->km;
    double* t = h->f[ATOMDEV_T];
    hipLaunchKernelGGL(k_coast_north_south, dim3((km + 63) / 64), dim3(64), 0, h->stream, t, h->flags, jm, km, 0);
    hipLaunchKernelGGL(k_coast_north_south, dim3((km + 63) / 64), dim3(64), 0, h->stream, t, h->flags, jm, km, 1);
    hipLaunchKernelGGL(k_coast_east_west, dim3((jm + 63) / 64), dim3(64), 0, h->stream, t, h->flags, jm, km, 0);
    hipLaunchKernelGGL(k_coast_east_west, dim3((jm + 63) / 64), dim3(64), 0, h->stream, t, h->flags, jm, km, 1);
    ATOMDEV_TRY(hipGetLastError());
    return 0;
}

__global__ void k_restrain_temperature(double* __restrict__ t, size_t n2, double shift, double cap) {
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n2) return;
    double v = t[q] - shift;
    if (v > cap) v = cap;
    t[q] = v;
}

extern "C" int atomdev_atm_calculate_mean_temperature(atomdev_t* h, float* mean) {
    ATOMDEV_REQUIRE(h && mean && h->model == ATOMDEV_ATM, "atomdev_atm_calculate_mean_temperature: needs an atmosphere handle and a result");
    ATOMDEV_TRY(hipSetDevice(h->device));
    const int jm = h->jm, km = h->km;
    std::vector<double> sea(h->n2);
    ATOMDEV_TRY(hipMemcpyAsync(sea.data(), h->f[ATOMDEV_T], h->n2 * 8, hipMemcpyDeviceToHost, h->stream));
    if (int rc__ = atomdev_sync_stream(h)) return rc__;
    // calculate_mean_temperature with the weights of calculate_node_weights (:952-988): one running sum in the reference's
    // order, the result a float
    double ret = 0., weight = 0.;
    for (int j = 0; j < jm; j++) {
        const double w = j <= 90 ? std::cos((90 - j) * M_PI / 180.0) : std::cos((j - 90) * M_PI / 180.0);
        for (int k = 0; k < km; k++) {
            ret += sea[(size_t)j * km + k] * w;
            weight += w;
        }
    }
    *mean = (float)((ret / weight - 1) * h->p.t_0);
    return 0;
}

extern "C" int atomdev_atm_restrain_temperature(atomdev_t* h, double mean_temperature) {
    ATOMDEV_REQUIRE(h && h->model == ATOMDEV_ATM, "atomdev_atm_restrain_temperature: needs an atmosphere handle");
    float mean = 0.f;
    if (int rc = atomdev_atm_calculate_mean_temperature(h, &mean)) return rc;
    const double t_0 = h->p.t_0;
    const double tmp_2 = mean;
    const double diff = tmp_2 - mean_temperature;
    hipLaunchKernelGGL(k_restrain_temperature, dim3((unsigned)((h->n2 + 255) / 256)), dim3(256), 0, h->stream, h->f[ATOMDEV_T], h->n2, diff / t_0,
                       1 + 40 / t_0);
    ATOMDEV_TRY(hipGetLastError());
    return 0;
}

extern "C" int atomdev_atm_bc_pressure(atomdev_t* h) {
    ATOMDEV_REQUIRE(h && h->model == ATOMDEV_ATM, "atomdev_atm_bc_pressure: needs an atmosphere handle");
    ATOMDEV_TRY(hipSetDevice(h->device));
    const atomdev_params& P = h->p;
    const double exp_pressure = P.g / (1.e-2 * P.gam * P.R_Air);
    hipLaunchKernelGGL(k_atm_bc_pressure, dim3((unsigned)((h->n2 + 255) / 256)), dim3(256), 0, h->stream, h->f[ATOMDEV_P_STAT], h->f[ATOMDEV_T],
                       h->im, h->n2, P.t_0, P.r_air, P.R_Air, P.gam, P.L_atm, exp_pressure);
    ATOMDEV_TRY(hipGetLastError());
    return 0;
}

extern "C" int atomdev_atm_bc_water_vapour(atomdev_t* h, const atomdev_atm_start* s) {
    ATOMDEV_REQUIRE(h && s && h->model == ATOMDEV_ATM, "atomdev_atm_bc_water_vapour: needs an atmosphere handle and the start values");
    ATOMDEV_TRY(hipSetDevice(h->device));
    if (int rc = atomdev_refresh_flags(h)) return rc;
    const atomdev_params& P = h->p;
    VapourStart S{ P.t_0, P.hp, P.ep, P.r_air, P.R_Air, P.R_WaterVapour, P.u_0, P.L_atm / P.u_0 * P.dt, P.dr * P.L_atm,  // :121-123
                   s->c_ocean, s->c_land, s->c_tropopause, tropopause_add(s->t_cretaceous, P.t_0) };
    int* trop = nullptr;
    if (int rc = upload_tropopause(h, s->im_tropopause, S.trop_add, "atomdev_atm_bc_water_vapour", &trop)) return rc;
    const unsigned nb = (unsigned)((h->n2 + 255) / 256);
    hipLaunchKernelGGL(k_vapour_surface, dim3(nb), dim3(256), 0, h->stream, h->f[ATOMDEV_C], h->f[ATOMDEV_P_STAT], h->f[ATOMDEV_T], h->f[ATOMDEV_V],
                       h->f[ATOMDEV_W], h->plane, h->flags, S, h->im, h->jm, h->km);
    hipLaunchKernelGGL(k_vapour_columns, dim3(nb), dim3(256), 0, h->stream, h->f[ATOMDEV_C], h->plane, h->flags, trop, S, h->im, h->jm, h->km);
    ATOMDEV_TRY(hipGetLastError());
    return 0;
}

extern "C" int atomdev_atm_bc_co2(atomdev_t* h, const atomdev_atm_start* s) {
    ATOMDEV_REQUIRE(h && s && h->model == ATOMDEV_ATM, "atomdev_atm_bc_co2: needs an atmosphere handle and the start values");
    ATOMDEV_TRY(hipSetDevice(h->device));
    if (int rc = atomdev_refresh_flags(h)) return rc;
    const double co2_0 = h->p.co2_0;
    // :797-818
    double co2_cretaceous = 3.2886 * std::pow((s->t_cretaceous + s->t_average), 2) - 32.8859 * (s->t_cretaceous + s->t_average) + 102.2148;
    const double co2_average = 3.2886 * std::pow(s->t_average, 2) - 32.8859 * s->t_average + 102.2148;
    co2_cretaceous = co2_cretaceous - co2_average;
    Co2Start S;
    const double co2_equator = s->co2_equator / co2_0;
    S.co2_pole = s->co2_pole / co2_0;
    S.co2_cretaceous = co2_cretaceous / co2_0;
    S.co2_land = s->co2_land / co2_0; S.co2_ocean = s->co2_ocean / co2_0;
    S.co2_vegetation = s->co2_vegetation / co2_0; S.co2_tropopause = s->co2_tropopause / co2_0;
    S.co2_eff = S.co2_pole - co2_equator;
    S.trop_add = tropopause_add(s->t_cretaceous, h->p.t_0);
    int* trop = nullptr;
    if (int rc = upload_tropopause(h, s->im_tropopause, S.trop_add, "atomdev_atm_bc_co2", &trop)) return rc;
    hipLaunchKernelGGL(k_atm_bc_co2, dim3((unsigned)((h->n2 + 255) / 256)), dim3(256), 0, h->stream, h->f[ATOMDEV_CO2], h->g[ATOMDEV_VEGETATION],
                       h->flags, trop, S, h->im, h->jm, h->km);
    ATOMDEV_TRY(hipGetLastError());
    return 0;
}
