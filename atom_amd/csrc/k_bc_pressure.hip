// Boundary fills (BC_Atmosphere, BC_Hydrosphere) and the pressure step
// (Pressure_Atm, Pressure_Hyd): one Jacobi sweep of the pressure Poisson
// equation framed by boundary fills of aux_u/v/w and p_dyn.
//
// Every fill kernel assigns one thread per boundary row/column, and threads of
// one launch never read what another thread of that launch writes; the
// reference's r -> theta -> phi order is kept as launch order on the handle's
// stream, so edges and corners receive the same values.
#include <cstdlib>

#include "atomdev_internal.h"

namespace {

constexpr int TPB = 256;
constexpr double c43 = 4. / 3., c13 = 1. / 3.;

struct Fields { double* p[9]; int n; };
struct PtrList3 { double* p[3]; };

enum { FILL_NEUMANN = 0, FILL_ZERO = 1 };
struct ThetaFill { double* p[9]; int mode[9]; int n; };

// rows j = 0 and j = jm-1 for every (i, k): x = 4/3 x1 - 1/3 x2, or 0; blockIdx.y = field
__global__ void k_fill_theta(ThetaFill f, int im, int jm, int km, int i0, int ni) {
    int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= ni * km) return;
    int k = id % km, i = i0 + id / km;
    const int n = blockIdx.y;
    double* x = f.p[n];
    if (f.mode[n] == FILL_ZERO) {
        x[idx3(jm, km, i, 0, k)] = 0.;
        x[idx3(jm, km, i, jm - 1, k)] = 0.;
    } else {
        x[idx3(jm, km, i, 0, k)] = c43 * x[idx3(jm, km, i, 1, k)] - c13 * x[idx3(jm, km, i, 2, k)];
        x[idx3(jm, km, i, jm - 1, k)] = c43 * x[idx3(jm, km, i, jm - 2, k)] - c13 * x[idx3(jm, km, i, jm - 3, k)];
    }
}

// columns k = 0 and k = km-1 for (i, j), j in [j0, j1): Neumann on both ends, then
// both set to their mean (e.g. BC_Atm.cpp:121-123); blockIdx.y = field
__global__ void k_fill_phi_seam(Fields f, int im, int jm, int km, int i0, int ni, int j0, int j1) {
    int id = blockIdx.x * blockDim.x + threadIdx.x;
    int nj = j1 - j0;
    if (id >= ni * nj) return;
    int j = j0 + id % nj, i = i0 + id / nj;
    double* x = f.p[blockIdx.y] + idx3(jm, km, i, j, 0);
    double lo = c43 * x[1] - c13 * x[2];
    double hi = c43 * x[km - 2] - c13 * x[km - 3];
    double m = (lo + hi) / 2.;
    x[0] = m; x[km - 1] = m;
}

// BC_Atmosphere::BC_radius, atmosphere/BC_Atm.cpp:39-73
__global__ void k_atm_bc_radius(double* t, double* u, double* v, double* w, Fields ex /* p_dyn c cloud ice co2 */,
                                double t_tropopause, int im, int jm, int km) {
    size_t n2 = (size_t)jm * km;
    size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n2) return;
    const size_t top = (size_t)(im - 1) * n2 + q;
    u[q] = 0.;
    v[q] = c43 * v[n2 + q] - c13 * v[2 * n2 + q];
    w[q] = c43 * w[n2 + q] - c13 * w[2 * n2 + q];
    t[top] = t_tropopause;
    u[top] = 0.; v[top] = 0.; w[top] = 0.;
    for (int n = 0; n < ex.n; n++) {
        double* x = ex.p[n];
        x[top] = x[top - 3 * n2] - 3. * x[top - 2 * n2] + 3. * x[top - n2];
    }
}

// BC_radius, BC_theta and BC_phi of the atmosphere (BC_Atm.cpp:39-150) in one launch.  The three fills read what
// their predecessors wrote only on the faces themselves, so every boundary point works out the values it would have
// read from the earlier fills from points that no fill writes (same formulas, same operands, same bits):
//   after_radius: a point of level 0 or im-1 as BC_radius leaves it, from the levels next to it;
//   after_theta:  a point of row 0 or jm-1 as BC_theta leaves it, from after_radius of rows 1, 2 / jm-2, jm-3;
//   the seam columns k = 0, km-1 of rows 1 .. jm-2 as BC_phi leaves them, from after_radius of columns 1, 2, km-2, km-3.
// One hazard is left: BC_theta on the seam columns reads rows 1, 2, jm-2, jm-3 of those columns as they were BEFORE
// BC_phi rewrites them.  The seam columns of a variable therefore belong to one workgroup, which first takes the
// row-0 / row-jm-1 values of both columns on every level into LDS and only then writes anything.
struct BcAll { double* x[9]; double t_tropopause; };   // t u v w p_dyn c cloud ice co2
enum { B_T = 0, B_U, B_V, B_W, B_P, B_C, B_CLOUD, B_ICE, B_CO2 };

__device__ __forceinline__ double after_radius(const BcAll& b, int f, int i, size_t q, int im, size_t n2) {
    const double* x = b.x[f];
    if (i == 0) {
        if (f == B_U) return 0.;
        if (f == B_V || f == B_W) return c43 * x[n2 + q] - c13 * x[2 * n2 + q];
        return x[q];
    }
    if (i == im - 1) {
        if (f == B_T) return b.t_tropopause;
        if (f == B_U || f == B_V || f == B_W) return 0.;
        const size_t top = (size_t)(im - 1) * n2 + q;
        return x[top - 3 * n2] - 3. * x[top - 2 * n2] + 3. * x[top - n2];
    }
    return x[(size_t)i * n2 + q];
}
__device__ __forceinline__ double after_theta(const BcAll& b, int f, int i, bool north, int k, int im, int jm, int km) {
    if (f == B_V || f == B_W || f == B_P) return 0.;
    const size_t n2 = (size_t)jm * km;
    const int j1 = north ? 1 : jm - 2, j2 = north ? 2 : jm - 3;
    return c43 * after_radius(b, f, i, (size_t)j1 * km + k, im, n2) - c13 * after_radius(b, f, i, (size_t)j2 * km + k, im, n2);
}

// the seam columns of row j on level i as BC_phi leaves them
__device__ __forceinline__ void seam_fill(const BcAll& b, int f, int i, int j, int im, int jm, int km) {
    const size_t n2 = (size_t)jm * km, q = (size_t)j * km;
    const double lo = c43 * after_radius(b, f, i, q + 1, im, n2) - c13 * after_radius(b, f, i, q + 2, im, n2);
    const double hi = c43 * after_radius(b, f, i, q + km - 2, im, n2) - c13 * after_radius(b, f, i, q + km - 3, im, n2);
    const double m = (lo + hi) / 2.;
    b.x[f][(size_t)i * n2 + q] = m; b.x[f][(size_t)i * n2 + q + km - 1] = m;
}

__global__ __launch_bounds__(256) void k_atm_bc_all(BcAll b, int im, int jm, int km, unsigned nb_r, unsigned nb_t, unsigned nb_p) {
    extern __shared__ double corner[];  // seam workgroups: [4 * im]
    const size_t n2 = (size_t)jm * km;
    const int f = (int)blockIdx.y;      // the variable
    double* x = b.x[f];
    if (blockIdx.x < nb_r) {  // levels 0 and im-1 inside their rim
        const unsigned id = blockIdx.x * blockDim.x + threadIdx.x;
        if (id >= (unsigned)(jm - 2) * (km - 2)) return;
        const int j = 1 + (int)(id / (unsigned)(km - 2)), k = 1 + (int)(id % (unsigned)(km - 2));
        const size_t q = (size_t)j * km + k;
        const bool bottom = f == B_U || f == B_V || f == B_W;  // BC_radius leaves the others alone on level 0
        const double lo = bottom ? after_radius(b, f, 0, q, im, n2) : 0., up = after_radius(b, f, im - 1, q, im, n2);
        if (bottom) x[q] = lo;
        x[(size_t)(im - 1) * n2 + q] = up;
        return;
    }
    if (blockIdx.x < nb_r + nb_t) {  // rows 0 and jm-1 between the seam columns, every level
        const unsigned id = (blockIdx.x - nb_r) * blockDim.x + threadIdx.x;
        if (id >= (unsigned)im * (km - 2)) return;
        const int i = (int)(id / (unsigned)(km - 2)), k = 1 + (int)(id % (unsigned)(km - 2));
        const double north = after_theta(b, f, i, true, k, im, jm, km), south = after_theta(b, f, i, false, k, im, jm, km);
        x[idx3(jm, km, i, 0, k)] = north;
        x[idx3(jm, km, i, jm - 1, k)] = south;
        return;
    }
    if (blockIdx.x < nb_r + nb_t + nb_p) {  // seam columns of rows 3 .. jm-4: nothing in this launch reads them
        const unsigned id = (blockIdx.x - nb_r - nb_t) * blockDim.x + threadIdx.x;
        if (id >= (unsigned)im * (unsigned)(jm - 6)) return;
        const int i = (int)(id / (unsigned)(jm - 6)), j = 3 + (int)(id % (unsigned)(jm - 6));
        seam_fill(b, f, i, j, im, jm, km);
        return;
    }
    // the seam columns of the variable on the rows that BC_theta reads there (1, 2, jm-3, jm-2), and their ends
    for (int c = threadIdx.x; c < 4 * im; c += blockDim.x) {
        const int i = c >> 2, k = (c & 1) ? km - 1 : 0;
        corner[c] = after_theta(b, f, i, (c & 2) == 0, k, im, jm, km);
    }
    __syncthreads();
    for (int c = threadIdx.x; c < 4 * im; c += blockDim.x) {
        const int i = c >> 2, k = (c & 1) ? km - 1 : 0, j = (c & 2) == 0 ? 0 : jm - 1;
        x[idx3(jm, km, i, j, k)] = corner[c];
    }
    for (int e = threadIdx.x; e < 4 * im; e += blockDim.x) {
        const int i = e >> 2, r = e & 3, j = r == 0 ? 1 : (r == 1 ? 2 : (r == 2 ? jm - 3 : jm - 2));
        if (r >= 2 && j <= 2) continue;  // (cannot happen: the library's grids have at least 7 rows)
        seam_fill(b, f, i, j, im, jm, km);
    }
}

// RB_radius, RB_theta and RB_phi of the hydrosphere (BC_Hyd.cpp:38-158) in one launch, the same way.  Here the phi fill
// comes last and takes every row, so nothing a later fill reads is rewritten by an earlier one except where the later
// one overwrites it anyway: no hazard, every point belongs to one thread.
struct BcHyd { double* x[6]; double ta, ca; };   // t u v w p_dyn c
enum { HB_T = 0, HB_U, HB_V, HB_W, HB_P, HB_C };
__device__ __forceinline__ double hyd_after_radius(const BcHyd& b, int f, int i, size_t q, int im, size_t n2) {
    const double* x = b.x[f];
    if (i == 0) {
        if (f == HB_U) return c43 * x[n2 + q] - c13 * x[2 * n2 + q];
        if (f == HB_T) return b.ta;
        if (f == HB_C) return b.ca;
        return 0.;  // v, w, p_dyn
    }
    if (i == im - 1) {
        const size_t top = (size_t)(im - 1) * n2 + q;
        if (f == HB_V || f == HB_W) return c43 * x[top - n2] - c13 * x[top - 2 * n2];
        if (f == HB_U || f == HB_C) return x[top - 3 * n2] - 3. * x[top - 2 * n2] + 3. * x[top - n2];
        return x[top];  // t, p_dyn: left alone on the surface level
    }
    return x[(size_t)i * n2 + q];
}
__global__ __launch_bounds__(256) void k_hyd_bc_all(BcHyd b, int im, int jm, int km, unsigned nb_r, unsigned nb_t) {
    const size_t n2 = (size_t)jm * km;
    const int f = (int)blockIdx.y;
    double* x = b.x[f];
    auto R = [&](int i, int j, int k) { return hyd_after_radius(b, f, i, (size_t)j * km + k, im, n2); };
    auto theta = [&](int i, bool north, int k) {  // RB_theta: v and w are zero on the two rows, the others extrapolated
        if (f == HB_V || f == HB_W) return 0.;
        const int j1 = north ? 1 : jm - 2, j2 = north ? 2 : jm - 3;
        return c43 * R(i, j1, k) - c13 * R(i, j2, k);
    };
    auto V = [&](int i, int j, int k) { return j == 0 ? theta(i, true, k) : (j == jm - 1 ? theta(i, false, k) : R(i, j, k)); };
    if (blockIdx.x < nb_r) {  // levels 0 and im-1 inside their rim
        const unsigned id = blockIdx.x * blockDim.x + threadIdx.x;
        if (id >= (unsigned)(jm - 2) * (km - 2)) return;
        const int j = 1 + (int)(id / (unsigned)(km - 2)), k = 1 + (int)(id % (unsigned)(km - 2));
        const double lo = R(0, j, k), up = R(im - 1, j, k);
        x[idx3(jm, km, 0, j, k)] = lo;
        if (f != HB_T && f != HB_P) x[idx3(jm, km, im - 1, j, k)] = up;
        return;
    }
    if (blockIdx.x < nb_r + nb_t) {  // rows 0 and jm-1 between the seam columns, every level
        const unsigned id = (blockIdx.x - nb_r) * blockDim.x + threadIdx.x;
        if (id >= (unsigned)im * (km - 2)) return;
        const int i = (int)(id / (unsigned)(km - 2)), k = 1 + (int)(id % (unsigned)(km - 2));
        const double north = theta(i, true, k), south = theta(i, false, k);
        x[idx3(jm, km, i, 0, k)] = north; x[idx3(jm, km, i, jm - 1, k)] = south;
        return;
    }
    // the seam columns of every row
    const unsigned id = (blockIdx.x - nb_r - nb_t) * blockDim.x + threadIdx.x;
    if (id >= (unsigned)im * (unsigned)jm) return;
    const int i = (int)(id / (unsigned)jm), j = (int)(id % (unsigned)jm);
    const double lo = c43 * V(i, j, 1) - c13 * V(i, j, 2), hi = c43 * V(i, j, km - 2) - c13 * V(i, j, km - 3);
    const double m = (lo + hi) / 2.;
    x[idx3(jm, km, i, j, 0)] = m; x[idx3(jm, km, i, j, km - 1)] = m;
}

// BC_Hydrosphere::RB_radius, hydrosphere/BC_Hyd.cpp:38-83
__global__ void k_hyd_bc_radius(double* t, double* u, double* v, double* w, double* p_dyn, double* c, double ta, double ca,
                                int im, int jm, int km) {
    size_t n2 = (size_t)jm * km;
    size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n2) return;
    const size_t top = (size_t)(im - 1) * n2 + q;
    u[q] = c43 * u[n2 + q] - c13 * u[2 * n2 + q];
    t[q] = ta; c[q] = ca; v[q] = 0.; w[q] = 0.; p_dyn[q] = 0.;
    v[top] = c43 * v[top - n2] - c13 * v[top - 2 * n2];
    w[top] = c43 * w[top - n2] - c13 * w[top - 2 * n2];
    u[top] = u[top - 3 * n2] - 3. * u[top - 2 * n2] + 3. * u[top - n2];
    c[top] = c[top - 3 * n2] - 3. * c[top - 2 * n2] + 3. * c[top - n2];
}

// radial faces of aux_u/v/w (Pressure_Atm.cpp:47-56, Pressure_Hyd.cpp:51-62)
__global__ void k_aux_fill_r(double* au, double* av, double* aw, int im, int jm, int km) {
    size_t n2 = (size_t)jm * km;
    size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n2) return;
    const size_t top = (size_t)(im - 1) * n2 + q;
    au[q] = 0.; au[top] = 0.;
    av[q] = c43 * av[n2 + q] - c13 * av[2 * n2 + q];
    av[top] = c43 * av[top - n2] - c13 * av[top - 2 * n2];
    aw[q] = c43 * aw[n2 + q] - c13 * aw[2 * n2 + q];
    aw[top] = c43 * aw[top - n2] - c13 * aw[top - 2 * n2];
}

// radial faces of p_dyn after the sweep (Pressure_Atm.cpp:163-168, Pressure_Hyd.cpp:176-181)
__global__ void k_p_fill_r(double* p, int im, int jm, int km) {
    size_t n2 = (size_t)jm * km;
    size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n2) return;
    const size_t top = (size_t)(im - 1) * n2 + q;
    p[q] = p[3 * n2 + q] - 3. * p[2 * n2 + q] + 3. * p[n2 + q];
    p[top] = c43 * p[top - n2] - c13 * p[top - 2 * n2];
}

// ---- the fills around the pressure step, each group of three in one launch ----
// Before the sweep: aux_u/v/w radially (k_aux_fill_r), then their theta rows, then the phi seam of EVERY row
// (Pressure_Atm.cpp:47-101).  After it: p_dyn radially (k_p_fill_r), then the phi seam of every row, then the theta rows
// of every column (Pressure_Atm.cpp:162-186).  As in k_atm_bc_all a face point works out what it would have read from
// the earlier fills; here nothing that a later fill reads is rewritten by an earlier one except where the later fill
// overwrites it anyway, so there is no hazard and every point belongs to one thread.
enum { AUX_FILLS = 0, P_FILLS = 1 };
template <int WHICH>
__device__ __forceinline__ double pr_after_r(const double* __restrict__ x, bool zero_faces, int i, size_t q, int im, size_t n2) {
    if (i == 0) {
        if (WHICH == AUX_FILLS) return zero_faces ? 0. : c43 * x[n2 + q] - c13 * x[2 * n2 + q];
        return x[3 * n2 + q] - 3. * x[2 * n2 + q] + 3. * x[n2 + q];
    }
    if (i == im - 1) {
        const size_t top = (size_t)(im - 1) * n2 + q;
        if (WHICH == AUX_FILLS && zero_faces) return 0.;
        return c43 * x[top - n2] - c13 * x[top - 2 * n2];
    }
    return x[(size_t)i * n2 + q];
}
template <int WHICH>
__global__ __launch_bounds__(256) void k_pressure_fills(PtrList3 fields, int im, int jm, int km, unsigned nb_r, unsigned nb_t) {
    double* __restrict__ x = fields.p[blockIdx.y];
    const bool zf = WHICH == AUX_FILLS && blockIdx.y == 0;  // aux_u is zero on both radial faces
    const size_t n2 = (size_t)jm * km;
    auto R = [&](int i, int j, int k) { return pr_after_r<WHICH>(x, zf, i, (size_t)j * km + k, im, n2); };
    // an interior column k of row j after the radial and (aux only: before the seam) the theta fill
    auto V = [&](int i, int j, int k) {
        if (WHICH == AUX_FILLS && (j == 0 || j == jm - 1)) {
            const int j1 = j == 0 ? 1 : jm - 2, j2 = j == 0 ? 2 : jm - 3;
            return c43 * R(i, j1, k) - c13 * R(i, j2, k);
        }
        return R(i, j, k);
    };
    auto seam = [&](int i, int j) {  // what BC_phi's rule makes of row j
        const double lo = c43 * V(i, j, 1) - c13 * V(i, j, 2), hi = c43 * V(i, j, km - 2) - c13 * V(i, j, km - 3);
        return (lo + hi) / 2.;
    };
    if (blockIdx.x < nb_r) {  // levels 0 and im-1 inside their rim
        const unsigned id = blockIdx.x * blockDim.x + threadIdx.x;
        if (id >= (unsigned)(jm - 2) * (km - 2)) return;
        const int j = 1 + (int)(id / (unsigned)(km - 2)), k = 1 + (int)(id % (unsigned)(km - 2));
        const double lo = R(0, j, k), up = R(im - 1, j, k);
        x[idx3(jm, km, 0, j, k)] = lo; x[idx3(jm, km, im - 1, j, k)] = up;
        return;
    }
    if (blockIdx.x < nb_r + nb_t) {  // rows 0 and jm-1: between the seam columns (aux), every column (p_dyn: the theta fill is last)
        const unsigned kn = WHICH == AUX_FILLS ? (unsigned)(km - 2) : (unsigned)km;
        const unsigned id = (blockIdx.x - nb_r) * blockDim.x + threadIdx.x;
        if (id >= (unsigned)im * kn) return;
        const int i = (int)(id / kn), k = (WHICH == AUX_FILLS ? 1 : 0) + (int)(id % kn);
        double north, south;
        if (WHICH == P_FILLS && (k == 0 || k == km - 1)) {
            north = c43 * seam(i, 1) - c13 * seam(i, 2); south = c43 * seam(i, jm - 2) - c13 * seam(i, jm - 3);
        } else {
            north = c43 * R(i, 1, k) - c13 * R(i, 2, k); south = c43 * R(i, jm - 2, k) - c13 * R(i, jm - 3, k);
        }
        x[idx3(jm, km, i, 0, k)] = north; x[idx3(jm, km, i, jm - 1, k)] = south;
        return;
    }
    // the seam columns: every row (aux), the rows between the theta rows (p_dyn)
    const unsigned jn = WHICH == AUX_FILLS ? (unsigned)jm : (unsigned)(jm - 2);
    const unsigned id = (blockIdx.x - nb_r - nb_t) * blockDim.x + threadIdx.x;
    if (id >= (unsigned)im * jn) return;
    const int i = (int)(id / jn), j = (WHICH == AUX_FILLS ? 0 : 1) + (int)(id % jn);
    const double m = seam(i, j);
    x[idx3(jm, km, i, j, 0)] = m; x[idx3(jm, km, i, j, km - 1)] = m;
}

// Interior Jacobi sweep: Pressure_Atm.cpp:104-160 / Pressure_Hyd.cpp:114-173.  Reads
// p_dynn (old), writes p_dyn.  hyd: aux_v / aux_w are read at level im-1 for every i
// (Pressure_Hyd.cpp:137-163), the radial land test is h[i+1] == 0., the western
// phi branch tests a land cell with water to the west, and land is not zeroed.
template <bool HYD>
__global__ void k_pressure3d(double* __restrict__ p_dyn, const double* __restrict__ pn, const double* __restrict__ au,
                             const double* __restrict__ av, const double* __restrict__ aw,
                             const uint8_t* __restrict__ flags, const double* __restrict__ rad,
                             const double* __restrict__ sinthe_t, double rho, DevConst K) {
    const int im = K.im, jm = K.jm, km = K.km;
    // Workgroups go to the eight XCDs in turn.  Hydrosphere: each XCD gets one contiguous run of (level, row, segment)
    // triples, so the neighbour rows and levels a workgroup reads are, most of the time, already in its XCD's L2 (42 ->
    // 39 us).  The atmosphere's form, which fetches most operands only where a branch needs them, is faster dealt round
    // robin (41 us; 57 us in runs).
    const unsigned gx = (unsigned)(km - 2 + (int)blockDim.x - 1) / blockDim.x, nblk = gx * (unsigned)(jm - 2) * (unsigned)(im - 2);
    const unsigned per = (nblk + 7) / 8, L = HYD ? (blockIdx.x & 7) * per + (blockIdx.x >> 3) : blockIdx.x;
    if ((HYD && (blockIdx.x >> 3) >= per) || L >= nblk) return;
    const int k = 1 + (int)(L % gx) * (int)blockDim.x + (int)threadIdx.x;
    const int j = 1 + (int)((L / gx) % (unsigned)(jm - 2)), i = 1 + (int)(L / (gx * (unsigned)(jm - 2)));
    // What depends on the row (i, j) only - ten divisions of the reference's statements - is worked out by one thread of
    // the block; every quotient by one of these divisors is then div_by with the divisor's correctly rounded reciprocal
    // (the bits of the division, atomdev_internal.h).  2 dphi rm sin (Pressure_Hyd) and 2 rm sin dphi (Pressure_Atm) are
    // the same number: doubling is exact and the product commutes.
    __shared__ double rc[12];
    if (threadIdx.x == 0) {
        const double rm = rad[i], rm2 = rm * rm;
        const double sinthe = sinthe_t[j];
        const double rmsinthe = rm * sinthe;
        const double rm2sinthe2 = rmsinthe * rmsinthe;
        const double denom = 2. / K.dr2 + 2. / (rm2 * K.dthe2) + 2. / (rm2sinthe2 * K.dphi2);
        rc[0] = denom; rc[1] = 1. / denom;
        rc[2] = 1. / (rm2 * K.dthe2); rc[3] = 1. / (rm2sinthe2 * K.dphi2);
        const double t2 = 2. * K.dthe * rm, t1 = K.dthe * rm, w2 = 2. * K.dphi * rmsinthe, w1 = K.dphi * rmsinthe;
        rc[4] = t2; rc[5] = 1. / t2; rc[6] = t1; rc[7] = 1. / t1;
        rc[8] = w2; rc[9] = 1. / w2; rc[10] = w1; rc[11] = 1. / w1;
    }
    __syncthreads();
    if (k > km - 2) return;
    const size_t row = km, lvl = (size_t)jm * km;
    const size_t p = idx3(jm, km, i, j, k);
    const double denom = rc[0], i_denom = rc[1], num1 = K.i_dr2, num2 = rc[2], num3 = rc[3];
    const double t2 = rc[4], i_t2 = rc[5], t1 = rc[6], i_t1 = rc[7], w2 = rc[8], i_w2 = rc[9], w1 = rc[10], i_w1 = rc[11];
    const double two_dr = K.two_dr, i_two_dr = K.i_two_dr, dr = K.dr, i_dr = K.i_dr;

    double r;
    if constexpr (HYD) {
        // (the hydrosphere reads aux_v / aux_w on one level for all: loading every operand before the first decision costs
        // no traffic there and saves the dependent round trips; 64 -> 54 us.  The atmosphere, where most points take the
        // central forms and skip the far operands, is faster as the reference writes it.)
        // every operand is loaded before the first decision (rows and columns two away exist only inside the index
        // windows: outside them the address is that of the centre, whose value then is not used)
        const bool jwin = (j >= 2) && (j <= jm - 3), kwin = (k >= 2) && (k <= km - 3), iwin = i <= im - 3;
        const size_t a = HYD ? idx3(jm, km, im - 1, j, k) : p;  // aux_v / aux_w base: own level (atm) or level im-1 (hyd)
        const size_t r2 = jwin ? 2 * row : 0, c2 = kwin ? 2 : 0, l2 = iwin ? 2 * lvl : 0;
        const uint8_t f_c = flags[p], f_up = flags[p + lvl], f_jp = flags[p + row], f_jm = flags[p - row], f_jp2 = flags[p + r2],
                      f_jm2 = flags[p - r2], f_kp = flags[p + 1], f_km = flags[p - 1], f_kp2 = flags[p + c2], f_km2 = flags[p - c2];
        const double au_c = au[p], au_p = au[p + lvl], au_m = au[p - lvl], au_p2 = au[p + l2];
        const double av_c = av[a], av_p = av[a + row], av_m = av[a - row], av_p2 = av[a + r2], av_m2 = av[a - r2];
        const double aw_c = aw[a], aw_p = aw[a + 1], aw_m = aw[a - 1], aw_p2 = aw[a + c2], aw_m2 = aw[a - c2];
        const double pn_ip = pn[p + lvl], pn_im = pn[p - lvl], pn_jp = pn[p + row], pn_jm = pn[p - row], pn_kp = pn[p + 1], pn_km = pn[p - 1];
        auto is_land = [](uint8_t f) { return (f & FL_LAND) != 0; };
        const bool lc = is_land(f_c);

        double drhs_udr = div_by(au_p - au_m, two_dr, i_two_dr);
        if (iwin) {
            const bool above_fluid = HYD ? ((f_up & FL_HZERO) != 0) : !is_land(f_up);
            if (lc && above_fluid) drhs_udr = div_by(-3. * au_c + 4. * au_p - au_p2, two_dr, i_two_dr);
        } else {
            drhs_udr = div_by(au_p - au_c, dr, i_dr);
        }

        double drhs_vdthe = div_by(av_p - av_m, t2, i_t2);
        if (lc && !is_land(f_jp)) drhs_vdthe = div_by(av_p - av_c, t1, i_t1);
        if (lc && !is_land(f_jm)) drhs_vdthe = div_by(av_m - av_c, t1, i_t1);
        if (jwin) {
            if (lc && (!is_land(f_jp) && !is_land(f_jp2))) drhs_vdthe = div_by(-3. * av_c + 4. * av_p - av_p2, t2, i_t2);
            if (lc && !is_land(f_jm) && !is_land(f_jm2)) drhs_vdthe = div_by(-3. * av_c + 4. * av_m - av_m2, t2, i_t2);
        }

        double drhs_wdphi = div_by(aw_p - aw_m, w2, i_w2);
        if (lc && !is_land(f_kp))
            drhs_wdphi = div_by(aw_p - aw_c, w1, i_w1);
        if (HYD ? (lc && !is_land(f_km)) : (!lc && is_land(f_km)))
            drhs_wdphi = div_by(aw_m - aw_c, w1, i_w1);
        if (kwin) {
            if (lc && !is_land(f_kp) && !is_land(f_kp2)) drhs_wdphi = div_by(-3. * aw_c + 4. * aw_p - aw_p2, w2, i_w2);
            if (lc && !is_land(f_km) && !is_land(f_km2)) drhs_wdphi = div_by(-3. * aw_c + 4. * aw_m - aw_m2, w2, i_w2);
        }

        r = div_by((pn_ip + pn_im) * num1 + (pn_jp + pn_jm) * num2 + (pn_kp + pn_km) * num3
                   - rho * (drhs_udr + drhs_vdthe + drhs_wdphi), denom, i_denom);
        if (!HYD && lc) r = .0;
    } else {
        auto land = [&](size_t q) { return (flags[q] & FL_LAND) != 0; };
        const bool lc = land(p);

        double drhs_udr = div_by(au[p + lvl] - au[p - lvl], two_dr, i_two_dr);
        if (i <= im - 3) {
            bool above_fluid = HYD ? ((flags[p + lvl] & FL_HZERO) != 0) : !land(p + lvl);
            if (lc && above_fluid) drhs_udr = div_by(-3. * au[p] + 4. * au[p + lvl] - au[p + 2 * lvl], two_dr, i_two_dr);
        } else {
            drhs_udr = div_by(au[p + lvl] - au[p], dr, i_dr);
        }

        // aux_v / aux_w base: own level (atm) or level im-1 (hyd)
        const size_t a = HYD ? idx3(jm, km, im - 1, j, k) : p;
        double drhs_vdthe = div_by(av[a + row] - av[a - row], t2, i_t2);
        if (lc && !land(p + row)) drhs_vdthe = div_by(av[a + row] - av[a], t1, i_t1);
        if (lc && !land(p - row)) drhs_vdthe = div_by(av[a - row] - av[a], t1, i_t1);
        if ((j >= 2) && (j <= jm - 3)) {
            if (lc && (!land(p + row) && !land(p + 2 * row)))
                drhs_vdthe = div_by(-3. * av[a] + 4. * av[a + row] - av[a + 2 * row], t2, i_t2);
            if (lc && !land(p - row) && !land(p - 2 * row))
                drhs_vdthe = div_by(-3. * av[a] + 4. * av[a - row] - av[a - 2 * row], t2, i_t2);
        }

        double drhs_wdphi = div_by(aw[a + 1] - aw[a - 1], w2, i_w2);
        if (lc && !land(p + 1))
            drhs_wdphi = div_by(aw[a + 1] - aw[a], w1, i_w1);
        if (HYD ? (lc && !land(p - 1)) : (!lc && land(p - 1)))
            drhs_wdphi = div_by(aw[a - 1] - aw[a], w1, i_w1);
        if ((k >= 2) && (k <= km - 3)) {
            if (lc && !land(p + 1) && !land(p + 2))
                drhs_wdphi = div_by(-3. * aw[a] + 4. * aw[a + 1] - aw[a + 2], w2, i_w2);
            if (lc && !land(p - 1) && !land(p - 2))
                drhs_wdphi = div_by(-3. * aw[a] + 4. * aw[a - 1] - aw[a - 2], w2, i_w2);
        }

        r = div_by((pn[p + lvl] + pn[p - lvl]) * num1 + (pn[p + row] + pn[p - row]) * num2 + (pn[p + 1] + pn[p - 1]) * num3
                   - rho * (drhs_udr + drhs_vdthe + drhs_wdphi), denom, i_denom);
        if (!HYD && lc) r = .0;
    }
    p_dyn[p] = r;
}

// Surface sweep at level L: Pressure_Atm.cpp:217-281 / Pressure_Hyd.cpp:228-298
__global__ void k_pressure2d(double* __restrict__ p_dyn, const double* __restrict__ pn, const double* __restrict__ av,
                             const double* __restrict__ aw, const uint8_t* __restrict__ flags, const double* __restrict__ rad,
                             const double* __restrict__ sinthe_t, double rho, int L, DevConst K) {
    const int jm = K.jm, km = K.km;
    const int k = 1 + blockIdx.x * blockDim.x + threadIdx.x;
    const int j = 1 + blockIdx.y;
    if (k > km - 2) return;
    const size_t row = km;
    const size_t p = idx3(jm, km, L, j, k);
    const double dthe = K.dthe, dphi = K.dphi;
    const double rm = rad[L], rm2 = rm * rm;
    const double sinthe = sinthe_t[j];
    const double rmsinthe = rm * sinthe;
    const double rm2sinthe2 = rmsinthe * rmsinthe;
    const double denom = 2. / (rm2 * K.dthe2) + 2. / (rm2sinthe2 * K.dphi2);
    const double num2 = 1. / (rm2 * K.dthe2), num3 = 1. / (rm2sinthe2 * K.dphi2);
    auto land = [&](size_t q) { return (flags[q] & FL_LAND) != 0; };
    const bool lc = land(p);

    double drhs_vdthe = (av[p + row] - av[p - row]) / (2. * dthe * rm);
    if (lc && !land(p + row)) {
        if (j < jm - 2 && !land(p + 2 * row)) drhs_vdthe = (-3. * av[p] + 4. * av[p + row] - av[p + 2 * row]) / (2. * dthe * rm);
        else drhs_vdthe = (av[p + row] - av[p]) / (dthe * rm);
    }
    if (lc && !land(p - row)) {
        if (j > 1 && !land(p - 2 * row)) drhs_vdthe = (-3. * av[p] + 4. * av[p - row] - av[p - 2 * row]) / (2. * dthe * rm);
        else drhs_vdthe = (av[p - row] - av[p]) / (dthe * rm);
    }
    double drhs_wdphi = (aw[p + 1] - aw[p - 1]) / (2. * dphi * rmsinthe);
    if (lc && !land(p + 1)) {
        if (k < km - 2 && !land(p + 2)) drhs_wdphi = (-3. * aw[p] + 4. * aw[p + 1] - aw[p + 2]) / (2. * rmsinthe * dphi);
        else drhs_wdphi = (aw[p + 1] - aw[p]) / (dphi * rmsinthe);
    }
    if (lc && !land(p - 1)) {
        if (k >= 2 && !land(p - 2)) drhs_wdphi = (-3. * aw[p] + 4. * aw[p - 1] - aw[p - 2]) / (2. * rmsinthe * dphi);
        else drhs_wdphi = (aw[p - 1] - aw[p]) / (dphi * rmsinthe);
    }
    p_dyn[p] = ((pn[p + row] + pn[p - row]) * num2 + (pn[p + 1] + pn[p - 1]) * num3 - rho * (drhs_vdthe + drhs_wdphi)) / denom;
}

// p_dyn = 0 on two rows after the surface sweep: (0, 0, :) and (0, jm-1, :) for the
// atmosphere (Pressure_Atm.cpp:284-288); (im-1, im-1, :) and (0, jm-1, :) for the
// hydrosphere, whose row index slips to im-1 (Pressure_Hyd.cpp:301-305).
__global__ void k_zero_two_rows(double* p, size_t row_a, size_t row_b, int km) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= km) return;
    p[row_a + k] = 0.;
    p[row_b + k] = 0.;
}

inline unsigned nb(size_t n) { return (unsigned)((n + TPB - 1) / TPB); }

int fill_theta(atomdev* h, const int* ids, const int* modes, int n, int i0, int ni) {
    ThetaFill f; f.n = n;
    for (int m = 0; m < n; m++) { f.p[m] = h->f[ids[m]]; f.mode[m] = modes[m]; }
    hipLaunchKernelGGL(k_fill_theta, dim3(nb((size_t)ni * h->km), n), dim3(TPB), 0, h->stream, f, h->im, h->jm, h->km, i0, ni);
    return atomdev_check(hipGetLastError(), "k_fill_theta");
}

int fill_phi(atomdev* h, const int* ids, int n, int i0, int ni, int j0, int j1) {
    Fields f; f.n = n;
    for (int m = 0; m < n; m++) f.p[m] = h->f[ids[m]];
    hipLaunchKernelGGL(k_fill_phi_seam, dim3(nb((size_t)ni * (j1 - j0)), n), dim3(TPB), 0, h->stream, f, h->im, h->jm, h->km,
                       i0, ni, j0, j1);
    return atomdev_check(hipGetLastError(), "k_fill_phi_seam");
}

int pressure3d(atomdev* h, bool hyd) {
    if (int rc = atomdev_refresh_flags(h)) return rc;
    static const int AUX[3] = { ATOMDEV_AUX_U, ATOMDEV_AUX_V, ATOMDEV_AUX_W };
    static const int NE3[3] = { FILL_NEUMANN, FILL_NEUMANN, FILL_NEUMANN };
    const size_t n2 = h->n2;
    const bool one_launch = h->jm >= 7 && h->km >= 7 && h->im >= 5 && !getenv("ATOMDEV_PRESSURE_FILLS_ONE_BY_ONE");
    const unsigned nb_r = nb((size_t)(h->jm - 2) * (h->km - 2));
    if (one_launch) {
        PtrList3 aux = { { h->f[ATOMDEV_AUX_U], h->f[ATOMDEV_AUX_V], h->f[ATOMDEV_AUX_W] } };
        const unsigned nb_t = nb((size_t)h->im * (h->km - 2)), nb_s = nb((size_t)h->im * h->jm);
        hipLaunchKernelGGL(k_pressure_fills<AUX_FILLS>, dim3(nb_r + nb_t + nb_s, 3), dim3(TPB), 0, h->stream, aux, h->im, h->jm, h->km, nb_r, nb_t);
        ATOMDEV_TRY(hipGetLastError());
    } else {
        hipLaunchKernelGGL(k_aux_fill_r, dim3(nb(n2)), dim3(TPB), 0, h->stream, h->f[ATOMDEV_AUX_U], h->f[ATOMDEV_AUX_V],
                           h->f[ATOMDEV_AUX_W], h->im, h->jm, h->km);
        ATOMDEV_TRY(hipGetLastError());
        if (int rc = fill_theta(h, AUX, NE3, 3, 0, h->im)) return rc;
        if (int rc = fill_phi(h, AUX, 3, 0, h->im, 0, h->jm)) return rc;
    }
    const unsigned nblk3 = (unsigned)((h->km - 2 + 127) / 128) * (unsigned)(h->jm - 2) * (unsigned)(h->im - 2);
    dim3 grid(8 * ((nblk3 + 7) / 8));
    const double rho = hyd ? h->p.r_0_water : h->p.r_air;
    if (hyd)
        hipLaunchKernelGGL(k_pressure3d<true>, grid, dim3(128), 0, h->stream, h->f[ATOMDEV_P_DYN], h->f[ATOMDEV_P_DYNN],
                           h->f[ATOMDEV_AUX_U], h->f[ATOMDEV_AUX_V], h->f[ATOMDEV_AUX_W], h->flags, h->rad, h->sinthe, rho, h->k);
    else
        hipLaunchKernelGGL(k_pressure3d<false>, grid, dim3(128), 0, h->stream, h->f[ATOMDEV_P_DYN], h->f[ATOMDEV_P_DYNN],
                           h->f[ATOMDEV_AUX_U], h->f[ATOMDEV_AUX_V], h->f[ATOMDEV_AUX_W], h->flags, h->rad, h->sinthe, rho, h->k);
    ATOMDEV_TRY(hipGetLastError());
    // p_dyn faces: r, then phi (all j), then theta (Pressure_Atm.cpp:162-186)
    if (one_launch) {
        PtrList3 pd = { { h->f[ATOMDEV_P_DYN], nullptr, nullptr } };
        const unsigned nb_t = nb((size_t)h->im * h->km), nb_s = nb((size_t)h->im * (h->jm - 2));
        hipLaunchKernelGGL(k_pressure_fills<P_FILLS>, dim3(nb_r + nb_t + nb_s, 1), dim3(TPB), 0, h->stream, pd, h->im, h->jm, h->km, nb_r, nb_t);
        return atomdev_check(hipGetLastError(), "k_pressure_fills");
    }
    hipLaunchKernelGGL(k_p_fill_r, dim3(nb(n2)), dim3(TPB), 0, h->stream, h->f[ATOMDEV_P_DYN], h->im, h->jm, h->km);
    ATOMDEV_TRY(hipGetLastError());
    static const int PD[1] = { ATOMDEV_P_DYN };
    if (int rc = fill_phi(h, PD, 1, 0, h->im, 0, h->jm)) return rc;
    if (int rc = fill_theta(h, PD, NE3, 1, 0, h->im)) return rc;
    return 0;
}

int pressure2d(atomdev* h, bool hyd) {
    if (int rc = atomdev_refresh_flags(h)) return rc;
    const int L = hyd ? h->im - 1 : 0;
    static const int AV[1] = { ATOMDEV_AUX_V };
    static const int NE[1] = { FILL_NEUMANN };
    static const int PD[1] = { ATOMDEV_P_DYN };
    // aux_v only: theta rows, then the phi seam for every j (Pressure_Atm.cpp:202-214)
    if (int rc = fill_theta(h, AV, NE, 1, L, 1)) return rc;
    if (int rc = fill_phi(h, AV, 1, L, 1, 0, h->jm)) return rc;
    dim3 grid((h->km - 2 + 127) / 128, h->jm - 2);
    const double rho = hyd ? h->p.r_0_water : h->p.r_air;
    hipLaunchKernelGGL(k_pressure2d, grid, dim3(128), 0, h->stream, h->f[ATOMDEV_P_DYN], h->f[ATOMDEV_P_DYNN],
                       h->f[ATOMDEV_AUX_V], h->f[ATOMDEV_AUX_W], h->flags, h->rad, h->sinthe, rho, L, h->k);
    ATOMDEV_TRY(hipGetLastError());
    size_t row_a, row_b = idx3(h->jm, h->km, 0, h->jm - 1, 0);
    if (hyd) {
        ATOMDEV_REQUIRE(h->jm > h->im - 1, "atomdev_hyd_pressure2d: the reference writes row j = im-1; the grid needs jm > im-1");
        row_a = idx3(h->jm, h->km, h->im - 1, h->im - 1, 0);
    } else {
        row_a = idx3(h->jm, h->km, 0, 0, 0);
    }
    hipLaunchKernelGGL(k_zero_two_rows, dim3(nb(h->km)), dim3(TPB), 0, h->stream, h->f[ATOMDEV_P_DYN], row_a, row_b, h->km);
    ATOMDEV_TRY(hipGetLastError());
    if (int rc = fill_phi(h, PD, 1, L, 1, 1, h->jm - 1)) return rc;
    return 0;
}

}  // namespace

// Neumann fill of rows j = 0, jm-1 (every i, k) / averaged-seam fill of columns k = 0, km-1 (every i,
// j in [j0, j1)) of one field, for operators in other translation units.
int atomdev_fill_theta_neumann(atomdev* h, int field) {
    const int mode = FILL_NEUMANN;
    return fill_theta(h, &field, &mode, 1, 0, h->im);
}
int atomdev_fill_phi_seam(atomdev* h, int field, int j0, int j1) { return fill_phi(h, &field, 1, 0, h->im, j0, j1); }

extern "C" int atomdev_atm_bc_radius(atomdev_t* h) {
    ATOMDEV_REQUIRE(h && h->model == ATOMDEV_ATM, "atomdev_atm_bc_radius: not an atmosphere handle");
    ATOMDEV_TRY(hipSetDevice(h->device));
    Fields ex; ex.n = 5;
    ex.p[0] = h->f[ATOMDEV_P_DYN]; ex.p[1] = h->f[ATOMDEV_C]; ex.p[2] = h->f[ATOMDEV_CLOUD]; ex.p[3] = h->f[ATOMDEV_ICE]; ex.p[4] = h->f[ATOMDEV_CO2];
    hipLaunchKernelGGL(k_atm_bc_radius, dim3(nb(h->n2)), dim3(TPB), 0, h->stream, h->f[ATOMDEV_T], h->f[ATOMDEV_U],
                       h->f[ATOMDEV_V], h->f[ATOMDEV_W], ex, h->p.t_tropopause, h->im, h->jm, h->km);
    ATOMDEV_TRY(hipGetLastError());
    return 0;
}

extern "C" int atomdev_atm_bc_radius_theta_phi(atomdev_t* h) {
    ATOMDEV_REQUIRE(h && h->model == ATOMDEV_ATM, "atomdev_atm_bc_radius_theta_phi: not an atmosphere handle");
    ATOMDEV_TRY(hipSetDevice(h->device));
    static const int ids[9] = { ATOMDEV_T, ATOMDEV_U, ATOMDEV_V, ATOMDEV_W, ATOMDEV_P_DYN, ATOMDEV_C, ATOMDEV_CLOUD, ATOMDEV_ICE, ATOMDEV_CO2 };
    BcAll b;
    for (int f = 0; f < 9; f++) b.x[f] = h->f[ids[f]];
    b.t_tropopause = h->p.t_tropopause;
    const unsigned nb_r = (unsigned)nb((size_t)(h->jm - 2) * (h->km - 2)), nb_t = (unsigned)nb((size_t)h->im * (h->km - 2));
    ATOMDEV_REQUIRE(h->jm >= 7 && h->km >= 7, "atomdev_atm_bc_radius_theta_phi: needs at least 7 rows and 7 columns");
    const unsigned nb_p = (unsigned)nb((size_t)h->im * (h->jm - 6));
    hipLaunchKernelGGL(k_atm_bc_all, dim3(nb_r + nb_t + nb_p + 1, 9), dim3(TPB), (size_t)4 * h->im * sizeof(double), h->stream, b,
                       h->im, h->jm, h->km, nb_r, nb_t, nb_p);
    return atomdev_check(hipGetLastError(), "k_atm_bc_all");
}

extern "C" int atomdev_atm_bc_theta(atomdev_t* h) {
    ATOMDEV_REQUIRE(h && h->model == ATOMDEV_ATM, "atomdev_atm_bc_theta: not an atmosphere handle");
    ATOMDEV_TRY(hipSetDevice(h->device));
    // BC_Atm.cpp:80-107
    static const int ids[9] = { ATOMDEV_T, ATOMDEV_U, ATOMDEV_V, ATOMDEV_W, ATOMDEV_P_DYN, ATOMDEV_C, ATOMDEV_CLOUD, ATOMDEV_ICE, ATOMDEV_CO2 };
    static const int md[9] = { FILL_NEUMANN, FILL_NEUMANN, FILL_ZERO, FILL_ZERO, FILL_ZERO, FILL_NEUMANN, FILL_NEUMANN, FILL_NEUMANN, FILL_NEUMANN };
    return fill_theta(h, ids, md, 9, 0, h->im);
}

extern "C" int atomdev_atm_bc_phi(atomdev_t* h) {
    ATOMDEV_REQUIRE(h && h->model == ATOMDEV_ATM, "atomdev_atm_bc_phi: not an atmosphere handle");
    ATOMDEV_TRY(hipSetDevice(h->device));
    // BC_Atm.cpp:115-150: rows j = 1 .. jm-2 only
    static const int ids[9] = { ATOMDEV_T, ATOMDEV_U, ATOMDEV_V, ATOMDEV_W, ATOMDEV_P_DYN, ATOMDEV_C, ATOMDEV_CLOUD, ATOMDEV_ICE, ATOMDEV_CO2 };
    return fill_phi(h, ids, 9, 0, h->im, 1, h->jm - 1);
}

extern "C" int atomdev_hyd_bc_radius(atomdev_t* h) {
    ATOMDEV_REQUIRE(h && h->model == ATOMDEV_HYD, "atomdev_hyd_bc_radius: not a hydrosphere handle");
    ATOMDEV_TRY(hipSetDevice(h->device));
    hipLaunchKernelGGL(k_hyd_bc_radius, dim3(nb(h->n2)), dim3(TPB), 0, h->stream, h->f[ATOMDEV_T], h->f[ATOMDEV_U],
                       h->f[ATOMDEV_V], h->f[ATOMDEV_W], h->f[ATOMDEV_P_DYN], h->f[ATOMDEV_C], h->p.ta, h->p.ca, h->im, h->jm, h->km);
    ATOMDEV_TRY(hipGetLastError());
    return 0;
}

extern "C" int atomdev_hyd_bc_radius_theta_phi(atomdev_t* h) {
    ATOMDEV_REQUIRE(h && h->model == ATOMDEV_HYD, "atomdev_hyd_bc_radius_theta_phi: not a hydrosphere handle");
    ATOMDEV_REQUIRE(h->jm >= 7 && h->km >= 7 && h->im >= 5, "atomdev_hyd_bc_radius_theta_phi: needs at least 5 levels, 7 rows and 7 columns");
    ATOMDEV_TRY(hipSetDevice(h->device));
    static const int ids[6] = { ATOMDEV_T, ATOMDEV_U, ATOMDEV_V, ATOMDEV_W, ATOMDEV_P_DYN, ATOMDEV_C };
    BcHyd b;
    for (int f = 0; f < 6; f++) b.x[f] = h->f[ids[f]];
    b.ta = h->p.ta; b.ca = h->p.ca;
    const unsigned nb_r = nb((size_t)(h->jm - 2) * (h->km - 2)), nb_t = nb((size_t)h->im * (h->km - 2)), nb_s = nb((size_t)h->im * h->jm);
    hipLaunchKernelGGL(k_hyd_bc_all, dim3(nb_r + nb_t + nb_s, 6), dim3(TPB), 0, h->stream, b, h->im, h->jm, h->km, nb_r, nb_t);
    return atomdev_check(hipGetLastError(), "k_hyd_bc_all");
}

extern "C" int atomdev_hyd_bc_theta(atomdev_t* h) {
    ATOMDEV_REQUIRE(h && h->model == ATOMDEV_HYD, "atomdev_hyd_bc_theta: not a hydrosphere handle");
    ATOMDEV_TRY(hipSetDevice(h->device));
    // BC_Hyd.cpp:90-118
    static const int ids[6] = { ATOMDEV_T, ATOMDEV_U, ATOMDEV_V, ATOMDEV_W, ATOMDEV_P_DYN, ATOMDEV_C };
    static const int md[6] = { FILL_NEUMANN, FILL_NEUMANN, FILL_ZERO, FILL_ZERO, FILL_NEUMANN, FILL_NEUMANN };
    return fill_theta(h, ids, md, 6, 0, h->im);
}

extern "C" int atomdev_hyd_bc_phi(atomdev_t* h) {
    ATOMDEV_REQUIRE(h && h->model == ATOMDEV_HYD, "atomdev_hyd_bc_phi: not a hydrosphere handle");
    ATOMDEV_TRY(hipSetDevice(h->device));
    // BC_Hyd.cpp:125-158: every j
    static const int ids[6] = { ATOMDEV_T, ATOMDEV_U, ATOMDEV_V, ATOMDEV_W, ATOMDEV_P_DYN, ATOMDEV_C };
    return fill_phi(h, ids, 6, 0, h->im, 0, h->jm);
}

extern "C" int atomdev_atm_pressure3d(atomdev_t* h) {
    ATOMDEV_REQUIRE(h && h->model == ATOMDEV_ATM, "atomdev_atm_pressure3d: not an atmosphere handle");
    ATOMDEV_TRY(hipSetDevice(h->device));
    return pressure3d(h, false);
}
extern "C" int atomdev_hyd_pressure3d(atomdev_t* h) {
    ATOMDEV_REQUIRE(h && h->model == ATOMDEV_HYD, "atomdev_hyd_pressure3d: not a hydrosphere handle");
    ATOMDEV_TRY(hipSetDevice(h->device));
    return pressure3d(h, true);
}
extern "C" int atomdev_atm_pressure2d(atomdev_t* h) {
    ATOMDEV_REQUIRE(h && h->model == ATOMDEV_ATM, "atomdev_atm_pressure2d: not an atmosphere handle");
    ATOMDEV_TRY(hipSetDevice(h->device));
    return pressure2d(h, false);
}
extern "C" int atomdev_hyd_pressure2d(atomdev_t* h) {
    ATOMDEV_REQUIRE(h && h->model == ATOMDEV_HYD, "atomdev_hyd_pressure2d: not a hydrosphere handle");
    ATOMDEV_TRY(hipSetDevice(h->device));
    return pressure2d(h, true);
}
