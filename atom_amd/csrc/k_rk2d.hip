// Surface (2-D) Runge-Kutta sweeps: solveRungeKutta_2D_Atmosphere + RK_RHS_2D_Atmosphere
// (atmosphere/RungeKutta_Atm.cpp:163-217, RHS_Atm.cpp:498-678) and the hydrosphere twins
// (hydrosphere/RungeKutta_Hyd.cpp:130-186, RHS_Hyd.cpp:453-653).
//
// Unlike the 3-D sweep, theta AND phi differences are live here, so the in-place
// lexicographic (j, k) sweep is a true 2-D Gauss-Seidel: point (j, k) reads the
// updated values at (j-1, k), (j-2, k), (j, k-1), (j, k-2) and the old values at
// (j+1, k), (j+2, k), (j, k+1), (j, k+2).  All points of an anti-diagonal
// j + k = d are mutually independent and see exactly those values if diagonals are
// processed in increasing d.  One workgroup walks the diagonals with a barrier
// between them (65 k points on the 1-degree grid: latency-bound, not worth more
// than one CU).
#include "atomdev_internal.h"

namespace {

struct Rk2dArgs {
    double *v, *w;                 // level base pointers (in/out)
    const double *vn, *wn, *p;     // level base pointers
    double *aux_v, *aux_w;
    const uint8_t* fl;             // flags of the level
    const double* sinthe; const double* costhe;
    double rm, rho;
    int hyd;
};

// One RHS evaluation at (j, k).  Statement order follows the reference because the
// branches overwrite each other (RHS_Atm.cpp:559-662).
__device__ void rhs2d(const Rk2dArgs& a, const DevConst& K, int j, int k, double vc, double wc, double out[2], double aux[2]) {
    const int jm = K.jm, km = K.km;
    const double dthe = K.dthe, dphi = K.dphi, re = K.re;
    const double dthe2 = K.dthe2, dphi2 = K.dphi2;
    const double rm = a.rm, rm2 = rm * rm;
    const double sinthe = a.sinthe[j], sinthe2 = sinthe * sinthe, costhe = a.costhe[j];
    const double rmsinthe = rm * sinthe, rm2sinthe = rm2 * sinthe, rm2sinthe2 = rm2 * sinthe2;
    const size_t q = (size_t)j * km + k;
    const size_t row = km;
    // the centre value is the caller's stage value; neighbours come from memory
#define V_(o) ((o) == 0 ? vc : a.v[q + (o)])
#define W_(o) ((o) == 0 ? wc : a.w[q + (o)])
#define P_(o) (a.p[q + (o)])
    auto LAND = [&](long o) { return (a.fl[q + o] & FL_LAND) != 0; };
    auto FLUID = [&](long o) { return (a.fl[q + o] & FL_LAND) == 0; };
    const bool centre_fluid = a.hyd ? ((a.fl[q] & FL_HZERO) != 0) : FLUID(0);
    const long R = (long)row;

    double h_0_j, h_d_j, h_0_k, h_d_k;
    if ((centre_fluid && LAND(R)) || (centre_fluid && LAND(-R))) { h_0_j = 1.; h_d_j = 0.; } else { h_0_j = 0.; h_d_j = 1.; }
    if ((centre_fluid && LAND(1)) || (centre_fluid && LAND(-1))) { h_0_k = 1.; h_d_k = 0.; } else { h_0_k = 0.; h_d_k = 1.; }

    double dvdthe = h_d_j * (a.v[q + R] - a.v[q - R]) / (2. * dthe);
    double dwdthe = h_d_j * (a.w[q + R] - a.w[q - R]) / (2. * dthe);
    double dpdthe = h_d_j * (P_(R) - P_(-R)) / (2. * dthe);
    double dvdphi = h_d_k * (a.v[q + 1] - a.v[q - 1]) / (2. * dphi);
    double dwdphi = h_d_k * (a.w[q + 1] - a.w[q - 1]) / (2. * dphi);
    double dpdphi = h_d_k * (P_(1) - P_(-1)) / (2. * dphi);
    double d2vdthe2 = h_d_j * (a.v[q + R] - 2. * vc + a.v[q - R]) / dthe2;
    double d2wdthe2 = h_d_j * (a.w[q + R] - 2. * wc + a.w[q - R]) / dthe2;
    double d2vdphi2 = h_d_k * (a.v[q + 1] - 2. * vc + a.v[q - 1]) / dphi2;
    double d2wdphi2 = h_d_k * (a.w[q + 1] - 2. * wc + a.w[q - 1]) / dphi2;

    if ((j >= 2) && (j < jm - 3)) {
        for (int s = +1; s >= -1; s -= 2) {
            const long o1 = s * R, o2 = 2 * s * R;
            if (LAND(0) && (FLUID(o1) && FLUID(o2))) {
                dvdthe = h_d_j * (-3. * vc + 4. * a.v[q + o1] - a.v[q + o2]) / (2. * dthe);
                dwdthe = h_d_j * (-3. * wc + 4. * a.w[q + o1] - a.w[q + o2]) / (2. * dthe);
                dpdthe = h_d_j * (-3. * P_(0) + 4. * P_(o1) - P_(o2)) / (2. * dthe);
                d2vdthe2 = h_d_j * (2. * vc - 2. * a.v[q + o1] + a.v[q + o2]) / dthe2;
                d2wdthe2 = h_d_j * (2. * wc - 2. * a.w[q + o1] + a.w[q + o2]) / dthe2;
            }
            if (LAND(0) && FLUID(o1)) {
                if (s > 0) {
                    dvdthe = h_d_j * (a.v[q + R] - vc) / dthe;
                    dwdthe = h_d_j * (a.w[q + R] - wc) / dthe;
                    dpdthe = h_d_j * (P_(R) - P_(0)) / dthe;
                } else {
                    dvdthe = h_d_j * (vc - a.v[q - R]) / dthe;
                    dwdthe = h_d_j * (wc - a.w[q - R]) / dthe;
                    dpdthe = h_d_j * (P_(0) - P_(-R)) / dthe;
                }
                d2vdthe2 = d2wdthe2 = 0.;
            }
        }
        d2vdthe2 = d2wdthe2 = 0.;
    }

    if ((k >= 2) && (k < km - 3)) {
        for (int s = +1; s >= -1; s -= 2) {
            const long o1 = s, o2 = 2 * s;
            if (LAND(0) && FLUID(o1) && FLUID(o2)) {
                dvdphi = h_d_k * (-3. * vc + 4. * a.v[q + o1] - a.v[q + o2]) / (2. * dphi);
                dwdphi = h_d_k * (-3. * wc + 4. * a.w[q + o1] - a.w[q + o2]) / (2. * dphi);
                dpdphi = h_d_k * (-3. * P_(0) + 4. * P_(o1) - P_(o2)) / (2. * dphi);
                // the reference assigns the theta second derivatives here (RHS_Atm.cpp:616-617,637-638)
                d2vdthe2 = h_d_k * (2. * vc - 2. * a.v[q + o1] + a.v[q + o2]) / dphi2;
                d2wdthe2 = h_d_k * (2. * wc - 2. * a.w[q + o1] + a.w[q + o2]) / dphi2;
            }
            if (LAND(0) && FLUID(o1)) {
                if (s > 0) {
                    dvdphi = h_d_k * (a.v[q + 1] - vc) / dphi;
                    dwdphi = h_d_k * (a.w[q + 1] - wc) / dphi;
                    dpdphi = h_d_k * (P_(1) - P_(0)) / dphi;
                } else {
                    dvdphi = h_d_k * (vc - a.v[q - 1]) / dphi;
                    dwdphi = h_d_k * (wc - a.w[q - 1]) / dphi;
                    dpdphi = h_d_k * (P_(0) - P_(-1)) / dphi;
                }
                d2vdphi2 = d2wdphi2 = 0.;
            }
        }
        d2vdphi2 = d2wdphi2 = 0.;
    } else {
        if (LAND(0) && FLUID(1)) {
            dvdphi = h_d_k * (a.v[q + 1] - vc) / dphi;
            dwdphi = h_d_k * (a.w[q + 1] - wc) / dphi;
            dpdphi = h_d_k * (P_(1) - P_(0)) / dphi;
        }
        if (centre_fluid && LAND(-1)) {
            dvdphi = h_d_k * (vc - a.v[q - 1]) / dphi;
            dwdphi = h_d_k * (wc - a.w[q - 1]) / dphi;
            dpdphi = h_d_k * (P_(0) - P_(-1)) / dphi;
        }
        d2vdthe2 = d2wdthe2 = 0.;
        d2vdphi2 = d2wdphi2 = 0.;
    }

    const double k_Force = 1.;
    const double rho = a.rho;
    // RHS_Atm.cpp:664-677 / RHS_Hyd.cpp:632-645
    double rhs_v = -(vc * dvdthe / rm + wc * dvdphi / rmsinthe)
                   + -h_d_j * dpdthe / rm / rho
                   - (d2vdthe2 / rm2 + dvdthe / rm2sinthe * costhe
                      - (1. + costhe * costhe / sinthe2) * h_d_j * vc
                      + d2vdphi2 / rm2sinthe2 - dwdphi * 2. * costhe / rm2sinthe2) / re
                   - h_0_j * vc * k_Force / dthe2;
    double rhs_w = -(vc * dwdthe / rm + wc * dwdphi / rmsinthe)
                   + -h_d_k * dpdphi / rmsinthe / rho
                   + (d2wdthe2 / rm2 + dwdthe / rm2sinthe * costhe
                      - (1. + costhe * costhe / sinthe2) * h_d_k * wc
                      + d2wdphi2 / rm2sinthe2 + dvdphi * 2. * costhe / rm2sinthe2) / re
                   - h_0_k * wc * k_Force / dphi2;
    aux[0] = rhs_v + h_d_j * dpdthe / rm / rho;
    aux[1] = rhs_w + h_d_k * dpdphi / rmsinthe / rho;
    out[0] = rhs_v; out[1] = rhs_w;
#undef V_
#undef W_
#undef P_
}

__global__ __launch_bounds__(1024) void k_rk2d(Rk2dArgs a, DevConst K) {
    const int jm = K.jm, km = K.km;
    const double dt = K.dt;
    const int dmax = (jm - 2) + (km - 2);
    for (int d = 2; d <= dmax; d++) {
        const int jlo = max(1, d - (km - 2)), jhi = min(jm - 2, d - 1);
        for (int j = jlo + (int)threadIdx.x; j <= jhi; j += blockDim.x) {
            const int k = d - j;
            const size_t q = (size_t)j * km + k;
            const double vn = a.vn[q], wn = a.wn[q];
            double vc = a.v[q], wc = a.w[q];
            double k1[2], k2[2], k3[2], k4[2], aux[2];
            rhs2d(a, K, j, k, vc, wc, k1, aux);
            vc = vn + k1[0] * .5 * dt; wc = wn + k1[1] * .5 * dt;
            rhs2d(a, K, j, k, vc, wc, k2, aux);
            vc = vn + k2[0] * .5 * dt; wc = wn + k2[1] * .5 * dt;
            rhs2d(a, K, j, k, vc, wc, k3, aux);
            vc = vn + k3[0] * dt; wc = wn + k3[1] * dt;
            rhs2d(a, K, j, k, vc, wc, k4, aux);
            a.v[q] = vn + dt * (k1[0] + 2. * k2[0] + 2. * k3[0] + k4[0]) / 6.;
            a.w[q] = wn + dt * (k1[1] + 2. * k2[1] + 2. * k3[1] + k4[1]) / 6.;
            a.aux_v[q] = aux[0]; a.aux_w[q] = aux[1];
        }
        __syncthreads();  // orders the global writes of diagonal d before the reads of d+1 within this workgroup
    }
}

int rk2d(atomdev* h, bool hyd) {
    if (int rc = atomdev_refresh_flags(h)) return rc;
    const int L = hyd ? h->im - 1 : 0;
    const size_t off = (size_t)L * h->n2;
    Rk2dArgs a;
    a.v = h->f[ATOMDEV_V] + off; a.w = h->f[ATOMDEV_W] + off;
    a.vn = h->f[ATOMDEV_VN] + off; a.wn = h->f[ATOMDEV_WN] + off; a.p = h->f[ATOMDEV_P_DYN] + off;
    a.aux_v = h->f[ATOMDEV_AUX_V] + off; a.aux_w = h->f[ATOMDEV_AUX_W] + off;
    a.fl = h->flags + off;
    a.sinthe = h->sinthe; a.costhe = h->costhe;
    double rm = h->p.r0;  // rad[L] by repeated addition
    for (int l = 1; l <= L; l++) rm = rm + h->p.dr;
    a.rm = rm;
    a.rho = hyd ? h->p.r_0_water : h->p.r_air;
    a.hyd = hyd ? 1 : 0;
    hipLaunchKernelGGL(k_rk2d, dim3(1), dim3(1024), 0, h->stream, a, h->k);
    return atomdev_check(hipGetLastError(), "k_rk2d");
}

}  // namespace

extern "C" int atomdev_atm_rk2d(atomdev_t* h) {
    ATOMDEV_REQUIRE(h && h->model == ATOMDEV_ATM, "atomdev_atm_rk2d: not an atmosphere handle");
    ATOMDEV_TRY(hipSetDevice(h->device));
    return rk2d(h, false);
}

extern "C" int atomdev_hyd_rk2d(atomdev_t* h) {
    ATOMDEV_REQUIRE(h && h->model == ATOMDEV_HYD, "atomdev_hyd_rk2d: not a hydrosphere handle");
    // RK_RHS_2D_Hydrosphere sets im = 41 inside the RHS object (hydrosphere/RHS_Hyd.cpp:466) while
    // the RK driver uses the grid's im-1: on any other grid the reference mixes two levels.
    ATOMDEV_REQUIRE(h->im == 41, "atomdev_hyd_rk2d: the reference hard-codes im = 41 for the surface sweep");
    ATOMDEV_TRY(hipSetDevice(h->device));
    return rk2d(h, true);
}
