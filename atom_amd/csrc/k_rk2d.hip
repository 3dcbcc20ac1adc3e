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
    double i_rm, i_rm2, i_rho, i_two_dphi, i_dphi;  // correctly rounded reciprocals (host)
    int hyd;
};

// per-row geometry (LDS): every divisor of the right-hand side with its correctly rounded reciprocal
struct RowGeo { double costhe, rmsin, i_rmsin, rm2sin, i_rm2sin, rm2sin2, i_rm2sin2, coef; };

// A finite difference of the reference in the one form all its variants share,
//     h * ((cc * centre + c1 * nA) + c2 * nB) / div,
// with small exact coefficients: multiplying by 0, +-1, +-2 is exact and a - b == a + (-b), so every
// variant keeps its bits (up to the sign of a zero).  Which variant applies depends on the land mask
// only, so the coefficients are chosen once per point and the four stages evaluate one formula.
struct Diff { double cc, c1, c2, h, div, idiv; };
__device__ __forceinline__ double diff(const Diff& D, double centre, double nA, double nB) {
    return div_by(D.h * ((D.cc * centre + D.c1 * nA) + D.c2 * nB), D.div, D.idiv);
}

// What survives of RK_RHS_2D_*'s cascade of overwriting branches (RHS_Atm.cpp:559-662, RHS_Hyd.cpp:516-630;
// L = land centre, f(o) = fluid neighbour at offset o, "window" = index in [2, n-4]):
//   theta first derivatives (v, w, p):  window_j && L && f(-R): backward two-point; else window_j && L && f(+R):
//       forward two-point; else central.  (The three-point one-sided forms are always overwritten by the
//       two-point ones that follow them.)
//   phi first derivatives: inside window_k the same with +-1; outside it, forward if L && f(+1), then
//       backward if the centre is fluid and (k-1) is land.
//   phi second derivatives: zeroed on every path.
//   theta second derivatives: central outside window_j, zero inside; then, inside window_k, replaced by the
//       PHI one-sided form h_d_k (2 x - 2 x(k+-1) + x(k+-2)) / dphi2 where L && f(+-1) && f(+-2) (the reference
//       assigns d2vdthe2 there: RHS_Atm.cpp:616-617, 637-638; the s = -1 side wins); outside window_k zero.
__global__ __launch_bounds__(256) void k_rk2d(Rk2dArgs a, DevConst K) {
    extern __shared__ RowGeo geo[];  // [jm]
    const int jm = K.jm, km = K.km;
    const double dt = K.dt;
    const double rm = a.rm, rm2 = rm * rm;
    {
        for (int j = threadIdx.x; j < jm; j += blockDim.x) {
            const double sinthe = a.sinthe[j], sinthe2 = sinthe * sinthe, costhe = a.costhe[j];
            RowGeo g;
            g.costhe = costhe;
            g.rmsin = rm * sinthe; g.rm2sin = rm2 * sinthe; g.rm2sin2 = rm2 * sinthe2;
            g.i_rmsin = 1. / g.rmsin; g.i_rm2sin = 1. / g.rm2sin; g.i_rm2sin2 = 1. / g.rm2sin2;
            g.coef = 1. + costhe * costhe / sinthe2;
            geo[j] = g;
        }
        __syncthreads();
    }
    const long R = km;
    const int dmax = (jm - 2) + (km - 2);
    for (int d = 2; d <= dmax; d++) {
        const int jlo = max(1, d - (km - 2)), jhi = min(jm - 2, d - 1);
        for (int j = jlo + (int)threadIdx.x; j <= jhi; j += blockDim.x) {
            const int k = d - j;
            const size_t q = (size_t)j * km + k;
            const bool win_j = (j >= 2) && (j < jm - 3), win_k = (k >= 2) && (k < km - 3);
            // ---- neighbourhood, read once: updated values on the side already swept, old ones ahead ----
            const double v_tp = a.v[q + R], v_tm = a.v[q - R], v_pp = a.v[q + 1], v_pm = a.v[q - 1];
            const double w_tp = a.w[q + R], w_tm = a.w[q - R], w_pp = a.w[q + 1], w_pm = a.w[q - 1];
            const double p_tp = a.p[q + R], p_tm = a.p[q - R], p_pp = a.p[q + 1], p_pm = a.p[q - 1], p_c = a.p[q];
            const uint8_t fc = a.fl[q];
            const bool L = (fc & FL_LAND) != 0;
            const bool fluid_c = a.hyd ? ((fc & FL_HZERO) != 0) : !L;
            const bool l_tp = (a.fl[q + R] & FL_LAND) != 0, l_tm = (a.fl[q - R] & FL_LAND) != 0;
            const bool l_pp = (a.fl[q + 1] & FL_LAND) != 0, l_pm = (a.fl[q - 1] & FL_LAND) != 0;
            bool l_pp2 = true, l_pm2 = true;
            if (win_k) { l_pp2 = (a.fl[q + 2] & FL_LAND) != 0; l_pm2 = (a.fl[q - 2] & FL_LAND) != 0; }
            const double vn = a.vn[q], wn = a.wn[q];
            double vc = a.v[q], wc = a.w[q];
            const RowGeo G = geo[j];

            double h_0_j, h_d_j, h_0_k, h_d_k;
            if (fluid_c && (l_tp || l_tm)) { h_0_j = 1.; h_d_j = 0.; } else { h_0_j = 0.; h_d_j = 1.; }
            if (fluid_c && (l_pp || l_pm)) { h_0_k = 1.; h_d_k = 0.; } else { h_0_k = 0.; h_d_k = 1.; }

            // ---- the surviving variant of every difference ----
            Diff T1 = { 0., 1., -1., h_d_j, K.two_dthe, K.i_two_dthe };  // central
            if (win_j && L && !l_tp) T1 = { -1., 1., 0., h_d_j, K.dthe, K.i_dthe };   // forward two-point
            if (win_j && L && !l_tm) T1 = { 1., 0., -1., h_d_j, K.dthe, K.i_dthe };   // backward two-point (wins)
            Diff P1 = { 0., 1., -1., h_d_k, K.two_dphi, a.i_two_dphi };
            if (win_k) {
                if (L && !l_pp) P1 = { -1., 1., 0., h_d_k, K.dphi, a.i_dphi };
                if (L && !l_pm) P1 = { 1., 0., -1., h_d_k, K.dphi, a.i_dphi };
            } else {
                if (L && !l_pp) P1 = { -1., 1., 0., h_d_k, K.dphi, a.i_dphi };
                if (fluid_c && l_pm) P1 = { 1., 0., -1., h_d_k, K.dphi, a.i_dphi };
            }
            // theta second derivative and its two neighbour operands (theta neighbours, or phi neighbours in the one-sided form)
            Diff T2 = { 0., 0., 0., h_d_j, K.dthe2, K.i_dthe2 };  // zero
            double v2A = v_tp, v2B = v_tm, w2A = w_tp, w2B = w_tm;
            if (!win_j) T2 = { -2., 1., 1., h_d_j, K.dthe2, K.i_dthe2 };  // central
            if (win_k) {
                if (L && !l_pp && !l_pp2) {
                    T2 = { 2., -2., 1., h_d_k, K.dphi2, K.i_dphi2 };
                    v2A = v_pp; v2B = a.v[q + 2]; w2A = w_pp; w2B = a.w[q + 2];
                }
                if (L && !l_pm && !l_pm2) {
                    T2 = { 2., -2., 1., h_d_k, K.dphi2, K.i_dphi2 };
                    v2A = v_pm; v2B = a.v[q - 2]; w2A = w_pm; w2B = a.w[q - 2];
                }
            } else {
                T2 = { 0., 0., 0., h_d_j, K.dthe2, K.i_dthe2 };
            }
            // pressure gradient terms do not change during the stages
            const double dpdthe = diff(T1, p_c, p_tp, p_tm), dpdphi = diff(P1, p_c, p_pp, p_pm);
            const double pg_v = div_by(div_by(h_d_j * dpdthe, rm, a.i_rm), a.rho, a.i_rho);         // h_d_j * dpdthe / rm / rho
            const double pg_w = div_by(div_by(h_d_k * dpdphi, G.rmsin, G.i_rmsin), a.rho, a.i_rho);

            double kv[4], kw[4];
#pragma unroll
            for (int s = 0; s < 4; s++) {
                const double dvdthe = diff(T1, vc, v_tp, v_tm), dwdthe = diff(T1, wc, w_tp, w_tm);
                const double dvdphi = diff(P1, vc, v_pp, v_pm), dwdphi = diff(P1, wc, w_pp, w_pm);
                const double d2vdthe2 = diff(T2, vc, v2A, v2B), d2wdthe2 = diff(T2, wc, w2A, w2B);
                // RHS_Atm.cpp:664-677 / RHS_Hyd.cpp:632-645 (k_Force = 1; the phi second derivatives are zero)
                kv[s] = -(div_by(vc * dvdthe, rm, a.i_rm) + div_by(wc * dvdphi, G.rmsin, G.i_rmsin))
                        + -pg_v
                        - div_by(div_by(d2vdthe2, rm2, a.i_rm2) + div_by(dvdthe, G.rm2sin, G.i_rm2sin) * G.costhe
                                 - G.coef * h_d_j * vc
                                 - div_by(dwdphi * 2. * G.costhe, G.rm2sin2, G.i_rm2sin2), K.re, K.i_re)
                        - div_by(h_0_j * vc, K.dthe2, K.i_dthe2);
                kw[s] = -(div_by(vc * dwdthe, rm, a.i_rm) + div_by(wc * dwdphi, G.rmsin, G.i_rmsin))
                        + -pg_w
                        + div_by(div_by(d2wdthe2, rm2, a.i_rm2) + div_by(dwdthe, G.rm2sin, G.i_rm2sin) * G.costhe
                                 - G.coef * h_d_k * wc
                                 + div_by(dvdphi * 2. * G.costhe, G.rm2sin2, G.i_rm2sin2), K.re, K.i_re)
                        - div_by(h_0_k * wc, K.dphi2, K.i_dphi2);
                if (s < 2) { vc = vn + kv[s] * .5 * dt; wc = wn + kw[s] * .5 * dt; }
                else if (s == 2) { vc = vn + kv[s] * dt; wc = wn + kw[s] * dt; }
            }
            a.v[q] = vn + div_by(dt * (kv[0] + 2. * kv[1] + 2. * kv[2] + kv[3]), 6., K.i_six);
            a.w[q] = wn + div_by(dt * (kw[0] + 2. * kw[1] + 2. * kw[2] + kw[3]), 6., K.i_six);
            a.aux_v[q] = kv[3] + pg_v; a.aux_w[q] = kw[3] + pg_w;
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
    a.i_rm = 1. / a.rm; a.i_rm2 = 1. / (a.rm * a.rm); a.i_rho = 1. / a.rho;
    a.i_two_dphi = 1. / h->k.two_dphi; a.i_dphi = 1. / h->k.dphi;
    hipLaunchKernelGGL(k_rk2d, dim3(1), dim3(256), (size_t)h->jm * sizeof(RowGeo), h->stream, a, h->k);
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
