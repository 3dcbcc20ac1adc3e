// Hydrosphere 3-D Runge-Kutta sweep: RungeKutta_Hydrosphere::solveRungeKutta_3D_Hydrosphere
// (hydrosphere/RungeKutta_Hyd.cpp:30-124) fused with RHS_Hydrosphere::RK_RHS_3D_Hydrosphere
// (hydrosphere/RHS_Hyd.cpp:60-447).
//
// Same structure as the atmosphere sweep (k_rk_atm.hip): an in-place
// lexicographic Gauss-Seidel sweep whose phi-derivatives are all multiplied by
// h_d_k == 0, so each (i, k) theta line only needs its own updated rows j-1,
// j-2.  Differences to the atmosphere:
//   * radial FIRST derivatives are live, as forward differences to level i+1
//     (RHS_Hyd.cpp:206-211), which the lexicographic order sees un-updated.
//     Lines of different levels therefore must not overwrite each other's
//     inputs: the sweep reads the current buffers and writes a second set, and
//     the handle swaps the two afterwards.
//   * level im-2 reads the ALREADY UPDATED salinity of level im-3
//     (RHS_Hyd.cpp:356-357), so it runs as a second launch that reads that
//     level from the output buffer.
//   * stage 3 advances w from its stage-2 value (RungeKutta_Hyd.cpp:101).
//   * land cells overwrite r_water / r_salt_water during the first RHS
//     evaluation and the later stages read the overwritten values (RHS_Hyd.cpp:395-402).
#include "atomdev_internal.h"

namespace {

struct HydRkArgs {
    const double *t, *u, *v, *w, *c;        // current state (read)
    double *to, *uo, *vo, *wo, *co;         // next state (written)
    const double *tn, *un, *vn, *wn, *cn;
    const double* p_dyn;
    double *r_water, *r_salt_water;         // read, rewritten on land
    double *aux_u, *aux_v, *aux_w, *buoy, *salt_balance, *salt_finger, *salt_diffusion;
    const uint8_t* flags;
    const double *bathy, *evap, *precip;
    const double *rad, *sinthe, *costhe;
    int i_begin;                            // first level of this launch
};

__device__ __forceinline__ void dtheta_h(int mode, double hdj, double xm2, double xm1, double x0, double xp1,
                                         const DevConst& K, double& d1, double& d2) {
    if (mode == 0) {  // RHS_Hyd.cpp:156-161, 177-181
        d1 = div_by(hdj * (xp1 - xm1), K.two_dthe, K.i_two_dthe);
        d2 = div_by(hdj * (xp1 - 2. * x0 + xm1), K.dthe2, K.i_dthe2);
    } else if (mode == 1) {  // :269-276
        d1 = div_by(hdj * (x0 - xm1), K.dthe, K.i_dthe);
        d2 = 0.;
    } else {  // :250-267
        d1 = div_by(hdj * (-3. * x0 + 4. * xm1 - xm2), K.two_dthe, K.i_two_dthe);
        d2 = div_by(hdj * (2 * x0 - 2. * xm1 + xm2), K.dthe2, K.i_dthe2);
    }
}

__global__ __launch_bounds__(64) void k_hyd_rk3d(HydRkArgs a, DevConst K) {
    const int im = K.im, jm = K.jm, km = K.km;
    const int k = 1 + blockIdx.x * 64 + threadIdx.x;
    const int i = a.i_begin + blockIdx.y;
    if (k > km - 2) return;

    const double rm = a.rad[i], rm2 = rm * rm;
    const double i_rm = 1. / rm, i_rm2 = 1. / rm2;
    const double hight = (double)i * K.topo_step;
    const size_t row = (size_t)km, lvl = (size_t)jm * km;
    const bool kwin = (k >= 2) && (k < km - 3);
    const bool top = (i == im - 2);
    size_t p = idx3(jm, km, i, 0, k);

    const double* X[5] = { a.t, a.u, a.v, a.w, a.c };
    double* XO[5] = { a.to, a.uo, a.vo, a.wo, a.co };
    const double* XN[5] = { a.tn, a.un, a.vn, a.wn, a.cn };
    double xm2[5], xm1[5], x0[5], xp1[5];
    double pm2 = 0., pm1, p0, pp1;
#pragma unroll
    for (int n = 0; n < 5; n++) { xm2[n] = 0.; xm1[n] = X[n][p]; x0[n] = X[n][p + row]; xp1[n] = X[n][p + 2 * row]; }
    pm1 = a.p_dyn[p]; p0 = a.p_dyn[p + row]; pp1 = a.p_dyn[p + 2 * row];

    for (int j = 1; j < jm - 1; j++) {
        p += row;
        const size_t q = (size_t)j * km + k;
        const double sinthe = a.sinthe[j], costhe = a.costhe[j];
        const double sinthe2 = sinthe * sinthe;
        const double rm2sinthe = rm2 * sinthe;
        const double i_rm2sinthe = 1. / rm2sinthe;
        const double coef_v = 1. + costhe * costhe / sinthe2;  // RHS_Hyd.cpp:422 (no rm here)

        const uint8_t fl = a.flags[p];
        const bool land = fl & FL_LAND;
        const double h_0_j = (fl & FL_H0J) ? 1. : 0.;
        const double h_0_k = (fl & FL_H0K) ? 1. : 0.;
        const double h_d_j = ((fl & FL_H0J) && (fl & FL_H0K)) ? 0. : 1.;  // RHS_Hyd.cpp:117-139
        double h_0_i = 1.;                                                 // RHS_Hyd.cpp:96-111
        if (land) {
            double topo_diff = fabs(hight - a.bathy[q]);
            if (!(topo_diff == K.topo_step)) h_0_i = div_by(topo_diff, K.topo_step, K.i_topo_step);
        }
        const bool window = (j >= 2) && (j < jm - 3);
        const int mode = !window ? 0 : ((fl & FL_ONESIDED_S) ? 2 : 1);
        const bool d2zero = kwin && (fl & FL_D2ZERO);          // RHS_Hyd.cpp:296-300, 326-330
        const bool rlive = (i < im - 2) && !(fl & FL_RDEAD);   // RHS_Hyd.cpp:191-214

        double xn[5], xa[5];
#pragma unroll
        for (int n = 0; n < 5; n++) { xn[n] = XN[n][p]; xa[n] = rlive ? X[n][p + lvl] : 0.; }
        double rw = a.r_water[p], rsw = a.r_salt_water[p];

        double dpdthe, d2p_unused;
        dtheta_h(mode, h_d_j, pm2, pm1, p0, pp1, K, dpdthe, d2p_unused);
        const double pg_num = div_by(h_d_j * dpdthe, rm, i_rm);  // h_d_j*dpdthe/rm, then / salt_water_ref per stage

        double c_top = 0., c_below = 0., ep = 0.;
        if (top) {  // RHS_Hyd.cpp:355-359
            c_top = a.c[p + lvl];      // level im-1: boundary, never updated
            c_below = a.co[p - lvl];   // level im-3: already updated
            ep = a.evap[q] - a.precip[q];
        }

        double xp2[5], pp2 = 0.;
        if (j + 2 < jm) {
#pragma unroll
            for (int n = 0; n < 5; n++) xp2[n] = X[n][p + 2 * row];
            pp2 = a.p_dyn[p + 2 * row];
        } else {
#pragma unroll
            for (int n = 0; n < 5; n++) xp2[n] = 0.;
        }

        double st[5], acc[5], rhs[5];
        double RS = 0., swr = 0., pg_v = 0., sb = 0.;
#pragma unroll
        for (int n = 0; n < 5; n++) st[n] = x0[n];

#pragma unroll
        for (int s = 0; s < 4; s++) {
            // order: t u v w c
            const double u = st[1], v = st[2];
            double d1[5], d2[5], dr1[5];
#pragma unroll
            for (int n = 0; n < 5; n++) {
                dtheta_h(mode, h_d_j, xm2[n], xm1[n], st[n], xp1[n], K, d1[n], d2[n]);
                if (d2zero) d2[n] = 0.;
                dr1[n] = rlive ? div_by(xa[n] - st[n], K.dr, K.i_dr) : 0.;
            }

            // RHS_Hyd.cpp:345-382
            swr = rw + .7 * st[4] * K.c_0;  // salt_water_ref
            double salinity_evaporation = 0.;
            if (top) {
                double salinity_surface = div_by(swr * (-3. * c_top + 4. * st[4] - c_below), K.two_dr, K.i_two_dr)
                                          * (1. - 2. * c_top) * ep;
                salinity_evaporation = K.coeff_salinity * salinity_surface;
                if (land) salinity_evaporation = 0.;
                if (salinity_evaporation >= 20.) salinity_evaporation = 20.;
                if (salinity_evaporation <= -20.) salinity_evaporation = -20.;
            }
            RS = K.buoy_g * (rsw - swr) / swr;
            sb = swr - rsw;
            if (land) { rw = 1007.; rsw = 1032.; }  // seen by the following stages

            pg_v = pg_num / swr;
#define ADV(n) (u * dr1[n] + div_by(v * d1[n], rm, i_rm))
            // RHS_Hyd.cpp:406-437 with the exact-zero terms removed
            rhs[0] = -ADV(0)
                     + div_by(div_by(dr1[0] * 2., rm, i_rm) + div_by(d2[0], rm2, i_rm2) + div_by(d1[0] * costhe, rm2sinthe, i_rm2sinthe), K.re_pr, K.i_re_pr)
                     - div_by(h_0_i * st[0], K.dr2, K.i_dr2);
            rhs[1] = -ADV(1)
                     + div_by(div_by(d2[1], rm2, i_rm2) + div_by(4. * dr1[1], rm, i_rm) + div_by(d1[1] * costhe, rm2sinthe, i_rm2sinthe), K.re, K.i_re)
                     + RS
                     - div_by(h_0_i * u, K.dthe2, K.i_dthe2);
            rhs[2] = -ADV(2) - pg_v
                     + div_by(div_by(dr1[2] * 2., rm, i_rm) + div_by(d2[2], rm2, i_rm2) + div_by(d1[2], rm2sinthe, i_rm2sinthe) * costhe
                              - coef_v * h_d_j * v + div_by(2. * d1[1], rm2, i_rm2), K.re, K.i_re)
                     - div_by(h_0_j * v, K.dthe2, K.i_dthe2);
            rhs[3] = -ADV(3)
                     + div_by(div_by(dr1[3] * 2., rm, i_rm) + div_by(d2[3], rm2, i_rm2) + div_by(d1[3], rm2sinthe, i_rm2sinthe) * costhe, K.re, K.i_re)
                     - div_by(h_0_k * st[3], K.dphi2, K.i_dphi2);
            rhs[4] = -ADV(4)
                     + div_by(div_by(dr1[4] * 2., rm, i_rm) + div_by(d2[4], rm2, i_rm2) + div_by(d1[4] * costhe, rm2sinthe, i_rm2sinthe), K.re_sc, K.i_re_sc);
            if (top) rhs[4] = rhs[4] + salinity_evaporation;
            rhs[4] = rhs[4] - div_by(h_0_i * st[4], K.dr2, K.i_dr2);
#undef ADV

            // RungeKutta_Hyd.cpp:56-120
            if (s == 0) {
#pragma unroll
                for (int n = 0; n < 5; n++) { acc[n] = rhs[n]; st[n] = xn[n] + rhs[n] * .5 * K.dt; }
            } else if (s == 1) {
#pragma unroll
                for (int n = 0; n < 5; n++) { acc[n] = acc[n] + 2. * rhs[n]; st[n] = xn[n] + rhs[n] * .5 * K.dt; }
            } else if (s == 2) {
#pragma unroll
                for (int n = 0; n < 5; n++) {
                    acc[n] = acc[n] + 2. * rhs[n];
                    const double base = (n == 3) ? st[3] : xn[n];  // w = w + kw3*dt
                    st[n] = base + rhs[n] * K.dt;
                }
            } else {
#pragma unroll
                for (int n = 0; n < 5; n++) st[n] = xn[n] + div_by(K.dt * (acc[n] + rhs[n]), 6., K.i_six);
            }
        }

        // side outputs of the 4th evaluation (RHS_Hyd.cpp:379-402, 440-446)
        double au = rhs[1], av = rhs[2] + pg_v, aw = rhs[3];
        double bf = RS * K.coeff_buoy * 1000.;
        double sfinger, sdiff;
        if (sb < 0.) { sdiff = sb; sfinger = 0.; } else { sfinger = sb; sdiff = 0.; }
        if (land) {
            au = 0.; av = 0.; aw = 0.; bf = 0.; sb = 0.; sfinger = 0.; sdiff = 0.;
            a.r_water[p] = 1007.; a.r_salt_water[p] = 1032.;
        }
        a.aux_u[p] = au; a.aux_v[p] = av; a.aux_w[p] = aw; a.buoy[p] = bf;
        a.salt_balance[p] = sb; a.salt_finger[p] = sfinger; a.salt_diffusion[p] = sdiff;
#pragma unroll
        for (int n = 0; n < 5; n++) {
            XO[n][p] = st[n];
            xm2[n] = xm1[n]; xm1[n] = st[n]; x0[n] = xp1[n]; xp1[n] = xp2[n];
        }
        pm2 = pm1; pm1 = p0; p0 = pp1; pp1 = pp2;
    }
}

}  // namespace

extern "C" int atomdev_hyd_rk3d(atomdev_t* h) {
    ATOMDEV_REQUIRE(h && h->model == ATOMDEV_HYD, "atomdev_hyd_rk3d: not a hydrosphere handle");
    ATOMDEV_TRY(hipSetDevice(h->device));
    if (int rc = atomdev_refresh_flags(h)) return rc;
    if (h->rk_variant == 3) return atomdev_hyd_rk3d_pipe(h);
    if (h->rk_variant == 2) return atomdev_hyd_rk3d_ws_fast(h);
    if (h->rk_variant == 0) return atomdev_hyd_rk3d_ws(h);
    static const int X[5] = { ATOMDEV_T, ATOMDEV_U, ATOMDEV_V, ATOMDEV_W, ATOMDEV_C };
    // the sweep only writes interior points: carry the boundary values over first
    for (int n = 0; n < 5; n++)
        ATOMDEV_TRY(hipMemcpyAsync(h->alt[X[n]], h->f[X[n]], h->n3 * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    HydRkArgs a;
    a.t = h->f[ATOMDEV_T]; a.u = h->f[ATOMDEV_U]; a.v = h->f[ATOMDEV_V]; a.w = h->f[ATOMDEV_W]; a.c = h->f[ATOMDEV_C];
    a.to = h->alt[ATOMDEV_T]; a.uo = h->alt[ATOMDEV_U]; a.vo = h->alt[ATOMDEV_V]; a.wo = h->alt[ATOMDEV_W]; a.co = h->alt[ATOMDEV_C];
    a.tn = h->f[ATOMDEV_TN]; a.un = h->f[ATOMDEV_UN]; a.vn = h->f[ATOMDEV_VN]; a.wn = h->f[ATOMDEV_WN]; a.cn = h->f[ATOMDEV_CN];
    a.p_dyn = h->f[ATOMDEV_P_DYN];
    a.r_water = h->f[ATOMDEV_R_WATER]; a.r_salt_water = h->f[ATOMDEV_R_SALT_WATER];
    a.aux_u = h->f[ATOMDEV_AUX_U]; a.aux_v = h->f[ATOMDEV_AUX_V]; a.aux_w = h->f[ATOMDEV_AUX_W]; a.buoy = h->f[ATOMDEV_BUOYANCY];
    a.salt_balance = h->f[ATOMDEV_SALT_BALANCE]; a.salt_finger = h->f[ATOMDEV_SALT_FINGER]; a.salt_diffusion = h->f[ATOMDEV_SALT_DIFFUSION];
    a.flags = h->flags;
    a.bathy = h->g[ATOMDEV_TOPOGRAPHY]; a.evap = h->g[ATOMDEV_EVAPORATION_DALTON]; a.precip = h->g[ATOMDEV_PRECIPITATION];
    a.rad = h->rad; a.sinthe = h->sinthe; a.costhe = h->costhe;
    const unsigned gx = (h->km - 2 + 63) / 64;
    if (int rc = atomdev_prof_begin(h)) return rc;
    if (h->im - 3 >= 1) {  // levels 1 .. im-3
        a.i_begin = 1;
        hipLaunchKernelGGL(k_hyd_rk3d, dim3(gx, h->im - 3), dim3(64), 0, h->stream, a, h->k);
        ATOMDEV_TRY(hipGetLastError());
    }
    a.i_begin = h->im - 2;  // level im-2, after level im-3 is complete
    hipLaunchKernelGGL(k_hyd_rk3d, dim3(gx, 1), dim3(64), 0, h->stream, a, h->k);
    ATOMDEV_TRY(hipGetLastError());
    if (int rc = atomdev_prof_end(h)) return rc;
    for (int n = 0; n < 5; n++) { double* t = h->f[X[n]]; h->f[X[n]] = h->alt[X[n]]; h->alt[X[n]] = t; }
    return 0;
}
