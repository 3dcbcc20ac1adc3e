// Atmosphere 3-D Runge-Kutta sweep: RungeKutta_Atmosphere::solveRungeKutta_3D_Atmosphere
// (atmosphere/RungeKutta_Atm.cpp:32-156) fused with RHS_Atmosphere::RK_RHS_3D_Atmosphere
// (atmosphere/RHS_Atm.cpp:68-493).
//
// What the reference computes.  The sweep visits (i, j, k) lexicographically and
// runs all four stages at one point before moving on, updating the fields in
// place: a nonlinear Gauss-Seidel sweep.  In the RHS every r- and phi-derivative
// is multiplied by h_d_i == 0 / h_d_k == 0 (RHS_Atm.cpp:121-124, 146-149), so a
// point depends only on its own theta line: on the already-updated rows j-1, j-2
// and on the not-yet-updated rows j+1 (pole-side rows only).  The single
// cross-line read is c at level 3 for the points of level 1 (RHS_Atm.cpp:420),
// which the lexicographic order sees un-updated; the launcher snapshots that
// plane first.
//
// Mapping.  One thread marches one (i, k) theta line, j = 1 .. jm-2, in place,
// carrying the new values of rows j-1 and j-2 in registers.  Lanes run along k
// (phi, the fastest index), so every load and store of a wave is one contiguous
// 512-byte row segment; i is uniform per workgroup, so radius and latitude
// factors are wave-uniform.  Terms the reference multiplies by an exact zero are
// dropped (x + (+-0) == x for the finite values of the model); the surviving
// terms keep the reference's association order, and nothing is contracted to
// FMA except inside div_by (exact quotient by a loop-invariant divisor).
#include "atomdev_internal.h"

namespace {

struct AtmRkArgs {
    double *t, *u, *v, *w, *c, *cloud, *ice, *co2;             // in/out
    const double *tn, *un, *vn, *wn, *cn, *cloudn, *icen, *co2n;
    const double* p_dyn;
    const double *S_v, *S_c, *S_i, *S_r, *S_s;
    double *aux_u, *aux_v, *aux_w, *buoy;
    const uint8_t* flags;
    const double *topo, *evap, *precip, *c3;                   // 2-D: Topography, Evaporation_Dalton, Precipitation, c[3] snapshot
    const double *rad, *sinthe, *costhe;
};

// theta derivative pair of one field at row j (RHS_Atm.cpp:162-170, 192-199, 245-312).
// mode 0: pole-side rows (j = 1, jm-3, jm-2): central; mode 1: backward two-point;
// mode 2: land cell with two fluid cells to the north: one-sided three-point.
__device__ __forceinline__ void dtheta(int mode, double hdj, double xm2, double xm1, double x0, double xp1,
                                       const DevConst& K, double& d1, double& d2) {
    if (mode == 0) {
        d1 = div_by(hdj * (xp1 - xm1), K.two_dthe, K.i_two_dthe);
        d2 = div_by(hdj * (xp1 - 2. * x0 + xm1), K.dthe2, K.i_dthe2);
    } else if (mode == 1) {
        d1 = div_by(hdj * (x0 - xm1), K.dthe, K.i_dthe);
        d2 = 0.;
    } else {
        d1 = div_by(hdj * (-3. * x0 + 4. * xm1 - xm2), K.two_dthe, K.i_two_dthe);
        d2 = div_by(hdj * (2 * x0 - 2. * xm1 + xm2), K.dthe2, K.i_dthe2);
    }
}

__global__ __launch_bounds__(64) void k_atm_rk3d(AtmRkArgs a, DevConst K) {
    const int im = K.im, jm = K.jm, km = K.km;
    const int k = 1 + blockIdx.x * 64 + threadIdx.x;
    const int i = 1 + blockIdx.y;
    if (k > km - 2) return;

    const double rm = a.rad[i], rm2 = rm * rm;
    const double i_rm = 1. / rm, i_rm2 = 1. / rm2;
    const double hight = (double)i * K.topo_step;
    const size_t row = (size_t)km;
    size_t p = idx3(jm, km, i, 0, k);  // (i, j, k), advanced by one row per step

    // rows j-2, j-1 (updated), j (current), j+1 (old) of the 8 prognostic fields and p_dyn
    double *X[8] = { a.t, a.u, a.v, a.w, a.c, a.cloud, a.ice, a.co2 };
    const double* XN[8] = { a.tn, a.un, a.vn, a.wn, a.cn, a.cloudn, a.icen, a.co2n };
    double xm2[8], xm1[8], x0[8], xp1[8];
    double pm2 = 0., pm1, p0, pp1;
#pragma unroll
    for (int n = 0; n < 8; n++) { xm2[n] = 0.; xm1[n] = X[n][p]; x0[n] = X[n][p + row]; xp1[n] = X[n][p + 2 * row]; }
    pm1 = a.p_dyn[p]; p0 = a.p_dyn[p + row]; pp1 = a.p_dyn[p + 2 * row];

    for (int j = 1; j < jm - 1; j++) {
        p += row;
        const size_t q = (size_t)j * km + k;  // 2-D index
        const double sinthe = a.sinthe[j], costhe = a.costhe[j];
        const double sinthe2 = sinthe * sinthe;
        const double rm2sinthe = rm2 * sinthe;
        const double i_rm2sinthe = 1. / rm2sinthe;
        const double coef_v = 1. + costhe * costhe / (rm * sinthe2);  // RHS_Atm.cpp:450

        const uint8_t fl = a.flags[p];
        const bool land = fl & FL_LAND;
        const double h_0_j = (fl & FL_H0J) ? 1. : 0.;
        const double h_0_k = (fl & FL_H0K) ? 1. : 0.;
        // h_d_j = 0 only for a cell that is both a theta- and a phi-coast (RHS_Atm.cpp:129-149)
        const double h_d_j = ((fl & FL_H0J) && (fl & FL_H0K)) ? 0. : 1.;
        // RHS_Atm.cpp:110-124: hat function of the height above ground, 1 in air
        double h_0_i = 1.;
        if (land) {
            double topo_diff = hight - a.topo[q];
            if (!(topo_diff == K.topo_step)) h_0_i = div_by(topo_diff, K.topo_step, K.i_topo_step);
        }
        const bool window = (j >= 2) && (j < jm - 3);
        const int mode = !window ? 0 : ((fl & FL_ONESIDED_S) ? 2 : 1);

        // stage-independent inputs
        double xn[8];
#pragma unroll
        for (int n = 0; n < 8; n++) xn[n] = XN[n][p];
        const double S_v = a.S_v[p], S_c = a.S_c[p], S_i = a.S_i[p], S_r = a.S_r[p], S_s = a.S_s[p];
        const double src_t1 = K.coeff_energy * (S_c + S_r), src_t2 = K.coeff_energy * (S_i + S_s);
        const double src_c = S_v * K.coeff_trans, src_cloud = S_c * K.coeff_trans, src_ice = S_i * K.coeff_trans;

        double dpdthe, d2p_unused;
        dtheta(mode, h_d_j, pm2, pm1, p0, pp1, K, dpdthe, d2p_unused);
        const double pg_v = div_by(div_by(h_d_j * dpdthe, rm, i_rm), K.r_air, K.i_r_air);  // h_d_j*dpdthe/rm/r_air

        // RHS_Atm.cpp:384-390: pressure at height from the sea-level temperature t[0][j][k]
        const double t0jk = a.t[q];  // level 0 is never written by the sweep
        const double t0K = t0jk * K.t_0;
        const double p_SL = .01 * (K.p_SL_factor * t0jk * K.t_0);
        const double p_h = pow((t0K - K.gam * hight * 1.e-2) / t0K, K.exp_pressure) * p_SL;
        const double p_h100 = 100. * p_h;

        // RHS_Atm.cpp:412-421 (level 1 only): evaporation minus precipitation, clamped
        double evap_precip = 0., c_lvl0 = 0., c_lvl3 = 0.;
        if (i == 1) {
            evap_precip = a.evap[q] - a.precip[q];
            if (evap_precip >= 6.) evap_precip = 6.;
            if (evap_precip <= -6.) evap_precip = -6.;
            c_lvl0 = a.c[q];
            c_lvl3 = a.c3[q];
        }

        // prefetch row j+2 for the next step (old values: this thread has not written them yet)
        double xp2[8], pp2 = 0.;
        if (j + 2 < jm) {
#pragma unroll
            for (int n = 0; n < 8; n++) xp2[n] = X[n][p + 2 * row];
            pp2 = a.p_dyn[p + 2 * row];
        } else {
#pragma unroll
            for (int n = 0; n < 8; n++) xp2[n] = 0.;
        }

        double st[8], acc[8], rhs[8];
        double aux_v_add = 0., RS = 0.;
#pragma unroll
        for (int n = 0; n < 8; n++) st[n] = x0[n];

#pragma unroll
        for (int s = 0; s < 4; s++) {
            // order of st/rhs: t u v w c cloud ice co2
            const double v = st[2];
            double d1[8], d2[8];
#pragma unroll
            for (int n = 0; n < 8; n++) dtheta(mode, h_d_j, xm2[n], xm1[n], st[n], xp1[n], K, d1[n], d2[n]);

            // buoyancy, RHS_Atm.cpp:390-404 (current-stage t and c)
            const double t_u = st[0] * K.t_0;
            const double r_dry = p_h100 / (K.R_Air * t_u);
            const double r_humid = r_dry * (1. + st[4]) / (1. + K.R_WV_over_R_Air * st[4]);
            RS = K.Buoyancy * (r_humid - r_dry) / r_dry * K.g;

            double vapour_evaporation = 0.;
            if (i == 1) {
                double vapour_surface = div_by(r_humid * (-3. * c_lvl0 + 4. * st[4] - c_lvl3), K.two_dr, K.i_two_dr)
                                        * (1. - 2. * c_lvl0) * evap_precip;
                vapour_evaporation = K.coeff_vapour * vapour_surface;
                if (land) vapour_evaporation = 0.;
            }

            // advection by v and the theta part of the diffusion bracket; every other
            // term of RHS_Atm.cpp:433-483 carries an exact zero factor
#define ADV(n) div_by(v * d1[n], rm, i_rm)
#define BRK_A(n) (div_by(d2[n], rm2, i_rm2) + div_by(d1[n] * costhe, rm2sinthe, i_rm2sinthe)) /* d2/rm2 + d1*cos/rm2sin */
#define BRK_B(n) (div_by(d2[n], rm2, i_rm2) + div_by(d1[n], rm2sinthe, i_rm2sinthe) * costhe) /* d2/rm2 + d1/rm2sin*cos */
            rhs[0] = -ADV(0) + div_by(BRK_A(0), K.re_pr, K.i_re_pr) + src_t1 + src_t2;
            rhs[1] = -ADV(1) + div_by(BRK_A(1), K.re, K.i_re) - RS - div_by(h_0_i * st[1], K.dr2, K.i_dr2);
            rhs[2] = -ADV(2) + -pg_v
                     + div_by(BRK_B(2) - coef_v * h_d_j * v + div_by(2. * d1[1], rm2, i_rm2), K.re, K.i_re)
                     - div_by(h_0_j * v, K.dthe2, K.i_dthe2);
            rhs[3] = -ADV(3) + div_by(BRK_B(3), K.re, K.i_re) - div_by(h_0_k * st[3], K.dphi2, K.i_dphi2);
            rhs[4] = -ADV(4) + div_by(BRK_A(4), K.re_sc_wv, K.i_re_sc_wv) + src_c;
            if (i == 1) rhs[4] = rhs[4] + vapour_evaporation;
            rhs[5] = -ADV(5) + div_by(BRK_A(5), K.re_sc_wv, K.i_re_sc_wv) + src_cloud - div_by(h_0_i * st[5], K.dr2, K.i_dr2);
            rhs[6] = -ADV(6) + div_by(BRK_A(6), K.re_sc_wv, K.i_re_sc_wv) + src_ice - div_by(h_0_i * st[6], K.dr2, K.i_dr2);
            rhs[7] = -ADV(7) + div_by(BRK_A(7), K.re_sc_co2, K.i_re_sc_co2) - div_by(h_0_i * st[7], K.dr2, K.i_dr2);
#undef ADV
#undef BRK_A
#undef BRK_B
            aux_v_add = pg_v;

            // RungeKutta_Atm.cpp:58-152
            if (s == 0) {
#pragma unroll
                for (int n = 0; n < 8; n++) { acc[n] = rhs[n]; st[n] = xn[n] + rhs[n] * .5 * K.dt; }
            } else if (s == 1) {
#pragma unroll
                for (int n = 0; n < 8; n++) { acc[n] = acc[n] + 2. * rhs[n]; st[n] = xn[n] + rhs[n] * .5 * K.dt; }
            } else if (s == 2) {
#pragma unroll
                for (int n = 0; n < 8; n++) { acc[n] = acc[n] + 2. * rhs[n]; st[n] = xn[n] + rhs[n] * K.dt; }
            } else {
#pragma unroll
                for (int n = 0; n < 8; n++) st[n] = xn[n] + div_by(K.dt * (acc[n] + rhs[n]), 6., K.i_six);
            }
        }

        // side outputs hold the values of the 4th RHS evaluation (RHS_Atm.cpp:406-410, 485-492)
        double au = rhs[1], av = rhs[2] + aux_v_add, aw = rhs[3];
        double bf = -RS * K.coeff_buoy * 1000.;
        if (land) { au = 0.; av = 0.; aw = 0.; bf = 0.; }
        a.aux_u[p] = au; a.aux_v[p] = av; a.aux_w[p] = aw; a.buoy[p] = bf;
#pragma unroll
        for (int n = 0; n < 8; n++) {
            X[n][p] = st[n];
            xm2[n] = xm1[n]; xm1[n] = st[n]; x0[n] = xp1[n]; xp1[n] = xp2[n];
        }
        pm2 = pm1; pm1 = p0; p0 = pp1; pp1 = pp2;
    }
}

}  // namespace

namespace {
int rk3d(atomdev* h, bool limit, bool limit_before) {
    ATOMDEV_TRY(hipSetDevice(h->device));
    if (int rc = atomdev_refresh_flags(h)) return rc;
    if (h->rk_variant == 2 || h->rk_variant == 0 || h->rk_variant == 3) {
        if (int rc = h->rk_variant != 0 ? atomdev_atm_rk3d_ws_fast(h, limit, limit_before)
                                        : atomdev_atm_rk3d_ws(h, limit, limit_before)) return rc;
        // (after a limiter that ran just before the sweep nothing is left for the second one outside the sweep's own
        // points: the faces and p_dyn are as the first one left them, and limiting twice changes nothing)
        return (limit && !limit_before) ? atomdev_atm_limit_rest(h) : 0;
    }
    if (limit_before) {
        if (int rc = atomdev_atm_value_limitation(h)) return rc;
        if (int rc = atomdev_atm_solid_ground(h)) return rc;
    }
    // c at level 3 as the level-1 points must see it: before the sweep touches it
    ATOMDEV_TRY(hipMemcpyAsync(h->plane, h->f[ATOMDEV_C] + 3 * h->n2, h->n2 * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    AtmRkArgs a;
    a.t = h->f[ATOMDEV_T]; a.u = h->f[ATOMDEV_U]; a.v = h->f[ATOMDEV_V]; a.w = h->f[ATOMDEV_W];
    a.c = h->f[ATOMDEV_C]; a.cloud = h->f[ATOMDEV_CLOUD]; a.ice = h->f[ATOMDEV_ICE]; a.co2 = h->f[ATOMDEV_CO2];
    a.tn = h->f[ATOMDEV_TN]; a.un = h->f[ATOMDEV_UN]; a.vn = h->f[ATOMDEV_VN]; a.wn = h->f[ATOMDEV_WN];
    a.cn = h->f[ATOMDEV_CN]; a.cloudn = h->f[ATOMDEV_CLOUDN]; a.icen = h->f[ATOMDEV_ICEN]; a.co2n = h->f[ATOMDEV_CO2N];
    a.p_dyn = h->f[ATOMDEV_P_DYN];
    a.S_v = h->f[ATOMDEV_S_V]; a.S_c = h->f[ATOMDEV_S_C]; a.S_i = h->f[ATOMDEV_S_I]; a.S_r = h->f[ATOMDEV_S_R]; a.S_s = h->f[ATOMDEV_S_S];
    a.aux_u = h->f[ATOMDEV_AUX_U]; a.aux_v = h->f[ATOMDEV_AUX_V]; a.aux_w = h->f[ATOMDEV_AUX_W]; a.buoy = h->f[ATOMDEV_BUOYANCY];
    a.flags = h->flags;
    a.topo = h->g[ATOMDEV_TOPOGRAPHY]; a.evap = h->g[ATOMDEV_EVAPORATION_DALTON]; a.precip = h->g[ATOMDEV_PRECIPITATION];
    a.c3 = h->plane;
    a.rad = h->rad; a.sinthe = h->sinthe; a.costhe = h->costhe;
    dim3 grid((h->km - 2 + 63) / 64, h->im - 2);
    if (int rc = atomdev_prof_begin(h)) return rc;
    hipLaunchKernelGGL(k_atm_rk3d, grid, dim3(64), 0, h->stream, a, h->k);
    ATOMDEV_TRY(hipGetLastError());
    if (int rc = atomdev_prof_end(h)) return rc;
    return limit ? atomdev_atm_value_limitation(h) : 0;
}
}  // namespace

extern "C" int atomdev_atm_rk3d(atomdev_t* h) {
    ATOMDEV_REQUIRE(h && h->model == ATOMDEV_ATM, "atomdev_atm_rk3d: not an atmosphere handle");
    return rk3d(h, false, false);
}

extern "C" int atomdev_atm_rk3d_limited(atomdev_t* h) {
    ATOMDEV_REQUIRE(h && h->model == ATOMDEV_ATM, "atomdev_atm_rk3d_limited: not an atmosphere handle");
    return rk3d(h, true, false);
}

extern "C" int atomdev_atm_limit_ground_rk3d_limit(atomdev_t* h) {
    ATOMDEV_REQUIRE(h && h->model == ATOMDEV_ATM, "atomdev_atm_limit_ground_rk3d_limit: not an atmosphere handle");
    return rk3d(h, true, true);
}
