// Atmosphere 3-D Runge-Kutta sweep, wave-specialised form (the default).
//
// Same arithmetic, statement for statement, as the line kernel in k_rk_atm.hip
// (reference: atmosphere/RungeKutta_Atm.cpp:32-156, RHS_Atm.cpp:68-493); only the
// mapping to the machine differs.
//
// Why.  The sweep is 14 001 independent theta-recurrences on the 1-degree grid
// (39 levels x 359 longitudes): as one thread per line that is 234 waves for the
// 1 024 SIMDs of the chip, and each of them issues ~1 800 fp64 instructions per
// theta step back to back (measured: 54 % of wave time in VALU issue at 4.1
// cycles per instruction, 32 % waiting on loads).  The chip is three quarters idle
// and the busy quarter is issue-bound.  The recurrence itself cannot be cut along
// theta, but the eight prognostic variables of one point are coupled only
// through a few stage values:
//     every variable  <- v      (advection velocity)
//     v               <- u      (2 du/dtheta / r^2 in the viscous bracket)
//     u               <- t, c   (buoyancy: dry and humid air density)
// So a workgroup of 8 waves takes 64 neighbouring lines (one level i, 64 longitudes k),
// ONE VARIABLE PER WAVE, and the waves exchange the stage values v, u, c and the
// dry density through LDS with one barrier per RK stage.  That puts 1 752 waves on
// the chip (two per SIMD on 219 CUs) and shortens the per-step critical path from
// eight variables to the heaviest one.
//
// The pressure-at-height term (one pow per point, static during the sweep) and the
// land height factor h_0_i are computed beforehand for all points in parallel
// (k_atm_prep), so they are off the sequential path.
//
// Pairing.  Waves w and w+4 of a workgroup share a SIMD; roles are numbered so that a
// heavy variable shares with a light one: (u, co2) (v, ice) (t, cloud) (w, c).
#include "atomdev_internal.h"

namespace {

enum Role { R_U = 0, R_V = 1, R_T = 2, R_W = 3, R_CO2 = 4, R_ICE = 5, R_CLOUD = 6, R_C = 7 };

struct AtmWsArgs {
    double *t, *u, *v, *w, *c, *cloud, *ice, *co2;
    const double *tn, *un, *vn, *wn, *cn, *cloudn, *icen, *co2n;
    const double* p_dyn;
    const double *S_v, *S_c, *S_i, *S_r, *S_s;
    double *aux_u, *aux_v, *aux_w, *buoy;
    const uint8_t* flags;
    const double *evap, *precip, *c3;
    const double *ph100, *h0i;                 // k_atm_prep outputs (3-D)
    const double *rad, *costhe;
    const double *tab_rm2sin, *tab_irm2sin, *tab_coefv;   // [i*jm + j]
};

struct Shared {
    double v[2][64], u[2][64], c[2][64], rdry[2][64];  // stage values, double-buffered by stage parity
    double ufin[64];                                    // updated u of the row just finished
};

// ---- prep: everything per point that does not change during the sweep -------------------------
// ph100 = 100 * p_h, p_h from RHS_Atm.cpp:384-389; h0i = h_0_i from RHS_Atm.cpp:110-124.
__global__ void k_atm_prep(const double* __restrict__ t, const double* __restrict__ topo, const uint8_t* __restrict__ flags,
                           double* __restrict__ ph100, double* __restrict__ h0i, DevConst K) {
    const int jm = K.jm, km = K.km;
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y, i = 1 + blockIdx.z;
    if (k >= km) return;
    const size_t q = (size_t)j * km + k;
    const size_t p = (size_t)i * jm * km + q;
    const double hight = (double)i * K.topo_step;
    const double t0jk = t[q];
    const double t0K = t0jk * K.t_0;
    const double p_SL = .01 * (K.p_SL_factor * t0jk * K.t_0);
    const double p_h = pow((t0K - K.gam * hight * 1.e-2) / t0K, K.exp_pressure) * p_SL;
    ph100[p] = 100. * p_h;
    double h_0_i = 1.;
    if (flags[p] & FL_LAND) {
        double topo_diff = hight - topo[q];
        if (!(topo_diff == K.topo_step)) h_0_i = div_by(topo_diff, K.topo_step, K.i_topo_step);
    }
    h0i[p] = h_0_i;
}

// theta derivative pair (RHS_Atm.cpp:162-170, 192-199, 245-312); see k_rk_atm.hip
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

// Per-step geometry, identical for all lanes of a workgroup.
struct StepGeo {
    double costhe, rm, i_rm, rm2, i_rm2, rm2sin, i_rm2sin, coef_v;
};

// d2/rm2 + d1*cos/rm2sin   (t, u, c, cloud, ice, co2)
__device__ __forceinline__ double brk_a(bool all_bw, double d1, double d2, const StepGeo& g) {
    const double x = div_by(d1 * g.costhe, g.rm2sin, g.i_rm2sin);
    return all_bw ? x : div_by(d2, g.rm2, g.i_rm2) + x;
}
// d2/rm2 + d1/rm2sin*cos   (v, w)
__device__ __forceinline__ double brk_b(bool all_bw, double d1, double d2, const StepGeo& g) {
    const double x = div_by(d1, g.rm2sin, g.i_rm2sin) * g.costhe;
    return all_bw ? x : div_by(d2, g.rm2, g.i_rm2) + x;
}

// RungeKutta_Atm.cpp:58-152 for one variable: returns the next stage value (or the final one)
__device__ __forceinline__ double rk_advance(int s, double xn, double rhs, double& acc, const DevConst& K) {
    if (s == 0) { acc = rhs; return xn + rhs * .5 * K.dt; }
    if (s == 1) { acc = acc + 2. * rhs; return xn + rhs * .5 * K.dt; }
    if (s == 2) { acc = acc + 2. * rhs; return xn + rhs * K.dt; }
    return xn + div_by(K.dt * (acc + rhs), 6., K.i_six);
}

// The whole sweep of one wave.  ROLE is a template parameter so that every wave runs
// straight-line code for its own variable; all roles execute the same number of barriers
// (one before the first step, four per step), whatever their lanes do.
template <int ROLE>
__device__ __forceinline__ void sweep_role(const AtmWsArgs& a, const DevConst& K, Shared& sh, double* __restrict__ X,
                                           const double* __restrict__ XN) {
    constexpr bool isV = ROLE == R_V, isU = ROLE == R_U, isT = ROLE == R_T, isC = ROLE == R_C;
    constexpr bool hasH0I = ROLE == R_U || ROLE == R_CO2 || ROLE == R_ICE || ROLE == R_CLOUD;
    const int jm = K.jm, km = K.km;
    const int lane = threadIdx.x & 63;
    const int kk = 1 + blockIdx.x * 64 + lane;
    const bool active = kk <= km - 2;
    const int k = active ? kk : km - 2;  // idle lanes shadow the last line: loads stay in bounds, stores are masked
    const int i = 1 + blockIdx.y;
    const bool lvl1 = isC && (i == 1);

    const size_t row = (size_t)km;
    size_t p = idx3(jm, km, i, 0, k);
    StepGeo g;
    g.rm = a.rad[i]; g.rm2 = g.rm * g.rm; g.i_rm = 1. / g.rm; g.i_rm2 = 1. / g.rm2;

    double xm2 = 0., xm1 = X[p], x0 = X[p + row], xp1 = X[p + 2 * row];
    double ym2 = 0., ym1 = 0., y0 = 0., yp1 = 0.;  // p_dyn rows (v wave)
    double um2 = 0., um1 = 0., up1 = 0.;            // u rows (v wave): updated j-2, j-1; old j+1
    if constexpr (isV) {
        ym1 = a.p_dyn[p]; y0 = a.p_dyn[p + row]; yp1 = a.p_dyn[p + 2 * row];
        um1 = a.u[p]; up1 = a.u[p + 2 * row];
    }

    // inputs of step j are loaded during step j-1
    double n_xn, n_a = 0., n_b = 0., n_c = 0., n_c0 = 0., n_c3 = 0.;
    double n_cos, n_rm2sin, n_irm2sin, n_coefv = 0.;
    uint8_t n_fl;
    auto load_inputs = [&](size_t pp, size_t qq, int jj) {
        n_xn = XN[pp];
        n_fl = a.flags[pp];
        const size_t gi = (size_t)i * jm + jj;
        n_cos = a.costhe[jj]; n_rm2sin = a.tab_rm2sin[gi]; n_irm2sin = a.tab_irm2sin[gi];
        if constexpr (isV) n_coefv = a.tab_coefv[gi];
        if constexpr (isT) {
            n_a = a.ph100[pp];
            n_b = K.coeff_energy * (a.S_c[pp] + a.S_r[pp]);
            n_c = K.coeff_energy * (a.S_i[pp] + a.S_s[pp]);
        }
        if constexpr (hasH0I) n_a = a.h0i[pp];
        if constexpr (ROLE == R_ICE) n_b = a.S_i[pp] * K.coeff_trans;
        if constexpr (ROLE == R_CLOUD) n_b = a.S_c[pp] * K.coeff_trans;
        if constexpr (isC) {
            n_b = a.S_v[pp] * K.coeff_trans;
            if (lvl1) {  // RHS_Atm.cpp:412-421
                double ep = a.evap[qq] - a.precip[qq];
                if (ep >= 6.) ep = 6.;
                if (ep <= -6.) ep = -6.;
                n_a = ep;
                n_c0 = a.c[qq];
                n_c3 = a.c3[qq];
            }
        }
    };
    load_inputs(p + row, (size_t)km + k, 1);

    // stage-1 values of row 1
    if constexpr (isV) sh.v[0][lane] = x0;
    if constexpr (isU) sh.u[0][lane] = x0;
    if constexpr (isC) sh.c[0][lane] = x0;
    if constexpr (isT) sh.rdry[0][lane] = n_a / (K.R_Air * (x0 * K.t_0));  // r_dry, RHS_Atm.cpp:385, 390
    lds_barrier();

    for (int j = 1; j < jm - 1; j++) {
        p += row;
        // ---- take over the prefetched inputs of this step, start loading the next step's ----
        const double xn = n_xn;
        const uint8_t fl = n_fl;
        const double e_a = n_a, e_b = n_b, e_c = n_c, c_lvl0 = n_c0, c_lvl3 = n_c3;
        g.costhe = n_cos; g.rm2sin = n_rm2sin; g.i_rm2sin = n_irm2sin; g.coef_v = n_coefv;
        const int jn = (j + 1 < jm - 1) ? j + 1 : j;  // the last step re-reads its own row (unused)
        load_inputs(p + (size_t)(jn - j) * row, (size_t)jn * km + k, jn);
        double xp2 = 0., yp2 = 0., up2 = 0.;
        if (j + 2 < jm) {
            xp2 = X[p + 2 * row];
            if constexpr (isV) { yp2 = a.p_dyn[p + 2 * row]; up2 = a.u[p + 2 * row]; }
        }
        const bool land = fl & FL_LAND;
        const double h_d_j = ((fl & FL_H0J) && (fl & FL_H0K)) ? 0. : 1.;  // RHS_Atm.cpp:129-149
        const bool window = (j >= 2) && (j < jm - 3);
        const int mode = !window ? 0 : ((fl & FL_ONESIDED_S) ? 2 : 1);
        const bool all_bw = __all(mode == 1);  // whole wave on the two-point backward form: d2 == 0

        double pg_v = 0., h_0_j = 0., h_0_k = 0.;
        if constexpr (isV) {
            h_0_j = (fl & FL_H0J) ? 1. : 0.;
            if (j > 1) um1 = sh.ufin[lane];  // updated u of row j-1 (row 0 is a boundary row)
            double dpdthe, d2p;
            dtheta(mode, h_d_j, ym2, ym1, y0, yp1, K, dpdthe, d2p);
            pg_v = div_by(div_by(h_d_j * dpdthe, g.rm, g.i_rm), K.r_air, K.i_r_air);  // h_d_j*dpdthe/rm/r_air
        }
        if constexpr (ROLE == R_W) h_0_k = (fl & FL_H0K) ? 1. : 0.;

        double st = x0, acc = 0., rhs = 0., RS = 0.;
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const int rb = s & 1, wb = rb ^ 1;
            double v;
            if constexpr (isV) v = st; else v = sh.v[rb][lane];
            double d1, d2 = 0., brk;
            if (all_bw) {
                d1 = div_by(h_d_j * (st - xm1), K.dthe, K.i_dthe);
                if constexpr (isV || ROLE == R_W) brk = div_by(d1, g.rm2sin, g.i_rm2sin) * g.costhe;
                else brk = div_by(d1 * g.costhe, g.rm2sin, g.i_rm2sin);
            } else {
                dtheta(mode, h_d_j, xm2, xm1, st, xp1, K, d1, d2);
                if constexpr (isV || ROLE == R_W) brk = div_by(d2, g.rm2, g.i_rm2) + div_by(d1, g.rm2sin, g.i_rm2sin) * g.costhe;
                else brk = div_by(d2, g.rm2, g.i_rm2) + div_by(d1 * g.costhe, g.rm2sin, g.i_rm2sin);
            }
            const double adv = div_by(v * d1, g.rm, g.i_rm);

            if constexpr (isT) {
                rhs = -adv + div_by(brk, K.re_pr, K.i_re_pr) + e_b + e_c;
            } else if constexpr (isU) {
                // buoyancy, RHS_Atm.cpp:390-404, from the current-stage dry density and vapour
                const double r_dry = sh.rdry[rb][lane], cs = sh.c[rb][lane];
                const double r_humid = r_dry * (1. + cs) / (1. + K.R_WV_over_R_Air * cs);
                RS = K.Buoyancy * (r_humid - r_dry) / r_dry * K.g;
                rhs = -adv + div_by(brk, K.re, K.i_re) - RS - div_by(e_a * st, K.dr2, K.i_dr2);
            } else if constexpr (isV) {
                double d1u, d2u;
                dtheta(mode, h_d_j, um2, um1, sh.u[rb][lane], up1, K, d1u, d2u);
                rhs = -adv + -pg_v
                      + div_by(brk - g.coef_v * h_d_j * v + div_by(2. * d1u, g.rm2, g.i_rm2), K.re, K.i_re)
                      - div_by(h_0_j * v, K.dthe2, K.i_dthe2);
            } else if constexpr (ROLE == R_W) {
                rhs = -adv + div_by(brk, K.re, K.i_re) - div_by(h_0_k * st, K.dphi2, K.i_dphi2);
            } else if constexpr (ROLE == R_CO2) {
                rhs = -adv + div_by(brk, K.re_sc_co2, K.i_re_sc_co2) - div_by(e_a * st, K.dr2, K.i_dr2);
            } else if constexpr (ROLE == R_ICE || ROLE == R_CLOUD) {
                rhs = -adv + div_by(brk, K.re_sc_wv, K.i_re_sc_wv) + e_b - div_by(e_a * st, K.dr2, K.i_dr2);
            } else {  // R_C
                rhs = -adv + div_by(brk, K.re_sc_wv, K.i_re_sc_wv) + e_b;
                if (lvl1) {  // RHS_Atm.cpp:412-430: surface evaporation source, c at levels 0, 1 (own), 3
                    const double r_dry = sh.rdry[rb][lane];
                    const double r_humid = r_dry * (1. + st) / (1. + K.R_WV_over_R_Air * st);
                    double vapour_surface = div_by(r_humid * (-3. * c_lvl0 + 4. * st - c_lvl3), K.two_dr, K.i_two_dr)
                                            * (1. - 2. * c_lvl0) * e_a;
                    double vapour_evaporation = K.coeff_vapour * vapour_surface;
                    if (land) vapour_evaporation = 0.;
                    rhs = rhs + vapour_evaporation;
                }
            }
            st = rk_advance(s, xn, rhs, acc, K);

            // publish: the next stage's value, or (after the last stage) stage 1 of the next row
            const double pub = (s < 3) ? st : xp1;
            if constexpr (isV) sh.v[wb][lane] = pub;
            if constexpr (isU) { sh.u[wb][lane] = pub; if (s == 3) sh.ufin[lane] = st; }
            if constexpr (isC) sh.c[wb][lane] = pub;
            if constexpr (isT) sh.rdry[wb][lane] = ((s < 3) ? e_a : n_a) / (K.R_Air * (pub * K.t_0));
            lds_barrier();
        }

        // ---- results of this row (side outputs hold the 4th evaluation: RHS_Atm.cpp:406-410, 485-492) ----
        if (active) {
            X[p] = st;
            if constexpr (isU) {
                a.aux_u[p] = land ? 0. : rhs;
                a.buoy[p] = land ? 0. : -RS * K.coeff_buoy * 1000.;
            }
            if constexpr (isV) a.aux_v[p] = land ? 0. : rhs + pg_v;
            if constexpr (ROLE == R_W) a.aux_w[p] = land ? 0. : rhs;
        }
        xm2 = xm1; xm1 = st; x0 = xp1; xp1 = xp2;
        if constexpr (isV) { ym2 = ym1; ym1 = y0; y0 = yp1; yp1 = yp2; um2 = um1; up1 = up2; }
    }
}

__global__ __launch_bounds__(512) void k_atm_rk3d_ws(AtmWsArgs a, DevConst K) {
    __shared__ Shared sh;
    switch (threadIdx.x >> 6) {
        case R_U: sweep_role<R_U>(a, K, sh, a.u, a.un); break;
        case R_V: sweep_role<R_V>(a, K, sh, a.v, a.vn); break;
        case R_T: sweep_role<R_T>(a, K, sh, a.t, a.tn); break;
        case R_W: sweep_role<R_W>(a, K, sh, a.w, a.wn); break;
        case R_CO2: sweep_role<R_CO2>(a, K, sh, a.co2, a.co2n); break;
        case R_ICE: sweep_role<R_ICE>(a, K, sh, a.ice, a.icen); break;
        case R_CLOUD: sweep_role<R_CLOUD>(a, K, sh, a.cloud, a.cloudn); break;
        default: sweep_role<R_C>(a, K, sh, a.c, a.cn); break;
    }
}

}  // namespace

int atomdev_atm_rk3d_ws(atomdev* h) {
    AtmWsArgs a;
    a.t = h->f[ATOMDEV_T]; a.u = h->f[ATOMDEV_U]; a.v = h->f[ATOMDEV_V]; a.w = h->f[ATOMDEV_W];
    a.c = h->f[ATOMDEV_C]; a.cloud = h->f[ATOMDEV_CLOUD]; a.ice = h->f[ATOMDEV_ICE]; a.co2 = h->f[ATOMDEV_CO2];
    a.tn = h->f[ATOMDEV_TN]; a.un = h->f[ATOMDEV_UN]; a.vn = h->f[ATOMDEV_VN]; a.wn = h->f[ATOMDEV_WN];
    a.cn = h->f[ATOMDEV_CN]; a.cloudn = h->f[ATOMDEV_CLOUDN]; a.icen = h->f[ATOMDEV_ICEN]; a.co2n = h->f[ATOMDEV_CO2N];
    a.p_dyn = h->f[ATOMDEV_P_DYN];
    a.S_v = h->f[ATOMDEV_S_V]; a.S_c = h->f[ATOMDEV_S_C]; a.S_i = h->f[ATOMDEV_S_I]; a.S_r = h->f[ATOMDEV_S_R]; a.S_s = h->f[ATOMDEV_S_S];
    a.aux_u = h->f[ATOMDEV_AUX_U]; a.aux_v = h->f[ATOMDEV_AUX_V]; a.aux_w = h->f[ATOMDEV_AUX_W]; a.buoy = h->f[ATOMDEV_BUOYANCY];
    a.flags = h->flags;
    a.evap = h->g[ATOMDEV_EVAPORATION_DALTON]; a.precip = h->g[ATOMDEV_PRECIPITATION];
    a.c3 = h->plane;
    a.ph100 = h->scr[0]; a.h0i = h->scr[1];
    a.rad = h->rad; a.costhe = h->costhe;
    a.tab_rm2sin = h->tab_rm2sin; a.tab_irm2sin = h->tab_irm2sin; a.tab_coefv = h->tab_coefv;
    if (int rc = atomdev_prof_begin(h)) return rc;
    dim3 pg((h->km + 127) / 128, h->jm, h->im - 2);
    hipLaunchKernelGGL(k_atm_prep, pg, dim3(128), 0, h->stream, h->f[ATOMDEV_T], h->g[ATOMDEV_TOPOGRAPHY], h->flags,
                       h->scr[0], h->scr[1], h->k);
    ATOMDEV_TRY(hipGetLastError());
    dim3 grid((h->km - 2 + 63) / 64, h->im - 2);
    hipLaunchKernelGGL(k_atm_rk3d_ws, grid, dim3(512), 0, h->stream, a, h->k);
    ATOMDEV_TRY(hipGetLastError());
    if (int rc = atomdev_prof_end(h)) return rc;
    return 0;
}
