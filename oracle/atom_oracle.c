/*
 * atom_oracle.c — CPU restatement of ATOM's dynamical-core hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity checker for the HIP
 * library: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may build, load or call it.  Nothing under atom_amd/ links or imports it.
 *
 * It restates, on flat (i*jm+j)*km+k arrays and in plain C, the arithmetic of
 * the reference classes, statement for statement where rounding is concerned
 * (same association order, same dead terms, same in-place lexicographic sweep
 * order, same quirks).  Each function cites the reference lines it follows.
 * Pinned bit-for-bit against the compiled reference (oracle/_ref) by
 * tests/test_oracle_vs_reference.py and against the .npz files of tests/golden, which
 * oracle/make_golden.py generated from that compiled reference.
 *
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared (x86-64 baseline: no FMA, like
 * the reference's own objects).
 */
#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/atomdev.h"

typedef struct oracle_ctx {
    int model, im, jm, km;
    atomdev_params p;
    double* f[ATOMDEV_N3D]; /* bound by the caller (numpy buffers), not owned */
    double* g[ATOMDEV_N2D];
    double *rad, *the, *phi; /* owned */
} oracle_ctx;

#define IDX(c, i, j, k) (((size_t)(i) * (size_t)(c)->jm + (size_t)(j)) * (size_t)(c)->km + (size_t)(k))
#define A3(c, F, i, j, k) ((c)->f[F][IDX(c, i, j, k)])
#define A2(c, G, j, k) ((c)->g[G][(size_t)(j) * (size_t)(c)->km + (size_t)(k)])

/* lib/Array_1D.cpp:54-65 — coordinates by repeated addition */
static void coordinates(double* z, int n, double z0, double dz) {
    z[0] = z0;
    for (int l = 1; l < n; l++) z[l] = z[l - 1] + dz;
}

oracle_ctx* oracle_create(int model, int im, int jm, int km, const atomdev_params* p) {
    oracle_ctx* c = (oracle_ctx*)calloc(1, sizeof(oracle_ctx));
    c->model = model; c->im = im; c->jm = jm; c->km = km; c->p = *p;
    c->rad = (double*)malloc(sizeof(double) * (size_t)(im > 41 ? im : 41));
    c->the = (double*)malloc(sizeof(double) * (size_t)jm);
    c->phi = (double*)malloc(sizeof(double) * (size_t)km);
    coordinates(c->rad, im > 41 ? im : 41, p->r0, p->dr);
    coordinates(c->the, jm, p->the0, p->dthe);
    coordinates(c->phi, km, p->phi0, p->dphi);
    return c;
}

void oracle_destroy(oracle_ctx* c) {
    if (!c) return;
    free(c->rad); free(c->the); free(c->phi); free(c);
}

void oracle_bind3d(oracle_ctx* c, int field, double* ptr) { c->f[field] = ptr; }
void oracle_bind2d(oracle_ctx* c, int field, double* ptr) { c->g[field] = ptr; }

/* lib/Utils.h:22-32 */
static inline int is_land(const oracle_ctx* c, int i, int j, int k) {
    return fabs(A3(c, ATOMDEV_H, i, j, k) - 1) < DBL_EPSILON;
}
static inline int is_fluid(const oracle_ctx* c, int i, int j, int k) { return !is_land(c, i, j, k); }

/* ------------------------------------------------------------------------- */
/* Finite differences of RK_RHS_3D_* for a list of fields.                   */
/* ------------------------------------------------------------------------- */

enum { DIR_R = 0, DIR_THE = 1, DIR_PHI = 2 };
#define MAXF 9

typedef struct { double d1[3][MAXF], d2[3][MAXF]; } derivs;

static inline double nb(const oracle_ctx* c, int F, int i, int j, int k, int dir, int o) {
    if (dir == DIR_R) return A3(c, F, i + o, j, k);
    if (dir == DIR_THE) return A3(c, F, i, j + o, k);
    return A3(c, F, i, j, k + o);
}

/* one-sided 3-point forms towards offset s = +1 / -1 */
static inline double one_sided_d1(const oracle_ctx* c, int F, int i, int j, int k, int dir, int s,
                                  double hd, double d) {
    return hd * (-3. * nb(c, F, i, j, k, dir, 0) + 4. * nb(c, F, i, j, k, dir, s) - nb(c, F, i, j, k, dir, 2 * s)) / (2. * d);
}
static inline double one_sided_d2(const oracle_ctx* c, int F, int i, int j, int k, int dir, int s,
                                  double hd, double dd) {
    return hd * (2 * nb(c, F, i, j, k, dir, 0) - 2. * nb(c, F, i, j, k, dir, s) + nb(c, F, i, j, k, dir, 2 * s)) / dd;
}

/*
 * F[0..nf): fields; index ip = position of p_dyn in F (it has no second
 * derivative), iu / ic = positions of u and c (the radial land branch builds
 * the c derivatives from u).  hyd selects the hydrosphere variants.
 * atmosphere/RHS_Atm.cpp:151-381, hydrosphere/RHS_Hyd.cpp:148-341.
 */
static void fd_derivatives(const oracle_ctx* c, const int* F, int nf, int ip, int iu, int ic,
                           int i, int j, int k, const double hd[3], int hyd, derivs* D) {
    const int im = c->im, jm = c->jm, km = c->km;
    const double d[3] = { c->p.dr, c->p.dthe, c->p.dphi };
    const double dd[3] = { c->p.dr * c->p.dr, c->p.dthe * c->p.dthe, c->p.dphi * c->p.dphi };

    for (int dir = 0; dir < 3; dir++)
        for (int n = 0; n < nf; n++) {
            D->d1[dir][n] = hd[dir] * (nb(c, F[n], i, j, k, dir, 1) - nb(c, F[n], i, j, k, dir, -1)) / (2. * d[dir]);
            D->d2[dir][n] = hd[dir] * (nb(c, F[n], i, j, k, dir, 1) - 2. * nb(c, F[n], i, j, k, dir, 0)
                                        + nb(c, F[n], i, j, k, dir, -1)) / dd[dir];
        }

    /* radial block: RHS_Atm.cpp:210-243, RHS_Hyd.cpp:191-215 */
    if (i < im - 2) {
        int above_is_fluid = hyd ? (A3(c, ATOMDEV_H, i + 1, j, k) == 0.) : is_fluid(c, i + 1, j, k);
        if (is_land(c, i, j, k) && above_is_fluid) {
            for (int n = 0; n < nf; n++) {
                int src = (n == ic) ? F[iu] : F[n]; /* c built from u: RHS_Atm.cpp:217,226; RHS_Hyd.cpp:198,204 */
                D->d1[DIR_R][n] = one_sided_d1(c, src, i, j, k, DIR_R, +1, hd[DIR_R], d[DIR_R]);
                if (n != ip) D->d2[DIR_R][n] = one_sided_d2(c, src, i, j, k, DIR_R, +1, hd[DIR_R], dd[DIR_R]);
            }
        } else {
            for (int n = 0; n < nf; n++) {
                double fwd = (nb(c, F[n], i, j, k, DIR_R, 1) - nb(c, F[n], i, j, k, DIR_R, 0));
                /* hydrosphere drops h_d_i here: RHS_Hyd.cpp:206-211 */
                D->d1[DIR_R][n] = hyd ? fwd / d[DIR_R] : hd[DIR_R] * fwd / d[DIR_R];
                if (n != ip) D->d2[DIR_R][n] = 0.;
            }
        }
    }

    /* theta block: "north" branch, then unconditionally the "south" branch.
       RHS_Atm.cpp:245-312, RHS_Hyd.cpp:218-278 */
    if ((j >= 2) && (j < jm - 3)) {
        for (int s = +1; s >= -1; s -= 2) {
            if (is_land(c, i, j, k) && (is_fluid(c, i, j + s, k) && is_fluid(c, i, j + 2 * s, k))) {
                for (int n = 0; n < nf; n++) {
                    D->d1[DIR_THE][n] = one_sided_d1(c, F[n], i, j, k, DIR_THE, s, hd[DIR_THE], d[DIR_THE]);
                    if (n != ip) D->d2[DIR_THE][n] = one_sided_d2(c, F[n], i, j, k, DIR_THE, s, hd[DIR_THE], dd[DIR_THE]);
                }
            } else {
                for (int n = 0; n < nf; n++) {
                    double a = nb(c, F[n], i, j, k, DIR_THE, s > 0 ? 1 : 0);
                    double b = nb(c, F[n], i, j, k, DIR_THE, s > 0 ? 0 : -1);
                    D->d1[DIR_THE][n] = hd[DIR_THE] * (a - b) / d[DIR_THE];
                    if (n != ip) D->d2[DIR_THE][n] = 0.;
                }
            }
        }
    }

    /* phi block. RHS_Atm.cpp:315-381, RHS_Hyd.cpp:281-341.  The hydrosphere's
       one-sided branches assign the *theta* second derivatives (RHS_Hyd.cpp:296-300,326-330). */
    if ((k >= 2) && (k < km - 3)) {
        for (int s = +1; s >= -1; s -= 2) {
            if (is_land(c, i, j, k) && is_fluid(c, i, j, k + s) && is_fluid(c, i, j, k + 2 * s)) {
                for (int n = 0; n < nf; n++) {
                    D->d1[DIR_PHI][n] = one_sided_d1(c, F[n], i, j, k, DIR_PHI, s, hd[DIR_PHI], d[DIR_PHI]);
                    if (n != ip) D->d2[hyd ? DIR_THE : DIR_PHI][n] =
                        one_sided_d2(c, F[n], i, j, k, DIR_PHI, s, hd[DIR_PHI], dd[DIR_PHI]);
                }
            } else {
                for (int n = 0; n < nf; n++) {
                    double a = nb(c, F[n], i, j, k, DIR_PHI, s > 0 ? 1 : 0);
                    double b = nb(c, F[n], i, j, k, DIR_PHI, s > 0 ? 0 : -1);
                    D->d1[DIR_PHI][n] = hd[DIR_PHI] * (a - b) / d[DIR_PHI];
                    if (n != ip) D->d2[DIR_PHI][n] = 0.;
                }
            }
        }
    }
}

/* Immersed-boundary factors. RHS_Atm.cpp:108-149, RHS_Hyd.cpp:94-139.
 * out: h0[3] = {h_0_i, h_0_j, h_0_k}, hd[3] = {h_d_i, h_d_j, h_d_k}. */
static void ib_factors(const oracle_ctx* c, int i, int j, int k, int hyd, double L, double h0[3], double hd[3]) {
    const double cc = 1., dist_coeff = 1.;
    const double dthe = c->p.dthe, dphi = c->p.dphi;
    double topo_step = L / (double)(c->im - 1);
    double hight = (double)i * topo_step;
    double topo_diff = hyd ? fabs(hight - A2(c, ATOMDEV_TOPOGRAPHY, j, k)) : hight - A2(c, ATOMDEV_TOPOGRAPHY, j, k);
    double h_0_i = topo_diff / topo_step;
    double h_d_i = 0, h_0_0 = 0;
    int below_solid = hyd ? (A3(c, ATOMDEV_H, i - 1, j, k) == 1.) : is_land(c, i - 1, j, k);
    if ((topo_diff < topo_step) && (is_fluid(c, i, j, k) && below_solid)) {
        h_0_0 = 1. - h_0_i;
        h_d_i = cc * (1. - h_0_0);
    }
    if ((topo_diff == topo_step) || is_fluid(c, i, j, k)) {
        h_0_i = 1.;
        h_d_i = cc * (1. - h_0_i);
    }
    double dist = 0, h_0_j = 0, h_d_j = 0;
    if ((is_fluid(c, i, j, k) && is_land(c, i, j - 1, k)) || (is_fluid(c, i, j, k) && is_land(c, i, j + 1, k))) {
        dist = dist_coeff * dthe;
        h_0_j = dist / dthe;
        h_d_j = cc * (1. - h_0_j);
    } else {
        h_0_j = 0.;
        h_d_j = cc * (1. - h_0_j);
    }
    double h_0_k = 0, h_d_k = 0;
    if ((is_fluid(c, i, j, k) && is_land(c, i, j, k - 1)) || (is_fluid(c, i, j, k) && is_land(c, i, j, k + 1))) {
        dist = dist_coeff * dphi;
        h_0_k = dist / dphi;
        h_d_k = cc * (1. - h_0_k);
    } else {
        h_0_k = 0.;
        h_d_j = cc * (1. - h_0_k); /* assigns h_d_j, not h_d_k: RHS_Atm.cpp:148, RHS_Hyd.cpp:138 */
    }
    h0[0] = h_0_i; h0[1] = h_0_j; h0[2] = h_0_k;
    hd[0] = h_d_i; hd[1] = h_d_j; hd[2] = h_d_k;
}

/* ------------------------------------------------------------------------- */
/* Atmosphere                                                                */
/* ------------------------------------------------------------------------- */

enum { AU = 0, AV, AW, AT, AP, AC, ACL, AIC, ACO };
static const int ATM_F[9] = { ATOMDEV_U, ATOMDEV_V, ATOMDEV_W, ATOMDEV_T, ATOMDEV_P_DYN,
                              ATOMDEV_C, ATOMDEV_CLOUD, ATOMDEV_ICE, ATOMDEV_CO2 };

/* RHS_Atmosphere::RK_RHS_3D_Atmosphere, atmosphere/RHS_Atm.cpp:68-493.
 * rhs order: t u v w c cloud ice co2. */
static void atm_rhs3d(oracle_ctx* c, int i, int j, int k, double rhs[8]) {
    const atomdev_params* P = &c->p;
    const double k_Force = 1.;
    const double dr = P->dr, dthe = P->dthe, dphi = P->dphi, re = P->re;
    double dr2 = dr * dr, dthe2 = dthe * dthe, dphi2 = dphi * dphi;
    double rm = c->rad[i], rm2 = rm * rm;
    double sinthe = sin(c->the[j]), sinthe2 = sinthe * sinthe, costhe = cos(c->the[j]);
    double rmsinthe = rm * sinthe, rm2sinthe = rm2 * sinthe, rm2sinthe2 = rm2 * sinthe2;

    double h0[3], hd[3];
    ib_factors(c, i, j, k, 0, P->L_atm, h0, hd);
    const double h_0_i = h0[0], h_0_j = h0[1], h_0_k = h0[2];
    const double h_d_i = hd[0], h_d_j = hd[1], h_d_k = hd[2];

    derivs D;
    fd_derivatives(c, ATM_F, 9, AP, AU, AC, i, j, k, hd, 0, &D);
#define DR(n) D.d1[DIR_R][n]
#define DT(n) D.d1[DIR_THE][n]
#define DP(n) D.d1[DIR_PHI][n]
#define D2R(n) D.d2[DIR_R][n]
#define D2T(n) D.d2[DIR_THE][n]
#define D2P(n) D.d2[DIR_PHI][n]

    const double u = A3(c, ATOMDEV_U, i, j, k), v = A3(c, ATOMDEV_V, i, j, k), w = A3(c, ATOMDEV_W, i, j, k);
    const double t0jk = A3(c, ATOMDEV_T, 0, j, k);

    /* RHS_Atm.cpp:384-410 */
    double exp_pressure = P->g / (1.e-2 * P->gam * P->R_Air);
    double t_u = A3(c, ATOMDEV_T, i, j, k) * P->t_0;
    double p_SL = .01 * (P->r_air * P->R_Air * t0jk * P->t_0);
    double hight = (double)i * (P->L_atm / (double)(c->im - 1));
    double p_h = pow(((t0jk * P->t_0 - P->gam * hight * 1.e-2) / (t0jk * P->t_0)), exp_pressure) * p_SL;
    double r_dry = 100. * p_h / (P->R_Air * t_u);

    double coeff_energy = P->L_atm / (P->cp_l * P->t_0 * P->u_0);
    double coeff_buoy = P->r_air * P->u_0 * P->u_0 / P->L_atm;
    double coeff_trans = P->L_atm / P->u_0;
    double vapour_evaporation = 0., vapour_surface = 0., evap_precip = 0.;
    double coeff_vapour = 1.1574e-5 * P->L_atm / P->u_0;

    double cc_ = A3(c, ATOMDEV_C, i, j, k);
    double r_humid = r_dry * (1. + cc_) / (1. + P->R_WaterVapour / P->R_Air * cc_);
    double RS_buoyancy_Momentum = P->Buoyancy * (r_humid - r_dry) / r_dry * P->g;

    A3(c, ATOMDEV_BUOYANCY, i, j, k) = -RS_buoyancy_Momentum * coeff_buoy * 1000.;
    if (is_land(c, i, j, k)) A3(c, ATOMDEV_BUOYANCY, i, j, k) = 0.;

    /* RHS_Atm.cpp:412-430 (note c at level 3, not 2) */
    if (i == 1) {
        evap_precip = A2(c, ATOMDEV_EVAPORATION_DALTON, j, k) - A2(c, ATOMDEV_PRECIPITATION, j, k);
        if (evap_precip >= 6.) evap_precip = 6.;
        if (evap_precip <= -6.) evap_precip = -6.;
        vapour_surface = r_humid * (-3. * A3(c, ATOMDEV_C, 0, j, k) + 4. * A3(c, ATOMDEV_C, 1, j, k)
                                    - A3(c, ATOMDEV_C, 3, j, k)) / (2. * dr)
                         * (1. - 2. * A3(c, ATOMDEV_C, 0, j, k)) * evap_precip;
        vapour_evaporation = +coeff_vapour * vapour_surface;
        if (is_land(c, i, j, k)) vapour_evaporation = 0.;
    } else {
        vapour_evaporation = 0.;
    }

    const double S_v = A3(c, ATOMDEV_S_V, i, j, k), S_c = A3(c, ATOMDEV_S_C, i, j, k),
                 S_i = A3(c, ATOMDEV_S_I, i, j, k), S_r = A3(c, ATOMDEV_S_R, i, j, k),
                 S_s = A3(c, ATOMDEV_S_S, i, j, k);

    /* RHS_Atm.cpp:432-483 */
    double rhs_t = -(u * DR(AT) + v * DT(AT) / rm + w * DP(AT) / rmsinthe)
                   + (D2R(AT) + DR(AT) * 2. / rm + D2T(AT) / rm2 + DT(AT) * costhe / rm2sinthe + D2P(AT) / rm2sinthe2) / (re * P->pr)
                   + coeff_energy * (S_c + S_r)
                   + coeff_energy * (S_i + S_s);

    double rhs_u = -(u * DR(AU) + v * DT(AU) / rm + w * DP(AU) / rmsinthe)
                   - h_d_i * DR(AP) / P->r_air
                   + (D2R(AU) + h_d_i * 2. * u / rm2 + D2T(AU) / rm2 + 4. * DR(AU) / rm + DT(AU) * costhe / rm2sinthe + D2P(AU) / rm2sinthe2) / re
                   - RS_buoyancy_Momentum
                   - h_0_i * u * k_Force / dr2;

    double rhs_v = -(u * DR(AV) + v * DT(AV) / rm + w * DP(AV) / rmsinthe)
                   + -h_d_j * DT(AP) / rm / P->r_air
                   + (D2R(AV) + DR(AV) * 2. / rm + D2T(AV) / rm2 + DT(AV) / rm2sinthe * costhe
                      - (1. + costhe * costhe / (rm * sinthe2)) * h_d_j * v + D2P(AV) / rm2sinthe2
                      + 2. * DT(AU) / rm2 - DP(AW) * 2. * costhe / rm2sinthe2) / re
                   - h_0_j * v * k_Force / dthe2;

    double rhs_w = -(u * DR(AW) + v * DT(AW) / rm + w * DP(AW) / rmsinthe)
                   + -h_d_k * DP(AP) / rmsinthe / P->r_air
                   + (D2R(AW) + DR(AW) * 2. / rm + D2T(AW) / rm2 + DT(AW) / rm2sinthe * costhe
                      - (1. + costhe * costhe / (rm * sinthe2)) * h_d_k * w
                      + D2P(AW) / rm2sinthe2 + 2. * DP(AU) / rm2sinthe + DP(AV) * 2. * costhe / rm2sinthe2) / re
                   - h_0_k * w * k_Force / dphi2;

    double rhs_c = -(u * DR(AC) + v * DT(AC) / rm + w * DP(AC) / rmsinthe)
                   + (D2R(AC) + DR(AC) * 2. / rm + D2T(AC) / rm2 + DT(AC) * costhe / rm2sinthe + D2P(AC) / rm2sinthe2) / (P->sc_WaterVapour * re)
                   + S_v * coeff_trans
                   + vapour_evaporation;

    double cloud = A3(c, ATOMDEV_CLOUD, i, j, k), ice = A3(c, ATOMDEV_ICE, i, j, k), co2 = A3(c, ATOMDEV_CO2, i, j, k);
    double rhs_cloud = -(u * DR(ACL) + v * DT(ACL) / rm + w * DP(ACL) / rmsinthe)
                       + (D2R(ACL) + DR(ACL) * 2. / rm + D2T(ACL) / rm2 + DT(ACL) * costhe / rm2sinthe + D2P(ACL) / rm2sinthe2) / (P->sc_WaterVapour * re)
                       + S_c * coeff_trans
                       - h_0_i * cloud * k_Force / dr2;

    double rhs_ice = -(u * DR(AIC) + v * DT(AIC) / rm + w * DP(AIC) / rmsinthe)
                     + (D2R(AIC) + DR(AIC) * 2. / rm + D2T(AIC) / rm2 + DT(AIC) * costhe / rm2sinthe + D2P(AIC) / rm2sinthe2) / (P->sc_WaterVapour * re)
                     + S_i * coeff_trans
                     - h_0_i * ice * k_Force / dr2;

    double rhs_co2 = -(u * DR(ACO) + v * DT(ACO) / rm + w * DP(ACO) / rmsinthe)
                     + (D2R(ACO) + DR(ACO) * 2. / rm + D2T(ACO) / rm2 + DT(ACO) * costhe / rm2sinthe + D2P(ACO) / rm2sinthe2) / (P->sc_CO2 * re)
                     - h_0_i * co2 * k_Force / dr2;

    /* RHS_Atm.cpp:485-492 */
    A3(c, ATOMDEV_AUX_U, i, j, k) = rhs_u + h_d_i * DR(AP) / P->r_air;
    A3(c, ATOMDEV_AUX_V, i, j, k) = rhs_v + h_d_j * DT(AP) / rm / P->r_air;
    A3(c, ATOMDEV_AUX_W, i, j, k) = rhs_w + h_d_k * DP(AP) / rmsinthe / P->r_air;
    if (is_land(c, i, j, k))
        A3(c, ATOMDEV_AUX_U, i, j, k) = A3(c, ATOMDEV_AUX_V, i, j, k) = A3(c, ATOMDEV_AUX_W, i, j, k) = 0.;

    rhs[0] = rhs_t; rhs[1] = rhs_u; rhs[2] = rhs_v; rhs[3] = rhs_w;
    rhs[4] = rhs_c; rhs[5] = rhs_cloud; rhs[6] = rhs_ice; rhs[7] = rhs_co2;
}

/* RungeKutta_Atmosphere::solveRungeKutta_3D_Atmosphere, atmosphere/RungeKutta_Atm.cpp:32-156.
 * In-place, lexicographic (i, j, k): a nonlinear Gauss-Seidel sweep. */
void oracle_atm_rk3d(oracle_ctx* c) {
    static const int X[8] = { ATOMDEV_T, ATOMDEV_U, ATOMDEV_V, ATOMDEV_W, ATOMDEV_C, ATOMDEV_CLOUD, ATOMDEV_ICE, ATOMDEV_CO2 };
    static const int XN[8] = { ATOMDEV_TN, ATOMDEV_UN, ATOMDEV_VN, ATOMDEV_WN, ATOMDEV_CN, ATOMDEV_CLOUDN, ATOMDEV_ICEN, ATOMDEV_CO2N };
    const double dt = c->p.dt;
    double k1[8], k2[8], k3[8], k4[8];
    for (int i = 1; i < c->im - 1; i++)
        for (int j = 1; j < c->jm - 1; j++)
            for (int k = 1; k < c->km - 1; k++) {
                atm_rhs3d(c, i, j, k, k1);
                for (int n = 0; n < 8; n++) A3(c, X[n], i, j, k) = A3(c, XN[n], i, j, k) + k1[n] * .5 * dt;
                atm_rhs3d(c, i, j, k, k2);
                for (int n = 0; n < 8; n++) A3(c, X[n], i, j, k) = A3(c, XN[n], i, j, k) + k2[n] * .5 * dt;
                atm_rhs3d(c, i, j, k, k3);
                for (int n = 0; n < 8; n++) A3(c, X[n], i, j, k) = A3(c, XN[n], i, j, k) + k3[n] * dt;
                atm_rhs3d(c, i, j, k, k4);
                for (int n = 0; n < 8; n++)
                    A3(c, X[n], i, j, k) = A3(c, XN[n], i, j, k) + dt * (k1[n] + 2. * k2[n] + 2. * k3[n] + k4[n]) / 6.;
            }
}

/*
 * 2-D surface RHS shared by both models: RHS_Atmosphere::RK_RHS_2D_Atmosphere
 * (atmosphere/RHS_Atm.cpp:498-678) and RHS_Hydrosphere::RK_RHS_2D_Hydrosphere
 * (hydrosphere/RHS_Hyd.cpp:453-653).  L = level (0 atm; the hydrosphere object
 * sets im = 41, so L = 40 there).  fluid_c: the hydrosphere tests h == 0.
 * where the atmosphere tests !is_land.  sin2_only: the atmosphere divides
 * cos^2 by sinthe2 too (no rm factor) in the 2-D form — both do.
 */
static void rhs2d(oracle_ctx* c, int L, int j, int k, int hyd, double rho, double out[2]) {
    const atomdev_params* P = &c->p;
    const double k_Force = 1., cc = 1., dist_coeff = 1.;
    const double dthe = P->dthe, dphi = P->dphi, re = P->re;
    const int jm = c->jm, km = c->km;
    double dthe2 = dthe * dthe, dphi2 = dphi * dphi;
    double rm = c->rad[L], rm2 = rm * rm;
    double sinthe = sin(c->the[j]), sinthe2 = sinthe * sinthe, costhe = cos(c->the[j]);
    double rmsinthe = rm * sinthe, rm2sinthe = rm2 * sinthe, rm2sinthe2 = rm2 * sinthe2;

#define V_(jj, kk) A3(c, ATOMDEV_V, L, jj, kk)
#define W_(jj, kk) A3(c, ATOMDEV_W, L, jj, kk)
#define P_(jj, kk) A3(c, ATOMDEV_P_DYN, L, jj, kk)
#define LAND(jj, kk) is_land(c, L, jj, kk)
#define FLUID(jj, kk) is_fluid(c, L, jj, kk)
    int centre_fluid = hyd ? (A3(c, ATOMDEV_H, L, j, k) == 0.) : FLUID(j, k);

    double dist = 0, h_0_j = 0, h_d_j = 0;
    if ((centre_fluid && LAND(j + 1, k)) || (centre_fluid && LAND(j - 1, k))) {
        dist = dist_coeff * dthe; h_0_j = dist / dthe; h_d_j = cc * (1. - h_0_j);
    } else { h_0_j = 0.; h_d_j = cc * (1. - h_0_j); }
    double h_0_k = 0, h_d_k = 0;
    if ((centre_fluid && LAND(j, k + 1)) || (centre_fluid && LAND(j, k - 1))) {
        dist = dist_coeff * dphi; h_0_k = dist / dphi; h_d_k = cc * (1. - h_0_k);
    } else { h_0_k = 0.; h_d_k = cc * (1. - h_0_k); }

    double dvdthe = h_d_j * (V_(j + 1, k) - V_(j - 1, k)) / (2. * dthe);
    double dwdthe = h_d_j * (W_(j + 1, k) - W_(j - 1, k)) / (2. * dthe);
    double dpdthe = h_d_j * (P_(j + 1, k) - P_(j - 1, k)) / (2. * dthe);
    double dvdphi = h_d_k * (V_(j, k + 1) - V_(j, k - 1)) / (2. * dphi);
    double dwdphi = h_d_k * (W_(j, k + 1) - W_(j, k - 1)) / (2. * dphi);
    double dpdphi = h_d_k * (P_(j, k + 1) - P_(j, k - 1)) / (2. * dphi);
    double d2vdthe2 = h_d_j * (V_(j + 1, k) - 2. * V_(j, k) + V_(j - 1, k)) / dthe2;
    double d2wdthe2 = h_d_j * (W_(j + 1, k) - 2. * W_(j, k) + W_(j - 1, k)) / dthe2;
    double d2vdphi2 = h_d_k * (V_(j, k + 1) - 2. * V_(j, k) + V_(j, k - 1)) / dphi2;
    double d2wdphi2 = h_d_k * (W_(j, k + 1) - 2. * W_(j, k) + W_(j, k - 1)) / dphi2;

    /* RHS_Atm.cpp:559-603, RHS_Hyd.cpp:526-570 */
    if ((j >= 2) && (j < jm - 3)) {
        for (int s = +1; s >= -1; s -= 2) {
            if (LAND(j, k) && (FLUID(j + s, k) && FLUID(j + 2 * s, k))) {
                dvdthe = h_d_j * (-3. * V_(j, k) + 4. * V_(j + s, k) - V_(j + 2 * s, k)) / (2. * dthe);
                dwdthe = h_d_j * (-3. * W_(j, k) + 4. * W_(j + s, k) - W_(j + 2 * s, k)) / (2. * dthe);
                dpdthe = h_d_j * (-3. * P_(j, k) + 4. * P_(j + s, k) - P_(j + 2 * s, k)) / (2. * dthe);
                d2vdthe2 = h_d_j * (2. * V_(j, k) - 2. * V_(j + s, k) + V_(j + 2 * s, k)) / dthe2;
                d2wdthe2 = h_d_j * (2. * W_(j, k) - 2. * W_(j + s, k) + W_(j + 2 * s, k)) / dthe2;
            }
            if (LAND(j, k) && FLUID(j + s, k)) {
                int a = s > 0 ? j + 1 : j, b = s > 0 ? j : j - 1;
                dvdthe = h_d_j * (V_(a, k) - V_(b, k)) / dthe;
                dwdthe = h_d_j * (W_(a, k) - W_(b, k)) / dthe;
                dpdthe = h_d_j * (P_(a, k) - P_(b, k)) / dthe;
                d2vdthe2 = d2wdthe2 = 0.;
            }
        }
        d2vdthe2 = d2wdthe2 = 0.;
    }

    /* RHS_Atm.cpp:605-662, RHS_Hyd.cpp:573-630; the one-sided branches assign
       d2vdthe2/d2wdthe2 (sic), and everything second-order ends up 0. */
    if ((k >= 2) && (k < km - 3)) {
        for (int s = +1; s >= -1; s -= 2) {
            if (LAND(j, k) && FLUID(j, k + s) && FLUID(j, k + 2 * s)) {
                dvdphi = h_d_k * (-3. * V_(j, k) + 4. * V_(j, k + s) - V_(j, k + 2 * s)) / (2. * dphi);
                dwdphi = h_d_k * (-3. * W_(j, k) + 4. * W_(j, k + s) - W_(j, k + 2 * s)) / (2. * dphi);
                dpdphi = h_d_k * (-3. * P_(j, k) + 4. * P_(j, k + s) - P_(j, k + 2 * s)) / (2. * dphi);
                d2vdthe2 = h_d_k * (2. * V_(j, k) - 2. * V_(j, k + s) + V_(j, k + 2 * s)) / dphi2;
                d2wdthe2 = h_d_k * (2. * W_(j, k) - 2. * W_(j, k + s) + W_(j, k + 2 * s)) / dphi2;
            }
            if (LAND(j, k) && FLUID(j, k + s)) {
                int a = s > 0 ? k + 1 : k, b = s > 0 ? k : k - 1;
                dvdphi = h_d_k * (V_(j, a) - V_(j, b)) / dphi;
                dwdphi = h_d_k * (W_(j, a) - W_(j, b)) / dphi;
                dpdphi = h_d_k * (P_(j, a) - P_(j, b)) / dphi;
                d2vdphi2 = d2wdphi2 = 0.;
            }
        }
        d2vdphi2 = d2wdphi2 = 0.;
    } else {
        if (LAND(j, k) && FLUID(j, k + 1)) {
            dvdphi = h_d_k * (V_(j, k + 1) - V_(j, k)) / dphi;
            dwdphi = h_d_k * (W_(j, k + 1) - W_(j, k)) / dphi;
            dpdphi = h_d_k * (P_(j, k + 1) - P_(j, k)) / dphi;
        }
        if (centre_fluid && LAND(j, k - 1)) {
            dvdphi = h_d_k * (V_(j, k) - V_(j, k - 1)) / dphi;
            dwdphi = h_d_k * (W_(j, k) - W_(j, k - 1)) / dphi;
            dpdphi = h_d_k * (P_(j, k) - P_(j, k - 1)) / dphi;
        }
        d2vdthe2 = d2wdthe2 = 0.;
        d2vdphi2 = d2wdphi2 = 0.;
    }

    double v = V_(j, k), w = W_(j, k);
    /* RHS_Atm.cpp:664-677, RHS_Hyd.cpp:632-645 */
    double rhs_v = -(v * dvdthe / rm + w * dvdphi / rmsinthe)
                   + -h_d_j * dpdthe / rm / rho
                   - (d2vdthe2 / rm2 + dvdthe / rm2sinthe * costhe
                      - (1. + costhe * costhe / sinthe2) * h_d_j * v
                      + d2vdphi2 / rm2sinthe2 - dwdphi * 2. * costhe / rm2sinthe2) / re
                   - h_0_j * v * k_Force / dthe2;
    double rhs_w = -(v * dwdthe / rm + w * dwdphi / rmsinthe)
                   + -h_d_k * dpdphi / rmsinthe / rho
                   + (d2wdthe2 / rm2 + dwdthe / rm2sinthe * costhe
                      - (1. + costhe * costhe / sinthe2) * h_d_k * w
                      + d2wdphi2 / rm2sinthe2 + dvdphi * 2. * costhe / rm2sinthe2) / re
                   - h_0_k * w * k_Force / dphi2;
    A3(c, ATOMDEV_AUX_V, L, j, k) = rhs_v + h_d_j * dpdthe / rm / rho;
    A3(c, ATOMDEV_AUX_W, L, j, k) = rhs_w + h_d_k * dpdphi / rmsinthe / rho;
    out[0] = rhs_v; out[1] = rhs_w;
#undef V_
#undef W_
#undef P_
#undef LAND
#undef FLUID
}

/* 4-stage in-place sweep over a surface: atmosphere/RungeKutta_Atm.cpp:163-217,
 * hydrosphere/RungeKutta_Hyd.cpp:130-186.  Lrhs = level the RHS object works
 * on, Lrk = level the RK driver reads/writes (they differ only if a hydrosphere
 * grid has im != 41). */
static void rk2d(oracle_ctx* c, int Lrhs, int Lrk, int hyd, double rho) {
    const double dt = c->p.dt;
    double k1[2], k2[2], k3[2], k4[2];
    for (int j = 1; j < c->jm - 1; j++)
        for (int k = 1; k < c->km - 1; k++) {
            rhs2d(c, Lrhs, j, k, hyd, rho, k1);
            A3(c, ATOMDEV_V, Lrk, j, k) = A3(c, ATOMDEV_VN, Lrk, j, k) + k1[0] * .5 * dt;
            A3(c, ATOMDEV_W, Lrk, j, k) = A3(c, ATOMDEV_WN, Lrk, j, k) + k1[1] * .5 * dt;
            rhs2d(c, Lrhs, j, k, hyd, rho, k2);
            A3(c, ATOMDEV_V, Lrk, j, k) = A3(c, ATOMDEV_VN, Lrk, j, k) + k2[0] * .5 * dt;
            A3(c, ATOMDEV_W, Lrk, j, k) = A3(c, ATOMDEV_WN, Lrk, j, k) + k2[1] * .5 * dt;
            rhs2d(c, Lrhs, j, k, hyd, rho, k3);
            A3(c, ATOMDEV_V, Lrk, j, k) = A3(c, ATOMDEV_VN, Lrk, j, k) + k3[0] * dt;
            A3(c, ATOMDEV_W, Lrk, j, k) = A3(c, ATOMDEV_WN, Lrk, j, k) + k3[1] * dt;
            rhs2d(c, Lrhs, j, k, hyd, rho, k4);
            A3(c, ATOMDEV_V, Lrk, j, k) = A3(c, ATOMDEV_VN, Lrk, j, k) + dt * (k1[0] + 2. * k2[0] + 2. * k3[0] + k4[0]) / 6.;
            A3(c, ATOMDEV_W, Lrk, j, k) = A3(c, ATOMDEV_WN, Lrk, j, k) + dt * (k1[1] + 2. * k2[1] + 2. * k3[1] + k4[1]) / 6.;
        }
}

void oracle_atm_rk2d(oracle_ctx* c) { rk2d(c, 0, 0, 0, c->p.r_air); }

/* ------------------------------------------------------------------------- */
/* Pressure                                                                  */
/* ------------------------------------------------------------------------- */

static const double c43 = 4. / 3., c13 = 1. / 3.;

/* Pressure_Atm::computePressure_3D (atmosphere/Pressure_Atm.cpp:39-192) and
 * Pressure_Hyd::computePressure_3D (hydrosphere/Pressure_Hyd.cpp:42-203). */
static void pressure3d(oracle_ctx* c, int hyd, double rho) {
    const int im = c->im, jm = c->jm, km = c->km;
    const double dr = c->p.dr, dthe = c->p.dthe, dphi = c->p.dphi;
    static const int AUX[3] = { ATOMDEV_AUX_U, ATOMDEV_AUX_V, ATOMDEV_AUX_W };
#define AU_(i, j, k) A3(c, ATOMDEV_AUX_U, i, j, k)
#define AV_(i, j, k) A3(c, ATOMDEV_AUX_V, i, j, k)
#define AW_(i, j, k) A3(c, ATOMDEV_AUX_W, i, j, k)
#define PD(i, j, k) A3(c, ATOMDEV_P_DYN, i, j, k)
#define PN(i, j, k) A3(c, ATOMDEV_P_DYNN, i, j, k)
    for (int j = 0; j < jm; j++)
        for (int k = 0; k < km; k++) {
            AU_(0, j, k) = 0.;
            AU_(im - 1, j, k) = 0.;
            AV_(0, j, k) = c43 * AV_(1, j, k) - c13 * AV_(2, j, k);
            AV_(im - 1, j, k) = c43 * AV_(im - 2, j, k) - c13 * AV_(im - 3, j, k);
            AW_(0, j, k) = c43 * AW_(1, j, k) - c13 * AW_(2, j, k);
            AW_(im - 1, j, k) = c43 * AW_(im - 2, j, k) - c13 * AW_(im - 3, j, k);
        }
    for (int k = 0; k < km; k++)
        for (int i = 0; i < im; i++)
            for (int n = 0; n < 3; n++) {
                A3(c, AUX[n], i, 0, k) = c43 * A3(c, AUX[n], i, 1, k) - c13 * A3(c, AUX[n], i, 2, k);
                A3(c, AUX[n], i, jm - 1, k) = c43 * A3(c, AUX[n], i, jm - 2, k) - c13 * A3(c, AUX[n], i, jm - 3, k);
            }
    for (int i = 0; i < im; i++)
        for (int j = 0; j < jm; j++)
            for (int n = 0; n < 3; n++) {
                A3(c, AUX[n], i, j, 0) = c43 * A3(c, AUX[n], i, j, 1) - c13 * A3(c, AUX[n], i, j, 2);
                A3(c, AUX[n], i, j, km - 1) = c43 * A3(c, AUX[n], i, j, km - 2) - c13 * A3(c, AUX[n], i, j, km - 3);
                A3(c, AUX[n], i, j, 0) = A3(c, AUX[n], i, j, km - 1) = (A3(c, AUX[n], i, j, 0) + A3(c, AUX[n], i, j, km - 1)) / 2.;
            }

    double dr2 = dr * dr, dthe2 = dthe * dthe, dphi2 = dphi * dphi;
    /* hydrosphere reads aux_v / aux_w at level im-1 for every i: Pressure_Hyd.cpp:137-163 */
    for (int i = 1; i < im - 1; i++) {
        double rm = c->rad[i], rm2 = rm * rm;
        const int la = hyd ? im - 1 : i;
        for (int j = 1; j < jm - 1; j++) {
            double sinthe = sin(c->the[j]);
            double rmsinthe = rm * sinthe;
            double rm2sinthe2 = rmsinthe * rmsinthe;
            double denom = 2. / dr2 + 2. / (rm2 * dthe2) + 2. / (rm2sinthe2 * dphi2);
            double num1 = 1. / dr2, num2 = 1. / (rm2 * dthe2), num3 = 1. / (rm2sinthe2 * dphi2);
            for (int k = 1; k < km - 1; k++) {
                double drhs_udr = (AU_(i + 1, j, k) - AU_(i - 1, j, k)) / (2. * dr);
                if (i <= im - 3) {
                    int above = hyd ? (A3(c, ATOMDEV_H, i + 1, j, k) == 0.) : is_fluid(c, i + 1, j, k);
                    if (is_land(c, i, j, k) && above)
                        drhs_udr = (-3. * AU_(i, j, k) + 4. * AU_(i + 1, j, k) - AU_(i + 2, j, k)) / (2. * dr);
                } else drhs_udr = (AU_(i + 1, j, k) - AU_(i, j, k)) / dr;

                double drhs_vdthe = (AV_(la, j + 1, k) - AV_(la, j - 1, k)) / (2. * dthe * rm);
                if (is_land(c, i, j, k) && is_fluid(c, i, j + 1, k))
                    drhs_vdthe = (AV_(la, j + 1, k) - AV_(la, j, k)) / (dthe * rm);
                if (is_land(c, i, j, k) && is_fluid(c, i, j - 1, k))
                    drhs_vdthe = (AV_(la, j - 1, k) - AV_(la, j, k)) / (dthe * rm);
                if ((j >= 2) && (j <= jm - 3)) {
                    if (is_land(c, i, j, k) && (is_fluid(c, i, j + 1, k) && is_fluid(c, i, j + 2, k)))
                        drhs_vdthe = (-3. * AV_(la, j, k) + 4. * AV_(la, j + 1, k) - AV_(la, j + 2, k)) / (2. * dthe * rm);
                    if (is_land(c, i, j, k) && is_fluid(c, i, j - 1, k) && is_fluid(c, i, j - 2, k))
                        drhs_vdthe = (-3. * AV_(la, j, k) + 4. * AV_(la, j - 1, k) - AV_(la, j - 2, k)) / (2. * dthe * rm);
                }

                /* operand order of the divisor differs: Pressure_Atm.cpp:140 vs Pressure_Hyd.cpp:152 */
                double drhs_wdphi = hyd ? (AW_(la, j, k + 1) - AW_(la, j, k - 1)) / (2. * dphi * rmsinthe)
                                        : (AW_(la, j, k + 1) - AW_(la, j, k - 1)) / (2. * rmsinthe * dphi);
                if (is_land(c, i, j, k) && is_fluid(c, i, j, k + 1))
                    drhs_wdphi = hyd ? (AW_(la, j, k + 1) - AW_(la, j, k)) / (dphi * rmsinthe)
                                     : (AW_(la, j, k + 1) - AW_(la, j, k)) / (rmsinthe * dphi);
                /* atmosphere: air cell with land to the west; hydrosphere: land cell with water to the west */
                if (hyd ? (is_land(c, i, j, k) && is_fluid(c, i, j, k - 1)) : (is_fluid(c, i, j, k) && is_land(c, i, j, k - 1)))
                    drhs_wdphi = hyd ? (AW_(la, j, k - 1) - AW_(la, j, k)) / (dphi * rmsinthe)
                                     : (AW_(la, j, k - 1) - AW_(la, j, k)) / (rmsinthe * dphi);
                if ((k >= 2) && (k <= km - 3)) {
                    if (is_land(c, i, j, k) && is_fluid(c, i, j, k + 1) && is_fluid(c, i, j, k + 2))
                        drhs_wdphi = (-3. * AW_(la, j, k) + 4. * AW_(la, j, k + 1) - AW_(la, j, k + 2)) / (2. * rmsinthe * dphi);
                    if (is_land(c, i, j, k) && is_fluid(c, i, j, k - 1) && is_fluid(c, i, j, k - 2))
                        drhs_wdphi = (-3. * AW_(la, j, k) + 4. * AW_(la, j, k - 1) - AW_(la, j, k - 2)) / (2. * rmsinthe * dphi);
                }

                PD(i, j, k) = ((PN(i + 1, j, k) + PN(i - 1, j, k)) * num1
                               + (PN(i, j + 1, k) + PN(i, j - 1, k)) * num2
                               + (PN(i, j, k + 1) + PN(i, j, k - 1)) * num3
                               - rho * (drhs_udr + drhs_vdthe + drhs_wdphi)) / denom;
                if (!hyd && is_land(c, i, j, k)) PD(i, j, k) = .0; /* Pressure_Atm.cpp:157 only */
            }
        }
    }
    for (int j = 0; j < jm; j++)
        for (int k = 0; k < km; k++) {
            PD(0, j, k) = PD(3, j, k) - 3. * PD(2, j, k) + 3. * PD(1, j, k);
            PD(im - 1, j, k) = c43 * PD(im - 2, j, k) - c13 * PD(im - 3, j, k);
        }
    for (int i = 0; i < im; i++)
        for (int j = 0; j < jm; j++) {
            PD(i, j, 0) = c43 * PD(i, j, 1) - c13 * PD(i, j, 2);
            PD(i, j, km - 1) = c43 * PD(i, j, km - 2) - c13 * PD(i, j, km - 3);
            PD(i, j, 0) = PD(i, j, km - 1) = (PD(i, j, 0) + PD(i, j, km - 1)) / 2.;
        }
    for (int k = 0; k < km; k++)
        for (int i = 0; i < im; i++) {
            PD(i, 0, k) = c43 * PD(i, 1, k) - c13 * PD(i, 2, k);
            PD(i, jm - 1, k) = c43 * PD(i, jm - 2, k) - c13 * PD(i, jm - 3, k);
        }
}

/* Pressure_Atm::computePressure_2D (atmosphere/Pressure_Atm.cpp:195-301) and
 * Pressure_Hyd::computePressure_2D (hydrosphere/Pressure_Hyd.cpp:206-317), level L. */
static void pressure2d(oracle_ctx* c, int L, int hyd, double rho) {
    const int im = c->im, jm = c->jm, km = c->km;
    const double dthe = c->p.dthe, dphi = c->p.dphi;
    for (int k = 0; k < km; k++) {
        AV_(L, 0, k) = c43 * AV_(L, 1, k) - c13 * AV_(L, 2, k);
        AV_(L, jm - 1, k) = c43 * AV_(L, jm - 2, k) - c13 * AV_(L, jm - 3, k);
    }
    for (int j = 0; j < jm; j++) {
        AV_(L, j, 0) = c43 * AV_(L, j, 1) - c13 * AV_(L, j, 2);
        AV_(L, j, km - 1) = c43 * AV_(L, j, km - 2) - c13 * AV_(L, j, km - 3);
        AV_(L, j, 0) = AV_(L, j, km - 1) = (AV_(L, j, 0) + AV_(L, j, km - 1)) / 2.;
    }
    double dthe2 = dthe * dthe, dphi2 = dphi * dphi;
    double rm = c->rad[L], rm2 = rm * rm;
    for (int j = 1; j < jm - 1; j++) {
        double sinthe = sin(c->the[j]);
        double rmsinthe = rm * sinthe;
        double rm2sinthe2 = rmsinthe * rmsinthe;
        double denom = 2. / (rm2 * dthe2) + 2. / (rm2sinthe2 * dphi2);
        double num2 = 1. / (rm2 * dthe2), num3 = 1. / (rm2sinthe2 * dphi2);
        for (int k = 1; k < km - 1; k++) {
            double drhs_vdthe = (AV_(L, j + 1, k) - AV_(L, j - 1, k)) / (2. * dthe * rm);
            if (is_land(c, L, j, k) && is_fluid(c, L, j + 1, k)) {
                if (j < jm - 2 && is_fluid(c, L, j + 2, k))
                    drhs_vdthe = (-3. * AV_(L, j, k) + 4. * AV_(L, j + 1, k) - AV_(L, j + 2, k)) / (2. * dthe * rm);
                else
                    drhs_vdthe = (AV_(L, j + 1, k) - AV_(L, j, k)) / (dthe * rm);
            }
            if (is_land(c, L, j, k) && is_fluid(c, L, j - 1, k)) {
                if (j > 1 && is_fluid(c, L, j - 2, k))
                    drhs_vdthe = (-3. * AV_(L, j, k) + 4. * AV_(L, j - 1, k) - AV_(L, j - 2, k)) / (2. * dthe * rm);
                else
                    drhs_vdthe = (AV_(L, j - 1, k) - AV_(L, j, k)) / (dthe * rm);
            }
            double drhs_wdphi = (AW_(L, j, k + 1) - AW_(L, j, k - 1)) / (2. * dphi * rmsinthe);
            if (is_land(c, L, j, k) && is_fluid(c, L, j, k + 1)) {
                if (k < km - 2 && is_fluid(c, L, j, k + 2))
                    drhs_wdphi = (-3. * AW_(L, j, k) + 4. * AW_(L, j, k + 1) - AW_(L, j, k + 2)) / (2. * rmsinthe * dphi);
                else
                    drhs_wdphi = (AW_(L, j, k + 1) - AW_(L, j, k)) / (dphi * rmsinthe);
            }
            if (is_land(c, L, j, k) && is_fluid(c, L, j, k - 1)) {
                /* k >= 2 (atm) and k > 1 (hyd) are the same test */
                if (k >= 2 && is_fluid(c, L, j, k - 2))
                    drhs_wdphi = (-3. * AW_(L, j, k) + 4. * AW_(L, j, k - 1) - AW_(L, j, k - 2)) / (2. * rmsinthe * dphi);
                else
                    drhs_wdphi = (AW_(L, j, k - 1) - AW_(L, j, k)) / (dphi * rmsinthe);
            }
            PD(L, j, k) = ((PN(L, j + 1, k) + PN(L, j - 1, k)) * num2
                           + (PN(L, j, k + 1) + PN(L, j, k - 1)) * num3
                           - rho * (drhs_vdthe + drhs_wdphi)) / denom;
        }
    }
    for (int k = 0; k < km; k++) {
        if (hyd) { /* index slip, Pressure_Hyd.cpp:301-305 */
            PD(im - 1, im - 1, k) = 0.;
            PD(0, jm - 1, k) = 0.;
        } else {
            PD(0, 0, k) = 0.;
            PD(0, jm - 1, k) = 0.;
        }
    }
    for (int j = 1; j < jm - 1; j++) {
        PD(L, j, 0) = c43 * PD(L, j, 1) - c13 * PD(L, j, 2);
        PD(L, j, km - 1) = c43 * PD(L, j, km - 2) - c13 * PD(L, j, km - 3);
        PD(L, j, 0) = PD(L, j, km - 1) = (PD(L, j, 0) + PD(L, j, km - 1)) / 2.;
    }
}
#undef AU_
#undef AV_
#undef AW_
#undef PD
#undef PN

void oracle_atm_pressure3d(oracle_ctx* c) { pressure3d(c, 0, c->p.r_air); }
void oracle_atm_pressure2d(oracle_ctx* c) { pressure2d(c, 0, 0, c->p.r_air); }
void oracle_hyd_pressure3d(oracle_ctx* c) { pressure3d(c, 1, c->p.r_0_water); }
void oracle_hyd_pressure2d(oracle_ctx* c) { pressure2d(c, c->im - 1, 1, c->p.r_0_water); }

/* ------------------------------------------------------------------------- */
/* Boundary fills                                                            */
/* ------------------------------------------------------------------------- */

#define NEUMANN_LO(F, a0, a1, a2) (F a0 = c43 * F a1 - c13 * F a2)

/* BC_Atmosphere::BC_radius, atmosphere/BC_Atm.cpp:39-73 */
void oracle_atm_bc_radius(oracle_ctx* c) {
    const int im = c->im;
    static const int EX[5] = { ATOMDEV_P_DYN, ATOMDEV_C, ATOMDEV_CLOUD, ATOMDEV_ICE, ATOMDEV_CO2 };
    for (int j = 0; j < c->jm; j++)
        for (int k = 0; k < c->km; k++) {
            A3(c, ATOMDEV_U, 0, j, k) = 0.;
            A3(c, ATOMDEV_V, 0, j, k) = c43 * A3(c, ATOMDEV_V, 1, j, k) - c13 * A3(c, ATOMDEV_V, 2, j, k);
            A3(c, ATOMDEV_W, 0, j, k) = c43 * A3(c, ATOMDEV_W, 1, j, k) - c13 * A3(c, ATOMDEV_W, 2, j, k);
            A3(c, ATOMDEV_T, im - 1, j, k) = c->p.t_tropopause;
            A3(c, ATOMDEV_U, im - 1, j, k) = 0.;
            A3(c, ATOMDEV_V, im - 1, j, k) = 0.;
            A3(c, ATOMDEV_W, im - 1, j, k) = 0.;
            for (int n = 0; n < 5; n++)
                A3(c, EX[n], im - 1, j, k) = A3(c, EX[n], im - 4, j, k) - 3. * A3(c, EX[n], im - 3, j, k) + 3. * A3(c, EX[n], im - 2, j, k);
        }
}

/* BC_Atmosphere::BC_theta, atmosphere/BC_Atm.cpp:80-107 */
void oracle_atm_bc_theta(oracle_ctx* c) {
    const int jm = c->jm;
    static const int NE[6] = { ATOMDEV_T, ATOMDEV_U, ATOMDEV_C, ATOMDEV_CLOUD, ATOMDEV_ICE, ATOMDEV_CO2 };
    static const int ZE[3] = { ATOMDEV_V, ATOMDEV_W, ATOMDEV_P_DYN };
    for (int k = 0; k < c->km; k++)
        for (int i = 0; i < c->im; i++) {
            for (int n = 0; n < 6; n++) {
                A3(c, NE[n], i, 0, k) = c43 * A3(c, NE[n], i, 1, k) - c13 * A3(c, NE[n], i, 2, k);
                A3(c, NE[n], i, jm - 1, k) = c43 * A3(c, NE[n], i, jm - 2, k) - c13 * A3(c, NE[n], i, jm - 3, k);
            }
            for (int n = 0; n < 3; n++) { A3(c, ZE[n], i, 0, k) = 0.; A3(c, ZE[n], i, jm - 1, k) = 0.; }
        }
}

static void phi_seam(oracle_ctx* c, const int* F, int nf, int j0, int j1) {
    const int km = c->km;
    for (int i = 0; i < c->im; i++)
        for (int j = j0; j < j1; j++)
            for (int n = 0; n < nf; n++) {
                A3(c, F[n], i, j, 0) = c43 * A3(c, F[n], i, j, 1) - c13 * A3(c, F[n], i, j, 2);
                A3(c, F[n], i, j, km - 1) = c43 * A3(c, F[n], i, j, km - 2) - c13 * A3(c, F[n], i, j, km - 3);
                A3(c, F[n], i, j, 0) = A3(c, F[n], i, j, km - 1) = (A3(c, F[n], i, j, 0) + A3(c, F[n], i, j, km - 1)) / 2.;
            }
}

/* BC_Atmosphere::BC_phi, atmosphere/BC_Atm.cpp:115-150 (j = 1 .. jm-2 only) */
void oracle_atm_bc_phi(oracle_ctx* c) {
    static const int F[9] = { ATOMDEV_T, ATOMDEV_U, ATOMDEV_V, ATOMDEV_W, ATOMDEV_P_DYN, ATOMDEV_C, ATOMDEV_CLOUD, ATOMDEV_ICE, ATOMDEV_CO2 };
    phi_seam(c, F, 9, 1, c->jm - 1);
}

/* BC_Hydrosphere::RB_radius, hydrosphere/BC_Hyd.cpp:38-83 */
void oracle_hyd_bc_radius(oracle_ctx* c) {
    const int im = c->im;
    for (int j = 0; j < c->jm; j++)
        for (int k = 0; k < c->km; k++) {
            A3(c, ATOMDEV_U, 0, j, k) = c43 * A3(c, ATOMDEV_U, 1, j, k) - c13 * A3(c, ATOMDEV_U, 2, j, k);
            A3(c, ATOMDEV_T, 0, j, k) = c->p.ta;
            A3(c, ATOMDEV_C, 0, j, k) = c->p.ca;
            A3(c, ATOMDEV_V, 0, j, k) = 0.;
            A3(c, ATOMDEV_W, 0, j, k) = 0.;
            A3(c, ATOMDEV_P_DYN, 0, j, k) = 0.;
            A3(c, ATOMDEV_V, im - 1, j, k) = c43 * A3(c, ATOMDEV_V, im - 2, j, k) - c13 * A3(c, ATOMDEV_V, im - 3, j, k);
            A3(c, ATOMDEV_W, im - 1, j, k) = c43 * A3(c, ATOMDEV_W, im - 2, j, k) - c13 * A3(c, ATOMDEV_W, im - 3, j, k);
            A3(c, ATOMDEV_U, im - 1, j, k) = A3(c, ATOMDEV_U, im - 4, j, k) - 3. * A3(c, ATOMDEV_U, im - 3, j, k) + 3. * A3(c, ATOMDEV_U, im - 2, j, k);
            A3(c, ATOMDEV_C, im - 1, j, k) = A3(c, ATOMDEV_C, im - 4, j, k) - 3. * A3(c, ATOMDEV_C, im - 3, j, k) + 3. * A3(c, ATOMDEV_C, im - 2, j, k);
        }
}

/* BC_Hydrosphere::RB_theta, hydrosphere/BC_Hyd.cpp:90-118 */
void oracle_hyd_bc_theta(oracle_ctx* c) {
    const int jm = c->jm;
    static const int NE[4] = { ATOMDEV_T, ATOMDEV_U, ATOMDEV_P_DYN, ATOMDEV_C };
    for (int i = 0; i < c->im; i++)
        for (int k = 0; k < c->km; k++) {
            for (int n = 0; n < 4; n++) {
                A3(c, NE[n], i, 0, k) = c43 * A3(c, NE[n], i, 1, k) - c13 * A3(c, NE[n], i, 2, k);
                A3(c, NE[n], i, jm - 1, k) = c43 * A3(c, NE[n], i, jm - 2, k) - c13 * A3(c, NE[n], i, jm - 3, k);
            }
            A3(c, ATOMDEV_V, i, 0, k) = 0.; A3(c, ATOMDEV_V, i, jm - 1, k) = 0.;
            A3(c, ATOMDEV_W, i, 0, k) = 0.; A3(c, ATOMDEV_W, i, jm - 1, k) = 0.;
        }
}

/* BC_Hydrosphere::RB_phi, hydrosphere/BC_Hyd.cpp:125-158 (all j) */
void oracle_hyd_bc_phi(oracle_ctx* c) {
    static const int F[6] = { ATOMDEV_T, ATOMDEV_U, ATOMDEV_V, ATOMDEV_W, ATOMDEV_P_DYN, ATOMDEV_C };
    phi_seam(c, F, 6, 0, c->jm);
}

/* ------------------------------------------------------------------------- */
/* Limiters, solid ground, old -> new copy                                   */
/* ------------------------------------------------------------------------- */

/* BC_Thermo::Value_Limitation_Atm, atmosphere/BC_Thermo.cpp:3235-3275 */
void oracle_atm_value_limitation(oracle_ctx* c) {
    for (int k = 0; k < c->km; k++)
        for (int j = 0; j < c->jm; j++)
            for (int i = 0; i < c->im; i++) {
                double* u = &A3(c, ATOMDEV_U, i, j, k); double* v = &A3(c, ATOMDEV_V, i, j, k);
                double* w = &A3(c, ATOMDEV_W, i, j, k); double* t = &A3(c, ATOMDEV_T, i, j, k);
                double* cc = &A3(c, ATOMDEV_C, i, j, k); double* cl = &A3(c, ATOMDEV_CLOUD, i, j, k);
                double* ic = &A3(c, ATOMDEV_ICE, i, j, k); double* co = &A3(c, ATOMDEV_CO2, i, j, k);
                if (*u >= .106) *u = .106;
                if (*u <= -.106) *u = -.106;
                if (*v >= .125) *v = .125;
                if (*v <= -.125) *v = -.125;
                if (*w >= 3.5) *w = 3.5;
                if (*w <= -.469) *w = -.469;
                if (*t >= 1.165) *t = 1.165;
                if (*t <= -.78) *t = -.78;
                if (*cc >= .03) *cc = .03;
                if (*cc < 0.) *cc = 0.;
                if (*cl >= .01) *cl = .01;
                if (*cl < 0.) *cl = 0.;
                if (*ic >= .005) *ic = .005;
                if (*ic < 0.) *ic = 0.;
                if (*co >= 5.36) *co = 5.36;
                if (*co <= 1.) *co = 1.;
                if (is_land(c, i, j, k)) {
                    *u = 0.; *v = 0.; *w = 0.; *cl = 0.; *ic = 0.; *co = 1.;
                    A3(c, ATOMDEV_P_DYN, i, j, k) = 0.;
                }
            }
}

/* BC_Bathymetry_Atmosphere::BC_SolidGround, atmosphere/BC_Bath_Atm.cpp:94-119
 * (levels im-2 .. 0; the top level is left alone) */
void oracle_atm_solid_ground(oracle_ctx* c) {
    for (int j = 0; j < c->jm; j++)
        for (int k = 0; k < c->km; k++)
            for (int i = c->im - 2; i >= 0; i--)
                if (is_land(c, i, j, k)) {
                    A3(c, ATOMDEV_U, i, j, k) = 0.; A3(c, ATOMDEV_V, i, j, k) = 0.; A3(c, ATOMDEV_W, i, j, k) = 0.;
                    A3(c, ATOMDEV_CLOUD, i, j, k) = 0.; A3(c, ATOMDEV_ICE, i, j, k) = 0.;
                    A3(c, ATOMDEV_CO2, i, j, k) = 1.; A3(c, ATOMDEV_P_DYN, i, j, k) = 0.;
                }
}

/* BC_Thermohalin::Value_Limitation_Hyd, hydrosphere/BC_Thermohalin.cpp:699-722 */
void oracle_hyd_value_limitation(oracle_ctx* c) {
    for (int k = 0; k < c->km; k++)
        for (int j = 0; j < c->jm; j++)
            for (int i = 0; i < c->im; i++) {
                double* u = &A3(c, ATOMDEV_U, i, j, k); double* v = &A3(c, ATOMDEV_V, i, j, k);
                double* w = &A3(c, ATOMDEV_W, i, j, k); double* t = &A3(c, ATOMDEV_T, i, j, k);
                double* cc = &A3(c, ATOMDEV_C, i, j, k);
                if (*u >= 11.11) *u = 11.11;
                if (*u <= -11.11) *u = -11.11;
                if (*v >= .552) *v = .552;
                if (*v <= -.552) *v = -.552;
                if (*w >= .552) *w = .552;
                if (*w <= -.552) *w = -.552;
                if (*t >= 1.147) *t = 1.147;
                if (*t <= -1.01464) *t = -1.01464;
                if (*cc >= 1.1) *cc = 1.1;
                if (*cc < .95) *cc = .95;
            }
}

/* Results_MSL_Atm::C_Dalton, atmosphere/Results_Atm.cpp:685-692 */
static double C_Dalton(double u_0, double v, double w) {
    double C_max = -.053, v_max = 10.;
    double vel_magnitude = sqrt(v * v + w * w) * u_0;
    return sqrt(C_max * C_max / v_max * vel_magnitude);
}

/* Results_MSL_Atm::run_MSL_data, atmosphere/Results_Atm.cpp:118-353: everything that changes model
 * fields; the averages and the printout that follow (:355-625) are not restated.  The reference tests
 * is_air(h, i+1, j, k) for i = im-1 as well, one level above the array, when the top cell is land;
 * here that cell counts as not air. */
void oracle_atm_msl_data(oracle_ctx* c, int RadiationModel) {
    const atomdev_params* P = &c->p;
    const int im = c->im, jm = c->jm, km = c->km;
    const double r_air = P->r_air, R_Air = P->R_Air, t_0 = P->t_0, g = P->g, L_atm = P->L_atm, ep = P->ep, hp = P->hp;
    const double cp_l = P->cp_l, lv = P->lv, u_0 = P->u_0, R_WaterVapour = P->R_WaterVapour;
    const double f_Penman = 2., c43 = 4. / 3., c13 = 1. / 3.;
#define T_(i) A3(c, ATOMDEV_T, i, j, k)
#define PS_(i) A3(c, ATOMDEV_P_STAT, i, j, k)
#define C_(i) A3(c, ATOMDEV_C, i, j, k)
#define V_(i) A3(c, ATOMDEV_V, i, j, k)
#define W_(i) A3(c, ATOMDEV_W, i, j, k)
    for (int k = 0; k < km; k++)
        for (int j = 0; j < jm; j++) {
            if (RadiationModel >= 1) A2(c, ATOMDEV_Q_RADIATION, j, k) = A3(c, ATOMDEV_RADIATION_3D, 0, j, k);
            A2(c, ATOMDEV_PRECIPITABLE_WATER, j, k) = 0.;
            A2(c, ATOMDEV_EVAPORATION_PENMAN, j, k) = 0.;
            A2(c, ATOMDEV_EVAPORATION_DALTON, j, k) = 0.;
        }
    for (int k = 0; k < km; k++)
        for (int j = 0; j < jm; j++)
            for (int i = 0; i < im; i++) {
                double t_Celsius, r_dry, r_humid, e, t_denom, E, Delta, sat_deficit, gamma, E_a;
                /* top of the topography (:153-201) */
                if (is_land(c, i, j, k) && (i + 1 < im) && is_fluid(c, i + 1, j, k)) {
                    if (i == 0) PS_(0) = (r_air * R_Air * T_(0) * t_0) * .01;
                    else PS_(i) = exp(-g * (double)i * (L_atm / (double)(im - 1)) / (R_Air * T_(i) * t_0)) * PS_(0);
                    t_Celsius = T_(i) * t_0 - t_0;
                    r_dry = 100. * PS_(i) / (R_Air * T_(i) * t_0);
                    r_humid = r_dry / (1. + (R_WaterVapour / R_Air - 1.) * C_(i));
                    e = C_(i) * PS_(i) / ep;
                    t_denom = t_Celsius + 234.175;
                    E = hp * exp(17.0809 * t_Celsius / t_denom);
                    Delta = 4000. * E / (t_denom * t_denom);
                    sat_deficit = (E - e);
                    gamma = PS_(0) * cp_l / (ep * lv);
                    E_a = .35 * (1. + .15 * sqrt((V_(i + 1) * V_(i + 1) + W_(i + 1) * W_(i + 1)) / 2.) * u_0 * 3.6) * sat_deficit;
                    A2(c, ATOMDEV_Q_EVAPORATION, j, k) = 2300.;
                    A2(c, ATOMDEV_Q_LATENT_2D, j, k) = A3(c, ATOMDEV_Q_LATENT, i, j, k);
                    A2(c, ATOMDEV_Q_SENSIBLE_2D, j, k) = A3(c, ATOMDEV_Q_SENSIBLE, i, j, k);
                    A2(c, ATOMDEV_Q_BOTTOM, j, k) = -(A2(c, ATOMDEV_Q_RADIATION, j, k) - A2(c, ATOMDEV_Q_LATENT_2D, j, k)
                                                      - A2(c, ATOMDEV_Q_SENSIBLE_2D, j, k));
                    double ed = C_Dalton(u_0, V_(i), W_(i)) * sat_deficit * 24.;
                    if (ed <= 0.) ed = 0.;
                    double epm = f_Penman * .0346 * ((A2(c, ATOMDEV_Q_RADIATION, j, k) + A2(c, ATOMDEV_Q_BOTTOM, j, k)) * Delta
                                                     + gamma * E_a) / (Delta + gamma);
                    if (epm <= 0.) epm = 0.;
                    A2(c, ATOMDEV_EVAPORATION_PENMAN, j, k) = epm;
                    A2(c, ATOMDEV_EVAPORATION_DALTON, j, k) = .5 * epm; /* the centre is land (:200) */
                    (void)ed; (void)r_humid;
                }
                /* sea surface (:204-240) */
                if ((i == 0) && is_fluid(c, 0, j, k)) {
                    PS_(0) = (r_air * R_Air * T_(0) * t_0) * .01;
                    t_Celsius = T_(0) * t_0 - t_0;
                    r_dry = 100. * PS_(0) / (R_Air * T_(0) * t_0);
                    r_humid = r_dry / (1. + (R_WaterVapour / R_Air - 1.) * C_(i));
                    e = C_(i) * PS_(i) / ep;
                    t_denom = t_Celsius + 234.175;
                    E = hp * exp(17.0809 * t_Celsius / t_denom);
                    Delta = 4000. * E / (t_denom * t_denom);
                    sat_deficit = (E - e);
                    gamma = PS_(0) * cp_l / (ep * lv);
                    E_a = .35 * (1. + .15 * sqrt((V_(1) * V_(1) + W_(1) * W_(1)) / 2.) * u_0 * 3.6) * sat_deficit;
                    if (t_Celsius >= -2.) A2(c, ATOMDEV_Q_EVAPORATION, j, k) = (2500.8 - 2.372 * (T_(0) * t_0 - t_0));
                    else A2(c, ATOMDEV_Q_EVAPORATION, j, k) = (2500.8 - 2.372 * (T_(0) * t_0 - t_0)) + 300.;
                    A2(c, ATOMDEV_Q_LATENT_2D, j, k) = A3(c, ATOMDEV_Q_LATENT, 0, j, k);
                    A2(c, ATOMDEV_Q_SENSIBLE_2D, j, k) = A3(c, ATOMDEV_Q_SENSIBLE, 0, j, k);
                    A2(c, ATOMDEV_Q_BOTTOM, j, k) = -(A2(c, ATOMDEV_Q_RADIATION, j, k) - A2(c, ATOMDEV_Q_LATENT_2D, j, k)
                                                      - A2(c, ATOMDEV_Q_SENSIBLE_2D, j, k));
                    double ed = C_Dalton(u_0, V_(0), W_(0)) * sat_deficit * 24.;
                    if (ed <= 0.) ed = 0.;
                    A2(c, ATOMDEV_EVAPORATION_DALTON, j, k) = ed;
                    double epm = f_Penman * .0346 * ((A2(c, ATOMDEV_Q_RADIATION, j, k) + A2(c, ATOMDEV_Q_BOTTOM, j, k)) * Delta
                                                     + gamma * E_a) / (Delta + gamma);
                    if (epm <= 0.) epm = 0.;
                    A2(c, ATOMDEV_EVAPORATION_PENMAN, j, k) = epm;
                    (void)r_humid;
                }
            }
    /* boundary values of BuoyancyForce (:248-278), r then theta then phi */
#define B_(i, j, k) A3(c, ATOMDEV_BUOYANCY, i, j, k)
    for (int k = 1; k < km - 1; k++)
        for (int j = 1; j < jm - 1; j++) {
            B_(0, j, k) = c43 * B_(1, j, k) - c13 * B_(2, j, k);
            B_(im - 1, j, k) = c43 * B_(im - 2, j, k) - c13 * B_(im - 3, j, k);
            if (is_land(c, 0, j, k)) B_(0, j, k) = 0.;
        }
    for (int k = 0; k < km; k++)
        for (int i = 0; i < im; i++) {
            B_(i, 0, k) = c43 * B_(i, 1, k) - c13 * B_(i, 2, k);
            B_(i, jm - 1, k) = c43 * B_(i, jm - 2, k) - c13 * B_(i, jm - 3, k);
        }
    for (int i = 0; i < im; i++)
        for (int j = 1; j < jm - 1; j++) {
            B_(i, j, 0) = c43 * B_(i, j, 1) - c13 * B_(i, j, 2);
            B_(i, j, km - 1) = c43 * B_(i, j, km - 2) - c13 * B_(i, j, km - 3);
            B_(i, j, 0) = B_(i, j, km - 1) = (B_(i, j, 0) + B_(i, j, km - 1)) / 2.;
        }
#undef B_
    /* :280-290 */
    static const int CL[5] = { ATOMDEV_C, ATOMDEV_CLOUD, ATOMDEV_ICE, ATOMDEV_P_RAIN, ATOMDEV_P_SNOW };
    for (int n = 0; n < 5; n++) {
        double* x = c->f[CL[n]];
        for (size_t q = 0; q < (size_t)im * jm * km; q++) if (x[q] < 0.) x[q] = 0.;
    }
    /* :328-353 */
    for (int j = 0; j < jm; j++)
        for (int k = 0; k < km; k++) {
            A2(c, ATOMDEV_CO2_TOTAL, j, k) = A3(c, ATOMDEV_CO2, 0, j, k);
            for (int i = 0; i < im; i++) {
                double e = 100. * C_(i) * PS_(i) / ep;
                double a = e / (R_WaterVapour * T_(i) * t_0);
                A2(c, ATOMDEV_PRECIPITABLE_WATER, j, k) += a * L_atm / (double)(im - 1);
            }
            double pr = 86400. * (A3(c, ATOMDEV_P_RAIN, 0, j, k) + A3(c, ATOMDEV_P_SNOW, 0, j, k));
            if (pr >= 25.) pr = 25.;
            if (pr <= 0) pr = 0.;
            A2(c, ATOMDEV_PRECIPITATION, j, k) = pr;
        }
#undef T_
#undef PS_
#undef C_
#undef V_
#undef W_
}

/* BC_Thermo::Latent_Heat, atmosphere/BC_Thermo.cpp:2418-2502, with i_topography as the constructor builds it
 * (:128-139: the highest land level at or below im-2, else 0).  Pinned bit for bit by the compiled class
 * (tests/test_oracle_vs_reference.py, tests/golden/bc_thermo.npz).  The
 * saturation pressures the function evaluates (:2450-2453, :2488-2491) are never used and are left out; the
 * bottom formula divides by R_WaterVapour (:2455), the column formula by r_water_vapour (:2493).  A column
 * whose topography reaches level im-2 would read level im (outside the array) and is skipped. */
void oracle_atm_latent_heat(oracle_ctx* c) {
    const atomdev_params* P = &c->p;
    const int im = c->im, jm = c->jm, km = c->km;
    const double step = P->L_atm / (double)(im - 1), dr = P->dr, ep = P->ep, t_0 = P->t_0;
    const double coeff_Lv = P->lv / step, coeff_Ls = P->ls / step, coeff_Q = P->cp_l * P->r_air * t_0 / step;
    const double coeff_lat = .079, coeff_sen = .15;
    for (int j = 0; j < jm; j++)
        for (int k = 0; k < km; k++) {
            int i_mount = 0;
            for (int i = im - 2; i >= 0; i--) if (is_land(c, i, j, k)) { i_mount = i; break; }
            if (i_mount + 2 > im - 1) continue;
            {
                const int i = i_mount;
                double e = .01 * A3(c, ATOMDEV_C, i, j, k) * A3(c, ATOMDEV_P_STAT, i, j, k) / ep;
                double a = e / (P->R_WaterVapour * A3(c, ATOMDEV_T, i, j, k) * t_0);
                double ql = -coeff_Lv * a * (-3. * A3(c, ATOMDEV_C, i, j, k) + 4. * A3(c, ATOMDEV_C, i + 1, j, k)
                                             - A3(c, ATOMDEV_C, i + 2, j, k)) / (2. * dr);
                double qi = -coeff_Ls * a * (-3. * A3(c, ATOMDEV_ICE, i, j, k) + 4. * A3(c, ATOMDEV_ICE, i + 1, j, k)
                                             - A3(c, ATOMDEV_ICE, i + 2, j, k)) / (2. * dr);
                A3(c, ATOMDEV_Q_LATENT, i, j, k) = coeff_lat * (ql + qi);
                A3(c, ATOMDEV_Q_SENSIBLE, i, j, k) = -coeff_sen * coeff_Q * (-3. * A3(c, ATOMDEV_T, i, j, k)
                    + 4. * A3(c, ATOMDEV_T, i + 1, j, k) - A3(c, ATOMDEV_T, i + 2, j, k)) / (2. * dr);
            }
        }
    for (int j = 0; j < jm; j++)
        for (int k = 0; k < km; k++) {
            int i_mount = 0;
            for (int i = im - 2; i >= 0; i--) if (is_land(c, i, j, k)) { i_mount = i; break; }
            for (int i = i_mount + 1; i < im - 2; i++) {
                double e = .01 * A3(c, ATOMDEV_C, i, j, k) * A3(c, ATOMDEV_P_STAT, i, j, k) / ep;
                double a = e / (P->r_water_vapour * A3(c, ATOMDEV_T, i, j, k) * t_0);
                double ql = -coeff_Lv * a * (A3(c, ATOMDEV_C, i + 1, j, k) - A3(c, ATOMDEV_C, i - 1, j, k)) / (2. * dr);
                double qi = -coeff_Ls * a * (A3(c, ATOMDEV_ICE, i + 1, j, k) - A3(c, ATOMDEV_ICE, i - 1, j, k)) / (2. * dr);
                A3(c, ATOMDEV_Q_LATENT, i, j, k) = coeff_lat * (ql + qi);
                A3(c, ATOMDEV_Q_SENSIBLE, i, j, k) = -coeff_sen * coeff_Q * (A3(c, ATOMDEV_T, i + 1, j, k)
                    - A3(c, ATOMDEV_T, i - 1, j, k)) / (2. * dr);
            }
        }
}

/* lib/Utils.h:79-81 */
static double exp_func(double T_K, double co_1, double co_2) { return exp(co_1 * (T_K - 273.15) / (T_K - co_2)); }

/* BC_Thermo::Two_Category_Ice_Scheme, atmosphere/BC_Thermo.cpp:2803-3143, pinned bit for bit by the compiled class
 * (tests/test_oracle_vs_reference.py, tests/golden/bc_thermo.npz): a statement-by-statement restatement of the source text in the reference's traversal order.
 * S_nuc and S_i_dep are class members there, assigned only under conditions (:2923-2931, :2995-3016), so their
 * values travel from grid point to grid point, pass to pass and call to call: state[0], state[1].  N_i (member)
 * and m_i (local) are carried too but are only read where they have just been assigned.  dt_snow_dim = 417,
 * dt_rain_dim = 250 (:113-119). */
void oracle_atm_two_category_ice_scheme(oracle_ctx* c, double state[2]) {
    const atomdev_params* P = &c->p;
    const int im = c->im, jm = c->jm, km = c->km;
    const double N_i_0 = 1.e2, m_i_0 = 1.e-12, m_i_max = 1.e-9, m_s_0 = 3.e-9;
    const double c_i_dep = 1.3e-5, c_c_au = 4.e-4, c_i_au = 1.e-3, c_ac = .24, c_rim = 18.6, c_agg = 10.3, c_i_cri = .24,
                 c_r_cri = 3.2e-5, a_ev = 1.e-3, b_ev = 5.9, c_s_dep = 1.8e-2, b_s_dep = 12.3, c_s_melt = 8.43e-5,
                 b_s_melt = 12.05, a_s_melt = 2.31e3, c_r_frz = 3.75e-2;
    const double t_nuc = 267.15, t_d = 248.15, t_hn = 236.15, t_r_frz = 271.15;
    const double dt_snow_dim = 417., dt_rain_dim = 250.;
    const double t_0 = P->t_0, g = P->g, gam = P->gam, R_Air = P->R_Air, r_air = P->r_air, R_WaterVapour = P->R_WaterVapour,
                 L_atm = P->L_atm, hp = P->hp, ep = P->ep, dr = P->dr;
    const double exp_pressure = g / (1.e-2 * gam * R_Air);
    double m_i = m_i_max, N_i = 0., S_nuc = state[0], S_i_dep = state[1];
    double S_c_au, S_i_au, S_c_frz, S_d_au, S_ac, S_rim, S_shed, S_agg, S_i_cri, S_r_cri, S_ev, S_s_dep, S_i_melt, S_s_melt, S_r_frz;
#define F_(N, i) A3(c, ATOMDEV_##N, i, j, k)
    static const int CL[5] = { ATOMDEV_C, ATOMDEV_CLOUD, ATOMDEV_ICE, ATOMDEV_P_RAIN, ATOMDEV_P_SNOW };
    for (int n = 0; n < 5; n++) {
        double* x = c->f[CL[n]];
        for (size_t q = 0; q < (size_t)im * jm * km; q++) if (x[q] < 0.) x[q] = 0.;
    }
    /* initial values for rain and snow (:2875-2954) */
    for (int k = 0; k < km; k++)
        for (int j = 0; j < jm; j++) {
            F_(P_RAIN, im - 1) = 0.; F_(P_SNOW, im - 1) = 0.; F_(S_R, im - 1) = 0.; F_(S_S, im - 1) = 0.;
            for (int i = im - 2; i >= 0; i--) {
                if (F_(C, i) < 0.) F_(C, i) = 0.;
                if (F_(CLOUD, i) < 0.) F_(CLOUD, i) = 0.;
                if (F_(ICE, i) < 0.) F_(ICE, i) = 0.;
                double t_Celsius = F_(T, i) * t_0 - t_0;
                double t_u = F_(T, i) * t_0;
                double p_SL = .01 * (r_air * R_Air * F_(T, 0) * t_0);
                double hight = (double)i * (L_atm / (double)(im - 1));
                double p_h;
                if (i != 0) p_h = pow(((t_u - gam * hight * 1.e-2) / (t_u)), exp_pressure) * p_SL;
                else p_h = p_SL;
                double r_dry = 100. * p_h / (R_Air * t_u);
                double r_humid = r_dry * (1. + F_(C, i)) / (1. + R_WaterVapour / R_Air * F_(C, i));
                double E_Ice = hp * exp_func(t_u, 21.8746, 7.66);
                double q_Ice = ep * E_Ice / (p_h - E_Ice);
                if ((t_u >= t_0) && (c_c_au * F_(CLOUD, i) > 0.)) S_c_au = c_c_au * F_(CLOUD, i);
                else S_c_au = 0.;
                if ((t_u < t_0) && (c_i_au * F_(ICE, i) > 0.)) S_i_au = c_i_au * F_(ICE, i);
                else S_i_au = 0.;
                if (t_u <= t_0) N_i = N_i_0 * exp(.2 * (t_0 - t_u));
                if (t_u <= t_0) {
                    m_i = r_humid * F_(ICE, i) / N_i;
                    if (m_i > m_i_max) { m_i = m_i_max; }
                    else m_i = r_humid * F_(ICE, i) / N_i;
                    if (m_i < m_i_0) { m_i = m_i_0; }
                }
                if (t_Celsius <= 0.) {
                    if (F_(C, i) > q_Ice) S_i_dep = c_i_dep * N_i * pow(m_i, (1. / 3.)) * (F_(C, i) - q_Ice);
                    if ((F_(C, i) < q_Ice) && (-F_(ICE, i) / dt_snow_dim > (F_(C, i) - q_Ice) / dt_snow_dim))
                        S_i_dep = -F_(ICE, i) / dt_snow_dim;
                } else S_i_dep = 0.;
                F_(S_R, i) = S_c_au;
                F_(S_S, i) = S_i_au;
                if (is_land(c, i, j, k) && is_land(c, i + 1, j, k)) { F_(S_R, i) = 0.; F_(S_S, i) = 0.; }
                if (F_(P_RAIN, i + 1) < 0.) F_(P_RAIN, i + 1) = 0.;
                if (F_(P_SNOW, i + 1) < 0.) F_(P_SNOW, i + 1) = 0.;
                F_(P_RAIN, i) = F_(P_RAIN, i + 1) + (F_(S_R, i) + F_(S_R, i + 1)) * .5 * r_humid * dr * 200.;
                F_(P_SNOW, i) = F_(P_SNOW, i + 1) + (F_(S_S, i) + F_(S_S, i + 1)) * .5 * r_humid * dr * 200.;
                if (F_(P_RAIN, i) < 0.) F_(P_RAIN, i) = 0.;
                if (F_(P_SNOW, i) < 0.) F_(P_SNOW, i) = 0.;
            }
        }
    /* main part (:2959-3135) */
    for (int iter_prec = 1; iter_prec <= 5; iter_prec++)
        for (int k = 0; k < km; k++)
            for (int j = 0; j < jm; j++) {
                F_(P_RAIN, im - 1) = 0.; F_(P_SNOW, im - 1) = 0.;
                for (int i = im - 2; i >= 0; i--) {
                    double t_u = F_(T, i) * t_0;
                    double t_Celsius = t_u - t_0;
                    double p_SL = .01 * (r_air * R_Air * F_(T, 0) * t_0);
                    double hight = (double)i * (L_atm / (double)(im - 1));
                    double p_h;
                    if (i != 0) p_h = pow(((t_u - gam * hight * 1.e-2) / (t_u)), exp_pressure) * p_SL;
                    else p_h = p_SL;
                    double r_dry = 100. * p_h / (R_Air * t_u);
                    double r_humid = r_dry * (1. + F_(C, i)) / (1. + R_WaterVapour / R_Air * F_(C, i));
                    double E_Rain = hp * exp_func(t_u, 17.2694, 35.86);
                    double E_Ice = hp * exp_func(t_u, 21.8746, 7.66);
                    double q_Rain = ep * E_Rain / (p_h - E_Rain);
                    double q_Ice = ep * E_Ice / (p_h - E_Ice);
                    const double cc = F_(C, i), cl = F_(CLOUD, i), ic = F_(ICE, i), pr = F_(P_RAIN, i), ps = F_(P_SNOW, i);
                    if (t_u <= t_0) N_i = N_i_0 * exp(.2 * (t_0 - t_u));
                    if (t_u <= t_0) {
                        m_i = r_humid * ic / N_i;
                        if (m_i > m_i_max) { m_i = m_i_max; }
                        else m_i = r_humid * ic / N_i;
                        if (m_i < m_i_0) { m_i = m_i_0; }
                    }
                    if (ic == 0.) {
                        if ((t_u < t_d) && (cc >= q_Ice)) S_nuc = m_i_0 / (r_humid * dt_snow_dim) * N_i;
                        if (((t_d <= t_u) && (t_u <= t_nuc)) && (cc >= q_Rain)) S_nuc = m_i_0 / (r_humid * dt_snow_dim) * N_i;
                    } else S_nuc = 0.;
                    if ((t_u < t_hn) && (cl > 0.)) S_c_frz = cl / dt_rain_dim;
                    else S_c_frz = 0.;
                    if (t_Celsius <= 0.) {
                        if (cc > q_Ice) S_i_dep = c_i_dep * N_i * pow(m_i, (1. / 3.)) * (cc - q_Ice);
                        if ((cc < q_Ice) && (-ic / dt_snow_dim > (cc - q_Ice) / dt_snow_dim)) S_i_dep = -ic / dt_snow_dim;
                    } else S_i_dep = 0.;
                    if ((t_u >= t_0) && (cl > 0.)) S_c_au = c_c_au * cl;
                    else S_c_au = 0.;
                    if ((t_u <= t_0) && (ic > 0.)) S_i_au = c_i_au * ic;
                    else S_i_au = 0.;
                    if (t_u <= t_0) S_d_au = S_i_dep / (1.5 * (pow((m_s_0 / m_i), (2. / 3.)) - 1.));
                    else S_d_au = 0.;
                    if (t_u > t_0) S_ac = c_ac * cl * pow(pr, (7. / 9.));
                    else S_ac = 0.;
                    if (t_u < t_0) S_rim = c_rim * cl * ps;
                    else S_rim = 0.;
                    if (t_u >= t_0) S_shed = c_rim * cl * ps;
                    else S_shed = 0.;
                    if (t_u <= t_0) {
                        S_agg = c_agg * ic * ps;
                        S_i_cri = c_i_cri * ic * pow(pr, (7. / 9.));
                        S_r_cri = c_r_cri * ic / m_i * pow(pr, (13. / 9.));
                    } else { S_agg = 0.; S_i_cri = 0.; S_r_cri = 0.; }
                    if (t_u >= t_0) S_ev = a_ev * (1. + b_ev * pow(pr, (1. / 6.))) * (q_Rain - cc) * pow(pr, (4. / 9.));
                    else S_ev = 0.;
                    if (t_u < t_0) S_s_dep = c_s_dep * (1. + b_s_dep * pow(ps, (5. / 26.))) * (cc - q_Ice) * pow(ps, (8. / 13.));
                    else S_s_dep = 0.;
                    if ((t_u > t_0) && (ic > 0.)) S_i_melt = ic / dt_snow_dim;
                    else S_i_melt = 0.;
                    if (t_u > t_0) {
                        p_SL = .01 * (r_air * R_Air * F_(T, 0) * t_0);
                        hight = (double)i * (L_atm / (double)(im - 1));
                        double p_t_in = pow(((t_0 - gam * hight * 1.e-2) / t_0), exp_pressure) * p_SL;
                        double E_Rain_t_in = hp * exp_func(t_0, 17.2694, 35.86);
                        double q_Rain_t_in = ep * E_Rain_t_in / (p_t_in - E_Rain_t_in);
                        S_s_melt = c_s_melt * (1. + b_s_melt * pow(ps, (5. / 26.))) * ((t_u - t_0) + a_s_melt * (cc - q_Rain_t_in))
                                   * pow(ps, (8. / 13.));
                    } else S_s_melt = 0.;
                    if (t_r_frz - t_u > 0.) S_r_frz = c_r_frz * pow((t_r_frz - t_u), (3. / 2.)) * pow(pr, (3. / 2.));
                    else S_r_frz = 0.;
                    F_(S_V, i) = -F_(S_C_C, i) + S_ev - S_i_dep - S_s_dep - S_nuc;
                    F_(S_C, i) = F_(S_C_C, i) - S_c_au - S_ac - S_c_frz + S_i_melt - S_rim - S_shed;
                    F_(S_I, i) = S_nuc + S_c_frz + S_i_dep - S_i_melt - S_i_au - S_d_au - S_agg - S_i_cri;
                    F_(S_R, i) = S_c_au + S_ac - S_ev + S_shed - S_r_cri - S_r_frz + S_s_melt;
                    F_(S_S, i) = S_d_au + S_s_dep + S_i_au + S_rim + S_agg + S_i_cri + S_r_cri + S_r_frz - S_s_melt;
                    if (is_land(c, i, j, k) && is_land(c, i + 1, j, k)) {
                        F_(S_C_C, i) = 0.; F_(S_V, i) = 0.; F_(S_C, i) = 0.; F_(S_I, i) = 0.; F_(S_R, i) = 0.; F_(S_S, i) = 0.;
                    }
                    if (F_(P_RAIN, i + 1) < 0.) F_(P_RAIN, i + 1) = 0.;
                    if (F_(P_SNOW, i + 1) < 0.) F_(P_SNOW, i + 1) = 0.;
                    F_(P_RAIN, i) = F_(P_RAIN, i + 1) + (F_(S_R, i) + F_(S_R, i + 1)) * .5 * r_humid * dr * 200.;
                    F_(P_SNOW, i) = F_(P_SNOW, i + 1) + (F_(S_S, i) + F_(S_S, i + 1)) * .5 * r_humid * dr * 200.;
                    if (F_(P_RAIN, i) < 0.) F_(P_RAIN, i) = 0.;
                    if (F_(P_SNOW, i) < 0.) F_(P_SNOW, i) = 0.;
                    if (F_(C, i) < 0.) F_(C, i) = 0.;
                    if (F_(CLOUD, i) < 0.) F_(CLOUD, i) = 0.;
                    if (F_(ICE, i) < 0.) F_(ICE, i) = 0.;
                }
            }
    state[0] = S_nuc; state[1] = S_i_dep;
#undef F_
}

/* BC_Thermo::Ice_Water_Saturation_Adjustment, atmosphere/BC_Thermo.cpp:2509-2796, pinned bit for bit by the compiled
 * class (tests/test_oracle_vs_reference.py, tests/golden/bc_thermo.npz): a statement-by-statement restatement of the source text in the reference's traversal
 * order (k outer, j, i upwards; a column's sea-level pressure uses t(0), which the column's first point has just
 * changed).  q_v_hyp is a class member assigned only under conditions (:2658-2661; none holds where cloud water or
 * cloud ice comes in negative), so it travels from grid point to grid point and call to call: *q_v_hyp_state. */
void oracle_atm_ice_water_saturation_adjustment(oracle_ctx* c, double* q_v_hyp_state) {
    const atomdev_params* P = &c->p;
    const int im = c->im, jm = c->jm, km = c->km;
    const double t_0 = P->t_0, g = P->g, gam = P->gam, R_Air = P->R_Air, r_air = P->r_air, R_WaterVapour = P->R_WaterVapour,
                 L_atm = P->L_atm, hp = P->hp, ep = P->ep, lv = P->lv, ls = P->ls, cp_l = P->cp_l;
    const double dt_dim = L_atm / P->u_0 * P->dt; /* :121 */
    const double t_00 = 236.15, t_Celsius_2 = t_00 - t_0;
    const double exp_pressure = g / (1.e-2 * gam * R_Air);
    double q_v_hyp = *q_v_hyp_state, T_it = 0.;
#define F_(N) A3(c, ATOMDEV_##N, i, j, k)
    for (int k = 0; k < km; k++)
        for (int j = 0; j < jm; j++)
            for (int i = 0; i < im; i++) {
                double t_u = F_(T) * t_0;
                double t_Celsius = t_u - t_0;
                double p_SL = .01 * (r_air * R_Air * A3(c, ATOMDEV_T, 0, j, k) * t_0);
                double hight = (double)i * (L_atm / (double)(im - 1));
                double p_h;
                if (i != 0) p_h = pow(((F_(T) * t_0 - gam * hight * 1.e-2) / (F_(T) * t_0)), exp_pressure) * p_SL;
                else p_h = p_SL;
                double E_Rain, E_Ice, q_Rain, q_Ice, q_T;
                if (t_Celsius >= 0.) { /* warm cloud phase */
                    q_T = F_(C) + F_(CLOUD);
                    E_Rain = hp * exp_func(t_u, 17.2694, 35.86);
                    q_Rain = ep * E_Rain / (p_h - E_Rain);
                    double q_Rain_n = q_Rain;
                    if (q_T <= q_Rain) {
                        F_(C) = q_T; F_(CLOUD) = 0.; F_(ICE) = 0.;
                        T_it = t_u;
                    } else {
                        for (int iter_prec = 1; iter_prec <= 20; iter_prec++) {
                            T_it = (t_u + lv / cp_l * F_(C) - lv / cp_l * q_Rain);
                            E_Rain = hp * exp_func(T_it, 17.2694, 35.86);
                            q_Rain = ep * E_Rain / (p_h - E_Rain);
                            q_Rain = .5 * (q_Rain_n + q_Rain);
                            F_(C) = q_Rain;
                            F_(CLOUD) = q_T - F_(C);
                            F_(ICE) = 0.;
                            q_T = F_(C) + F_(CLOUD);
                            if (F_(C) < 0.) F_(C) = 0.;
                            if (F_(CLOUD) < 0.) F_(CLOUD) = 0.;
                            if ((q_Rain_n) > DBL_EPSILON && fabs(q_Rain / q_Rain_n - 1.) <= 1.e-5) break;
                            q_Rain_n = q_Rain;
                        }
                    }
                    F_(CN) = F_(C); F_(CLOUDN) = F_(CLOUD); F_(ICEN) = F_(ICE);
                    F_(T) = T_it / t_0;
                }
                if (t_Celsius < 0.) { /* mixed cloud phase */
                    if (t_Celsius < t_Celsius_2) F_(CLOUD) = 0.;
                    if (t_Celsius > 0.) F_(ICE) = 0.;
                    double q_v_b = F_(C), q_c_b = F_(CLOUD), q_i_b = F_(ICE);
                    q_T = q_v_b + q_c_b + q_i_b;
                    t_u = F_(T) * t_0;
                    double T = t_u;
                    E_Rain = hp * exp_func(T, 17.2694, 35.86);
                    E_Ice = hp * exp_func(T, 21.8746, 7.66);
                    q_Rain = ep * E_Rain / (p_h - E_Rain);
                    q_Ice = ep * E_Ice / (p_h - E_Ice);
                    if ((q_c_b > 0.) && (q_i_b > 0.)) q_v_hyp = (q_c_b * q_Rain + q_i_b * q_Ice) / (q_c_b + q_i_b);
                    if ((q_c_b >= 0.) && (q_i_b == 0.)) q_v_hyp = q_Rain;
                    if ((q_c_b == 0.) && (q_i_b > 0.)) q_v_hyp = q_Ice;
                    for (int iter_prec = 1; iter_prec <= 20; iter_prec++) {
                        double CND = (T - t_00) / (t_0 - t_00);
                        if (T < t_00) CND = 0.;
                        double DEP = (t_0 - T) / (t_0 - t_00);
                        if (T > t_0) DEP = 0.;
                        double d_q_v = q_v_hyp - q_v_b;
                        double d_q_c = -d_q_v * CND;
                        double d_q_i = -d_q_v * DEP;
                        double d_t = (lv * d_q_c + ls * d_q_i) / cp_l;
                        T = T + d_t;
                        q_v_b = F_(C) + d_q_v;
                        q_c_b = F_(CLOUD) + d_q_c;
                        q_i_b = F_(ICE) + d_q_i;
                        if (q_v_b < 0.) q_v_b = 0.;
                        if (q_c_b < 0.) q_c_b = 0.;
                        if (q_i_b < 0.) q_i_b = 0.;
                        p_SL = .01 * (r_air * R_Air * A3(c, ATOMDEV_T, 0, j, k) * t_0);
                        hight = (double)i * (L_atm / (double)(im - 1));
                        if (i != 0) {
                            if (T > gam * hight * 1.e-2) p_h = pow(((T - gam * hight * 1.e-2) / (T)), exp_pressure) * p_SL;
                            else p_h = p_SL; /* the reference logs a warning here */
                        } else p_h = p_SL;
                        E_Rain = hp * exp_func(T, 17.2694, 35.86);
                        E_Ice = hp * exp_func(T, 21.8746, 7.66);
                        q_Rain = ep * E_Rain / (p_h - E_Rain);
                        q_Ice = ep * E_Ice / (p_h - E_Ice);
                        if ((q_c_b > 0.) && (q_i_b > 0.)) q_v_hyp = (q_c_b * q_Rain + q_i_b * q_Ice) / (q_c_b + q_i_b);
                        if ((q_c_b >= 0.) && (q_i_b == 0.)) q_v_hyp = q_Rain;
                        if ((q_c_b == 0.) && (q_i_b > 0.)) q_v_hyp = q_Ice;
                        F_(S_C_C) = .5 * (F_(CN) - F_(C)) / dt_dim;
                        if (is_land(c, i, j, k)) F_(S_C_C) = 0.;
                        q_T = q_v_b + q_c_b + q_i_b;
                        if (iter_prec >= 3 && (q_v_hyp) > DBL_EPSILON && fabs(q_v_b / q_v_hyp - 1.) <= 1.e-5) break;
                        q_v_b = .5 * (q_v_hyp + q_v_b);
                    }
                    F_(CN) = F_(C) = q_v_b;
                    F_(CLOUDN) = F_(CLOUD) = q_c_b;
                    F_(ICEN) = F_(ICE) = q_i_b;
                    if (t_Celsius < t_Celsius_2) F_(CLOUDN) = F_(CLOUD) = 0.;
                    if (t_Celsius > 0.) F_(ICEN) = F_(ICE) = 0.;
                    F_(T) = T / t_0;
                }
                (void)q_T;
            }
    static const int CL[3] = { ATOMDEV_C, ATOMDEV_CLOUD, ATOMDEV_ICE };
    for (int n = 0; n < 3; n++) {
        double* x = c->f[CL[n]];
        for (size_t q = 0; q < (size_t)im * jm * km; q++) if (x[q] < 0.) x[q] = 0.;
    }
    *q_v_hyp_state = q_v_hyp;
#undef F_
}

/* BC_Thermo::BC_Radiation_multi_layer, atmosphere/BC_Thermo.cpp:166-424, pinned bit for bit by the compiled class
 * (tests/test_oracle_vs_reference.py, tests/golden/bc_thermo.npz): a statement-by-statement restatement, including the transmission
 * matrix CC whose recurrence starts from an element that is never set (so all of it stays 0, :341-344) and the
 * sums CCC / DDD that are not reset inside the level loop (:312, :357-362). */
static double parabola(double x) { return x * x - 2 * x; } /* lib/Utils.h:41-43 */
void oracle_atm_radiation_multi_layer(oracle_ctx* c, const atomdev_radiation* R, const int* i_trop_of_row) {
    const atomdev_params* P = &c->p;
    const int im = c->im, jm = c->jm, km = c->km;
    const double t_0 = P->t_0, ep = P->ep, sigma = R->sigma, t_tropopause = P->t_tropopause;
    const double rad_eff = R->rad_pole - R->rad_equator, albedo_co2_eff = R->albedo_pole - R->albedo_equator;
    double j_max_half = (jm - 1) / 2;
#define T_(i) A3(c, ATOMDEV_T, i, j, k)
#define EPS_(i) A3(c, ATOMDEV_EPSILON_3D, i, j, k)
#define RAD_(i) A3(c, ATOMDEV_RADIATION_3D, i, j, k)
    for (int j = 0; j < jm; j++)
        for (int k = 0; k < km; k++)
            for (int i = 0; i < im - 1; i++)
                if ((i == 0 && !is_land(c, i, j, k)) || (is_land(c, i, j, k) && !is_land(c, i + 1, j, k)))
                    A2(c, ATOMDEV_ALBEDO, j, k) = albedo_co2_eff * parabola(j / j_max_half) + R->albedo_pole;
    double epsilon_eff_max;
    const double epsilon_eff_2D = R->epsilon_pole - R->epsilon_equator;
    for (int j = 0; j < jm; j++) {
        const int i_trop = i_trop_of_row[j];
        epsilon_eff_max = epsilon_eff_2D * parabola(j / j_max_half) + R->epsilon_pole;
        for (int k = 0; k < km; k++) {
            int i_mount = 0;
            for (int i = im - 2; i >= 0; i--) if (is_land(c, i, j, k)) { i_mount = i; break; }
            A2(c, ATOMDEV_RADIATION_SURFACE, j, k) = rad_eff * parabola(j / j_max_half) + R->rad_pole;
            for (int i = 0; i <= i_trop; i++) {
                if (A3(c, ATOMDEV_C, i, j, k) < 0.) A3(c, ATOMDEV_C, i, j, k) = 0.;
                if (A3(c, ATOMDEV_CLOUD, i, j, k) < 0.) A3(c, ATOMDEV_CLOUD, i, j, k) = 0.;
                if (A3(c, ATOMDEV_ICE, i, j, k) < 0.) A3(c, ATOMDEV_ICE, i, j, k) = 0.;
                double e = (A3(c, ATOMDEV_C, i, j, k) + A3(c, ATOMDEV_CLOUD, i, j, k) + A3(c, ATOMDEV_ICE, i, j, k))
                           * A3(c, ATOMDEV_P_STAT, i, j, k) / ep;
                double epsilon_eff = epsilon_eff_max - (R->epsilon_tropopause - epsilon_eff_max) * parabola((double)i / (im - 1));
                if (i >= i_mount) {
                    EPS_(i) = R->co2_coeff * epsilon_eff + .0416 * sqrt(e);
                    RAD_(i) = (1. - EPS_(i)) * sigma * pow(T_(i) * t_0, 4.);
                }
                if (EPS_(i) > 1.) EPS_(i) = 1.;
            }
            A2(c, ATOMDEV_EPSILON, j, k) = EPS_(i_trop);
            for (int i = i_mount - 1; i >= 0; i--) { EPS_(i) = EPS_(i_mount); RAD_(i) = RAD_(i_mount); }
            for (int i = i_trop; i < im; i++) {
                EPS_(i) = EPS_(i_trop);
                T_(i) = t_tropopause;
                RAD_(i) = (1. - EPS_(i)) * sigma * pow(T_(i) * t_0, 4.);
            }
        }
    }
    double* alfa = (double*)malloc(sizeof(double) * im); double* beta = (double*)malloc(sizeof(double) * im);
    double* AA = (double*)malloc(sizeof(double) * im); double* CC = (double*)malloc(sizeof(double) * im * im);
#define CC_(a, b) CC[(size_t)(a) * im + (b)]
    for (int iter_rad = 1; iter_rad <= 4; iter_rad++)
        for (int j = 0; j < jm; j++) {
            const int i_trop = i_trop_of_row[j];
            for (int k = 0; k < km; k++) {
                int i_mount = 0;
                for (int i = im - 2; i >= 0; i--) if (is_land(c, i, j, k)) { i_mount = i; break; }
                for (int i = 0; i < im; i++) { alfa[i] = beta[i] = AA[i] = 0.; }
                for (int i = 0; i < im * im; i++) CC[i] = 0.;
                double CCC = 0, DDD = 0, aa, bb, cc, dd;
                RAD_(i_trop) = (1. - EPS_(i_trop)) * sigma * pow(T_(i_trop) * t_0, 4.);
                double radiation_back = EPS_(i_mount + 1) * sigma * pow(T_(i_mount + 1) * t_0, 4.);
                double atmospheric_window = .1007 * A2(c, ATOMDEV_RADIATION_SURFACE, j, k);
                double rad_surf_diff = radiation_back + A2(c, ATOMDEV_RADIATION_SURFACE, j, k) - atmospheric_window;
                double fac_rad = (double)i_mount * .07 + 1.;
                rad_surf_diff = fac_rad * rad_surf_diff;
                AA[i_mount] = rad_surf_diff / RAD_(i_trop);
                CC_(i_mount, i_mount) = 0.;
                RAD_(i_mount) = (1. - EPS_(i_mount)) * sigma * pow(T_(i_mount) * t_0, 4.) / RAD_(i_trop);
                for (int i = i_mount + 1; i <= i_trop; i++) {
                    AA[i] = AA[i - 1] * (1. - EPS_(i));
                    double tmp = sigma * pow(T_(i) * t_0, 4.) / RAD_(i_trop);
                    CC_(i, i) = EPS_(i) * tmp;
                    RAD_(i) = (1. - EPS_(i)) * tmp;
                    for (int l = i_mount + 1; l <= i_trop; l++) CC_(i, l) = CC_(i, l - 1) * (1. - EPS_(l));
                }
                for (int i = i_mount; i < i_trop; i++) {
                    if (i == i_mount) {
                        bb = -RAD_(i); cc = RAD_(i + 1); dd = -AA[i];
                        alfa[i] = cc / bb; beta[i] = dd / bb;
                    } else {
                        for (int l = i_mount + 1; l <= i - 1; l++) CCC = CCC + CC_(l, i);
                        for (int l = i_mount + 1; l <= i - 2; l++) DDD = DDD + CC_(l, i - 1);
                        aa = RAD_(i - 1); bb = -2. * RAD_(i); cc = RAD_(i + 1);
                        dd = -AA[i - 1] + AA[i] + CCC - DDD;
                        alfa[i] = cc / (bb - aa * alfa[i - 1]);
                        beta[i] = (dd - aa * beta[i - 1]) / (bb - aa * alfa[i - 1]);
                    }
                }
                T_(i_trop) = t_tropopause;
                RAD_(i_trop) = (1. - EPS_(i_trop)) * sigma * pow(T_(i_trop) * t_0, 4.);
                for (int i = i_trop - 1; i >= i_mount; i--) {
                    RAD_(i) = -alfa[i] * RAD_(i + 1) + beta[i];
                    T_(i) = .5 * (T_(i) + pow(RAD_(i) / sigma, (1. / 4.)) / t_0);
                }
                for (int i = i_trop; i < im; i++) {
                    T_(i) = t_tropopause;
                    RAD_(i) = (1. - EPS_(i_trop)) * sigma * pow(T_(i) * t_0, 4.);
                }
            }
        }
    free(alfa); free(beta); free(AA); free(CC);
#undef CC_
#undef T_
#undef EPS_
#undef RAD_
}

/* MinMax_Atm::searchMinMax_3D / _2D (atmosphere/MinMax_Atm.cpp:74-196), MinMax_Hyd::searchMinMax_3D / _2D
 * (hydrosphere/MinMax_Hyd.cpp:51-260): the scans, without unit conversion and printout.  The model of the
 * context selects the class.  out = {max, min}; loc = {imax, jmax, kmax, imin, jmin, kmin}. */
void oracle_minmax3d(oracle_ctx* c, int field, double out[2], int loc[6]) {
    const int im = c->im, jm = c->jm, km = c->km;
    double maxValue, minValue;
    for (int n = 0; n < 6; n++) loc[n] = 0;
    if (c->model == ATOMDEV_ATM) {
        minValue = maxValue = A3(c, field, 0, 0, 0);
        for (int j = 0; j < jm; j++) for (int k = 0; k < km; k++) for (int i = 0; i < im; i++) {
            double v = A3(c, field, i, j, k);
            if (v > maxValue) { maxValue = v; loc[0] = i; loc[1] = j; loc[2] = k; }
            else if (v < minValue) { minValue = v; loc[3] = i; loc[4] = j; loc[5] = k; }
        }
    } else {
        maxValue = 0.;
        for (int j = 0; j < jm; j++) for (int k = 0; k < km; k++) for (int i = 0; i < im; i++) {
            double v = A3(c, field, i, j, k);
            if (v > maxValue) { maxValue = v; loc[0] = i; loc[1] = j; loc[2] = k; }
        }
        minValue = maxValue;
        for (int j = 0; j < jm; j++) for (int k = 0; k < km; k++) for (int i = 0; i < im; i++) {
            double v = A3(c, field, i, j, k);
            if (v < minValue) { minValue = v; loc[3] = i; loc[4] = j; loc[5] = k; }
        }
    }
    out[0] = maxValue; out[1] = minValue;
}
void oracle_minmax2d(oracle_ctx* c, int field, double out[2], int loc[6]) {
    const int jm = c->jm, km = c->km;
    double maxValue, minValue;
    for (int n = 0; n < 6; n++) loc[n] = 0;
    if (c->model == ATOMDEV_ATM) {
        minValue = maxValue = A2(c, field, 0, 0);
        for (int j = 1; j < jm - 1; j++) for (int k = 1; k < km - 1; k++) {
            double v = A2(c, field, j, k);
            if (v > maxValue) { maxValue = v; loc[1] = j; loc[2] = k; }
            else if (v < minValue) { minValue = v; loc[4] = j; loc[5] = k; }
        }
    } else {
        maxValue = 0.;
        for (int j = 1; j < jm - 1; j++) for (int k = 1; k < km - 1; k++) {
            double v = A2(c, field, j, k);
            if (v > maxValue) { maxValue = v; loc[1] = j; loc[2] = k; }
        }
        minValue = maxValue;
        for (int j = 1; j < jm - 1; j++) for (int k = 1; k < km - 1; k++) {
            double v = A2(c, field, j, k);
            if ((v < minValue) && (v != 0.)) { minValue = v; loc[4] = j; loc[5] = k; }
        }
        if (minValue < 0.) minValue = 0.;
    }
    out[0] = maxValue; out[1] = minValue;
}

/* BC_Bathymetry_Atmosphere::vegetationDistribution, atmosphere/BC_Bath_Atm.cpp:129-142 */
void oracle_atm_vegetation_distribution(oracle_ctx* c, double max_Precipitation) {
    for (int j = 0; j < c->jm; j++)
        for (int k = 0; k < c->km; k++) {
            if (max_Precipitation > 0 && is_land(c, 0, j, k) && !(A3(c, ATOMDEV_T, 0, j, k) < 1.))
                A2(c, ATOMDEV_VEGETATION, j, k) = A2(c, ATOMDEV_PRECIPITATION, j, k) / max_Precipitation;
            else
                A2(c, ATOMDEV_VEGETATION, j, k) = 0.;
        }
}

/* PostProcess_Atmosphere::Atmosphere_v_w_Transfer (atmosphere/PostProcess_Atm.cpp:98-118) followed by
 * PostProcess_Hydrosphere::Atmosphere_TransferFile_read (hydrosphere/PostProcess_Hyd.cpp:642-668): every value
 * goes through fixed notation with 4 decimals, as the file between the two models would carry it. */
static double through_text(double x) {
    char buf[400];
    snprintf(buf, sizeof buf, "%.4f", x);
    return strtod(buf, NULL);
}
void oracle_transfer_atm_to_hyd(oracle_ctx* a, oracle_ctx* h) {
    const int jm = a->jm, km = a->km, top = h->im - 1;
    for (int j = 0; j < jm; j++)
        for (int k = 0; k < km; k++) {
            A3(h, ATOMDEV_V, top, j, k) = through_text(A3(a, ATOMDEV_V, 0, j, k) * a->p.u_0);
            A3(h, ATOMDEV_W, top, j, k) = through_text(A3(a, ATOMDEV_W, 0, j, k) * a->p.u_0);
            A3(h, ATOMDEV_T, top, j, k) = through_text(A3(a, ATOMDEV_T, 0, j, k));
            A3(h, ATOMDEV_P_DYN, top, j, k) = 0.;
            A2(h, ATOMDEV_EVAPORATION_DALTON, j, k) = through_text(A2(a, ATOMDEV_EVAPORATION_DALTON, j, k));
            A2(h, ATOMDEV_PRECIPITATION, j, k) = through_text(A2(a, ATOMDEV_PRECIPITATION, j, k));
        }
}

/* Results_Hyd::run_data, hydrosphere/Results_Hyd.cpp:102-320, without the printout.  The reference integrates a
 * buffer that it fills at indices i_Ekman..im-2 over indices 0..i_diff (lib/Utils.h:58-65); the buffer is restated
 * as it is used.  The phi term of BottomWater is (x - x) / (2 rmsinthe dphi) with a dphi the class never sets
 * (Results_Hyd.h:26): evaluated here with the grid's dphi, which gives the same +0 as any finite positive value. */
static double simpson_as_shipped(int n, double dstep, const double* value) {
    double sum_even = 0, sum_odd = 0;
    if (n % 2 == 0) {
        for (int i = 1; i < n; i += 2) sum_odd += 4 * value[i];
        for (int i = 2; i < n; i += 2) sum_even += 2 * value[i];
    }
    return dstep / 3 * (value[0] + sum_odd + sum_even + value[n]);
}
void oracle_hyd_run_data(oracle_ctx* c) {
    const atomdev_params* P = &c->p;
    const int im = c->im, jm = c->jm, km = c->km;
    const double c43 = 4. / 3., c13 = 1. / 3., dr = P->dr, dthe = P->dthe, dphi = P->dphi;
    static const int G0[7] = { ATOMDEV_UPWELLING, ATOMDEV_DOWNWELLING, ATOMDEV_BOTTOM_WATER, ATOMDEV_SALT_FINGER_2D,
                               ATOMDEV_SALT_DIFFUSION_2D, ATOMDEV_BUOYANCY_FORCE_2D, ATOMDEV_SALT_TOTAL };
    double* aux_v = (double*)calloc((size_t)jm * km, sizeof(double));
    double* aux_w = (double*)calloc((size_t)jm * km, sizeof(double));
    double* aux_grad_v = (double*)calloc((size_t)im + 1, sizeof(double));
    double* aux_grad_w = (double*)calloc((size_t)im + 1, sizeof(double));
    for (int n = 0; n < 7; n++) memset(c->g[G0[n]], 0, sizeof(double) * (size_t)jm * km);
    const int i_half = (im - 1) / 2;
    const double i_Ekman_layer = 500.;
    const double coeff = i_Ekman_layer / P->L_hyd;
    const int i_Ekman = (im - 1) * (1. - coeff);
    const int j_half = (jm - 1) / 2, i_max = im - 1, i_diff = i_max - i_Ekman;
#define AV(j, k) aux_v[(size_t)(j) * km + (k)]
#define AW(j, k) aux_w[(size_t)(j) * km + (k)]
    for (int k = 0; k < km; k++)
        for (int j = 0; j < jm; j++) {
            for (int i = i_Ekman; i < im - 1; i++) {
                if (is_fluid(c, i, j, k)) { aux_grad_v[i] = A3(c, ATOMDEV_V, i, j, k); aux_grad_w[i] = A3(c, ATOMDEV_W, i, j, k); }
                else { AV(j, k) = 0.; AW(j, k) = 0.; aux_grad_v[i] = 0.; aux_grad_w[i] = 0.; }
            }
            if (i_diff % 2 == 0) {
                AV(j, k) = simpson_as_shipped(i_diff, dr, aux_grad_v);
                AW(j, k) = simpson_as_shipped(i_diff, dr, aux_grad_w);
            }
        }
    for (int k = 1; k < km - 1; k++) {
        for (int j = 1; j < jm - 1; j++) {
            if (!(j < j_half - 1 || j >= j_half + 2)) continue;
            double rmsinthe = c->rad[im - 1] * sin(c->the[j]);
            double b = (AV(j + 1, k) - AV(j - 1, k)) / (2. * c->rad[im - 1] * dthe) + (AW(j, k + 1) - AW(j, k + 1)) / (2. * rmsinthe * dphi);
            A2(c, ATOMDEV_BOTTOM_WATER, j, k) = b * P->u_0;
        }
        for (int j = 1; j < jm - 1; j++) {
            double b = A2(c, ATOMDEV_BOTTOM_WATER, j, k);
            A2(c, ATOMDEV_UPWELLING, j, k) = b >= 0. ? b : 0.;
            A2(c, ATOMDEV_DOWNWELLING, j, k) = b < 0. ? b : 0.;
        }
    }
    for (int k = 0; k < km; k++)
        for (int j = 0; j < jm; j++) A2(c, ATOMDEV_DOWNWELLING, j, k) = fabs(A2(c, ATOMDEV_DOWNWELLING, j, k));
    static const int UD[2] = { ATOMDEV_UPWELLING, ATOMDEV_DOWNWELLING };
    for (int k = 0; k < km; k++)
        for (int n = 0; n < 2; n++) {
            A2(c, UD[n], 0, k) = c43 * A2(c, UD[n], 1, k) - c13 * A2(c, UD[n], 2, k);
            A2(c, UD[n], jm - 1, k) = c43 * A2(c, UD[n], jm - 2, k) - c13 * A2(c, UD[n], jm - 3, k);
        }
    for (int j = 0; j < jm; j++)
        for (int n = 0; n < 2; n++) {
            A2(c, UD[n], j, 0) = c43 * A2(c, UD[n], j, 1) - c13 * A2(c, UD[n], j, 2);
            A2(c, UD[n], j, km - 1) = c43 * A2(c, UD[n], j, km - 2) - c13 * A2(c, UD[n], j, km - 3);
            A2(c, UD[n], j, 0) = A2(c, UD[n], j, km - 1) = (A2(c, UD[n], j, 0) + A2(c, UD[n], j, km - 1)) / 2.;
        }
    for (int k = 0; k < km; k++)
        for (int j = 0; j < jm; j++)
            for (int i = i_half; i < im; i++)
                if (is_fluid(c, i, j, k)) {
                    A2(c, ATOMDEV_SALT_FINGER_2D, j, k) += A3(c, ATOMDEV_SALT_FINGER, i, j, k);
                    A2(c, ATOMDEV_SALT_DIFFUSION_2D, j, k) += A3(c, ATOMDEV_SALT_DIFFUSION, i, j, k);
                    A2(c, ATOMDEV_BUOYANCY_FORCE_2D, j, k) += A3(c, ATOMDEV_BUOYANCY, i, j, k);
                    A2(c, ATOMDEV_SALT_TOTAL, j, k) += A3(c, ATOMDEV_C, i, j, k) * P->c_0;
                }
    static const int F4[4] = { ATOMDEV_BUOYANCY, ATOMDEV_SALT_FINGER, ATOMDEV_SALT_DIFFUSION, ATOMDEV_SALT_BALANCE };
    for (int n = 0; n < 4; n++) {
        const int F = F4[n];
        for (int k = 0; k < km; k++)
            for (int j = 0; j < jm; j++) {
                A3(c, F, 0, j, k) = c43 * A3(c, F, 1, j, k) - c13 * A3(c, F, 2, j, k);
                A3(c, F, im - 1, j, k) = c43 * A3(c, F, im - 2, j, k) - c13 * A3(c, F, im - 3, j, k);
            }
        for (int k = 0; k < km; k++)
            for (int i = 0; i < im; i++) {
                A3(c, F, i, 0, k) = c43 * A3(c, F, i, 1, k) - c13 * A3(c, F, i, 2, k);
                A3(c, F, i, jm - 1, k) = c43 * A3(c, F, i, jm - 2, k) - c13 * A3(c, F, i, jm - 3, k);
            }
        for (int i = 0; i < im; i++)
            for (int j = 0; j < jm; j++) {
                A3(c, F, i, j, 0) = c43 * A3(c, F, i, j, 1) - c13 * A3(c, F, i, j, 2);
                A3(c, F, i, j, km - 1) = c43 * A3(c, F, i, j, km - 2) - c13 * A3(c, F, i, j, km - 3);
                A3(c, F, i, j, 0) = A3(c, F, i, j, km - 1) = (A3(c, F, i, j, 0) + A3(c, F, i, j, km - 1)) / 2.;
            }
    }
#undef AV
#undef AW
    free(aux_v); free(aux_w); free(aux_grad_v); free(aux_grad_w);
}

/* PostProcess_Atmosphere::Atmosphere_PlotData (atmosphere/PostProcess_Atm.cpp:624-652) and
 * PostProcess_Hydrosphere::Hydrosphere_PlotData (hydrosphere/PostProcess_Hyd.cpp:676-705): "%.4f" is what an
 * ofstream with precision(4) and ios::fixed prints. */
int oracle_write_plot_data(oracle_ctx* c, const char* path) {
    FILE* f = fopen(path, "w");
    if (!f) return 1;
    const int atm = c->model == ATOMDEV_ATM, L = atm ? 0 : c->im - 1;
    const double u_0 = c->p.u_0, t_0 = c->p.t_0;
    if (atm) fprintf(f, "lons(deg), lats(deg), topography, v-velocity(m/s), w-velocity(m/s), velocity-mag(m/s), temperature(Celsius), water_vapour(g/kg), precipitation(mm), precipitable water(mm)\n");
    else fprintf(f, "lons(deg), lats(deg), topography, v-velocity(m/s), w-velocity(m/s), velocity-mag(m/s), temperature(Celsius), salinity(psu), bottom_water(m/s), upwelling(m/s), downwelling(m/s)\n");
    for (int k = 0; k < c->km; k++)
        for (int j = 0; j < c->jm; j++) {
            const double v = A3(c, ATOMDEV_V, L, j, k) * u_0, w = A3(c, ATOMDEV_W, L, j, k) * u_0;
            const double vel_mag = sqrt(pow(v, 2) + pow(w, 2));
            if (atm)
                fprintf(f, "%d %d %.4f %.4f %.4f %.4f %.4f %.4f %.4f %.4f \n", k, 90 - j, A3(c, ATOMDEV_H, L, j, k), v, w, vel_mag,
                        A3(c, ATOMDEV_T, L, j, k) * t_0 - t_0, A3(c, ATOMDEV_C, L, j, k) * 1000., A2(c, ATOMDEV_PRECIPITATION, j, k),
                        A2(c, ATOMDEV_PRECIPITABLE_WATER, j, k));
            else
                fprintf(f, "%d %d %.4f %.4f %.4f %.4f %.4f %.4f %.4f %.4f   %.4f \n", k, 90 - j, A3(c, ATOMDEV_H, L, j, k), v, w, vel_mag,
                        A3(c, ATOMDEV_T, L, j, k) * 273.15 - 273.15, A3(c, ATOMDEV_C, L, j, k) * 35., A2(c, ATOMDEV_BOTTOM_WATER, j, k),
                        A2(c, ATOMDEV_UPWELLING, j, k), A2(c, ATOMDEV_DOWNWELLING, j, k));
        }
    return fclose(f) != 0;
}

/* The ASCII VTK sections: PostProcess_Atmosphere::paraview_vtk_radial / _zonal / _longal
 * (atmosphere/PostProcess_Atm.cpp:240-389, 391-520, 522-621; dump_* helpers :50-96) and
 * PostProcess_Hydrosphere::paraview_vtk_longal / _radial / _zonal (hydrosphere/PostProcess_Hyd.cpp:373-449, 456-552,
 * 559-636), one function per writer, statement order as there.  What the writers store into aux_u / aux_v / aux_w
 * and into the Celsius copy of temperature_NASA stays in the bound arrays, as in the reference. */
enum { SEC_RADIAL = 0, SEC_LONGAL = 1, SEC_ZONAL = 2 };

static double sec_value(const oracle_ctx* c, int F, int sec, int fixed, int a, int b) {
    return sec == SEC_RADIAL ? A3(c, F, fixed, a, b) : (sec == SEC_LONGAL ? A3(c, F, a, fixed, b) : A3(c, F, a, b, fixed));
}
static void sec_dims(const oracle_ctx* c, int sec, int* na, int* nb_) {
    *na = sec == SEC_RADIAL ? c->jm : c->im;
    *nb_ = sec == SEC_ZONAL ? c->jm : c->km;
}
/* dump_radial / dump_longal / dump_zonal */
static void vtk_dump(FILE* f, const oracle_ctx* c, const char* desc, int F, double multiplier, int sec, int fixed) {
    int na, nb_;
    sec_dims(c, sec, &na, &nb_);
    fprintf(f, "SCALARS %s float 1\nLOOKUP_TABLE default\n", desc);
    for (int a = 0; a < na; a++)
        for (int b = 0; b < nb_; b++) fprintf(f, "%.4f\n", sec_value(c, F, sec, fixed, a, b) * multiplier);
}
/* dump_radial_2d */
static void vtk_dump_2d(FILE* f, const oracle_ctx* c, const char* desc, int G, double multiplier) {
    fprintf(f, "SCALARS %s float 1\nLOOKUP_TABLE default\n", desc);
    for (int j = 0; j < c->jm; j++)
        for (int k = 0; k < c->km; k++) fprintf(f, "%.4f\n", A2(c, G, j, k) * multiplier);
}
static void vtk_head(FILE* f, const oracle_ctx* c, const char* title, int sec, double d_inner) {
    int na, nb_;
    sec_dims(c, sec, &na, &nb_);
    fprintf(f, "# vtk DataFile Version 3.0\n%s\nASCII\nDATASET STRUCTURED_GRID\nDIMENSIONS %d %d 1\nPOINTS %d float\n", title, nb_, na, na * nb_);
    double x = 0., y = 0., z = 0.;
    const double dx = .1;
    for (int a = 0; a < na; a++) {
        for (int b = 0; b < nb_; b++) {
            if (sec == SEC_LONGAL) { if (b == 0) z = 0.; else z = z + d_inner; }
            else { if (b == 0) y = 0.; else y = y + d_inner; }
            fprintf(f, "%.4f %.4f %.4f\n", x, y, z);
        }
        if (sec == SEC_LONGAL) z = 0.; else y = 0.;
        x = x + dx;
    }
    fprintf(f, "POINT_DATA %d\n", na * nb_);
}
static void vtk_temperature_celsius(FILE* f, const oracle_ctx* c, int sec, int fixed) {
    int na, nb_;
    sec_dims(c, sec, &na, &nb_);
    fprintf(f, "SCALARS Temperature float 1\nLOOKUP_TABLE default\n");
    for (int a = 0; a < na; a++)
        for (int b = 0; b < nb_; b++) fprintf(f, "%.4f\n", sec_value(c, ATOMDEV_T, sec, fixed, a, b) * c->p.t_0 - c->p.t_0);
}
static void vtk_vectors(FILE* f, const oracle_ctx* c, const char* head, int sec, int fixed) {
    int na, nb_;
    sec_dims(c, sec, &na, &nb_);
    fprintf(f, "%s\n", head);
    for (int a = 0; a < na; a++)
        for (int b = 0; b < nb_; b++) {
            const double u = sec_value(c, ATOMDEV_U, sec, fixed, a, b), v = sec_value(c, ATOMDEV_V, sec, fixed, a, b),
                         w = sec_value(c, ATOMDEV_W, sec, fixed, a, b);
            if (sec == SEC_RADIAL) fprintf(f, "%.4f %.4f %.4f\n", v, w, 0.);
            else if (sec == SEC_LONGAL) fprintf(f, "%.4f %.4f %.4f\n", u, 0., w);
            else fprintf(f, "%.4f %.4f %.4f\n", u, v, 0.);
        }
}

static void vtk_atm_radial(FILE* f, oracle_ctx* c, int Ma, int i_radial) {
    const atomdev_params* P = &c->p;
    const int im = c->im, jm = c->jm, km = c->km, S = SEC_RADIAL;
    vtk_head(f, c, "Radial_Data_Atmosphere_Circulation", S, .1);
    /* the POINT_DATA line follows the Ma block in the reference; the block writes nothing to the file */
    if (Ma == 0)
        for (int k = 0; k < km; k++)
            for (int j = 0; j < jm; j++) {
                A2(c, ATOMDEV_TEMP_NASA_CELSIUS, j, k) = A2(c, ATOMDEV_TEMPERATURE_NASA, j, k) * P->t_0 - P->t_0;
                for (int i = im - 2; i >= 0; i--) {
                    if (is_fluid(c, i + 1, j, k) && is_land(c, i, j, k))
                        A3(c, ATOMDEV_AUX_V, i_radial, j, k) = (A3(c, ATOMDEV_T, 0, j, k) - A2(c, ATOMDEV_TEMPERATURE_NASA, j, k)) * P->t_0;
                    if (is_fluid(c, 0, j, k))
                        A3(c, ATOMDEV_AUX_V, i_radial, j, k) = (A3(c, ATOMDEV_T, 0, j, k) - A2(c, ATOMDEV_TEMPERATURE_NASA, j, k)) * P->t_0;
                    A3(c, ATOMDEV_AUX_W, i_radial, j, k) = A2(c, ATOMDEV_EVAPORATION_DALTON, j, k) - A2(c, ATOMDEV_PRECIPITATION, j, k);
                }
            }
    vtk_dump(f, c, "u-Component", ATOMDEV_U, 1., S, i_radial);
    vtk_dump(f, c, "v-Component", ATOMDEV_V, 1., S, i_radial);
    vtk_dump(f, c, "w-Component", ATOMDEV_W, 1., S, i_radial);
    vtk_temperature_celsius(f, c, S, i_radial);
    vtk_dump_2d(f, c, "Temp_NASA", ATOMDEV_TEMP_NASA_CELSIUS, 1.);
    vtk_dump(f, c, "Temp_NASA_diff", ATOMDEV_AUX_V, 1., S, i_radial);
    vtk_dump(f, c, "Topography", ATOMDEV_H, 1., S, i_radial);
    vtk_dump_2d(f, c, "Topography_m", ATOMDEV_TOPOGRAPHY, 1.);
    vtk_dump(f, c, "WaterVapour", ATOMDEV_C, 1000., S, i_radial);
    vtk_dump(f, c, "CloudWater", ATOMDEV_CLOUD, 1000., S, i_radial);
    vtk_dump(f, c, "CloudIce", ATOMDEV_ICE, 1000., S, i_radial);
    vtk_dump(f, c, "PrecipitationRain", ATOMDEV_P_RAIN, 86400., S, i_radial);
    vtk_dump(f, c, "PrecipitationSnow", ATOMDEV_P_SNOW, 86400., S, i_radial);
    vtk_dump_2d(f, c, "Precipitation", ATOMDEV_PRECIPITATION, 1.);
    vtk_dump_2d(f, c, "PrecipitableWater_2D", ATOMDEV_PRECIPITABLE_WATER, 1.);
    vtk_dump_2d(f, c, "Precipitation_NASA", ATOMDEV_PRECIPITATION_NASA, 1.);
    vtk_dump(f, c, "PressureDynamic", ATOMDEV_P_DYN, P->u_0 * P->u_0 * P->r_air * .01, S, i_radial);
    vtk_dump(f, c, "PressureStatic", ATOMDEV_P_STAT, 1., S, i_radial);
    vtk_dump(f, c, "Epsilon_3D", ATOMDEV_EPSILON_3D, 1., S, i_radial);
    vtk_dump_2d(f, c, "albedo_2D", ATOMDEV_ALBEDO, 1.);
    vtk_dump_2d(f, c, "epsilon_2D", ATOMDEV_EPSILON, 1.);
    vtk_dump_2d(f, c, "Q_radiation_2D", ATOMDEV_Q_RADIATION, 1.);
    vtk_dump_2d(f, c, "Q_bottom_2D", ATOMDEV_Q_BOTTOM, 1.);
    vtk_dump_2d(f, c, "Q_latent_2D", ATOMDEV_Q_LATENT_2D, 1.);
    vtk_dump_2d(f, c, "Q_sensible_2D", ATOMDEV_Q_SENSIBLE_2D, 1.);
    vtk_dump(f, c, "BuoyancyForce", ATOMDEV_BUOYANCY, 1., S, i_radial);
    vtk_dump(f, c, "Q_Radiation", ATOMDEV_RADIATION_3D, 1., S, i_radial);
    vtk_dump(f, c, "Q_Latent", ATOMDEV_Q_LATENT, 1., S, i_radial);
    vtk_dump(f, c, "Q_Sensible", ATOMDEV_Q_SENSIBLE, 1., S, i_radial);
    vtk_dump_2d(f, c, "Evaporation_Penman", ATOMDEV_EVAPORATION_PENMAN, 1.);
    vtk_dump_2d(f, c, "Evaporation_Dalton", ATOMDEV_EVAPORATION_DALTON, 1.);
    vtk_dump_2d(f, c, "Heat_Evaporation", ATOMDEV_Q_EVAPORATION, 1.);
    vtk_dump(f, c, "Evap-Precip", ATOMDEV_AUX_W, 1., S, i_radial);
    vtk_dump_2d(f, c, "Vegetation", ATOMDEV_VEGETATION, 1.);
    vtk_dump(f, c, "CO2-Concentration", ATOMDEV_CO2, P->co2_0, S, i_radial);
    vtk_vectors(f, c, "VECTORS v-w-Cell float ", S, i_radial);
}

static void vtk_atm_zonal(FILE* f, oracle_ctx* c, int k_zonal) {
    const atomdev_params* P = &c->p;
    const int im = c->im, jm = c->jm, S = SEC_ZONAL;
    const double t_0 = P->t_0, r_air = P->r_air, R_Air = P->R_Air, g = P->g, L_atm = P->L_atm, hp = P->hp, ep = P->ep;
    vtk_head(f, c, "Zonal_Data_Atmosphere_Circulation", S, .05);
    vtk_dump(f, c, "u-Component", ATOMDEV_U, 1., S, k_zonal);
    vtk_dump(f, c, "v-Component", ATOMDEV_V, 1., S, k_zonal);
    vtk_dump(f, c, "w-Component", ATOMDEV_W, 1., S, k_zonal);
    vtk_temperature_celsius(f, c, S, k_zonal);
    for (int i = 0; i < im; i++)
        for (int j = 0; j < jm; j++) {
            A3(c, ATOMDEV_AUX_W, i, j, k_zonal) = A3(c, ATOMDEV_C, i, j, k_zonal) + A3(c, ATOMDEV_CLOUD, i, j, k_zonal) + A3(c, ATOMDEV_ICE, i, j, k_zonal);
            double t_u = A3(c, ATOMDEV_T, i, j, k_zonal) * t_0;
            double T = t_u;
            double p_SL = .01 * (r_air * R_Air * A3(c, ATOMDEV_T, 0, j, k_zonal) * t_0);
            double p_h;
            if (i != 0) p_h = exp(-g * (double)i * (L_atm / (double)(im - 1)) / (R_Air * t_u)) * p_SL;
            else p_h = p_SL;
            double E_Rain = hp * exp_func(T, 17.2694, 35.86);
            double E_Ice = hp * exp_func(T, 21.8746, 7.66);
            double q_Rain = ep * E_Rain / (p_h - E_Rain);
            double q_Ice = ep * E_Ice / (p_h - E_Ice);
            A3(c, ATOMDEV_AUX_U, i, j, k_zonal) = A3(c, ATOMDEV_C, i, j, k_zonal) / q_Rain;
            if (T <= t_0) A3(c, ATOMDEV_AUX_V, i, j, k_zonal) = A3(c, ATOMDEV_C, i, j, k_zonal) / q_Ice;
            else A3(c, ATOMDEV_AUX_V, i, j, k_zonal) = 0.;
        }
    vtk_dump(f, c, "Topography", ATOMDEV_H, 1., S, k_zonal);
    vtk_dump(f, c, "WaterVapour", ATOMDEV_C, 1000., S, k_zonal);
    vtk_dump(f, c, "CloudWater", ATOMDEV_CLOUD, 1000., S, k_zonal);
    vtk_dump(f, c, "CloudIce", ATOMDEV_ICE, 1000., S, k_zonal);
    vtk_dump(f, c, "Saturation_Water", ATOMDEV_AUX_U, 1., S, k_zonal);
    vtk_dump(f, c, "Saturation_Ice", ATOMDEV_AUX_V, 1., S, k_zonal);
    vtk_dump(f, c, "Total_Water", ATOMDEV_AUX_W, 1000., S, k_zonal);
    vtk_dump(f, c, "PrecipitationRain", ATOMDEV_P_RAIN, 86400., S, k_zonal);
    vtk_dump(f, c, "PrecipitationSnow", ATOMDEV_P_SNOW, 86400., S, k_zonal);
    vtk_dump(f, c, "Source_WaterVapour", ATOMDEV_S_V, 1000., S, k_zonal);
    vtk_dump(f, c, "Source_CloudWater", ATOMDEV_S_C, 1000., S, k_zonal);
    vtk_dump(f, c, "Source_CloudIce", ATOMDEV_S_I, 1000., S, k_zonal);
    vtk_dump(f, c, "Source_Rain", ATOMDEV_S_R, 1000., S, k_zonal);
    vtk_dump(f, c, "Source_Snow", ATOMDEV_S_S, 1000., S, k_zonal);
    vtk_dump(f, c, "Source_CloudWater_CondEvap", ATOMDEV_S_C_C, 1000., S, k_zonal);
    vtk_dump(f, c, "BuoyancyForce", ATOMDEV_BUOYANCY, 1., S, k_zonal);
    vtk_dump(f, c, "Q_Radiation", ATOMDEV_RADIATION_3D, 1., S, k_zonal);
    vtk_dump(f, c, "Epsilon_3D", ATOMDEV_EPSILON_3D, 1., S, k_zonal);
    vtk_dump(f, c, "Q_Latent", ATOMDEV_Q_LATENT, 1., S, k_zonal);
    vtk_dump(f, c, "Q_Sensible", ATOMDEV_Q_SENSIBLE, 1., S, k_zonal);
    vtk_dump(f, c, "PressureDynamic", ATOMDEV_P_DYN, P->u_0 * P->u_0 * r_air * .01, S, k_zonal);
    vtk_dump(f, c, "PressureStatic", ATOMDEV_P_STAT, 1., S, k_zonal);
    vtk_dump(f, c, "CO2-Concentration", ATOMDEV_CO2, P->co2_0, S, k_zonal);
    vtk_vectors(f, c, "VECTORS u-v-Cell float", S, k_zonal);
}

static void vtk_atm_longal(FILE* f, oracle_ctx* c, int j_longal) {
    const atomdev_params* P = &c->p;
    const int S = SEC_LONGAL;
    vtk_head(f, c, "Longitudinal_Data_Atmosphere_Circulation", S, .025);
    vtk_dump(f, c, "u-Component", ATOMDEV_U, 1., S, j_longal);
    vtk_dump(f, c, "v-Component", ATOMDEV_V, 1., S, j_longal);
    vtk_dump(f, c, "w-Component", ATOMDEV_W, 1., S, j_longal);
    vtk_temperature_celsius(f, c, S, j_longal);
    for (int i = 0; i < c->im; i++)
        for (int k = 0; k < c->km; k++)
            A3(c, ATOMDEV_AUX_W, i, j_longal, k) = A3(c, ATOMDEV_C, i, j_longal, k) + A3(c, ATOMDEV_CLOUD, i, j_longal, k) + A3(c, ATOMDEV_ICE, i, j_longal, k);
    vtk_dump(f, c, "Topography", ATOMDEV_H, 1., S, j_longal);
    vtk_dump(f, c, "WaterVapour", ATOMDEV_C, 1000., S, j_longal);
    vtk_dump(f, c, "CloudWater", ATOMDEV_CLOUD, 1000., S, j_longal);
    vtk_dump(f, c, "CloudIce", ATOMDEV_ICE, 1000., S, j_longal);
    vtk_dump(f, c, "Total_Water", ATOMDEV_AUX_W, 1000., S, j_longal);
    vtk_dump(f, c, "PrecipitationRain", ATOMDEV_P_RAIN, 86400., S, j_longal);
    vtk_dump(f, c, "PrecipitationSnow", ATOMDEV_P_SNOW, 86400., S, j_longal);
    vtk_dump(f, c, "PressureDynamic", ATOMDEV_P_DYN, P->u_0 * P->u_0 * P->r_air * .01, S, j_longal);
    vtk_dump(f, c, "PressureStatic", ATOMDEV_P_STAT, 1., S, j_longal);
    vtk_dump(f, c, "BuoyancyForce", ATOMDEV_BUOYANCY, 1., S, j_longal);
    vtk_dump(f, c, "Q_Latent", ATOMDEV_Q_LATENT, 1., S, j_longal);
    vtk_dump(f, c, "Q_Sensible", ATOMDEV_Q_SENSIBLE, 1., S, j_longal);
    vtk_dump(f, c, "Epsilon_3D", ATOMDEV_EPSILON_3D, 1., S, j_longal);
    vtk_dump(f, c, "CO2-Concentration", ATOMDEV_CO2, P->co2_0, S, j_longal);
    vtk_vectors(f, c, "VECTORS u-w-Cell float", S, j_longal);
}

/* the columns the three hydrosphere writers share (PostProcess_Hyd.cpp:425-437, 525-536, 613-625) */
static void vtk_hyd_common(FILE* f, oracle_ctx* c, int S, int fixed, int with_temperature) {
    const atomdev_params* P = &c->p;
    vtk_dump(f, c, "u-Component", ATOMDEV_U, 1000., S, fixed);
    vtk_dump(f, c, "v-Component", ATOMDEV_V, 1., S, fixed);
    vtk_dump(f, c, "w-Component", ATOMDEV_W, 1., S, fixed);
    if (with_temperature) vtk_dump(f, c, "Temperature", ATOMDEV_T, 1., S, fixed);
    vtk_dump(f, c, "PressureDynamic", ATOMDEV_P_DYN, P->u_0 * P->u_0 * P->r_0_water * 1e-3, S, fixed);
    vtk_dump(f, c, "PressureStatic", ATOMDEV_P_STAT, 1., S, fixed);
    vtk_dump(f, c, "Salinity", ATOMDEV_C, 1., S, fixed);
    vtk_dump(f, c, "DensityWater", ATOMDEV_R_WATER, 1., S, fixed);
    vtk_dump(f, c, "DensitySaltWater", ATOMDEV_R_SALT_WATER, 1., S, fixed);
    vtk_dump(f, c, "SaltFinger", ATOMDEV_SALT_FINGER, 1., S, fixed);
    vtk_dump(f, c, "SaltDiffusion", ATOMDEV_SALT_DIFFUSION, 1., S, fixed);
    vtk_dump(f, c, "SaltBalance", ATOMDEV_SALT_BALANCE, 1., S, fixed);
    vtk_dump(f, c, "BuoyancyForce", ATOMDEV_BUOYANCY, 1., S, fixed);
}

static void vtk_hyd_radial(FILE* f, oracle_ctx* c, int i_radial) {
    const int S = SEC_RADIAL;
    vtk_head(f, c, "Radial_Data_Hydrosphere_Circulation", S, .1);
    fprintf(f, "SCALARS Temperature float 1\nLOOKUP_TABLE default\n");
    for (int j = 0; j < c->jm; j++)
        for (int k = 0; k < c->km; k++) {
            fprintf(f, "%.4f\n", A3(c, ATOMDEV_T, i_radial, j, k) * c->p.t_0 - c->p.t_0);
            A3(c, ATOMDEV_AUX_V, i_radial, j, k) = A2(c, ATOMDEV_EVAPORATION_DALTON, j, k) - A2(c, ATOMDEV_PRECIPITATION, j, k);
            if (is_land(c, 0, j, k)) A3(c, ATOMDEV_AUX_V, i_radial, j, k) = 0.;
        }
    vtk_dump(f, c, "Topography", ATOMDEV_H, 1., S, i_radial);
    vtk_dump_2d(f, c, "Bathymetry_m", ATOMDEV_TOPOGRAPHY, 1.);
    vtk_hyd_common(f, c, S, i_radial, 0);
    vtk_dump_2d(f, c, "Upwelling", ATOMDEV_UPWELLING, 1.);
    vtk_dump_2d(f, c, "Downwelling", ATOMDEV_DOWNWELLING, 1.);
    vtk_dump_2d(f, c, "BottomWater", ATOMDEV_BOTTOM_WATER, 1.);
    vtk_dump_2d(f, c, "Evaporation_Dalton", ATOMDEV_EVAPORATION_DALTON, 1.);
    vtk_dump_2d(f, c, "Precipitation", ATOMDEV_PRECIPITATION, 1.);
    vtk_dump(f, c, "Evap-Precip", ATOMDEV_AUX_V, 1., S, i_radial);
    vtk_vectors(f, c, "VECTORS v-w-Cell float", S, i_radial);
}

/* paraview_panorama_vts of both models (PostProcess_Atm.cpp:120-238 with dump_array :33-48; PostProcess_Hyd.cpp:269-366).
 * A string the writers end with "\n" << endl carries two newlines; dump_array ends its lines with one. */
static void vts_dump_array(FILE* f, const oracle_ctx* c, const char* name, int F, double multiplier) {
    fprintf(f, "    <DataArray type=\"Float32\" Name=\"%s\" format=\"ascii\">\n", name);
    for (int k = 0; k < c->km; k++) {
        for (int j = 0; j < c->jm; j++) {
            for (int i = 0; i < c->im; i++) fprintf(f, "%.4f\n", A3(c, F, i, j, k) * multiplier);
            fprintf(f, "\n");
        }
        fprintf(f, "\n");
    }
    fprintf(f, "\n");
    fprintf(f, "    </DataArray>\n");
}

int oracle_write_panorama_vts(oracle_ctx* c, const char* path) {
    FILE* f = fopen(path, "w");
    if (!f) return 1;
    const int atm = c->model == ATOMDEV_ATM, im = c->im, jm = c->jm, km = c->km;
    const atomdev_params* P = &c->p;
    fprintf(f, "<?xml version=\"1.0\"?>\n\n");
    fprintf(f, "<VTKFile type=\"StructuredGrid\" version=\"0.1\" byte_order=\"LittleEndian\">\n\n");
    fprintf(f, " <StructuredGrid WholeExtent=\"%d %d %d %d %d %d\">\n\n", 1, im, 1, jm, 1, km);
    fprintf(f, "  <Piece Extent=\"%d %d %d %d %d %d\">\n\n", 1, im, 1, jm, 1, km);
    if (atm) fprintf(f, "   <PointData Vectors=\"Velocity\" Scalars=\"Topography u-component v-component w-component Temperature CondensationTemp EvaporationTemp Epsilon_3D PressureDynamic PressureStatic WaterVapour CloudWater CloudIce CO2-Concentration Q_Latent Rain RainSuper Ice PrecipitationRain PrecipitationSnow PrecipitationConv Updraft Downdraft\">\n\n");
    else fprintf(f, "   <PointData Vectors=\"Velocity\" Scalars=\"Topography Temperature PressureDynamic PressureStatic Salinity\">\n\n");
    fprintf(f, "    <DataArray type=\"Float32\" NumberOfComponents=\"3\" Name=\"Velocity\" format=\"ascii\">\n\n");
    for (int k = 0; k < km; k++) {
        for (int j = 0; j < jm; j++) {
            for (int i = 0; i < im; i++)
                fprintf(f, "%.4f %.4f %.4f\n", A3(c, ATOMDEV_U, i, j, k), A3(c, ATOMDEV_V, i, j, k), A3(c, ATOMDEV_W, i, j, k));
            fprintf(f, "\n\n");
        }
        fprintf(f, "\n\n");
    }
    fprintf(f, "\n\n");
    fprintf(f, "    </DataArray>\n\n");
    if (atm) {
        vts_dump_array(f, c, "Topography", ATOMDEV_H, 1.0);
        vts_dump_array(f, c, "u-component", ATOMDEV_U, 1.0);
        vts_dump_array(f, c, "v-component", ATOMDEV_V, 1.0);
        vts_dump_array(f, c, "w-component", ATOMDEV_W, 1.0);
        fprintf(f, "    <DataArray type=\"Float32\" Name=\"Temperature\" format=\"ascii\">\n\n");
        for (int k = 0; k < km; k++) {
            for (int j = 0; j < jm; j++) {
                for (int i = 0; i < im; i++) fprintf(f, "%.4f\n", A3(c, ATOMDEV_T, i, j, k) * P->t_0 - P->t_0);
                fprintf(f, "\n\n");
            }
            fprintf(f, "\n\n");
        }
        fprintf(f, "\n\n");
        fprintf(f, "    </DataArray>\n\n");
        vts_dump_array(f, c, "Epsilon_3D", ATOMDEV_EPSILON_3D, 1.0);
        vts_dump_array(f, c, "WaterVapour", ATOMDEV_C, 1000.0);
        vts_dump_array(f, c, "CloudWater", ATOMDEV_CLOUD, 1000.0);
        vts_dump_array(f, c, "CloudIce", ATOMDEV_ICE, 1000.0);
        vts_dump_array(f, c, "PrecipitationRain", ATOMDEV_P_RAIN, 86400.);
        vts_dump_array(f, c, "PrecipitationSnow", ATOMDEV_P_SNOW, 86400.);
        vts_dump_array(f, c, "PressureDynamic", ATOMDEV_P_DYN, P->u_0 * P->u_0 * P->r_air * .01);
        vts_dump_array(f, c, "PressureStatic", ATOMDEV_P_STAT, 1.0);
        vts_dump_array(f, c, "BuoyancyForce", ATOMDEV_BUOYANCY, 1.0);
        vts_dump_array(f, c, "CO2-Concentration", ATOMDEV_CO2, 1.0);
        vts_dump_array(f, c, "Q_Latent", ATOMDEV_Q_LATENT, 1.0);
        vts_dump_array(f, c, "Q_Sensible", ATOMDEV_Q_SENSIBLE, 1.0);
    } else {
        vts_dump_array(f, c, "Topography", ATOMDEV_H, 1.0);
        vts_dump_array(f, c, "u-velocity", ATOMDEV_U, 1000.0);
        vts_dump_array(f, c, "v-velocity", ATOMDEV_V, 1.0);
        vts_dump_array(f, c, "w-velocity", ATOMDEV_W, 1.0);
        vts_dump_array(f, c, "Temperature", ATOMDEV_T, 1.0);
        vts_dump_array(f, c, "PressureDynamic", ATOMDEV_P_DYN, P->u_0 * P->u_0 * P->r_0_water * 1e-3);
        vts_dump_array(f, c, "PressureStatic", ATOMDEV_P_STAT, 1.0);
        vts_dump_array(f, c, "Salinity", ATOMDEV_C, 1.0);
        vts_dump_array(f, c, "DensityWater", ATOMDEV_R_WATER, 1.0);
        vts_dump_array(f, c, "DensitySaltWater", ATOMDEV_R_SALT_WATER, 1.0);
        vts_dump_array(f, c, "Salt_Finger", ATOMDEV_SALT_FINGER, 1.0);
        vts_dump_array(f, c, "SaltDiffusion", ATOMDEV_SALT_DIFFUSION, 1.0);
        vts_dump_array(f, c, "SaltBalance", ATOMDEV_SALT_BALANCE, 1.0);
        vts_dump_array(f, c, "BuoyancyForce", ATOMDEV_BUOYANCY, 1.0);
    }
    fprintf(f, "   </PointData>\n\n");
    fprintf(f, "   <Points>\n\n");
    fprintf(f, "    <DataArray type=\"Float32\" NumberOfComponents=\"3\" format=\"ascii\">\n\n");
    double x = 0., y = 0., z = 0., dx = atm ? .1 : .025, dy = .1, dz = .1;
    for (int k = 0; k < km; k++) {
        for (int j = 0; j < jm; j++) {
            for (int i = 0; i < im; i++) {
                if (k == 0 || j == 0) x = 0.;
                else x = x + dx;
                fprintf(f, "%.4f %.4f %.4f\n", x, y, z);
            }
            x = 0;
            y = y + dy;
            fprintf(f, "\n\n");
        }
        y = 0;
        z = z + dz;
        fprintf(f, "\n\n");
    }
    fprintf(f, "    </DataArray>\n\n");
    fprintf(f, "   </Points>\n\n");
    fprintf(f, "  </Piece>\n\n");
    fprintf(f, " </StructuredGrid>\n\n");
    fprintf(f, "</VTKFile>\n\n");
    return fclose(f) != 0;
}

int oracle_write_vtk(oracle_ctx* c, int section, int index, int Ma, const char* path) {
    FILE* f = fopen(path, "w");
    if (!f) return 1;
    if (c->model == ATOMDEV_ATM) {
        if (section == SEC_RADIAL) vtk_atm_radial(f, c, Ma, index);
        else if (section == SEC_LONGAL) vtk_atm_longal(f, c, index);
        else vtk_atm_zonal(f, c, index);
    } else if (section == SEC_RADIAL) {
        vtk_hyd_radial(f, c, index);
    } else {
        vtk_head(f, c, section == SEC_LONGAL ? "Longitudinal_Data_Hydrosphere_Circulation" : "Zonal_Data_Hydrosphere_Circulation",
                 section, section == SEC_LONGAL ? .025 : .05);
        vtk_dump(f, c, "Topography", ATOMDEV_H, 1., section, index);
        vtk_hyd_common(f, c, section, index, 1);
        vtk_vectors(f, c, section == SEC_LONGAL ? "VECTORS u-w-Cell float" : "VECTORS u-v-Cell float", section, index);
    }
    return fclose(f) != 0;
}

/* ---- the hydrosphere's start of a time slice (cHydrosphereModel.cpp:244-292) ---- */

/* BC_Thermohalin::IC_v_w_EkmanSpiral, hydrosphere/BC_Thermohalin.cpp:88-360: six latitude bands written out as in
 * the reference (three assignments in a row, each testing what the one before left), then the walk down the columns. */
static void ekman_band(oracle_ctx* c, int j0, int j1, int sel, int third, const double sv[3], const double sw[3],
                       double water_unused, double Ekman_angle) {
    const int L = c->im - 1;
    (void)water_unused;
    for (int j = j0; j < j1; j++)
        for (int k = 0; k < c->km; k++)
            if (is_fluid(c, L, j, k)) {
                double* v = &A3(c, ATOMDEV_V, L, j, k);
                double* w = &A3(c, ATOMDEV_W, L, j, k);
                double vel_magnitude = sqrt(*v * *v + *w * *w);
                if (*w == 0.) *w = 1.e-6;
                if (*v == 0.) *v = 1.e-6;
                if (vel_magnitude == 0.) vel_magnitude = 1.e-6;
                double alfa = asin(fabs(*w) / vel_magnitude), angle;
                int plus = sel == 0 ? *w >= 0. : (sel == 1 ? *v >= 0. : *v <= 0.);
                if (plus) angle = +alfa - Ekman_angle;
                else angle = -alfa - Ekman_angle;
                if (*w >= 0. && angle >= 0.) { *v = sv[0] * vel_magnitude * cos(angle); *w = sw[0] * vel_magnitude * sin(angle); }
                if (*w >= 0. && angle <= 0.) { *v = sv[1] * vel_magnitude * cos(angle); *w = sw[1] * vel_magnitude * sin(angle); }
                if (third == 0 ? *w <= 0. : (third == 1 ? *v <= 0. : *v >= 0.)) {
                    *v = sv[2] * vel_magnitude * cos(angle); *w = sw[2] * vel_magnitude * sin(angle);
                }
            }
}

void oracle_hyd_ic_ekman_spiral(oracle_ctx* c, int i_beg) {
    const int im = c->im, jm = c->jm, km = c->km, j_half = (jm - 1) / 2;
    const double water_wind = .03, u_0 = c->p.u_0;
    const double pi180 = 180. / M_PI, Ekman_angle = 45.0 / pi180;
    static const double P3[3] = { +1., +1., +1. }, M3[3] = { -1., -1., -1. };
    static const double ferrel_n_v[3] = { +1., +1., -1. }, ferrel_n_w[3] = { +1., -1., -1. };
    static const double ferrel_s_v[3] = { -1., -1., +1. }, ferrel_s_w[3] = { +1., -1., -1. };
    for (int j = 0; j < jm; j++)
        for (int k = 0; k < km; k++)
            if (is_fluid(c, im - 1, j, k)) {
                A3(c, ATOMDEV_V, im - 1, j, k) = water_wind * A3(c, ATOMDEV_V, im - 1, j, k) / u_0;
                A3(c, ATOMDEV_W, im - 1, j, k) = water_wind * A3(c, ATOMDEV_W, im - 1, j, k) / u_0;
            }
    ekman_band(c, 0, 30, 0, 0, P3, P3, 0., Ekman_angle);                    /* :106-140 */
    ekman_band(c, 30, 60, 1, 1, ferrel_n_v, ferrel_n_w, 0., Ekman_angle);   /* :143-177 */
    ekman_band(c, 60, 90, 0, 0, P3, P3, 0., Ekman_angle);                   /* :180-214 */
    ekman_band(c, 90, 120, 0, 0, M3, P3, 0., Ekman_angle);                  /* :217-251 */
    ekman_band(c, 120, 150, 2, 2, ferrel_s_v, ferrel_s_w, 0., Ekman_angle); /* :254-288 */
    ekman_band(c, 150, jm, 0, 0, M3, P3, 0., Ekman_angle);                  /* :291-324 */
    for (int j = 0; j < jm; j++)
        for (int i = im - 2; i >= i_beg; i--) {
            double gam_z = M_PI * (double)(i - i_beg) / (double)(im - 1 - i_beg);
            double exp_gam_z = exp(-gam_z), sin_gam_z = sin(gam_z), cos_gam_z = cos(gam_z);
            for (int k = 0; k < km; k++)
                if (is_fluid(c, i, j, k)) {
                    double v_g = A3(c, ATOMDEV_V, i + 1, j, k), w_g = A3(c, ATOMDEV_W, i + 1, j, k);
                    if (j > j_half) sin_gam_z = -sin(gam_z);
                    A3(c, ATOMDEV_V, i, j, k) = w_g * exp_gam_z * sin_gam_z + v_g * (1. - exp_gam_z * cos_gam_z);
                    A3(c, ATOMDEV_W, i, j, k) = w_g * (1. - exp_gam_z * cos_gam_z) - v_g * exp_gam_z * sin_gam_z;
                }
        }
}

/* BC_Thermohalin::IC_CircumPolar_Current, hydrosphere/BC_Thermohalin.cpp:676-690 */
void oracle_hyd_ic_circumpolar_current(oracle_ctx* c, int i_beg) {
    for (int i = i_beg; i < c->im; i++)
        for (int j = 147; j < 153; j++)
            for (int k = 0; k < c->km; k++)
                if (is_fluid(c, i, j, k)) A3(c, ATOMDEV_W, i, j, k) = .01 / c->p.u_0;
}

/* BC_Thermohalin::BC_Temperature_Salinity, hydrosphere/BC_Thermohalin.cpp:367-513 (without the printout) */
void oracle_hyd_bc_temperature_salinity(oracle_ctx* c, const atomdev_hyd_start* s) {
    const int im = c->im, jm = c->jm, km = c->km, i_beg = s->i_beg, i_max = s->i_max, Ma = s->Ma;
    const double ta = c->p.ta, ca = c->p.ca, t_0 = c->p.t_0, c_0 = c->p.c_0;
    double t_cretaceous_coeff = s->t_cretaceous_max / ((double)s->Ma_max_half - (double)(s->Ma_max_half * s->Ma_max_half / s->Ma_max));
    double t_cretaceous = t_cretaceous_coeff * (double)(-(Ma * Ma) / s->Ma_max + Ma);
    if (Ma == 0) t_cretaceous = 0.;
    t_cretaceous = 0.;
    double c_average = (s->t_average + 346.) / 10.;
    double c_cretaceous = (s->t_average + t_cretaceous + 346.) / 10.;
    c_cretaceous = c_cretaceous - c_average;
    if (Ma == 0) c_cretaceous = 0.;
    for (int k = 0; k < km; k++)
        for (int j = 0; j < jm; j++) {
            if (is_fluid(c, im - 1, j, k)) {
                A3(c, ATOMDEV_T, im - 1, j, k) = A3(c, ATOMDEV_T, im - 1, j, k) + t_cretaceous;
                double t_Celsius = A3(c, ATOMDEV_T, im - 1, j, k) * t_0 - t_0;
                if (t_Celsius < 4.) A3(c, ATOMDEV_T, im - 1, j, k) = ta + t_cretaceous;
                if (Ma == 0) A3(c, ATOMDEV_C, im - 1, j, k) = A3(c, ATOMDEV_C, im - 1, j, k) + c_cretaceous / c_0;
            } else {
                A3(c, ATOMDEV_T, im - 1, j, k) = ta + t_cretaceous;
                A3(c, ATOMDEV_C, im - 1, j, k) = ca + c_cretaceous / c_0;
            }
        }
    for (int k = 0; k < km; k++)
        for (int j = 0; j < jm; j++) {
            double tm_tbeg = (A3(c, ATOMDEV_T, im - 1, j, k) - ta) / (double)(i_max * i_max - i_beg * i_beg);
            double cm_cbeg = (A3(c, ATOMDEV_C, im - 1, j, k) - ca) / (double)(i_max * i_max - i_beg * i_beg);
            for (int i = i_beg; i < im - 1; i++) {
                if (is_fluid(c, i, j, k)) {
                    A3(c, ATOMDEV_T, i, j, k) = ta + tm_tbeg * (double)(i * i - i_beg * i_beg);
                    double t_Celsius = A3(c, ATOMDEV_T, i, j, k) * t_0 - t_0;
                    A3(c, ATOMDEV_C, i, j, k) = ca + cm_cbeg * (double)(i * i - i_beg * i_beg);
                    if (t_Celsius < 4.) { A3(c, ATOMDEV_T, i, j, k) = ta; A3(c, ATOMDEV_C, i, j, k) = ca; }
                } else {
                    A3(c, ATOMDEV_T, i, j, k) = ta; A3(c, ATOMDEV_C, i, j, k) = ca;
                }
            }
        }
    for (int i = 0; i < i_beg; i++)
        for (int j = 0; j < jm; j++)
            for (int k = 0; k < km; k++)
                if (is_fluid(c, i, j, k)) { A3(c, ATOMDEV_T, i, j, k) = ta; A3(c, ATOMDEV_C, i, j, k) = ca; }
}

/* BC_Thermohalin::BC_Surface_Salinity_NASA, hydrosphere/BC_Thermohalin.cpp:636-670, for files whose entries are all
 * numbers (the reference's stream extraction and fscanf part ways on anything else). */
int oracle_hyd_read_surface_salinity(oracle_ctx* c, const char* path) {
    FILE* f = fopen(path, "r");
    if (!f) return 1;
    double dummy_1, dummy_2, dummy_3 = 0.;
    for (int k = 0; k < c->km && !feof(f); k++)
        for (int j = 0; j < c->jm; j++) {
            if (fscanf(f, "%lf %lf %lf", &dummy_1, &dummy_2, &dummy_3) != 3) dummy_3 = 0.;
            if (dummy_3 < 0.) dummy_3 = c->p.ca;
            else A3(c, ATOMDEV_C, c->im - 1, j, k) = dummy_3 / c->p.c_0;
        }
    fclose(f);
    return 0;
}

/* lib/Utils.cpp:31-37 */
static void move_data(double* data, int len) {
    for (int i = 0; i < len / 2; i++) { double x = data[i]; data[i] = data[len / 2 + i]; data[len / 2 + i] = x; }
    data[len - 1] = data[0];
}

/* BC_Bathymetry_Atmosphere::BC_MountainSurface, atmosphere/BC_Bath_Atm.cpp:39-90, for well-formed files of jm x km lines */
int oracle_atm_read_mountain_surface(oracle_ctx* c, const char* path) {
    const int im = c->im, jm = c->jm, km = c->km;
    FILE* f = fopen(path, "r");
    if (!f) return 1;
    memset(c->f[ATOMDEV_H], 0, sizeof(double) * (size_t)im * jm * km);
    for (int j = 0; j < jm; j++)
        for (int k = 0; k < km; k++) {
            double lon, lat, height = -999;
            if (fscanf(f, "%lf %lf %lf", &lon, &lat, &height) != 3) { fclose(f); return 2; }
            if (height < 0.) {
                A3(c, ATOMDEV_H, 0, j, k) = A2(c, ATOMDEV_TOPOGRAPHY, j, k) = 0.;
            } else {
                int i_h = (int)floor(height / c->p.L_atm * (im - 1));
                A2(c, ATOMDEV_TOPOGRAPHY, j, k) = height;
                for (int i = 0; i <= i_h && i < im; i++) A3(c, ATOMDEV_H, i, j, k) = 1.;
            }
        }
    fclose(f);
    for (int j = 0; j < jm; j++) {
        move_data(&A2(c, ATOMDEV_TOPOGRAPHY, j, 0), km);
        for (int i = 0; i < im; i++) move_data(&A3(c, ATOMDEV_H, i, j, 0), km);
    }
    return 0;
}

/* BC_Bathymetry_Hydrosphere::BC_SeaGround, hydrosphere/BC_Bath_Hyd.cpp:31-114, for well-formed files of jm x km lines */
int oracle_hyd_read_sea_ground(oracle_ctx* c, const char* path) {
    const int im = c->im, jm = c->jm, km = c->km;
    FILE* f = fopen(path, "r");
    if (!f) return 1;
    memset(c->f[ATOMDEV_H], 0, sizeof(double) * (size_t)im * jm * km);
    for (int j = 0; j < jm; j++)
        for (int k = 0; k < km; k++) {
            double lon, lat, depth = 999;
            if (fscanf(f, "%lf %lf %lf", &lon, &lat, &depth) != 3) { fclose(f); return 2; }
            if (depth > 0.) depth = 0;
            int i_boden = (im - 1) + (int)floor(depth / c->p.L_hyd * (im - 1));
            A2(c, ATOMDEV_TOPOGRAPHY, j, k) = -depth;
            for (int i = 0; i <= i_boden; i++) A3(c, ATOMDEV_H, i, j, k) = 1.;
        }
    fclose(f);
    for (int j = 0; j < jm; j++) {
        move_data(&A2(c, ATOMDEV_TOPOGRAPHY, j, 0), km);
        for (int i = 0; i < im; i++) move_data(&A3(c, ATOMDEV_H, i, j, 0), km);
    }
    double h_center = 0.;
    for (int k = 2; k < km - 2; k++)
        for (int j = 2; j < jm - 2; j++)
            for (int i = 2; i < im - 2; i++) {
                if (is_land(c, i, j, k) && (is_fluid(c, i, j + 1, k) && is_fluid(c, i, j - 1, k))
                    && (is_fluid(c, i, j, k + 1) && is_fluid(c, i, j, k - 1))) {
                    h_center = A3(c, ATOMDEV_H, i, j, k);
                    A3(c, ATOMDEV_H, i, j, k) = 0.;
                }
                if (h_center == 1. && is_fluid(c, i, j - 2, k) && is_fluid(c, i, j, k + 2)) A3(c, ATOMDEV_H, i, j - 1, k + 1) = 0.;
                if (h_center == 1. && is_fluid(c, i, j - 2, k) && is_fluid(c, i, j, k - 2)) A3(c, ATOMDEV_H, i, j - 1, k - 1) = 0.;
                if (h_center == 1. && is_fluid(c, i, j + 2, k) && is_fluid(c, i, j, k - 2)) A3(c, ATOMDEV_H, i, j + 1, k - 1) = 0.;
                if (h_center == 1. && is_fluid(c, i, j + 2, k) && is_fluid(c, i, j, k + 2)) A3(c, ATOMDEV_H, i, j + 1, k + 1) = 0.;
            }
    return 0;
}

/* BC_Thermohalin::BC_Pressure_Density, hydrosphere/BC_Thermohalin.cpp:518-583.  The local
 * r_air / R_Air of the reference function (:533-534) shadow the model parameters. */
void oracle_hyd_pressure_density(oracle_ctx* c) {
    const atomdev_params* P = &c->p;
    const int im = c->im;
    const double E_water = 2.15e9, beta_water = 8.8e-5, r_air = 1.2041, R_Air = 287.1;
    const double t_0 = P->t_0, c_0 = P->c_0;
    for (int k = 0; k < c->km; k++)
        for (int j = 0; j < c->jm; j++) {
            double t1 = A3(c, ATOMDEV_T, im - 1, j, k), cc = A3(c, ATOMDEV_C, im - 1, j, k);
            A3(c, ATOMDEV_P_STAT, im - 1, j, k) = .01 * (r_air * R_Air * t1 * t_0) / 1000.;
            A3(c, ATOMDEV_R_WATER, im - 1, j, k) = P->r_0_water;
            double t_Celsius_1 = t1 * t_0 - t_0;
            double C_p = 999.83, beta_p = .808;
            double alfa_t_p = .0708 * (1. + .068 * t_Celsius_1);
            double gamma_t_p = .003 * (1. - .012 * t_Celsius_1);
            A3(c, ATOMDEV_R_SALT_WATER, im - 1, j, k) = C_p + beta_p * cc * c_0 - alfa_t_p * t_Celsius_1
                                                        - gamma_t_p * (c_0 - cc * c_0) * t_Celsius_1;
        }
    for (int k = 0; k < c->km; k++)
        for (int j = 0; j < c->jm; j++)
            for (int i = im - 2; i >= 0; i--) {
                double d_i = (double)(im - 1 - i);
                double cc = A3(c, ATOMDEV_C, i, j, k);
                double t_Celsius_1 = A3(c, ATOMDEV_T, i, j, k) * t_0 - t_0;
                double t_Celsius_0 = A3(c, ATOMDEV_T, i + 1, j, k) * t_0 - t_0;
                double rw1 = A3(c, ATOMDEV_R_WATER, i + 1, j, k);
                double ps = rw1 * P->g * d_i * (P->L_hyd / (double)(im - 1)) / 100000. + P->p_0 / 1000.;
                A3(c, ATOMDEV_P_STAT, i, j, k) = ps;
                A3(c, ATOMDEV_R_WATER, i, j, k) = rw1 / (1. + beta_water * (t_Celsius_1 - t_Celsius_0))
                                                  / (1. - (ps - A3(c, ATOMDEV_P_STAT, i + 1, j, k)) / E_water * 1e5);
                double p_km = (double)(im - 1 - i) * (P->L_hyd / (double)(im - 1)) / 1000.;
                double C_p = 999.83 + 5.053 * p_km - .048 * p_km * p_km;
                double beta_p = .808 - .0085 * p_km;
                double alfa_t_p = .0708 * (1. + .351 * p_km + .068 * (1. - .0683 * p_km) * t_Celsius_1);
                double gamma_t_p = .003 * (1. - .059 * p_km - .012 * (1. - .064 * p_km) * t_Celsius_1);
                A3(c, ATOMDEV_R_SALT_WATER, i, j, k) = C_p + beta_p * cc * c_0 - alfa_t_p * t_Celsius_1
                                                       - gamma_t_p * (c_0 - cc * c_0) * t_Celsius_1;
            }
}

/* BC_Bathymetry_Hydrosphere::BC_SolidGround, hydrosphere/BC_Bath_Hyd.cpp:120-139 */
void oracle_hyd_solid_ground(oracle_ctx* c) {
    for (int i = 0; i < c->im; i++)
        for (int j = 0; j < c->jm; j++)
            for (int k = 0; k < c->km; k++)
                if (is_land(c, i, j, k)) {
                    A3(c, ATOMDEV_P_DYN, i, j, k) = A3(c, ATOMDEV_P_DYNN, i, j, k) = c->p.pa;
                    A3(c, ATOMDEV_T, i, j, k) = A3(c, ATOMDEV_TN, i, j, k) = c->p.ta;
                    A3(c, ATOMDEV_C, i, j, k) = A3(c, ATOMDEV_CN, i, j, k) = c->p.ca;
                    A3(c, ATOMDEV_U, i, j, k) = A3(c, ATOMDEV_UN, i, j, k) = 0.;
                    A3(c, ATOMDEV_V, i, j, k) = A3(c, ATOMDEV_VN, i, j, k) = 0.;
                    A3(c, ATOMDEV_W, i, j, k) = A3(c, ATOMDEV_WN, i, j, k) = 0.;
                }
}

/* AtomUtils::move_data_to_new_arrays, lib/Utils.cpp:41-68, with the lists of
 * cAtmosphereModel.cpp:60-63 / cHydrosphereModel (u v w t p_dyn c). */
void oracle_move_data_to_new_arrays(oracle_ctx* c, int level) {
    static const int O3[9] = { ATOMDEV_U, ATOMDEV_V, ATOMDEV_W, ATOMDEV_T, ATOMDEV_P_DYN, ATOMDEV_C, ATOMDEV_CLOUD, ATOMDEV_ICE, ATOMDEV_CO2 };
    static const int O2[3] = { ATOMDEV_V, ATOMDEV_W, ATOMDEV_P_DYN };
    const double coeff = 1.;
    const int n3 = c->model == ATOMDEV_ATM ? 9 : 6;
    const int i0 = level < 0 ? 0 : level, i1 = level < 0 ? c->im : level + 1;
    const int* L = level < 0 ? O3 : O2;
    const int nl = level < 0 ? n3 : 3;
    for (int i = i0; i < i1; i++)
        for (int j = 0; j < c->jm; j++)
            for (int k = 0; k < c->km; k++)
                for (int n = 0; n < nl; n++)
                    A3(c, L[n] + 9, i, j, k) = coeff * A3(c, L[n], i, j, k); /* xn ids = x ids + 9 */
}

/* ------------------------------------------------------------------------- */
/* Accuracy                                                                  */
/* ------------------------------------------------------------------------- */

/* Accuracy_Atm::residuumQuery_3D, atmosphere/Accuracy_Atm.cpp:82-116 (strict >,
 * state carried by the caller) and Accuracy_Hyd::residuumQuery_3D,
 * hydrosphere/Accuracy_Hyd.cpp:88-116 (>=, state reset). */
static void residuum3d(oracle_ctx* c, int hyd, double* min, int loc[3]) {
    const double dr = c->p.dr, dthe = c->p.dthe, dphi = c->p.dphi;
    if (hyd) *min = 0.;
    for (int i = 1; i < c->im - 1; i++)
        for (int j = 1; j < c->jm - 1; j++) {
            double sinthe = sin(c->the[j]), costhe = cos(c->the[j]);
            double rmsinthe = c->rad[i] * sinthe;
            for (int k = 1; k < c->km - 1; k++) {
                double dudr = (A3(c, ATOMDEV_U, i + 1, j, k) - A3(c, ATOMDEV_U, i - 1, j, k)) / (2. * dr);
                double dvdthe = (A3(c, ATOMDEV_V, i, j + 1, k) - A3(c, ATOMDEV_V, i, j - 1, k)) / (2. * dthe);
                double dwdphi = (A3(c, ATOMDEV_W, i, j, k + 1) - A3(c, ATOMDEV_W, i, j, k - 1)) / (2. * dphi);
                double residuum = dudr + 2. * A3(c, ATOMDEV_U, i, j, k) / c->rad[i] + dvdthe / c->rad[i]
                                  + costhe / rmsinthe * A3(c, ATOMDEV_V, i, j, k) + dwdphi / rmsinthe;
                if (c->f[ATOMDEV_RESIDUUM]) A3(c, ATOMDEV_RESIDUUM, i, j, k) = residuum;
                if (hyd ? (fabs(residuum) >= *min) : (fabs(residuum) > *min)) {
                    *min = residuum; loc[0] = i; loc[1] = j; loc[2] = k;
                }
            }
        }
}
void oracle_atm_residuum3d(oracle_ctx* c, double* min, int loc[3]) { residuum3d(c, 0, min, loc); }
void oracle_hyd_residuum3d(oracle_ctx* c, double* min, int loc[3]) { residuum3d(c, 1, min, loc); }

/* Accuracy_Atm::residuumQuery_2D, atmosphere/Accuracy_Atm.cpp:54-80 */
void oracle_atm_residuum2d(oracle_ctx* c, double* min, int loc[3]) {
    const double dthe = c->p.dthe, dphi = c->p.dphi;
    for (int j = 1; j < c->jm - 1; j++) {
        double sinthe = sin(c->the[j]), costhe = cos(c->the[j]);
        double rmsinthe = c->rad[0] * sinthe;
        for (int k = 1; k < c->km - 1; k++) {
            double dvdthe = (A3(c, ATOMDEV_V, 0, j + 1, k) - A3(c, ATOMDEV_V, 0, j - 1, k)) / (2. * dthe);
            double dwdphi = (A3(c, ATOMDEV_W, 0, j, k + 1) - A3(c, ATOMDEV_W, 0, j, k - 1)) / (2. * dphi);
            double residuum = dvdthe / c->rad[0] + costhe / rmsinthe * A3(c, ATOMDEV_V, 0, j, k) + dwdphi / rmsinthe;
            if (c->f[ATOMDEV_RESIDUUM]) A3(c, ATOMDEV_RESIDUUM, 0, j, k) = residuum;
            if (fabs(residuum) > *min) { *min = residuum; loc[0] = 0; loc[1] = j; loc[2] = k; }
        }
    }
}

static void steady(oracle_ctx* c, const int* F, int nf, int i0, int i1, int ge, double* out, int* loc) {
    for (int n = 0; n < nf; n++) { out[n] = 0.; loc[3 * n] = loc[3 * n + 1] = loc[3 * n + 2] = 0; }
    for (int i = i0; i < i1; i++)
        for (int j = 0; j < c->jm; j++)
            for (int k = 0; k < c->km; k++)
                for (int n = 0; n < nf; n++) {
                    double tmp = fabs(A3(c, F[n], i, j, k) - A3(c, F[n] + 9, i, j, k));
                    if (ge ? (tmp >= out[n]) : (tmp > out[n])) {
                        out[n] = tmp; loc[3 * n] = i; loc[3 * n + 1] = j; loc[3 * n + 2] = k;
                    }
                }
}
/* Accuracy_Atm::steadyQuery_3D, atmosphere/Accuracy_Atm.cpp:192-305 */
void oracle_atm_steady3d(oracle_ctx* c, double out[9], int loc[27]) {
    static const int F[9] = { ATOMDEV_U, ATOMDEV_V, ATOMDEV_W, ATOMDEV_T, ATOMDEV_C, ATOMDEV_CLOUD, ATOMDEV_ICE, ATOMDEV_CO2, ATOMDEV_P_DYN };
    steady(c, F, 9, 0, c->im, 0, out, loc);
}
/* Accuracy_Atm::steadyQuery_2D, atmosphere/Accuracy_Atm.cpp:118-153 */
void oracle_atm_steady2d(oracle_ctx* c, double out[3], int loc[9]) {
    static const int F[3] = { ATOMDEV_V, ATOMDEV_W, ATOMDEV_P_DYN };
    steady(c, F, 3, 0, 1, 0, out, loc);
}
/* Accuracy_Hyd::steadyQuery_3D, hydrosphere/Accuracy_Hyd.cpp:121-184 */
void oracle_hyd_steady3d(oracle_ctx* c, double out[6], int loc[18]) {
    static const int F[6] = { ATOMDEV_U, ATOMDEV_V, ATOMDEV_W, ATOMDEV_T, ATOMDEV_C, ATOMDEV_P_DYN };
    steady(c, F, 6, 0, c->im, 1, out, loc);
}

/* ------------------------------------------------------------------------- */
/* Hydrosphere RK                                                            */
/* ------------------------------------------------------------------------- */

enum { HU = 0, HV, HW, HT, HP, HC };
static const int HYD_F[6] = { ATOMDEV_U, ATOMDEV_V, ATOMDEV_W, ATOMDEV_T, ATOMDEV_P_DYN, ATOMDEV_C };

/* RHS_Hydrosphere::RK_RHS_3D_Hydrosphere, hydrosphere/RHS_Hyd.cpp:60-447.
 * rhs order: t u v w c. */
static void hyd_rhs3d(oracle_ctx* c, int i, int j, int k, double rhs[5]) {
    const atomdev_params* P = &c->p;
    const int im = c->im;
    const double k_Force = 1.;
    const double dr = P->dr, dthe = P->dthe, dphi = P->dphi, re = P->re;
    double dr2 = dr * dr, dthe2 = dthe * dthe, dphi2 = dphi * dphi;
    double rm = c->rad[i], rm2 = rm * rm;
    double sinthe = sin(c->the[j]), sinthe2 = sinthe * sinthe, costhe = cos(c->the[j]);
    double rmsinthe = rm * sinthe, rm2sinthe = rm2 * sinthe, rm2sinthe2 = rm2 * sinthe2;

    double h0[3], hd[3];
    ib_factors(c, i, j, k, 1, P->L_hyd, h0, hd);
    const double h_0_i = h0[0], h_0_j = h0[1], h_0_k = h0[2];
    const double h_d_i = hd[0], h_d_j = hd[1], h_d_k = hd[2];

    derivs D;
    fd_derivatives(c, HYD_F, 6, HP, HU, HC, i, j, k, hd, 1, &D);

    const double u = A3(c, ATOMDEV_U, i, j, k), v = A3(c, ATOMDEV_V, i, j, k), w = A3(c, ATOMDEV_W, i, j, k);
    const double t = A3(c, ATOMDEV_T, i, j, k), cs = A3(c, ATOMDEV_C, i, j, k);

    /* RHS_Hyd.cpp:345-402 */
    double salinity_evaporation = 0., salinity_surface = 0.;
    double drodc = .7;
    double salt_water_ref = A3(c, ATOMDEV_R_WATER, i, j, k) + drodc * cs * P->c_0;
    double coeff_buoy = P->r_0_water * P->u_0 * P->u_0 / P->L_hyd;
    double coeff_salinity = 1.1574e-5 * P->L_hyd / P->u_0;

    if (i == im - 2) {
        salinity_surface = salt_water_ref * (-3. * A3(c, ATOMDEV_C, im - 1, j, k) + 4. * A3(c, ATOMDEV_C, im - 2, j, k)
                                             - A3(c, ATOMDEV_C, im - 3, j, k)) / (2. * dr)
                           * (1. - 2. * A3(c, ATOMDEV_C, im - 1, j, k))
                           * (A2(c, ATOMDEV_EVAPORATION_DALTON, j, k) - A2(c, ATOMDEV_PRECIPITATION, j, k));
        salinity_evaporation = +coeff_salinity * salinity_surface;
        if (is_land(c, i, j, k)) salinity_evaporation = 0.;
        if (salinity_evaporation >= 20.) salinity_evaporation = 20.;
        if (salinity_evaporation <= -20.) salinity_evaporation = -20.;
    } else salinity_evaporation = 0.;

    double RS_buoyancy_Momentum = P->Buoyancy * P->g * (A3(c, ATOMDEV_R_SALT_WATER, i, j, k) - salt_water_ref) / salt_water_ref;
    A3(c, ATOMDEV_BUOYANCY, i, j, k) = RS_buoyancy_Momentum * coeff_buoy * 1000.;
    A3(c, ATOMDEV_SALT_BALANCE, i, j, k) = salt_water_ref - A3(c, ATOMDEV_R_SALT_WATER, i, j, k);
    if (A3(c, ATOMDEV_SALT_BALANCE, i, j, k) < 0.) {
        A3(c, ATOMDEV_SALT_DIFFUSION, i, j, k) = A3(c, ATOMDEV_SALT_BALANCE, i, j, k);
        A3(c, ATOMDEV_SALT_FINGER, i, j, k) = 0.;
    } else {
        A3(c, ATOMDEV_SALT_FINGER, i, j, k) = A3(c, ATOMDEV_SALT_BALANCE, i, j, k);
        A3(c, ATOMDEV_SALT_DIFFUSION, i, j, k) = 0.;
    }
    if (is_land(c, i, j, k)) {
        A3(c, ATOMDEV_SALT_BALANCE, i, j, k) = 0.;
        A3(c, ATOMDEV_SALT_FINGER, i, j, k) = 0.;
        A3(c, ATOMDEV_SALT_DIFFUSION, i, j, k) = 0.;
        A3(c, ATOMDEV_BUOYANCY, i, j, k) = 0.;
        A3(c, ATOMDEV_R_WATER, i, j, k) = 1007.;
        A3(c, ATOMDEV_R_SALT_WATER, i, j, k) = 1032.;
    }

    /* RHS_Hyd.cpp:404-437 */
    double rhs_t = -(u * DR(HT) + v * DT(HT) / rm + w * DP(HT) / rmsinthe)
                   + (D2R(HT) + DR(HT) * 2. / rm + D2T(HT) / rm2 + DT(HT) * costhe / rm2sinthe + D2P(HT) / rm2sinthe2) / (re * P->pr)
                   - h_0_i * t * k_Force / dr2;

    double rhs_u = -(u * DR(HU) + v * DT(HU) / rm + w * DP(HU) / rmsinthe)
                   + h_d_i * DR(HP) / salt_water_ref
                   + (D2R(HU) + h_d_i * 2. * u / rm2 + D2T(HU) / rm2 + 4. * DR(HU) / rm + DT(HU) * costhe / rm2sinthe + D2P(HU) / rm2sinthe2) / re
                   + RS_buoyancy_Momentum
                   - h_0_i * u * k_Force / dthe2;

    double rhs_v = -(u * DR(HV) + v * DT(HV) / rm + w * DP(HV) / rmsinthe)
                   - h_d_j * DT(HP) / rm / salt_water_ref
                   + (D2R(HV) + DR(HV) * 2. / rm + D2T(HV) / rm2 + DT(HV) / rm2sinthe * costhe
                      - (1. + costhe * costhe / sinthe2) * h_d_j * v + D2P(HV) / rm2sinthe2
                      + 2. * DT(HU) / rm2 - DP(HW) * 2. * costhe / rm2sinthe2) / re
                   - h_0_j * v * k_Force / dthe2;

    double rhs_w = -(u * DR(HW) + v * DT(HW) / rm + w * DP(HW) / rmsinthe)
                   - h_d_k * DP(HP) / rmsinthe / salt_water_ref
                   + (D2R(HW) + DR(HW) * 2. / rm + D2T(HW) / rm2 + DT(HW) / rm2sinthe * costhe
                      - (1. + costhe * costhe / sinthe2) * h_d_k * w + D2P(HW) / rm2sinthe2
                      + 2. * DP(HU) / rm2sinthe + DP(HV) * 2. * costhe / rm2sinthe2) / re
                   - h_0_k * w * k_Force / dphi2;

    double rhs_c = -(u * DR(HC) + v * DT(HC) / rm + w * DP(HC) / rmsinthe)
                   + (D2R(HC) + DR(HC) * 2. / rm + D2T(HC) / rm2 + DT(HC) * costhe / rm2sinthe + D2P(HC) / rm2sinthe2) / (P->sc * re)
                   + salinity_evaporation
                   - h_0_i * cs * k_Force / dr2;

    /* RHS_Hyd.cpp:439-446 */
    A3(c, ATOMDEV_AUX_U, i, j, k) = rhs_u + h_d_i * DR(HP) / salt_water_ref;
    A3(c, ATOMDEV_AUX_V, i, j, k) = rhs_v + h_d_j * DT(HP) / rm / salt_water_ref;
    A3(c, ATOMDEV_AUX_W, i, j, k) = rhs_w + h_d_k * DP(HP) / rmsinthe / salt_water_ref;
    if (is_land(c, i, j, k))
        A3(c, ATOMDEV_AUX_U, i, j, k) = A3(c, ATOMDEV_AUX_V, i, j, k) = A3(c, ATOMDEV_AUX_W, i, j, k) = 0.;

    rhs[0] = rhs_t; rhs[1] = rhs_u; rhs[2] = rhs_v; rhs[3] = rhs_w; rhs[4] = rhs_c;
}

/* RungeKutta_Hydrosphere::solveRungeKutta_3D_Hydrosphere, hydrosphere/RungeKutta_Hyd.cpp:30-124.
 * Stage 3 advances w from its stage-2 value, not from wn (line 101). */
void oracle_hyd_rk3d(oracle_ctx* c) {
    static const int X[5] = { ATOMDEV_T, ATOMDEV_U, ATOMDEV_V, ATOMDEV_W, ATOMDEV_C };
    const double dt = c->p.dt;
    double k1[5], k2[5], k3[5], k4[5];
    for (int i = 1; i < c->im - 1; i++)
        for (int j = 1; j < c->jm - 1; j++)
            for (int k = 1; k < c->km - 1; k++) {
                hyd_rhs3d(c, i, j, k, k1);
                for (int n = 0; n < 5; n++) A3(c, X[n], i, j, k) = A3(c, X[n] + 9, i, j, k) + k1[n] * .5 * dt;
                hyd_rhs3d(c, i, j, k, k2);
                for (int n = 0; n < 5; n++) A3(c, X[n], i, j, k) = A3(c, X[n] + 9, i, j, k) + k2[n] * .5 * dt;
                hyd_rhs3d(c, i, j, k, k3);
                for (int n = 0; n < 5; n++) {
                    double base = (n == 3) ? A3(c, ATOMDEV_W, i, j, k) : A3(c, X[n] + 9, i, j, k);
                    A3(c, X[n], i, j, k) = base + k3[n] * dt;
                }
                hyd_rhs3d(c, i, j, k, k4);
                for (int n = 0; n < 5; n++)
                    A3(c, X[n], i, j, k) = A3(c, X[n] + 9, i, j, k) + dt * (k1[n] + 2. * k2[n] + 2. * k3[n] + k4[n]) / 6.;
            }
}

/* solveRungeKutta_2D_Hydrosphere: the RHS object forces im = 41 (RHS_Hyd.cpp:466),
 * the RK driver uses the grid's own im-1. */
void oracle_hyd_rk2d(oracle_ctx* c) { rk2d(c, 40, c->im - 1, 1, c->p.r_0_water); }

/* Reference defaults (param.py:32-177; grid constants cAtmosphereModel.cpp:38-55). */
void oracle_default_params(int model, atomdev_params* p) {
    memset(p, 0, sizeof(*p));
    const double pi180 = 180. / M_PI;
    p->dt = 0.00001; p->dr = 0.025; p->dthe = 1. / pi180; p->dphi = 1. / pi180;
    p->r0 = 1.; p->the0 = 0.; p->phi0 = 0.;
    p->g = 9.8066; p->Buoyancy = 1.0; p->t_0 = 273.15; p->p_0 = 1013.25; /* param.py:92,151 */
    p->ep = .623; p->hp = 6.1078; p->lv = 2.52e6; p->ls = 2.83e6; p->r_water_vapour = .0094; /* param.py:77-85 */
    p->co2_0 = 280.; /* param.py:95 */
    if (model == ATOMDEV_ATM) {
        p->re = 1000.; p->pr = .7179; p->u_0 = 8.; p->c_0 = .035;
    } else {
        p->re = 10.; p->pr = 6.957; p->u_0 = .25; p->c_0 = 34.6;
    }
    p->sc_WaterVapour = .61; p->sc_CO2 = .96; p->gam = .65; p->r_air = 1.2041; p->R_Air = 287.1;
    p->R_WaterVapour = 461.6; p->L_atm = 16000.; p->cp_l = 1005.; p->t_tropopause = .798;
    p->sc = 1.7329; p->L_hyd = 1000.; p->r_0_water = 1000.; p->pa = 0.; p->ca = 1.;
    p->ta = model == ATOMDEV_ATM ? 1. : 1.01464; /* param.py:102, 163 */
}

/* ------------------------------------------------------------------------- */
/* The atmosphere's start of a time slice (atmosphere/cAtmosphereModel.cpp:220-302)  */
/* ------------------------------------------------------------------------- */

/* BC_Thermo's i_topography (atmosphere/BC_Thermo.cpp:128-139): the highest land level at or below im-2, else 0 */
static int topography_level(const oracle_ctx* c, int j, int k) {
    for (int i = c->im - 2; i >= 0; i--) if (is_land(c, i, j, k)) return i;
    return 0;
}

/* BC_Thermo::GetTropopauseHightAdd, atmosphere/BC_Thermo.cpp:3298-3302, called with t_cretaceous / t_0 */
static int tropopause_add(double t_cretaceous, double t_0) { return (int)round(((t_cretaceous / t_0) * t_0) / 2.6); }

/* BC_Thermo::IC_CellStructure, atmosphere/BC_Thermo.cpp:934-2307.  The reference's loops never combine different
 * longitudes, so the statements run here for one longitude after the other, each in the reference's order. */
static void cell_rows_u(oracle_ctx* c, int k, int j0, int j1, double ua, const int* trop) { /* e.g. :1014-1024 */
    for (int j = j0; j <= j1; j++) {
        const int i_max = trop[j], i_half = trop[j] / 2;
        const double d_i_half = (double)i_half;
        for (int i = 0; i <= i_max; i++) {
            const double d_i = (double)i;
            A3(c, ATOMDEV_U, i, j, k) = -ua * (d_i * d_i / (d_i_half * d_i_half) - 2. * d_i / d_i_half);
        }
    }
}
/* below the tropopause (e.g. :1082-1094); F = ATOMDEV_V or ATOMDEV_W */
static void cell_rows_low(oracle_ctx* c, int F, int k, int j0, int j1, double x_trop, double x_sl, const int* trop) {
    for (int j = j0; j <= j1; j++) {
        const int i_max = trop[j];
        const double d_i_max = (double)i_max;
        for (int i = 0; i < i_max; i++) {
            const double d_i = (double)i;
            A3(c, F, i, j, k) = (x_trop - x_sl) * d_i / d_i_max + x_sl;
        }
    }
}
/* from the tropopause up (e.g. :1095-1106); guard: 0 none (:1330-1340), 1 "d_i_max == 0." (:1098), 2 "i_max == 40" (:1044) */
static void cell_rows_high(oracle_ctx* c, int F, int k, int j0, int j1, double x_trop, int guard, const int* trop) {
    const int im = c->im;
    for (int j = j0; j <= j1; j++) {
        const int i_max = trop[j];
        double d_i_max = (double)(i_max - (im - 1));
        if (guard == 1 && d_i_max == 0.) d_i_max = 1.e-6;
        if (guard == 2 && i_max == 40) d_i_max = 1.e-6;
        for (int i = i_max; i < im; i++) {
            const double d_i = (double)(i - (im - 1));
            A3(c, F, i, j, k) = x_trop * d_i / d_i_max;
        }
    }
}
/* x[j] = (x[ja] - x[jb]) * (j - jb) / (ja - jb) + x[jb] for j from j0 to j1 in steps of dir, in place (e.g. :1472-1484) */
static void cell_interpolate(oracle_ctx* c, int F, int k, int j0, int j1, int dir, int ja, int jb) {
    const double d_b = (double)jb, d_diff = (double)ja - (double)jb;
    for (int j = j0; dir > 0 ? j <= j1 : j >= j1; j += dir) {
        const double d_j = (double)j;
        for (int i = 0; i < c->im; i++)
            A3(c, F, i, j, k) = (A3(c, F, i, ja, k) - A3(c, F, i, jb, k)) * (d_j - d_b) / d_diff + A3(c, F, i, jb, k);
    }
}
/* seven rows around a cell border (e.g. :2144-2157) */
static void cell_smooth(oracle_ctx* c, int k, int jc) {
    static const int F[3] = { ATOMDEV_U, ATOMDEV_V, ATOMDEV_W };
    const int j_s = jc - 3, j_n = jc + 3;
    for (int i = 0; i < c->im; i++)
        for (int j = j_s; j <= j_n; j++)
            for (int n = 0; n < 3; n++)
                A3(c, F[n], i, j, k) = (A3(c, F[n], i, j_n, k) - A3(c, F[n], i, j_s, k)) / ((int)(j_n - j_s)) * (int)(j - j_s) + A3(c, F[n], i, j_s, k);
}

int oracle_atm_ic_cell_structure(oracle_ctx* c, const int* trop) {
    if (c->jm != 181) return 1;   /* the cell borders are rows of the 1-degree grid (:987-999) */
    const int im = c->im, jm = c->jm, km = c->km, V = ATOMDEV_V, W = ATOMDEV_W;
    /* :944-984 */
    const double ua_00 = 1., va_equator_SL = 0.000, va_equator_Tropopause = 0.000, wa_equator_SL = -1., wa_equator_Tropopause = -7.5;
    const double ua_30 = -1., va_Hadley_SL = .25, va_Hadley_Tropopause = -1., va_Hadley_SL_15 = 1., va_Hadley_Tropopause_15 = -1.,
                 wa_Hadley_SL = 1., wa_Hadley_Tropopause = 30.;
    const double ua_60 = 0.5, va_Ferrel_SL = 0.5, va_Ferrel_Tropopause = 1., va_Ferrel_SL_45 = -.1, va_Ferrel_Tropopause_45 = 1.,
                 wa_Ferrel_SL = -.2, wa_Ferrel_Tropopause = 10.;
    const double ua_90 = -0.5, va_Polar_SL = 0., va_Polar_Tropopause = 0., va_Polar_SL_75 = .5, va_Polar_Tropopause_75 = -1.,
                 wa_Polar_SL = -0.01, wa_Polar_Tropopause = 0.;
    const int j_aeq = 90, j_pol_n = 0, j_pol_s = jm - 1, j_pol_v_n = 15, j_pol_v_s = 165, j_fer_n = 30, j_fer_s = 150,
              j_fer_v_n = 45, j_fer_v_s = 135, j_had_n = 60, j_had_s = 120, j_had_v_n = 75, j_had_v_s = 105;
    for (int k = 0; k < km; k++) {
        /* equator, :1014-1052 */
        cell_rows_u(c, k, j_aeq, j_aeq, ua_00, trop);
        cell_rows_low(c, V, k, j_aeq, j_aeq, va_equator_Tropopause, va_equator_SL, trop);
        cell_rows_low(c, W, k, j_aeq, j_aeq, wa_equator_Tropopause, wa_equator_SL, trop);
        cell_rows_high(c, V, k, j_aeq, j_aeq, va_equator_Tropopause, 2, trop);
        cell_rows_high(c, W, k, j_aeq, j_aeq, wa_equator_Tropopause, 2, trop);
        /* northern polar cell, :1068-1133 */
        cell_rows_u(c, k, 0, j_pol_n, ua_90, trop);
        cell_rows_low(c, V, k, j_pol_n, j_fer_n, va_Polar_Tropopause, va_Polar_SL, trop);
        cell_rows_low(c, W, k, j_pol_n, j_fer_n, wa_Polar_Tropopause, wa_Polar_SL, trop);
        cell_rows_high(c, V, k, j_pol_n, j_fer_n, va_Polar_Tropopause, 1, trop);
        cell_rows_high(c, W, k, j_pol_n, j_fer_n, wa_Polar_Tropopause, 1, trop);
        cell_rows_low(c, V, k, j_pol_v_n, j_pol_v_n, va_Polar_Tropopause_75, va_Polar_SL_75, trop);
        cell_rows_high(c, V, k, j_pol_v_n, j_pol_v_n, va_Polar_Tropopause_75, 1, trop);
        /* northern Ferrel cell, :1143-1209 */
        cell_rows_u(c, k, j_fer_n, j_fer_n, ua_60, trop);
        cell_rows_low(c, V, k, j_fer_n, j_fer_n, va_Ferrel_Tropopause, va_Ferrel_SL, trop);
        cell_rows_low(c, W, k, j_fer_n, j_fer_n, wa_Ferrel_Tropopause, wa_Ferrel_SL, trop);
        cell_rows_high(c, V, k, j_fer_n, j_fer_n, va_Ferrel_Tropopause, 1, trop);
        cell_rows_high(c, W, k, j_fer_n, j_fer_n, wa_Ferrel_Tropopause, 1, trop);
        cell_rows_low(c, V, k, j_fer_v_n, j_fer_v_n, va_Ferrel_Tropopause_45, va_Ferrel_SL_45, trop);
        cell_rows_high(c, V, k, j_fer_v_n, j_fer_v_n, va_Ferrel_Tropopause_45, 1, trop);
        /* northern Hadley cell, :1218-1283 */
        cell_rows_u(c, k, j_had_n, j_had_n, ua_30, trop);
        cell_rows_low(c, V, k, j_had_n, j_had_n, va_Hadley_Tropopause, va_Hadley_SL, trop);
        cell_rows_low(c, W, k, j_had_n, j_had_n, wa_Hadley_Tropopause, wa_Hadley_SL, trop);
        cell_rows_high(c, V, k, j_had_n, j_had_n, va_Hadley_Tropopause, 1, trop);
        cell_rows_high(c, W, k, j_had_n, j_had_n, wa_Hadley_Tropopause, 1, trop);
        cell_rows_low(c, V, k, j_had_v_n, j_had_v_n, va_Hadley_Tropopause_15, va_Hadley_SL_15, trop);
        cell_rows_high(c, V, k, j_had_v_n, j_had_v_n, va_Hadley_Tropopause_15, 1, trop);
        /* southern Hadley cell, :1302-1366 (no guard at 30 S) */
        cell_rows_u(c, k, j_had_s, j_had_s, ua_30, trop);
        cell_rows_low(c, V, k, j_had_s, j_had_s, va_Hadley_Tropopause, va_Hadley_SL, trop);
        cell_rows_low(c, W, k, j_had_s, j_had_s, wa_Hadley_Tropopause, wa_Hadley_SL, trop);
        cell_rows_high(c, V, k, j_had_s, j_had_s, va_Hadley_Tropopause, 0, trop);
        cell_rows_high(c, W, k, j_had_s, j_had_s, wa_Hadley_Tropopause, 0, trop);
        cell_rows_low(c, V, k, j_had_v_s, j_had_v_s, va_Hadley_Tropopause_15, va_Hadley_SL_15, trop);
        cell_rows_high(c, V, k, j_had_v_s, j_had_v_s, va_Hadley_Tropopause_15, 1, trop);
        /* southern Ferrel cell (no guard at 60 S) */
        cell_rows_u(c, k, j_fer_s, j_fer_s, ua_60, trop);
        cell_rows_low(c, V, k, j_fer_s, j_fer_s, va_Ferrel_Tropopause, va_Ferrel_SL, trop);
        cell_rows_low(c, W, k, j_fer_s, j_fer_s, wa_Ferrel_Tropopause, wa_Ferrel_SL, trop);
        cell_rows_high(c, V, k, j_fer_s, j_fer_s, va_Ferrel_Tropopause, 0, trop);
        cell_rows_high(c, W, k, j_fer_s, j_fer_s, wa_Ferrel_Tropopause, 0, trop);
        cell_rows_low(c, V, k, j_fer_v_s, j_fer_v_s, va_Ferrel_Tropopause_45, va_Ferrel_SL_45, trop);
        cell_rows_high(c, V, k, j_fer_v_s, j_fer_v_s, va_Ferrel_Tropopause_45, 1, trop);
        /* southern polar cell (no guard between 60 S and the pole) */
        cell_rows_u(c, k, j_pol_s, jm - 1, ua_90, trop);
        cell_rows_low(c, V, k, j_fer_s + 1, j_pol_s, va_Polar_Tropopause, va_Polar_SL, trop);
        cell_rows_low(c, W, k, j_fer_s + 1, j_pol_s, wa_Polar_Tropopause, wa_Polar_SL, trop);
        cell_rows_high(c, V, k, j_fer_s + 1, j_pol_s, va_Polar_Tropopause, 0, trop);
        cell_rows_high(c, W, k, j_fer_s + 1, j_pol_s, wa_Polar_Tropopause, 0, trop);
        cell_rows_low(c, V, k, j_pol_v_s, j_pol_v_s, va_Polar_Tropopause_75, va_Polar_SL_75, trop);
        cell_rows_high(c, V, k, j_pol_v_s, j_pol_v_s, va_Polar_Tropopause_75, 1, trop);
        /* linear connections between the rows, :1468-1700, in the reference's order.  u and w of one loop are independent. */
        cell_interpolate(c, ATOMDEV_U, k, j_pol_n, j_fer_n, 1, j_fer_n, j_pol_n);
        cell_interpolate(c, W, k, j_pol_n, j_fer_n, 1, j_fer_n, j_pol_n);
        cell_interpolate(c, V, k, j_pol_n, j_pol_v_n, 1, j_pol_v_n, j_pol_n);
        cell_interpolate(c, V, k, j_pol_v_n, j_fer_n, 1, j_fer_n, j_pol_v_n);
        cell_interpolate(c, ATOMDEV_U, k, j_fer_n + 1, j_had_n, 1, j_had_n, j_fer_n);
        cell_interpolate(c, W, k, j_fer_n + 1, j_had_n, 1, j_had_n, j_fer_n);
        cell_interpolate(c, V, k, j_fer_n, j_fer_v_n, 1, j_fer_v_n, j_fer_n);
        cell_interpolate(c, V, k, j_fer_v_n, j_had_n, 1, j_had_n, j_fer_v_n);
        cell_interpolate(c, V, k, j_fer_v_s, j_had_s, -1, j_fer_v_s, j_had_s);
        cell_interpolate(c, ATOMDEV_U, k, j_had_n, j_aeq, 1, j_aeq, j_had_n);
        cell_interpolate(c, W, k, j_had_n, j_aeq, 1, j_aeq, j_had_n);
        cell_interpolate(c, V, k, j_had_v_n, j_aeq, 1, j_aeq, j_had_v_n);
        cell_interpolate(c, V, k, j_had_n, j_had_v_n, 1, j_had_v_n, j_had_n);
        cell_interpolate(c, ATOMDEV_U, k, j_fer_s, j_pol_s, 1, j_fer_s, j_pol_s);
        cell_interpolate(c, W, k, j_fer_s, j_pol_s, 1, j_fer_s, j_pol_s);
        cell_interpolate(c, V, k, j_pol_s, j_pol_v_s, -1, j_pol_s, j_pol_v_s);
        cell_interpolate(c, V, k, j_pol_v_s, j_fer_s, -1, j_pol_v_s, j_fer_s);
        cell_interpolate(c, ATOMDEV_U, k, j_had_s, j_fer_s, 1, j_had_s, j_fer_s);
        cell_interpolate(c, W, k, j_had_s, j_fer_s, 1, j_had_s, j_fer_s);
        cell_interpolate(c, V, k, j_fer_s, j_fer_v_s, -1, j_fer_s, j_fer_v_s);
        cell_interpolate(c, ATOMDEV_U, k, j_aeq, j_had_s, 1, j_aeq, j_had_s);
        cell_interpolate(c, W, k, j_aeq, j_had_s, 1, j_aeq, j_had_s);
        cell_interpolate(c, V, k, j_had_s, j_had_v_s, -1, j_had_s, j_had_v_s);
        cell_interpolate(c, V, k, j_had_v_s, j_aeq, -1, j_had_v_s, j_aeq);
        /* change in sign of the v-component in the southern hemisphere */
        for (int i = 0; i < im; i++)
            for (int j = j_aeq + 1; j < jm; j++) A3(c, V, i, j, k) = -A3(c, V, i, j, k);
        /* smoothing around the cell borders, :2144-2283 */
        cell_smooth(c, k, j_had_n); cell_smooth(c, k, j_fer_n); cell_smooth(c, k, j_had_v_n); cell_smooth(c, k, j_fer_v_n);
        cell_smooth(c, k, j_aeq); cell_smooth(c, k, j_had_s); cell_smooth(c, k, j_fer_s); cell_smooth(c, k, j_had_v_s);
        cell_smooth(c, k, j_fer_v_s);
        /* :2288-2303 */
        for (int i = 0; i < im; i++)
            for (int j = 0; j < jm; j++) {
                A3(c, ATOMDEV_U, i, j, k) = A3(c, ATOMDEV_U, i, j, k) / c->p.u_0;
                A3(c, V, i, j, k) = A3(c, V, i, j, k) / c->p.u_0;
                A3(c, W, i, j, k) = A3(c, W, i, j, k) / c->p.u_0;
                if (is_land(c, i, j, k)) A3(c, ATOMDEV_U, i, j, k) = A3(c, V, i, j, k) = A3(c, W, i, j, k) = 0.;
            }
    }
    return 0;
}

/* BC_Thermo::BC_Surface_Temperature_NASA, atmosphere/BC_Thermo.cpp:2310-2350, for files whose entries are all numbers */
int oracle_atm_read_surface_temperature(oracle_ctx* c, const char* path) {
    FILE* f = fopen(path, "r");
    if (!f) return 1;
    const int jm = c->jm, km = c->km, k_half = (km - 1) / 2;
    for (int k = 0; k < km && !feof(f); k++)
        for (int j = 0; j < jm; j++) {
            double lat, lon, temperature = 0.;
            if (fscanf(f, "%lf %lf %lf", &lat, &lon, &temperature) != 3) { /* the stream has failed: the value stays */ }
            A3(c, ATOMDEV_T, 0, j, k) = A2(c, ATOMDEV_TEMPERATURE_NASA, j, k) = (temperature + c->p.t_0) / c->p.t_0;
        }
    fclose(f);
    for (int j = 0; j < jm; j++) {
        A3(c, ATOMDEV_T, 0, j, k_half) = (A3(c, ATOMDEV_T, 0, j, k_half + 1) + A3(c, ATOMDEV_T, 0, j, k_half - 1)) / 2.;
        A2(c, ATOMDEV_TEMPERATURE_NASA, j, k_half) = (A2(c, ATOMDEV_TEMPERATURE_NASA, j, k_half + 1) + A2(c, ATOMDEV_TEMPERATURE_NASA, j, k_half - 1)) / 2.;
    }
    return 0;
}

/* BC_Thermo::BC_Surface_Precipitation_NASA, atmosphere/BC_Thermo.cpp:2352-2385 */
int oracle_atm_read_surface_precipitation(oracle_ctx* c, const char* path) {
    FILE* f = fopen(path, "r");
    if (!f) return 1;
    for (int k = 0; k < c->km && !feof(f); k++)
        for (int j = 0; j < c->jm; j++) {
            double lat, lon, precipitation = 0.;
            if (fscanf(f, "%lf %lf %lf", &lat, &lon, &precipitation) != 3) { }
            A2(c, ATOMDEV_PRECIPITATION_NASA, j, k) = precipitation;
        }
    fclose(f);
    return 0;
}

/* BC_Thermo::GetPoleTemperature, atmosphere/BC_Thermo.cpp:3305-3332, with the table of :544-561 */
static double pole_temperature(int Ma) {
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

/* BC_Thermo::BC_Temperature, atmosphere/BC_Thermo.cpp:427-676 */
void oracle_atm_bc_temperature(oracle_ctx* c, const atomdev_atm_start* s) {
    const int im = c->im, jm = c->jm, km = c->km, Ma = s->Ma;
    const double t_0 = c->p.t_0, t_cretaceous_add = s->t_cretaceous_add;
    if (Ma > 0 && s->sun == 1) {   /* :485-536, temperature distribution at a prescribed sun position */
        int j_par = s->sun_position_lat;
        j_par = j_par + s->declination;
        const int j_pol = jm - 1;
        const double j_par_f = (double)j_par, j_pol_f = (double)j_pol;
        double aa = (s->t_equator - s->t_pole) / (((j_par_f * j_par_f) - (j_pol_f * j_pol_f)) - 2. * j_par_f * (j_par_f - j_pol_f));
        double bb = -2. * aa * j_par_f;
        double cc = s->t_equator + aa * j_par_f * j_par_f;
        const double j_d = sqrt((cc - s->t_pole) / aa);
        const double dd = 2. * aa * j_d + bb;
        const double e = s->t_pole;
        for (int k = 0; k < km; k++)
            for (int j = 0; j < jm; j++) {
                const double d_j = (double)j;
                if (d_j <= j_d) A3(c, ATOMDEV_T, 0, j, k) = dd * d_j + e + t_cretaceous_add;
                if (d_j > j_d) A3(c, ATOMDEV_T, 0, j, k) = aa * d_j * d_j + bb * d_j + cc + t_cretaceous_add;
            }
        const int k_par = s->sun_position_lon, k_pol = km - 1;
        const double t_360 = (t_0 + 5.) / t_0;
        for (int j = 0; j < jm; j++) {
            const double jm_temp_asym = A3(c, ATOMDEV_T, 0, j, 20);
            for (int k = 0; k < km; k++) {
                const double k_par_f = (double)k_par, k_pol_f = (double)k_pol, d_k = (double)k;
                aa = (jm_temp_asym - t_360) / (((k_par_f * k_par_f) - (k_pol_f * k_pol_f)) - 2. * k_par_f * (k_par_f - k_pol_f));
                bb = -2. * aa * k_par_f;
                cc = jm_temp_asym + aa * k_par_f * k_par_f;
                A3(c, ATOMDEV_T, 0, j, k) = aa * d_k * d_k + bb * d_k + cc;
            }
        }
    }
    const double d_j_half = (double)(jm - 1) / 2.0;
    if (s->RadiationModel == 1) {   /* :568-633 */
        double t_pole_diff_ocean = 0.;
        if (!s->first_time_slice) t_pole_diff_ocean = pole_temperature(Ma) - pole_temperature(s->Ma_previous);
        const double t_pole_diff_land = pole_temperature(Ma);
        const double t_eff = s->t_pole - s->t_equator;
        for (int k = 0; k < km; k++)
            for (int j = 0; j < jm; j++) {
                const int i_mount = topography_level(c, j, k);
                const double d_j = (double)j;
                if (s->NASATemperature == 0) {
                    A3(c, ATOMDEV_T, i_mount, j, k) = t_eff * parabola(d_j / d_j_half) + s->t_pole + t_cretaceous_add;
                    if (is_land(c, 0, j, k))
                        A3(c, ATOMDEV_T, i_mount, j, k) = t_eff * parabola(d_j / d_j_half) + s->t_pole + t_cretaceous_add + s->t_land;
                } else {
                    if (is_land(c, 0, j, k)) {
                        A3(c, ATOMDEV_T, i_mount, j, k) = t_eff * parabola(d_j / d_j_half) + s->t_pole + t_cretaceous_add + s->t_land;
                        A3(c, ATOMDEV_T, i_mount, j, k) += t_pole_diff_land * fabs(parabola(d_j / d_j_half) + 1.) / t_0;
                    } else if (Ma > 0) {
                        A3(c, ATOMDEV_T, 0, j, k) += t_cretaceous_add + t_pole_diff_ocean * fabs(parabola(d_j / d_j_half) + 1.) / t_0;
                    }
                }
            }
    }
    /* :635-664, up to the tropopause and above */
    double t_tropopause_pole = -4.;
    t_tropopause_pole = c->p.t_tropopause + t_tropopause_pole / t_0;
    const double t_eff_tropo = t_tropopause_pole - c->p.t_tropopause;
    for (int j = 0; j < jm; j++) {
        const int i_trop = s->im_tropopause[j] + tropopause_add(s->t_cretaceous, t_0);
        const double temp_tropopause = t_eff_tropo * parabola(j / ((jm - 1) / 2.0)) + t_tropopause_pole + t_cretaceous_add;
        for (int k = 0; k < km; k++) {
            const int i_mount = topography_level(c, j, k);
            double tmp = A3(c, ATOMDEV_T, i_mount, j, k);
            tmp = (temp_tropopause - tmp) * ((double)i_mount / i_trop) + tmp;
            for (int i = 0; i < im; i++) {
                if (i < i_trop + 1) {
                    if (i > i_mount) A3(c, ATOMDEV_T, i, j, k) = (temp_tropopause - tmp) * ((double)(i - i_mount) / i_trop) + tmp;
                    else A3(c, ATOMDEV_T, i, j, k) = tmp;
                } else {
                    A3(c, ATOMDEV_T, i, j, k) = temp_tropopause;
                }
            }
        }
    }
}

/* BC_Thermo::IC_Temperature_WestEastCoast, atmosphere/BC_Thermo.cpp:3151-3233 */
void oracle_atm_ic_temperature_west_east_coast(oracle_ctx* c) {
    const int jm = c->jm, km = c->km, j_grad = 7, k_grad = 7;
#define T0_(j, k) A3(c, ATOMDEV_T, 0, j, k)
    int j_air = 0, flip = 0, ll;
    for (int k = 1; k < km - 1; k++) {   /* north coasts */
        for (int j = j_grad; j < jm - 1; j++) {
            if (is_fluid(c, 0, j, k)) { j_air = 0; flip = 0; }
            else j_air = 1;
            if (flip == 0 && j_air == 1) {
                ll = j - j_grad;
                for (int l = ll; l < j; l++) T0_(l, k) = T0_(ll, k);
                flip = 1;
            }
        }
        flip = 0;
    }
    j_air = 0;
    int j_sequel = 1;
    for (int k = 1; k < km - 1; k++) {   /* south coasts */
        for (int j = 0; j < jm - j_grad; j++) {
            if (is_land(c, 0, j, k)) j_sequel = 0;
            if (is_fluid(c, 0, j, k) && j_sequel == 0) j_air = 0;
            else j_air = 1;
            if (is_fluid(c, 0, j, k) && j_air == 0) {
                ll = j + j_grad;
                for (int l = ll; l > j; l--) T0_(l, k) = T0_(ll, k);
                j_sequel = 1;
            }
        }
        j_air = 0;
    }
    int k_air = 0, k_sequel = 1;
    for (int j = 1; j < jm - 1; j++) {   /* east coasts */
        for (int k = k_grad; k < km - k_grad; k++) {
            if (is_land(c, 0, j, k)) k_sequel = 0;
            if (is_fluid(c, 0, j, k) && k_sequel == 0) k_air = 0;
            else k_air = 1;
            if (is_fluid(c, 0, j, k) && k_air == 0) {
                ll = k + k_grad;
                for (int l = ll; l > k; l--) T0_(j, l) = T0_(j, ll);
                k_sequel = 1;
            }
        }
        k_air = 0;
    }
    k_air = 0; flip = 0;
    for (int j = 1; j < jm - 1; j++) {   /* west coasts */
        for (int k = k_grad; k < km - 1; k++) {
            if (is_fluid(c, 0, j, k)) { k_air = 0; flip = 0; }
            else k_air = 1;
            if (flip == 0 && k_air == 1) {
                ll = k - k_grad;
                for (int l = ll; l < k; l++) T0_(j, l) = T0_(j, ll);
                flip = 1;
            }
        }
        flip = 0;
    }
#undef T0_
}

/* BC_Thermo::BC_Pressure, atmosphere/BC_Thermo.cpp:2387-2407 */
void oracle_atm_bc_pressure(oracle_ctx* c) {
    const atomdev_params* P = &c->p;
    const int im = c->im, jm = c->jm, km = c->km;
    const double exp_pressure = P->g / (1.e-2 * P->gam * P->R_Air);
    for (int k = 0; k < km; k++)
        for (int j = 0; j < jm; j++) A3(c, ATOMDEV_P_STAT, 0, j, k) = .01 * (P->r_air * P->R_Air * A3(c, ATOMDEV_T, 0, j, k) * P->t_0);
    for (int k = 0; k < km; k++)
        for (int j = 0; j < jm; j++)
            for (int i = 1; i < im; i++) {
                const double hight = (double)i * (P->L_atm / (double)(im - 1));
                A3(c, ATOMDEV_P_STAT, i, j, k) = pow(((A3(c, ATOMDEV_T, 0, j, k) * P->t_0 - P->gam * hight * 1.e-2) /
                                                      (A3(c, ATOMDEV_T, 0, j, k) * P->t_0)), exp_pressure) * A3(c, ATOMDEV_P_STAT, 0, j, k);
            }
}

/* BC_Thermo::BC_WaterVapour, atmosphere/BC_Thermo.cpp:679-780.  Dalton_Evaporation is computed over the ocean only; a land
 * column adds what the ocean column visited before it (k outer, j inner) left in that variable (:744). */
void oracle_atm_bc_water_vapour(oracle_ctx* c, const atomdev_atm_start* s) {
    const atomdev_params* P = &c->p;
    const int im = c->im, jm = c->jm, km = c->km;
    const double t_0 = P->t_0, hp = P->hp, ep = P->ep, r_air = P->r_air, R_Air = P->R_Air;
    const double dt_dim = P->L_atm / P->u_0 * P->dt, dr_dim = P->dr * P->L_atm;   /* :121-123 */
    double Dalton_Evaporation = 0;
#define C_(i) A3(c, ATOMDEV_C, i, j, k)
#define T_(i) A3(c, ATOMDEV_T, i, j, k)
    for (int k = 0; k < km; k++)
        for (int j = 0; j < jm; j++) {
            const int i_mount = topography_level(c, j, k);
            if (is_fluid(c, 0, j, k)) {
                C_(i_mount) = hp * ep * exp(17.0809 * (T_(i_mount) * t_0 - t_0) / (234.175 + (T_(i_mount) * t_0 - t_0))) /
                              ((r_air * R_Air * T_(i_mount) * t_0) * .01);
                C_(i_mount) = s->c_ocean * C_(i_mount);
                A3(c, ATOMDEV_P_STAT, i_mount, j, k) = (r_air * R_Air * T_(i_mount) * t_0) * .01;
                const double p_stat = A3(c, ATOMDEV_P_STAT, i_mount, j, k);
                const double t_u = T_(i_mount) * t_0;
                const double r_dry = 100. * p_stat / (R_Air * T_(i_mount) * t_0);
                const double r_humid = r_dry / (1. + (P->R_WaterVapour / R_Air - 1.) * C_(i_mount));
                const double e = C_(i_mount) * p_stat / ep;
                const double E = hp * exp_func(t_u, 17.2694, 35.86);
                const double sat_difference = (E - e);
                Dalton_Evaporation = 8.46e-4 * C_Dalton(P->u_0, A3(c, ATOMDEV_V, i_mount, j, k), A3(c, ATOMDEV_W, i_mount, j, k)) *
                                     sat_difference * dt_dim / (r_humid * dr_dim) * 24.;
                C_(i_mount) = Dalton_Evaporation + C_(i_mount);
            }
            if (is_land(c, 0, j, k)) {
                C_(i_mount) = hp * ep * exp(17.0809 * (T_(i_mount) * t_0 - t_0) / (234.175 + (T_(i_mount) * t_0 - t_0))) /
                              ((r_air * R_Air * T_(i_mount) * t_0) * .01);
                C_(i_mount) = s->c_land * C_(i_mount);
                C_(i_mount) = Dalton_Evaporation + C_(i_mount);
            }
        }
    for (int j = 0; j < jm; j++) {
        const int i_trop = s->im_tropopause[j] + tropopause_add(s->t_cretaceous, t_0);
        const double d_i_max = (double)i_trop;
        for (int k = 0; k < km; k++) {
            const int i_mount = topography_level(c, j, k);
            for (int i = 0; i < im; i++) {
                if (i < i_trop) {
                    const double d_i = (double)i;
                    C_(i) = C_(i_mount) - (s->c_tropopause - C_(i_mount)) * (d_i / d_i_max * (d_i / d_i_max - 2.));
                } else {
                    C_(i) = s->c_tropopause;
                }
            }
        }
    }
#undef C_
#undef T_
}

/* BC_Thermo::BC_CO2, atmosphere/BC_Thermo.cpp:788-881 */
void oracle_atm_bc_co2(oracle_ctx* c, const atomdev_atm_start* s) {
    const int im = c->im, jm = c->jm, km = c->km;
    const double co2_0 = c->p.co2_0, t_0 = c->p.t_0;
    const int j_max = jm - 1, j_half = j_max / 2;
    double co2_cretaceous = 3.2886 * pow((s->t_cretaceous + s->t_average), 2) - 32.8859 * (s->t_cretaceous + s->t_average) + 102.2148;
    const double co2_average = 3.2886 * pow(s->t_average, 2) - 32.8859 * s->t_average + 102.2148;
    co2_cretaceous = co2_cretaceous - co2_average;
    const double d_j_half = (double)j_half;
    const double co2_equator = s->co2_equator / co2_0, co2_pole = s->co2_pole / co2_0;
    co2_cretaceous = co2_cretaceous / co2_0;
    const double co2_land = s->co2_land / co2_0, co2_ocean = s->co2_ocean / co2_0, co2_vegetation = s->co2_vegetation / co2_0,
                 co2_tropopause = s->co2_tropopause / co2_0;
    const double co2_eff = co2_pole - co2_equator;
#define CO2_(i) A3(c, ATOMDEV_CO2, i, j, k)
    for (int k = 0; k < km; k++)
        for (int j = 0; j < jm; j++) {
            const int i_mount = topography_level(c, j, k);
            if (is_fluid(c, i_mount, j, k)) {
                const double d_j = (double)j;
                CO2_(i_mount) = co2_eff * (d_j * d_j / (d_j_half * d_j_half) - 2. * d_j / d_j_half) + co2_pole + co2_cretaceous + co2_ocean;
            }
            if (is_land(c, i_mount, j, k)) {
                const double d_j = (double)j;
                CO2_(i_mount) = co2_eff * (d_j * d_j / (d_j_half * d_j_half) - 2. * d_j / d_j_half) + co2_pole + co2_cretaceous + co2_land -
                                co2_vegetation * A2(c, ATOMDEV_VEGETATION, j, k);
            }
        }
    for (int j = 0; j < jm; j++) {
        const int i_trop = s->im_tropopause[j] + tropopause_add(s->t_cretaceous, t_0);
        const double d_i_max = (double)i_trop;
        for (int k = 0; k < km; k++) {
            const int i_mount = topography_level(c, j, k);
            for (int i = 1; i <= im - 1; i++) {
                if (i <= i_trop) {
                    const double d_i = (double)i;
                    CO2_(i) = CO2_(i_mount) - (co2_tropopause - CO2_(i_mount)) * (d_i / d_i_max * (d_i / d_i_max - 2.));
                } else {
                    CO2_(i) = co2_tropopause;
                }
            }
            for (int i = i_trop - 1; i >= 0; i--)
                if (is_land(c, i, j, k) != (is_land(c, i, j, k) && is_fluid(c, i + 1, j, k))) CO2_(i) = CO2_(i_mount);
        }
    }
#undef CO2_
}

/* cAtmosphereModel::restrain_temperature with calculate_mean_temperature and calculate_node_weights
 * (atmosphere/cAtmosphereModel.cpp:952-1006).  mean_temperature_curve: get_mean_temperature_from_curve(Ma), the float
 * the reference gets, widened.  The weighted mean passes through float like the reference's return value. */
void oracle_atm_restrain_temperature(oracle_ctx* c, double mean_temperature_curve) {
    const int jm = c->jm, km = c->km;
    const double t_0 = c->p.t_0;
    double ret = 0., weight = 0.;
    for (int j = 0; j < jm; j++) {
        const double w = j <= 90 ? cos((90 - j) * M_PI / 180.0) : cos((j - 90) * M_PI / 180.0);
        for (int k = 0; k < km; k++) {
            ret += A3(c, ATOMDEV_T, 0, j, k) * w;
            weight += w;
        }
    }
    const double tmp_1 = mean_temperature_curve;
    const double tmp_2 = (float)((ret / weight - 1) * t_0);
    const double diff = tmp_2 - tmp_1;
    for (int j = 0; j < jm; j++)
        for (int k = 0; k < km; k++) {
            A3(c, ATOMDEV_T, 0, j, k) -= diff / t_0;
            if (A3(c, ATOMDEV_T, 0, j, k) > (1 + 40 / t_0)) A3(c, ATOMDEV_T, 0, j, k) = 1 + 40 / t_0;
        }
}

/* PostProcess_Atmosphere::Atmosphere_v_w_Transfer (atmosphere/PostProcess_Atm.cpp:98-118): "%.4f" is what an ofstream
 * with precision(4) and ios::fixed prints */
int oracle_atm_write_transfer_file(oracle_ctx* c, const char* path) {
    FILE* f = fopen(path, "w");
    if (!f) return 1;
    for (int j = 0; j < c->jm; j++)
        for (int k = 0; k < c->km; k++)
            fprintf(f, "%.4f %.4f %.4f %.4f %.4f %.4f\n", A3(c, ATOMDEV_V, 0, j, k) * c->p.u_0, A3(c, ATOMDEV_W, 0, j, k) * c->p.u_0,
                    A3(c, ATOMDEV_T, 0, j, k), A3(c, ATOMDEV_P_DYN, 0, j, k), A2(c, ATOMDEV_EVAPORATION_DALTON, j, k),
                    A2(c, ATOMDEV_PRECIPITATION, j, k));
    return fclose(f) != 0;
}
