// Atmosphere 3-D Runge-Kutta sweep, pipelined form (variant 3; tolerance form only).
//
// Reference: atmosphere/RungeKutta_Atm.cpp:32-156 with RHS_Atm.cpp:68-493 at every stage.  The arithmetic is the
// affine restatement of k_rk_atm_ws.hip's tolerance form; what changes is who does which part of it, and when.
//
// Why.  A theta line is a recurrence of jm-2 rows, and in the wave-specialised form a row costs one memory latency
// (its loads are issued one row ahead) or four publish -> barrier -> read round trips between the eight waves that
// share the row's stage values, whichever is longer: about a microsecond either way, 179 times in a row.  Neither
// is necessary.  With the neighbour rows fixed during the four stages of a row, the right-hand side of a variable
// x is affine in its stage value apart from the advection product with v,
//     rhs = -v (cA x + cB) + cP x + cQ      (+ cR u for v, - KK ((1 + c) / (1 + k c) - 1) for u, evaporation for c on level 1)
// and cB, cQ are themselves affine in the results of the two rows before,
//     cB = H1 x[j-1] + H2 x[j-2] + KB        cQ = q1 x[j-1] + q2 x[j-2] + KQ
// where cA, cP, H1, H2, q1, q2, KB, KQ depend on the mask, the geometry, the source fields and OLD field values
// only.  So everything that needs memory or many instructions can be worked out rows ahead by waves that do
// nothing else, and what remains on the recurrence is a handful of fused multiply-adds per variable and stage,
// short enough for ONE wave to carry four variables through a row in a third of a microsecond.
//
// Mapping.  A workgroup owns 64 neighbouring theta lines of one level and has six live waves:
//   four "prep" waves, taking the rows in turn (row r belongs to wave r mod 4): each loads everything its row
//       needs (old values, base values, sources, flags, four rows of p_dyn), turns it into the row's coefficient
//       packet in a ring of four LDS slots, and issues the loads of its next row, four rows on, before it rests -
//       so a load has three rows' time to arrive and nobody ever waits for memory on the recurrence;
//   chain wave A carries v, u, c, t (the coupled ones: v feeds all, u feeds v, c feeds u, t feeds c on level 1)
//       through the four stages of row j, leaves the four stage values of v in LDS, and stores its results;
//   chain wave B carries w, cloud, ice, co2 one row behind A, with the stage values of v that A left.
// One LDS barrier per row orders all of it: in epoch e the prep wave of row e+1 writes its packet, A works on row e,
// B on row e-1.  The chain waves issue no loads inside the loop and the prep waves no stores, so there is no
// vector-memory wait to place by hand anywhere.  Waves 4 and 5 of the 512-thread workgroup end at once: waves w and
// w+4 share a SIMD, and the two chain waves are to have theirs to themselves.
#include "atomdev_internal.h"

namespace {

enum Var { V_T = 0, V_U = 1, V_V = 2, V_W = 3, V_C = 4, V_CLOUD = 5, V_ICE = 6, V_CO2 = 7 };
// diffusion divisor class of a variable (re*pr, re, re*sc_WaterVapour, re*sc_CO2) and its slot among the cP columns
__device__ constexpr int q_class(int var) { return var == V_T ? 0 : (var == V_U || var == V_V || var == V_W) ? 1 : var == V_CO2 ? 3 : 2; }
__device__ constexpr int cp_slot(int var) { return var == V_T ? 0 : var == V_U ? 1 : var == V_V ? 2 : var == V_W ? 3 : var == V_C ? 4 : var == V_CO2 ? 6 : 5; }

struct PipeArgs {
    double* x[8];
    const double* xn[8];
    const double* p_dyn;
    const double *S_v, *S_c, *S_i, *S_r, *S_s;
    double *aux_u, *aux_v, *aux_w, *buoy;
    const uint8_t* flags;
    const double *evap, *precip, *c3;          // 2-D
    const double *ph100, *h0i;                 // 3-D, from the preparing kernel of k_rk_atm_ws.hip
    const double *rad, *costhe, *tab_irm2sin, *tab_coefv;
};
struct PipeParams {
    PipeArgs a;
    DevConst K;
};

// One row's packet: 64 lanes wide, everything a chain wave needs of the row besides its own last two results.  The
// first half is written by the prep wave that works for chain B (and holds what both chains read), the second by the
// one that works for chain A.
enum { QA_RE = 0, QA_WV = 1, QA_PR = 2 };     // diffusion classes of chain A's variables: u and v, c, t
enum { QB_RE = 0, QB_WV = 1, QB_CO2 = 2 };    // ... and of chain B's: w, cloud and ice, co2
struct Slot {
    double cA[64], H1[64], H2[64];
    double q1B[3][64], q2B[3][64];
    double cPB[3][64];                         // w, cloud = ice, co2
    double x0B[4][64], xnB[4][64], KQB[4][64], KBB[4][64];   // w, cloud, ice, co2 (KB, and KQ of w and co2: pole-side rows only)
    unsigned fl[64];
    double q1A[3][64], q2A[3][64];
    double cPA[4][64];                         // v, u, c, t
    double x0A[4][64], xnA[4][64], KQA[4][64], KBA[4][64];   // v, u, c, t (KB, and KQ of u: pole-side rows only)
    double cR[64], qu1[64], qu2[64], pgv[64];
    double cFac[64], cMc[64], phk[64];         // level 1: the evaporation source of c
};
constexpr int NSLOT = 4;
struct PipeShared {
    Slot ring[NSLOT];
    double vst[2][4][64];      // stage values of v of a row, by row parity
#ifdef ATOMDEV_EXP_PIPE_STAMP
    unsigned long long stamps[10][4][8];
    unsigned long long wall[2], cyc[2];
#endif
};

#ifdef ATOMDEV_EXP_PIPE_STAMP  // timing experiment only: cycle stamps of four epochs of one workgroup, kept in LDS
#define PSTAMP(wave, n)                                                                                       \
    do {                                                                                                      \
        if (e >= 60 && e < 64 && lane == 0) sh.stamps[wave][e - 60][n] = __builtin_readcyclecounter();        \
    } while (0)
#else
#define PSTAMP(wave, n) do {} while (0)
#endif

typedef const __attribute__((address_space(1))) char* gcchar_t;
typedef __attribute__((address_space(1))) char* gchar_t;
// Field element at byte offset `off` from a (wave-uniform) base pointer: the base stays in scalar registers and the
// offset is one 32-bit vector register shared by all fields of the wave.
__device__ __forceinline__ double gld(const double* base, unsigned off) {
    return *(const __attribute__((address_space(1))) double*)((gcchar_t)base + off);
}
__device__ __forceinline__ void gst(double* base, unsigned off, double x) {
    *(__attribute__((address_space(1))) double*)((gchar_t)base + off) = x;
}
#define PIN(x) asm volatile("" : "+v"(x))
// A constant of the argument block as a value in scalar registers: without this the compiler may turn a per-lane choice
// between two such constants into a per-lane choice of ADDRESS and a vector load in the middle of the loop.
#define SPIN(x) asm volatile("" : "+s"(x))

__device__ __forceinline__ double rcp2(double b) {  // hardware estimate and two Newton steps: full precision
    double y = __builtin_amdgcn_rcp(b);
    double e = __builtin_fma(-b, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-b, y, 1.0);
    return __builtin_fma(y, e, y);
}

// BC_Thermo::Value_Limitation_Atm (BC_Thermo.cpp:3235-3275) for one variable, the comparisons in its order
template <int VAR>
__device__ __forceinline__ double limited(double x, bool land) {
    if constexpr (VAR == V_U) { if (x >= .106) x = .106; if (x <= -.106) x = -.106; if (land) x = 0.; }
    if constexpr (VAR == V_V) { if (x >= .125) x = .125; if (x <= -.125) x = -.125; if (land) x = 0.; }
    if constexpr (VAR == V_W) { if (x >= 3.5) x = 3.5; if (x <= -.469) x = -.469; if (land) x = 0.; }
    if constexpr (VAR == V_T) { if (x >= 1.165) x = 1.165; if (x <= -.78) x = -.78; }
    if constexpr (VAR == V_C) { if (x >= .03) x = .03; if (x < 0.) x = 0.; }
    if constexpr (VAR == V_CLOUD) { if (x >= .01) x = .01; if (x < 0.) x = 0.; if (land) x = 0.; }
    if constexpr (VAR == V_ICE) { if (x >= .005) x = .005; if (x < 0.) x = 0.; if (land) x = 0.; }
    if constexpr (VAR == V_CO2) { if (x >= 5.36) x = 5.36; if (x <= 1.) x = 1.; if (land) x = 1.; }
    return x;
}

// ---- a prep wave: rows r = q, q + 4, ... (r >= 1), the half of the packet that chain A (FORA) or chain B reads ----
template <bool L1, bool FORA>
__device__ __forceinline__ void prep_wave(const PipeArgs& a, const DevConst& K, PipeShared& sh, const int q, const int i, const int bx) {
    const int jm = K.jm, km = K.km;
    const int lane = threadIdx.x & 63;
    const int kk = 1 + bx * 64 + lane;
    const int k = kk <= km - 2 ? kk : km - 2;  // idle lanes shadow the last line
    const int NR = jm - 2, NE = NR + 2;
    const unsigned rowB = (unsigned)km * 8u;
    unsigned o_lev = (unsigned)(((size_t)i * jm * km + k) * 8);   // row 0 of the line, as a byte offset into a 3-D field
    unsigned o_sfc = (unsigned)k * 8u;                             // ... into a 2-D field
    PIN(o_lev); PIN(o_sfc);
    Slot& S = sh.ring[q];
    constexpr int VARS[4] = { FORA ? V_V : V_W, FORA ? V_U : V_CLOUD, FORA ? V_C : V_ICE, FORA ? V_T : V_CO2 };
    const int stamp_wave = 2 + 2 * q + (FORA ? 0 : 1);
    (void)stamp_wave;

    const double rm = a.rad[i];
    const double i_rm = 1. / rm, i_rm2 = 1. / (rm * rm);
    const double two = 2. * i_rm2 * K.i_re;
    const double ce = K.coeff_energy, ctr = K.coeff_trans;
    const double i_phk = 1. / (K.R_Air * K.t_0);
    const double evf = K.coeff_vapour * K.i_two_dr;
    double i_dthe = K.i_dthe, i_two_dthe = K.i_two_dthe, i_dthe2 = K.i_dthe2, i_dr2 = K.i_dr2, i_dphi2 = K.i_dphi2;
    double i_re = K.i_re, i_re_sc_wv = K.i_re_sc_wv, i_re_x = FORA ? K.i_re_pr : K.i_re_sc_co2, i_r_air = K.i_r_air;
    SPIN(i_dthe); SPIN(i_two_dthe); SPIN(i_dthe2); SPIN(i_dr2); SPIN(i_dphi2); SPIN(i_re); SPIN(i_re_sc_wv); SPIN(i_re_x); SPIN(i_r_air);
    // diffusion divisors of this half's classes and variables
    const double idc[3] = { i_re, i_re_sc_wv, i_re_x };
    constexpr int CLS[4] = { 0, FORA ? 0 : 1, 1, 2 };   // A: v, u -> re; c -> wv; t -> pr.  B: w -> re; cloud, ice -> wv; co2

    // what a row's loads leave behind until the row's turn comes
    double lx[4], lxn[4], lxp[4], lS[5], lh0i = 0., lp[4];
    double lev = 0., lpr = 0., lc0 = 0., lc3 = 0., lph = 0.;
    double lcos, lirs, lcoefv = 0.;  // geometry of the row: one address for the whole wave, fetched with the rest of the row
    unsigned lfl;
    auto issue = [&](const int r) {
        const unsigned o = o_lev + (unsigned)r * rowB;
        lfl = *(const __attribute__((address_space(1))) uint8_t*)((gcchar_t)a.flags + (o >> 3));
#pragma unroll
        for (int n = 0; n < 4; n++) { lx[n] = gld(a.x[VARS[n]], o); lxn[n] = gld(a.xn[VARS[n]], o); }
        lcos = gld(a.costhe + r, 0u); lirs = gld(a.tab_irm2sin + (size_t)i * jm + r, 0u);
        lh0i = gld(a.h0i, o);
        if constexpr (FORA) {
            lS[0] = gld(a.S_c, o); lS[1] = gld(a.S_r, o); lS[2] = gld(a.S_i, o); lS[3] = gld(a.S_s, o); lS[4] = gld(a.S_v, o);
            lcoefv = gld(a.tab_coefv + (size_t)i * jm + r, 0u);
            lp[0] = r >= 2 ? gld(a.p_dyn, o - 2u * rowB) : 0.;
            lp[1] = gld(a.p_dyn, o - rowB);
            lp[2] = gld(a.p_dyn, o);
            lp[3] = gld(a.p_dyn, o + rowB);
        } else {
            lS[0] = gld(a.S_c, o); lS[2] = gld(a.S_i, o);
        }
        const bool window = r >= 2 && r < jm - 3;
        if (!window) {
#pragma unroll
            for (int n = 0; n < 4; n++) lxp[n] = gld(a.x[VARS[n]], o + rowB);
        }
        if constexpr (L1 && FORA) {
            const unsigned oq = o_sfc + (unsigned)r * rowB;
            lev = gld(a.evap, oq); lpr = gld(a.precip, oq);
            lc0 = gld(a.x[V_C], oq); lc3 = gld(a.c3, oq);
            lph = gld(a.ph100, o);
        }
    };
    const int first = q == 0 ? 4 : q;
    if (first <= NR) issue(first);

    for (int e = 0; e < NE; e++) {
        const int r = e + 1;
        PSTAMP(stamp_wave, 0);
        if ((r & 3) == q && r <= NR) {
            const bool window = r >= 2 && r < jm - 3;
            const double g = lcos * lirs;
            // the mask: RHS_Atm.cpp:110-149
            const unsigned fl = lfl;
            PSTAMP(stamp_wave, 1);
            const bool land = fl & FL_LAND;
            const double hdj = ((fl & FL_H0J) && (fl & FL_H0K)) ? 0. : 1.;
            // which theta form stands (RHS_Atm.cpp:162-170, 192-199, 245-312), as factors of x, x[j-1], x[j-2], x[j+1]
            const int mode = !window ? 0 : ((fl & FL_ONESIDED_S) ? 2 : 1);
            const double fa = mode == 0 ? 0. : (mode == 1 ? 1. : -3.), fe = mode == 0 ? -2. : (mode == 1 ? 0. : 2.);
            const double b1 = mode == 2 ? 4. : -1., b2 = mode == 2 ? -1. : 0., b3 = mode == 0 ? 1. : 0.;
            const double f1 = mode == 0 ? 1. : (mode == 1 ? 0. : -2.), f2 = mode == 2 ? 1. : 0., f3 = mode == 0 ? 1. : 0.;
            const double hD = hdj * (mode == 1 ? i_dthe : i_two_dthe), hE = hdj * i_dthe2;
            const double A1 = fa * hD, A2 = fe * hE;
            const double Pb = __builtin_fma(A2, i_rm2, A1 * g);
            const double hDr = hD * i_rm, hDg = hD * g, hEr = hE * i_rm2;
            const double G1 = __builtin_fma(f1, hEr, b1 * hDg), G2 = __builtin_fma(f2, hEr, b2 * hDg);
            const double dmp_i = lh0i * i_dr2;
            double KQ[4], KB[4] = { 0., 0., 0., 0. };
            if constexpr (FORA) {
#pragma unroll
                for (int c = 0; c < 3; c++) { S.q1A[c][lane] = G1 * idc[c]; S.q2A[c][lane] = G2 * idc[c]; }
                // factor of the stage value outside the advection product: diffusion bracket and damping
                const double p_re = Pb * i_re;
                const double h0j = (fl & FL_H0J) ? 1. : 0.;
                S.cPA[0][lane] = p_re - lcoefv * hdj * i_re - h0j * i_dthe2;
                S.cPA[1][lane] = p_re - dmp_i;
                S.cPA[2][lane] = Pb * i_re_sc_wv;
                S.cPA[3][lane] = Pb * i_re_x;
                // pressure gradient of the v equation (p_dyn is not changed by the sweep)
                const double dp = hD * (fa * lp[2] + b1 * lp[1] + b2 * lp[0] + b3 * lp[3]);
                const double pgv = hdj * dp * i_rm * i_r_air;
                S.pgv[lane] = pgv;
                const double twoD = two * hD;
                S.cR[lane] = two * A1; S.qu1[lane] = twoD * b1; S.qu2[lane] = twoD * b2;
                // constant terms: sources, and on the pole-side rows the old row j+1
                KQ[0] = -pgv; KQ[1] = 0.; KQ[2] = lS[4] * ctr; KQ[3] = ce * (lS[0] + lS[1]) + ce * (lS[2] + lS[3]);
                if (!window) KQ[0] = __builtin_fma(twoD * b3, lxp[1], KQ[0]);
                if constexpr (L1) {  // RHS_Atm.cpp:412-430
                    double ep = lev - lpr;
                    if (ep >= 6.) ep = 6.;
                    if (ep <= -6.) ep = -6.;
                    S.cFac[lane] = land ? 0. : evf * (1. - 2. * lc0) * ep;
                    S.cMc[lane] = -3. * lc0 - lc3;
                    S.phk[lane] = lph * i_phk;
                }
            } else {
                S.cA[lane] = A1 * i_rm; S.H1[lane] = b1 * hDr; S.H2[lane] = b2 * hDr;
#pragma unroll
                for (int c = 0; c < 3; c++) { S.q1B[c][lane] = G1 * idc[c]; S.q2B[c][lane] = G2 * idc[c]; }
                const double h0k = (fl & FL_H0K) ? 1. : 0.;
                const double p_wv = Pb * i_re_sc_wv;
                S.cPB[0][lane] = __builtin_fma(-h0k, i_dphi2, Pb * i_re);
                S.cPB[1][lane] = p_wv - dmp_i;
                S.cPB[2][lane] = __builtin_fma(Pb, i_re_x, -dmp_i);
                KQ[0] = 0.; KQ[1] = lS[0] * ctr; KQ[2] = lS[2] * ctr; KQ[3] = 0.;
                S.fl[lane] = fl;
            }
            if (!window) {
                const double G3 = __builtin_fma(f3, hEr, b3 * hDg), H3 = b3 * hDr;
#pragma unroll
                for (int n = 0; n < 4; n++) {
                    KB[n] = H3 * lxp[n];
                    KQ[n] = __builtin_fma(G3 * idc[CLS[n]], lxp[n], KQ[n]);
                }
            }
            double (*x0)[64] = FORA ? S.x0A : S.x0B, (*xn)[64] = FORA ? S.xnA : S.xnB;
            double (*kq)[64] = FORA ? S.KQA : S.KQB, (*kb)[64] = FORA ? S.KBA : S.KBB;
#pragma unroll
            for (int n = 0; n < 4; n++) {
                x0[n][lane] = lx[n]; xn[n][lane] = lxn[n];
                const bool zero_in_window = FORA ? n == 1 : (n == 0 || n == 3);   // u; w, co2: no source term
                if (!window || !zero_in_window) kq[n][lane] = KQ[n];
                if (!window) kb[n][lane] = KB[n];
            }
            PSTAMP(stamp_wave, 2);
            if (r + 4 <= NR) issue(r + 4);
            PSTAMP(stamp_wave, 3);
        }
        PSTAMP(stamp_wave, 4);
        lds_barrier();
        PSTAMP(stamp_wave, 5);
    }
}

// One variable's four stages need: the row's shared factors, its own packet entries, and v.
struct RowShared { double cA, H1, H2; };

// ---- chain wave A: v, u, c, t ------------------------------------------------------------------------------------
template <bool L1, bool LIMIT>
__device__ __forceinline__ void chain_a(const PipeArgs& a, const DevConst& K, PipeShared& sh, const int i, const int bx) {
    const int jm = K.jm, km = K.km;
    const int lane = threadIdx.x & 63;
    const int kk = 1 + bx * 64 + lane;
    const int k = kk <= km - 2 ? kk : km - 2;
    const int NR = jm - 2, NE = NR + 2;
    const unsigned rowB = (unsigned)km * 8u;
    unsigned o_lev = (unsigned)(((size_t)i * jm * km + k) * 8);
    PIN(o_lev);
    constexpr int VARS[4] = { V_V, V_U, V_C, V_T };
    constexpr int CLS[4] = { QA_RE, QA_RE, QA_WV, QA_PR };
    enum { cV = 0, cU = 1, cC = 2, cT = 3 };

    const double kWV = K.R_WV_over_R_Air, KK = K.Buoyancy * K.g, hdt = .5 * K.dt, dt = K.dt, dt6 = K.dt * K.i_six;
    const double cb1000 = K.coeff_buoy * 1000.;
    double xm1[4], xm2[4] = { 0., 0., 0., 0. };
#pragma unroll
    for (int n = 0; n < 4; n++) xm1[n] = gld(a.x[VARS[n]], o_lev);

    for (int e = 0; e < NE; e++) {
        PSTAMP(0, 0);
        if (e >= 1 && e <= NR) {
            const int j = e;
            const bool window = j >= 2 && j < jm - 3;
            const Slot& S = sh.ring[j & 3];
            double (*vst)[64] = sh.vst[j & 1];
            const unsigned o = o_lev + (unsigned)j * rowB;
            const bool land = S.fl[lane] & FL_LAND;
            const double cA = S.cA[lane], H1 = S.H1[lane], H2 = S.H2[lane];
            double st[4], xn[4], cB[4], cQ[4], cP[4], acc[4], rhs[4];
#pragma unroll
            for (int n = 0; n < 4; n++) {
                st[n] = S.x0A[n][lane]; xn[n] = S.xnA[n][lane]; cP[n] = S.cPA[n][lane];
                const double kb = window ? 0. : S.KBA[n][lane];
                const double kq = (window && n == cU) ? 0. : S.KQA[n][lane];
                cB[n] = __builtin_fma(H1, xm1[n], __builtin_fma(H2, xm2[n], kb));
                cQ[n] = __builtin_fma(S.q1A[CLS[n]][lane], xm1[n], __builtin_fma(S.q2A[CLS[n]][lane], xm2[n], kq));
            }
            cQ[cV] = __builtin_fma(S.qu1[lane], xm1[cU], __builtin_fma(S.qu2[lane], xm2[cU], cQ[cV]));  // 2 du/dthe / rm^2 / re
            const double cR = S.cR[lane], pgv = S.pgv[lane];
            double cFac = 0., cMc = 0., phk = 0.;
            if constexpr (L1) { cFac = S.cFac[lane]; cMc = S.cMc[lane]; phk = S.phk[lane]; }
            double RS = 0.;
            PSTAMP(0, 1);
#pragma unroll
            for (int s = 0; s < 4; s++) {
                const double v = st[cV];
                vst[s][lane] = v;
                // buoyancy (RHS_Atm.cpp:390-404): K g (r_humid - r_dry) / r_dry, the dry density cancels
                const double y = rcp2(__builtin_fma(kWV, st[cC], 1.));
                const double hod = (1. + st[cC]) * y;   // humid over dry density
                RS = KK * (hod - 1.);
                double lin[4];
#pragma unroll
                for (int n = 0; n < 4; n++) lin[n] = __builtin_fma(cP[n], st[n], cQ[n]);
                lin[cU] -= RS;
                lin[cV] = __builtin_fma(cR, st[cU], lin[cV]);
                if constexpr (L1) {  // surface evaporation source of c (RHS_Atm.cpp:412-430), dry density from this stage's t
                    const double r_humid = phk * rcp2(st[cT]) * hod;
                    lin[cC] = __builtin_fma(r_humid * __builtin_fma(4., st[cC], cMc), cFac, lin[cC]);
                }
#pragma unroll
                for (int n = 0; n < 4; n++) {
                    rhs[n] = __builtin_fma(-v, __builtin_fma(cA, st[n], cB[n]), lin[n]);
                    // RungeKutta_Atm.cpp:58-152
                    if (s == 0) { acc[n] = rhs[n]; st[n] = __builtin_fma(rhs[n], hdt, xn[n]); }
                    else if (s == 1) { acc[n] = __builtin_fma(2., rhs[n], acc[n]); st[n] = __builtin_fma(rhs[n], hdt, xn[n]); }
                    else if (s == 2) { acc[n] = __builtin_fma(2., rhs[n], acc[n]); st[n] = __builtin_fma(rhs[n], dt, xn[n]); }
                    else st[n] = __builtin_fma(acc[n] + rhs[n], dt6, xn[n]);
                }
            }
            PSTAMP(0, 2);
            // results and the side outputs of the fourth evaluation (RHS_Atm.cpp:406-410, 485-492)
            gst(a.x[V_V], o, LIMIT ? limited<V_V>(st[cV], land) : st[cV]);
            gst(a.x[V_U], o, LIMIT ? limited<V_U>(st[cU], land) : st[cU]);
            gst(a.x[V_C], o, LIMIT ? limited<V_C>(st[cC], land) : st[cC]);
            gst(a.x[V_T], o, LIMIT ? limited<V_T>(st[cT], land) : st[cT]);
            gst(a.aux_u, o, land ? 0. : rhs[cU]);
            gst(a.aux_v, o, land ? 0. : rhs[cV] + pgv);
            gst(a.buoy, o, land ? 0. : -RS * cb1000);
#pragma unroll
            for (int n = 0; n < 4; n++) { xm2[n] = xm1[n]; xm1[n] = st[n]; }
            PSTAMP(0, 3);
        }
        PSTAMP(0, 4);
        lds_barrier();
        PSTAMP(0, 5);
    }
}

// ---- chain wave B: w, cloud, ice, co2, one row behind A ------------------------------------------------------------
template <bool LIMIT>
__device__ __forceinline__ void chain_b(const PipeArgs& a, const DevConst& K, PipeShared& sh, const int i, const int bx) {
    const int jm = K.jm, km = K.km;
    const int lane = threadIdx.x & 63;
    const int kk = 1 + bx * 64 + lane;
    const int k = kk <= km - 2 ? kk : km - 2;
    const int NR = jm - 2, NE = NR + 2;
    const unsigned rowB = (unsigned)km * 8u;
    unsigned o_lev = (unsigned)(((size_t)i * jm * km + k) * 8);
    PIN(o_lev);
    constexpr int VARS[4] = { V_W, V_CLOUD, V_ICE, V_CO2 };
    constexpr int CLS[4] = { QB_RE, QB_WV, QB_WV, QB_CO2 };
    constexpr int CPS[4] = { 0, 1, 1, 2 };

    const double hdt = .5 * K.dt, dt = K.dt, dt6 = K.dt * K.i_six;
    double xm1[4], xm2[4] = { 0., 0., 0., 0. };
#pragma unroll
    for (int n = 0; n < 4; n++) xm1[n] = gld(a.x[VARS[n]], o_lev);

    for (int e = 0; e < NE; e++) {
        PSTAMP(1, 0);
        if (e >= 2 && e <= NR + 1) {
            const int j = e - 1;
            const bool window = j >= 2 && j < jm - 3;
            const Slot& S = sh.ring[j & 3];
            const double (*vst)[64] = sh.vst[j & 1];
            const unsigned o = o_lev + (unsigned)j * rowB;
            const bool land = S.fl[lane] & FL_LAND;
            const double cA = S.cA[lane], H1 = S.H1[lane], H2 = S.H2[lane];
            double st[4], xn[4], cB[4], cQ[4], cP[4], acc[4], rhs[4];
#pragma unroll
            for (int n = 0; n < 4; n++) {
                st[n] = S.x0B[n][lane]; xn[n] = S.xnB[n][lane]; cP[n] = S.cPB[CPS[n]][lane];
                const double kb = window ? 0. : S.KBB[n][lane];
                const double kq = (window && (n == 0 || n == 3)) ? 0. : S.KQB[n][lane];
                cB[n] = __builtin_fma(H1, xm1[n], __builtin_fma(H2, xm2[n], kb));
                cQ[n] = __builtin_fma(S.q1B[CLS[n]][lane], xm1[n], __builtin_fma(S.q2B[CLS[n]][lane], xm2[n], kq));
            }
#pragma unroll
            for (int s = 0; s < 4; s++) {
                const double v = vst[s][lane];
#pragma unroll
                for (int n = 0; n < 4; n++) {
                    rhs[n] = __builtin_fma(-v, __builtin_fma(cA, st[n], cB[n]), __builtin_fma(cP[n], st[n], cQ[n]));
                    if (s == 0) { acc[n] = rhs[n]; st[n] = __builtin_fma(rhs[n], hdt, xn[n]); }
                    else if (s == 1) { acc[n] = __builtin_fma(2., rhs[n], acc[n]); st[n] = __builtin_fma(rhs[n], hdt, xn[n]); }
                    else if (s == 2) { acc[n] = __builtin_fma(2., rhs[n], acc[n]); st[n] = __builtin_fma(rhs[n], dt, xn[n]); }
                    else st[n] = __builtin_fma(acc[n] + rhs[n], dt6, xn[n]);
                }
            }
            gst(a.x[V_W], o, LIMIT ? limited<V_W>(st[0], land) : st[0]);
            gst(a.x[V_CLOUD], o, LIMIT ? limited<V_CLOUD>(st[1], land) : st[1]);
            gst(a.x[V_ICE], o, LIMIT ? limited<V_ICE>(st[2], land) : st[2]);
            gst(a.x[V_CO2], o, LIMIT ? limited<V_CO2>(st[3], land) : st[3]);
            gst(a.aux_w, o, land ? 0. : rhs[0]);
#pragma unroll
            for (int n = 0; n < 4; n++) { xm2[n] = xm1[n]; xm1[n] = st[n]; }
            PSTAMP(1, 3);
        }
        PSTAMP(1, 4);
        lds_barrier();
        PSTAMP(1, 5);
    }
}

// Waves of the 1 024-thread workgroup: w and w + 4 (and w + 8, w + 12) share a SIMD.  Wave 0 is chain A, wave 1 chain B,
// and the waves that would share their SIMDs end at once; the eight waves on the other two SIMDs are the prep waves:
// rows r = q mod 4 belong to the pair (2 + 4 q, 3 + 4 q), the first working for chain A, the second for chain B.
template <bool LIMIT>
__global__ __launch_bounds__(1024) void k_atm_rk3d_pipe(const PipeParams* __restrict__ P) {
    const PipeArgs& a = P->a;
    const DevConst& K = P->K;
    __shared__ PipeShared sh;
    // workgroup -> (level, longitude block): all blocks of a level on one XCD (as in k_rk_atm_ws.hip)
    const int gx = (K.km - 2 + 63) / 64;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int lev = xcd + 8 * (idx / gx), bx = idx % gx;
    if (lev >= K.im - 2) return;
    const int i = 1 + lev;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#ifdef ATOMDEV_EXP_PIPE_STAMP
    if (threadIdx.x == 0) { sh.wall[0] = wall_clock64(); sh.cyc[0] = __builtin_readcyclecounter(); }
    for (int n = threadIdx.x; n < 10 * 4 * 8; n += blockDim.x) (&sh.stamps[0][0][0])[n] = 0;
    __syncthreads();
#endif
    if (wave == 0) {
        if (i == 1) chain_a<true, LIMIT>(a, K, sh, i, bx);
        else chain_a<false, LIMIT>(a, K, sh, i, bx);
    } else if (wave == 1) {
        chain_b<LIMIT>(a, K, sh, i, bx);
    } else if ((wave & 3) == 2) {
        if (i == 1) prep_wave<true, true>(a, K, sh, wave >> 2, i, bx);
        else prep_wave<false, true>(a, K, sh, wave >> 2, i, bx);
    } else if ((wave & 3) == 3) {
        prep_wave<false, false>(a, K, sh, wave >> 2, i, bx);
    }
#ifdef ATOMDEV_EXP_PIPE_STAMP
    if (threadIdx.x == 0 && bx == 2 && i == ATOMDEV_EXP_PIPE_STAMP) {  // wave 0 leaves the loop after the last barrier: every stamp is there
        sh.wall[1] = wall_clock64(); sh.cyc[1] = __builtin_readcyclecounter();
        for (int n = 0; n < 10 * 4 * 8; n++) a.aux_w[n] = (double)(long long)(&sh.stamps[0][0][0])[n];
        a.aux_w[400] = (double)(long long)(sh.wall[1] - sh.wall[0]); a.aux_w[401] = (double)(long long)(sh.cyc[1] - sh.cyc[0]);
    }
#endif
}

}  // namespace

// The sweep kernel alone; the caller (k_rk_atm_ws.hip, tolerance form) has run the preparing kernel and holds the
// profiling bracket.
int atomdev_atm_rk3d_pipe_launch(atomdev* h, bool limit) {
    if (!h->pipe_params) {
        PipeParams P;
        PipeArgs& a = P.a;
        const int X[8] = { ATOMDEV_T, ATOMDEV_U, ATOMDEV_V, ATOMDEV_W, ATOMDEV_C, ATOMDEV_CLOUD, ATOMDEV_ICE, ATOMDEV_CO2 };
        const int XN[8] = { ATOMDEV_TN, ATOMDEV_UN, ATOMDEV_VN, ATOMDEV_WN, ATOMDEV_CN, ATOMDEV_CLOUDN, ATOMDEV_ICEN, ATOMDEV_CO2N };
        for (int n = 0; n < 8; n++) { a.x[n] = h->f[X[n]]; a.xn[n] = h->f[XN[n]]; }
        a.p_dyn = h->f[ATOMDEV_P_DYN];
        a.S_v = h->f[ATOMDEV_S_V]; a.S_c = h->f[ATOMDEV_S_C]; a.S_i = h->f[ATOMDEV_S_I]; a.S_r = h->f[ATOMDEV_S_R]; a.S_s = h->f[ATOMDEV_S_S];
        a.aux_u = h->f[ATOMDEV_AUX_U]; a.aux_v = h->f[ATOMDEV_AUX_V]; a.aux_w = h->f[ATOMDEV_AUX_W]; a.buoy = h->f[ATOMDEV_BUOYANCY];
        a.flags = h->flags;
        a.evap = h->g[ATOMDEV_EVAPORATION_DALTON]; a.precip = h->g[ATOMDEV_PRECIPITATION];
        a.c3 = h->plane;
        a.ph100 = h->scr[0]; a.h0i = h->scr[1];
        a.rad = h->rad; a.costhe = h->costhe;
        a.tab_irm2sin = h->tab_irm2sin; a.tab_coefv = h->tab_coefv;
        P.K = h->k;
        // field pointers of the atmosphere never change after creation: one upload per handle
        ATOMDEV_TRY(hipMalloc(&h->pipe_params, sizeof(P)));
        ATOMDEV_TRY(hipMemcpy(h->pipe_params, &P, sizeof(P), hipMemcpyHostToDevice));
    }
    ATOMDEV_REQUIRE(h->im >= 3 && h->jm >= 3 && h->km >= 3, "atm_rk3d: grid without interior points");
    ATOMDEV_REQUIRE(h->n3 * sizeof(double) < ((size_t)1 << 32), "atm_rk3d: field larger than 4 GiB");
    const unsigned gx = (h->km - 2 + 63) / 64, rounds = (h->im - 2 + 7) / 8;
    dim3 grid(8 * rounds * gx);
    if (limit)
        hipLaunchKernelGGL(k_atm_rk3d_pipe<true>, grid, dim3(1024), 0, h->stream, (const PipeParams*)h->pipe_params);
    else
        hipLaunchKernelGGL(k_atm_rk3d_pipe<false>, grid, dim3(1024), 0, h->stream, (const PipeParams*)h->pipe_params);
    ATOMDEV_TRY(hipGetLastError());
    return 0;
}
