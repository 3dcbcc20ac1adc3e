// Hydrosphere 3-D Runge-Kutta sweep, pipelined form (variant 3; tolerance form only).
//
// Reference: hydrosphere/RungeKutta_Hyd.cpp:30-124 with RHS_Hyd.cpp:60-447 at every stage.  The arithmetic is the
// affine restatement of k_rk_hyd_ws.hip's tolerance form, the division of labour the one of k_rk_atm_pipe.hip:
// with the neighbour rows and the level above fixed during the four stages of a row, the right-hand side of a
// variable x is affine in its stage value apart from the two advection products,
//     rhs = -u (cAr x + cBr) - v (cA x + cB) + cP x + cQ
//           (+ cR u - pgn / rho for v, + g' (rsw - rho) / rho for u, evaporation for c on level im-2;  rho = rw + .7 c c_0)
// and cB, cQ are affine in the results of the two rows before,
//     cB = H1 x[j-1] + H2 x[j-2] + KB        cQ = q1 x[j-1] + q2 x[j-2] + KQ
// where everything else depends on the mask, the geometry, the level above and OLD values only.
//
// Mapping.  A workgroup owns 64 neighbouring theta lines of one level.  Of its twelve waves ten live:
//   eight "memory" waves in four pairs that take the rows in turn (row r belongs to pair r mod 4; one wave of the pair
//       works for chain A, the other for chain B): loads, the row's coefficient packet into a ring of four LDS slots,
//       the stores of the row that stood in the slot before, the loads of the row four on;
//   chain wave A carries v, u, c (u and v feed every variable, c the density in u and v) through the four stages and
//       leaves the stage values of u and v in LDS;
//   chain wave B carries t and w behind it.
// The chain waves touch LDS only; the waves follow each other's progress words in LDS, no barrier inside the loop.
//
// Levels.  The sweep is out of place (see k_rk_hyd.hip): level i reads the OLD level i+1, so all levels run side by
// side; the boundary shell of the alternate buffers is refreshed by extra workgroups of the same launch.  Level
// im-2 reads the salinity level im-3 has just computed (RHS_Hyd.cpp:356-357): the memory waves of a level im-3
// workgroup store c with device-scope stores and, once the memory system has acknowledged them, publish the row in a
// progress word of their longitude block and pair; the memory wave of the level im-2 workgroup that builds the same
// row's packet waits for that word and fetches the row past its caches.  A wait is bounded (an error word is set and
// the sweep finishes with whatever it reads); producers have the lower workgroup index, so they are dispatched first.
// Consecutive levels go to the same XCD (a chunk of levels per XCD), so that the rows of level i+1 a workgroup reads
// are, most of the time, rows the workgroup of level i+1 has just brought into the XCD's L2.
#include <algorithm>
#include <cstdlib>

#include "atomdev_internal.h"

namespace {

struct PipeArgs {
    const double* x[5];      // v, u, c, t, w: current state
    double* xo[5];           // next state
    const double* xn[5];
    const double* p_dyn;
    double *r_water, *r_salt_water;   // read; rewritten on land
    double *aux_u, *aux_v, *aux_w, *buoy, *salt_balance, *salt_finger, *salt_diffusion;
    const uint8_t* flags;
    const double *bathy, *evap, *precip;   // 2-D
    const double *rad, *costhe, *tab_irm2sin, *tab_coefv;
};
constexpr int MAX_SLOTS = 256;
struct PipeParams {
    PipeArgs a;
    DevConst K;
    int chunk;                    // level slots per XCD
    int level_of_slot[MAX_SLOTS]; // [xcd * chunk + position]: level index (i - 1), or -1 for an empty slot
};
enum { X_V = 0, X_U = 1, X_C = 2, X_T = 3, X_W = 4 };

// One row's packet, 64 lanes wide, one half per chain wave.  On the three pole-side rows (central differences: x[j-2]
// does not enter, old x[j+1] does) the arrays of the x[j-2] factors carry the constant terms KB instead: q2[0..1] and,
// for chain A, qu2.
struct HalfA {
    double cA[64], H1[64], H2[64], cAr[64];
    double q1[2][64], q2[2][64];               // diffusion classes: u and v (re), c (re sc)
    double cP[3][64], x0[3][64], xn[3][64], KQ[3][64], cBr[3][64];   // v, u, c
    double cR[64], qu1[64], qu2[64], pgn[64];
    double rw[64], rsw[64], rw2[64], rsw2[64];  // densities of the first evaluation and of the following (RHS_Hyd.cpp:395-402)
    double cFac4[64], cMF[64];                  // level im-2: the evaporation source of c
};
struct HalfB {
    double cA[64], H1[64], H2[64], cAr[64];
    double q1[2][64], q2[2][64];               // t (re pr), w (re)
    double cP[2][64], x0[2][64], xn[2][64], KQ[2][64], cBr[2][64];   // t, w
};
constexpr int NSLOT = 4;
constexpr int GEO_ROWS = 768;                  // grids with more rows fetch the geometry with the row's other loads
struct Ctl { int rdy, stored; };
struct PipeShared {
    HalfA ha[NSLOT];
    HalfB hb[NSLOT];
    double outA[NSLOT][7][64];                 // v, u, c, aux_u, aux_v, buoyancy, salt balance of a finished row
    double outB[NSLOT][3][64];                 // t, w, aux_w
    double ust[2][4][64], vst[2][4][64];       // stage values of u and v of a row, by row parity
    double geo[GEO_ROWS][2];                   // the level's geometry by row: cos(the) / (rm^2 sin(the)), the viscous v coefficient
    Ctl ctlA[NSLOT], ctlB[NSLOT];
    int doneA, doneB;
};

typedef const __attribute__((address_space(1))) char* gcchar_t;
typedef __attribute__((address_space(1))) char* gchar_t;
__device__ __forceinline__ double gld(const double* base, unsigned off) {
    return *(const __attribute__((address_space(1))) double*)((gcchar_t)base + off);
}
__device__ __forceinline__ void gst(double* base, unsigned off, double x) {
    *(__attribute__((address_space(1))) double*)((gchar_t)base + off) = x;
}
// The same with the non-temporal hint, for what a step touches once (the old-time-level fields and the source terms read,
// the side outputs written): the last-level cache then keeps the fields the next kernels read again.
__device__ __forceinline__ double gld_once(const double* base, unsigned off) {
    return __builtin_nontemporal_load((const __attribute__((address_space(1))) double*)((gcchar_t)base + off));
}
__device__ __forceinline__ void gst_once(double* base, unsigned off, double x) {
    __builtin_nontemporal_store(x, (__attribute__((address_space(1))) double*)((gchar_t)base + off));
}
// device-scope accesses (written through / fetched past the XCD's L2), for the rows that cross workgroups
__device__ __forceinline__ void gst_agent(double* base, unsigned off, double x) {
    __hip_atomic_store((__attribute__((address_space(1))) double*)((gchar_t)base + off), x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double gld_agent(const double* base, unsigned off) {
    return __hip_atomic_load((const __attribute__((address_space(1))) double*)((gcchar_t)base + off), __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
}
#define PIN(x) asm volatile("" : "+v"(x))
#define SPIN(x) asm volatile("" : "+s"(x))

typedef volatile __attribute__((address_space(3))) int* lds_word_t;
#define LDS_WORD(p) ((lds_word_t)(p))
// Progress words in LDS: see k_rk_atm_pipe.hip (a wave's LDS operations are carried out in issue order; only lane 0
// writes; a wait gives up after about a second and says so in *err).
__device__ __forceinline__ void announce(int* word, int row, int lane) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) *LDS_WORD(word) = row;
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ void announce_in_order(int* word, int row, int lane) {
    asm volatile("" ::: "memory");
    if (lane == 0) *LDS_WORD(word) = row;
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ void wait_for(const int* word, int need, int* err) {
    asm volatile("" ::: "memory");
    for (int n = 0; __builtin_amdgcn_readfirstlane(*LDS_WORD(word)) < need; n++) {
        __builtin_amdgcn_s_sleep(2);
        if (n > (1 << 22)) {
            __hip_atomic_store(err, 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            break;
        }
    }
    asm volatile("" ::: "memory");
}
// Progress words in device memory, between the workgroups of levels im-3 and im-2 (see k_rk_hyd_ws.hip for why these
// are sc1 accesses in program order and not release / acquire).
__device__ __forceinline__ void announce_rows(int* progress, int rows, int lane) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the rows' stores have been acknowledged
    if (lane == 0) __hip_atomic_store(progress, rows, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ void wait_rows(const int* progress, int need, int* err) {
    asm volatile("" ::: "memory");
    for (int n = 0;; n++) {
        const int seen = __hip_atomic_load((const __attribute__((address_space(1))) int*)progress, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__builtin_amdgcn_readfirstlane(seen) >= need) break;
        __builtin_amdgcn_s_sleep(4);
        if (n > (1 << 21)) {
            __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            break;
        }
    }
    asm volatile("" ::: "memory");
}
// a value every lane holds alike, moved to scalar registers
__device__ __forceinline__ double uniform(double x) {
    const unsigned long long b = __builtin_bit_cast(unsigned long long, x);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double rcp2(double b) {  // hardware estimate and two Newton steps: full precision
    double y = __builtin_amdgcn_rcp(b);
    double e = __builtin_fma(-b, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-b, y, 1.0);
    return __builtin_fma(y, e, y);
}

// ---- a memory wave: rows r = q, q + 4, ... (r >= 1), the half of the row that chain A (FORA) or chain B works on ----
// TOP: level im-2 (no radial differences; chain A's half carries the evaporation source and waits for level im-3);
// FEEDS: level im-3 (chain A's half publishes its rows of c for level im-2).
// GEO: the level's geometry table is in LDS (grids of up to GEO_ROWS rows).
template <bool TOP, bool FEEDS, bool FORA, bool GEO>
__device__ __forceinline__ void memory_wave(const PipeArgs& a, const DevConst& K, PipeShared& sh, const int q, const int i, const int bx,
                                            int* progress, int* err) {
    const int jm = K.jm, km = K.km;
    const int lane = threadIdx.x & 63;
    const int kk = 1 + bx * 64 + lane;
    const int k = kk <= km - 2 ? kk : km - 2;  // idle lanes shadow the last line
    const bool kwin = (k >= 2) && (k < km - 3);
    const int NR = jm - 2;
    const unsigned rowB = (unsigned)km * 8u;
    const unsigned lvlB = (unsigned)jm * rowB;
    unsigned o_lev = (unsigned)(((size_t)i * jm * km + k) * 8);   // row 0 of the line, as a byte offset into a 3-D field
    unsigned o_sfc = (unsigned)k * 8u;                             // ... into a 2-D field
    PIN(o_lev); PIN(o_sfc);
    HalfA& SA = sh.ha[q];
    HalfB& SB = sh.hb[q];
    Ctl* ctl = FORA ? &sh.ctlA[q] : &sh.ctlB[q];
    const int* done = FORA ? &sh.doneA : &sh.doneB;
    constexpr int NV = FORA ? 3 : 2;
    constexpr int VARS[3] = { FORA ? X_V : X_T, FORA ? X_U : X_W, X_C };
    int* const word = progress + 4 * bx + q;   // of this longitude block and pair (levels im-3 and im-2 only)

    const double rm = a.rad[i];
    const double i_rm = 1. / rm, i_rm2 = 1. / (rm * rm);
    const double hight = (double)i * K.topo_step;
    double i_dthe = K.i_dthe, i_two_dthe = K.i_two_dthe, i_dthe2 = K.i_dthe2, i_dr2 = K.i_dr2, i_dphi2 = K.i_dphi2, i_dr = K.i_dr;
    double i_re = K.i_re, i_re_x = FORA ? K.i_re_sc : K.i_re_pr, topo_step = K.topo_step, i_topo_step = K.i_topo_step;
    SPIN(i_dthe); SPIN(i_two_dthe); SPIN(i_dthe2); SPIN(i_dr2); SPIN(i_dphi2); SPIN(i_dr); SPIN(i_re); SPIN(i_re_x); SPIN(topo_step); SPIN(i_topo_step);
    const double two = 2. * i_rm2 * i_re;
    const double evf = K.coeff_salinity * K.i_two_dr;
    // diffusion classes and divisors: A: {v, u} -> re, c -> re sc;  B: t -> re pr, w -> re
    const double idc[2] = { FORA ? i_re : i_re_x, FORA ? i_re_x : i_re };
    constexpr int CLS[3] = { 0, FORA ? 0 : 1, 1 };
    // factor of dr1 / rm in the diffusion bracket: 4 for u, 2 for the others (RHS_Hyd.cpp:406-437)
    const double krs[3] = { 2. * i_rm, (FORA ? 4. : 2.) * i_rm, 2. * i_rm };

    // what a row's loads leave behind until the row's turn comes
    double lx[3], lxn[3], lxa[3], lxp[3], lp[4], lbathy = 0., lrw = 0., lrsw = 0., lev = 0., lpr = 0.;
    constexpr bool geo_lds = GEO;
    double lcos = 0., lirs = 0., lcoefv = 0.;
    unsigned lfl;
    auto issue = [&](const int r) {
#ifdef ATOMDEV_EXP_HP_NOMEM  // timing experiment only (results are wrong): no vector-memory traffic inside the loop
        if (r > 8) return;
#endif
        const unsigned o = o_lev + (unsigned)r * rowB;
        const unsigned oq = o_sfc + (unsigned)r * rowB;
        lfl = *(const __attribute__((address_space(1))) uint8_t*)((gcchar_t)a.flags + (o >> 3));
#pragma unroll
        for (int n = 0; n < NV; n++) {
            lx[n] = gld(a.x[VARS[n]], o); lxn[n] = gld_once(a.xn[VARS[n]], o);
#ifdef ATOMDEV_EXP_HP_NOXA  // timing experiment only (results are wrong): the level above is not fetched
            lxa[n] = 0.;
#else
            lxa[n] = gld(a.x[VARS[n]], o + lvlB);   // level i+1 <= im-1 always exists
#endif
        }
        if constexpr (!geo_lds) { lcos = gld(a.costhe + r, 0u); lirs = gld(a.tab_irm2sin + (size_t)i * jm + r, 0u); }
        lbathy = gld(a.bathy, oq);
        if constexpr (FORA) {
            if constexpr (!geo_lds) lcoefv = gld(a.tab_coefv + (size_t)i * jm + r, 0u);
            lrw = gld_once(a.r_water, o); lrsw = gld_once(a.r_salt_water, o);
            lp[0] = r >= 2 ? gld(a.p_dyn, o - 2u * rowB) : 0.;
            lp[1] = gld(a.p_dyn, o - rowB);
            lp[2] = gld(a.p_dyn, o);
            lp[3] = gld(a.p_dyn, o + rowB);
            if constexpr (TOP) { lev = gld(a.evap, oq); lpr = gld(a.precip, oq); }
        }
        const bool window = r >= 2 && r < jm - 3;
        if (!window) {
#pragma unroll
            for (int n = 0; n < NV; n++) lxp[n] = gld(a.x[VARS[n]], o + rowB);
        }
    };
    // Level im-2: c of the level below, row r, once level im-3 has published it.  Called a row of this wave ahead, when
    // the wave would otherwise only wait for its chain wave: the word's and the row's latencies stay off the row's path.
    double lcb = 0.;
    auto fetch_below = [&](const int r) {
        const unsigned o = o_lev + (unsigned)r * rowB;
#ifdef ATOMDEV_EXP_HP_NOPAIR  // timing experiment only (results are wrong): level im-2 takes a stale salinity row
        lcb = gld(a.x[X_C], o - lvlB);
#else
        wait_rows(word, r, err);
        lcb = gld_agent(a.xo[X_C], o - lvlB);
#endif
    };
    // The results of row `rs`, which the chain wave left in the slot's output arrays; side outputs zero on land, where
    // the two densities are rewritten instead (RHS_Hyd.cpp:379-402, 440-446).
    auto store_row = [&](const int rs, const bool land) {
#ifdef ATOMDEV_EXP_HP_NOMEM
        if (rs > 8) { announce(&ctl->stored, rs, lane); return; }
#endif
        const unsigned o = o_lev + (unsigned)rs * rowB;
        if constexpr (FORA) {
            const double (*out)[64] = sh.outA[q];
            const double v = out[0][lane], u = out[1][lane], c = out[2][lane];
            const double au = out[3][lane], av = out[4][lane], bu = out[5][lane], sb = out[6][lane];
            gst(a.xo[X_V], o, v);
            gst(a.xo[X_U], o, u);
            if constexpr (FEEDS) gst_agent(a.xo[X_C], o, c); else gst(a.xo[X_C], o, c);
            double sfinger, sdiff;
            if (sb < 0.) { sdiff = sb; sfinger = 0.; } else { sfinger = sb; sdiff = 0.; }
            gst_once(a.aux_u, o, land ? 0. : au);
            gst_once(a.aux_v, o, land ? 0. : av);
            gst_once(a.buoy, o, land ? 0. : bu);
            gst_once(a.salt_balance, o, land ? 0. : sb);
            gst_once(a.salt_finger, o, land ? 0. : sfinger);
            gst_once(a.salt_diffusion, o, land ? 0. : sdiff);
            if (land) { gst(a.r_water, o, 1007.); gst(a.r_salt_water, o, 1032.); }
        } else {
            const double (*out)[64] = sh.outB[q];
            const double t = out[0][lane], w = out[1][lane], aw = out[2][lane];
            gst(a.xo[X_T], o, t);
            gst(a.xo[X_W], o, w);
            gst_once(a.aux_w, o, land ? 0. : aw);
        }
        announce(&ctl->stored, rs, lane);   // (waits until the values have left LDS)
    };
    const int first = q == 0 ? 4 : q;
    if (first <= NR) {
        issue(first);
        if constexpr (TOP && FORA) fetch_below(first);
    }
    bool land_prev = false;
    int r_last = 0;

    for (int r = first; r <= NR; r += 4) {
        // the slot is free once the chain wave has finished the row that stood in it; its results go to memory first
        wait_for(done, r - 4, err);
        if (r > 4) store_row(r - 4, land_prev);
        const bool window = r >= 2 && r < jm - 3;
        const double g = geo_lds ? sh.geo[r][0] : lcos * lirs;
        const double coefv = geo_lds ? sh.geo[r][1] : lcoefv;
        // the mask: RHS_Hyd.cpp:96-139, 191-214, 296-300, 326-330
        const unsigned fl = lfl;
        const bool land = fl & FL_LAND;
        const double hdj = ((fl & FL_H0J) && (fl & FL_H0K)) ? 0. : 1.;
        const bool d2zero = kwin && (fl & FL_D2ZERO);
        const bool rlive = !TOP && !(fl & FL_RDEAD);
        double h0i = 1.;
        if (land) {
            const double topo_diff = fabs(hight - lbathy);
            if (!(topo_diff == topo_step)) h0i = topo_diff * i_topo_step;
        }
        // which theta form stands (RHS_Hyd.cpp:156-161, 177-181, 250-276), as factors of x, x[j-1], x[j-2], x[j+1]
        const int mode = !window ? 0 : ((fl & FL_ONESIDED_S) ? 2 : 1);
        const double fa = mode == 0 ? 0. : (mode == 1 ? 1. : -3.), fe = mode == 0 ? -2. : (mode == 1 ? 0. : 2.);
        const double b1 = mode == 2 ? 4. : -1., b2 = mode == 2 ? -1. : 0., b3 = mode == 0 ? 1. : 0.;
        const double f1 = mode == 0 ? 1. : (mode == 1 ? 0. : -2.), f2 = mode == 2 ? 1. : 0., f3 = mode == 0 ? 1. : 0.;
        const double hD = hdj * (mode == 1 ? i_dthe : i_two_dthe), hE = d2zero ? 0. : hdj * i_dthe2;
        const double A1 = fa * hD, A2 = fe * hE;
        const double hDr = hD * i_rm, hDg = hD * g, hEr = hE * i_rm2;
        const double G1 = __builtin_fma(f1, hEr, b1 * hDg), G2 = __builtin_fma(f2, hEr, b2 * hDg);
        const double G3 = __builtin_fma(f3, hEr, b3 * hDg);
        const double Pth = __builtin_fma(A2, i_rm2, A1 * g);     // theta part of the bracket's factor of x
        const double cAr = rlive ? -i_dr : 0.;
        const double H3 = b3 * hDr;
        double (*x0)[64] = FORA ? SA.x0 : SB.x0, (*xn)[64] = FORA ? SA.xn : SB.xn, (*kq)[64] = FORA ? SA.KQ : SB.KQ;
        double (*cbr)[64] = FORA ? SA.cBr : SB.cBr, (*cp)[64] = FORA ? SA.cP : SB.cP;
        double (*q1)[64] = FORA ? SA.q1 : SB.q1, (*q2)[64] = FORA ? SA.q2 : SB.q2;
        (FORA ? SA.cA : SB.cA)[lane] = A1 * i_rm;
        (FORA ? SA.H1 : SB.H1)[lane] = b1 * hDr;
        (FORA ? SA.H2 : SB.H2)[lane] = b2 * hDr;
        (FORA ? SA.cAr : SB.cAr)[lane] = cAr;
        double KB[3] = { 0., 0., 0. };
        if (!window) {
#pragma unroll
            for (int n = 0; n < NV; n++) KB[n] = H3 * lxp[n];
        }
#pragma unroll
        for (int c = 0; c < 2; c++) { q1[c][lane] = G1 * idc[c]; q2[c][lane] = window ? G2 * idc[c] : KB[c]; }
        // damping of the variable (RHS_Hyd.cpp:406-437: u is damped with 1 / dthe^2)
        double dmp[3];
        if constexpr (FORA) {
            dmp[0] = ((fl & FL_H0J) ? 1. : 0.) * i_dthe2 + coefv * hdj * i_re;
            dmp[1] = h0i * i_dthe2;
            dmp[2] = h0i * i_dr2;
        } else {
            dmp[0] = h0i * i_dr2;
            dmp[1] = ((fl & FL_H0K) ? 1. : 0.) * i_dphi2;
            dmp[2] = 0.;
        }
#pragma unroll
        for (int n = 0; n < NV; n++) {
            const double iv = idc[CLS[n]];
            const double cBr = rlive ? lxa[n] * i_dr : 0.;
            x0[n][lane] = lx[n]; xn[n][lane] = lxn[n];
            cbr[n][lane] = cBr;
            cp[n][lane] = __builtin_fma(__builtin_fma(krs[n], cAr, Pth), iv, -dmp[n]);
            double KQ = krs[n] * cBr * iv;
            if (!window) KQ = __builtin_fma(G3 * iv, lxp[n], KQ);
            if (FORA && n == 0 && !window) KQ = __builtin_fma(two * hD * b3, lxp[1], KQ);   // 2 du/dthe / rm^2 / re: old u[j+1]
            kq[n][lane] = KQ;
        }
        if constexpr (FORA) {
            // pressure gradient of the v equation, before the division by the density (p_dyn is not changed by the sweep)
            const double dp = hD * (fa * lp[2] + b1 * lp[1] + b2 * lp[0] + b3 * lp[3]);
            SA.pgn[lane] = hdj * dp * i_rm;
            const double twoD = two * hD;
            SA.cR[lane] = two * A1; SA.qu1[lane] = twoD * b1; SA.qu2[lane] = window ? twoD * b2 : KB[2];
            SA.rw[lane] = lrw; SA.rsw[lane] = lrsw;
            SA.rw2[lane] = land ? 1007. : lrw; SA.rsw2[lane] = land ? 1032. : lrsw;
            if constexpr (TOP) {  // RHS_Hyd.cpp:355-372; c of level im-1 (never updated) is the "level above"
                const double c_below = lcb;
                const double c_top = lxa[2];
                const double cFac = land ? 0. : evf * (1. - 2. * c_top) * (lev - lpr);
                const double cM = -3. * c_top - c_below;
                SA.cFac4[lane] = 4. * cFac; SA.cMF[lane] = cM * cFac;
            }
        }
        land_prev = land;
        announce(&ctl->rdy, r, lane);
        r_last = r;
        if (r + 4 <= NR) {
            issue(r + 4);
            if constexpr (TOP && FORA) fetch_below(r + 4);
        }
        if constexpr (FEEDS && FORA) {
            if (r > 4) announce_rows(word, r - 4, lane);
        }
    }
    if (r_last) {  // the wave's last row
        wait_for(done, r_last, err);
        store_row(r_last, land_prev);
        if constexpr (FEEDS && FORA) announce_rows(word, r_last, lane);
    }
    // the next sweep starts from "no row finished": the reader of the word is the last to touch it
    if constexpr (TOP && FORA) {
        if (lane == 0) __hip_atomic_store(word, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---- chain wave A: v, u, c ---------------------------------------------------------------------------------------
template <bool TOP>
__device__ __forceinline__ void chain_a(const PipeArgs& a, const DevConst& K, PipeShared& sh, const int i, const int bx, int* err) {
    const int jm = K.jm, km = K.km;
    const int lane = threadIdx.x & 63;
    const int kk = 1 + bx * 64 + lane;
    const int k = kk <= km - 2 ? kk : km - 2;
    const int NR = jm - 2;
    const unsigned o_lev = (unsigned)(((size_t)i * jm * km + k) * 8);
    constexpr int CLS[3] = { 0, 0, 1 };
    enum { cV = 0, cU = 1, cC = 2 };

    // (wave-uniform constants, kept in scalar registers: the vector registers are needed for two packets)
    double c07 = uniform(.7 * K.c_0), KK = K.buoy_g, hdt = uniform(.5 * K.dt), dt = K.dt, dt6 = uniform(K.dt * K.i_six);
    double cb1000 = uniform(K.coeff_buoy * 1000.);
    SPIN(KK); SPIN(dt);
    double xm1[3], xm2[3] = { 0., 0., 0. };
#pragma unroll
    for (int n = 0; n < 3; n++) xm1[n] = gld(a.x[n], o_lev);

    // A row's packet in registers.  The packet of row j+1 is fetched while row j is worked on (its slot was written rows
    // ago), so two of these are alive at a time and the loop below takes the rows in twos.
    struct Packet {
        int w_rdy, w_st;
        double cA, H1, H2, cAr, cR, pgn, qu1, qu2, rw, rsw, rw2, rsw2, cFac4, cMF;
        double q1[2], q2[2], st[3], xn[3], cP[3], kq[3], cBr[3];
    };
    // Words first, then the packet: a packet read behind a word that says "there" is the packet (a wave's LDS reads are
    // carried out in order); nothing is waited for here.  In two parts: one is fetched while the row before is in its
    // stages, the other (what the top of a row needs: the factors of the two rows before, then the coefficients the
    // stages do not change) when the registers of that row's stages are free again.
    auto fetch_main = [&](Packet& P, const int j) {
        const HalfA& S = sh.ha[j & 3];
        asm volatile("" ::: "memory");
        P.w_rdy = *LDS_WORD(&sh.ctlA[j & 3].rdy); P.w_st = *LDS_WORD(&sh.ctlA[j & 3].stored);
        asm volatile("" ::: "memory");
        P.cA = S.cA[lane]; P.pgn = S.pgn[lane];
        P.rw = S.rw[lane]; P.rsw = S.rsw[lane]; P.rw2 = S.rw2[lane]; P.rsw2 = S.rsw2[lane];
#pragma unroll
        for (int m = 0; m < 3; m++) { P.st[m] = S.x0[m][lane]; P.xn[m] = S.xn[m][lane]; }
        if constexpr (TOP) { P.cFac4 = S.cFac4[lane]; P.cMF = S.cMF[lane]; } else { P.cFac4 = 0.; P.cMF = 0.; }
        asm volatile("" ::: "memory");
    };
    auto fetch_top = [&](Packet& P, const int j) {
        const HalfA& S = sh.ha[j & 3];
        asm volatile("" ::: "memory");
        P.H1 = S.H1[lane]; P.H2 = S.H2[lane]; P.qu1 = S.qu1[lane]; P.qu2 = S.qu2[lane];
#pragma unroll
        for (int c = 0; c < 2; c++) { P.q1[c] = S.q1[c][lane]; P.q2[c] = S.q2[c][lane]; }
#pragma unroll
        for (int m = 0; m < 3; m++) P.kq[m] = S.KQ[m][lane];
        P.cAr = S.cAr[lane]; P.cR = S.cR[lane];   // (first needed a little later: by the first evaluation)
#pragma unroll
        for (int m = 0; m < 3; m++) { P.cP[m] = S.cP[m][lane]; P.cBr[m] = S.cBr[m][lane]; }
        asm volatile("" ::: "memory");
    };
    auto row = [&](const int j, Packet& P, Packet& N) {
        const bool window = j >= 2 && j < jm - 3;
        double (*vst)[64] = sh.vst[j & 1];
        double (*ust)[64] = sh.ust[j & 1];
        double (*out)[64] = sh.outA[j & 3];
        // the packet was fetched a row ago; if its word said "not yet" then, the word is waited for and it is read again
        if (__builtin_amdgcn_readfirstlane(P.w_rdy) < j) {
            wait_for(&sh.ctlA[j & 3].rdy, j, err);
            fetch_main(P, j);
        }
        fetch_top(P, j);
        const int w_b = *LDS_WORD(&sh.doneB);   // (looked at below, before the first stage value is written)
        double cB[3], cQ[3], acc[3], rhs[3];
#pragma unroll
        for (int n = 0; n < 3; n++) {
            const double alias = n < 2 ? P.q2[n] : P.qu2;   // pole-side rows: KB of the variable
            const double kb = window ? 0. : alias;
            const double q2n = window ? P.q2[CLS[n]] : 0.;
            cB[n] = __builtin_fma(P.H1, xm1[n], __builtin_fma(window ? P.H2 : 0., xm2[n], kb));
            cQ[n] = __builtin_fma(P.q1[CLS[n]], xm1[n], __builtin_fma(q2n, xm2[n], P.kq[n]));
        }
        cQ[cV] = __builtin_fma(P.qu1, xm1[cU], __builtin_fma(window ? P.qu2 : 0., xm2[cU], cQ[cV]));  // 2 du/dthe / rm^2 / re
#pragma unroll
        for (int n = 0; n < 3; n++) { PIN(cB[n]); PIN(cQ[n]); }
        // the next row's packet, read while this row's stages are worked out
        if (j + 1 <= NR) fetch_main(N, j + 1);
        // the buffer of the stage values is free once chain B has done row j-2
        if (__builtin_amdgcn_readfirstlane(w_b) < j - 2) wait_for(&sh.doneB, j - 2, err);
        double rw = P.rw, rsw = P.rsw;
        double RS = 0., sb = 0., pgv = 0.;
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const double v = P.st[cV], u = P.st[cU];
            vst[s][lane] = v; ust[s][lane] = u;
            // RHS_Hyd.cpp:345, 374-377: density of the salt water, buoyancy, pressure gradient
            const double swr = __builtin_fma(c07, P.st[cC], rw);
            const double y = rcp2(swr);
            RS = KK * __builtin_fma(rsw, y, -1.);
            sb = swr - rsw;
            pgv = P.pgn * y;
            double lin[3];
#pragma unroll
            for (int n = 0; n < 3; n++) lin[n] = __builtin_fma(P.cP[n], P.st[n], cQ[n]);
            lin[cU] += RS;
            lin[cV] = __builtin_fma(P.cR, u, lin[cV]) - pgv;
            if constexpr (TOP) {  // surface evaporation source of c (RHS_Hyd.cpp:355-372)
                double sal = swr * __builtin_fma(P.cFac4, P.st[cC], P.cMF);
                sal = fmin(fmax(sal, -20.), 20.);
                lin[cC] += sal;
            }
            if (s == 0) { rw = P.rw2; rsw = P.rsw2; }   // on land the following evaluations see the rewritten densities
#pragma unroll
            for (int n = 0; n < 3; n++) {
                const double tv = __builtin_fma(-v, __builtin_fma(P.cA, P.st[n], cB[n]), lin[n]);
                rhs[n] = TOP ? tv : __builtin_fma(-u, __builtin_fma(P.cAr, P.st[n], P.cBr[n]), tv);
                // RungeKutta_Hyd.cpp:56-120
                if (s == 0) { acc[n] = rhs[n]; P.st[n] = __builtin_fma(rhs[n], hdt, P.xn[n]); }
                else if (s == 1) { acc[n] = __builtin_fma(2., rhs[n], acc[n]); P.st[n] = __builtin_fma(rhs[n], hdt, P.xn[n]); }
                else if (s == 2) { acc[n] = __builtin_fma(2., rhs[n], acc[n]); P.st[n] = __builtin_fma(rhs[n], dt, P.xn[n]); }
                else P.st[n] = __builtin_fma(acc[n] + rhs[n], dt6, P.xn[n]);
            }
        }
#pragma unroll
        for (int n = 0; n < 3; n++) { PIN(P.st[n]); PIN(rhs[n]); }
        PIN(RS); PIN(sb); PIN(pgv);
        // results and the side outputs of the fourth evaluation, for the memory wave of the slot to store; the slot's
        // output arrays are free once row j-4 has been stored
        if (__builtin_amdgcn_readfirstlane(P.w_st) < j - 4) wait_for(&sh.ctlA[j & 3].stored, j - 4, err);
#pragma unroll
        for (int n = 0; n < 3; n++) { out[n][lane] = P.st[n]; xm2[n] = xm1[n]; xm1[n] = P.st[n]; }
        out[3][lane] = rhs[cU]; out[4][lane] = rhs[cV] + pgv; out[5][lane] = RS * cb1000; out[6][lane] = sb;
        announce_in_order(&sh.doneA, j, lane);
    };
    Packet X, Y;
    fetch_main(X, 1);
    for (int j = 1; j <= NR; j += 2) {
        row(j, X, Y);
        if (j + 1 <= NR) row(j + 1, Y, X);
    }
}

// ---- chain wave B: t, w, behind A --------------------------------------------------------------------------------
template <bool TOP>
__device__ __forceinline__ void chain_b(const PipeArgs& a, const DevConst& K, PipeShared& sh, const int i, const int bx, int* err) {
    const int jm = K.jm, km = K.km;
    const int lane = threadIdx.x & 63;
    const int kk = 1 + bx * 64 + lane;
    const int k = kk <= km - 2 ? kk : km - 2;
    const int NR = jm - 2;
    const unsigned o_lev = (unsigned)(((size_t)i * jm * km + k) * 8);

    const double hdt = .5 * K.dt, dt = K.dt, dt6 = K.dt * K.i_six;
    double xm1[2], xm2[2] = { 0., 0. };
    xm1[0] = gld(a.x[X_T], o_lev); xm1[1] = gld(a.x[X_W], o_lev);

    for (int j = 1; j <= NR; j++) {
        const bool window = j >= 2 && j < jm - 3;
        const HalfB& S = sh.hb[j & 3];
        const double (*vst)[64] = sh.vst[j & 1];
        const double (*ust)[64] = sh.ust[j & 1];
        double (*out)[64] = sh.outB[j & 3];
        double cA, H1, H2, cAr, vs[4], us[4];
        double st[2], xn[2], cB[2], cQ[2], cP[2], cBr[2], acc[2], rhs[2], q1[2], q2[2], kq[2];
        for (int n = 0;; n++) {   // words and packet in one go, as in chain A; here also the stage values of u and v of the row
            asm volatile("" ::: "memory");
            const int w_rdy = *LDS_WORD(&sh.ctlB[j & 3].rdy), w_st = *LDS_WORD(&sh.ctlB[j & 3].stored);
            const int w_a = *LDS_WORD(&sh.doneA);
            asm volatile("" ::: "memory");
            cA = S.cA[lane]; H1 = S.H1[lane]; H2 = S.H2[lane]; cAr = S.cAr[lane];
#pragma unroll
            for (int m = 0; m < 2; m++) {
                q1[m] = S.q1[m][lane]; q2[m] = S.q2[m][lane];
                st[m] = S.x0[m][lane]; xn[m] = S.xn[m][lane]; cP[m] = S.cP[m][lane]; kq[m] = S.KQ[m][lane]; cBr[m] = S.cBr[m][lane];
            }
#pragma unroll
            for (int m = 0; m < 4; m++) { vs[m] = vst[m][lane]; us[m] = TOP ? 0. : ust[m][lane]; }
            asm volatile("" ::: "memory");
            if (__builtin_amdgcn_readfirstlane(w_rdy) >= j && __builtin_amdgcn_readfirstlane(w_st) >= j - 4 &&
                __builtin_amdgcn_readfirstlane(w_a) >= j) break;
            __builtin_amdgcn_s_sleep(1);
            if (n > (1 << 22)) {
                __hip_atomic_store(err, 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
        }
#pragma unroll
        for (int n = 0; n < 2; n++) {
            const double kb = window ? 0. : q2[n];   // pole-side rows: KB of the variable
            const double q2n = window ? q2[n] : 0.;
            cB[n] = __builtin_fma(H1, xm1[n], __builtin_fma(window ? H2 : 0., xm2[n], kb));
            cQ[n] = __builtin_fma(q1[n], xm1[n], __builtin_fma(q2n, xm2[n], kq[n]));
        }
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const double v = vs[s], u = us[s];
#pragma unroll
            for (int n = 0; n < 2; n++) {
                const double tv = __builtin_fma(-v, __builtin_fma(cA, st[n], cB[n]), __builtin_fma(cP[n], st[n], cQ[n]));
                rhs[n] = TOP ? tv : __builtin_fma(-u, __builtin_fma(cAr, st[n], cBr[n]), tv);
                if (s == 0) { acc[n] = rhs[n]; st[n] = __builtin_fma(rhs[n], hdt, xn[n]); }
                else if (s == 1) { acc[n] = __builtin_fma(2., rhs[n], acc[n]); st[n] = __builtin_fma(rhs[n], hdt, xn[n]); }
                // stage 3 advances w from its stage-2 value (RungeKutta_Hyd.cpp:101)
                else if (s == 2) { acc[n] = __builtin_fma(2., rhs[n], acc[n]); st[n] = __builtin_fma(rhs[n], dt, n == 1 ? st[n] : xn[n]); }
                else st[n] = __builtin_fma(acc[n] + rhs[n], dt6, xn[n]);
            }
        }
#pragma unroll
        for (int n = 0; n < 2; n++) { out[n][lane] = st[n]; xm2[n] = xm1[n]; xm1[n] = st[n]; }
        out[2][lane] = rhs[1];
        announce_in_order(&sh.doneB, j, lane);
    }
}

// Copies the boundary shell (i, j or k on a face) of the five variables from the current buffers to the alternate
// ones: the sweep writes interior points only.  Work of the `ncopy` extra workgroups at the end of the sweep's grid.
// Per variable: levels 0 and im-1 whole, then of every other level the rows j = 0, jm-1 and the columns k = 0, km-1.
__device__ void copy_shell(const PipeArgs& a, unsigned part, unsigned parts, int im, int jm, int km) {
    // (32-bit index arithmetic: the host checks that five shells are fewer than 2^31 points)
    const unsigned n2 = (unsigned)jm * km, rim = 2u * km + 2u * jm;
    const unsigned per_field = 2u * n2 + (unsigned)(im - 2) * rim;
    const unsigned total = 5u * per_field, stride = parts * blockDim.x;
    auto offset = [&](unsigned e) -> size_t {
        if (e < n2) return e;
        if (e < 2u * n2) return (size_t)(im - 1) * n2 + (e - n2);
        const unsigned r = e - 2u * n2, i = 1u + r / rim, x = r % rim;
        size_t q;
        if (x < (unsigned)km) q = x;
        else if (x < 2u * km) q = (size_t)(jm - 1) * km + (x - km);
        else if (x < 2u * km + jm) q = (size_t)(x - 2u * km) * km;
        else q = (size_t)(x - 2u * km - jm) * km + (km - 1);
        return (size_t)i * n2 + q;
    };
    for (unsigned idx = part * blockDim.x + threadIdx.x; idx < total; idx += stride) {
        const unsigned f = idx / per_field;
        const size_t o = offset(idx - f * per_field);
        a.xo[f][o] = a.x[f][o];
    }
}

// Waves of the 768-thread workgroup: w, w + 4 and w + 8 share a SIMD, and with three waves a SIMD a wave may hold 168
// vector registers (chain A holds two packets).  Wave 0 is chain A and has its SIMD to itself (waves 4 and 8 end at once);
// wave 1 is chain B.  Rows r = q mod 4 belong to pair q of the memory waves; the wave of a pair that works for chain A
// has about twice the instructions of the one that works for chain B: two of the former and one of the latter on each
// of SIMDs 2 and 3, the other two beside chain B.
constexpr int WG = 768;
// role of a wave: 0 chain A, 1 chain B, 2 + 2 q memory wave of pair q for chain A, 3 + 2 q for chain B, -1 none
__constant__ const signed char ROLE[WG / 64] = { 0, 1, 2, 4,  -1, 3, 6, 8,  -1, 5, 7, 9 };
__global__ __launch_bounds__(WG) void k_hyd_rk3d_pipe(const PipeParams* __restrict__ P, int* __restrict__ progress, int* err, int ncopy) {
    const PipeArgs& a = P->a;
    const DevConst& K = P->K;
    if (blockIdx.x >= gridDim.x - (unsigned)ncopy) {  // before anything a workgroup does together
        copy_shell(a, blockIdx.x - (gridDim.x - (unsigned)ncopy), (unsigned)ncopy, K.im, K.jm, K.km);
        return;
    }
    __shared__ PipeShared sh;
    // workgroup -> (level, longitude block): the host's table gives every XCD a chunk of consecutive levels (see the launch)
    const int gx = (K.km - 2 + 63) / 64;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int bx = idx % gx;
#ifdef ATOMDEV_EXP_HP_OLDMAP  // timing experiment only: levels dealt round robin to the XCDs
    const int lev = xcd + 8 * (idx / gx) < K.im - 2 ? xcd + 8 * (idx / gx) : -1;
#else
    // (all blocks of a level, then the next level of the chunk; the chunk's levels block by block instead - so that the
    // workgroups of levels i and i+1 of a block run side by side where a launch takes several rounds - was no faster
    // on the 0.25-degree grid)
    const int lev = P->level_of_slot[xcd * P->chunk + idx / gx];
#endif
    if (lev < 0) return;
    const int i = 1 + lev;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (threadIdx.x < NSLOT) { sh.ctlA[threadIdx.x] = { 0, 0 }; sh.ctlB[threadIdx.x] = { 0, 0 }; }
    if (threadIdx.x == 0) { sh.doneA = 0; sh.doneB = 0; }
    if (K.jm <= GEO_ROWS)
        for (int j = threadIdx.x; j < K.jm; j += (int)blockDim.x) {
            sh.geo[j][0] = a.costhe[j] * a.tab_irm2sin[(size_t)i * K.jm + j];
            sh.geo[j][1] = a.tab_coefv[(size_t)i * K.jm + j];
        }
    __syncthreads();   // the only barrier: from here on the waves follow each other's progress words
    const bool top = i == K.im - 2, feeds = i == K.im - 3;
    const int role = ROLE[wave];
    const bool fora = !(role & 1);
    if (role == 0) {
        if (top) chain_a<true>(a, K, sh, i, bx, err);
        else chain_a<false>(a, K, sh, i, bx, err);
    } else if (role == 1) {
        if (top) chain_b<true>(a, K, sh, i, bx, err);
        else chain_b<false>(a, K, sh, i, bx, err);
    } else if (role >= 2) {
        const int q = (role - 2) >> 1;
        if (K.jm <= GEO_ROWS) {
            if (fora) {
                if (top) memory_wave<true, false, true, true>(a, K, sh, q, i, bx, progress, err);
                else if (feeds) memory_wave<false, true, true, true>(a, K, sh, q, i, bx, progress, err);
                else memory_wave<false, false, true, true>(a, K, sh, q, i, bx, progress, err);
            } else {
                if (top) memory_wave<true, false, false, true>(a, K, sh, q, i, bx, progress, err);
                else memory_wave<false, false, false, true>(a, K, sh, q, i, bx, progress, err);
            }
        } else {
            if (fora) {
                if (top) memory_wave<true, false, true, false>(a, K, sh, q, i, bx, progress, err);
                else if (feeds) memory_wave<false, true, true, false>(a, K, sh, q, i, bx, progress, err);
                else memory_wave<false, false, true, false>(a, K, sh, q, i, bx, progress, err);
            } else {
                if (top) memory_wave<true, false, false, false>(a, K, sh, q, i, bx, progress, err);
                else memory_wave<false, false, false, false>(a, K, sh, q, i, bx, progress, err);
            }
        }
    }
}

}  // namespace

int atomdev_hyd_rk3d_pipe(atomdev* h) {
    static const int X[5] = { ATOMDEV_V, ATOMDEV_U, ATOMDEV_C, ATOMDEV_T, ATOMDEV_W };
    static const int XN[5] = { ATOMDEV_VN, ATOMDEV_UN, ATOMDEV_CN, ATOMDEV_TN, ATOMDEV_WN };
    ATOMDEV_REQUIRE(h->n3 * sizeof(double) < ((size_t)1 << 32), "hyd_rk3d: field larger than 4 GiB");
    ATOMDEV_REQUIRE(h->im >= 5 && h->jm >= 3 && h->km >= 3, "hyd_rk3d: the grid needs at least 5 levels and interior points");
    const unsigned gx = (h->km - 2 + 63) / 64;
    ATOMDEV_REQUIRE(gx <= 128, "hyd_rk3d: more than 128 longitude blocks");
    const int nlev = h->im - 2;
    ATOMDEV_REQUIRE(8 * ((nlev + 7) / 8) <= MAX_SLOTS, "hyd_rk3d: more than 256 levels");
    // Two argument blocks, one per orientation of the (current, alternate) buffer pair; the sweep swaps the pair.
    struct Blocks { PipeParams p[2]; int progress[4 * 128]; };
    if (!h->pipe_params) {
        Blocks* B = new Blocks();
        for (int o = 0; o < 2; o++) {
            PipeArgs& a = B->p[o].a;
            for (int n = 0; n < 5; n++) {
                a.x[n] = o ? h->alt[X[n]] : h->f[X[n]];
                a.xo[n] = o ? h->f[X[n]] : h->alt[X[n]];
                a.xn[n] = h->f[XN[n]];
            }
            a.p_dyn = h->f[ATOMDEV_P_DYN];
            a.r_water = h->f[ATOMDEV_R_WATER]; a.r_salt_water = h->f[ATOMDEV_R_SALT_WATER];
            a.aux_u = h->f[ATOMDEV_AUX_U]; a.aux_v = h->f[ATOMDEV_AUX_V]; a.aux_w = h->f[ATOMDEV_AUX_W]; a.buoy = h->f[ATOMDEV_BUOYANCY];
            a.salt_balance = h->f[ATOMDEV_SALT_BALANCE]; a.salt_finger = h->f[ATOMDEV_SALT_FINGER];
            a.salt_diffusion = h->f[ATOMDEV_SALT_DIFFUSION];
            a.flags = h->flags;
            a.bathy = h->g[ATOMDEV_TOPOGRAPHY]; a.evap = h->g[ATOMDEV_EVAPORATION_DALTON]; a.precip = h->g[ATOMDEV_PRECIPITATION];
            a.rad = h->rad; a.costhe = h->costhe; a.tab_irm2sin = h->tab_irm2sin; a.tab_coefv = h->tab_coefv;
            B->p[o].K = h->k;
            // Levels to XCDs (workgroup n runs on XCD n % 8): consecutive levels in chunks, so that the rows of level i+1 a
            // workgroup reads are mostly rows the workgroup of level i+1 has just brought into the same L2.  Levels im-3
            // and im-2 hand rows over through memory and do that faster from different XCDs: level im-3 changes places
            // with the last level of the chunk before; level im-2 takes the last slot of its XCD, so that its workgroups
            // come after those of level im-3 in dispatch order.
            const int chunk = (nlev + 7) / 8;
            int* T = B->p[o].level_of_slot;
            for (int s = 0; s < MAX_SLOTS; s++) T[s] = s < nlev ? s : -1;
            const int lp = nlev - 2, lt = nlev - 1;
            if (lp / chunk == lt / chunk && lt / chunk >= 1) {
                const int q = (lt / chunk) * chunk - 1;
                T[q] = lp; T[lp] = q;
            }
            const int last = (lt / chunk) * chunk + chunk - 1;
            if (last != lt) { T[last] = lt; T[lt] = -1; }
            B->p[o].chunk = chunk;
        }
        hipError_t e = hipMalloc(&h->pipe_params, sizeof(Blocks));
        if (e == hipSuccess) e = hipMemcpy(h->pipe_params, B, sizeof(Blocks), hipMemcpyHostToDevice);
        h->pipe_cur0 = h->f[ATOMDEV_T];
        delete B;
        ATOMDEV_TRY(e);
    }
    const int o = (h->f[ATOMDEV_T] == h->pipe_cur0) ? 0 : 1;
    Blocks* D = (Blocks*)h->pipe_params;
    if (int rc = atomdev_prof_begin(h)) return rc;
    const size_t shell = (size_t)5 * (2 * h->n2 + (size_t)(h->im - 2) * 2 * (h->km + h->jm));
    ATOMDEV_REQUIRE(shell + (size_t)256 * WG < ((size_t)1 << 31), "hyd_rk3d: boundary shell too large for 32-bit indices");
    const unsigned chunk = (h->im - 2 + 7) / 8;
    const unsigned sweep_blocks = 8 * chunk * gx;
    // the shell copy: about 32 points a thread, but on a grid that fits the chip in one go no more workgroups than the
    // sweep leaves compute units for (a workgroup of either kind takes a whole compute unit)
    int cus = 0;
    ATOMDEV_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device));
    unsigned ncopy = (unsigned)std::min<size_t>(256, std::max<size_t>(16, shell / ((size_t)WG * 32)));
    if (sweep_blocks < (unsigned)cus && sweep_blocks + ncopy > (unsigned)cus) ncopy = std::max(8u, (unsigned)cus - sweep_blocks);
#ifdef ATOMDEV_EXP_HP_NOSHELL  // timing experiment only (results are wrong after the first sweep)
    ncopy = 0;
#endif
    hipLaunchKernelGGL(k_hyd_rk3d_pipe, dim3(sweep_blocks + ncopy), dim3(WG), 0, h->stream, (const PipeParams*)&D->p[o], D->progress,
                       (int*)atomdev_kernel_error_word(h), (int)ncopy);
    ATOMDEV_TRY(hipGetLastError());
    if (int rc = atomdev_prof_end(h)) return rc;
    static const int S[5] = { ATOMDEV_T, ATOMDEV_U, ATOMDEV_V, ATOMDEV_W, ATOMDEV_C };
    for (int n = 0; n < 5; n++) { double* t = h->f[S[n]]; h->f[S[n]] = h->alt[S[n]]; h->alt[S[n]] = t; }
    return 0;
}
