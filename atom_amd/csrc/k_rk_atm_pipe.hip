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
// nothing else, and what remains on the recurrence is a handful of fused multiply-adds per variable and stage.
//
// Mapping.  A workgroup owns 64 neighbouring theta lines of one level.  Of its sixteen waves ten live:
//   eight "memory" waves in four pairs, taking the rows in turn (row r belongs to pair r mod 4; one wave of the pair
//       works for chain A, the other for chain B): a wave loads what its half of the row needs (old values, base values,
//       sources, flags, four rows of p_dyn), turns it into the row's coefficient packet in a ring of four LDS slots,
//       stores the results of the row that stood in the slot before (which the chain wave left in LDS), and issues the
//       loads of its next row, four rows on, before it rests - so a load has three rows' time to arrive;
//   chain wave A carries v, u, c, t (the coupled ones: v feeds all, u feeds v, c feeds u, t feeds c on level 1)
//       through the four stages of row j and leaves the four stage values of v in LDS;
//   chain wave B carries w, cloud, ice, co2 behind A, with the stage values of v that A left.
// The chain waves touch LDS only: a compute unit's vector-memory pipe takes about 11 bytes a cycle, the memory
// waves keep it full, and a store issued by a chain wave would queue behind a row's worth of their loads.
// No barrier inside the loop: the waves follow each other's progress words in LDS (which row's packet stands in a
// slot, which row a chain has finished, which row's results have been stored), so each runs at its own pace and the
// sweep goes at the pace of the slowest on average, not of the slowest in every row.  Waves w, w+4, w+8, w+12 share a
// SIMD; the six that would share the chain waves' SIMDs end at once.
#include "atomdev_internal.h"

namespace {

enum Var { V_T = 0, V_U = 1, V_V = 2, V_W = 3, V_C = 4, V_CLOUD = 5, V_ICE = 6, V_CO2 = 7 };

struct PipeArgs {
    double* x[8];
    const double* xn[8];
    const double* p_dyn;
    const double *S_v, *S_c, *S_i, *S_r, *S_s;
    double *aux_u, *aux_v, *aux_w, *buoy;
    const uint8_t* flags;
    const double *evap, *precip, *c3, *topo;   // 2-D
    const double* ph100;                       // 3-D, from the preparing kernel of k_rk_atm_ws.hip (level 1 is read)
    const double *rad, *costhe, *tab_irm2sin, *tab_coefv;
};
struct PipeParams {
    PipeArgs a;
    DevConst K;
};

// One row's packet: 64 lanes wide, everything a chain wave needs of the row besides its own last two results; one
// half per chain wave.  On the three pole-side rows (central differences: x[j-2] does not enter, old x[j+1] does)
// the arrays of the x[j-2] factors carry the constant terms KB of the four variables instead: q2[0..2] and, for
// chain A, qu2, for chain B, H2.
enum { QA_RE = 0, QA_WV = 1, QA_PR = 2 };     // diffusion classes of chain A's variables: u and v, c, t
enum { QB_RE = 0, QB_WV = 1, QB_CO2 = 2 };    // ... and of chain B's: w, cloud and ice, co2
struct HalfB {
    double cA[64], H1[64], H2[64];
    double q1[3][64], q2[3][64];
    double cP[3][64];                          // w, cloud = ice, co2
    double x0[4][64], xn[4][64], KQ[4][64];    // w, cloud, ice, co2 (KQ of w and co2: pole-side rows only)
};
struct HalfA {
    double cA[64], H1[64], H2[64];
    double q1[3][64], q2[3][64];
    double cP[4][64];                          // v, u, c, t
    double x0[4][64], xn[4][64], KQ[4][64];    // v, u, c, t (KQ of u: pole-side rows only)
    double cR[64], qu1[64], qu2[64], pgv[64];
    double cFac[64], cMc[64], phk[64];         // level 1: the evaporation source of c
};
constexpr int NSLOT = 4;
constexpr int GEO_ROWS = 768;                  // grids with more rows fetch the geometry with the row's other loads
struct Ctl { int rdy, stored; };               // per slot and half: the row whose packet stands there, the row last stored from there
struct PipeShared {
    HalfA ha[NSLOT];
    HalfB hb[NSLOT];
    double outA[NSLOT][7][64];                 // v, u, c, t, aux_u, aux_v, buoyancy of a finished row
    double outB[NSLOT][5][64];                 // w, cloud, ice, co2, aux_w
    double vst[2][4][64];                      // stage values of v of a row, by row parity
    double geo[GEO_ROWS][2];                   // the level's geometry by row: cos(the) / (rm^2 sin(the)), the viscous v coefficient
    Ctl ctlA[NSLOT], ctlB[NSLOT];
    int doneA, doneB;                          // the last row each chain wave has finished
#ifdef ATOMDEV_EXP_PIPE_STAMP
    unsigned long long stamps[10][4][8];
    unsigned long long wall[2], cyc[2];
#endif
};

#ifdef ATOMDEV_EXP_PIPE_STAMP  // timing experiment only: cycle stamps of four rows of one workgroup, kept in LDS
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
// The same with the non-temporal hint, for what a step touches once (the old-time-level fields and the source terms read,
// the side outputs written): the last-level cache then keeps the fields the next kernels read again.
__device__ __forceinline__ double gld_once(const double* base, unsigned off) {
    return __builtin_nontemporal_load((const __attribute__((address_space(1))) double*)((gcchar_t)base + off));
}
__device__ __forceinline__ void gst_once(double* base, unsigned off, double x) {
    __builtin_nontemporal_store(x, (__attribute__((address_space(1))) double*)((gchar_t)base + off));
}
#define PIN(x) asm volatile("" : "+v"(x))
// A constant of the argument block as a value in scalar registers: without this the compiler may turn a per-lane choice
// between two such constants into a per-lane choice of ADDRESS and a vector load in the middle of the loop.
#define SPIN(x) asm volatile("" : "+s"(x))

typedef volatile __attribute__((address_space(3))) int* lds_word_t;   // (typed as LDS: a generic pointer would go through the vector-memory pipe)
#define LDS_WORD(p) ((lds_word_t)(p))
// Progress words in LDS.  A wave's LDS operations are carried out in the order it issues them, so a word written after
// the data it announces is seen after that data by whoever reads the word first and the data afterwards; the compiler
// barriers keep the compiler from moving an access across the word.  Only lane 0 writes; every lane reads (one address:
// a broadcast).  A wait gives up after about a second and says so in *err (page-locked host memory that the library's
// next synchronising call looks at), so that every wave reaches the end of the kernel whatever happens.
__device__ __forceinline__ void announce(int* word, int row, int lane) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) *LDS_WORD(word) = row;
    asm volatile("" ::: "memory");
}
// the same without waiting for the wave's earlier LDS operations to complete: they are carried out before the word's write
__device__ __forceinline__ void announce_in_order(int* word, int row, int lane) {
    asm volatile("" ::: "memory");
    if (lane == 0) *LDS_WORD(word) = row;
    asm volatile("" ::: "memory");
}
#ifdef ATOMDEV_EXP_PIPE_COUNT  // experiment only: how often each kind of wave found its input not there yet
#define PCOUNT(x) ((x)++)
#else
#define PCOUNT(x) do {} while (0)
#endif
__device__ __forceinline__ void wait_for(const int* word, int need, int* err, int& spins) {
    asm volatile("" ::: "memory");
    for (int n = 0; __builtin_amdgcn_readfirstlane(*LDS_WORD(word)) < need; n++) {
        PCOUNT(spins);
        __builtin_amdgcn_s_sleep(2);
        if (n > (1 << 22)) {
            __hip_atomic_store(err, 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            break;
        }
    }
    asm volatile("" ::: "memory");
}
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

// ---- a memory wave: rows r = q, q + 4, ... (r >= 1), the half of the row that chain A (FORA) or chain B works on ----
template <bool L1, bool FORA, bool LIMIT>
__device__ __forceinline__ void memory_wave(const PipeArgs& a, const DevConst& K, PipeShared& sh, const int q, const int i, const int bx, int* err) {
    const bool geo_lds = K.jm <= GEO_ROWS;
    const int jm = K.jm, km = K.km;
    const int lane = threadIdx.x & 63;
    const int kk = 1 + bx * 64 + lane;
    const int k = kk <= km - 2 ? kk : km - 2;  // idle lanes shadow the last line
    const int NR = jm - 2;
    const unsigned rowB = (unsigned)km * 8u;
    unsigned o_lev = (unsigned)(((size_t)i * jm * km + k) * 8);   // row 0 of the line, as a byte offset into a 3-D field
    unsigned o_sfc = (unsigned)k * 8u;                             // ... into a 2-D field
    PIN(o_lev); PIN(o_sfc);
    HalfA& SA = sh.ha[q];
    HalfB& SB = sh.hb[q];
    Ctl* ctl = FORA ? &sh.ctlA[q] : &sh.ctlB[q];
    const int* done = FORA ? &sh.doneA : &sh.doneB;
    constexpr int VARS[4] = { FORA ? V_V : V_W, FORA ? V_U : V_CLOUD, FORA ? V_C : V_ICE, FORA ? V_T : V_CO2 };
    const int stamp_wave = 2 + 2 * q + (FORA ? 0 : 1);
    (void)stamp_wave;

    const double rm = a.rad[i];
    const double i_rm = 1. / rm, i_rm2 = 1. / (rm * rm);
    const double ce = K.coeff_energy, ctr = K.coeff_trans;
    const double i_phk = 1. / (K.R_Air * K.t_0);
    const double hight = (double)i * K.topo_step;
    double topo_step = K.topo_step, i_topo_step = K.i_topo_step;
    SPIN(topo_step); SPIN(i_topo_step);
    const double evf = K.coeff_vapour * K.i_two_dr;
    double i_dthe = K.i_dthe, i_two_dthe = K.i_two_dthe, i_dthe2 = K.i_dthe2, i_dr2 = K.i_dr2, i_dphi2 = K.i_dphi2;
    double i_re = K.i_re, i_re_sc_wv = K.i_re_sc_wv, i_re_x = FORA ? K.i_re_pr : K.i_re_sc_co2, i_r_air = K.i_r_air;
    SPIN(i_dthe); SPIN(i_two_dthe); SPIN(i_dthe2); SPIN(i_dr2); SPIN(i_dphi2); SPIN(i_re); SPIN(i_re_sc_wv); SPIN(i_re_x); SPIN(i_r_air);
    const double two = 2. * i_rm2 * i_re;
    // diffusion divisors of this half's classes and variables
    const double idc[3] = { i_re, i_re_sc_wv, i_re_x };
    constexpr int CLS[4] = { 0, FORA ? 0 : 1, 1, 2 };   // A: v, u -> re; c -> wv; t -> pr.  B: w -> re; cloud, ice -> wv; co2

    // what a row's loads leave behind until the row's turn comes
    double lx[4], lxn[4], lxp[4], lS[5], ltopo = 0., lp[4];
    double lev = 0., lpr = 0., lc0 = 0., lc3 = 0., lph = 0.;
    double lcos = 0., lirs = 0., lcoefv = 0.;  // geometry of the row where the level's table is not in LDS: one address for the whole wave
    unsigned lfl;
    auto issue = [&](const int r) {
#ifdef ATOMDEV_EXP_PIPE_NOMEM  // timing experiment only (results are wrong): no vector-memory traffic inside the loop
        if (r > 8) return;
#endif
        const unsigned o = o_lev + (unsigned)r * rowB;
        lfl = *(const __attribute__((address_space(1))) uint8_t*)((gcchar_t)a.flags + (o >> 3));
#pragma unroll
        for (int n = 0; n < 4; n++) { lx[n] = gld(a.x[VARS[n]], o); lxn[n] = gld_once(a.xn[VARS[n]], o); }
        if (!geo_lds) { lcos = gld(a.costhe + r, 0u); lirs = gld(a.tab_irm2sin + (size_t)i * jm + r, 0u); }
        ltopo = gld(a.topo, o_sfc + (unsigned)r * rowB);   // (a 2-D field: stays in the L2)
        if constexpr (FORA) {
            lS[0] = gld_once(a.S_c, o); lS[1] = gld_once(a.S_r, o); lS[2] = gld_once(a.S_i, o); lS[3] = gld_once(a.S_s, o); lS[4] = gld_once(a.S_v, o);
            if (!geo_lds) lcoefv = gld(a.tab_coefv + (size_t)i * jm + r, 0u);
            lp[0] = r >= 2 ? gld(a.p_dyn, o - 2u * rowB) : 0.;
            lp[1] = gld(a.p_dyn, o - rowB);
            lp[2] = gld(a.p_dyn, o);
            lp[3] = gld(a.p_dyn, o + rowB);
        } else {
            lS[0] = gld_once(a.S_c, o); lS[2] = gld_once(a.S_i, o);
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
    // The results of row `rs`, which the chain wave left in the slot's output arrays: limited as the limiter that follows
    // the sweep in the model's loop would leave them (LIMIT), side outputs zero on land (RHS_Atm.cpp:406-410, 485-492).
    auto store_row = [&](const int rs, const bool land) {
#ifdef ATOMDEV_EXP_PIPE_NOMEM
        if (rs > 8) { announce(&ctl->stored, rs, lane); return; }
#endif
        const unsigned o = o_lev + (unsigned)rs * rowB;
        if constexpr (FORA) {
            const double (*out)[64] = sh.outA[q];
            const double v = out[0][lane], u = out[1][lane], c = out[2][lane], t = out[3][lane];
            const double au = out[4][lane], av = out[5][lane], bu = out[6][lane];
            gst(a.x[V_V], o, LIMIT ? limited<V_V>(v, land) : v);
            gst(a.x[V_U], o, LIMIT ? limited<V_U>(u, land) : u);
            gst(a.x[V_C], o, LIMIT ? limited<V_C>(c, land) : c);
            gst(a.x[V_T], o, LIMIT ? limited<V_T>(t, land) : t);
            gst_once(a.aux_u, o, land ? 0. : au);
            gst_once(a.aux_v, o, land ? 0. : av);
            gst_once(a.buoy, o, land ? 0. : bu);
        } else {
            const double (*out)[64] = sh.outB[q];
            const double w = out[0][lane], cl = out[1][lane], ic = out[2][lane], co = out[3][lane], aw = out[4][lane];
            gst(a.x[V_W], o, LIMIT ? limited<V_W>(w, land) : w);
            gst(a.x[V_CLOUD], o, LIMIT ? limited<V_CLOUD>(cl, land) : cl);
            gst(a.x[V_ICE], o, LIMIT ? limited<V_ICE>(ic, land) : ic);
            gst(a.x[V_CO2], o, LIMIT ? limited<V_CO2>(co, land) : co);
            gst_once(a.aux_w, o, land ? 0. : aw);
        }
        announce(&ctl->stored, rs, lane);   // (waits until the values have left LDS)
    };
    const int first = q == 0 ? 4 : q;
    if (first <= NR) issue(first);
    bool land_prev = false;
    int r_last = 0, spins = 0;

    for (int r = first; r <= NR; r += 4) {
        const int e = r;
        PSTAMP(stamp_wave, 0);
        // The row's packet is worked out in registers first (its loads were issued a turn ago); only writing it needs the
        // slot, which is free once the chain wave has finished the row that stood in it.  That row's results go to memory
        // after the packet has been announced: the chain wave looks at `stored` only when it is about to write results.
        {
            const bool window = r >= 2 && r < jm - 3;
            const double g = geo_lds ? sh.geo[r][0] : lcos * lirs;
            const double coefv = geo_lds ? sh.geo[r][1] : lcoefv;
            // the mask: RHS_Atm.cpp:110-149
            const unsigned fl = lfl;
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
            // h_0_i (RHS_Atm.cpp:110-124): 1 over fluid, the relative height above the ground over land
            double h0i = 1.;
            if (land) {
                const double topo_diff = hight - ltopo;
                if (!(topo_diff == topo_step)) h0i = div_by(topo_diff, topo_step, i_topo_step);
            }
            const double dmp_i = h0i * i_dr2;
            double KQ[4], KB[4] = { 0., 0., 0., 0. };
            double o_cA, o_H1, o_H2, o_q1[3], o_q2[3], o_cP[4] = { 0., 0., 0., 0. }, o_pgv = 0., o_cR = 0., o_qu1 = 0., o_qu2 = 0., o_cFac = 0., o_cMc = 0., o_phk = 0.;
            if (!window) {  // b2 = f2 = 0 on these rows: the x[j-2] arrays are free for KB
                const double H3 = b3 * hDr;
#pragma unroll
                for (int n = 0; n < 4; n++) KB[n] = H3 * lxp[n];
            }
            if constexpr (FORA) {
                o_cA = A1 * i_rm; o_H1 = b1 * hDr; o_H2 = b2 * hDr;
#pragma unroll
                for (int c = 0; c < 3; c++) { o_q1[c] = G1 * idc[c]; o_q2[c] = window ? G2 * idc[c] : KB[c]; }
                // factor of the stage value outside the advection product: diffusion bracket and damping
                const double p_re = Pb * i_re;
                const double h0j = (fl & FL_H0J) ? 1. : 0.;
                o_cP[0] = p_re - coefv * hdj * i_re - h0j * i_dthe2;
                o_cP[1] = p_re - dmp_i;
                o_cP[2] = Pb * i_re_sc_wv;
                o_cP[3] = Pb * i_re_x;
                // pressure gradient of the v equation (p_dyn is not changed by the sweep)
                const double dp = hD * (fa * lp[2] + b1 * lp[1] + b2 * lp[0] + b3 * lp[3]);
                const double pgv = hdj * dp * i_rm * i_r_air;
                o_pgv = pgv;
                const double twoD = two * hD;
                o_cR = two * A1; o_qu1 = twoD * b1; o_qu2 = window ? twoD * b2 : KB[3];
                // constant terms: sources, and on the pole-side rows the old row j+1
                KQ[0] = -pgv; KQ[1] = 0.; KQ[2] = lS[4] * ctr; KQ[3] = ce * (lS[0] + lS[1]) + ce * (lS[2] + lS[3]);
                if (!window) KQ[0] = __builtin_fma(twoD * b3, lxp[1], KQ[0]);
                if constexpr (L1) {  // RHS_Atm.cpp:412-430
                    double ep = lev - lpr;
                    if (ep >= 6.) ep = 6.;
                    if (ep <= -6.) ep = -6.;
                    o_cFac = land ? 0. : evf * (1. - 2. * lc0) * ep;
                    o_cMc = -3. * lc0 - lc3;
                    o_phk = lph * i_phk;
                }
            } else {
                o_cA = A1 * i_rm; o_H1 = b1 * hDr; o_H2 = window ? b2 * hDr : KB[3];
#pragma unroll
                for (int c = 0; c < 3; c++) { o_q1[c] = G1 * idc[c]; o_q2[c] = window ? G2 * idc[c] : KB[c]; }
                const double h0k = (fl & FL_H0K) ? 1. : 0.;
                const double p_wv = Pb * i_re_sc_wv;
                o_cP[0] = __builtin_fma(-h0k, i_dphi2, Pb * i_re);
                o_cP[1] = p_wv - dmp_i;
                o_cP[2] = __builtin_fma(Pb, i_re_x, -dmp_i);
                KQ[0] = 0.; KQ[1] = lS[0] * ctr; KQ[2] = lS[2] * ctr; KQ[3] = 0.;
            }
            if (!window) {
                const double G3 = __builtin_fma(f3, hEr, b3 * hDg);
#pragma unroll
                for (int n = 0; n < 4; n++) KQ[n] = __builtin_fma(G3 * idc[CLS[n]], lxp[n], KQ[n]);
            }
            // (kept from moving below the wait)
            PIN(o_cA); PIN(o_H1); PIN(o_H2); PIN(o_cP[0]); PIN(o_cP[1]); PIN(o_cP[2]);
#pragma unroll
            for (int c = 0; c < 3; c++) { PIN(o_q1[c]); PIN(o_q2[c]); }
#pragma unroll
            for (int n = 0; n < 4; n++) PIN(KQ[n]);
            if constexpr (FORA) { PIN(o_cP[3]); PIN(o_pgv); PIN(o_cR); PIN(o_qu1); PIN(o_qu2); }
            PSTAMP(stamp_wave, 1);
            wait_for(done, r - 4, err, spins);
            if constexpr (FORA) {
                SA.cA[lane] = o_cA; SA.H1[lane] = o_H1; SA.H2[lane] = o_H2; SA.cR[lane] = o_cR; SA.pgv[lane] = o_pgv;
                SA.qu1[lane] = o_qu1; SA.qu2[lane] = o_qu2;
#pragma unroll
                for (int c = 0; c < 3; c++) { SA.q1[c][lane] = o_q1[c]; SA.q2[c][lane] = o_q2[c]; }
#pragma unroll
                for (int n = 0; n < 4; n++) SA.cP[n][lane] = o_cP[n];
                if constexpr (L1) { SA.cFac[lane] = o_cFac; SA.cMc[lane] = o_cMc; SA.phk[lane] = o_phk; }
            } else {
                SB.cA[lane] = o_cA; SB.H1[lane] = o_H1; SB.H2[lane] = o_H2;
#pragma unroll
                for (int c = 0; c < 3; c++) { SB.q1[c][lane] = o_q1[c]; SB.q2[c][lane] = o_q2[c]; SB.cP[c][lane] = o_cP[c]; }
            }
            double (*x0)[64] = FORA ? SA.x0 : SB.x0, (*xn)[64] = FORA ? SA.xn : SB.xn, (*kq)[64] = FORA ? SA.KQ : SB.KQ;
#pragma unroll
            for (int n = 0; n < 4; n++) {
                x0[n][lane] = lx[n]; xn[n][lane] = lxn[n];
                const bool zero_in_window = FORA ? n == 1 : (n == 0 || n == 3);   // u; w, co2: no source term
                if (!window || !zero_in_window) kq[n][lane] = KQ[n];
            }
            PSTAMP(stamp_wave, 2);
            announce(&ctl->rdy, r, lane);
            if (r > 4) store_row(r - 4, land_prev);
            land_prev = land;
            r_last = r;
            if (r + 4 <= NR) issue(r + 4);
            PSTAMP(stamp_wave, 3);
        }
    }
    if (r_last) {  // the wave's last row
        wait_for(done, r_last, err, spins);
        store_row(r_last, land_prev);
    }
#ifdef ATOMDEV_EXP_PIPE_COUNT
    if (lane == 0 && bx == 2 && i == ATOMDEV_EXP_PIPE_COUNT) a.aux_w[16 + 2 * q + (FORA ? 0 : 1)] = (double)spins;
#endif
}

// ---- chain wave A: v, u, c, t ------------------------------------------------------------------------------------
template <bool L1>
__device__ __forceinline__ void chain_a(const PipeArgs& a, const DevConst& K, PipeShared& sh, const int i, const int bx, int* err) {
    const int jm = K.jm, km = K.km;
    const int lane = threadIdx.x & 63;
    const int kk = 1 + bx * 64 + lane;
    const int k = kk <= km - 2 ? kk : km - 2;
    const int NR = jm - 2;
    const unsigned o_lev = (unsigned)(((size_t)i * jm * km + k) * 8);
    constexpr int VARS[4] = { V_V, V_U, V_C, V_T };
    constexpr int CLS[4] = { QA_RE, QA_RE, QA_WV, QA_PR };
    enum { cV = 0, cU = 1, cC = 2, cT = 3 };

    const double kWV = K.R_WV_over_R_Air, KK = K.Buoyancy * K.g, hdt = .5 * K.dt, dt = K.dt, dt6 = K.dt * K.i_six;
    const double cb1000 = K.coeff_buoy * 1000.;
    double xm1[4], xm2[4] = { 0., 0., 0., 0. };
#pragma unroll
    for (int n = 0; n < 4; n++) xm1[n] = gld(a.x[VARS[n]], o_lev);
    int retries = 0, why[3] = { 0, 0, 0 };
    (void)retries; (void)why;

    for (int j = 1; j <= NR; j++) {
        const int e = j;
        PSTAMP(0, 0);
        // the row's packet; the buffer for the stage values of v free (chain B done with row j-2)
        const bool window = j >= 2 && j < jm - 3;
        const HalfA& S = sh.ha[j & 3];
        double (*vst)[64] = sh.vst[j & 1];
        double (*out)[64] = sh.outA[j & 3];
        double cA, H1, H2, cR, pgv, qu1, qu2, cFac = 0., cMc = 0., phk = 0.;
        int stored_seen = 0;
        double st[4], xn[4], cB[4], cQ[4], cP[4], acc[4], rhs[4], q1[3], q2[3], kq[4];
        // The words and the packet are read in one go: the words first, so a packet read behind words that say "there" is
        // the packet (a wave's LDS reads are carried out in order); if the words say "not yet", the words are waited for and
        // everything is read again.
        for (int n = 0;; n++) {
            asm volatile("" ::: "memory");
            const int w_rdy = *LDS_WORD(&sh.ctlA[j & 3].rdy), w_st = *LDS_WORD(&sh.ctlA[j & 3].stored);
            const int w_b = *LDS_WORD(&sh.doneB);
            asm volatile("" ::: "memory");
            cA = S.cA[lane]; H1 = S.H1[lane]; H2 = S.H2[lane]; cR = S.cR[lane]; pgv = S.pgv[lane]; qu1 = S.qu1[lane]; qu2 = S.qu2[lane];
#pragma unroll
            for (int c = 0; c < 3; c++) { q1[c] = S.q1[c][lane]; q2[c] = S.q2[c][lane]; }
#pragma unroll
            for (int m = 0; m < 4; m++) { st[m] = S.x0[m][lane]; xn[m] = S.xn[m][lane]; cP[m] = S.cP[m][lane]; kq[m] = S.KQ[m][lane]; }
            if constexpr (L1) { cFac = S.cFac[lane]; cMc = S.cMc[lane]; phk = S.phk[lane]; }
            asm volatile("" ::: "memory");
            stored_seen = __builtin_amdgcn_readfirstlane(w_st);
            if (__builtin_amdgcn_readfirstlane(w_rdy) >= j && __builtin_amdgcn_readfirstlane(w_b) >= j - 2) break;
            PCOUNT(retries);
#ifdef ATOMDEV_EXP_PIPE_COUNT
            if (__builtin_amdgcn_readfirstlane(w_rdy) < j) why[0]++; else why[2]++;
#endif
            // not there: the words alone are polled (the LDS stays free for the other waves), then everything is read again
            wait_for(&sh.ctlA[j & 3].rdy, j, err, why[0]);
            wait_for(&sh.doneB, j - 2, err, why[2]);
            if (n > 2) {
                __hip_atomic_store(err, 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
        }
        PSTAMP(0, 1);
#pragma unroll
        for (int n = 0; n < 4; n++) {
            const double alias = n < 3 ? q2[n] : qu2;   // pole-side rows: KB of the variable
            const double kb = window ? 0. : alias;
            const double q2n = window ? q2[CLS[n]] : 0.;
            const double kqn = (window && n == cU) ? 0. : kq[n];
            cB[n] = __builtin_fma(H1, xm1[n], __builtin_fma(window ? H2 : 0., xm2[n], kb));
            cQ[n] = __builtin_fma(q1[CLS[n]], xm1[n], __builtin_fma(q2n, xm2[n], kqn));
        }
        cQ[cV] = __builtin_fma(qu1, xm1[cU], __builtin_fma(window ? qu2 : 0., xm2[cU], cQ[cV]));  // 2 du/dthe / rm^2 / re
        double RS = 0.;
#ifdef ATOMDEV_EXP_PIPE_NOCHAIN  // timing experiment only (results are wrong): the recurrence without its arithmetic
#pragma unroll
        for (int n = 0; n < 4; n++) { rhs[n] = cB[n] + cQ[n] + cP[n]; st[n] += xn[n]; }
        if (j < 0)
#endif
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
        // results and the side outputs of the fourth evaluation, for the memory wave of the slot to store; the slot's
        // output arrays are free once row j-4 has been stored (looked at here, not at the top: the memory wave stores a
        // row after it has announced the packet of the row four on)
        if (stored_seen < j - 4) { wait_for(&sh.ctlA[j & 3].stored, j - 4, err, why[1]); }
#pragma unroll
        for (int n = 0; n < 4; n++) { out[n][lane] = st[n]; xm2[n] = xm1[n]; xm1[n] = st[n]; }
        out[4][lane] = rhs[cU]; out[5][lane] = rhs[cV] + pgv; out[6][lane] = -RS * cb1000;
        announce_in_order(&sh.doneA, j, lane);
        PSTAMP(0, 3);
    }
#ifdef ATOMDEV_EXP_PIPE_COUNT
    if (lane == 0 && bx == 2 && i == ATOMDEV_EXP_PIPE_COUNT) { a.aux_w[0] = retries; a.aux_w[1] = why[0]; a.aux_w[2] = why[1]; a.aux_w[3] = why[2]; }
#endif
}

// ---- chain wave B: w, cloud, ice, co2, behind A --------------------------------------------------------------------
__device__ __forceinline__ void chain_b(const PipeArgs& a, const DevConst& K, PipeShared& sh, const int i, const int bx, int* err) {
    const int jm = K.jm, km = K.km;
    const int lane = threadIdx.x & 63;
    const int kk = 1 + bx * 64 + lane;
    const int k = kk <= km - 2 ? kk : km - 2;
    const int NR = jm - 2;
    const unsigned o_lev = (unsigned)(((size_t)i * jm * km + k) * 8);
    constexpr int VARS[4] = { V_W, V_CLOUD, V_ICE, V_CO2 };
    constexpr int CLS[4] = { QB_RE, QB_WV, QB_WV, QB_CO2 };
    constexpr int CPS[4] = { 0, 1, 1, 2 };

    const double hdt = .5 * K.dt, dt = K.dt, dt6 = K.dt * K.i_six;
    double xm1[4], xm2[4] = { 0., 0., 0., 0. };
#pragma unroll
    for (int n = 0; n < 4; n++) xm1[n] = gld(a.x[VARS[n]], o_lev);
    int retries = 0, why[3] = { 0, 0, 0 };
    (void)retries; (void)why;

    for (int j = 1; j <= NR; j++) {
        const int e = j;
        PSTAMP(1, 0);
        const bool window = j >= 2 && j < jm - 3;
        const HalfB& S = sh.hb[j & 3];
        const double (*vst)[64] = sh.vst[j & 1];
        double (*out)[64] = sh.outB[j & 3];
        double cA, H1, H2, vs[4];
        int stored_seen = 0;
        double st[4], xn[4], cB[4], cQ[4], cP[4], acc[4], rhs[4], q1[3], q2[3], kq[4];
        // The packet is nearly always there long before chain A has done the row: it is read together with its word (and
        // again if the word says "not yet"); then chain A's word alone is polled, and the four stage values of v are read
        // once it says "done".  (Reading the whole packet at every poll would keep the LDS busy while chain A needs it.)
        for (int n = 0;; n++) {
            asm volatile("" ::: "memory");
            const int w_rdy = *LDS_WORD(&sh.ctlB[j & 3].rdy), w_st = *LDS_WORD(&sh.ctlB[j & 3].stored);
            asm volatile("" ::: "memory");
            cA = S.cA[lane]; H1 = S.H1[lane]; H2 = S.H2[lane];
#pragma unroll
            for (int c = 0; c < 3; c++) { q1[c] = S.q1[c][lane]; q2[c] = S.q2[c][lane]; }
#pragma unroll
            for (int m = 0; m < 4; m++) { st[m] = S.x0[m][lane]; xn[m] = S.xn[m][lane]; cP[m] = S.cP[CPS[m]][lane]; kq[m] = S.KQ[m][lane]; }
            asm volatile("" ::: "memory");
            stored_seen = __builtin_amdgcn_readfirstlane(w_st);
            if (__builtin_amdgcn_readfirstlane(w_rdy) >= j) break;
            PCOUNT(retries); PCOUNT(why[0]);
            __builtin_amdgcn_s_sleep(1);
            if (n > (1 << 22)) {
                __hip_atomic_store(err, 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
        }
        wait_for(&sh.doneA, j, err, why[2]);
#pragma unroll
        for (int m = 0; m < 4; m++) vs[m] = vst[m][lane];
        PSTAMP(1, 1);
#pragma unroll
        for (int n = 0; n < 4; n++) {
            const double alias = n < 3 ? q2[n] : H2;   // pole-side rows: KB of the variable
            const double kb = window ? 0. : alias;
            const double q2n = window ? q2[CLS[n]] : 0.;
            const double kqn = (window && (n == 0 || n == 3)) ? 0. : kq[n];
            cB[n] = __builtin_fma(H1, xm1[n], __builtin_fma(window ? H2 : 0., xm2[n], kb));
            cQ[n] = __builtin_fma(q1[CLS[n]], xm1[n], __builtin_fma(q2n, xm2[n], kqn));
        }
#ifdef ATOMDEV_EXP_PIPE_NOCHAIN
#pragma unroll
        for (int n = 0; n < 4; n++) { rhs[n] = cB[n] + cQ[n] + cP[n] + vs[n]; st[n] += xn[n]; }
        if (j < 0)
#endif
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const double v = vs[s];
#pragma unroll
            for (int n = 0; n < 4; n++) {
                rhs[n] = __builtin_fma(-v, __builtin_fma(cA, st[n], cB[n]), __builtin_fma(cP[n], st[n], cQ[n]));
                if (s == 0) { acc[n] = rhs[n]; st[n] = __builtin_fma(rhs[n], hdt, xn[n]); }
                else if (s == 1) { acc[n] = __builtin_fma(2., rhs[n], acc[n]); st[n] = __builtin_fma(rhs[n], hdt, xn[n]); }
                else if (s == 2) { acc[n] = __builtin_fma(2., rhs[n], acc[n]); st[n] = __builtin_fma(rhs[n], dt, xn[n]); }
                else st[n] = __builtin_fma(acc[n] + rhs[n], dt6, xn[n]);
            }
        }
        if (stored_seen < j - 4) { wait_for(&sh.ctlB[j & 3].stored, j - 4, err, why[1]); }
#pragma unroll
        for (int n = 0; n < 4; n++) { out[n][lane] = st[n]; xm2[n] = xm1[n]; xm1[n] = st[n]; }
        out[4][lane] = rhs[0];
        announce_in_order(&sh.doneB, j, lane);
        PSTAMP(1, 3);
    }
#ifdef ATOMDEV_EXP_PIPE_COUNT
    if (lane == 0 && bx == 2 && i == ATOMDEV_EXP_PIPE_COUNT) { a.aux_w[4] = retries; a.aux_w[5] = why[0]; a.aux_w[6] = why[1]; a.aux_w[7] = why[2]; }
#endif
}

// Waves of the 1 024-thread workgroup: w and w + 4 (and w + 8, w + 12) share a SIMD.  Wave 0 is chain A, wave 1 chain B,
// and the waves that would share their SIMDs end at once; the eight waves on the other two SIMDs are the memory waves:
// rows r = q mod 4 belong to pair q.  One wave of a pair works for chain A, the other for chain B; chain A's half of a row
// is about twice the instructions of chain B's, so which is which alternates with q and each of the two SIMDs carries
// two waves of either kind.  (Spreading the memory waves over the chain waves' SIMDs too - rows 1 to 3 of the table
// below - changed the sweep's time by 2 per cent at most.)
template <bool LIMIT>
__global__ __launch_bounds__(1024) void k_atm_rk3d_pipe(const PipeParams* __restrict__ P, int* err) {
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
    if (threadIdx.x < NSLOT) { sh.ctlA[threadIdx.x] = { 0, 0 }; sh.ctlB[threadIdx.x] = { 0, 0 }; }
    if (threadIdx.x == 0) { sh.doneA = 0; sh.doneB = 0; }
    if (K.jm <= GEO_ROWS)
        for (int j = threadIdx.x; j < K.jm; j += (int)blockDim.x) {
            sh.geo[j][0] = a.costhe[j] * a.tab_irm2sin[(size_t)i * K.jm + j];
            sh.geo[j][1] = a.tab_coefv[(size_t)i * K.jm + j];
        }
    __syncthreads();   // the only barrier: from here on the waves follow each other's progress words
#ifdef ATOMDEV_EXP_PIPE_STAMP
    if (threadIdx.x == 0) { sh.wall[0] = wall_clock64(); sh.cyc[0] = __builtin_readcyclecounter(); }
    for (int n = threadIdx.x; n < 10 * 4 * 8; n += blockDim.x) (&sh.stamps[0][0][0])[n] = 0;
    __syncthreads();
#endif
    // role of a wave: 0 chain A, 1 chain B, 2 + 2 q memory wave of pair q for chain A, 3 + 2 q for chain B, -1 none
#ifndef ATOMDEV_EXP_ROLEMAP
#define ATOMDEV_EXP_ROLEMAP 0
#endif
    static constexpr signed char ROLE[4][16] = {
        // SIMD:   0   1   2   3    0   1   2   3    0   1   2   3    0   1   2   3
        {          0,  1,  2,  3,  -1, -1,  5,  4,  -1, -1,  6,  7,  -1, -1,  9,  8 },
        {          0,  1,  2,  4,   3,  5,  6,  8,  -1, -1,  7,  9,  -1, -1, -1, -1 },
        {          0,  1,  2,  4,  -1,  3,  6,  8,  -1,  5,  7,  9,  -1, -1, -1, -1 },
        {          0,  1,  2,  4,   3,  5,  6,  8,   7,  9, -1, -1,  -1, -1, -1, -1 } };
    const int role = ROLE[ATOMDEV_EXP_ROLEMAP][wave];
    if (role == 0) {
        if (i == 1) chain_a<true>(a, K, sh, i, bx, err);
        else chain_a<false>(a, K, sh, i, bx, err);
    } else if (role == 1) {
        chain_b(a, K, sh, i, bx, err);
    } else if (role >= 2 && !(role & 1)) {
        if (i == 1) memory_wave<true, true, LIMIT>(a, K, sh, (role - 2) >> 1, i, bx, err);
        else memory_wave<false, true, LIMIT>(a, K, sh, (role - 2) >> 1, i, bx, err);
    } else if (role >= 2) {
        memory_wave<false, false, LIMIT>(a, K, sh, (role - 2) >> 1, i, bx, err);
    }
#ifdef ATOMDEV_EXP_PIPE_STAMP
    __syncthreads();
    if (threadIdx.x == 0 && bx == 2 && i == ATOMDEV_EXP_PIPE_STAMP) {
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
        a.ph100 = h->scr[0]; a.topo = h->g[ATOMDEV_TOPOGRAPHY];
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
        hipLaunchKernelGGL(k_atm_rk3d_pipe<true>, grid, dim3(1024), 0, h->stream, (const PipeParams*)h->pipe_params, (int*)atomdev_kernel_error_word(h));
    else
        hipLaunchKernelGGL(k_atm_rk3d_pipe<false>, grid, dim3(1024), 0, h->stream, (const PipeParams*)h->pipe_params, (int*)atomdev_kernel_error_word(h));
    ATOMDEV_TRY(hipGetLastError());
    return 0;
}
