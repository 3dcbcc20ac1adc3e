// Hydrosphere 3-D Runge-Kutta sweep, wave-specialised form (the default).
//
// Same arithmetic, statement for statement, as the line kernel in k_rk_hyd.hip
// (reference: hydrosphere/RungeKutta_Hyd.cpp:30-124, RHS_Hyd.cpp:60-447); the mapping
// to the machine is the one of the atmosphere sweep (k_rk_atm_ws.hip): a workgroup
// takes 64 neighbouring theta lines of one level, ONE PROGNOSTIC VARIABLE PER WAVE,
// and the waves exchange stage values through LDS with one barrier per RK stage:
//     every variable  <- u, v   (radial and meridional advection velocity)
//     v               <- u      (2 du/dtheta / r^2 in the viscous bracket)
//     u, v            <- c      (salt_water_ref = r_water + .7 c c_0 in buoyancy and pressure gradient)
// The c wave publishes salt_water_ref together with its Newton-refined reciprocal, so that the
// two divisions by it (u: buoyancy, v: pressure gradient) cost the three-instruction quotient
// step of the fp64 division sequence instead of the whole sequence; the quotient bits are those
// of a / b for operands that need no range scaling (a water density is ~1e3).
//
// Five waves on four SIMDs: t and w, the two lightest roles, share one (waves 0 and 4).
//
// Level im-2 reads the salinity level im-3 has just computed (RHS_Hyd.cpp:356-357).  The c wave of a level im-3
// workgroup therefore stores its rows with device-scope (write-through) stores and, at the end of every step, when
// its wait for the step's loads has also completed the stores of the step before, publishes "rows up to j-2 are in
// memory" in a progress word of its longitude block; the c wave of the level im-2 workgroup of the same block checks
// that word (read one step ahead, like its other inputs) before it prefetches a row of the level below, and so runs
// a few rows behind.  The poll is
// bounded (an error word is set and the sweep finishes with whatever it reads, atomdev_sync reports it); producer
// workgroups have the lower block index, so they are dispatched before their consumers.  (With three shadow waves
// recomputing level im-3 inside the workgroups of level im-2 the sweep took 0.53 ms instead of 0.40 ms; as a launch
// of its own, level im-2 costs a whole second sweep time on six workgroups.)
//
// The sweep is out of place (see k_rk_hyd.hip): it reads the current buffers and writes the
// alternate set, whose boundary shell is refreshed by extra workgroups of the same launch (copy_shell).
#include <algorithm>
#include <cstdlib>

#include "atomdev_internal.h"

// Compiled twice, like k_rk_atm_ws.hip: as it stands every quotient carries the bits of the reference's division;
// with -DATOMDEV_RK_FAST -ffp-contract=fast divisions by loop-invariant divisors become multiplications by the
// reciprocal and products feeding sums fuse (tolerance form, about 1e-13 from the exact form).
// The tolerance form also uses that, with the neighbour rows and the level above fixed during the four stages of a
// row, every right-hand side is affine in the variable's own stage value x apart from the two advection products:
//     rhs = -u (Ar x + Br) - v (A x + B) + P x + Q   (+ R u for v, the density terms of u and v, the surface source of c)
// Ar, Br, A, B, P, Q are worked out once per row; a stage is five or six fused multiply-adds.
#ifdef ATOMDEV_RK_FAST
#define HYD_WS_ENTRY atomdev_hyd_rk3d_ws_fast
#define k_hyd_rk3d_ws k_hyd_rk3d_wsf
#define DIVC(a, b, rb) ((a) * (rb))
#ifndef ATOMDEV_RK_NOAFFINE
#define HYD_WS_AFFINE 1
#endif
#else
#define HYD_WS_ENTRY atomdev_hyd_rk3d_ws
#define DIVC(a, b, rb) div_by(a, b, rb)
#endif
#ifndef HYD_WS_AFFINE
#define HYD_WS_AFFINE 0
#endif

namespace {

enum Role { R_T = 0, R_V = 1, R_U = 2, R_C = 3, R_W = 4 };

struct HydWsArgs {
    const double *t, *u, *v, *w, *c;        // current state (read)
    double *to, *uo, *vo, *wo, *co;         // next state (written)
    const double *tn, *un, *vn, *wn, *cn;
    const double* p_dyn;
    double *r_water, *r_salt_water;         // read, rewritten on land
    double *aux_u, *aux_v, *aux_w, *buoy, *salt_balance, *salt_finger, *salt_diffusion;
    const uint8_t* flags;
    const double *bathy, *evap, *precip;
    const double *rad, *sinthe, *costhe;
};

// Arguments live in device memory (see k_rk_atm_ws.hip): each role reads only what it needs.
struct HydWsParams {
    HydWsArgs a;
    DevConst K;
};

struct Shared {
    // stage values, double-buffered by stage parity
    double u[2][64], v[2][64];
    double swr[2][64], yswr[2][64];  // salt_water_ref and its refined reciprocal
    double ufin[64];                 // updated u of the row just finished
};

#define PIN(x) asm volatile("" : "+v"(x))

#ifdef ATOMDEV_EXP_STAMP  // timing experiment only: cycle stamps of step 60 of one workgroup go to level 0 of aux_w
#define STAMP(n)                                                                                      \
    do {                                                                                              \
        if (j == 60 && bx == 2 && i == ATOMDEV_EXP_STAMP && lane == 0)                                \
            a.aux_w[ROLE * 16 + (n)] = (double)(long long)__builtin_readcyclecounter();               \
    } while (0)
#else
#define STAMP(n) do {} while (0)
#endif

typedef const __attribute__((address_space(1))) char* gcchar_t;
typedef __attribute__((address_space(1))) char* gchar_t;
__device__ __forceinline__ double ldg(const double* base, unsigned off) {
    return *(const __attribute__((address_space(1))) double*)((gcchar_t)base + off);
}
__device__ __forceinline__ uint8_t ldg_u8(const uint8_t* base, unsigned idx) {
    return *(const __attribute__((address_space(1))) uint8_t*)((gcchar_t)base + idx);
}
__device__ __forceinline__ void stg(double* base, unsigned off, double x) {
    *(__attribute__((address_space(1))) double*)((gchar_t)base + off) = x;
}

// Device-scope accesses (sc1: written through / fetched past the XCD's L2), for data that crosses workgroups.
__device__ __forceinline__ void stg_agent(double* base, unsigned off, double x) {
    __hip_atomic_store((__attribute__((address_space(1))) double*)((gchar_t)base + off), x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double ldg_agent(const double* base, unsigned off) {
    return __hip_atomic_load((const __attribute__((address_space(1))) double*)((gcchar_t)base + off), __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int ld_progress(const int* progress) {
    return __hip_atomic_load((const __attribute__((address_space(1))) int*)progress, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// The hand-off between two workgroups of one launch.  Producer: write-through (sc1) stores of the rows, a wait until the
// memory system has acknowledged them (vmcnt(0), issued by the caller), then the progress word.  Consumer: the progress
// word, then sc1 loads of the rows.  The hardware issues a wave's memory instructions in program order; the two
// compiler barriers keep the compiler from moving an access across the word, which relaxed atomics alone would allow.
// (Release / acquire orderings at agent scope would be the portable spelling; on gfx950 they lower to a write-back of
// the whole L2 (buffer_wbl2) and an invalidate of the L1 (buffer_inv) per hand-off, microseconds each, where the rows
// handed over here are a few hundred bytes that bypass both caches already: every store and load of them carries sc1.)
__device__ __forceinline__ void announce_rows(int* progress, int rows, int lane) {
    asm volatile("" ::: "memory");
    if (lane == 0) __hip_atomic_store(progress, rows, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("" ::: "memory");
}
// Waits until the producer's progress word has reached `need` and returns what it read; gives up after about a second
// and says so in *err (page-locked host memory the next synchronising call of the library looks at).
__device__ __forceinline__ int wait_rows(const int* progress, int need, int* err) {
    int seen = ld_progress(progress);
    for (int n = 0; seen < need; n++) {
        __builtin_amdgcn_s_sleep(4);
        if (n > (1 << 21)) {
            __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            seen = need;
            break;
        }
        seen = ld_progress(progress);
    }
    asm volatile("" ::: "memory");
    return seen;
}

// 1/b to the accuracy the fp64 division sequence uses: hardware estimate plus two Newton steps.
__device__ __forceinline__ double rcp_refined(double b) {
    double y = __builtin_amdgcn_rcp(b);
    double e = __builtin_fma(-b, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-b, y, 1.0);
    return __builtin_fma(y, e, y);
}

struct ThetaK { double dthe, i_dthe, two_dthe, i_two_dthe, dthe2, i_dthe2; };

// Theta derivatives (RHS_Hyd.cpp:156-161, 177-181, 250-276) in the one form  hdj ((a x + B) + C) / D  that covers the
// three forms the reference's cascade of branches leaves standing (see k_rk_atm_ws.hip: same coefficients, same
// roundings); a lane whose second derivative the phi block zeroes (RHS_Hyd.cpp:296-300, 326-330) gets e = F = G = 0.
struct ThetaForm {
    double a, e;          // factors of the stage value in d1 and d2
    double D1, iD1;       // divisor of d1 and its reciprocal
};
__device__ __forceinline__ ThetaForm theta_form(int mode, bool d2zero, const ThetaK& K) {
    ThetaForm f;
    f.a = mode == 0 ? 0. : (mode == 1 ? 1. : -3.);
    f.e = d2zero ? 0. : (mode == 0 ? -2. : (mode == 1 ? 0. : 2.));
    f.D1 = mode == 1 ? K.dthe : K.two_dthe;
    f.iD1 = mode == 1 ? K.i_dthe : K.i_two_dthe;
    return f;
}
__device__ __forceinline__ void theta_terms(int mode, bool d2zero, double xm2, double xm1, double xp1, double& B, double& C,
                                            double& F, double& G) {
    B = mode == 0 ? xp1 : (mode == 1 ? -xm1 : 4. * xm1);
    C = mode == 0 ? -xm1 : (mode == 1 ? 0. : -xm2);
    F = d2zero ? 0. : (mode == 0 ? xp1 : (mode == 1 ? 0. : -2. * xm1));
    G = d2zero ? 0. : (mode == 0 ? xm1 : (mode == 1 ? 0. : xm2));
}

// The whole sweep of one wave.  All roles execute the same number of barriers (one before the
// first step, four per step).  Memory schedule of a step (row j): every load (row j+1 of xn,
// flags, the level above and the role's extras, row j+2 of the own variable) is issued at the
// top, the stores of row j-1's results in stage 2; nothing loaded in a step is used before the
// next one, and the end-of-step wait leaves only the stores outstanding.
// TOPC: the instantiation for level im-2 (the others carry no test for it inside their stages).  There the radial terms
// are dead: the tolerance form drops the radial advection and, for the waves that need it for nothing else, u.
template <int ROLE, bool TOPC = false>
__device__ __forceinline__ void sweep_role(const HydWsArgs& a, const DevConst& K, Shared& sh, const double* tab,
                                           int bx, int i, int* progress, int* err,
                                           const double* __restrict__ X, double* __restrict__ XO,
                                           const double* __restrict__ XN) {
    // progress: the progress word of this longitude block (level im-3 writes it, level im-2 polls it), null on
    // the other levels; err: where a poll that gave up leaves its mark.
    constexpr bool STORE = true;
    constexpr int lag = 0, extra = 0;
    constexpr bool isT = ROLE == R_T, isU = ROLE == R_U, isV = ROLE == R_V, isW = ROLE == R_W, isC = ROLE == R_C;
    constexpr bool hasH0I = isT || isU || isC;
    constexpr int NST = isU ? 7 : (isV || isW || isC) ? 2 : 1;  // stores per step
    const int im = K.im, jm = K.jm, km = K.km;
    const int lane = threadIdx.x & 63;
    const int kk = 1 + bx * 64 + lane;
    const int k = kk <= km - 2 ? kk : km - 2;  // idle lanes shadow the last line and reproduce its bits
    constexpr bool top = TOPC;  // i == im - 2: the kernel picks the instantiation
    const bool feeds_top = ROLE == R_C && progress && !top;   // c wave of level im-3
    const bool polls = ROLE == R_C && progress && top;        // c wave of level im-2
    const bool kwin = (k >= 2) && (k < km - 3);

    // ---- constants of the dependent chain, in vector registers ----
    ThetaK TK = { K.dthe, K.i_dthe, K.two_dthe, K.i_two_dthe, K.dthe2, K.i_dthe2 };
    PIN(TK.dthe); PIN(TK.i_dthe); PIN(TK.two_dthe); PIN(TK.i_two_dthe); PIN(TK.dthe2); PIN(TK.i_dthe2);
    double rm = a.rad[i];
    double rm2 = rm * rm, i_rm = 1. / rm, i_rm2 = 1. / rm2;
    PIN(rm); PIN(rm2); PIN(i_rm); PIN(i_rm2);
    double dr = K.dr, i_dr = K.i_dr;
    PIN(dr); PIN(i_dr);
    // divisor of the diffusion bracket and of the damping term of this variable
    double dv, i_dv, dm, i_dm;
    if constexpr (isT) { dv = K.re_pr; i_dv = K.i_re_pr; dm = K.dr2; i_dm = K.i_dr2; }
    else if constexpr (isU) { dv = K.re; i_dv = K.i_re; dm = K.dthe2; i_dm = K.i_dthe2; }
    else if constexpr (isV) { dv = K.re; i_dv = K.i_re; dm = K.dthe2; i_dm = K.i_dthe2; }
    else if constexpr (isW) { dv = K.re; i_dv = K.i_re; dm = K.dphi2; i_dm = K.i_dphi2; }
    else { dv = K.re_sc; i_dv = K.i_re_sc; dm = K.dr2; i_dm = K.i_dr2; }
    PIN(dv); PIN(i_dv); PIN(dm); PIN(i_dm);
    double dt = K.dt, hdt = .5 * K.dt, i_six = K.i_six, six = 6.;
    PIN(dt); PIN(hdt); PIN(i_six); PIN(six);
    double k1 = 0., k2 = 0., k3 = 0., k4 = 0.;  // role-specific scalars
    if constexpr (isU) { k1 = K.buoy_g; k2 = K.coeff_buoy; }
    if constexpr (isC) { k1 = K.c_0; k2 = K.coeff_salinity; k3 = K.two_dr; k4 = K.i_two_dr; }
    PIN(k1); PIN(k2); PIN(k3); PIN(k4);
    double topo_step = K.topo_step, i_topo_step = K.i_topo_step;
    const double hight = (double)i * topo_step;
    const uint8_t* __restrict__ FL = a.flags;

    // ---- byte offsets of rows j, j+1, j+2 of this line, advanced by one row per step ----
    const unsigned rowB = (unsigned)km * 8u;
    const unsigned lvlB = (unsigned)jm * rowB;
    unsigned o0 = (unsigned)(idx3(jm, km, i, 1, k) * 8);
    unsigned o1 = o0 + rowB, o2 = o1 + rowB;
    unsigned q1 = (unsigned)(((size_t)2 * km + k) * 8);  // row j+1 of a 2-D field
    PIN(o0); PIN(o1); PIN(o2); PIN(q1);
    const double *E1 = nullptr, *E2 = nullptr, *E3 = nullptr;
    if constexpr (isV) { E1 = a.p_dyn; E2 = a.u; }
    if constexpr (isU) E1 = a.r_salt_water;
    if constexpr (isC) { E1 = a.r_water; E2 = a.evap; E3 = a.precip; }
    const double* BA = a.bathy;
    // side-output fields of the role
    double *O1 = nullptr, *O2 = nullptr, *O3 = nullptr, *O4 = nullptr, *O5 = nullptr, *O6 = nullptr;
    if constexpr (isU) { O1 = a.aux_u; O2 = a.buoy; O3 = a.salt_balance; O4 = a.salt_finger; O5 = a.salt_diffusion; O6 = a.r_salt_water; }
    if constexpr (isV) O1 = a.aux_v;
    if constexpr (isW) O1 = a.aux_w;
    if constexpr (isC) O1 = a.r_water;

    double xm2 = 0., xm1 = ldg(X, o0 - rowB), x0 = ldg(X, o0), xp1 = ldg(X, o1);
    double ym2 = 0., ym1 = 0., y0 = 0., yp1 = 0.;  // p_dyn rows (v wave)
    double um2 = 0., um1 = 0., up1 = 0.;            // u rows (v wave): updated j-2, j-1; old j+1
    if constexpr (isV) {
        ym1 = ldg(E1, o0 - rowB); y0 = ldg(E1, o0); yp1 = ldg(E1, o1);
        um1 = ldg(E2, o0 - rowB); up1 = ldg(E2, o1);
    }

    // raw inputs of the next step, loaded during this one
    double n_xn, n_xa, n_bathy = 0., n_r1 = 0., n_cb = 0., n_ev = 0., n_pr = 0.;
    double n_cos, n_rm2sin, n_irm2sin, n_coefv = 0.;
    uint8_t n_fl;
    auto load_next = [&](unsigned o, unsigned q) {
        n_xn = ldg(XN, o);
        n_fl = ldg_u8(FL, o >> 3);
        n_xa = ldg(X, o + lvlB);  // level i+1 <= im-1 always exists
        if constexpr (hasH0I) n_bathy = ldg(BA, q);
        if constexpr (isU) n_r1 = ldg(E1, o);
        if constexpr (isC) {
            if (top) {  // RHS_Hyd.cpp:355-359
                n_cb = polls ? ldg_agent(XO, o - lvlB) : ldg(XO, o - lvlB);  // c at level im-3: already updated (level im-1, in n_xa, never is)
                n_ev = ldg(E2, q); n_pr = ldg(E3, q);
            }
        }
    };
    auto load_table = [&](int jj) {
        const double* row = tab + 4 * jj;
        n_cos = row[0]; n_rm2sin = row[1]; n_irm2sin = row[2];
        if constexpr (isV) n_coefv = row[3];
    };
    // level im-2: rows of the level below known to be in memory, and the progress word as read during the last step
    int known = 0, n_prog = 0;
    if (polls) known = wait_rows(progress, 1, err);
    load_next(o0, q1 - rowB);
    load_table(1);
    // c wave: r_water of rows j and j+1; row j+2 is fetched at the top of step j
    double rw_cur = 0., rw_nxt = 0., n_rw2 = 0.;
    if constexpr (isC) { rw_cur = ldg(E1, o0); rw_nxt = ldg(E1, o1); }

    // what the u and v waves need of a new c value (RHS_Hyd.cpp:345)
    double c_swr = 0.;  // c wave: the salt water density it published for the value it is working with
    auto publish_c = [&](int buf, double cval, double rw) {
        const double swr = rw + .7 * cval * k1;
        c_swr = swr;
        sh.swr[buf][lane] = swr;
        sh.yswr[buf][lane] = rcp_refined(swr);
    };

    // stage-1 values of row 1
    if constexpr (isU) sh.u[0][lane] = x0;
    if constexpr (isV) sh.v[0][lane] = x0;
    if constexpr (isC) publish_c(0, x0, rw_cur);
    lds_barrier();
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): nothing pending when the loop starts

    double so1 = 0., so2 = 0., so3 = 0., so4 = 0., so5 = 0., so6 = 0.;  // side outputs of the previous row

    for (int tick = 1; tick < jm - 1 + extra; tick++) {
        const int j = tick - lag;
        if (j < 1 || j > jm - 2) {  // this level idles: keep the barrier rhythm of the workgroup
#pragma unroll
            for (int s = 0; s < 4; s++) lds_barrier();
            continue;
        }
        if (feeds_top && j >= 3) {
            // Only the stores of the step before (row j-2, issued three stages ago) can still be in flight here:
            // once they have completed, the row may be announced.
            __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
            announce_rows(progress, j - 2, lane);
        }
        STAMP(0);
        // ---- take over the prefetched inputs of this step, start loading the next step's ----
        const double xn = n_xn, xa_raw = n_xa;
        const uint8_t fl = n_fl;
        const double costhe = n_cos, rm2sin = n_rm2sin, i_rm2sin = n_irm2sin, coef_v = n_coefv;
        const double bathy = n_bathy, rsw_in = n_r1;
        const double c_below = n_cb;
        const double c_top = xa_raw, ep = n_ev - n_pr;
        const double m3ct = -3. * c_top, om2ct = 1. - 2. * c_top;
        if (polls) {  // row j+1 of the level below must be in memory before it is fetched (row jm-1 is a boundary row)
            const int need = j + 1 < jm - 2 ? j + 1 : jm - 2;
            known = n_prog > known ? n_prog : known;
            if (known < need) known = wait_rows(progress, need, err);
        }
        load_next(o1, q1);  // row j+1 <= jm-1 always exists
        if (polls) n_prog = ld_progress(progress);  // arrives with the step's other loads: the next step rarely has to wait
        double xp2 = 0., yp2 = 0., up2 = 0.;
        if (j + 2 < jm) {
            xp2 = ldg(X, o2);
            if constexpr (isC) n_rw2 = ldg(E1, o2);
            if constexpr (isV) { yp2 = ldg(E1, o2); up2 = ldg(E2, o2); }
        }
        STAMP(1);

        const bool land = fl & FL_LAND;
        const double h_d_j = ((fl & FL_H0J) && (fl & FL_H0K)) ? 0. : 1.;  // RHS_Hyd.cpp:117-139
        const bool window = (j >= 2) && (j < jm - 3);
        const int mode = !window ? 0 : ((fl & FL_ONESIDED_S) ? 2 : 1);
        const bool all_bw = __all(mode == 1);                    // whole wave on the two-point backward form: d2 == 0
        const bool d2zero = kwin && (fl & FL_D2ZERO);            // RHS_Hyd.cpp:296-300, 326-330
        const bool rlive = (i < im - 2) && !(fl & FL_RDEAD);     // RHS_Hyd.cpp:191-214
        const double xa = rlive ? xa_raw : 0.;
        const ThetaForm TF = theta_form(mode, d2zero, TK);
        double xB = 0., xC = 0., xF = 0., xG = 0., uB = 0., uC = 0.;
        if (!all_bw) theta_terms(mode, d2zero, xm2, xm1, xp1, xB, xC, xF, xG);

        double h_0 = 0.;
        if constexpr (hasH0I) {  // h_0_i, RHS_Hyd.cpp:96-111
            h_0 = 1.;
            if (land) {
                const double topo_diff = fabs(hight - bathy);
                if (!(topo_diff == topo_step)) h_0 = div_by(topo_diff, topo_step, i_topo_step);
            }
        }
        if constexpr (isV) h_0 = (fl & FL_H0J) ? 1. : 0.;
        if constexpr (isW) h_0 = (fl & FL_H0K) ? 1. : 0.;

        double pg_num = 0.;
        if constexpr (isV) {
            if (j > 1) um1 = sh.ufin[lane];  // updated u of row j-1 (row 0 is a boundary row)
            double pB, pC, pF, pG;
            theta_terms(mode, true, ym2, ym1, yp1, pB, pC, pF, pG);
            const double dpdthe = DIVC(h_d_j * ((TF.a * y0 + pB) + pC), TF.D1, TF.iD1);
            pg_num = DIVC(h_d_j * dpdthe, rm, i_rm);  // h_d_j*dpdthe/rm, then / salt_water_ref per stage
            theta_terms(mode, true, um2, um1, up1, uB, uC, pF, pG);
        }
        double rsw = rsw_in;  // u wave: r_salt_water, overwritten on land after the first evaluation
        double rw = rw_cur;   // c wave: r_water, likewise

        double st = x0, acc = 0., rhs = 0., RS = 0., pg_v = 0., sb = 0.;
        // ---- affine form: coefficients of the row ----
        constexpr bool AFFINE = HYD_WS_AFFINE;
        double cAr = 0., cBr = 0., cA = 0., cB = 0., cP = 0., cQ = 0., cR = 0., cFac = 0., cM = 0., cFac4 = 0., cMF = 0.;
        if constexpr (AFFINE) {
            double A1, B1, A2 = 0., B2 = 0., B1u = 0.;
            if (all_bw) {
                A1 = h_d_j * TK.i_dthe;
                B1 = -xm1 * A1;
                if constexpr (isV) B1u = -um1 * A1;
            } else {
                const double hD = h_d_j * TF.iD1, hE = h_d_j * TK.i_dthe2;
                A1 = TF.a * hD; B1 = (xB + xC) * hD;
                A2 = TF.e * hE; B2 = (xF + xG) * hE;
                if constexpr (isV) B1u = (uB + uC) * hD;
            }
            cAr = rlive ? -i_dr : 0.; cBr = rlive ? xa * i_dr : 0.;   // dr1 = cAr x + cBr
            const double g = costhe * i_rm2sin;
            // the diffusion bracket: (2 or 4) dr1 / rm + d2 / rm2 + d1 cos / (rm2 sin)
            const double kr = (isU ? 4. : 2.) * i_rm;
            double Pb = __builtin_fma(A1, g, kr * cAr), Qb = __builtin_fma(B1, g, kr * cBr);
            if (!all_bw) { Pb = __builtin_fma(A2, i_rm2, Pb); Qb = __builtin_fma(B2, i_rm2, Qb); }
            cA = A1 * i_rm; cB = B1 * i_rm;
            cP = __builtin_fma(-h_0, i_dm, Pb * i_dv); cQ = Qb * i_dv;
            if constexpr (isV) {
                const double two = 2. * i_rm2 * i_dv;
                cP = __builtin_fma(-coef_v * h_d_j, i_dv, cP);
                cR = two * A1;
                cQ = __builtin_fma(two, B1u, cQ);
            }
            if constexpr (isC) {
                if (top) {  // RHS_Hyd.cpp:355-372
                    cFac = land ? 0. : k2 * k4 * (1. - 2. * c_top) * ep;
                    cM = -3. * c_top - c_below;
                    cFac4 = 4. * cFac; cMF = cM * cFac;
                }
            }
        }
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const int rb = s & 1, wb = rb ^ 1;
            if (STORE && s == 1 && j > 1) {  // results of the previous row
                if (feeds_top) stg_agent(XO, o0 - rowB, xm1); else stg(XO, o0 - rowB, xm1);
                if constexpr (isU) {
                    stg(O1, o0 - rowB, so1); stg(O2, o0 - rowB, so2); stg(O3, o0 - rowB, so3);
                    stg(O4, o0 - rowB, so4); stg(O5, o0 - rowB, so5); stg(O6, o0 - rowB, so6);
                }
                if constexpr (isV || isW || isC) stg(O1, o0 - rowB, so1);
            }
            if (s == 3) load_table(j + 1);
            double u, v;
            if constexpr (isU) u = st;
            else if constexpr (!(TOPC && HYD_WS_AFFINE) || isV) u = sh.u[rb][lane];
            else u = 0.;
            if constexpr (isV) v = st; else v = sh.v[rb][lane];

          if constexpr (AFFINE) {
            const double tr = __builtin_fma(cAr, st, cBr);
            const double tv = __builtin_fma(cA, st, cB);
            double lin = __builtin_fma(cP, st, cQ);
            if constexpr (isU) {
                const double swr = sh.swr[rb][lane];
                RS = k1 * __builtin_fma(rsw, sh.yswr[rb][lane], -1.);  // RHS_Hyd.cpp:374: g (r_salt_water - ref) / ref
                sb = swr - rsw;
                if (land) rsw = 1032.;
                lin += RS;
            }
            if constexpr (isV) {
                pg_v = pg_num * sh.yswr[rb][lane];
                lin = __builtin_fma(cR, u, lin) - pg_v;
            }
            if constexpr (isC) {
                if (top) {
                    double sal = c_swr * __builtin_fma(cFac4, st, cMF);   // = swr (4 st + cM) cFac
                    sal = fmin(fmax(sal, -20.), 20.);
                    lin += sal;
                }
                if (land) rw = 1007.;  // seen by the following stages (RHS_Hyd.cpp:395-402)
            }
            if constexpr (TOPC) rhs = __builtin_fma(-v, tv, lin);
            else rhs = __builtin_fma(-u, tr, __builtin_fma(-v, tv, lin));
          } else {
            double d1, d2 = 0.;
            if (all_bw) {
                d1 = DIVC(h_d_j * (st - xm1), TK.dthe, TK.i_dthe);
            } else {
                d1 = DIVC(h_d_j * ((TF.a * st + xB) + xC), TF.D1, TF.iD1);
                d2 = DIVC(h_d_j * ((TF.e * st + xF) + xG), TK.dthe2, TK.i_dthe2);
            }
            // RHS_Hyd.cpp:206-211.  On level im-2 no radial difference is live: dr1 is +0 there and so is its quotient
            // by rm, which the whole workgroup skips (the sums below still see the two zeros).
            double dr1 = 0., dr1_rm = 0.;  // dr1 and (2 or 4) dr1 / rm
            if (!top) {
                dr1 = rlive ? DIVC(xa - st, dr, i_dr) : 0.;
                dr1_rm = DIVC(dr1 * (isU ? 4. : 2.), rm, i_rm);
            }
            const double adv = u * dr1 + DIVC(v * d1, rm, i_rm);

            // RHS_Hyd.cpp:406-437 with the exact-zero terms removed
            if constexpr (isT || isC) {
                double brk;
                if (all_bw) brk = dr1_rm + DIVC(d1 * costhe, rm2sin, i_rm2sin);
                else brk = dr1_rm + DIVC(d2, rm2, i_rm2) + DIVC(d1 * costhe, rm2sin, i_rm2sin);
                rhs = -adv + DIVC(brk, dv, i_dv);
                if constexpr (isC) {
                    if (top) {  // RHS_Hyd.cpp:355-372
                        // the density is the one published with this stage value (same operands, same operations);
                        // 4 st is exact, so the fused form of -3 c_top + 4 st rounds once, like the sum it replaces
                        double salinity_surface = DIVC(c_swr * (__builtin_fma(4., st, m3ct) - c_below), k3, k4)
                                                  * om2ct * ep;
                        double salinity_evaporation = k2 * salinity_surface;
                        if (land) salinity_evaporation = 0.;
                        if (salinity_evaporation >= 20.) salinity_evaporation = 20.;
                        if (salinity_evaporation <= -20.) salinity_evaporation = -20.;
                        rhs = rhs + salinity_evaporation;
                    }
                    if (land) rw = 1007.;  // seen by the following stages (RHS_Hyd.cpp:395-402)
                }
                rhs = rhs - DIVC(h_0 * st, dm, i_dm);
            } else if constexpr (isU) {
                double brk;
                if (all_bw) brk = dr1_rm + DIVC(d1 * costhe, rm2sin, i_rm2sin);
                else brk = DIVC(d2, rm2, i_rm2) + dr1_rm + DIVC(d1 * costhe, rm2sin, i_rm2sin);
                const double swr = sh.swr[rb][lane];
                RS = DIVC(k1 * (rsw - swr), swr, sh.yswr[rb][lane]);  // RHS_Hyd.cpp:374
                sb = swr - rsw;
                if (land) rsw = 1032.;
                rhs = -adv + DIVC(brk, dv, i_dv) + RS - DIVC(h_0 * u, dm, i_dm);
            } else if constexpr (isV) {
                const double d1u = DIVC(h_d_j * ((TF.a * u + uB) + uC), TF.D1, TF.iD1);
                pg_v = DIVC(pg_num, sh.swr[rb][lane], sh.yswr[rb][lane]);
                double brk;
                if (all_bw) brk = dr1_rm + DIVC(d1, rm2sin, i_rm2sin) * costhe;
                else brk = dr1_rm + DIVC(d2, rm2, i_rm2) + DIVC(d1, rm2sin, i_rm2sin) * costhe;
                rhs = -adv - pg_v
                      + DIVC(brk - coef_v * h_d_j * v + DIVC(2. * d1u, rm2, i_rm2), dv, i_dv)
                      - DIVC(h_0 * v, dm, i_dm);
            } else {  // R_W
                double brk;
                if (all_bw) brk = dr1_rm + DIVC(d1, rm2sin, i_rm2sin) * costhe;
                else brk = dr1_rm + DIVC(d2, rm2, i_rm2) + DIVC(d1, rm2sin, i_rm2sin) * costhe;
                rhs = -adv + DIVC(brk, dv, i_dv) - DIVC(h_0 * st, dm, i_dm);
            }
          }

            // RungeKutta_Hyd.cpp:56-120 (stage 3 advances w from its stage-2 value, :101)
            // rhs * .5 * dt == rhs * (.5 * dt) and acc + 2. * rhs == fma(2, rhs, acc) bit for bit (scaling by a
            // power of two is exact), one instruction less each
            if (s == 0) { acc = rhs; st = xn + rhs * hdt; }
            else if (s == 1) { acc = __builtin_fma(2., rhs, acc); st = xn + rhs * hdt; }
            else if (s == 2) { acc = __builtin_fma(2., rhs, acc); st = (isW ? st : xn) + rhs * dt; }
            else st = xn + DIVC(dt * (acc + rhs), six, i_six);

            // publish: the next stage's value, or (after the last stage) stage 1 of the next row
            const double pub = (s < 3) ? st : xp1;
            if constexpr (isU) { sh.u[wb][lane] = pub; if (s == 3) sh.ufin[lane] = st; }
            if constexpr (isV) sh.v[wb][lane] = pub;
            if constexpr (isC) {
                publish_c(wb, pub, (s < 3) ? rw : rw_nxt);
            }
            STAMP(2 + 2 * s);
            lds_barrier();
            STAMP(3 + 2 * s);
        }

        // ---- side outputs of the 4th evaluation (RHS_Hyd.cpp:379-402, 440-446) ----
        if constexpr (isU) {
            double sfinger, sdiff;
            if (sb < 0.) { sdiff = sb; sfinger = 0.; } else { sfinger = sb; sdiff = 0.; }
            so1 = land ? 0. : rhs;
            so2 = land ? 0. : RS * k2 * 1000.;
            so3 = land ? 0. : sb; so4 = land ? 0. : sfinger; so5 = land ? 0. : sdiff;
            so6 = land ? 1032. : rsw_in;
        }
        if constexpr (isV) so1 = land ? 0. : rhs + pg_v;
        if constexpr (isW) so1 = land ? 0. : rhs;
        if constexpr (isC) so1 = land ? 1007. : rw_cur;
        // Vector-memory operations complete in issue order: waiting until only this step's stores may
        // be outstanding completes its loads without waiting for the write acknowledgements.
        if (STORE && j > 1) __builtin_amdgcn_s_waitcnt(0x0F70 | NST);  // vmcnt(NST)
        else __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the first step stores nothing
        STAMP(10);
        xm2 = xm1; xm1 = st; x0 = xp1; xp1 = xp2;
        if constexpr (isV) { ym2 = ym1; ym1 = y0; y0 = yp1; yp1 = yp2; um2 = um1; up1 = up2; }
        if constexpr (isC) { rw_cur = rw_nxt; rw_nxt = n_rw2; }
        o0 += rowB; o1 += rowB; o2 += rowB; q1 += rowB;
    }
    // the next sweep starts from "no row finished": the reader of the word is the last to touch it
    if (polls && lane == 0) __hip_atomic_store(progress, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // row jm-2
    if (feeds_top) {
        stg_agent(XO, o0 - rowB, xm1);
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
        announce_rows(progress, jm - 2, lane);
    } else {
        stg(XO, o0 - rowB, xm1);
    }
    if constexpr (isU) {
        stg(O1, o0 - rowB, so1); stg(O2, o0 - rowB, so2); stg(O3, o0 - rowB, so3);
        stg(O4, o0 - rowB, so4); stg(O5, o0 - rowB, so5); stg(O6, o0 - rowB, so6);
    }
    if constexpr (isV || isW || isC) stg(O1, o0 - rowB, so1);
}

// Levels 1 .. im-2, one workgroup of five waves (one variable each) per 64 lines (NB = 1) - or TEN waves for two
// neighbouring blocks of 64 lines of one level (NB = 2).  A compute unit has four SIMDs and a SIMD issues one
// instruction every ~4 cycles whichever of its waves it comes from: with five waves the SIMD that carries two of them
// (t + w, ~950 instructions a row against 550 - 780 on the other three) sets the pace of every stage.  Waves go to
// SIMDs round robin (wave n on SIMD n % 4); with ten waves the order below puts {c, v} of a block on a SIMD of their
// own and {t, u, w} of a block together on another: ~1 500 instructions a row for two blocks - a fifth less per
// block, but half as long again per row.  So two blocks per workgroup pay where the grid has more blocks than the
// chip can hold side by side (0.25 degree: -12 %, tolerance form -23 %), and cost where every block has a compute
// unit of its own anyway (1 degree: +24 %); the launch picks.  The two blocks are independent recurrences that only
// share the stage barrier.
struct WaveJob { int block, role; };
template <int NB>
__device__ __forceinline__ WaveJob wave_job(int wave) {
    if (NB == 1) return { 0, wave };  // t, v, u, c, w
    // wave:  0    1    2    3    4    5    6    7    8    9
    // SIMD:  0    1    2    3    0    1    2    3    0    1
    //       A.t  B.t  A.c  B.c  A.u  B.u  A.v  B.v  A.w  B.w
    const int role[5] = { R_T, R_C, R_U, R_V, R_W };
    return { wave & 1, role[wave >> 1] };
}

// Copies the boundary shell (i, j or k on a face) of the five variables from the current buffers to the alternate
// ones: the sweep writes interior points only.  Work of the `ncopy` extra workgroups at the end of the sweep's grid,
// which run beside it on whatever the sweep's own workgroups leave free.  Per variable: levels 0 and im-1 whole, then
// of every other level the rows j = 0, jm-1 and the columns k = 0, km-1.
__device__ void copy_shell(const double* const* __restrict__ src, double* const* __restrict__ dst, unsigned part, unsigned parts,
                           int im, int jm, int km) {
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
    for (unsigned idx = part * blockDim.x + threadIdx.x; idx < total; idx += 4 * stride) {
        double v[4]; size_t o[4]; unsigned f[4]; bool ok[4];
#pragma unroll
        for (int n = 0; n < 4; n++) {
            const unsigned id = idx + n * stride;
            ok[n] = id < total;
            f[n] = ok[n] ? id / per_field : 0u;
            o[n] = ok[n] ? offset(id - f[n] * per_field) : 0;
            v[n] = ok[n] ? src[f[n]][o[n]] : 0.;
        }
#pragma unroll
        for (int n = 0; n < 4; n++) if (ok[n]) dst[f[n]][o[n]] = v[n];
    }
}

template <int NB>
__global__ __launch_bounds__(320 * NB) void k_hyd_rk3d_ws(const HydWsParams* __restrict__ P, int nlev, int* __restrict__ progress,
                                                      int* __restrict__ err, const double* const* __restrict__ shell_src,
                                                      double* const* __restrict__ shell_dst, int ncopy) {
    const HydWsArgs& a = P->a;
    const DevConst& K = P->K;
    if (blockIdx.x >= gridDim.x - (unsigned)ncopy) {  // before anything a workgroup does together
        copy_shell(shell_src, shell_dst, blockIdx.x - (gridDim.x - (unsigned)ncopy), (unsigned)ncopy, K.im, K.jm, K.km);
        return;
    }
    __shared__ Shared sh2[NB];
    extern __shared__ double tab[];  // [jm][4]: cos(the), rm^2 sin(the), its reciprocal, viscous v coefficient
    // all longitude blocks of a level on one XCD (they share cache lines; see k_rk_atm_ws.hip); level im-3 comes
    // one workgroup index before level im-2 of the same longitude blocks
    const int gx = (K.km - 2 + 63) / 64, gp = (gx + NB - 1) / NB;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int lev = xcd + 8 * (idx / gp), pair = idx % gp;
    if (lev >= nlev) return;
    const WaveJob job = wave_job<NB>(threadIdx.x >> 6);
    const int bx = NB * pair + job.block;
    const int wave = job.role;
    Shared& sh = sh2[job.block];
    const int i = 1 + lev;
    const int jm = K.jm;
#ifdef ATOMDEV_EXP_WGTIME  // timing experiment only: duration of every workgroup goes to level 0 of aux_u
    const long long wg_t0 = __builtin_readcyclecounter();
    const long long wg_w0 = wall_clock64();
#endif
    {
        const double rm = a.rad[i], rm2 = rm * rm;
        for (int j = threadIdx.x; j < jm; j += (int)blockDim.x) {
            const double sinthe = a.sinthe[j], costhe = a.costhe[j];
            const double rm2sinthe = rm2 * sinthe;
            tab[4 * j + 0] = costhe;
            tab[4 * j + 1] = rm2sinthe;
            tab[4 * j + 2] = 1. / rm2sinthe;
            tab[4 * j + 3] = 1. + costhe * costhe / (sinthe * sinthe);  // RHS_Hyd.cpp:422 (no rm here)
        }
        __syncthreads();
    }
    if (bx >= gx) {  // the level has an odd number of blocks: these waves only keep the barrier rhythm of the workgroup
        for (int n = 0; n < 1 + 4 * (jm - 2); n++) lds_barrier();
        return;
    }
#ifdef ATOMDEV_EXP_NOPAIR  // timing experiment only: level im-2 takes a stale salinity row
    int* pw = nullptr;
#else
    int* pw = (i >= K.im - 3) ? progress + bx : nullptr;
#endif
    if (i == K.im - 2) {  // a uniform choice: the whole workgroup is on one level
        switch (wave) {
            case R_T: sweep_role<R_T, true>(a, K, sh, tab, bx, i, pw, err, a.t, a.to, a.tn); break;
            case R_V: sweep_role<R_V, true>(a, K, sh, tab, bx, i, pw, err, a.v, a.vo, a.vn); break;
            case R_U: sweep_role<R_U, true>(a, K, sh, tab, bx, i, pw, err, a.u, a.uo, a.un); break;
            case R_C: sweep_role<R_C, true>(a, K, sh, tab, bx, i, pw, err, a.c, a.co, a.cn); break;
            default: sweep_role<R_W, true>(a, K, sh, tab, bx, i, pw, err, a.w, a.wo, a.wn); break;
        }
    } else {
        switch (wave) {
            case R_T: sweep_role<R_T>(a, K, sh, tab, bx, i, pw, err, a.t, a.to, a.tn); break;
            case R_V: sweep_role<R_V>(a, K, sh, tab, bx, i, pw, err, a.v, a.vo, a.vn); break;
            case R_U: sweep_role<R_U>(a, K, sh, tab, bx, i, pw, err, a.u, a.uo, a.un); break;
            case R_C: sweep_role<R_C>(a, K, sh, tab, bx, i, pw, err, a.c, a.co, a.cn); break;
            default: sweep_role<R_W>(a, K, sh, tab, bx, i, pw, err, a.w, a.wo, a.wn); break;
        }
    }
#ifdef ATOMDEV_EXP_WGTIME
    if ((threadIdx.x & 63) == 0) {
        a.aux_u[(lev * gx + bx) * 8 + wave] = (double)(__builtin_readcyclecounter() - wg_t0);
        if (threadIdx.x == 0) { a.aux_u[16384 + lev * gx + bx] = (double)wg_w0; a.aux_u[32768 + lev * gx + bx] = (double)wall_clock64(); }
    }
#endif
}

}  // namespace

int HYD_WS_ENTRY(atomdev* h) {
    static const int X[5] = { ATOMDEV_T, ATOMDEV_U, ATOMDEV_V, ATOMDEV_W, ATOMDEV_C };
    ATOMDEV_REQUIRE(h->n3 * sizeof(double) < ((size_t)1 << 32), "hyd_rk3d: field larger than 4 GiB");
    // Two argument blocks, one per orientation of the (current, alternate) buffer pair; the sweep
    // swaps the pair, so consecutive sweeps alternate between them.
    struct Blocks { HydWsParams p[2]; const double* src[2][5]; double* dst[2][5]; int progress[128]; };
    if (!h->ws_params) {
        Blocks* B = new Blocks();
        for (int o = 0; o < 2; o++) {
            HydWsArgs& a = B->p[o].a;
            double* cur[5]; double* alt[5];
            for (int n = 0; n < 5; n++) { cur[n] = o ? h->alt[X[n]] : h->f[X[n]]; alt[n] = o ? h->f[X[n]] : h->alt[X[n]]; }
            a.t = cur[0]; a.u = cur[1]; a.v = cur[2]; a.w = cur[3]; a.c = cur[4];
            a.to = alt[0]; a.uo = alt[1]; a.vo = alt[2]; a.wo = alt[3]; a.co = alt[4];
            a.tn = h->f[ATOMDEV_TN]; a.un = h->f[ATOMDEV_UN]; a.vn = h->f[ATOMDEV_VN]; a.wn = h->f[ATOMDEV_WN]; a.cn = h->f[ATOMDEV_CN];
            a.p_dyn = h->f[ATOMDEV_P_DYN];
            a.r_water = h->f[ATOMDEV_R_WATER]; a.r_salt_water = h->f[ATOMDEV_R_SALT_WATER];
            a.aux_u = h->f[ATOMDEV_AUX_U]; a.aux_v = h->f[ATOMDEV_AUX_V]; a.aux_w = h->f[ATOMDEV_AUX_W]; a.buoy = h->f[ATOMDEV_BUOYANCY];
            a.salt_balance = h->f[ATOMDEV_SALT_BALANCE]; a.salt_finger = h->f[ATOMDEV_SALT_FINGER];
            a.salt_diffusion = h->f[ATOMDEV_SALT_DIFFUSION];
            a.flags = h->flags;
            a.bathy = h->g[ATOMDEV_TOPOGRAPHY]; a.evap = h->g[ATOMDEV_EVAPORATION_DALTON]; a.precip = h->g[ATOMDEV_PRECIPITATION];
            a.rad = h->rad; a.sinthe = h->sinthe; a.costhe = h->costhe;
            B->p[o].K = h->k;
            for (int n = 0; n < 5; n++) { B->src[o][n] = cur[n]; B->dst[o][n] = alt[n]; }
        }
        hipError_t e = hipMalloc(&h->ws_params, sizeof(Blocks));
        if (e == hipSuccess) e = hipMemcpy(h->ws_params, B, sizeof(Blocks), hipMemcpyHostToDevice);
        h->ws_cur0 = h->f[ATOMDEV_T];
        delete B;
        ATOMDEV_TRY(e);
    }
    const int o = (h->f[ATOMDEV_T] == h->ws_cur0) ? 0 : 1;
    Blocks* D = (Blocks*)h->ws_params;
    const HydWsParams* P = &D->p[o];
    const unsigned gx = (h->km - 2 + 63) / 64;
    ATOMDEV_REQUIRE(gx <= 128, "hyd_rk3d: more than 128 longitude blocks");
    ATOMDEV_REQUIRE(h->im >= 5, "hyd_rk3d: the grid needs at least 5 levels");
    if (int rc = atomdev_prof_begin(h)) return rc;
    const size_t lds = (size_t)h->jm * 4 * sizeof(double);
    // levels 1 .. im-2; two blocks of 64 lines per workgroup where one block each would be more workgroups than two per
    // compute unit (see the kernel)
    int cus = 0;
    ATOMDEV_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device));
    const char* force = std::getenv("ATOMDEV_HYD_BLOCKS_PER_GROUP");  // 1 or 2: A/B timing
    const bool two = force ? force[0] == '2' : (size_t)(h->im - 2) * gx > (size_t)2 * cus;
    // the shell of the alternate buffers is refreshed by extra workgroups of the same launch: about 32 points a thread
    const size_t shell = (size_t)5 * (2 * h->n2 + (size_t)(h->im - 2) * 2 * (h->km + h->jm));
    ATOMDEV_REQUIRE(shell + (size_t)4 * 256 * 640 < ((size_t)1 << 31), "hyd_rk3d: boundary shell too large for 32-bit indices");
    const unsigned threads = two ? 640 : 320;
    const unsigned ncopy = (unsigned)std::min<size_t>(256, std::max<size_t>(16, shell / ((size_t)threads * 32)));
    const unsigned sweep_blocks = 8 * ((h->im - 2 + 7) / 8) * (two ? (gx + 1) / 2 : gx);
    if (two)
        hipLaunchKernelGGL(k_hyd_rk3d_ws<2>, dim3(sweep_blocks + ncopy), dim3(640), lds, h->stream, P, h->im - 2, D->progress,
                           (int*)atomdev_kernel_error_word(h), &D->src[o][0], &D->dst[o][0], (int)ncopy);
    else
        hipLaunchKernelGGL(k_hyd_rk3d_ws<1>, dim3(sweep_blocks + ncopy), dim3(320), lds, h->stream, P, h->im - 2, D->progress,
                           (int*)atomdev_kernel_error_word(h), &D->src[o][0], &D->dst[o][0], (int)ncopy);
    ATOMDEV_TRY(hipGetLastError());
    if (int rc = atomdev_prof_end(h)) return rc;
    for (int n = 0; n < 5; n++) { double* t = h->f[X[n]]; h->f[X[n]] = h->alt[X[n]]; h->alt[X[n]] = t; }
    return 0;
}
