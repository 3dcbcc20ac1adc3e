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
// Pairing.  Waves w and w+4 of a workgroup share a SIMD, and the SIMD issues one fp64 instruction
// every ~4 cycles whichever wave it comes from, so the stage time is the largest instruction count
// of a pair.  Roles are numbered to even the pairs out: (v, w) (t, co2) (u, ice) (c, cloud).
//
// Divisions.  The buoyancy term divides by two per-point quantities, 1 + k c and the dry density.
// The waves that produce c and t also publish the Newton-refined reciprocals of those divisors
// (the first half of the fp64 division sequence), so that the u wave, which has the most work per
// stage, only runs the three-instruction quotient step.  The instruction sequence, and therefore
// every bit of the quotient, is the one the compiler emits for a / b on operands that need no
// range scaling, which holds for densities and mixing ratios.
#include <cstdlib>

#include "atomdev_internal.h"

// This file is compiled twice.  As it stands: every quotient carries the bits of the reference's division
// (div_by, no contraction).  With -DATOMDEV_RK_FAST -ffp-contract=fast: division by a loop-invariant divisor
// is one multiplication by its reciprocal and products feeding sums fuse, which costs up to one unit in the
// last place per operation (the sweep then agrees with the reference to about 1e-13 instead of bit for bit
// plus pow) and removes a third of the instructions of every RK stage.
//
// The tolerance form goes one step further.  With the neighbour rows fixed during the four stages of a row, every
// right-hand side is affine in the variable's own stage value x except for the advection product with v:
//     rhs = -v (A x + B) + P x + Q   (+ R u for v, - K (humid/dry - 1) for u, + the evaporation source of c at level 1)
// so A, B, P, Q are worked out once per row from the same derivative forms, geometry factors and source terms, and a
// stage is three or four fused multiply-adds.  The buoyancy term k (r_humid - r_dry) / r_dry with
// r_humid = r_dry (1 + c) / (1 + k c) is evaluated as k ((1 + c) / (1 + k c) - 1): the dry density drops out, and the
// t wave publishes it only on level 1, where the evaporation source of c needs it.
#ifdef ATOMDEV_RK_FAST
#define ATM_WS_ENTRY atomdev_atm_rk3d_ws_fast
#define k_atm_rk3d_ws k_atm_rk3d_wsf
#define k_atm_prep k_atm_prepf
#define k_atm_limit_prep k_atm_limit_prepf
#define DIVC(a, b, rb) ((a) * (rb))
#ifndef ATOMDEV_RK_NOAFFINE
#define ATM_WS_AFFINE 1
#endif
#else
#define ATM_WS_ENTRY atomdev_atm_rk3d_ws
#define DIVC(a, b, rb) div_by(a, b, rb)
#endif
#ifndef ATM_WS_AFFINE
#define ATM_WS_AFFINE 0
#endif
#ifndef ATM_WS_TOPO
#define ATM_WS_TOPO 0
#endif

namespace {

enum Role { R_V = 0, R_T = 1, R_U = 2, R_C = 3, R_W = 4, R_CO2 = 5, R_ICE = 6, R_CLOUD = 7 };

struct AtmWsArgs {
    double *t, *u, *v, *w, *c, *cloud, *ice, *co2;
    const double *tn, *un, *vn, *wn, *cn, *cloudn, *icen, *co2n;
    const double* p_dyn;
    const double *S_v, *S_c, *S_i, *S_r, *S_s;
    double *aux_u, *aux_v, *aux_w, *buoy;
    const uint8_t* flags;
    const double *evap, *precip, *c3;
    const double *ph100, *h0i;                 // k_atm_prep outputs (3-D)
    const double* topo;                        // 2-D (read by the ATM_WS_TOPO build only)
    const double *rad, *costhe;
    const double *tab_rm2sin, *tab_irm2sin, *tab_coefv;   // [i*jm + j]
};

// Arguments live in device memory, not in the kernel-argument segment: a role then reads the dozen
// pointers and constants it needs with scalar loads inside its own branch, instead of the compiler
// loading all ~100 of them ahead of the role switch and spilling them.
struct AtmWsParams {
    AtmWsArgs a;
    DevConst K;
};

struct Shared {
    // stage values, double-buffered by stage parity
    double v[2][64], u[2][64];
    double opc[2][64], cb[2][64], cy[2][64];  // from c: 1 + c, b = 1 + k c, refined 1/b
    double rdry[2][64], yr[2][64];            // from t: dry density and its refined reciprocal
    double ufin[64];                          // updated u of the row just finished
};

// ---- prep: everything per point that does not change during the sweep -------------------------
// ph100 = 100 * p_h, p_h from RHS_Atm.cpp:384-389 (t0jk: t at the ground of the column)
#if ATM_WS_AFFINE
// Tolerance form: x^y as exp(y log x), a few units in the last place off and half the instructions.  Three kernels work
// this value out for the same points (the preparing pass, the limiting pass and its faces-and-rows form) and must agree
// to the bit, while what the compiler fuses - inside an inlined exp or log, or of a product that one kernel finds to be
// the same for a whole loop and another does not - depends on the code around it: so this is one function that is
// called, not inlined.  The tolerance form needs the value on level 1 only; the call costs nothing there.
__device__ __attribute__((noinline)) double ph100_called(double t0jk, double hight, double t_0, double p_SL_factor, double gam,
                                                         double exp_pressure) {
    const double t0K = t0jk * t_0;
    const double p_SL = .01 * (p_SL_factor * t0jk * t_0);
    return 100. * (exp(exp_pressure * log((t0K - gam * hight * 1.e-2) / t0K)) * p_SL);
}
__device__ __forceinline__ double ph100_of(double t0jk, double hight, const DevConst& K) {
    return ph100_called(t0jk, hight, K.t_0, K.p_SL_factor, K.gam, K.exp_pressure);
}
#else
__device__ __forceinline__ double ph100_of(double t0jk, double hight, const DevConst& K) {
    const double t0K = t0jk * K.t_0;
    const double p_SL = .01 * (K.p_SL_factor * t0jk * K.t_0);
    const double p_h = pow((t0K - K.gam * hight * 1.e-2) / t0K, K.exp_pressure) * p_SL;
    return 100. * p_h;
}
#endif
// h_0_i from RHS_Atm.cpp:110-124: static, depends on the mask and the topography only
__device__ __forceinline__ double h0i_of(uint8_t fl, double hight, double topo, const DevConst& K) {
    double h_0_i = 1.;
    if (fl & FL_LAND) {
        const double topo_diff = hight - topo;
        if (!(topo_diff == K.topo_step)) h_0_i = div_by(topo_diff, K.topo_step, K.i_topo_step);
    }
    return h_0_i;
}
// c3 (when given): c at level 3 as the sweep found it, which the level-1 points read (RHS_Atm.cpp:412-430) while
// level 3 itself is being updated.
template <bool WITH_H0I>
__global__ void k_atm_prep(const double* __restrict__ t, const double* __restrict__ topo, const uint8_t* __restrict__ flags,
                           double* __restrict__ ph100, double* __restrict__ h0i, const double* __restrict__ c,
                           double* __restrict__ c3, DevConst K) {
    const int jm = K.jm, km = K.km;
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y, i = 1 + blockIdx.z;
    if (k >= km) return;
    const size_t q = (size_t)j * km + k;
    const size_t p = (size_t)i * jm * km + q;
    if (c3 && i == 3) c3[q] = c[p];
    const double hight = (double)i * K.topo_step;
    // (the tolerance form reads 100 p_h on level 1 only - the dry density has dropped out of the buoyancy - and no h_0_i array)
    if (!ATM_WS_AFFINE || i == 1) ph100[p] = ph100_of(t[q], hight, K);
    if constexpr (WITH_H0I) h0i[p] = h0i_of(flags[p], hight, topo[q], K);
}

// Values that every stage of every step needs are pinned into vector registers: the
// kernel has few waves per SIMD and hundreds of free VGPRs, but only ~100 scalar
// registers, and a constant evicted from those is re-read from the kernel-argument
// segment with a full scalar-cache round trip in the middle of the dependent chain.
#define PIN(x) asm volatile("" : "+v"(x))

#ifdef ATOMDEV_EXP_STAMP  // timing experiment only: cycle stamps of step 60 of one workgroup go to level 0 of aux_w
#define STAMP(n)                                                                                      \
    do {                                                                                              \
        if (j == 60 && bx == 2 && i == ATOMDEV_EXP_STAMP && lane == 0)                                           \
            a.aux_w[ROLE * 16 + (n)] = (double)(long long)__builtin_readcyclecounter(); \
    } while (0)
#else
#define STAMP(n) do {} while (0)
#endif

// Field element at byte offset `off` from a (wave-uniform) base pointer: the base stays in scalar
// registers and the offset is one 32-bit vector register shared by all fields of the role.
// Pointers read from the argument block are generic to the compiler; typing the access as global
// keeps it on global_load/global_store (flat instructions would also count against the LDS wait
// counter of every stage barrier).
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
// The same with the non-temporal hint, for what a step touches once: the old-time-level fields and the source terms
// (read), the side outputs (written).  The last-level cache then keeps the fields the next kernels read again.
__device__ __forceinline__ double ldg_once(const double* base, unsigned off) {
    return __builtin_nontemporal_load((const __attribute__((address_space(1))) double*)((gcchar_t)base + off));
}
__device__ __forceinline__ void stg_once(double* base, unsigned off, double x) {
#ifdef ATOMDEV_EXP_PLAIN_AUX   // timing experiment only
    *(__attribute__((address_space(1))) double*)((gchar_t)base + off) = x;
#else
    __builtin_nontemporal_store(x, (__attribute__((address_space(1))) double*)((gchar_t)base + off));
#endif
}

// 1/b to the accuracy the fp64 division sequence uses: hardware estimate plus two Newton steps.
__device__ __forceinline__ double rcp_refined(double b) {
    double y = __builtin_amdgcn_rcp(b);
    double e = __builtin_fma(-b, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-b, y, 1.0);
    return __builtin_fma(y, e, y);
}
// a / b for operands that need no range scaling (finite, normal, quotient far from overflow and
// underflow): the compiler's own sequence without its scaling and fix-up instructions.
__device__ __forceinline__ double fdiv_normal(double a, double b) { return DIVC(a, b, rcp_refined(b)); }
// 1/b from the hardware estimate (4.6e-8) and one Newton step: 2.2e-15 (tolerance form, where the chain waits for it)
__device__ __forceinline__ double rcp_one_step(double b) {
    const double y = __builtin_amdgcn_rcp(b);
    return __builtin_fma(y, __builtin_fma(-b, y, 1.0), y);
}

struct ThetaK { double dthe, i_dthe, two_dthe, i_two_dthe, dthe2, i_dthe2; };

// Theta derivatives (RHS_Atm.cpp:162-170, 192-199, 245-312).  The reference's cascade of branches leaves one
// of three forms standing at a point (see k_rk_atm.hip):
//   0, outside the index window:        d1 = hdj (x[j+1] - x[j-1]) / 2dthe         d2 = hdj (x[j+1] - 2 x + x[j-1]) / dthe2
//   1, inside it:                       d1 = hdj (x - x[j-1]) / dthe               d2 = 0
//   2, land cell, fluid at j-1 and j-2: d1 = hdj (-3 x + 4 x[j-1] - x[j-2]) / 2dthe  d2 = hdj (2 x - 2 x[j-1] + x[j-2]) / dthe2
// All three are  hdj ((a x + B) + C) / D  with per-point a, B, C, D fixed for the four stages of a row, evaluated
// with the reference's roundings: a x is exact or the reference's own product (-3 x), B and C are neighbour
// values times 1, -1, 4 or -2 (exact), and the two additions are the reference's two.  So every lane runs the
// same instructions whatever its form is.
struct ThetaForm {
    double a, e;          // factors of the stage value in d1 and d2
    double D1, iD1;       // divisor of d1 and its reciprocal
};
__device__ __forceinline__ ThetaForm theta_form(int mode, const ThetaK& K) {
    ThetaForm f;
    f.a = mode == 0 ? 0. : (mode == 1 ? 1. : -3.);
    f.e = mode == 0 ? -2. : (mode == 1 ? 0. : 2.);
    f.D1 = mode == 1 ? K.dthe : K.two_dthe;
    f.iD1 = mode == 1 ? K.i_dthe : K.i_two_dthe;
    return f;
}
// B, C of d1 and F, G of d2 for one variable's neighbour rows
__device__ __forceinline__ void theta_terms(int mode, double xm2, double xm1, double xp1, double& B, double& C, double& F, double& G) {
    B = mode == 0 ? xp1 : (mode == 1 ? -xm1 : 4. * xm1);
    C = mode == 0 ? -xm1 : (mode == 1 ? 0. : -xm2);
    F = mode == 0 ? xp1 : (mode == 1 ? 0. : -2. * xm1);
    G = mode == 0 ? xm1 : (mode == 1 ? 0. : xm2);
}

// BC_Thermo::Value_Limitation_Atm (BC_Thermo.cpp:3235-3275) for the variable of one role: the same comparisons
// in the same order as the pointwise kernel (a value that is not a number passes, unless the cell is land).
template <int ROLE>
__device__ __forceinline__ double limited(double x, bool land) {
    if constexpr (ROLE == R_U) { if (x >= .106) x = .106; if (x <= -.106) x = -.106; if (land) x = 0.; }
    if constexpr (ROLE == R_V) { if (x >= .125) x = .125; if (x <= -.125) x = -.125; if (land) x = 0.; }
    if constexpr (ROLE == R_W) { if (x >= 3.5) x = 3.5; if (x <= -.469) x = -.469; if (land) x = 0.; }
    if constexpr (ROLE == R_T) { if (x >= 1.165) x = 1.165; if (x <= -.78) x = -.78; }
    if constexpr (ROLE == R_C) { if (x >= .03) x = .03; if (x < 0.) x = 0.; }
    if constexpr (ROLE == R_CLOUD) { if (x >= .01) x = .01; if (x < 0.) x = 0.; if (land) x = 0.; }
    if constexpr (ROLE == R_ICE) { if (x >= .005) x = .005; if (x < 0.) x = 0.; if (land) x = 0.; }
    if constexpr (ROLE == R_CO2) { if (x >= 5.36) x = 5.36; if (x <= 1.) x = 1.; if (land) x = 1.; }
    return x;
}

// Value_Limitation_Atm on every point and, in the same pass, what k_atm_prep prepares - from the limited values: a
// point of levels 1 .. im-2 limits the ground temperature of its column itself (limiting what another thread has
// already limited changes nothing).  BC_SolidGround (BC_Bath_Atm.cpp:94-119) right after this limiter would write
// what the limiter's land rule has just written, so the pass stands for both.
struct LimitFields { double *u, *v, *w, *t, *c, *cloud, *ice, *co2, *p_dyn; };
template <bool WITH_H0I>
__device__ __forceinline__ void limit_prep_point(const LimitFields& f, const double* __restrict__ topo, const uint8_t* __restrict__ flags,
                                                 double* __restrict__ ph100, double* __restrict__ h0i, double* __restrict__ c3,
                                                 const DevConst& K, size_t p, size_t q, int i) {
    // every load first, then the pressure power (which needs the ground temperature only) while they are on their way
    const uint8_t fl = flags[p];
    const bool inner = i >= 1 && i <= K.im - 2;
    const double t_ground = (inner && (!ATM_WS_AFFINE || i == 1)) ? f.t[q] : 1.;
    const double topo_q = (WITH_H0I && inner) ? topo[q] : 0.;
    const double u0 = f.u[p], v0 = f.v[p], w0 = f.w[p], t0 = f.t[p], c0 = f.c[p], cl0 = f.cloud[p], ic0 = f.ice[p], co0 = f.co2[p];
    const double hight = (double)i * K.topo_step;
    if (inner) {
        if (!ATM_WS_AFFINE || i == 1) ph100[p] = ph100_of(limited<R_T>(t_ground, false), hight, K);
        if constexpr (WITH_H0I) h0i[p] = h0i_of(fl, hight, topo_q, K);
    }
    const bool land = fl & FL_LAND;
    auto put = [&](double* x, double was, double now) {
        if (__double_as_longlong(was) != __double_as_longlong(now)) x[p] = now;
    };
    put(f.u, u0, limited<R_U>(u0, land)); put(f.v, v0, limited<R_V>(v0, land)); put(f.w, w0, limited<R_W>(w0, land));
    put(f.t, t0, limited<R_T>(t0, land));
    const double c_now = limited<R_C>(c0, land);
    put(f.c, c0, c_now);
    put(f.cloud, cl0, limited<R_CLOUD>(cl0, land)); put(f.ice, ic0, limited<R_ICE>(ic0, land));
    put(f.co2, co0, limited<R_CO2>(co0, land));
    if (land) f.p_dyn[p] = 0.;
    if (c3 && i == 3) c3[q] = c_now;
}
template <bool WITH_H0I>
__global__ __launch_bounds__(256) void k_atm_limit_prep(LimitFields f, const double* __restrict__ topo,
                                                        const uint8_t* __restrict__ flags, double* __restrict__ ph100,
                                                        double* __restrict__ h0i, double* __restrict__ c3, DevConst K, size_t n) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const size_t n2 = (size_t)K.jm * K.km;
    const int i = (int)(p / n2);
    limit_prep_point<WITH_H0I>(f, topo, flags, ph100, h0i, c3, K, p, p - (size_t)i * n2, i);
}
// The form for a sweep that limits the old values of its own points as it takes them over (tolerance form): only the
// points on the six faces, which the sweep reads but does not store, are limited here (the first nb_faces workgroups:
// levels 0 and im-1, then of every other level the rows j = 0, jm-1 and the columns k = 0, km-1); of the others (a
// workgroup a row) p_dyn is zeroed on land, level 1 gets its pressure term and level 3's c goes to the plane as the
// limiter would leave it.
__global__ __launch_bounds__(256) void k_atm_limit_faces_prep(LimitFields f, const uint8_t* __restrict__ flags,
                                                              double* __restrict__ ph100, double* __restrict__ c3, DevConst K,
                                                              unsigned nb_faces) {
    const int im = K.im, jm = K.jm, km = K.km;
    const size_t n2 = (size_t)jm * km;
    if (blockIdx.x < nb_faces) {
        const unsigned rim = 2u * (unsigned)km + 2u * (unsigned)(jm - 2), total = 2u * (unsigned)n2 + (unsigned)(im - 2) * rim;
        const unsigned idx = blockIdx.x * blockDim.x + threadIdx.x;
        if (idx >= total) return;
        int i, j, k;
        if (idx < 2u * (unsigned)n2) {
            i = idx < (unsigned)n2 ? 0 : im - 1;
            const unsigned q = idx < (unsigned)n2 ? idx : idx - (unsigned)n2;
            j = (int)(q / (unsigned)km); k = (int)(q - (unsigned)j * (unsigned)km);
        } else {
            const unsigned r = idx - 2u * (unsigned)n2, e = r % rim;
            i = 1 + (int)(r / rim);
            if (e < (unsigned)km) { j = 0; k = (int)e; }
            else if (e < 2u * (unsigned)km) { j = jm - 1; k = (int)e - km; }
            else { j = 1 + (int)((e - 2u * (unsigned)km) >> 1); k = (e & 1u) ? km - 1 : 0; }
        }
        const size_t q = (size_t)j * km + k;
        limit_prep_point<false>(f, nullptr, flags, ph100, nullptr, c3, K, (size_t)i * n2 + q, q, i);
        return;
    }
    const unsigned row = blockIdx.x - nb_faces;   // (level, row) of interior points
    const int i = 1 + (int)(row / (unsigned)(jm - 2)), j = 1 + (int)(row % (unsigned)(jm - 2));
    const double hight = (double)i * K.topo_step;
    for (int k = 1 + (int)threadIdx.x; k <= km - 2; k += (int)blockDim.x) {
        const size_t q = (size_t)j * km + k, p = (size_t)i * n2 + q;
        const bool land = flags[p] & FL_LAND;
        if (i == 1) ph100[p] = ph100_of(limited<R_T>(f.t[q], false), hight, K);
        if (land) f.p_dyn[p] = 0.;
        if (c3 && i == 3) c3[q] = limited<R_C>(f.c[p], land);
    }
}

// The whole sweep of one wave.  ROLE is a template parameter so that every wave runs
// straight-line code for its own variable; all roles execute the same number of barriers
// (one before the first step, four per step), whatever their lanes do.
//
// Memory schedule of a step (row j): every load (xn, flags and source terms of row j+1, row j+2 of the
// own variable) is issued at the top, the stores of row j-1's results in stage 2, the geometry table
// of row j+1 is read from LDS in stage 4.  Nothing loaded in a step is used before the next step.
// L1: the instantiation for level 1 (the t and c waves only: the evaporation source of c and the dry density it needs),
// so that the other levels carry no test for it inside their stages.
// LOADLIM: the old values of the sweep's own points (the variable's rows 1 .. jm-2, and for the v wave the rows of u it
// reads; p_dyn, which the sweep does not store, has been zeroed on land in memory) pass through the limiter as they are taken over - what a limiter and BC_SolidGround run just before
// the sweep would have left in memory; the rows 0 and jm-1 have been limited in memory by k_atm_limit_prep<.., true>.
template <int ROLE, bool limit, bool L1 = false, bool LOADLIM = false>
__device__ __forceinline__ void sweep_role(const AtmWsArgs& a, const DevConst& K, Shared& sh, const double* tab,
                                           int bx, int i, double* __restrict__ X, const double* __restrict__ XN) {
    // limit: what goes to memory is the value as the limiter that follows the sweep in the model's loop would leave it;
    // the sweep itself goes on with the value it computed, as it does when the limiter runs afterwards.
    constexpr bool isV = ROLE == R_V, isU = ROLE == R_U, isT = ROLE == R_T, isC = ROLE == R_C, isW = ROLE == R_W;
    constexpr bool hasH0I = ROLE == R_U || ROLE == R_CO2 || ROLE == R_ICE || ROLE == R_CLOUD;
    // (Tried in the tolerance form: these roles fetching the column's topography, a 2-D field that stays in the L2, and
    // working h_0_i out themselves instead of reading a 3-D array of it.  8 bytes a point less traffic, and slower: 0.187
    // -> 0.189 ms; the few instructions stand in the row's dependent chain.  ATM_WS_TOPO=1 builds that form.)
    constexpr bool AFFINE_TOPO = ATM_WS_TOPO != 0;
    constexpr bool needs_ph = isT && (!ATM_WS_AFFINE || L1);   // 100 p_h: tolerance form on level 1 only
    const int jm = K.jm, km = K.km;
    const int lane = threadIdx.x & 63;
    const int kk = 1 + bx * 64 + lane;
    const int k = kk <= km - 2 ? kk : km - 2;  // idle lanes shadow the last line and reproduce its bits
    const bool lvl1 = isC && (ATM_WS_AFFINE ? L1 : (i == 1));  // (the exact form keeps the test: its roles are large as they are)
    constexpr int NST = isU ? 3 : (isV || isW) ? 2 : 1;  // stores per step

    // ---- constants of the dependent chain, in vector registers ----
    ThetaK TK = { K.dthe, K.i_dthe, K.two_dthe, K.i_two_dthe, K.dthe2, K.i_dthe2 };
    PIN(TK.dthe); PIN(TK.i_dthe); PIN(TK.two_dthe); PIN(TK.i_two_dthe); PIN(TK.dthe2); PIN(TK.i_dthe2);
    double rm = a.rad[i];
    double rm2 = rm * rm, i_rm = 1. / rm, i_rm2 = 1. / rm2;
    PIN(rm); PIN(rm2); PIN(i_rm); PIN(i_rm2);
    // divisor of the diffusion bracket and of the damping term of this variable
    double dv, i_dv, dm = 1., i_dm = 1.;
    if constexpr (isT) { dv = K.re_pr; i_dv = K.i_re_pr; }
    else if constexpr (isU) { dv = K.re; i_dv = K.i_re; dm = K.dr2; i_dm = K.i_dr2; }
    else if constexpr (isV) { dv = K.re; i_dv = K.i_re; dm = K.dthe2; i_dm = K.i_dthe2; }
    else if constexpr (isW) { dv = K.re; i_dv = K.i_re; dm = K.dphi2; i_dm = K.i_dphi2; }
    else if constexpr (ROLE == R_CO2) { dv = K.re_sc_co2; i_dv = K.i_re_sc_co2; dm = K.dr2; i_dm = K.i_dr2; }
    else if constexpr (isC) { dv = K.re_sc_wv; i_dv = K.i_re_sc_wv; }
    else { dv = K.re_sc_wv; i_dv = K.i_re_sc_wv; dm = K.dr2; i_dm = K.i_dr2; }
    PIN(dv); PIN(i_dv); PIN(dm); PIN(i_dm);
    double dt = K.dt, hdt = .5 * K.dt, i_six = K.i_six, six = 6.;
    PIN(dt); PIN(hdt); PIN(i_six); PIN(six);
    double k1 = 0., k2 = 0., k3 = 0., k4 = 0., i_k4 = 0.;  // role-specific scalars
    if constexpr (isT) { k1 = K.R_Air; k2 = K.t_0; k3 = K.coeff_energy; }
    if constexpr (isU) { k1 = K.R_WV_over_R_Air; k2 = K.Buoyancy; k3 = K.g; k4 = K.coeff_buoy; }
    if constexpr (isV) { k1 = K.r_air; k2 = K.i_r_air; }
    if constexpr (isC) { k1 = K.R_WV_over_R_Air; k2 = K.coeff_trans; k3 = K.coeff_vapour; k4 = K.two_dr; i_k4 = K.i_two_dr; }
    if constexpr (ROLE == R_ICE || ROLE == R_CLOUD) k2 = K.coeff_trans;
    PIN(k1); PIN(k2); PIN(k3); PIN(k4); PIN(i_k4);
    const uint8_t* __restrict__ FL = a.flags;

    // ---- byte offsets of rows j, j+1, j+2 of this line, advanced by one row per step ----
    const unsigned rowB = (unsigned)km * 8u;
    unsigned o0 = (unsigned)(idx3(jm, km, i, 1, k) * 8);
    unsigned o1 = o0 + rowB, o2 = o1 + rowB;
    unsigned q1 = (unsigned)(((size_t)2 * km + k) * 8);  // row j+1 of a 2-D field
    PIN(o0); PIN(o1); PIN(o2); PIN(q1);
    // source-term fields of the role (bases stay scalar)
    const double *E1 = nullptr, *E2 = nullptr, *E3 = nullptr, *E4 = nullptr, *E5 = nullptr;
    if constexpr (isT) { E1 = a.ph100; E2 = a.S_c; E3 = a.S_r; E4 = a.S_i; E5 = a.S_s; }
    if constexpr (hasH0I) E1 = AFFINE_TOPO ? a.topo : a.h0i;
    if constexpr (ROLE == R_ICE) E2 = a.S_i;
    if constexpr (ROLE == R_CLOUD) E2 = a.S_c;
    if constexpr (isC) { E2 = a.S_v; E1 = a.evap; E3 = a.precip; E4 = a.c; E5 = a.c3; }
    if constexpr (isV) { E1 = a.p_dyn; E2 = a.u; }
    double *O1 = nullptr, *O2 = nullptr;  // side outputs
    if constexpr (isU) { O1 = a.aux_u; O2 = a.buoy; }
    if constexpr (isV) O1 = a.aux_v;
    if constexpr (isW) O1 = a.aux_w;

    double xm2 = 0., xm1 = ldg(X, o0 - rowB), x0 = ldg(X, o0), xp1 = ldg(X, o1);
    double ym2 = 0., ym1 = 0., y0 = 0., yp1 = 0.;  // p_dyn rows (v wave)
    double um2 = 0., um1 = 0., up1 = 0.;            // u rows (v wave): updated j-2, j-1; old j+1
    if constexpr (isV) {
        ym1 = ldg(E1, o0 - rowB); y0 = ldg(E1, o0); yp1 = ldg(E1, o1);
        um1 = ldg(E2, o0 - rowB); up1 = ldg(E2, o1);
    }
    // LOADLIM: the mask byte of row j+2, whose old values are taken over at the end of step j (a step then loads that
    // byte instead of the one of row j+1, which it already has), and the byte of row j+1
    unsigned n_fl2 = 0, fl_nn = 0;
    if constexpr (LOADLIM) {
        fl_nn = ldg_u8(FL, o1 >> 3);
        const bool land1 = ldg_u8(FL, o0 >> 3) & FL_LAND, land2 = fl_nn & FL_LAND;
        x0 = limited<ROLE>(x0, land1); xp1 = limited<ROLE>(xp1, land2);
        if constexpr (isV) up1 = limited<R_U>(up1, land2);   // (p_dyn has been zeroed on land in memory: the sweep does not store it)
    }

    // raw inputs of the next step, loaded during this one
    double n_xn, n_r1 = 0., n_r2 = 0., n_r3 = 0., n_r4 = 0., n_r5 = 0.;
    double n_cos, n_rm2sin, n_irm2sin, n_coefv = 0.;
    uint8_t n_fl;
    auto load_top = [&](unsigned o) {  // xn, flags
        n_xn = ldg_once(XN, o);
        if constexpr (!LOADLIM) n_fl = ldg_u8(FL, o >> 3);
    };
    auto load_sources = [&](unsigned o, unsigned q) {
        if constexpr (isT) { n_r2 = ldg_once(E2, o); n_r3 = ldg_once(E3, o); n_r4 = ldg_once(E4, o); n_r5 = ldg_once(E5, o); }
        if constexpr (hasH0I) n_r1 = ldg(E1, AFFINE_TOPO ? q : o);
        if constexpr (ROLE == R_ICE || ROLE == R_CLOUD || isC) n_r2 = ldg_once(E2, o);
        if constexpr (isC) {
            if (lvl1) { n_r1 = ldg(E1, q); n_r3 = ldg(E3, q); n_r4 = ldg(E4, q); n_r5 = ldg(E5, q); }
        }
    };
    auto load_table = [&](int jj) {
        const double* row = tab + 4 * jj;
        n_cos = row[0]; n_rm2sin = row[1]; n_irm2sin = row[2];
        if constexpr (isV) n_coefv = row[3];
    };
    load_top(o0);
    if constexpr (LOADLIM) n_fl = ldg_u8(FL, o0 >> 3);
    load_sources(o0, q1 - rowB);
    load_table(1);
    // t wave: 100 p_h of rows j and j+1; row j+2 is fetched at the top of step j
    double ph_cur = 0., ph_nxt = 0., n_ph2 = 0.;
    if constexpr (needs_ph) { ph_cur = ldg(E1, o0); ph_nxt = ldg(E1, o1); }

    // what the other waves need of a new t or c value (RHS_Atm.cpp:385, 390-404)
    constexpr bool AFFINE = ATM_WS_AFFINE;
    constexpr bool t_feeds_c = L1;  // affine form: the dry density has one reader, the c wave of level 1
    double k12 = k1 * k2; PIN(k12);
    auto publish_t = [&](int buf, double tval, double ph) {
        if constexpr (AFFINE) {
            if (t_feeds_c) sh.rdry[buf][lane] = ph * rcp_one_step(k12 * tval);  // level 1 is the last to finish: every step of this chain counts
        } else {
            const double r_dry = fdiv_normal(ph, k1 * (tval * k2));
            sh.rdry[buf][lane] = r_dry;
            sh.yr[buf][lane] = rcp_refined(r_dry);
        }
    };
    double c_opc = 0., c_cb = 0., c_cy = 0.;   // c wave: what it published for the value it is working with
    auto publish_c = [&](int buf, double cval) {
        c_cb = 1. + k1 * cval;
        c_opc = 1. + cval;
        c_cy = rcp_refined(c_cb);
        if constexpr (AFFINE) {
            sh.opc[buf][lane] = __builtin_fma(c_opc, c_cy, -1.);  // humid over dry density, minus one
        } else {
            sh.opc[buf][lane] = c_opc;
            sh.cb[buf][lane] = c_cb;
            sh.cy[buf][lane] = c_cy;
        }
    };

    // stage-1 values of row 1
    if constexpr (isV) sh.v[0][lane] = x0;
    if constexpr (isU) sh.u[0][lane] = x0;
    if constexpr (isC) publish_c(0, x0);
    if constexpr (isT) publish_t(0, x0, ph_cur);
    lds_barrier();

    // Everything loaded so far has arrived before the loop starts, so that the only vector-memory
    // wait inside the loop is the one at the end of a step.
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    double so1 = 0., so2 = 0.;  // side outputs of the previous row, stored with it during the next step
    bool land_prev = false;     // ... and whether that row's cell is land

    for (int j = 1; j < jm - 1; j++) {
        // (Idle lanes shadow the last line: they load what its owner loads and store the same bits.)
        STAMP(0);
        if constexpr (LOADLIM) {
            if (j > 1) {   // (rows 1 and 2 were limited before the loop)
                const bool land_p1 = fl_nn & FL_LAND;
                xp1 = limited<ROLE>(xp1, land_p1);
                if constexpr (isV) up1 = limited<R_U>(up1, land_p1);
            }
        }
        // ---- take over the prefetched inputs of this step, start loading the next step's ----
        const double xn = n_xn;
        const uint8_t fl = n_fl;
        const double costhe = n_cos, rm2sin = n_rm2sin, i_rm2sin = n_irm2sin, coef_v = n_coefv;
        double e_a = n_r1, e_b = 0., e_c = 0., c_lvl0 = 0., c_lvl3 = 0., m3c0 = 0., om2c0 = 0.;
        if constexpr (isT) e_a = ph_cur;
        if constexpr (isT) { e_b = k3 * (n_r2 + n_r3); e_c = k3 * (n_r4 + n_r5); }
        if constexpr (ROLE == R_ICE || ROLE == R_CLOUD || isC) e_b = n_r2 * k2;
        if constexpr (isC) {
            if (lvl1) {  // RHS_Atm.cpp:412-421
                double ep = n_r1 - n_r3;
                if (ep >= 6.) ep = 6.;
                if (ep <= -6.) ep = -6.;
                e_a = ep; c_lvl0 = n_r4; c_lvl3 = n_r5;
                m3c0 = -3. * c_lvl0; om2c0 = 1. - 2. * c_lvl0;
            }
        }
#ifdef ATOMDEV_EXP_NOMEM  // timing experiment only (results are wrong): bit 0 no loads, bit 1 no stores inside the loop
        const bool exp_mem = !(ATOMDEV_EXP_NOMEM & 1) || jm < 0, exp_st = !(ATOMDEV_EXP_NOMEM & 2) || jm < 0;
#else
        const bool exp_mem = true, exp_st = true;
#endif
        if (exp_mem) load_top(o1);  // row j+1 <= jm-1 always exists
        double xp2 = 0., yp2 = 0., up2 = 0.;
        if (exp_mem && j + 2 < jm) {
            xp2 = ldg(X, o2);
            if constexpr (LOADLIM) n_fl2 = ldg_u8(FL, o2 >> 3);
            if constexpr (needs_ph) n_ph2 = ldg(E1, o2);
            if constexpr (isV) { yp2 = ldg(E1, o2); up2 = ldg(E2, o2); }
        }
        if (exp_mem) load_sources(o1, q1);
        STAMP(1);
        const bool land = fl & FL_LAND;
        const double h_d_j = ((fl & FL_H0J) && (fl & FL_H0K)) ? 0. : 1.;  // RHS_Atm.cpp:129-149
        const bool window = (j >= 2) && (j < jm - 3);
        const int mode = !window ? 0 : ((fl & FL_ONESIDED_S) ? 2 : 1);
        const bool all_bw = __all(mode == 1);  // whole wave on the two-point backward form: d2 == 0
        const ThetaForm TF = theta_form(mode, TK);
        double xB = 0., xC = 0., xF = 0., xG = 0., uB = 0., uC = 0.;
        if (!all_bw) theta_terms(mode, xm2, xm1, xp1, xB, xC, xF, xG);

        double pg_v = 0., h_0 = 0.;
        if constexpr (isV) {
            h_0 = (fl & FL_H0J) ? 1. : 0.;
            if (j > 1) um1 = sh.ufin[lane];  // updated u of row j-1 (row 0 is a boundary row)
            double pB, pC, pF, pG;
            theta_terms(mode, ym2, ym1, yp1, pB, pC, pF, pG);
            const double dpdthe = DIVC(h_d_j * ((TF.a * y0 + pB) + pC), TF.D1, TF.iD1);
            pg_v = DIVC(DIVC(h_d_j * dpdthe, rm, i_rm), k1, k2);  // h_d_j*dpdthe/rm/r_air
            theta_terms(mode, um2, um1, up1, uB, uC, pF, pG);
        }
        if constexpr (isW) h_0 = (fl & FL_H0K) ? 1. : 0.;
        if constexpr (hasH0I) h_0 = AFFINE_TOPO ? h0i_of(fl, (double)i * K.topo_step, e_a, K) : e_a;  // h_0_i

        double st = x0, acc = 0., rhs = 0., RS = 0.;
        // ---- affine form: coefficients of the row ----
        double cA = 0., cB = 0., cP = 0., cQ = 0., cR = 0., cKK = 0., cFac = 0., cMc = 0.;
        if constexpr (AFFINE) {
            double A1, B1, Pb, Qb, B1u = 0.;
            const double g = costhe * i_rm2sin;
            if (all_bw) {
                A1 = h_d_j * TK.i_dthe;
                B1 = -xm1 * A1;
                Pb = A1 * g; Qb = B1 * g;
                if constexpr (isV) B1u = -um1 * A1;
            } else {
                const double hD = h_d_j * TF.iD1, hE = h_d_j * TK.i_dthe2;
                A1 = TF.a * hD; B1 = (xB + xC) * hD;
                const double A2 = TF.e * hE, B2 = (xF + xG) * hE;
                Pb = __builtin_fma(A2, i_rm2, A1 * g); Qb = __builtin_fma(B2, i_rm2, B1 * g);
                if constexpr (isV) B1u = (uB + uC) * hD;
            }
            cA = A1 * i_rm; cB = B1 * i_rm;
            cP = Pb * i_dv; cQ = Qb * i_dv;
            if constexpr (isT) cQ += e_b + e_c;
            if constexpr (isU || isW || ROLE == R_CO2 || ROLE == R_ICE || ROLE == R_CLOUD) cP = __builtin_fma(-h_0, i_dm, cP);
            if constexpr (ROLE == R_ICE || ROLE == R_CLOUD || isC) cQ += e_b;
            if constexpr (isU) cKK = k2 * k3;
            if constexpr (isV) {
                const double two = 2. * i_rm2 * i_dv;
                cP = cP - coef_v * h_d_j * i_dv - h_0 * i_dm;
                cR = two * A1;
                cQ = __builtin_fma(two, B1u, cQ) - pg_v;
            }
            if constexpr (isC) {
                if (lvl1) {  // RHS_Atm.cpp:412-430
                    cFac = land ? 0. : k3 * i_k4 * om2c0 * e_a;
                    cMc = m3c0 - c_lvl3;
                }
            }
        }
        auto store_prev = [&]() {  // results of the previous row
            stg(X, o0 - rowB, limit ? limited<ROLE>(xm1, land_prev) : xm1);
            if constexpr (isU) { stg_once(O1, o0 - rowB, so1); stg_once(O2, o0 - rowB, so2); }
            if constexpr (isV || isW) stg_once(O1, o0 - rowB, so1);
        };
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const int rb = s & 1, wb = rb ^ 1;
            if (exp_st && s == 1 && j > 1) store_prev();
            if (s == 3) load_table(j + 1);
            double v;
            if constexpr (isV) v = st; else v = sh.v[rb][lane];
          if constexpr (AFFINE) {
            const double t1 = __builtin_fma(cA, st, cB);
            double lin = __builtin_fma(cP, st, cQ);
            if constexpr (isU) { RS = cKK * sh.opc[rb][lane]; lin -= RS; }
            if constexpr (isV) lin = __builtin_fma(cR, sh.u[rb][lane], lin);
            if constexpr (isC) {
                if (lvl1) {
                    const double r_humid = sh.rdry[rb][lane] * (c_opc * c_cy);
                    lin = __builtin_fma(r_humid * __builtin_fma(4., st, cMc), cFac, lin);
                }
            }
            rhs = __builtin_fma(-v, t1, lin);
          } else {
            double d1, d2 = 0., brk;
            if (all_bw) {
                d1 = DIVC(h_d_j * (st - xm1), TK.dthe, TK.i_dthe);
                if constexpr (isV || isW) brk = DIVC(d1, rm2sin, i_rm2sin) * costhe;
                else brk = DIVC(d1 * costhe, rm2sin, i_rm2sin);
            } else {
                d1 = DIVC(h_d_j * ((TF.a * st + xB) + xC), TF.D1, TF.iD1);
                d2 = DIVC(h_d_j * ((TF.e * st + xF) + xG), TK.dthe2, TK.i_dthe2);
                if constexpr (isV || isW) brk = DIVC(d2, rm2, i_rm2) + DIVC(d1, rm2sin, i_rm2sin) * costhe;
                else brk = DIVC(d2, rm2, i_rm2) + DIVC(d1 * costhe, rm2sin, i_rm2sin);
            }
            const double adv = DIVC(v * d1, rm, i_rm);

            if constexpr (isT) {
                rhs = -adv + DIVC(brk, dv, i_dv) + e_b + e_c;
            } else if constexpr (isU) {
                // buoyancy, RHS_Atm.cpp:390-404, from the current-stage dry density and vapour
                const double r_dry = sh.rdry[rb][lane];
                const double r_humid = DIVC(r_dry * sh.opc[rb][lane], sh.cb[rb][lane], sh.cy[rb][lane]);
                RS = DIVC(k2 * (r_humid - r_dry), r_dry, sh.yr[rb][lane]) * k3;
                rhs = -adv + DIVC(brk, dv, i_dv) - RS - DIVC(h_0 * st, dm, i_dm);
            } else if constexpr (isV) {
                const double d1u = DIVC(h_d_j * ((TF.a * sh.u[rb][lane] + uB) + uC), TF.D1, TF.iD1);
                rhs = -adv + -pg_v
                      + DIVC(brk - coef_v * h_d_j * v + DIVC(2. * d1u, rm2, i_rm2), dv, i_dv)
                      - DIVC(h_0 * v, dm, i_dm);
            } else if constexpr (isW || ROLE == R_CO2) {
                rhs = -adv + DIVC(brk, dv, i_dv) - DIVC(h_0 * st, dm, i_dm);
            } else if constexpr (ROLE == R_ICE || ROLE == R_CLOUD) {
                rhs = -adv + DIVC(brk, dv, i_dv) + e_b - DIVC(h_0 * st, dm, i_dm);
            } else {  // R_C
                rhs = -adv + DIVC(brk, dv, i_dv) + e_b;
                if (lvl1) {  // RHS_Atm.cpp:412-430: surface evaporation source, c at levels 0, 1 (own), 3
                    // 1 + c, 1 + k c and the refined reciprocal of the latter are what this wave published for st
                    const double r_dry = sh.rdry[rb][lane];
                    const double r_humid = DIVC(r_dry * c_opc, c_cb, c_cy);
                    double vapour_surface = DIVC(r_humid * (m3c0 + 4. * st - c_lvl3), k4, i_k4) * om2c0 * e_a;
                    double vapour_evaporation = k3 * vapour_surface;
                    if (land) vapour_evaporation = 0.;
                    rhs = rhs + vapour_evaporation;
                }
            }
          }
            // RungeKutta_Atm.cpp:58-152.  rhs * .5 * dt == rhs * (.5 * dt) and acc + 2. * rhs == fma(2, rhs, acc)
            // bit for bit (scaling by a power of two is exact), one instruction less each.
            if (s == 0) { acc = rhs; st = xn + rhs * hdt; }
            else if (s == 1) { acc = __builtin_fma(2., rhs, acc); st = xn + rhs * hdt; }
            else if (s == 2) { acc = __builtin_fma(2., rhs, acc); st = xn + rhs * dt; }
            else st = xn + DIVC(dt * (acc + rhs), six, i_six);

            // publish: the next stage's value, or (after the last stage) stage 1 of the next row
            const double pub = (s < 3) ? st : xp1;
            if constexpr (isV) sh.v[wb][lane] = pub;
            if constexpr (isU) { sh.u[wb][lane] = pub; if (s == 3) sh.ufin[lane] = st; }
            if constexpr (isC) publish_c(wb, pub);
            if constexpr (isT) publish_t(wb, pub, (s < 3) ? ph_cur : ph_nxt);
            STAMP(2 + 2 * s);
            lds_barrier();
            STAMP(3 + 2 * s);
        }

        // ---- side outputs hold the 4th evaluation (RHS_Atm.cpp:406-410, 485-492) ----
        if constexpr (isU) { so1 = land ? 0. : rhs; so2 = land ? 0. : -RS * k4 * 1000.; }
        if constexpr (isV) so1 = land ? 0. : rhs + pg_v;
        if constexpr (isW) so1 = land ? 0. : rhs;
        // Vector-memory operations complete in issue order: the step's loads were issued at its top and
        // its stores after them, so waiting until only the stores may be outstanding completes the loads
        // (four rounds old) without waiting for the write acknowledgements, and leaves no pending state
        // for the next step's first uses.
        if (j > 1) __builtin_amdgcn_s_waitcnt(0x0F70 | NST);  // vmcnt(NST)
        else __builtin_amdgcn_s_waitcnt(0x0F70);              // vmcnt(0): the first step stores nothing
        STAMP(10);
        if constexpr (LOADLIM) {
            // (the loaded values are taken as they are here, behind the wait: worked on any earlier, the compiler waits
            // for the step's loads in the middle of the stages; the limiting itself is the next step's, which has room
            // for it - inside the window of backward differences row j+1 is not needed before the step's last stage)
            asm volatile("" : "+v"(xp2), "+v"(n_fl2));
            if constexpr (isV) asm volatile("" : "+v"(up2));
            n_fl = (uint8_t)fl_nn; fl_nn = n_fl2;
        }
        xm2 = xm1; xm1 = st; x0 = xp1; xp1 = xp2;
        land_prev = land;
        if constexpr (isV) { ym2 = ym1; ym1 = y0; y0 = yp1; yp1 = yp2; um2 = um1; up1 = up2; }
        if constexpr (isT) { ph_cur = ph_nxt; ph_nxt = n_ph2; }
        o0 += rowB; o1 += rowB; o2 += rowB; q1 += rowB;
    }
    stg(X, o0 - rowB, limit ? limited<ROLE>(xm1, land_prev) : xm1);  // row jm-2
    if constexpr (isU) { stg_once(O1, o0 - rowB, so1); stg_once(O2, o0 - rowB, so2); }
    if constexpr (isV || isW) stg_once(O1, o0 - rowB, so1);
}

template <bool LIMIT, bool LOADLIM = false>
__global__ __launch_bounds__(512) void k_atm_rk3d_ws(const AtmWsParams* __restrict__ P) {
    const AtmWsArgs& a = P->a;
    const DevConst& K = P->K;
    __shared__ Shared sh;
    extern __shared__ double tab[];  // [jm][4]: cos(the), rm^2 sin(the), its reciprocal, viscous v coefficient
    // Workgroup -> (level, longitude block).  Consecutive workgroup ids go to consecutive XCDs, each
    // with its own L2; the 512-byte row segments of neighbouring longitude blocks share cache lines
    // (rows are not line-aligned), so all blocks of one level are given to the same XCD.
    const int gx = (K.km - 2 + 63) / 64;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int lev = xcd + 8 * (idx / gx), bx = idx % gx;
    if (lev >= K.im - 2) return;  // padding workgroups of the last round of levels
    const int i = 1 + lev;
#ifdef ATOMDEV_EXP_WGTIME  // timing experiment only: duration of every workgroup goes to level 0 of aux_u
    const long long wg_t0 = __builtin_readcyclecounter();
    const long long wg_w0 = wall_clock64();
#endif
    {
        const int jm = K.jm;
        for (int j = threadIdx.x; j < jm; j += blockDim.x) {
            tab[4 * j + 0] = a.costhe[j];
            tab[4 * j + 1] = a.tab_rm2sin[i * jm + j];
            tab[4 * j + 2] = a.tab_irm2sin[i * jm + j];
            tab[4 * j + 3] = a.tab_coefv[i * jm + j];
        }
        __syncthreads();
    }
    switch (threadIdx.x >> 6) {
        case R_U: sweep_role<R_U, LIMIT, false, LOADLIM>(a, K, sh, tab, bx, i, a.u, a.un); break;
        case R_V: sweep_role<R_V, LIMIT, false, LOADLIM>(a, K, sh, tab, bx, i, a.v, a.vn); break;
        case R_T:
            if (ATM_WS_AFFINE && i == 1) sweep_role<R_T, LIMIT, ATM_WS_AFFINE != 0, LOADLIM>(a, K, sh, tab, bx, i, a.t, a.tn);
            else sweep_role<R_T, LIMIT, false, LOADLIM>(a, K, sh, tab, bx, i, a.t, a.tn);
            break;
        case R_W: sweep_role<R_W, LIMIT, false, LOADLIM>(a, K, sh, tab, bx, i, a.w, a.wn); break;
        case R_CO2: sweep_role<R_CO2, LIMIT, false, LOADLIM>(a, K, sh, tab, bx, i, a.co2, a.co2n); break;
        case R_ICE: sweep_role<R_ICE, LIMIT, false, LOADLIM>(a, K, sh, tab, bx, i, a.ice, a.icen); break;
        case R_CLOUD: sweep_role<R_CLOUD, LIMIT, false, LOADLIM>(a, K, sh, tab, bx, i, a.cloud, a.cloudn); break;
        default:
            if (ATM_WS_AFFINE && i == 1) sweep_role<R_C, LIMIT, ATM_WS_AFFINE != 0, LOADLIM>(a, K, sh, tab, bx, i, a.c, a.cn);
            else sweep_role<R_C, LIMIT, false, LOADLIM>(a, K, sh, tab, bx, i, a.c, a.cn);
            break;
    }
#ifdef ATOMDEV_EXP_WGTIME
    if ((threadIdx.x & 63) == 0) {
        a.aux_u[(lev * gx + bx) * 8 + (threadIdx.x >> 6)] = (double)(__builtin_readcyclecounter() - wg_t0);
        if (threadIdx.x == 0) { a.aux_u[16384 + lev * gx + bx] = (double)wg_w0; a.aux_u[32768 + lev * gx + bx] = (double)wall_clock64(); }
    }
#endif
}

}  // namespace

int ATM_WS_ENTRY(atomdev* h, bool limit, bool limit_before) {
    AtmWsParams P;
    AtmWsArgs& a = P.a;
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
    a.ph100 = h->scr[0]; a.h0i = h->scr[1]; a.topo = h->g[ATOMDEV_TOPOGRAPHY];
    a.rad = h->rad; a.costhe = h->costhe;
    a.tab_rm2sin = h->tab_rm2sin; a.tab_irm2sin = h->tab_irm2sin; a.tab_coefv = h->tab_coefv;
    P.K = h->k;
    // field pointers of the atmosphere never change after creation: one upload per handle
    if (!h->ws_params) {
        ATOMDEV_TRY(hipMalloc(&h->ws_params, sizeof(P)));
        ATOMDEV_TRY(hipMemcpy(h->ws_params, &P, sizeof(P), hipMemcpyHostToDevice));
    }
    dim3 pg((h->km + 127) / 128, h->jm, h->im - 2);
    // c at level 3 as the level-1 points must see it: the preparing kernel copies it where it passes level 3
    double* c3 = nullptr;
    if (h->im >= 5) c3 = h->plane;
    else ATOMDEV_TRY(hipMemcpyAsync(h->plane, h->f[ATOMDEV_C] + 3 * h->n2, h->n2 * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    if (h->profile_rk == 2)
        if (int rc = atomdev_prof_begin(h)) return rc;  // the bracket of the whole operator
    // tolerance form, wave-specialised sweep: the limiter in front of the sweep touches the six faces only and the sweep
    // limits the old values of its own points as it takes them over (same bits; the pass over all points is 36 us)
    const bool load_limited = ATM_WS_AFFINE && limit_before && limit && h->rk_variant == 2 && !getenv("ATOMDEV_NO_LOAD_LIMIT");
    if (limit_before) {
        LimitFields lf = { h->f[ATOMDEV_U], h->f[ATOMDEV_V], h->f[ATOMDEV_W], h->f[ATOMDEV_T], h->f[ATOMDEV_C],
                           h->f[ATOMDEV_CLOUD], h->f[ATOMDEV_ICE], h->f[ATOMDEV_CO2], h->f[ATOMDEV_P_DYN] };
        const unsigned nb = (unsigned)((h->n3 + 255) / 256);
        if (load_limited && (ATM_WS_TOPO || h->h0i_valid))
        {
            const size_t faces = 2 * h->n2 + (size_t)(h->im - 2) * (2 * (size_t)h->km + 2 * (size_t)(h->jm - 2));
            ATOMDEV_REQUIRE(faces < ((size_t)1 << 31), "atm_rk3d: the six faces hold more than 2^31 points");
            const unsigned nb_faces = (unsigned)((faces + 255) / 256);
            hipLaunchKernelGGL(k_atm_limit_faces_prep, dim3(nb_faces + (unsigned)(h->im - 2) * (unsigned)(h->jm - 2)), dim3(256), 0, h->stream, lf,
                               h->flags, h->scr[0], c3, h->k, nb_faces);
        }
        else if (ATM_WS_TOPO || h->h0i_valid)
            hipLaunchKernelGGL(k_atm_limit_prep<false>, dim3(nb), dim3(256), 0, h->stream, lf, h->g[ATOMDEV_TOPOGRAPHY], h->flags,
                               h->scr[0], h->scr[1], c3, h->k, h->n3);
        else
            hipLaunchKernelGGL(k_atm_limit_prep<true>, dim3(nb), dim3(256), 0, h->stream, lf, h->g[ATOMDEV_TOPOGRAPHY], h->flags,
                               h->scr[0], h->scr[1], c3, h->k, h->n3);
    } else if (ATM_WS_TOPO || h->h0i_valid)
        hipLaunchKernelGGL(k_atm_prep<false>, pg, dim3(128), 0, h->stream, h->f[ATOMDEV_T], h->g[ATOMDEV_TOPOGRAPHY], h->flags,
                           h->scr[0], h->scr[1], h->f[ATOMDEV_C], c3, h->k);
    else
        hipLaunchKernelGGL(k_atm_prep<true>, pg, dim3(128), 0, h->stream, h->f[ATOMDEV_T], h->g[ATOMDEV_TOPOGRAPHY], h->flags,
                           h->scr[0], h->scr[1], h->f[ATOMDEV_C], c3, h->k);
    ATOMDEV_TRY(hipGetLastError());
    const bool h0i_was_valid = h->h0i_valid;   // (the first call of a mask takes the pass that also prepares h_0_i)
    if (!ATM_WS_TOPO) h->h0i_valid = true;
    if (h->profile_rk == 1)
        if (int rc = atomdev_prof_begin(h)) return rc;  // the bracket of the sweep kernel alone
    const unsigned gx = (h->km - 2 + 63) / 64, rounds = (h->im - 2 + 7) / 8;
    dim3 grid(8 * rounds * gx);
    ATOMDEV_REQUIRE(h->n3 * sizeof(double) < ((size_t)1 << 32), "atm_rk3d: field larger than 4 GiB");
    const size_t tab_bytes = (size_t)h->jm * 4 * sizeof(double);
    if (h->rk_variant == 3) {  // the pipelined sweep takes the same prepared inputs
        if (int rc = atomdev_atm_rk3d_pipe_launch(h, limit)) return rc;
    } else if (load_limited && (ATM_WS_TOPO || h0i_was_valid))
        hipLaunchKernelGGL((k_atm_rk3d_ws<true, ATM_WS_AFFINE != 0>), grid, dim3(512), tab_bytes, h->stream, (const AtmWsParams*)h->ws_params);
    else if (limit)
        hipLaunchKernelGGL(k_atm_rk3d_ws<true>, grid, dim3(512), tab_bytes, h->stream, (const AtmWsParams*)h->ws_params);
    else
        hipLaunchKernelGGL(k_atm_rk3d_ws<false>, grid, dim3(512), tab_bytes, h->stream, (const AtmWsParams*)h->ws_params);
    ATOMDEV_TRY(hipGetLastError());
    if (int rc = atomdev_prof_end(h)) return rc;
    return 0;
}
