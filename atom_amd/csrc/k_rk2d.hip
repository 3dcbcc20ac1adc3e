// Surface (2-D) Runge-Kutta sweeps: solveRungeKutta_2D_Atmosphere + RK_RHS_2D_Atmosphere
// (atmosphere/RungeKutta_Atm.cpp:163-217, RHS_Atm.cpp:498-678) and the hydrosphere twins
// (hydrosphere/RungeKutta_Hyd.cpp:130-186, RHS_Hyd.cpp:453-653).
//
// Unlike the 3-D sweep, theta AND phi differences are live here, so the in-place
// lexicographic (j, k) sweep is a true 2-D Gauss-Seidel: point (j, k) reads the
// updated values at (j-1, k), (j-2, k), (j, k-1), (j, k-2) and the old values at
// (j+1, k), (j+2, k), (j, k+1), (j, k+2).  All points of an anti-diagonal
// j + k = d are mutually independent and see exactly those values if diagonals are
// processed in increasing d: the sweep is a chain of jm + km - 4 dependent point
// evaluations, and what it costs is the time of ONE point evaluation times that length.
// So the work is split to make a point short, not to make the machine full:
//   * everything of a point that does not depend on results of the sweep (which of the
//     reference's difference forms survives, its coefficients, the old neighbour values,
//     the pressure gradient) is computed for all points at once by k_rk2d_prepare;
//   * the chain itself runs as waves of 32 rows with TWO lanes per row, one for v and one
//     for w (the two right-hand sides have the same shape), which swap their stage values
//     with a lane permutation; a wave is a workgroup of its own, on its own compute unit,
//     and hands its last row to the wave of the next 32 rows through memory.  That
//     dependency points one way only, so the waves run as a pipeline and the latency of
//     the hand-over is paid once per wave, not once per diagonal.
#include <vector>

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
struct PointIn {
    double vc, wc, vn, wn;                                   // centre: current and "new"-array values
    double v_tp, v_tm, v_pp, v_pm, w_tp, w_tm, w_pp, w_pm;   // theta +-1, phi +-1 neighbours
    double v_pp2, v_pm2, w_pp2, w_pm2;                       // phi +-2 (read only inside window_k)
    double p_c, p_tp, p_tm, p_pp, p_pm;
    unsigned m;                                              // land tests, M_* bits (the +-2 ones set outside window_k)
};
enum : unsigned { M_L = 1, M_FLUID_C = 2, M_TP = 4, M_TM = 8, M_PP = 16, M_PM = 32, M_PP2 = 64, M_PM2 = 128 };
struct PointOut { double v, w, aux_v, aux_w; };

// What the mask, the old values and the geometry decide about a point before any stage runs.
struct PointCoef {
    Diff T1, P1, T2;                      // theta first, phi first, "theta second" difference
    double h_0_j, h_d_j, h_0_k, h_d_k;
    bool o_fwd, o_bwd;                    // T2 is the phi one-sided form on the + / - side
    double pg_v, pg_w;                    // pressure gradient terms (p does not change during the sweep)
};

__device__ __forceinline__ PointCoef point_coef(unsigned m, double p_c, double p_tp, double p_tm, double p_pp, double p_pm,
                                                const RowGeo& G, bool win_j, bool win_k, const Rk2dArgs& a, const DevConst& K) {
    PointCoef C;
    const bool L = m & M_L, fluid_c = m & M_FLUID_C;
    const bool l_tp = m & M_TP, l_tm = m & M_TM, l_pp = m & M_PP, l_pm = m & M_PM, l_pp2 = m & M_PP2, l_pm2 = m & M_PM2;
    if (fluid_c && (l_tp || l_tm)) { C.h_0_j = 1.; C.h_d_j = 0.; } else { C.h_0_j = 0.; C.h_d_j = 1.; }
    if (fluid_c && (l_pp || l_pm)) { C.h_0_k = 1.; C.h_d_k = 0.; } else { C.h_0_k = 0.; C.h_d_k = 1.; }
    // central {0, 1, -1} / 2 d; forward two-point {-1, 1, 0} / d; backward two-point {1, 0, -1} / d (wins)
    const bool t_fwd = win_j && L && !l_tp, t_bwd = win_j && L && !l_tm;
    const bool p_fwd = L && !l_pp, p_bwd = win_k ? (L && !l_pm) : (fluid_c && l_pm);
    const int t_sel = t_bwd ? 2 : (t_fwd ? 1 : 0), p_sel = p_bwd ? 2 : (p_fwd ? 1 : 0);
    Diff &T1 = C.T1, &P1 = C.P1, &T2 = C.T2;
    T1.cc = t_sel == 2 ? 1. : (t_sel == 1 ? -1. : 0.); T1.c1 = t_sel == 2 ? 0. : 1.; T1.c2 = t_sel == 1 ? 0. : -1.;
    T1.h = C.h_d_j; T1.div = t_sel ? K.dthe : K.two_dthe; T1.idiv = t_sel ? K.i_dthe : K.i_two_dthe;
    P1.cc = p_sel == 2 ? 1. : (p_sel == 1 ? -1. : 0.); P1.c1 = p_sel == 2 ? 0. : 1.; P1.c2 = p_sel == 1 ? 0. : -1.;
    P1.h = C.h_d_k; P1.div = p_sel ? K.dphi : K.two_dphi; P1.idiv = p_sel ? a.i_dphi : a.i_two_dphi;
    // theta second derivative and its two neighbour operands: zero {0, 0, 0} inside window_j, central {-2, 1, 1}
    // outside it, zero again outside window_k; inside window_k the PHI one-sided form {2, -2, 1} on (k, k+-1, k+-2)
    // where the centre is land and both cells on that side are fluid (the s = -1 side wins)
    C.o_fwd = win_k && L && !l_pp && !l_pp2; C.o_bwd = win_k && L && !l_pm && !l_pm2;
    const bool one_sided = C.o_fwd || C.o_bwd, central2 = !win_j && win_k && !one_sided;
    T2.cc = one_sided ? 2. : (central2 ? -2. : 0.); T2.c1 = one_sided ? -2. : (central2 ? 1. : 0.);
    T2.c2 = (one_sided || central2) ? 1. : 0.;
    T2.h = one_sided ? C.h_d_k : C.h_d_j; T2.div = one_sided ? K.dphi2 : K.dthe2; T2.idiv = one_sided ? K.i_dphi2 : K.i_dthe2;
    const double dpdthe = diff(T1, p_c, p_tp, p_tm), dpdphi = diff(P1, p_c, p_pp, p_pm);
    C.pg_v = div_by(div_by(C.h_d_j * dpdthe, a.rm, a.i_rm), a.rho, a.i_rho);         // h_d_j * dpdthe / rm / rho
    C.pg_w = div_by(div_by(C.h_d_k * dpdphi, G.rmsin, G.i_rmsin), a.rho, a.i_rho);
    return C;
}

// All four RK stages at one surface point (RungeKutta_Atm.cpp:176-214 around RHS_Atm.cpp:498-678), v and w in
// one thread: the form the row-major kernel k_rk2d uses.
__device__ __forceinline__ PointOut rk_point(const PointIn& in, const RowGeo& G, bool win_j, bool win_k,
                                             const Rk2dArgs& a, const DevConst& K) {
    const double dt = K.dt;
    const double rm = a.rm, rm2 = rm * rm;
    const PointCoef C = point_coef(in.m, in.p_c, in.p_tp, in.p_tm, in.p_pp, in.p_pm, G, win_j, win_k, a, K);
    const Diff &T1 = C.T1, &P1 = C.P1, &T2 = C.T2;
    const double h_0_j = C.h_0_j, h_d_j = C.h_d_j, h_0_k = C.h_0_k, h_d_k = C.h_d_k;
    const bool o_fwd = C.o_fwd, o_bwd = C.o_bwd;
    // (operands copied out of the struct as opaque values: a select between two struct members otherwise becomes a
    // load through a selected address and pins the whole struct in scratch memory)
    double s_tp[2] = { in.v_tp, in.w_tp }, s_tm[2] = { in.v_tm, in.w_tm }, s_pp[2] = { in.v_pp, in.w_pp };
    double s_pm[2] = { in.v_pm, in.w_pm }, s_pp2[2] = { in.v_pp2, in.w_pp2 }, s_pm2[2] = { in.v_pm2, in.w_pm2 };
#pragma unroll
    for (int n = 0; n < 2; n++) {
        asm volatile("" : "+v"(s_tp[n]), "+v"(s_tm[n]), "+v"(s_pp[n]), "+v"(s_pm[n]), "+v"(s_pp2[n]), "+v"(s_pm2[n]));
    }
    const double v2A = o_bwd ? s_pm[0] : (o_fwd ? s_pp[0] : s_tp[0]), v2B = o_bwd ? s_pm2[0] : (o_fwd ? s_pp2[0] : s_tm[0]);
    const double w2A = o_bwd ? s_pm[1] : (o_fwd ? s_pp[1] : s_tp[1]), w2B = o_bwd ? s_pm2[1] : (o_fwd ? s_pp2[1] : s_tm[1]);
    const double pg_v = C.pg_v, pg_w = C.pg_w;

    double vc = in.vc, wc = in.wc;
    const double vn = in.vn, wn = in.wn;
    double kv[4], kw[4];
#pragma unroll
    for (int s = 0; s < 4; s++) {
        const double dvdthe = diff(T1, vc, in.v_tp, in.v_tm), dwdthe = diff(T1, wc, in.w_tp, in.w_tm);
        const double dvdphi = diff(P1, vc, in.v_pp, in.v_pm), dwdphi = diff(P1, wc, in.w_pp, in.w_pm);
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
    PointOut o;
    o.v = vn + div_by(dt * (kv[0] + 2. * kv[1] + 2. * kv[2] + kv[3]), 6., K.i_six);
    o.w = wn + div_by(dt * (kw[0] + 2. * kw[1] + 2. * kw[2] + kw[3]), 6., K.i_six);
    o.aux_v = kv[3] + pg_v; o.aux_w = kw[3] + pg_w;
    return o;
}

// M_* bits from the flags of the centre and its four neighbours; the +-2 bits start set
__device__ __forceinline__ unsigned land_mask(uint8_t fc, int hyd, uint8_t tp, uint8_t tm, uint8_t pp, uint8_t pm) {
    const bool L = (fc & FL_LAND) != 0;
    const bool fluid_c = hyd ? ((fc & FL_HZERO) != 0) : !L;
    return (L ? M_L : 0u) | (fluid_c ? M_FLUID_C : 0u) | ((tp & FL_LAND) ? M_TP : 0u) | ((tm & FL_LAND) ? M_TM : 0u)
           | ((pp & FL_LAND) ? M_PP : 0u) | ((pm & FL_LAND) ? M_PM : 0u) | M_PP2 | M_PM2;
}

__device__ __forceinline__ RowGeo row_geometry(const Rk2dArgs& a, int j) {
    const double rm = a.rm, rm2 = rm * rm;
    const double sinthe = a.sinthe[j], sinthe2 = sinthe * sinthe, costhe = a.costhe[j];
    RowGeo g;
    g.costhe = costhe;
    g.rmsin = rm * sinthe; g.rm2sin = rm2 * sinthe; g.rm2sin2 = rm2 * sinthe2;
    g.i_rmsin = 1. / g.rmsin; g.i_rm2sin = 1. / g.rm2sin; g.i_rm2sin2 = 1. / g.rm2sin2;
    g.coef = 1. + costhe * costhe / sinthe2;
    return g;
}

// The plain form (atomdev_set_rk_variant(1), kept for A/B runs: same bits): one workgroup walks the diagonals
// with a barrier between them, the fields stay in their row-major layout, a thread takes whatever rows of the
// diagonal fall to it, and every neighbour goes through global memory (the barrier orders the writes of
// diagonal d before the reads of d + 1).
__global__ __launch_bounds__(256) void k_rk2d(Rk2dArgs a, DevConst K) {
    extern __shared__ RowGeo geo[];  // [jm]
    const int jm = K.jm, km = K.km;
    for (int j = threadIdx.x; j < jm; j += blockDim.x) geo[j] = row_geometry(a, j);
    __syncthreads();
    const long R = km;
    const int dmax = (jm - 2) + (km - 2);
    for (int d = 2; d <= dmax; d++) {
        const int jlo = max(1, d - (km - 2)), jhi = min(jm - 2, d - 1);
        for (int j = jlo + (int)threadIdx.x; j <= jhi; j += blockDim.x) {
            const int k = d - j;
            const size_t q = (size_t)j * km + k;
            const bool win_j = (j >= 2) && (j < jm - 3), win_k = (k >= 2) && (k < km - 3);
            PointIn in;
            in.v_tp = a.v[q + R]; in.v_tm = a.v[q - R]; in.v_pp = a.v[q + 1]; in.v_pm = a.v[q - 1];
            in.w_tp = a.w[q + R]; in.w_tm = a.w[q - R]; in.w_pp = a.w[q + 1]; in.w_pm = a.w[q - 1];
            in.p_tp = a.p[q + R]; in.p_tm = a.p[q - R]; in.p_pp = a.p[q + 1]; in.p_pm = a.p[q - 1]; in.p_c = a.p[q];
            in.m = land_mask(a.fl[q], a.hyd, a.fl[q + R], a.fl[q - R], a.fl[q + 1], a.fl[q - 1]);
            in.v_pp2 = in.v_pm2 = in.w_pp2 = in.w_pm2 = 0.;
            if (win_k) {
                if (!(a.fl[q + 2] & FL_LAND)) in.m &= ~M_PP2;
                if (!(a.fl[q - 2] & FL_LAND)) in.m &= ~M_PM2;
                in.v_pp2 = a.v[q + 2]; in.v_pm2 = a.v[q - 2]; in.w_pp2 = a.w[q + 2]; in.w_pm2 = a.w[q - 2];
            }
            in.vn = a.vn[q]; in.wn = a.wn[q];
            in.vc = a.v[q]; in.wc = a.w[q];
            const PointOut o = rk_point(in, geo[j], win_j, win_k, a, K);
            a.v[q] = o.v; a.w[q] = o.w; a.aux_v[q] = o.aux_v; a.aux_w[q] = o.aux_w;
        }
        __syncthreads();
    }
}

// ---- the sweep as a pipeline of narrow waves ----
constexpr int CH_ROWS = 32;   // rows of a chain wave (two lanes per row: v and w)
constexpr int PK = 13;        // 16-byte entries of a lane's packet for one point
constexpr int PKF = 8;        // ... in the tolerance form (the packets keep the stride of PK entries)

// Packet of lane x (0: v, 1: w) for point (j, k), as doubles:
//   0 xc   1 xn                                  old centre value, "new"-array value
//   2 t1.cc  3 t1.c1 * x(j+1,k)  4 t1.c2  5 t1.h  6 t1.div  7 t1.idiv
//   8 p1.cc  9 p1.c1 * x(j,k+1) 10 p1.c2 11 p1.h 12 p1.div 13 p1.idiv
//  14 t2.cc 15 t2.c1 16 t2.c2 17 t2.h 18 t2.div 19 t2.idiv
//  20 sA  21 sB      T2's operands where they are old values: x(j,k+1), x(j,k+2) (o_fwd) or x(j+1,k) (central)
//  22 pg_x  23 coef * h_d_x  24 h_0_x
//  25 (as two 32-bit integers) low word 1: T2's operands are this sweep's x(j,k-1), x(j,k-2); high word 1: operand
//     B is x(j-1,k)
// Group g holds rows 1 + 32 g ... and stores the packets of its step n (row r of the group is at k = n + 1 - r)
// together: pk[((g * ndd + n) * PK + e) * 64 + lane], 1 KB per entry and wave.
struct Chain {
    double2* pk;
    double* bnd;   // [group][x][k]: the row above the group's first row (row 0 itself for group 0)
    int ndd;       // steps stored per group
};
constexpr unsigned long long BND_EMPTY = 0xFFF4A70DA70DA70Dull;  // "not written yet" (a NaN no arithmetic produces)

struct BatchItem { Rk2dArgs a; Chain c; DevConst K; int* err; };   // err: the item's own handle's failure word

// FAST: the packet of the tolerance form (below, at k_rk2d_chain)
template <bool FAST>
__global__ __launch_bounds__(256) void k_rk2d_prepare(Rk2dArgs a, Chain c, DevConst K, int groups,
                                                      const BatchItem* __restrict__ items) {
    if (items) { a = items[blockIdx.y].a; c = items[blockIdx.y].c; K = items[blockIdx.y].K; }
    const int jm = K.jm, km = K.km;
    const int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= jm * km) return;
    const int j = id / km, k = id - j * km;
    // the row above every group: static for group 0, "not written yet" for the others
    if (j < groups) {
        c.bnd[((size_t)j * 2 + 0) * km + k] = j == 0 ? a.v[k] : __longlong_as_double((long long)BND_EMPTY);
        c.bnd[((size_t)j * 2 + 1) * km + k] = j == 0 ? a.w[k] : __longlong_as_double((long long)BND_EMPTY);
    }
    if (j < 1 || j > jm - 2 || k < 1 || k > km - 2) return;
    const size_t q = id;
    const long R = km;
    const bool win_j = (j >= 2) && (j < jm - 3), win_k = (k >= 2) && (k < km - 3);
    unsigned m = land_mask(a.fl[q], a.hyd, a.fl[q + R], a.fl[q - R], a.fl[q + 1], a.fl[q - 1]);
    double x_pp2[2] = { 0., 0. };
    if (win_k) {
        if (!(a.fl[q + 2] & FL_LAND)) m &= ~M_PP2;
        if (!(a.fl[q - 2] & FL_LAND)) m &= ~M_PM2;
        x_pp2[0] = a.v[q + 2]; x_pp2[1] = a.w[q + 2];
    }
    const RowGeo G = row_geometry(a, j);
    const PointCoef C = point_coef(m, a.p[q], a.p[q + R], a.p[q - R], a.p[q + 1], a.p[q - 1], G, win_j, win_k, a, K);
    const int g = (j - 1) / CH_ROWS, r = (j - 1) - g * CH_ROWS, n = k - 1 + r;
    double* out = (double*)(c.pk + ((size_t)g * c.ndd + n) * PK * 64);
    const bool one_sided = C.o_fwd || C.o_bwd;
    const long long sel = (C.o_bwd ? 1ll : 0ll) | (!one_sided ? (1ll << 32) : 0ll);
#pragma unroll
    for (int x = 0; x < 2; x++) {
        const double* f = x ? a.w : a.v;
        const double* fn = x ? a.wn : a.vn;
        const double x_tp = f[q + R], x_pp = f[q + 1];
        double e[2 * PK];
        e[0] = f[q]; e[1] = fn[q];
        if constexpr (FAST) {
            // dx/dthe / rm = A_th x + B_th0 + B_th1 x(j-1,k);  dx/dphi / (rm sin) = A_ph x + B_ph0 + B_ph1 x(j,k-1)
            const double tf = C.T1.h * C.T1.idiv * a.i_rm, pf = C.P1.h * C.P1.idiv * G.i_rmsin;
            const double A_th = tf * C.T1.cc, A_ph = pf * C.P1.cc;
            e[2] = A_th; e[3] = tf * (C.T1.c1 * x_tp); e[4] = tf * C.T1.c2;
            e[5] = A_ph; e[6] = pf * (C.P1.c1 * x_pp); e[7] = pf * C.P1.c2;
            // what multiplies the lane's own stage value outside the advection products, and the partner's
            const double sgn = x ? 1. : -1.;
            const double t2f = C.T2.h * C.T2.idiv * a.i_rm2;
            const double gth = G.costhe * G.i_rmsin;                    // dx/dthe cos / (rm2 sin) = (dx/dthe / rm) gth
            const double coefh = G.coef * (x ? C.h_d_k : C.h_d_j), h0 = x ? C.h_0_k : C.h_0_j;
            e[8] = sgn * K.i_re * (t2f * C.T2.cc + gth * A_th - coefh) - h0 * (x ? K.i_dphi2 : K.i_dthe2);
            e[9] = K.i_re * 2. * gth * A_ph;                            // d other/dphi 2 cos / (rm2 sin2) / re
            e[10] = x ? C.pg_w : C.pg_v;
            e[11] = sgn * K.i_re * t2f * C.T2.c1; e[12] = sgn * K.i_re * t2f * C.T2.c2;
            e[13] = C.o_fwd ? x_pp : x_tp; e[14] = x_pp2[x];
            e[15] = __longlong_as_double(sel);
            const int lane = 2 * r + x;
#pragma unroll
            for (int n2 = 0; n2 < PKF; n2++) {
                out[((size_t)n2 * 64 + lane) * 2] = e[2 * n2];
                out[((size_t)n2 * 64 + lane) * 2 + 1] = e[2 * n2 + 1];
            }
            continue;
        }
        e[2] = C.T1.cc; e[3] = C.T1.c1 * x_tp; e[4] = C.T1.c2; e[5] = C.T1.h; e[6] = C.T1.div; e[7] = C.T1.idiv;
        e[8] = C.P1.cc; e[9] = C.P1.c1 * x_pp; e[10] = C.P1.c2; e[11] = C.P1.h; e[12] = C.P1.div; e[13] = C.P1.idiv;
        e[14] = C.T2.cc; e[15] = C.T2.c1; e[16] = C.T2.c2; e[17] = C.T2.h; e[18] = C.T2.div; e[19] = C.T2.idiv;
        e[20] = C.o_fwd ? x_pp : x_tp; e[21] = x_pp2[x];
        e[22] = x ? C.pg_w : C.pg_v;
        e[23] = G.coef * (x ? C.h_d_k : C.h_d_j);
        e[24] = x ? C.h_0_k : C.h_0_j;
        e[25] = __longlong_as_double(sel);
        const int lane = 2 * r + x;
#pragma unroll
        for (int n2 = 0; n2 < PK; n2++) {
            out[((size_t)n2 * 64 + lane) * 2] = e[2 * n2];
            out[((size_t)n2 * 64 + lane) * 2 + 1] = e[2 * n2 + 1];
        }
    }
}

// the value of the other lane of the pair (lanes 2 r and 2 r + 1 swap)
__device__ __forceinline__ double pair_swap(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_mov_dpp(lo, 0xB1, 0xF, 0xF, true);  // quad_perm [1, 0, 3, 2]
    hi = __builtin_amdgcn_mov_dpp(hi, 0xB1, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double flip_sign(double x, int mask) {  // mask: 0 or the sign bit
    return __hiloint2double(__double2hiint(x) ^ mask, __double2loint(x));
}
// pointers known to be global memory (the items of a batch are read from memory, which hides that from the compiler)
typedef __attribute__((address_space(1))) double* gdouble_t;
typedef const __attribute__((address_space(1))) double* gcdouble_t;
typedef double pair_t __attribute__((ext_vector_type(2)));   // one 16-byte entry of a packet
typedef const __attribute__((address_space(1))) pair_t* gcpair_t;
__device__ __forceinline__ double ld_agent(gcdouble_t p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(gdouble_t p, double x) {
    __hip_atomic_store(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct Packet { pair_t e[PK]; };

// One wave = rows j0 .. j0 + 31, lane 2 r + x.  Step n: row r is at k = n + 1 - r.  Of the values this sweep
// produces, x(j, k-1) and x(j, k-2) are the lane's own last two results, x(j-1, k) is the result the lane two
// below produced in the step before (the first row takes it from the wave of the rows above, through `bnd`, device
// scope); the partner lane's centre value and phi derivative of every stage come by lane permutation.  Everything
// else is in the packet, loaded one step ahead.  Both right-hand sides (RHS_Atm.cpp:664-677) in one form:
//   k_x = -(v dx/dthe / rm + w dx/dphi / (rm sin)) - pg_x + s ( (d2x / rm2 + dx/dthe / (rm2 sin) cos - coef h x)
//          + s (d other/dphi 2 cos / (rm2 sin2)) ) / re - h0 x / d2,       s = -1 for v, +1 for w
// where adding a negated value is the subtraction the reference writes, bit for bit.
// FAST, the tolerance form (about 1e-15 from the exact one per sweep): with the neighbours fixed during the four
// stages, both right-hand sides are
//   k_x = -v (A_th x + B_th) - w (A_ph x + B_ph) + C x + F x' + D        (x' the partner's stage value)
// where A_th, A_ph, C, F come ready in the packet and B_th, B_ph, D take this sweep's neighbour values once per
// step: a stage is six fused multiply-adds and one lane swap instead of ~65 instructions.
template <bool FAST>
__global__ __launch_bounds__(64) void k_rk2d_chain(Rk2dArgs a, Chain c, DevConst K, const BatchItem* __restrict__ items,
                                                   int* __restrict__ err) {
    if (items) { a = items[blockIdx.y].a; c = items[blockIdx.y].c; K = items[blockIdx.y].K; err = items[blockIdx.y].err; }
    const int jm = K.jm, km = K.km;
    const int g = blockIdx.x, lane = threadIdx.x, r = lane >> 1, x = lane & 1;
    const int j0 = 1 + g * CH_ROWS, j = j0 + r;
    const bool row = j <= jm - 2;
    const int rlast = min(CH_ROWS, jm - 1 - j0) - 1;
    const bool feeds = r == rlast && g + 1 < (int)gridDim.x, polls = r == 0;
    const RowGeo G = row_geometry(a, row ? j : 1);
    const double rm = a.rm, rm2 = rm * rm, dt = K.dt;
    const double hdt = .5 * K.dt, dt6 = K.dt * K.i_six;
    // tolerance form: what the theta difference and the partner's phi difference are multiplied by in the viscous bracket
    const double e_th = (x ? 1. : -1.) * K.i_re * G.costhe * G.i_rmsin, g_ph = K.i_re * 2. * G.costhe * G.i_rmsin;
    const double d2 = x ? K.dphi2 : K.dthe2, i_d2 = x ? K.i_dphi2 : K.i_dthe2;
    const int smask = x ? 0 : (int)0x80000000;
    const gcpair_t pk = (gcpair_t)(c.pk + (size_t)g * c.ndd * PK * 64 + lane);
    const gcdouble_t bin = (gcdouble_t)(c.bnd + ((size_t)g * 2 + x) * km);
    const gdouble_t bout = (gdouble_t)(c.bnd + ((size_t)(g + 1) * 2 + x) * km);
    const gdouble_t out = (gdouble_t)((x ? a.w : a.v) + (size_t)(row ? j : 1) * km);
    const gdouble_t out_aux = (gdouble_t)((x ? a.aux_w : a.aux_v) + (size_t)(row ? j : 1) * km);
    const double xb = out[0], aux_b = out_aux[0];   // (j, 0): what the row starts from (and keeps)
    double x_m1 = xb, x_m2 = 0.;         // own results at k-1, k-2 (the first step of a row finds (j, 0) there)
    bool dead = false;
    const gdouble_t idle = (gdouble_t)(c.bnd + (size_t)gridDim.x * 2 * km + blockIdx.x * 64 + lane);

    auto load = [&](int n) {
        Packet P;
        const gcpair_t p = pk + (size_t)n * PK * 64;
#pragma unroll
        for (int e = 0; e < (FAST ? PKF : PK); e++) P.e[e] = p[e * 64];
        return P;
    };
    const int nlast = km - 3 + rlast;
    double b_nx = polls ? ld_agent(bin + 1) : 0.;
    // tolerance form: a step is shorter than a trip to memory, so the row above is asked for four to seven steps ahead,
    // with the packets (a value that was not there yet when asked for is asked for again when its step comes)
    auto load_above = [&](int n) { return ld_agent(bin + min(max(n + 1 - r, 0), km - 1)); };
    auto step = [&](const Packet& cur, int n, double b_ahead = 0.) {
        const int k = n + 1 - r;
        const bool active = row && k >= 1 && k <= km - 2;
        double x_tm = __shfl_up(x_m1, 2);
        {
            double b = FAST ? b_ahead : b_nx;
            // the row above belongs to another wave: its value was asked for a step ago and is nearly always there
            const bool missing = polls && active && !dead && __double_as_longlong(b) == (long long)BND_EMPTY;
            if (__builtin_expect(__any(missing), 0)) {
                if (missing) {
                    for (int tries = 0; __double_as_longlong(b) == (long long)BND_EMPTY; tries++) {
                        if (tries > (1 << 20)) {  // never delivered: say so, and stop waiting for the rest of the sweep
                            __hip_atomic_store(err, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                            dead = true;
                            break;
                        }
                        __builtin_amdgcn_s_sleep(1);
                        b = ld_agent(bin + k);
                    }
                }
            }
            if (polls) x_tm = b;
        }
        if constexpr (!FAST) b_nx = ld_agent(bin + min(max(k + 1, 0), km - 1));  // every lane, so that nothing has to wait for it here
        if constexpr (FAST) {
            const double xn = cur.e[0].y;
            const double A_th = cur.e[1].x, B_th0 = cur.e[1].y, B_th1 = cur.e[2].x, A_ph = cur.e[2].y, B_ph0 = cur.e[3].x, B_ph1 = cur.e[3].y;
            const double Cx = cur.e[4].x, F = cur.e[4].y, pg = cur.e[5].x, S_A = cur.e[5].y, S_B = cur.e[6].x;
            const double sA = cur.e[6].y, sB = cur.e[7].x;
            const bool own = __double2loint(cur.e[7].y) != 0, above = __double2hiint(cur.e[7].y) != 0;
            const double x2A = own ? x_m1 : sA;
            const double x2B = own ? x_m2 : (above ? x_tm : sB);
            const double B_th = __builtin_fma(B_th1, x_tm, B_th0), B_ph = __builtin_fma(B_ph1, x_m1, B_ph0);
            const double oB_ph = pair_swap(B_ph);
            double D = __builtin_fma(e_th, B_th, -pg);
            D = __builtin_fma(g_ph, oB_ph, D);
            D = __builtin_fma(S_A, x2A, D);
            D = __builtin_fma(S_B, x2B, D);
            // lane v: v is the lane's own value, w the partner's; lane w: the other way round
            const double A_own = x ? A_ph : A_th, B_own = x ? B_ph : B_th, A_oth = x ? A_th : A_ph, B_oth = x ? B_th : B_ph;
            double xc = cur.e[0].x, acc = 0., kx = 0.;
#pragma unroll
            for (int s = 0; s < 4; s++) {
                const double oc = pair_swap(xc);
                kx = __builtin_fma(Cx, xc, D);
                kx = __builtin_fma(F, oc, kx);
                kx = __builtin_fma(-xc, __builtin_fma(A_own, xc, B_own), kx);
                kx = __builtin_fma(-oc, __builtin_fma(A_oth, xc, B_oth), kx);
                if (s == 0) { acc = kx; xc = __builtin_fma(kx, hdt, xn); }
                else if (s == 1) { acc = __builtin_fma(2., kx, acc); xc = __builtin_fma(kx, hdt, xn); }
                else if (s == 2) { acc = __builtin_fma(2., kx, acc); xc = __builtin_fma(kx, dt, xn); }
            }
            const double x_new = __builtin_fma(acc + kx, dt6, xn);
            out[active ? k : 0] = active ? x_new : xb;
            out_aux[active ? k : 0] = active ? kx + pg : aux_b;
            st_agent(active && feeds ? bout + k : idle, x_new);
            if (active) { x_m2 = x_m1; x_m1 = x_new; }
            return;
        }
        const double xn = cur.e[0].y;
        const double t1_cc = cur.e[1].x, t1_a = cur.e[1].y, t1_c2 = cur.e[2].x, t1_h = cur.e[2].y, t1_div = cur.e[3].x, t1_idiv = cur.e[3].y;
        const double p1_cc = cur.e[4].x, p1_a = cur.e[4].y, p1_c2 = cur.e[5].x, p1_h = cur.e[5].y, p1_div = cur.e[6].x, p1_idiv = cur.e[6].y;
        const double t2_cc = cur.e[7].x, t2_c1 = cur.e[7].y, t2_c2 = cur.e[8].x, t2_h = cur.e[8].y, t2_div = cur.e[9].x, t2_idiv = cur.e[9].y;
        const double sA = cur.e[10].x, sB = cur.e[10].y, pg = cur.e[11].x, coefh = cur.e[11].y, h0 = cur.e[12].x;
        const bool own = __double2loint(cur.e[12].y) != 0, above = __double2hiint(cur.e[12].y) != 0;
        const double x2A = own ? x_m1 : sA;
        const double x2B = own ? x_m2 : (above ? x_tm : sB);
        const double t1_b = t1_c2 * x_tm, p1_b = p1_c2 * x_m1, t2_a = t2_c1 * x2A, t2_b = t2_c2 * x2B;
        double xc = cur.e[0].x;
        double kx[4];
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const double dxdthe = div_by(t1_h * ((t1_cc * xc + t1_a) + t1_b), t1_div, t1_idiv);
            const double dxdphi = div_by(p1_h * ((p1_cc * xc + p1_a) + p1_b), p1_div, p1_idiv);
            const double d2x = div_by(t2_h * ((t2_cc * xc + t2_a) + t2_b), t2_div, t2_idiv);
            const double oc = pair_swap(xc), odphi = pair_swap(dxdphi);
            const double vc = x ? oc : xc, wc = x ? xc : oc;
            const double adv = div_by(vc * dxdthe, rm, a.i_rm) + div_by(wc * dxdphi, G.rmsin, G.i_rmsin);
            const double inner = div_by(d2x, rm2, a.i_rm2) + div_by(dxdthe, G.rm2sin, G.i_rm2sin) * G.costhe - coefh * xc;
            const double cross = flip_sign(div_by(odphi * 2. * G.costhe, G.rm2sin2, G.i_rm2sin2), smask);
            const double visc = flip_sign(div_by(inner + cross, K.re, K.i_re), smask);
            kx[s] = -adv + -pg + visc - div_by(h0 * xc, d2, i_d2);
            if (s < 2) xc = xn + kx[s] * .5 * dt;
            else if (s == 2) xc = xn + kx[s] * dt;
        }
        const double x_new = xn + div_by(dt * (kx[0] + 2. * kx[1] + 2. * kx[2] + kx[3]), 6., K.i_six);
        // every lane stores in every step (an idle lane into a slot of its own that nobody reads): with a fixed
        // number of stores per step the waits for the loads can be counted so that they never wait for a store
        out[active ? k : 0] = active ? x_new : xb;
        out_aux[active ? k : 0] = active ? kx[3] + pg : aux_b;
        st_agent(active && feeds ? bout + k : idle, x_new);
        if (active) { x_m2 = x_m1; x_m1 = x_new; }
    };
    // two steps per trip, each with packet registers of its own: the loads of a step are issued a whole step
    // before their first use and nothing waits for the stores (a step beyond the last one has no active lane)
    if constexpr (FAST) {
        // Eight steps per trip, in two halves with packet registers of their own: what a half needs is asked for behind
        // the first step of the half before, three to six steps ahead.  (The compiler waits for everything outstanding
        // at the first use of a value loaded before the loop's back edge: loads issued just before that use, or a step
        // before the end of a trip, would cost a whole trip to memory per trip.)
        Packet X0 = load(0), X1 = load(1), X2 = load(2), X3 = load(3);
        double bx0 = load_above(0), bx1 = load_above(1), bx2 = load_above(2), bx3 = load_above(3);
        // (Holding the results of a half back and storing them behind the next half's wait, so that it is never a wait for
        // a store's acknowledgement, was slower: the row handed to the next wave left four steps later.)
        for (int n = 0; n <= nlast; n += 8) {
            // (every value that was asked for before the back edge is declared used here, where nothing newer is under way)
            asm volatile("" :: "v"(bx0), "v"(bx1), "v"(bx2), "v"(bx3), "v"(X0.e[PKF - 1].y), "v"(X1.e[PKF - 1].y), "v"(X2.e[PKF - 1].y), "v"(X3.e[PKF - 1].y));
            step(X0, n, bx0);
            const Packet Y0 = load(n + 4), Y1 = load(n + 5), Y2 = load(n + 6), Y3 = load(n + 7);
            const double by0 = load_above(n + 4), by1 = load_above(n + 5), by2 = load_above(n + 6), by3 = load_above(n + 7);
            step(X1, n + 1, bx1); step(X2, n + 2, bx2); step(X3, n + 3, bx3);
            step(Y0, n + 4, by0);
            X0 = load(n + 8); X1 = load(n + 9); X2 = load(n + 10); X3 = load(n + 11);
            bx0 = load_above(n + 8); bx1 = load_above(n + 9); bx2 = load_above(n + 10); bx3 = load_above(n + 11);
            step(Y1, n + 5, by1); step(Y2, n + 6, by2); step(Y3, n + 7, by3);
        }
        return;
    }
    Packet A = load(0);
    for (int n = 0; n <= nlast; n += 2) {
        const Packet B = load(n + 1);
        step(A, n);
        A = load(n + 2);
        step(B, n + 1);
    }
}

// arguments of one model's surface sweep; allocates the packets on first use
int rk2d_item(atomdev* h, BatchItem& it) {
    if (int rc = atomdev_refresh_flags(h)) return rc;
    const bool hyd = h->model == ATOMDEV_HYD;
    const int L = hyd ? h->im - 1 : 0;
    const size_t off = (size_t)L * h->n2;
    Rk2dArgs& a = it.a;
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
    it.K = h->k;
    const int groups = (h->jm - 2 + CH_ROWS - 1) / CH_ROWS;
    const int ndd = h->km + CH_ROWS + 12;   // steps 0 .. km - 3 + 31, and up to eleven more that are loaded ahead
    const size_t pk_bytes = (size_t)groups * ndd * PK * 64 * sizeof(double2);
    const size_t bnd_bytes = ((size_t)groups * 2 * h->km + (size_t)groups * 64) * sizeof(double);  // and a slot per lane
    if (!h->surf) {
        ATOMDEV_TRY(hipMalloc(&h->surf, pk_bytes + bnd_bytes));
        ATOMDEV_TRY(hipMemsetAsync(h->surf, 0, pk_bytes + bnd_bytes, h->stream));
    }
    it.c.pk = (double2*)h->surf;
    it.c.bnd = (double*)((char*)h->surf + pk_bytes);
    it.c.ndd = ndd;
    return 0;
}

// prepare + chain for n models (items == nullptr: one model, arguments of `one` by value) on `stream`
// fast: the tolerance form (the handles' rk_variant 2 and 3)
int rk2d_launch(const BatchItem& one, const BatchItem* items, int n, int jm, int km, hipStream_t stream, int* err, bool fast) {
    const int groups = (jm - 2 + CH_ROWS - 1) / CH_ROWS;
    const unsigned nb = (unsigned)(((size_t)jm * km + 255) / 256);
    if (fast) hipLaunchKernelGGL(k_rk2d_prepare<true>, dim3(nb, n), dim3(256), 0, stream, one.a, one.c, one.K, groups, items);
    else hipLaunchKernelGGL(k_rk2d_prepare<false>, dim3(nb, n), dim3(256), 0, stream, one.a, one.c, one.K, groups, items);
    ATOMDEV_TRY(hipGetLastError());
    if (fast) hipLaunchKernelGGL(k_rk2d_chain<true>, dim3(groups, n), dim3(64), 0, stream, one.a, one.c, one.K, items, err);
    else hipLaunchKernelGGL(k_rk2d_chain<false>, dim3(groups, n), dim3(64), 0, stream, one.a, one.c, one.K, items, err);
    return atomdev_check(hipGetLastError(), "k_rk2d_chain");
}

int rk2d(atomdev* h, bool hyd) {
    (void)hyd;
    BatchItem it;
    if (int rc = rk2d_item(h, it)) return rc;
    if (h->rk_variant == 1 || h->jm < 3 || h->km < 4) {
        hipLaunchKernelGGL(k_rk2d, dim3(1), dim3(256), (size_t)h->jm * sizeof(RowGeo), h->stream, it.a, h->k);
        return atomdev_check(hipGetLastError(), "k_rk2d");
    }
    return rk2d_launch(it, nullptr, 1, h->jm, h->km, h->stream, (int*)atomdev_kernel_error_word(h), h->rk_variant >= 2);
}

}  // namespace

// The surface sweep is a chain of dependent anti-diagonals that occupies a handful of waves, so one model leaves most
// of the 256 compute units idle: the sweeps of independent models (the time slices of a paleo series, each with its
// own handle) go into one launch.  Work already enqueued on the handles' streams is waited
// for through events, and each stream in turn waits for the batch, so the ordering rules of single calls hold.
extern "C" int atomdev_rk2d_batch(atomdev_t* const* handles, int n) {
    ATOMDEV_REQUIRE(handles && n >= 1, "atomdev_rk2d_batch: bad argument");
    atomdev* h0 = handles[0];
    ATOMDEV_REQUIRE(h0 != nullptr, "atomdev_rk2d_batch: null handle");
    ATOMDEV_REQUIRE(h0->jm >= 3 && h0->km >= 4, "atomdev_rk2d_batch: surface too small");
    for (int m = 0; m < n; m++) {
        atomdev* h = handles[m];
        ATOMDEV_REQUIRE(h != nullptr, "atomdev_rk2d_batch: null handle");
        ATOMDEV_REQUIRE(h->device == h0->device, "atomdev_rk2d_batch: handles on different devices");
        ATOMDEV_REQUIRE(h->jm == h0->jm && h->km == h0->km, "atomdev_rk2d_batch: surfaces of different size");
        ATOMDEV_REQUIRE(h->model != ATOMDEV_HYD || h->im == 41,
                        "atomdev_rk2d_batch: the reference hard-codes im = 41 for the hydrosphere surface sweep");
        ATOMDEV_REQUIRE((h->rk_variant >= 2) == (h0->rk_variant >= 2), "atomdev_rk2d_batch: handles with the exact and the tolerance form of the sweep in one batch");
        for (int q = 0; q < m; q++) ATOMDEV_REQUIRE(handles[q] != h, "atomdev_rk2d_batch: a handle appears twice");
    }
    ATOMDEV_TRY(hipSetDevice(h0->device));
    std::vector<BatchItem> items(n);
    for (int m = 0; m < n; m++) {
        if (int rc = rk2d_item(handles[m], items[m])) return rc;
        items[m].err = (int*)atomdev_kernel_error_word(handles[m]);   // a give-up is reported on the model it happened in
    }
    const size_t bytes = (size_t)n * sizeof(BatchItem);
    if (h0->batch_bytes < bytes) {
        ATOMDEV_TRY(hipStreamSynchronize(h0->stream));
        if (h0->batch) ATOMDEV_TRY(hipFree(h0->batch));
        h0->batch = nullptr; h0->batch_bytes = 0;
        ATOMDEV_TRY(hipMalloc(&h0->batch, bytes));
        h0->batch_bytes = bytes;
    }
    // the first handle's stream carries the batch
    for (int m = 1; m < n; m++) {
        ATOMDEV_TRY(hipEventRecord(handles[m]->ev1, handles[m]->stream));
        ATOMDEV_TRY(hipStreamWaitEvent(h0->stream, handles[m]->ev1, 0));
    }
    ATOMDEV_TRY(hipMemcpyAsync(h0->batch, items.data(), bytes, hipMemcpyHostToDevice, h0->stream));
    ATOMDEV_TRY(hipStreamSynchronize(h0->stream));  // `items` is host memory that goes out of scope
    if (int rc = rk2d_launch(items[0], (const BatchItem*)h0->batch, n, h0->jm, h0->km, h0->stream,
                             (int*)atomdev_kernel_error_word(h0), h0->rk_variant >= 2)) return rc;
    ATOMDEV_TRY(hipEventRecord(h0->ev1, h0->stream));
    for (int m = 1; m < n; m++) ATOMDEV_TRY(hipStreamWaitEvent(handles[m]->stream, h0->ev1, 0));
    return 0;
}

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
