// Surface (2-D) Runge-Kutta sweeps: solveRungeKutta_2D_Atmosphere + RK_RHS_2D_Atmosphere
// (atmosphere/RungeKutta_Atm.cpp:163-217, RHS_Atm.cpp:498-678) and the hydrosphere twins
// (hydrosphere/RungeKutta_Hyd.cpp:130-186, RHS_Hyd.cpp:453-653).
//
// Unlike the 3-D sweep, theta AND phi differences are live here, so the in-place
// lexicographic (j, k) sweep is a true 2-D Gauss-Seidel: point (j, k) reads the
// updated values at (j-1, k), (j-2, k), (j, k-1), (j, k-2) and the old values at
// (j+1, k), (j+2, k), (j, k+1), (j, k+2).  All points of an anti-diagonal
// j + k = d are mutually independent and see exactly those values if diagonals are
// processed in increasing d.  One workgroup walks the diagonals with a barrier
// between them (65 k points on the 1-degree grid: latency-bound, not worth more
// than one CU).
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

// All four RK stages at one surface point (RungeKutta_Atm.cpp:176-214 around RHS_Atm.cpp:498-678).
__device__ __forceinline__ PointOut rk_point(const PointIn& in, const RowGeo& G, bool win_j, bool win_k,
                                             const Rk2dArgs& a, const DevConst& K) {
    const double dt = K.dt;
    const double rm = a.rm, rm2 = rm * rm;
    const bool L = in.m & M_L, fluid_c = in.m & M_FLUID_C;
    const bool l_tp = in.m & M_TP, l_tm = in.m & M_TM, l_pp = in.m & M_PP, l_pm = in.m & M_PM, l_pp2 = in.m & M_PP2, l_pm2 = in.m & M_PM2;
    double h_0_j, h_d_j, h_0_k, h_d_k;
    if (fluid_c && (l_tp || l_tm)) { h_0_j = 1.; h_d_j = 0.; } else { h_0_j = 0.; h_d_j = 1.; }
    if (fluid_c && (l_pp || l_pm)) { h_0_k = 1.; h_d_k = 0.; } else { h_0_k = 0.; h_d_k = 1.; }

    // ---- the surviving variant of every difference ----
    // central {0, 1, -1} / 2 d; forward two-point {-1, 1, 0} / d; backward two-point {1, 0, -1} / d (wins)
    const bool t_fwd = win_j && L && !l_tp, t_bwd = win_j && L && !l_tm;
    const bool p_fwd = L && !l_pp, p_bwd = win_k ? (L && !l_pm) : (fluid_c && l_pm);
    const int t_sel = t_bwd ? 2 : (t_fwd ? 1 : 0), p_sel = p_bwd ? 2 : (p_fwd ? 1 : 0);
    Diff T1, P1, T2;
    T1.cc = t_sel == 2 ? 1. : (t_sel == 1 ? -1. : 0.); T1.c1 = t_sel == 2 ? 0. : 1.; T1.c2 = t_sel == 1 ? 0. : -1.;
    T1.h = h_d_j; T1.div = t_sel ? K.dthe : K.two_dthe; T1.idiv = t_sel ? K.i_dthe : K.i_two_dthe;
    P1.cc = p_sel == 2 ? 1. : (p_sel == 1 ? -1. : 0.); P1.c1 = p_sel == 2 ? 0. : 1.; P1.c2 = p_sel == 1 ? 0. : -1.;
    P1.h = h_d_k; P1.div = p_sel ? K.dphi : K.two_dphi; P1.idiv = p_sel ? a.i_dphi : a.i_two_dphi;
    // theta second derivative and its two neighbour operands: zero {0, 0, 0} inside window_j, central {-2, 1, 1}
    // outside it, zero again outside window_k; inside window_k the PHI one-sided form {2, -2, 1} on (k, k+-1, k+-2)
    // where the centre is land and both cells on that side are fluid (the s = -1 side wins)
    const bool o_fwd = win_k && L && !l_pp && !l_pp2, o_bwd = win_k && L && !l_pm && !l_pm2;
    const bool one_sided = o_fwd || o_bwd, central2 = !win_j && win_k && !one_sided;
    T2.cc = one_sided ? 2. : (central2 ? -2. : 0.); T2.c1 = one_sided ? -2. : (central2 ? 1. : 0.);
    T2.c2 = (one_sided || central2) ? 1. : 0.;
    T2.h = one_sided ? h_d_k : h_d_j; T2.div = one_sided ? K.dphi2 : K.dthe2; T2.idiv = one_sided ? K.i_dphi2 : K.i_dthe2;
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
    // pressure gradient terms do not change during the stages
    const double dpdthe = diff(T1, in.p_c, in.p_tp, in.p_tm), dpdphi = diff(P1, in.p_c, in.p_pp, in.p_pm);
    const double pg_v = div_by(div_by(h_d_j * dpdthe, rm, a.i_rm), a.rho, a.i_rho);         // h_d_j * dpdthe / rm / rho
    const double pg_w = div_by(div_by(h_d_k * dpdphi, G.rmsin, G.i_rmsin), a.rho, a.i_rho);

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

// Fallback for grids with more interior rows than one workgroup has threads: the fields stay in their
// row-major layout, a thread takes whatever rows of the diagonal fall to it, and every neighbour goes
// through global memory (the barrier orders the writes of diagonal d before the reads of d + 1).
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

// ---- the same sweep with one thread per row ----
// Surface fields in a skewed copy: point (j, k) lives at [j + k][j] (pitch P), so the points of one anti-
// diagonal, and each of their neighbour sets, are contiguous across the rows = across the lanes of a wave.
struct Skewed {
    double *v, *w, *p, *vn, *wn, *aux_v, *aux_w;
    uint8_t* fl;
    int P;
};

// One surface of a batched launch (several independent models in one grid: blockIdx.y of the copy kernels,
// blockIdx.x of the sweep selects the model).  A launch for a single model passes its arguments by value.
struct BatchItem { Rk2dArgs a; Skewed s; DevConst K; };

__global__ void k_skew(Rk2dArgs a, Skewed s, int jm, int km, const BatchItem* __restrict__ items) {
    if (items) { a = items[blockIdx.y].a; s = items[blockIdx.y].s; }
    const int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= jm * km) return;
    const int j = id / km, k = id - j * km;
    const size_t o = (size_t)(j + k) * s.P + j;
    s.v[o] = a.v[id]; s.w[o] = a.w[id]; s.p[o] = a.p[id]; s.vn[o] = a.vn[id]; s.wn[o] = a.wn[id]; s.fl[o] = a.fl[id];
}

__global__ void k_unskew(Rk2dArgs a, Skewed s, int jm, int km, const BatchItem* __restrict__ items) {
    if (items) { a = items[blockIdx.y].a; s = items[blockIdx.y].s; }
    const int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= jm * km) return;
    const int j = id / km, k = id - j * km;
    if (j < 1 || j > jm - 2 || k < 1 || k > km - 2) return;  // the sweep writes interior points only
    const size_t o = (size_t)(j + k) * s.P + j;
    a.v[id] = s.v[o]; a.w[id] = s.w[o]; a.aux_v[id] = s.aux_v[o]; a.aux_w[id] = s.aux_w[o];
}

// What a thread reads from memory for its point on a diagonal; fetched one diagonal ahead.
struct Fetched {
    PointIn in;          // without v_tm, w_tm, v_pm, w_pm, v_pm2, w_pm2 (results of this sweep)
    double v_b, w_b;     // (j, 0): the phi-boundary values, wanted when k == 1
    double v_t0, w_t0;   // (0, k): the theta-boundary row, wanted by row 1
    bool active;
};

// Thread t owns row j = t + 1 and walks it with the diagonals (k = d - j).  Of the values the sweep itself
// produces, (j, k-1) and (j, k-2) are the thread's own last two results (registers) and (j-1, k) is the
// neighbour thread's result of the previous diagonal (LDS, double-buffered); everything else is old or
// static data, fetched from the skewed copy one diagonal ahead.  So no global memory access sits on the
// chain of 538 dependent diagonals (1-degree grid), only an LDS exchange and one barrier per diagonal.
template <int MAX_THREADS>
__global__ __launch_bounds__(MAX_THREADS) void k_rk2d_rows(Rk2dArgs a, Skewed s, DevConst K,
                                                           const BatchItem* __restrict__ items) {
    if (items) { a = items[blockIdx.x].a; s = items[blockIdx.x].s; K = items[blockIdx.x].K; }
    extern __shared__ double ex[];  // [2][P][2]: new (v, w) of row j on the last two diagonals
    const int jm = K.jm, km = K.km, P = s.P;
    const int j = (int)threadIdx.x + 1;
    const bool row = j <= jm - 2;
    const bool win_j = (j >= 2) && (j < jm - 3);
    RowGeo G = row_geometry(a, row ? j : 1);
    const int dmax = (jm - 2) + (km - 2);

    auto fetch = [&](int d) {
        Fetched f;
        const int k = d - j;
        f.active = row && k >= 1 && k <= km - 2;
        f.v_b = f.w_b = f.v_t0 = f.w_t0 = 0.;
        PointIn& in = f.in;
        in.vc = in.wc = in.vn = in.wn = in.v_tp = in.v_tm = in.v_pp = in.v_pm = in.w_tp = in.w_tm = in.w_pp = in.w_pm = 0.;
        in.v_pp2 = in.v_pm2 = in.w_pp2 = in.w_pm2 = in.p_c = in.p_tp = in.p_tm = in.p_pp = in.p_pm = 0.;
        in.m = 0;
        if (f.active) {
            const size_t o = (size_t)d * P + j;           // (j, k)
            const size_t o_tp = o + P + 1, o_tm = o - P - 1, o_pp = o + P, o_pm = o - P;
            in.vc = s.v[o]; in.wc = s.w[o]; in.vn = s.vn[o]; in.wn = s.wn[o];
            in.v_tp = s.v[o_tp]; in.w_tp = s.w[o_tp]; in.v_pp = s.v[o_pp]; in.w_pp = s.w[o_pp];
            in.p_c = s.p[o]; in.p_tp = s.p[o_tp]; in.p_tm = s.p[o_tm]; in.p_pp = s.p[o_pp]; in.p_pm = s.p[o_pm];
            in.m = land_mask(s.fl[o], a.hyd, s.fl[o_tp], s.fl[o_tm], s.fl[o_pp], s.fl[o_pm]);
            if ((k >= 2) && (k < km - 3)) {
                if (!(s.fl[o_pp + P] & FL_LAND)) in.m &= ~M_PP2;
                if (!(s.fl[o_pm - P] & FL_LAND)) in.m &= ~M_PM2;
                in.v_pp2 = s.v[o_pp + P]; in.w_pp2 = s.w[o_pp + P];
            }
            if (k == 1) { f.v_b = s.v[o_pm]; f.w_b = s.w[o_pm]; }
            if (j == 1) { f.v_t0 = s.v[o_tm]; f.w_t0 = s.w[o_tm]; }
        }
        return f;
    };

    double v_m1 = 0., w_m1 = 0., v_m2 = 0., w_m2 = 0.;  // own results at k-1, k-2
    Fetched nx = fetch(2);
    for (int d = 2; d <= dmax; d++) {
        const Fetched cur = nx;
        nx = fetch(d + 1);
        if (cur.active) {
            const int k = d - j;
            if (k == 1) { v_m1 = cur.v_b; w_m1 = cur.w_b; }
            PointIn in = cur.in;
            if (j == 1) { in.v_tm = cur.v_t0; in.w_tm = cur.w_t0; }
            else {
                const double* e = ex + ((size_t)((d - 1) & 1) * P + (j - 1)) * 2;
                in.v_tm = e[0]; in.w_tm = e[1];
            }
            in.v_pm = v_m1; in.w_pm = w_m1; in.v_pm2 = v_m2; in.w_pm2 = w_m2;
            const bool win_k = (k >= 2) && (k < km - 3);
            const PointOut o = rk_point(in, G, win_j, win_k, a, K);
            const size_t q = (size_t)d * P + j;
            s.v[q] = o.v; s.w[q] = o.w; s.aux_v[q] = o.aux_v; s.aux_w[q] = o.aux_w;
            double* e = ex + ((size_t)(d & 1) * P + j) * 2;
            e[0] = o.v; e[1] = o.w;
            v_m2 = v_m1; w_m2 = w_m1; v_m1 = o.v; w_m1 = o.w;
        }
        lds_barrier();
    }
}

// arguments of one model's surface sweep; allocates the skewed copy on first use
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
    // skewed copy of the surface: 7 fp64 arrays and the flags, jm + km + 2 diagonals of pitch P
    const int P = (h->jm + 15) & ~15;
    const size_t ndiag = (size_t)h->jm + h->km + 2, cells = ndiag * P;
    if (!h->surf) {
        ATOMDEV_TRY(hipMalloc(&h->surf, cells * (7 * sizeof(double) + 1)));
        ATOMDEV_TRY(hipMemsetAsync(h->surf, 0, cells * (7 * sizeof(double) + 1), h->stream));
    }
    Skewed& s = it.s;
    double* base = (double*)h->surf;
    s.v = base; s.w = base + cells; s.p = base + 2 * cells; s.vn = base + 3 * cells; s.wn = base + 4 * cells;
    s.aux_v = base + 5 * cells; s.aux_w = base + 6 * cells; s.fl = (uint8_t*)(base + 7 * cells);
    s.P = P;
    return 0;
}

// skew, sweep, unskew for n models (items == nullptr: one model, arguments of `one` by value) on `stream`
int rk2d_launch(const BatchItem& one, const BatchItem* items, int n, int jm, int km, hipStream_t stream) {
    const int rows = jm - 2, threads = (rows + 63) / 64 * 64;
    const int P = one.s.P;
    const unsigned nb = (unsigned)(((size_t)jm * km + 255) / 256);
    hipLaunchKernelGGL(k_skew, dim3(nb, n), dim3(256), 0, stream, one.a, one.s, jm, km, items);
    ATOMDEV_TRY(hipGetLastError());
    const size_t lds = (size_t)2 * P * 2 * sizeof(double);
    if (threads <= 256)
        hipLaunchKernelGGL(k_rk2d_rows<256>, dim3(n), dim3(threads), lds, stream, one.a, one.s, one.K, items);
    else if (threads <= 768)  // three waves per SIMD: the register allocation still fits without spills
        hipLaunchKernelGGL(k_rk2d_rows<768>, dim3(n), dim3(threads), lds, stream, one.a, one.s, one.K, items);
    else
        hipLaunchKernelGGL(k_rk2d_rows<1024>, dim3(n), dim3(threads), lds, stream, one.a, one.s, one.K, items);
    ATOMDEV_TRY(hipGetLastError());
    hipLaunchKernelGGL(k_unskew, dim3(nb, n), dim3(256), 0, stream, one.a, one.s, jm, km, items);
    return atomdev_check(hipGetLastError(), "k_unskew");
}

int rk2d(atomdev* h, bool hyd) {
    (void)hyd;
    BatchItem it;
    const int rows = h->jm - 2, threads = (rows + 63) / 64 * 64;
    if (threads > 1024 || h->rk_variant == 1) {
        if (int rc = rk2d_item(h, it)) return rc;
        hipLaunchKernelGGL(k_rk2d, dim3(1), dim3(256), (size_t)h->jm * sizeof(RowGeo), h->stream, it.a, h->k);
        return atomdev_check(hipGetLastError(), "k_rk2d");
    }
    if (int rc = rk2d_item(h, it)) return rc;
    return rk2d_launch(it, nullptr, 1, h->jm, h->km, h->stream);
}

}  // namespace

// The surface sweep is a chain of dependent anti-diagonals that occupies one workgroup, so one model leaves 255
// of 256 compute units idle: the sweeps of independent models (the time slices of a paleo series, each with its
// own handle) go into one launch, one workgroup each.  Work already enqueued on the handles' streams is waited
// for through events, and each stream in turn waits for the batch, so the ordering rules of single calls hold.
extern "C" int atomdev_rk2d_batch(atomdev_t* const* handles, int n) {
    ATOMDEV_REQUIRE(handles && n >= 1, "atomdev_rk2d_batch: bad argument");
    atomdev* h0 = handles[0];
    ATOMDEV_REQUIRE(h0 != nullptr, "atomdev_rk2d_batch: null handle");
    const int rows = h0->jm - 2, threads = (rows + 63) / 64 * 64;
    ATOMDEV_REQUIRE(threads <= 1024, "atomdev_rk2d_batch: more than 1024 interior rows");
    for (int m = 0; m < n; m++) {
        atomdev* h = handles[m];
        ATOMDEV_REQUIRE(h != nullptr, "atomdev_rk2d_batch: null handle");
        ATOMDEV_REQUIRE(h->device == h0->device, "atomdev_rk2d_batch: handles on different devices");
        ATOMDEV_REQUIRE(h->jm == h0->jm && h->km == h0->km, "atomdev_rk2d_batch: surfaces of different size");
        ATOMDEV_REQUIRE(h->model != ATOMDEV_HYD || h->im == 41,
                        "atomdev_rk2d_batch: the reference hard-codes im = 41 for the hydrosphere surface sweep");
        for (int q = 0; q < m; q++) ATOMDEV_REQUIRE(handles[q] != h, "atomdev_rk2d_batch: a handle appears twice");
    }
    ATOMDEV_TRY(hipSetDevice(h0->device));
    std::vector<BatchItem> items(n);
    for (int m = 0; m < n; m++)
        if (int rc = rk2d_item(handles[m], items[m])) return rc;
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
    if (int rc = rk2d_launch(items[0], (const BatchItem*)h0->batch, n, h0->jm, h0->km, h0->stream)) return rc;
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
