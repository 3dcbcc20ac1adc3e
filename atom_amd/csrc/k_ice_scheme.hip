// BC_Thermo::Two_Category_Ice_Scheme (atmosphere/BC_Thermo.cpp:2803-3143) on the device.
//
// Every term of the scheme is local to a grid point except two: the column integrals of rain and snow (top
// down, inside a column) and the class members S_nuc and S_i_dep, which the reference assigns only under
// conditions (:2923-2931, :2995-3016) - where no condition holds, a grid point works with whatever the point
// visited before it left behind (k outer, j, i from im-2 down; then the next pass; then the next call).  The
// values that ARE assigned depend on local data only (t, c, ice; the scheme changes none of them), so the
// inherited value of a point is simply the last assignment before it in the reference's visiting order, the
// same in each of the five passes.  That order is restored in three steps: (1) every column finds its own last
// assignments; (2) a scan over the columns in the reference's (k, j) order gives each column what it inherits
// from the columns before it, and the last assignment of the whole grid, which is what the first columns
// inherit from the previous pass; (3) every column runs the preparatory and the five full passes on its own.
#include "atomdev_internal.h"
#include "cr_math.h"

namespace {

constexpr int TPB = 128;

struct IceArgs {
    double *c, *cloud, *ice, *p_rain, *p_snow, *s_v, *s_c, *s_i, *s_r, *s_s, *s_c_c;
    const double* t;
    const uint8_t* flags;
    int im, jm, km;
    double t_0, g, gam, R_Air, r_air, R_WaterVapour, L_atm, hp, ep, dr;
    // per column, indexed by its rank k * jm + j in the reference's visiting order
    double *out_nuc, *out_dep, *in_nuc, *in_dep;   // in_*: from the columns before it inside its chunk of CHUNK ranks
    uint8_t *out_def, *in_def;   // bit 0: S_nuc, bit 1: S_i_dep
    double *chunk_nuc, *chunk_dep;                 // per chunk: its own last assignments, then (after the carry kernel)
    uint8_t* chunk_def;                            // what the chunk inherits from the chunks before it
    int nchunks;
    double* last;                // {def S_nuc, S_nuc, def S_i_dep, S_i_dep} of the whole grid
    double state_nuc, state_dep; // the members as the call found them
};

// lib/Utils.h:79-81
__device__ __forceinline__ double exp_func(double T_K, double co_1, double co_2) { return cr_exp(co_1 * (T_K - 273.15) / (T_K - co_2)); }

constexpr double N_i_0 = 1.e2, m_i_0 = 1.e-12, m_i_max = 1.e-9, m_s_0 = 3.e-9;
constexpr double c_i_dep = 1.3e-5, c_c_au = 4.e-4, c_i_au = 1.e-3, c_ac = .24, c_rim = 18.6, c_agg = 10.3, c_i_cri = .24,
                 c_r_cri = 3.2e-5, a_ev = 1.e-3, b_ev = 5.9, c_s_dep = 1.8e-2, b_s_dep = 12.3, c_s_melt = 8.43e-5,
                 b_s_melt = 12.05, a_s_melt = 2.31e3, c_r_frz = 3.75e-2;
constexpr double t_nuc = 267.15, t_d = 248.15, t_hn = 236.15, t_r_frz = 271.15;
constexpr double dt_snow_dim = 417., dt_rain_dim = 250.;  // BC_Thermo.cpp:113-119

// what a grid point's state (t, c, ice; clamped) fixes, whatever the pass
struct Local {
    double t_u, r_humid, q_Rain, q_Ice, N_i, m_i;
    bool cold;              // t_u <= t_0
    bool nuc_def, dep_def;  // the point assigns S_nuc / S_i_dep
    double nuc_val, dep_val;
};

__device__ __forceinline__ Local local_terms(const IceArgs& a, int i, double t, double t_surface, double cc, double ic) {
    Local L;
    const double t_0 = a.t_0;
    const double exp_pressure = a.g / (1.e-2 * a.gam * a.R_Air);
    L.t_u = t * t_0;
    const double p_SL = .01 * (a.r_air * a.R_Air * t_surface * t_0);
    const double hight = (double)i * (a.L_atm / (double)(a.im - 1));
    double p_h;
    if (i != 0) p_h = cr_pow(((L.t_u - a.gam * hight * 1.e-2) / (L.t_u)), exp_pressure) * p_SL;
    else p_h = p_SL;
    const double r_dry = 100. * p_h / (a.R_Air * L.t_u);
    L.r_humid = r_dry * (1. + cc) / (1. + a.R_WaterVapour / a.R_Air * cc);
    const double E_Rain = a.hp * exp_func(L.t_u, 17.2694, 35.86);
    const double E_Ice = a.hp * exp_func(L.t_u, 21.8746, 7.66);
    L.q_Rain = a.ep * E_Rain / (p_h - E_Rain);
    L.q_Ice = a.ep * E_Ice / (p_h - E_Ice);
    L.cold = L.t_u <= t_0;
    L.N_i = 0.; L.m_i = m_i_max;
    if (L.cold) {  // :2985-2992 (where t_u > t_0 neither is read)
        L.N_i = N_i_0 * cr_exp(.2 * (t_0 - L.t_u));
        L.m_i = L.r_humid * ic / L.N_i;
        if (L.m_i > m_i_max) L.m_i = m_i_max;
        if (L.m_i < m_i_0) L.m_i = m_i_0;
    }
    // S_nuc, :2995-3002
    L.nuc_def = false; L.nuc_val = 0.;
    if (ic == 0.) {
        if (((L.t_u < t_d) && (cc >= L.q_Ice)) || (((t_d <= L.t_u) && (L.t_u <= t_nuc)) && (cc >= L.q_Rain))) {
            L.nuc_def = true; L.nuc_val = m_i_0 / (L.r_humid * dt_snow_dim) * L.N_i;
        }
    } else { L.nuc_def = true; L.nuc_val = 0.; }
    // S_i_dep, :3008-3016 (t_Celsius = t_u - t_0 <= 0 exactly where t_u <= t_0)
    L.dep_def = false; L.dep_val = 0.;
    if (L.cold) {
        if (cc > L.q_Ice) { L.dep_def = true; L.dep_val = c_i_dep * L.N_i * cr_pow(L.m_i, (1. / 3.)) * (cc - L.q_Ice); }
        if ((cc < L.q_Ice) && (-ic / dt_snow_dim > (cc - L.q_Ice) / dt_snow_dim)) { L.dep_def = true; L.dep_val = -ic / dt_snow_dim; }
    } else { L.dep_def = true; L.dep_val = 0.; }
    return L;
}

__device__ __forceinline__ double at_least_zero(double x) { return x < 0. ? 0. : x; }

// step 1: clamp the column's c, cloud, ice, P_rain, P_snow (:2861-2871) and record its last assignments
__global__ void k_ice_column_summary(IceArgs a) {
    const size_t n2 = (size_t)a.jm * a.km;
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n2) return;
    const int j = (int)(q / a.km), k = (int)(q - (size_t)j * a.km);
    const size_t rank = (size_t)k * a.jm + j;
    double* cl5[5] = { a.c, a.cloud, a.ice, a.p_rain, a.p_snow };
    for (int i = 0; i < a.im; i++)
        for (int n = 0; n < 5; n++) {
            double* p = cl5[n] + (size_t)i * n2 + q;
            if (*p < 0.) *p = 0.;
        }
    const double t_surface = a.t[q];
    uint8_t def = 0; double nuc = 0., dep = 0.;
    for (int i = a.im - 2; i >= 0; i--) {  // the later a point is visited, the later its assignment
        const size_t id = (size_t)i * n2 + q;
        const Local L = local_terms(a, i, a.t[id], t_surface, a.c[id], a.ice[id]);
        if (L.nuc_def) { def |= 1; nuc = L.nuc_val; }
        if (L.dep_def) { def |= 2; dep = L.dep_val; }
    }
    a.out_def[rank] = def; a.out_nuc[rank] = nuc; a.out_dep[rank] = dep;
}

// step 2: what every column inherits from the columns before it.  "The last assignment so far" is an associative
// operation without arithmetic (take the right operand where it is defined), so it is scanned in parallel: inside
// chunks of CHUNK consecutive ranks by one workgroup each, then over the chunk totals by one thread.
constexpr int CHUNK = 1024;

__global__ __launch_bounds__(CHUNK) void k_ice_chunk_scan(IceArgs a) {
    __shared__ uint8_t sd[2][CHUNK];
    __shared__ double sn[2][CHUNK], sp[2][CHUNK];
    const size_t n2 = (size_t)a.jm * a.km;
    const int t = threadIdx.x;
    const size_t r = (size_t)blockIdx.x * CHUNK + t;
    uint8_t def = 0; double nuc = 0., dep = 0.;
    if (r < n2) { def = a.out_def[r]; nuc = a.out_nuc[r]; dep = a.out_dep[r]; }
    int cur = 0;
    sd[0][t] = def; sn[0][t] = nuc; sp[0][t] = dep;
    __syncthreads();
    for (int off = 1; off < CHUNK; off <<= 1) {  // inclusive scan
        if (t >= off) {
            const uint8_t ld = sd[cur][t - off];
            if (!(def & 1) && (ld & 1)) { def |= 1; nuc = sn[cur][t - off]; }
            if (!(def & 2) && (ld & 2)) { def |= 2; dep = sp[cur][t - off]; }
        }
        cur ^= 1;
        sd[cur][t] = def; sn[cur][t] = nuc; sp[cur][t] = dep;
        __syncthreads();
    }
    if (r < n2) {  // exclusive: what the columns before this one in the chunk left
        a.in_def[r] = t ? sd[cur][t - 1] : 0; a.in_nuc[r] = t ? sn[cur][t - 1] : 0.; a.in_dep[r] = t ? sp[cur][t - 1] : 0.;
    }
    if (t == CHUNK - 1) { a.chunk_def[blockIdx.x] = def; a.chunk_nuc[blockIdx.x] = nuc; a.chunk_dep[blockIdx.x] = dep; }
}

__global__ void k_ice_chunk_carry(IceArgs a) {
    if (blockIdx.x || threadIdx.x) return;
    uint8_t def = 0; double nuc = 0., dep = 0.;
    for (int c = 0; c < a.nchunks; c++) {
        const uint8_t d = a.chunk_def[c]; const double n = a.chunk_nuc[c], p = a.chunk_dep[c];
        a.chunk_def[c] = def; a.chunk_nuc[c] = nuc; a.chunk_dep[c] = dep;  // the total becomes the inheritance
        if (d & 1) { def |= 1; nuc = n; }
        if (d & 2) { def |= 2; dep = p; }
    }
    a.last[0] = (def & 1) ? 1. : 0.; a.last[1] = nuc; a.last[2] = (def & 2) ? 1. : 0.; a.last[3] = dep;
}

// the members as a point finds them when neither it nor the levels above it in its column assign them: from the
// columns before it, else (first pass) the caller's value or (later passes) the grid's last assignment
struct Inherited { double nuc_first, nuc_later, dep; };
__device__ __forceinline__ Inherited inherited(const IceArgs& a, size_t rank, bool have_n, double vn, bool have_d, double vd) {
    const uint8_t in_def = a.in_def[rank];
    if (!have_n && (in_def & 1)) { have_n = true; vn = a.in_nuc[rank]; }
    if (!have_d && (in_def & 2)) { have_d = true; vd = a.in_dep[rank]; }
    const size_t c = rank / CHUNK;
    const uint8_t ch_def = a.chunk_def[c];
    if (!have_n && (ch_def & 1)) { have_n = true; vn = a.chunk_nuc[c]; }
    if (!have_d && (ch_def & 2)) { have_d = true; vd = a.chunk_dep[c]; }
    Inherited r;
    r.nuc_first = have_n ? vn : a.state_nuc;
    r.nuc_later = have_n ? vn : (a.last[0] != 0. ? a.last[1] : a.state_nuc);
    r.dep = have_d ? vd : (a.last[2] != 0. ? a.last[3] : a.state_dep);  // the preparatory pass assigns S_i_dep like the others
    return r;
}

// step 3: the preparatory pass (:2875-2954) and the five full passes (:2959-3135) of one column
__global__ void k_ice_column_passes(IceArgs a) {
    const size_t n2 = (size_t)a.jm * a.km;
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n2) return;
    const int im = a.im;
    const int j = (int)(q / a.km), k = (int)(q - (size_t)j * a.km);
    const size_t rank = (size_t)k * a.jm + j;
    const double t_0 = a.t_0, dr = a.dr;
    const double exp_pressure = a.g / (1.e-2 * a.gam * a.R_Air);
    const double t_surface = a.t[q];
#define AT(F, i) a.F[(size_t)(i) * n2 + q]
    const Inherited inh = inherited(a, rank, false, 0., false, 0.);
    const double nuc_first = inh.nuc_first, nuc_later = inh.nuc_later, dep_in = inh.dep;

    // preparatory pass
    AT(p_rain, im - 1) = 0.; AT(p_snow, im - 1) = 0.; AT(s_r, im - 1) = 0.; AT(s_s, im - 1) = 0.;
    {
        double pr_above = 0., ps_above = 0., sr_above = 0., ss_above = 0.;
        bool land_above = (AT(flags, im - 1) & FL_LAND) != 0;
        for (int i = im - 2; i >= 0; i--) {
            const double cc = AT(c, i), cl = AT(cloud, i), ic = AT(ice, i);
            const Local L = local_terms(a, i, AT(t, i), t_surface, cc, ic);
            double S_c_au = ((L.t_u >= t_0) && (c_c_au * cl > 0.)) ? c_c_au * cl : 0.;
            double S_i_au = ((L.t_u < t_0) && (c_i_au * ic > 0.)) ? c_i_au * ic : 0.;
            const bool land = (AT(flags, i) & FL_LAND) != 0;
            double sr = S_c_au, ss = S_i_au;
            if (land && land_above) { sr = 0.; ss = 0.; }
            AT(s_r, i) = sr; AT(s_s, i) = ss;
            const double pr = at_least_zero(pr_above + (sr + sr_above) * .5 * L.r_humid * dr * 200.);
            const double ps = at_least_zero(ps_above + (ss + ss_above) * .5 * L.r_humid * dr * 200.);
            AT(p_rain, i) = pr; AT(p_snow, i) = ps;
            pr_above = pr; ps_above = ps; sr_above = sr; ss_above = ss; land_above = land;
        }
    }
    // full passes
    for (int iter_prec = 1; iter_prec <= 5; iter_prec++) {
        double S_nuc = iter_prec == 1 ? nuc_first : nuc_later, S_i_dep = dep_in;
        AT(p_rain, im - 1) = 0.; AT(p_snow, im - 1) = 0.;
        double pr_above = 0., ps_above = 0., sr_above = AT(s_r, im - 1), ss_above = AT(s_s, im - 1);
        bool land_above = (AT(flags, im - 1) & FL_LAND) != 0;
        for (int i = im - 2; i >= 0; i--) {
            const double cc = AT(c, i), cl = AT(cloud, i), ic = AT(ice, i), pr0 = AT(p_rain, i), ps0 = AT(p_snow, i);
            const Local L = local_terms(a, i, AT(t, i), t_surface, cc, ic);
            const double t_u = L.t_u;
            if (L.nuc_def) S_nuc = L.nuc_val;
            if (L.dep_def) S_i_dep = L.dep_val;
            const double S_c_frz = ((t_u < t_hn) && (cl > 0.)) ? cl / dt_rain_dim : 0.;
            const double S_c_au = ((t_u >= t_0) && (cl > 0.)) ? c_c_au * cl : 0.;
            const double S_i_au = ((t_u <= t_0) && (ic > 0.)) ? c_i_au * ic : 0.;
            const double S_d_au = (t_u <= t_0) ? S_i_dep / (1.5 * (cr_pow((m_s_0 / L.m_i), (2. / 3.)) - 1.)) : 0.;
            const double S_ac = (t_u > t_0) ? c_ac * cl * cr_pow(pr0, (7. / 9.)) : 0.;
            const double S_rim = (t_u < t_0) ? c_rim * cl * ps0 : 0.;
            const double S_shed = (t_u >= t_0) ? c_rim * cl * ps0 : 0.;
            double S_agg = 0., S_i_cri = 0., S_r_cri = 0.;
            if (t_u <= t_0) {
                S_agg = c_agg * ic * ps0;
                S_i_cri = c_i_cri * ic * cr_pow(pr0, (7. / 9.));
                S_r_cri = c_r_cri * ic / L.m_i * cr_pow(pr0, (13. / 9.));
            }
            const double S_ev = (t_u >= t_0) ? a_ev * (1. + b_ev * cr_pow(pr0, (1. / 6.))) * (L.q_Rain - cc) * cr_pow(pr0, (4. / 9.)) : 0.;
            const double S_s_dep = (t_u < t_0) ? c_s_dep * (1. + b_s_dep * cr_pow(ps0, (5. / 26.))) * (cc - L.q_Ice) * cr_pow(ps0, (8. / 13.)) : 0.;
            const double S_i_melt = ((t_u > t_0) && (ic > 0.)) ? ic / dt_snow_dim : 0.;
            double S_s_melt = 0.;
            if (t_u > t_0) {
                const double p_SL = .01 * (a.r_air * a.R_Air * t_surface * t_0);
                const double hight = (double)i * (a.L_atm / (double)(im - 1));
                const double p_t_in = cr_pow(((t_0 - a.gam * hight * 1.e-2) / t_0), exp_pressure) * p_SL;
                double t0v = t_0;
                const double E_Rain_t_in = a.hp * exp_func(t0v, 17.2694, 35.86);
                const double q_Rain_t_in = a.ep * E_Rain_t_in / (p_t_in - E_Rain_t_in);
                S_s_melt = c_s_melt * (1. + b_s_melt * cr_pow(ps0, (5. / 26.))) * ((t_u - t_0) + a_s_melt * (cc - q_Rain_t_in))
                           * cr_pow(ps0, (8. / 13.));
            }
            const double S_r_frz = (t_r_frz - t_u > 0.) ? c_r_frz * cr_pow((t_r_frz - t_u), (3. / 2.)) * cr_pow(pr0, (3. / 2.)) : 0.;
            const double scc = AT(s_c_c, i);
            double sv = -scc + S_ev - S_i_dep - S_s_dep - S_nuc;
            double sc = scc - S_c_au - S_ac - S_c_frz + S_i_melt - S_rim - S_shed;
            double si = S_nuc + S_c_frz + S_i_dep - S_i_melt - S_i_au - S_d_au - S_agg - S_i_cri;
            double sr = S_c_au + S_ac - S_ev + S_shed - S_r_cri - S_r_frz + S_s_melt;
            double ss = S_d_au + S_s_dep + S_i_au + S_rim + S_agg + S_i_cri + S_r_cri + S_r_frz - S_s_melt;
            const bool land = (AT(flags, i) & FL_LAND) != 0;
            if (land && land_above) { AT(s_c_c, i) = 0.; sv = 0.; sc = 0.; si = 0.; sr = 0.; ss = 0.; }
            AT(s_v, i) = sv; AT(s_c, i) = sc; AT(s_i, i) = si; AT(s_r, i) = sr; AT(s_s, i) = ss;
            const double pr = at_least_zero(pr_above + (sr + sr_above) * .5 * L.r_humid * dr * 200.);
            const double ps = at_least_zero(ps_above + (ss + ss_above) * .5 * L.r_humid * dr * 200.);
            AT(p_rain, i) = pr; AT(p_snow, i) = ps;
            pr_above = pr; ps_above = ps; sr_above = sr; ss_above = ss; land_above = land;
        }
    }
#undef AT
}

// ---- the same scheme, one thread per grid point -----------------------------------------------------------
// Inside a pass only the column integrals of rain and snow run along a column; the sources of a grid point use
// the integrals of the pass BEFORE (a point reads P_rain(i), P_snow(i) before it overwrites them).  So a workgroup
// takes COLS (= blockDim.x) neighbouring columns with one thread per level: every thread keeps what its point fixes for all
// passes (densities, saturation values, the inherited members) in registers, publishes S_r, S_s of the pass in
// LDS, and 2 x COLS threads integrate rain and snow down their columns between two barriers.  Used while
// COLS * im threads fit a workgroup; taller grids take the column kernels above.
constexpr int MAX_LEVELS = 128;  // at least 8 columns per workgroup
// columns per workgroup: as many as fit 1024 threads (one workgroup then fills a CU's 16 wave slots at 128 VGPRs)
inline int cols_per_group(int im) { const int c = 1024 / im; return c > 32 ? 32 : c; }

struct PointGeom {
    size_t q, id, rank;
    int i;
    bool valid, physics;  // inside the grid; below the top level
};

__device__ __forceinline__ PointGeom point_geom(const IceArgs& a) {
    PointGeom g;
    const size_t n2 = (size_t)a.jm * a.km;
    g.q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    g.i = (int)threadIdx.y;
    g.valid = g.q < n2;
    if (!g.valid) g.q = n2 - 1;
    g.id = (size_t)g.i * n2 + g.q;
    const size_t j = g.q / a.km, k = g.q - j * a.km;
    g.rank = k * a.jm + j;
    g.physics = g.valid && g.i <= a.im - 2;
    return g;
}

// step 1 of 3: clamps (:2861-2871) and the column's last assignments of S_nuc and S_i_dep
__global__ __launch_bounds__(1024) void k_ice_points_summary(IceArgs a) {
    extern __shared__ double lds[];  // [2][im][COLS] doubles, then [im][COLS] bytes
    const int COLS = (int)blockDim.x;
    const int plane = a.im * COLS;
    double *s_nuc = lds, *s_dep = lds + plane;
    uint8_t* s_def = (uint8_t*)(lds + 2 * plane);
    const PointGeom g = point_geom(a);
    const int tx = threadIdx.x;
    uint8_t def = 0; double nuc = 0., dep = 0.;
    if (g.valid) {
        double* cl5[5] = { a.c, a.cloud, a.ice, a.p_rain, a.p_snow };
        double v[5];
        for (int n = 0; n < 5; n++) { v[n] = cl5[n][g.id]; if (v[n] < 0.) { v[n] = 0.; cl5[n][g.id] = 0.; } }
        if (g.physics) {
            const Local L = local_terms(a, g.i, a.t[g.id], a.t[g.q], v[0], v[2]);
            if (L.nuc_def) { def |= 1; nuc = L.nuc_val; }
            if (L.dep_def) { def |= 2; dep = L.dep_val; }
        }
    }
    s_def[(g.i) * COLS + tx] = def; s_nuc[(g.i) * COLS + tx] = nuc; s_dep[(g.i) * COLS + tx] = dep;
    __syncthreads();
    if (g.i == 0 && g.valid) {  // the lowest assigning level is the one visited last
        uint8_t od = 0; double on = 0., odp = 0.;
        for (int i = a.im - 2; i >= 0; i--) {
            const uint8_t d = s_def[(i) * COLS + tx];
            if (d & 1) { od |= 1; on = s_nuc[(i) * COLS + tx]; }
            if (d & 2) { od |= 2; odp = s_dep[(i) * COLS + tx]; }
        }
        a.out_def[g.rank] = od; a.out_nuc[g.rank] = on; a.out_dep[g.rank] = odp;
    }
}

// step 3 of 3: the preparatory pass and the five full passes, all levels of COLS columns at once
__global__ __launch_bounds__(1024) void k_ice_points_passes(IceArgs a) {
    extern __shared__ double lds[];  // [5][im][COLS] doubles, then [im][COLS] bytes
    const int COLS = (int)blockDim.x;
    const int plane = a.im * COLS;
    double *s_r = lds, *s_s = lds + plane, *s_rh = lds + 2 * plane, *s_pr = lds + 3 * plane, *s_ps = lds + 4 * plane;
    double *s_nuc = s_pr, *s_dep = s_ps;  // the assignments are only needed before the first integration
    uint8_t* s_def = (uint8_t*)(lds + 5 * plane);
    const PointGeom g = point_geom(a);
    const int tx = threadIdx.x, i = g.i, im = a.im;
    const size_t n2 = (size_t)a.jm * a.km;
    const double t_0 = a.t_0, dr = a.dr;
    // what the point fixes for all passes
    double cc = 0., cl = 0., ic = 0., scc = 0.;
    Local L = {};
    bool land_rule = false;
    if (g.physics) {
        cc = a.c[g.id]; cl = a.cloud[g.id]; ic = a.ice[g.id]; scc = a.s_c_c[g.id];
        L = local_terms(a, i, a.t[g.id], a.t[g.q], cc, ic);
        land_rule = (a.flags[g.id] & FL_LAND) && (a.flags[g.id + n2] & FL_LAND);
    }
    s_def[(i) * COLS + tx] = (uint8_t)((L.nuc_def ? 1 : 0) | (L.dep_def ? 2 : 0));
    s_nuc[(i) * COLS + tx] = L.nuc_val; s_dep[(i) * COLS + tx] = L.dep_val;
    s_rh[(i) * COLS + tx] = L.r_humid;
    __syncthreads();
    // the members as this point finds them: its own assignment, else the nearest one above it in the column, else
    // what the column inherits, else (first pass) the caller's value or (later) the grid's last assignment
    double S_nuc_first = 0., S_nuc_later = 0., S_i_dep = 0.;
    if (g.physics) {
        bool have_n = L.nuc_def, have_d = L.dep_def;
        double vn = L.nuc_val, vd = L.dep_val;
        for (int l = i + 1; l <= im - 2 && !(have_n && have_d); l++) {
            const uint8_t d = s_def[(l) * COLS + tx];
            if (!have_n && (d & 1)) { have_n = true; vn = s_nuc[(l) * COLS + tx]; }
            if (!have_d && (d & 2)) { have_d = true; vd = s_dep[(l) * COLS + tx]; }
        }
        const Inherited inh = inherited(a, g.rank, have_n, vn, have_d, vd);
        S_nuc_first = inh.nuc_first; S_nuc_later = inh.nuc_later; S_i_dep = inh.dep;
    }
    // sources that no pass changes
    const double t_u = L.t_u;
    double S_c_frz = 0., S_c_au = 0., S_i_au = 0., S_d_au = 0., S_i_melt = 0., melt = 0., frz = 0.;
    double sr = 0., ss = 0.;
    if (g.physics) {
        S_c_frz = ((t_u < t_hn) && (cl > 0.)) ? cl / dt_rain_dim : 0.;
        S_c_au = ((t_u >= t_0) && (cl > 0.)) ? c_c_au * cl : 0.;
        S_i_au = ((t_u <= t_0) && (ic > 0.)) ? c_i_au * ic : 0.;
        S_d_au = (t_u <= t_0) ? S_i_dep / (1.5 * (cr_pow((m_s_0 / L.m_i), (2. / 3.)) - 1.)) : 0.;
        S_i_melt = ((t_u > t_0) && (ic > 0.)) ? ic / dt_snow_dim : 0.;
        if (t_u > t_0) {  // the bracket of S_s_melt, :3087-3096
            const double exp_pressure = a.g / (1.e-2 * a.gam * a.R_Air);
            const double p_SL = .01 * (a.r_air * a.R_Air * a.t[g.q] * t_0);
            const double hight = (double)i * (a.L_atm / (double)(im - 1));
            const double p_t_in = cr_pow(((t_0 - a.gam * hight * 1.e-2) / t_0), exp_pressure) * p_SL;
            const double E_Rain_t_in = a.hp * exp_func(t_0, 17.2694, 35.86);
            const double q_Rain_t_in = a.ep * E_Rain_t_in / (p_t_in - E_Rain_t_in);
            melt = ((t_u - t_0) + a_s_melt * (cc - q_Rain_t_in));
        }
        if (t_r_frz - t_u > 0.) frz = cr_pow((t_r_frz - t_u), (3. / 2.));
        // preparatory pass, :2875-2954
        sr = ((t_u >= t_0) && (c_c_au * cl > 0.)) ? c_c_au * cl : 0.;
        ss = ((t_u < t_0) && (c_i_au * ic > 0.)) ? c_i_au * ic : 0.;
        if (land_rule) { sr = 0.; ss = 0.; }
    }
    double sv = 0., sc = 0., si = 0.;
    for (int pass = 0; pass <= 5; pass++) {
        if (pass > 0 && g.physics) {
            const double pr0 = s_pr[(i) * COLS + tx], ps0 = s_ps[(i) * COLS + tx];
            const double S_nuc = pass == 1 ? S_nuc_first : S_nuc_later;
            const double S_ac = (t_u > t_0) ? c_ac * cl * cr_pow(pr0, (7. / 9.)) : 0.;
            const double S_rim = (t_u < t_0) ? c_rim * cl * ps0 : 0.;
            const double S_shed = (t_u >= t_0) ? c_rim * cl * ps0 : 0.;
            double S_agg = 0., S_i_cri = 0., S_r_cri = 0.;
            if (t_u <= t_0) {
                S_agg = c_agg * ic * ps0;
                S_i_cri = c_i_cri * ic * cr_pow(pr0, (7. / 9.));
                S_r_cri = c_r_cri * ic / L.m_i * cr_pow(pr0, (13. / 9.));
            }
            const double S_ev = (t_u >= t_0) ? a_ev * (1. + b_ev * cr_pow(pr0, (1. / 6.))) * (L.q_Rain - cc) * cr_pow(pr0, (4. / 9.)) : 0.;
            const double S_s_dep = (t_u < t_0) ? c_s_dep * (1. + b_s_dep * cr_pow(ps0, (5. / 26.))) * (cc - L.q_Ice) * cr_pow(ps0, (8. / 13.)) : 0.;
            const double S_s_melt = (t_u > t_0) ? c_s_melt * (1. + b_s_melt * cr_pow(ps0, (5. / 26.))) * melt * cr_pow(ps0, (8. / 13.)) : 0.;
            const double S_r_frz = (t_r_frz - t_u > 0.) ? c_r_frz * frz * cr_pow(pr0, (3. / 2.)) : 0.;
            sv = -scc + S_ev - S_i_dep - S_s_dep - S_nuc;
            sc = scc - S_c_au - S_ac - S_c_frz + S_i_melt - S_rim - S_shed;
            si = S_nuc + S_c_frz + S_i_dep - S_i_melt - S_i_au - S_d_au - S_agg - S_i_cri;
            sr = S_c_au + S_ac - S_ev + S_shed - S_r_cri - S_r_frz + S_s_melt;
            ss = S_d_au + S_s_dep + S_i_au + S_rim + S_agg + S_i_cri + S_r_cri + S_r_frz - S_s_melt;
            if (land_rule) { scc = 0.; sv = 0.; sc = 0.; si = 0.; sr = 0.; ss = 0.; }
        }
        s_r[(i) * COLS + tx] = sr; s_s[(i) * COLS + tx] = ss;  // the top level keeps its zeros (:2878-2881)
        __syncthreads();
        if (i < 2) {  // level 0's thread integrates rain, level 1's snow (:2946-2953, :3118-3125)
            const double* src = i == 0 ? s_r : s_s;
            double* dst = i == 0 ? s_pr : s_ps;
            double p_above = 0.;
            dst[(im - 1) * COLS + tx] = 0.;
            for (int l = im - 2; l >= 0; l--) {
                p_above = at_least_zero(p_above + (src[(l) * COLS + tx] + src[(l + 1) * COLS + tx]) * .5 * s_rh[(l) * COLS + tx] * dr * 200.);
                dst[(l) * COLS + tx] = p_above;
            }
        }
        __syncthreads();
    }
    if (!g.valid) return;
    a.p_rain[g.id] = s_pr[(i) * COLS + tx]; a.p_snow[g.id] = s_ps[(i) * COLS + tx];
    a.s_r[g.id] = sr; a.s_s[g.id] = ss;
    if (g.physics) {
        a.s_v[g.id] = sv; a.s_c[g.id] = sc; a.s_i[g.id] = si;
        if (land_rule) a.s_c_c[g.id] = 0.;
    }
}

// ---------------------------------------------------------------------------------------------------------
// BC_Thermo::Ice_Water_Saturation_Adjustment (atmosphere/BC_Thermo.cpp:2509-2796)
//
// Local to a grid point but for (1) the column's sea-level pressure, which uses t(0) as the column's first
// point has just left it, and (2) the class member q_v_hyp: a cold point assigns it before its iteration only
// if its cloud water and cloud ice come in non-negative (:2658-2661); a point whose input is negative starts
// from what the cold point visited before it (k outer, j, i upwards) left behind.  Every cold point assigns it
// inside its iteration, so only a column whose FIRST cold point has negative input depends on the columns
// before it.  Such columns are rare (extrapolated boundary values); they are left untouched by the column
// kernel and done in visiting order by the scan kernel, which also hands the member on from column to column.
struct AdjArgs {
    double *c, *cn, *cloud, *cloudn, *ice, *icen, *t, *s_c_c;
    const uint8_t* flags;
    int im, jm, km;
    double t_0, g, gam, R_Air, r_air, R_WaterVapour, L_atm, hp, ep, lv, ls, cp_l, dt_dim;
    double* out;      // per column rank: q_v_hyp after the column
    uint8_t* kind;    // per column rank: 0 no cold point, 1 out[] valid, 2 depends on the columns before it
    double* last;     // [0] q_v_hyp after the whole grid, [1] != 0: columns were left for the serial kernel
    double* chunk_out;   // per chunk of CHUNK ranks: out[] of its last column with a cold point
    uint8_t* chunk_kind; // per chunk: bit 0 chunk_out valid, bit 1 a column of the chunk was left for the serial kernel
    int nchunks;
    double state;     // q_v_hyp as the call found it
    // the resumable point kernel (k_adj_pass) and its scan
    double* hyp;          // per grid point: q_v_hyp as the finished cold point left it
    uint8_t* status;      // per grid point: bit 0 finished, bit 1 it was a cold point
    uint8_t* col_state;   // per column rank: 0 unfinished, 1 finished (out[] valid), 2 finished without a cold point
    double* carry;        // per column rank: out[] of the nearest column before it that is not of state 2 ...
    uint8_t* carry_state; // ... and that column's state: 1 the value is known, 0 not yet, 2 look in the chunk's entry
    double* chunk_carry; uint8_t* chunk_carry_state;  // the same per chunk, from the chunks before it
    int* chunk_unfinished;
};

constexpr double t_00 = 236.15;

// does the member get a value before the iteration? (:2658-2661)
__device__ __forceinline__ bool hyp_assigned(double q_c_b, double q_i_b) {
    return ((q_c_b > 0.) && (q_i_b > 0.)) || ((q_c_b >= 0.) && (q_i_b == 0.)) || ((q_c_b == 0.) && (q_i_b > 0.));
}

__device__ __forceinline__ double hyp_value(double q_c_b, double q_i_b, double q_Rain, double q_Ice, double q_v_hyp) {
    if ((q_c_b > 0.) && (q_i_b > 0.)) q_v_hyp = (q_c_b * q_Rain + q_i_b * q_Ice) / (q_c_b + q_i_b);
    if ((q_c_b >= 0.) && (q_i_b == 0.)) q_v_hyp = q_Rain;
    if ((q_c_b == 0.) && (q_i_b > 0.)) q_v_hyp = q_Ice;
    return q_v_hyp;
}

// 0: no cold point; 1: the first cold point assigns q_v_hyp itself; 2: it needs the inherited value
__device__ int adj_column_kind(const AdjArgs& a, size_t q) {
    const size_t n2 = (size_t)a.jm * a.km;
    for (int i = 0; i < a.im; i++) {
        const size_t id = (size_t)i * n2 + q;
        const double t_Celsius = a.t[id] * a.t_0 - a.t_0;
        if (t_Celsius < 0.) {
            const double q_c_b = (t_Celsius < t_00 - a.t_0) ? 0. : a.cloud[id];
            return hyp_assigned(q_c_b, a.ice[id]) ? 1 : 2;
        }
    }
    return 0;
}

// What one grid point makes of its input.  branch: 0 neither (t is not a number), 1 warm, 2 cold.
struct AdjOut { double c, cloud, ice, t, q_v_hyp; int branch; };

__device__ AdjOut adj_point(const AdjArgs& a, int i, double t_in, double t_ground, double cc, double cl, double ic,
                            double q_v_hyp) {
    const double t_0 = a.t_0, lv = a.lv, ls = a.ls, cp_l = a.cp_l;
    const double t_Celsius_2 = t_00 - t_0;
    const double exp_pressure = a.g / (1.e-2 * a.gam * a.R_Air);
    AdjOut o;
    const double t_u = t_in * t_0;
    const double t_Celsius = t_u - t_0;
    const double p_SL = .01 * (a.r_air * a.R_Air * t_ground * t_0);
    const double hight = (double)i * (a.L_atm / (double)(a.im - 1));
    double p_h;
    if (i != 0) p_h = cr_pow(((t_in * t_0 - a.gam * hight * 1.e-2) / (t_in * t_0)), exp_pressure) * p_SL;
    else p_h = p_SL;
    if (t_Celsius >= 0.) {  // warm cloud phase, :2566-2629
        double q_T = cc + cl;
        double E_Rain = a.hp * exp_func(t_u, 17.2694, 35.86);
        double q_Rain = a.ep * E_Rain / (p_h - E_Rain);
        double q_Rain_n = q_Rain, T_it = t_u;
        if (q_T <= q_Rain) {
            cc = q_T; cl = 0.; ic = 0.;
        } else {
            for (int iter_prec = 1; iter_prec <= 20; iter_prec++) {
                T_it = (t_u + lv / cp_l * cc - lv / cp_l * q_Rain);
                E_Rain = a.hp * exp_func(T_it, 17.2694, 35.86);
                q_Rain = a.ep * E_Rain / (p_h - E_Rain);
                q_Rain = .5 * (q_Rain_n + q_Rain);
                cc = q_Rain;
                cl = q_T - cc;
                ic = 0.;
                q_T = cc + cl;
                if (cc < 0.) cc = 0.;
                if (cl < 0.) cl = 0.;
                if ((q_Rain_n) > 2.220446049250313e-16 && fabs(q_Rain / q_Rain_n - 1.) <= 1.e-5) break;
                q_Rain_n = q_Rain;
            }
        }
        o.t = T_it / t_0; o.branch = 1;
    } else if (t_Celsius < 0.) {  // mixed cloud phase, :2637-2765
        if (t_Celsius < t_Celsius_2) cl = 0.;
        double q_v_b = cc, q_c_b = cl, q_i_b = ic;
        double T = t_u;
        double E_Rain = a.hp * exp_func(T, 17.2694, 35.86);
        double E_Ice = a.hp * exp_func(T, 21.8746, 7.66);
        double q_Rain = a.ep * E_Rain / (p_h - E_Rain);
        double q_Ice = a.ep * E_Ice / (p_h - E_Ice);
        q_v_hyp = hyp_value(q_c_b, q_i_b, q_Rain, q_Ice, q_v_hyp);
        for (int iter_prec = 1; iter_prec <= 20; iter_prec++) {
            double CND = (T - t_00) / (t_0 - t_00);
            if (T < t_00) CND = 0.;
            double DEP = (t_0 - T) / (t_0 - t_00);
            if (T > t_0) DEP = 0.;
            const double d_q_v = q_v_hyp - q_v_b;
            const double d_q_c = -d_q_v * CND;
            const double d_q_i = -d_q_v * DEP;
            const double d_t = (lv * d_q_c + ls * d_q_i) / cp_l;
            T = T + d_t;
            q_v_b = cc + d_q_v;
            q_c_b = cl + d_q_c;
            q_i_b = ic + d_q_i;
            if (q_v_b < 0.) q_v_b = 0.;
            if (q_c_b < 0.) q_c_b = 0.;
            if (q_i_b < 0.) q_i_b = 0.;
            if (i != 0) {
                if (T > a.gam * hight * 1.e-2) p_h = cr_pow(((T - a.gam * hight * 1.e-2) / (T)), exp_pressure) * p_SL;
                else p_h = p_SL;  // the reference logs a warning here
            } else p_h = p_SL;
            E_Rain = a.hp * exp_func(T, 17.2694, 35.86);
            E_Ice = a.hp * exp_func(T, 21.8746, 7.66);
            q_Rain = a.ep * E_Rain / (p_h - E_Rain);
            q_Ice = a.ep * E_Ice / (p_h - E_Ice);
            q_v_hyp = hyp_value(q_c_b, q_i_b, q_Rain, q_Ice, q_v_hyp);
            if (iter_prec >= 3 && (q_v_hyp) > 2.220446049250313e-16 && fabs(q_v_b / q_v_hyp - 1.) <= 1.e-5) break;
            q_v_b = .5 * (q_v_hyp + q_v_b);
        }
        cc = q_v_b; cl = q_c_b; ic = q_i_b;
        if (t_Celsius < t_Celsius_2) cl = 0.;
        o.t = T / t_0; o.branch = 2;
    } else {
        o.t = t_in; o.branch = 0;
    }
    o.c = cc; o.cloud = cl; o.ice = ic; o.q_v_hyp = q_v_hyp;
    return o;
}

// a point's results into the fields; c_in: the point's water vapour as it came in
__device__ __forceinline__ void adj_store(const AdjArgs& a, size_t id, const AdjOut& o, double c_in) {
    if (o.branch) {
        // S_c_c from the point's incoming cn and c (:2732; both are overwritten only after the iteration)
        if (o.branch == 2) a.s_c_c[id] = (a.flags[id] & FL_LAND) ? 0. : .5 * (a.cn[id] - c_in) / a.dt_dim;
        a.cn[id] = o.c; a.cloudn[id] = o.cloud; a.icen[id] = o.ice;
        a.t[id] = o.t;
    }
    // the closing clamp of c, cloud and ice (:2770-2778) touches the old-level arrays only
    a.c[id] = at_least_zero(o.c); a.cloud[id] = at_least_zero(o.cloud); a.ice[id] = at_least_zero(o.ice);
}

// one column, bottom to top; returns q_v_hyp as the column leaves it
__device__ double adj_column(const AdjArgs& a, size_t q, double q_v_hyp) {
    const size_t n2 = (size_t)a.jm * a.km;
    double t_ground = a.t[q];
    for (int i = 0; i < a.im; i++) {
        const size_t id = (size_t)i * n2 + q;
        const double c_in = a.c[id];
        const AdjOut o = adj_point(a, i, a.t[id], t_ground, c_in, a.cloud[id], a.ice[id], q_v_hyp);
        adj_store(a, id, o, c_in);
        q_v_hyp = o.q_v_hyp;
        if (i == 0) t_ground = o.t;  // the levels above see the ground temperature as its own point left it
    }
    return q_v_hyp;
}

__global__ void k_adj_columns(AdjArgs a) {
    const size_t n2 = (size_t)a.jm * a.km;
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n2) return;
    const int j = (int)(q / a.km), k = (int)(q - (size_t)j * a.km);
    const size_t rank = (size_t)k * a.jm + j;
    const int kind = adj_column_kind(a, q);
    a.kind[rank] = (uint8_t)kind;
    if (kind == 2) return;  // left for the serial kernel
    const double out = adj_column(a, q, 0.);  // the start value is never read: the first cold point assigns it
    a.out[rank] = out;
}

// The same with one thread per grid point (COLS neighbouring columns and all their levels per workgroup), in passes.
// A point can be finished when the ground point of its column is (the levels above need its new t) and, if it
// inherits q_v_hyp, when the cold point it inherits from is: the nearest one below it in its column, or - for the
// first cold point of a column - the last cold point of the nearest column before it that has one (carry).  A pass
// finishes every point whose inputs are there and leaves the others untouched; the scan between two passes hands
// the finished columns' last values on.  The first pass finishes everything but the runs of inheriting points at
// the first cold point of a column (in the model's loop these sit on the rows and columns that the boundary fills
// extrapolate); every further pass finishes at least the first unfinished column in visiting order.
__global__ __launch_bounds__(1024) void k_adj_pass(AdjArgs a) {
    extern __shared__ double lds[];  // [im][COLS] q_v_hyp of finished cold points, [COLS] ground t, then bytes
    const int COLS = (int)blockDim.x;
    const int im = a.im, tx = threadIdx.x, i = threadIdx.y;
    double* s_hyp = lds;
    double* s_t0 = lds + (size_t)im * COLS;
    uint8_t* s_flag = (uint8_t*)(s_t0 + COLS);   // [im][COLS]: 1 cold, 2 inherits q_v_hyp, 4 finished
    const size_t n2 = (size_t)a.jm * a.km;
    size_t q = (size_t)blockIdx.x * COLS + tx;
    const bool valid = q < n2;
    if (!valid) q = n2 - 1;
    const size_t id = (size_t)i * n2 + q;
    const size_t jj = q / a.km, kk = q - jj * a.km;
    const size_t rank = kk * a.jm + jj;
    const bool col_open = valid && a.col_state[rank] == 0;
    if (!__syncthreads_or(col_open)) return;  // nothing left to do for these columns
    // the point's own state; its input fields are untouched as long as it is unfinished
    const uint8_t st = a.status[id];
    bool done = (st & 1) != 0, cold = (st & 2) != 0, inherits = false;
    double t_in = 0., c_in = 0., cl_in = 0., ic_in = 0.;
    if (!done) {
        t_in = a.t[id]; c_in = a.c[id]; cl_in = a.cloud[id]; ic_in = a.ice[id];
        const double t_Celsius = t_in * a.t_0 - a.t_0;
        cold = t_Celsius < 0.;
        inherits = cold && !hyp_assigned((t_Celsius < t_00 - a.t_0) ? 0. : cl_in, ic_in);
    } else if (cold) s_hyp[i * COLS + tx] = a.hyp[id];
    s_flag[i * COLS + tx] = (uint8_t)((cold ? 1 : 0) | (inherits ? 2 : 0) | (done ? 4 : 0));
    // what the column inherits, if that is known by now
    bool carry_known = false; double carry = 0.;
    if (col_open) {
        uint8_t cs = a.carry_state[rank]; carry = a.carry[rank];
        if (cs == 2) { const size_t c = rank / CHUNK; cs = a.chunk_carry_state[c]; carry = a.chunk_carry[c]; }
        carry_known = cs == 1;
    }
    __syncthreads();
    bool first_cold = cold;  // no cold point below this one
    for (int l = i - 1; l >= 0 && first_cold; l--) if (s_flag[l * COLS + tx] & 1) first_cold = false;
    auto finish = [&](double t_ground, double hyp_in) {
        const AdjOut o = adj_point(a, i, t_in, t_ground, c_in, cl_in, ic_in, hyp_in);
        adj_store(a, id, o, c_in);
        a.status[id] = (uint8_t)(1 | (cold ? 2 : 0));
        if (cold) { a.hyp[id] = o.q_v_hyp; s_hyp[i * COLS + tx] = o.q_v_hyp; }
        done = true;
        return o;
    };
    if (i == 0) {  // the ground point
        double t0 = a.t[id];
        if (col_open && !done && (!inherits || carry_known)) t0 = finish(t_in, carry).t;
        s_t0[tx] = t0;
        if (done) s_flag[tx] |= 4;
    }
    __syncthreads();
    const bool ground_done = (s_flag[tx] & 4) != 0;
    const bool open = col_open && !done && i > 0 && ground_done;
    if (open && !inherits) finish(s_t0[tx], 0.);
    __syncthreads();
    if (done) s_flag[i * COLS + tx] |= 4;
    // the inheriting points, lowest first: every round finishes those whose cold point below is finished
    for (int round = 0; round < im; round++) {
        __syncthreads();
        bool ready = false; double hyp_in = 0.;
        if (open && !done) {
            if (first_cold) { ready = carry_known; hyp_in = carry; }
            else
                for (int l = i - 1; l >= 0; l--) {
                    const uint8_t f = s_flag[l * COLS + tx];
                    if (f & 1) { ready = (f & 4) != 0; hyp_in = s_hyp[l * COLS + tx]; break; }
                }
        }
        if (!__syncthreads_or(ready)) break;
        if (ready) { finish(s_t0[tx], hyp_in); s_flag[i * COLS + tx] |= 4; }
    }
    __syncthreads();
    if (i == 0 && col_open) {  // finished columns hand their last value on: the highest cold level is visited last
        bool all = true; int top_cold = -1;
        for (int l = 0; l < im; l++) {
            const uint8_t f = s_flag[l * COLS + tx];
            if (!(f & 4)) all = false;
            if (f & 1) top_cold = l;
        }
        if (all) {
            if (top_cold >= 0) a.out[rank] = s_hyp[top_cold * COLS + tx];
            a.col_state[rank] = top_cold >= 0 ? 1 : 2;
        }
    }
}

// Between two passes: for every column the value it inherits (exclusive scan of "the last column that is not of
// state 2" inside chunks of CHUNK ranks), the chunk totals, and the number of unfinished columns.
__global__ __launch_bounds__(CHUNK) void k_adj_carry_scan(AdjArgs a) {
    __shared__ uint8_t ss[2][CHUNK];
    __shared__ double sv[2][CHUNK];
    const size_t n2 = (size_t)a.jm * a.km;
    const int t = threadIdx.x;
    const size_t r = (size_t)blockIdx.x * CHUNK + t;
    uint8_t state = 2; double val = 0.;
    if (r < n2) { state = a.col_state[r]; if (state == 1) val = a.out[r]; }
    const int unfinished = __syncthreads_count(state == 0);
    int cur = 0;
    ss[0][t] = state; sv[0][t] = val;
    __syncthreads();
    for (int off = 1; off < CHUNK; off <<= 1) {  // inclusive scan: the right operand wins unless it is of state 2
        if (t >= off && state == 2) { state = ss[cur][t - off]; val = sv[cur][t - off]; }
        cur ^= 1;
        ss[cur][t] = state; sv[cur][t] = val;
        __syncthreads();
    }
    if (r < n2) { a.carry_state[r] = t ? ss[cur][t - 1] : 2; a.carry[r] = t ? sv[cur][t - 1] : 0.; }
    if (t == CHUNK - 1) { a.chunk_carry_state[blockIdx.x] = state; a.chunk_carry[blockIdx.x] = val; a.chunk_unfinished[blockIdx.x] = unfinished; }
}

__global__ void k_adj_carry_finish(AdjArgs a) {
    if (blockIdx.x || threadIdx.x) return;
    uint8_t state = 1; double val = a.state;  // before the first column: the member as the call found it
    int unfinished = 0;
    for (int c = 0; c < a.nchunks; c++) {
        const uint8_t cs = a.chunk_carry_state[c]; const double cv = a.chunk_carry[c];
        a.chunk_carry_state[c] = state; a.chunk_carry[c] = val;  // the total becomes what the chunk inherits
        if (cs != 2) { state = cs; val = cv; }
        unfinished += a.chunk_unfinished[c];
    }
    a.last[0] = val; a.last[1] = (double)unfinished;
}

// No column left over (the usual case): the member ends as the last column with a cold point left it - the
// highest such rank, found per chunk by a workgroup each and over the chunks by one thread.
__global__ __launch_bounds__(CHUNK) void k_adj_chunk_last(AdjArgs a) {
    __shared__ int s_last[CHUNK / 64];
    const size_t n2 = (size_t)a.jm * a.km;
    const int t = threadIdx.x;
    const size_t r = (size_t)blockIdx.x * CHUNK + t;
    const uint8_t kind = r < n2 ? a.kind[r] : 0;
    const int left_over = __syncthreads_or(kind == 2);
    const unsigned long long m = __ballot(kind == 1);  // highest lane of the wave whose column has a cold point
    if ((t & 63) == 0) s_last[t >> 6] = m ? (t | (63 - __builtin_clzll(m))) : -1;
    __syncthreads();
    if (t == 0) {
        int last = -1;
        for (int w = 0; w < CHUNK / 64; w++) if (s_last[w] >= 0) last = s_last[w];
        a.chunk_kind[blockIdx.x] = (uint8_t)((last >= 0 ? 1 : 0) | (left_over ? 2 : 0));
        a.chunk_out[blockIdx.x] = last >= 0 ? a.out[(size_t)blockIdx.x * CHUNK + last] : 0.;
    }
}

__global__ void k_adj_finish(AdjArgs a) {
    if (blockIdx.x || threadIdx.x) return;
    double v = a.state; bool left_over = false;
    for (int c = 0; c < a.nchunks; c++) {
        const uint8_t kd = a.chunk_kind[c];
        if (kd & 1) v = a.chunk_out[c];
        if (kd & 2) left_over = true;
    }
    a.last[0] = v; a.last[1] = left_over ? 1. : 0.;
}

// columns were left over: one workgroup; each thread owns a run of consecutive columns and takes its turn in
// visiting order
__global__ __launch_bounds__(1024) void k_adj_inherit(AdjArgs a) {
    __shared__ double carry;
    const size_t n2 = (size_t)a.jm * a.km;
    const size_t per = (n2 + blockDim.x - 1) / blockDim.x;
    const size_t r0 = (size_t)threadIdx.x * per, r1 = r0 + per < n2 ? r0 + per : n2;
    // a run without dependent columns is summarised ahead of its turn
    bool has_dep = false, def = false; double out = 0.;
    for (size_t r = r0; r < r1; r++) {
        const uint8_t kd = a.kind[r];
        if (kd == 2) has_dep = true;
        if (kd == 1) { def = true; out = a.out[r]; }
    }
    if (threadIdx.x == 0) carry = a.state;
    __syncthreads();
    for (unsigned turn = 0; turn < blockDim.x; turn++) {  // every thread runs all turns: the barrier is uniform
        if (turn == threadIdx.x) {
            if (!has_dep) { if (def) carry = out; }
            else {
                double v = carry;
                for (size_t r = r0; r < r1; r++) {
                    const uint8_t kd = a.kind[r];
                    if (kd == 1) v = a.out[r];
                    else if (kd == 2) {
                        const size_t k = r / a.jm, j = r - k * a.jm;
                        v = adj_column(a, j * a.km + k, v);
                    }
                }
                carry = v;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) a.last[0] = carry;
}

}  // namespace

extern "C" int atomdev_atm_two_category_ice_scheme(atomdev_t* h, double state[2]) {
    ATOMDEV_REQUIRE(h && h->model == ATOMDEV_ATM && state, "atomdev_atm_two_category_ice_scheme: bad argument");
    ATOMDEV_REQUIRE(h->im >= 8, "atomdev_atm_two_category_ice_scheme: needs at least 8 levels (scratch layout)");
    ATOMDEV_TRY(hipSetDevice(h->device));
    if (int rc = atomdev_refresh_flags(h)) return rc;
    const atomdev_params& P = h->p;
    IceArgs a;
    a.c = h->f[ATOMDEV_C]; a.cloud = h->f[ATOMDEV_CLOUD]; a.ice = h->f[ATOMDEV_ICE];
    a.p_rain = h->f[ATOMDEV_P_RAIN]; a.p_snow = h->f[ATOMDEV_P_SNOW];
    a.s_v = h->f[ATOMDEV_S_V]; a.s_c = h->f[ATOMDEV_S_C]; a.s_i = h->f[ATOMDEV_S_I]; a.s_r = h->f[ATOMDEV_S_R];
    a.s_s = h->f[ATOMDEV_S_S]; a.s_c_c = h->f[ATOMDEV_S_C_C];
    a.t = h->f[ATOMDEV_T]; a.flags = h->flags;
    a.im = h->im; a.jm = h->jm; a.km = h->km;
    a.t_0 = P.t_0; a.g = P.g; a.gam = P.gam; a.R_Air = P.R_Air; a.r_air = P.r_air; a.R_WaterVapour = P.R_WaterVapour;
    a.L_atm = P.L_atm; a.hp = P.hp; a.ep = P.ep; a.dr = P.dr;
    // per-column scratch in the first volume scratch field (the sweep's prep kernel refills it on every call)
    const size_t n2 = h->n2;
    double* s = h->scr[0];
    a.out_nuc = s; a.out_dep = s + n2; a.in_nuc = s + 2 * n2; a.in_dep = s + 3 * n2;
    a.out_def = (uint8_t*)(s + 4 * n2); a.in_def = a.out_def + n2;
    a.last = s + 6 * n2;
    a.nchunks = (int)((n2 + CHUNK - 1) / CHUNK);
    a.chunk_nuc = a.last + 8; a.chunk_dep = a.chunk_nuc + a.nchunks; a.chunk_def = (uint8_t*)(a.chunk_dep + a.nchunks);
    ATOMDEV_REQUIRE(6 * n2 + 8 + 3 * (size_t)a.nchunks <= h->n3, "atomdev_atm_two_category_ice_scheme: grid too small for the scratch layout");
    a.state_nuc = state[0]; a.state_dep = state[1];
    if (h->im <= MAX_LEVELS && h->im >= 2) {
        const int COLS = cols_per_group(h->im);
        const unsigned nb = (unsigned)((n2 + COLS - 1) / COLS);
        const size_t plane = (size_t)h->im * COLS;
        hipLaunchKernelGGL(k_ice_points_summary, dim3(nb), dim3(COLS, h->im), 2 * plane * sizeof(double) + plane, h->stream, a);
        hipLaunchKernelGGL(k_ice_chunk_scan, dim3(a.nchunks), dim3(CHUNK), 0, h->stream, a);
        hipLaunchKernelGGL(k_ice_chunk_carry, dim3(1), dim3(64), 0, h->stream, a);
        hipLaunchKernelGGL(k_ice_points_passes, dim3(nb), dim3(COLS, h->im), 5 * plane * sizeof(double) + plane, h->stream, a);
    } else {
        const unsigned nb = (unsigned)((n2 + TPB - 1) / TPB);
        hipLaunchKernelGGL(k_ice_column_summary, dim3(nb), dim3(TPB), 0, h->stream, a);
        hipLaunchKernelGGL(k_ice_chunk_scan, dim3(a.nchunks), dim3(CHUNK), 0, h->stream, a);
        hipLaunchKernelGGL(k_ice_chunk_carry, dim3(1), dim3(64), 0, h->stream, a);
        hipLaunchKernelGGL(k_ice_column_passes, dim3(nb), dim3(TPB), 0, h->stream, a);
    }
    ATOMDEV_TRY(hipGetLastError());
    double last[4];
    ATOMDEV_TRY(hipMemcpyAsync(last, a.last, sizeof(last), hipMemcpyDeviceToHost, h->stream));
    if (int rc__ = atomdev_sync_stream(h)) return rc__;
    if (last[0] != 0.) state[0] = last[1];
    if (last[2] != 0.) state[1] = last[3];
    return 0;
}

extern "C" int atomdev_atm_ice_water_saturation_adjustment(atomdev_t* h, double* q_v_hyp) {
    ATOMDEV_REQUIRE(h && h->model == ATOMDEV_ATM && q_v_hyp, "atomdev_atm_ice_water_saturation_adjustment: bad argument");
    ATOMDEV_REQUIRE(h->im >= 4, "atomdev_atm_ice_water_saturation_adjustment: needs at least 4 levels (scratch layout)");
    ATOMDEV_TRY(hipSetDevice(h->device));
    if (int rc = atomdev_refresh_flags(h)) return rc;
    const atomdev_params& P = h->p;
    AdjArgs a;
    a.c = h->f[ATOMDEV_C]; a.cn = h->f[ATOMDEV_CN]; a.cloud = h->f[ATOMDEV_CLOUD]; a.cloudn = h->f[ATOMDEV_CLOUDN];
    a.ice = h->f[ATOMDEV_ICE]; a.icen = h->f[ATOMDEV_ICEN]; a.t = h->f[ATOMDEV_T]; a.s_c_c = h->f[ATOMDEV_S_C_C];
    a.flags = h->flags;
    a.im = h->im; a.jm = h->jm; a.km = h->km;
    a.t_0 = P.t_0; a.g = P.g; a.gam = P.gam; a.R_Air = P.R_Air; a.r_air = P.r_air; a.R_WaterVapour = P.R_WaterVapour;
    a.L_atm = P.L_atm; a.hp = P.hp; a.ep = P.ep; a.lv = P.lv; a.ls = P.ls; a.cp_l = P.cp_l;
    a.dt_dim = P.L_atm / P.u_0 * P.dt;  // BC_Thermo.cpp:121
    const size_t n2 = h->n2, n3 = h->n3;
    a.nchunks = (int)((n2 + CHUNK - 1) / CHUNK);
    a.state = *q_v_hyp;
    double last[2];
    if (h->im <= MAX_LEVELS) {
        // scratch: the per-point values in the first scratch field, everything else in the second (the sweep's
        // prep kernel rebuilds what it keeps there)
        a.hyp = h->scr[0];
        double* s = h->scr[1];
        h->h0i_valid = false;
        a.out = s; a.carry = s + n2; a.chunk_carry = s + 2 * n2; a.last = a.chunk_carry + a.nchunks;
        a.chunk_unfinished = (int*)(a.last + 2);
        a.status = (uint8_t*)(a.chunk_unfinished + a.nchunks + (a.nchunks & 1));
        a.col_state = a.status + n3; a.carry_state = a.col_state + n2; a.chunk_carry_state = a.carry_state + n2;
        const size_t bytes = (size_t)(a.chunk_carry_state + a.nchunks - (uint8_t*)s);
        ATOMDEV_REQUIRE(bytes <= n3 * sizeof(double), "atomdev_atm_ice_water_saturation_adjustment: grid too small for the scratch layout");
        ATOMDEV_TRY(hipMemsetAsync(a.status, 0, n3 + n2, h->stream));  // no point finished, every column open
        const int COLS = cols_per_group(h->im);
        const size_t lds = ((size_t)h->im * COLS + COLS) * sizeof(double) + (size_t)h->im * COLS;
        size_t unfinished = n2;
        for (size_t pass = 0;; pass++) {
            hipLaunchKernelGGL(k_adj_carry_scan, dim3(a.nchunks), dim3(CHUNK), 0, h->stream, a);
            hipLaunchKernelGGL(k_adj_carry_finish, dim3(1), dim3(64), 0, h->stream, a);
            if (pass > 0) {
                ATOMDEV_TRY(hipGetLastError());
                ATOMDEV_TRY(hipMemcpyAsync(last, a.last, sizeof(last), hipMemcpyDeviceToHost, h->stream));
                if (int rc__ = atomdev_sync_stream(h)) return rc__;
                const size_t left = (size_t)last[1];
                if (left == 0) break;
                // the first unfinished column in visiting order always has what it needs
                ATOMDEV_REQUIRE(left < unfinished, "atomdev_atm_ice_water_saturation_adjustment: a pass finished no column");
                unfinished = left;
            }
            hipLaunchKernelGGL(k_adj_pass, dim3((unsigned)((n2 + COLS - 1) / COLS)), dim3(COLS, h->im), lds, h->stream, a);
        }
    } else {
        // very tall grids: one thread per column, columns that inherit from the column before them one by one
        double* s = h->scr[0];
        a.out = s; a.kind = (uint8_t*)(s + n2); a.last = s + 2 * n2;
        a.chunk_out = a.last + 2; a.chunk_kind = (uint8_t*)(a.chunk_out + a.nchunks);
        ATOMDEV_REQUIRE(2 * n2 + 2 + 2 * (size_t)a.nchunks <= n3, "atomdev_atm_ice_water_saturation_adjustment: grid too small for the scratch layout");
        hipLaunchKernelGGL(k_adj_columns, dim3((unsigned)((n2 + TPB - 1) / TPB)), dim3(TPB), 0, h->stream, a);
        hipLaunchKernelGGL(k_adj_chunk_last, dim3(a.nchunks), dim3(CHUNK), 0, h->stream, a);
        hipLaunchKernelGGL(k_adj_finish, dim3(1), dim3(64), 0, h->stream, a);
        ATOMDEV_TRY(hipGetLastError());
        ATOMDEV_TRY(hipMemcpyAsync(last, a.last, sizeof(last), hipMemcpyDeviceToHost, h->stream));
        if (int rc__ = atomdev_sync_stream(h)) return rc__;
        if (last[1] != 0.) {
            hipLaunchKernelGGL(k_adj_inherit, dim3(1), dim3(1024), 0, h->stream, a);
            ATOMDEV_TRY(hipGetLastError());
            ATOMDEV_TRY(hipMemcpyAsync(last, a.last, sizeof(double), hipMemcpyDeviceToHost, h->stream));
            if (int rc__ = atomdev_sync_stream(h)) return rc__;
        }
    }
    *q_v_hyp = last[0];
    return 0;
}
