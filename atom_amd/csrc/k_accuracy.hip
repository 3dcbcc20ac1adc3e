// Accuracy queries: continuity residuum (Accuracy_Atm::residuumQuery_3D/_2D,
// atmosphere/Accuracy_Atm.cpp:54-116; Accuracy_Hyd::residuumQuery_3D,
// hydrosphere/Accuracy_Hyd.cpp:88-116) and steady-state differences
// (Accuracy_Atm::steadyQuery_3D/_2D, Accuracy_Atm.cpp:118-153, 192-305;
// Accuracy_Hyd::steadyQuery_3D, Accuracy_Hyd.cpp:121-184).
//
// The residuum "minimum" of the reference is an order-dependent scan,
//     if (fabs(res) > min) { min = res; loc = here; }      (>= for the hydrosphere)
// with a SIGNED store and an unsigned compare: after a negative value is stored
// the next point replaces it whatever it is.  It is not a max-reduction, so it is
// evaluated as what it is, a state machine over the lexicographic sequence — but the
// machine can jump:
//   * state m >= 0 at position s: let P be the FIRST position of the maximum of |res|
//     over [s, n).  If that maximum does not beat m, nothing in the suffix does and the
//     scan is over.  Otherwise P replaces whatever state reaches it (every earlier
//     point of the suffix is strictly smaller in magnitude, a negative state is always
//     replaced), so the points in between do not matter: continue at P+1 with state res[P].
//   * state m < 0 at position s: every point replaces it until the first non-negative
//     one, q; continue at q+1 with state res[q] (if there is none, the last point wins).
// Each jump is two block-wide searches over a per-chunk summary table plus one partial
// chunk; the number of jumps is the number of sign changes along the chain of suffix
// maxima, a handful on any field (logarithmic in expectation on random data).
// Steps: (1) every point's residuum in parallel; (2) per chunk, in parallel: first
// position of the maximum magnitude and first non-negative position; (3) one workgroup
// runs the jumps.  NaNs never compare true: they neither replace nor are they stored.
#include <math.h>

#include "atomdev_internal.h"

namespace {

constexpr int TPB = 256;
constexpr int CHUNK = 2048;    // sequence elements per chunk

struct ChunkSum {
    double amax;               // maximum |res| of the chunk (-1: none, all NaN)
    int pmax;                  // first position (sequence index) attaining it
    int pnn;                   // first position with res >= 0 (-1: none)
};

template <bool GE>
__device__ __forceinline__ bool beats(double r, double m) { return GE ? (fabs(r) >= m) : (fabs(r) > m); }

// residuum at every interior point, stored in the field (full layout) and in the
// compact lexicographic sequence seq[(i-1)(jm-2)(km-2) + (j-1)(km-2) + (k-1)]
__global__ void k_residuum3d(const double* __restrict__ u, const double* __restrict__ v, const double* __restrict__ w,
                             double* __restrict__ field, double* __restrict__ seq, const double* __restrict__ rad,
                             const double* __restrict__ sint, const double* __restrict__ cost, DevConst K) {
    const int jm = K.jm, km = K.km;
    const int k = 1 + blockIdx.x * blockDim.x + threadIdx.x;
    const int j = 1 + blockIdx.y, i = 1 + blockIdx.z;
    if (k > km - 2) return;
    const size_t row = km, lvl = (size_t)jm * km;
    const size_t p = idx3(jm, km, i, j, k);
    const double sinthe = sint[j], costhe = cost[j];
    const double rmsinthe = rad[i] * sinthe;
    double dudr = (u[p + lvl] - u[p - lvl]) / (2. * K.dr);
    double dvdthe = (v[p + row] - v[p - row]) / (2. * K.dthe);
    double dwdphi = (w[p + 1] - w[p - 1]) / (2. * K.dphi);
    double res = dudr + 2. * u[p] / rad[i] + dvdthe / rad[i] + costhe / rmsinthe * v[p] + dwdphi / rmsinthe;
    field[p] = res;
    if (seq) seq[((size_t)(i - 1) * (jm - 2) + (j - 1)) * (km - 2) + (k - 1)] = res;
}

// surface form, level 0 (Accuracy_Atm.cpp:58-78)
__global__ void k_residuum2d(const double* __restrict__ v, const double* __restrict__ w, double* __restrict__ field,
                             double* __restrict__ seq, const double* __restrict__ rad, const double* __restrict__ sint,
                             const double* __restrict__ cost, DevConst K) {
    const int km = K.km, jm = K.jm;
    const int k = 1 + blockIdx.x * blockDim.x + threadIdx.x;
    const int j = 1 + blockIdx.y;
    if (k > km - 2) return;
    const size_t row = km;
    const size_t p = (size_t)j * km + k;
    const double sinthe = sint[j], costhe = cost[j];
    const double rmsinthe = rad[0] * sinthe;
    double dvdthe = (v[p + row] - v[p - row]) / (2. * K.dthe);
    double dwdphi = (w[p + 1] - w[p - 1]) / (2. * K.dphi);
    double res = dvdthe / rad[0] + costhe / rmsinthe * v[p] + dwdphi / rmsinthe;
    field[p] = res;
    seq[(size_t)(j - 1) * (km - 2) + (k - 1)] = res;
    (void)jm;
}

// first-argmax / first-position combiners
__device__ __forceinline__ void take_max(double& v, long& p, double v2, long p2) {
    if (v2 > v || (v2 == v && p2 >= 0 && (p < 0 || p2 < p))) { v = v2; p = p2; }
}
__device__ __forceinline__ void take_first(long& p, long p2) {
    if (p2 >= 0 && (p < 0 || p2 < p)) p = p2;
}

__global__ __launch_bounds__(TPB) void k_chunk_summary(const double* __restrict__ seq, size_t n, ChunkSum* __restrict__ sum) {
    __shared__ double sv[TPB];
    __shared__ long sp[TPB], sq[TPB];
    const size_t b = (size_t)blockIdx.x * CHUNK;
    const size_t e = (b + CHUNK < n) ? b + CHUNK : n;
    double v = -1.; long p = -1, q = -1;
    for (size_t t = b + threadIdx.x; t < e; t += TPB) {
        const double x = seq[t];
        take_max(v, p, fabs(x), (long)t);  // NaN: both compares false
        if (x >= 0.) take_first(q, (long)t);
    }
    sv[threadIdx.x] = v; sp[threadIdx.x] = p; sq[threadIdx.x] = q;
    __syncthreads();
    for (int st = TPB / 2; st > 0; st >>= 1) {
        if (threadIdx.x < st) {
            take_max(sv[threadIdx.x], sp[threadIdx.x], sv[threadIdx.x + st], sp[threadIdx.x + st]);
            take_first(sq[threadIdx.x], sq[threadIdx.x + st]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { sum[blockIdx.x].amax = sv[0]; sum[blockIdx.x].pmax = (int)sp[0]; sum[blockIdx.x].pnn = (int)sq[0]; }
}

// One workgroup: the jumps described at the top of the file.
template <bool GE>
__global__ __launch_bounds__(TPB) void k_scan_jumps(const double* __restrict__ seq, size_t n, const ChunkSum* __restrict__ sum,
                                                    int nchunks, double m_in, double* __restrict__ out /* {min, pos} */) {
    __shared__ double sv[TPB];
    __shared__ long sp[TPB], sq[TPB];
    double m = m_in;
    long pos = -1;  // -1: the carried-in state survives, the caller's location stays
    long s = 0;
    const long N = (long)n;
    while (s < N) {
        // suffix [s, N): the rest of the chunk s lies in, element by element, then whole chunks from the table
        const long c0 = s / CHUNK;
        const long e0 = ((c0 + 1) * CHUNK < N) ? (c0 + 1) * CHUNK : N;
        double v = -1.; long p = -1, q = -1;
        for (long t = s + threadIdx.x; t < e0; t += TPB) {
            const double x = seq[t];
            take_max(v, p, fabs(x), t);
            if (x >= 0.) take_first(q, t);
        }
        for (long c = c0 + 1 + threadIdx.x; c < nchunks; c += TPB) {
            take_max(v, p, sum[c].amax, (long)sum[c].pmax);
            take_first(q, (long)sum[c].pnn);
        }
        sv[threadIdx.x] = v; sp[threadIdx.x] = p; sq[threadIdx.x] = q;
        __syncthreads();
        for (int st = TPB / 2; st > 0; st >>= 1) {
            if (threadIdx.x < st) {
                take_max(sv[threadIdx.x], sp[threadIdx.x], sv[threadIdx.x + st], sp[threadIdx.x + st]);
                take_first(sq[threadIdx.x], sq[threadIdx.x + st]);
            }
            __syncthreads();
        }
        const double g = sv[0]; const long P = sp[0], Q = sq[0];
        __syncthreads();  // everyone has read the results before the next round overwrites them
        if (m < 0.) {
            // every (non-NaN) point replaces a negative state, up to and including the first non-negative one
            if (Q >= 0) { m = seq[Q]; pos = Q; s = Q + 1; continue; }
            long t = N - 1;  // no non-negative point left: the last non-NaN point of the suffix wins
            while (t >= s && !(fabs(seq[t]) >= 0.)) t--;
            if (t >= s) { m = seq[t]; pos = t; }
            break;
        }
        if (P < 0 || !(GE ? (g >= m) : (g > m))) break;  // nothing in the suffix beats m (also: m is NaN)
        m = seq[P]; pos = P; s = P + 1;
    }
    if (threadIdx.x == 0) { out[0] = m; out[1] = (double)pos; }
}

// ---- steady queries: max |x - xn| and its first (atm, strict >) or last (hyd, >=) location ----

struct SteadyArgs { const double* x[9]; const double* xn[9]; int n; };
struct Best { double v; long long idx; };

template <bool LAST>
__device__ __forceinline__ void take(Best& a, double v, long long idx) {
    if (v > a.v || (v == a.v && (LAST ? idx > a.idx : idx < a.idx))) { a.v = v; a.idx = idx; }
}

template <bool LAST>
__global__ void k_steady_partial(SteadyArgs s, size_t off, size_t n, Best* __restrict__ part) {
    __shared__ Best sh[TPB];
    const int f = blockIdx.y;  // one field per grid row: all fields stream concurrently
    const double* __restrict__ x = s.x[f] + off;
    const double* __restrict__ xn = s.xn[f] + off;
    Best b; b.v = -1.; b.idx = LAST ? -1 : (long long)n;
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (size_t)gridDim.x * blockDim.x) {
        double d = fabs(x[q] - xn[q]);
        take<LAST>(b, d, (long long)q);
    }
    sh[threadIdx.x] = b;
    __syncthreads();
    for (int st = TPB / 2; st > 0; st >>= 1) {
        if (threadIdx.x < st) take<LAST>(sh[threadIdx.x], sh[threadIdx.x + st].v, sh[threadIdx.x + st].idx);
        __syncthreads();
    }
    if (threadIdx.x == 0) part[(size_t)f * gridDim.x + blockIdx.x] = sh[0];
}

template <bool LAST>
__global__ void k_steady_final(const Best* __restrict__ part, int nblocks, int nf, Best* __restrict__ out) {
    __shared__ Best sh[TPB];
    const int f = blockIdx.x;  // one workgroup per field
    Best b; b.v = -1.; b.idx = LAST ? -1 : (1LL << 62);
    for (int i = threadIdx.x; i < nblocks; i += blockDim.x) take<LAST>(b, part[(size_t)f * nblocks + i].v, part[(size_t)f * nblocks + i].idx);
    sh[threadIdx.x] = b;
    __syncthreads();
    for (int st = TPB / 2; st > 0; st >>= 1) {
        if (threadIdx.x < st) take<LAST>(sh[threadIdx.x], sh[threadIdx.x + st].v, sh[threadIdx.x + st].idx);
        __syncthreads();
    }
    if (threadIdx.x == 0) out[f] = sh[0];
    (void)nf;
}

// Scratch of the residuum query that no call allocates: one array per GPU in the library's own device
// image, for callers that must not allocate (atomdev_atm_residuum3d_noalloc).  It holds the compact
// residuum sequence, the per-chunk summaries and the two result words of one query of up to
// STATIC_SEQ_MAX interior points (the 41 x 181 x 361 grid has 2 506 179).  Every query writes what it
// reads before it reads it, so nothing has to start at zero; two queries that overlap in time on one
// GPU would share the array and race.
constexpr size_t STATIC_SEQ_MAX = 2621440;   // 20 MiB of doubles
constexpr size_t STATIC_SCRATCH_BYTES = STATIC_SEQ_MAX * sizeof(double) + (STATIC_SEQ_MAX / CHUNK + 1) * sizeof(ChunkSum) + 256;
static __device__ __align__(256) unsigned char scratch[STATIC_SCRATCH_BYTES];

// where a query keeps its intermediate results: the handle's reduction buffer (grown on demand) or the static array
struct ResScratch { void* base; size_t bytes; double* host_result; };

int residuum(atomdev* h, bool three_d, bool ge, double* min, int loc[3], const ResScratch* fixed = nullptr) {
    const int im = h->im, jm = h->jm, km = h->km;
    const size_t n = three_d ? (size_t)(im - 2) * (jm - 2) * (km - 2) : (size_t)(jm - 2) * (km - 2);
    const int nchunks = (int)((n + CHUNK - 1) / CHUNK);
    // scratch: seq | chunk functions | result
    const size_t need = ((n * sizeof(double) + 63) / 64) * 64 + (size_t)nchunks * sizeof(ChunkSum) + 128;
    void* base; size_t bytes;
    if (fixed) {
        base = fixed->base; bytes = fixed->bytes;
        if (need > bytes) {
            atomdev_set_error("residuum query without allocation: the grid has " + std::to_string(n) +
                              " interior points, the static scratch holds at most " + std::to_string(STATIC_SEQ_MAX));
            return 1;
        }
    } else {
        if (need > h->red_bytes) {
            if (int rc__ = atomdev_sync_stream(h)) return rc__;
            ATOMDEV_TRY(hipFree(h->red));
            h->red = nullptr; h->red_bytes = 0;
            ATOMDEV_TRY(hipMalloc(&h->red, need));
            h->red_bytes = need;
        }
        base = h->red; bytes = h->red_bytes;
    }
    double* seq = (double*)base;
    ChunkSum* fn = (ChunkSum*)((char*)base + ((n * sizeof(double) + 63) / 64) * 64);
    double* out = (double*)((char*)fn + (size_t)nchunks * sizeof(ChunkSum));
    if ((char*)(out + 2) > (char*)base + bytes) { atomdev_set_error("residuum: scratch layout"); return 1; }
    if (three_d) {
        dim3 grid((km - 2 + 127) / 128, jm - 2, im - 2);
        hipLaunchKernelGGL(k_residuum3d, grid, dim3(128), 0, h->stream, h->f[ATOMDEV_U], h->f[ATOMDEV_V], h->f[ATOMDEV_W],
                           h->f[ATOMDEV_RESIDUUM], seq, h->rad, h->sinthe, h->costhe, h->k);
    } else {
        dim3 grid((km - 2 + 127) / 128, jm - 2);
        hipLaunchKernelGGL(k_residuum2d, grid, dim3(128), 0, h->stream, h->f[ATOMDEV_V], h->f[ATOMDEV_W],
                           h->f[ATOMDEV_RESIDUUM], seq, h->rad, h->sinthe, h->costhe, h->k);
    }
    ATOMDEV_TRY(hipGetLastError());
    const double m_in = ge ? 0. : *min;  // the hydrosphere object resets its state (Accuracy_Hyd.cpp:91)
    hipLaunchKernelGGL(k_chunk_summary, dim3(nchunks), dim3(TPB), 0, h->stream, seq, n, fn);
    if (ge) hipLaunchKernelGGL(k_scan_jumps<true>, dim3(1), dim3(TPB), 0, h->stream, seq, n, fn, nchunks, m_in, out);
    else hipLaunchKernelGGL(k_scan_jumps<false>, dim3(1), dim3(TPB), 0, h->stream, seq, n, fn, nchunks, m_in, out);
    ATOMDEV_TRY(hipGetLastError());
    double stack_res[2];
    double* res = fixed ? fixed->host_result : stack_res;   // pinned memory of the handle / pageable
    ATOMDEV_TRY(hipMemcpyAsync(res, out, 2 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    if (int rc__ = atomdev_sync_stream(h)) return rc__;
    *min = res[0];
    long pos = (long)res[1];
    if (pos >= 0) {
        const long nk = km - 2, nj = jm - 2;
        if (three_d) { loc[0] = (int)(pos / (nj * nk)) + 1; loc[1] = (int)((pos / nk) % nj) + 1; loc[2] = (int)(pos % nk) + 1; }
        else { loc[0] = 0; loc[1] = (int)(pos / nk) + 1; loc[2] = (int)(pos % nk) + 1; }
    }
    return 0;
}

int steady(atomdev* h, const int* ids, int nf, size_t off, size_t n, bool last, double* outv, int* loc) {
    const int nblocks = 512;
    const size_t need = ((size_t)nf * nblocks + 16) * sizeof(Best);
    if (need > h->red_bytes) { atomdev_set_error("steady: scratch too small"); return 1; }
    Best* part = (Best*)h->red;
    Best* fin = part + (size_t)nf * nblocks;
    SteadyArgs s; s.n = nf;
    for (int f = 0; f < nf; f++) { s.x[f] = h->f[ids[f]]; s.xn[f] = h->f[ids[f] + 9]; }
    if (last) {
        hipLaunchKernelGGL(k_steady_partial<true>, dim3(nblocks, nf), dim3(TPB), 0, h->stream, s, off, n, part);
        hipLaunchKernelGGL(k_steady_final<true>, dim3(nf), dim3(TPB), 0, h->stream, part, nblocks, nf, fin);
    } else {
        hipLaunchKernelGGL(k_steady_partial<false>, dim3(nblocks, nf), dim3(TPB), 0, h->stream, s, off, n, part);
        hipLaunchKernelGGL(k_steady_final<false>, dim3(nf), dim3(TPB), 0, h->stream, part, nblocks, nf, fin);
    }
    ATOMDEV_TRY(hipGetLastError());
    Best res[9];
    ATOMDEV_TRY(hipMemcpyAsync(res, fin, nf * sizeof(Best), hipMemcpyDeviceToHost, h->stream));
    if (int rc__ = atomdev_sync_stream(h)) return rc__;
    const long long n2 = (long long)h->n2;
    for (int f = 0; f < nf; f++) {
        double v = res[f].v;
        long long q = res[f].idx + (long long)off;
        if (v < 0. || (!last && v == 0.)) { v = 0.; q = 0; }  // nothing exceeded the initial 0: location stays (0, 0, 0)
        outv[f] = v;
        loc[3 * f] = (int)(q / n2); loc[3 * f + 1] = (int)((q % n2) / h->km); loc[3 * f + 2] = (int)(q % h->km);
    }
    return 0;
}

// ---- extreme values and where the reference's scans find them ----
// MinMax_Atm::searchMinMax_3D / _2D (atmosphere/MinMax_Atm.cpp:74-196) and MinMax_Hyd::searchMinMax_3D / _2D
// (hydrosphere/MinMax_Hyd.cpp:51-260) walk the field in (j, k, i) order with strict comparisons, so each result
// is the largest / smallest value at its first position in that order; what differs between the four is the
// seed of the scan, applied on the host afterwards.  key = rank of a point in the (j, k, i) walk.
struct Ext { double vmax; long long kmax; double vmin; long long kmin; };

__device__ __forceinline__ void ext_take(Ext& a, double vmax, long long kmax, double vmin, long long kmin) {
    if (vmax > a.vmax || (vmax == a.vmax && kmax < a.kmax)) { a.vmax = vmax; a.kmax = kmax; }
    if (vmin < a.vmin || (vmin == a.vmin && kmin < a.kmin)) { a.vmin = vmin; a.kmin = kmin; }
}

__device__ __forceinline__ void ext_block_reduce(Ext& b, Ext* sh) {
    sh[threadIdx.x] = b;
    __syncthreads();
    for (int st = TPB / 2; st > 0; st >>= 1) {
        if (threadIdx.x < st) {
            const Ext o = sh[threadIdx.x + st];
            ext_take(sh[threadIdx.x], o.vmax, o.kmax, o.vmin, o.kmin);
        }
        __syncthreads();
    }
    b = sh[0];
}

// two_d: interior points of a jm x km surface (im = 1); skip_zero_min: zeros do not count for the minimum
__global__ void k_minmax_partial(const double* __restrict__ x, int im, int jm, int km, int two_d, int skip_zero_min,
                                 Ext* __restrict__ part) {
    __shared__ Ext sh[TPB];
    const long long none = 1LL << 62;
    Ext b = { -INFINITY, none, INFINITY, none };
    const size_t n = (size_t)im * jm * km;
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (size_t)gridDim.x * blockDim.x) {
        const int k = (int)(q % km), j = (int)((q / km) % jm), i = (int)(q / ((size_t)km * jm));
        if (two_d && (j < 1 || j > jm - 2 || k < 1 || k > km - 2)) continue;
        const long long key = ((long long)j * km + k) * im + i;
        const double v = x[q];
        if (v > b.vmax || (v == b.vmax && key < b.kmax)) { b.vmax = v; b.kmax = key; }
        if (!(skip_zero_min && v == 0.) && (v < b.vmin || (v == b.vmin && key < b.kmin))) { b.vmin = v; b.kmin = key; }
    }
    ext_block_reduce(b, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = b;
}

__global__ void k_minmax_final(const Ext* __restrict__ part, int nblocks, const double* __restrict__ x, Ext* __restrict__ out,
                               double* __restrict__ first_value) {
    __shared__ Ext sh[TPB];
    const long long none = 1LL << 62;
    Ext b = { -INFINITY, none, INFINITY, none };
    for (int i = threadIdx.x; i < nblocks; i += blockDim.x) ext_take(b, part[i].vmax, part[i].kmax, part[i].vmin, part[i].kmin);
    ext_block_reduce(b, sh);
    if (threadIdx.x == 0) { *out = b; *first_value = x[0]; }
}

int minmax(atomdev* h, const double* x, bool two_d, double out[2], int loc[6]) {
    const bool hyd = h->model == ATOMDEV_HYD;
    const int nblocks = 512;
    Ext* part = (Ext*)h->red;
    Ext* fin = part + nblocks;
    double* first = (double*)(fin + 1);
    if ((size_t)((char*)(first + 1) - (char*)h->red) > h->red_bytes) { atomdev_set_error("minmax: scratch too small"); return 1; }
    const int im = two_d ? 1 : h->im;
    hipLaunchKernelGGL(k_minmax_partial, dim3(nblocks), dim3(TPB), 0, h->stream, x, im, h->jm, h->km, two_d ? 1 : 0,
                       (hyd && two_d) ? 1 : 0, part);
    hipLaunchKernelGGL(k_minmax_final, dim3(1), dim3(TPB), 0, h->stream, part, nblocks, x, fin, first);
    ATOMDEV_TRY(hipGetLastError());
    struct { Ext e; double first; } r;
    ATOMDEV_TRY(hipMemcpyAsync(&r, fin, sizeof(r), hipMemcpyDeviceToHost, h->stream));
    if (int rc__ = atomdev_sync_stream(h)) return rc__;
    const long long none = 1LL << 62;
    // seeds: MinMax_Atm starts both scans from the first element (which the 2-D scan, interior only, never visits);
    // MinMax_Hyd starts the maximum from 0 and the minimum from the maximum it found
    double vmax = hyd ? 0. : r.first, vmin;
    long long kmax = 0, kmin = 0;
    if (r.e.kmax != none && r.e.vmax > vmax) { vmax = r.e.vmax; kmax = r.e.kmax; }
    vmin = hyd ? vmax : r.first;
    if (r.e.kmin != none && r.e.vmin < vmin) { vmin = r.e.vmin; kmin = r.e.kmin; }
    if (hyd && two_d && vmin < 0.) vmin = 0.;  // MinMax_Hyd.cpp:258
    out[0] = vmax; out[1] = vmin;
    const long long keys[2] = { kmax, kmin };
    for (int n = 0; n < 2; n++) {
        const long long key = keys[n];
        loc[3 * n] = (int)(key % im); loc[3 * n + 1] = (int)(key / ((long long)h->km * im)); loc[3 * n + 2] = (int)((key / im) % h->km);
    }
    return 0;
}

}  // namespace

extern "C" int atomdev_minmax3d(atomdev_t* h, int field3d, double out[2], int loc[6]) {
    ATOMDEV_REQUIRE(h && out && loc && field3d >= 0 && field3d < ATOMDEV_N3D && h->f[field3d], "atomdev_minmax3d: bad argument");
    ATOMDEV_TRY(hipSetDevice(h->device));
    return minmax(h, h->f[field3d], false, out, loc);
}
extern "C" int atomdev_minmax2d(atomdev_t* h, int field2d, double out[2], int loc[6]) {
    ATOMDEV_REQUIRE(h && out && loc && field2d >= 0 && field2d < ATOMDEV_N2D, "atomdev_minmax2d: bad argument");
    ATOMDEV_TRY(hipSetDevice(h->device));
    return minmax(h, h->g[field2d], true, out, loc);
}

extern "C" int atomdev_atm_residuum3d(atomdev_t* h, double* min, int loc[3]) {
    ATOMDEV_REQUIRE(h && h->model == ATOMDEV_ATM && min && loc, "atomdev_atm_residuum3d: bad argument");
    ATOMDEV_TRY(hipSetDevice(h->device));
    return residuum(h, true, false, min, loc);
}
// The address of this GPU's copy of the static scratch, looked up once per handle at set-up with the handle's
// device current (atomdev_create).  Host code must use this pointer: the array's name is a host-side stand-in.
int atomdev_static_scratch_init(atomdev* h) {
    void* p = nullptr;
    ATOMDEV_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
    ATOMDEV_REQUIRE(p != nullptr, "atomdev_create: the static scratch of this device has no address");
    h->static_scratch = p;
    h->static_scratch_bytes = STATIC_SCRATCH_BYTES;
    return 0;
}

extern "C" int atomdev_atm_residuum3d_noalloc(atomdev_t* h, double* min, int loc[3]) {
    ATOMDEV_REQUIRE(h && h->model == ATOMDEV_ATM && min && loc, "atomdev_atm_residuum3d_noalloc: bad argument");
    ATOMDEV_REQUIRE(h->static_scratch && h->pinned, "atomdev_atm_residuum3d_noalloc: the handle has no static scratch");
    ATOMDEV_TRY(hipSetDevice(h->device));
    const ResScratch fixed = { h->static_scratch, h->static_scratch_bytes, (double*)h->pinned };
    return residuum(h, true, false, min, loc, &fixed);
}

extern "C" int atomdev_atm_residuum2d(atomdev_t* h, double* min, int loc[3]) {
    ATOMDEV_REQUIRE(h && h->model == ATOMDEV_ATM && min && loc, "atomdev_atm_residuum2d: bad argument");
    ATOMDEV_TRY(hipSetDevice(h->device));
    return residuum(h, false, false, min, loc);
}
extern "C" int atomdev_hyd_residuum3d(atomdev_t* h, double* min, int loc[3]) {
    ATOMDEV_REQUIRE(h && h->model == ATOMDEV_HYD && min && loc, "atomdev_hyd_residuum3d: bad argument");
    ATOMDEV_TRY(hipSetDevice(h->device));
    return residuum(h, true, true, min, loc);
}
// ---- the queries of a model that holds a slab of longitudes of a wider grid ----
// The residuum scan runs over the lexicographic sequence of the WHOLE grid, in which the slabs' points interleave row by
// row; what a slab can contribute to one jump of the scan (see the top of the file) is a summary of its own points of the
// suffix: the maximum magnitude with its first position, the first non-negative point, the last point that is no NaN, each
// with the value there.  The caller combines the slabs' summaries and repeats for the next jump.

namespace {

struct Suffix {
    double amax, pmax, xmax;   // largest |res|, its first position in the global sequence, res there (amax -1: none)
    double pnn, xnn;           // first position with res >= 0 (-1: none), res there
    double plast, xlast;       // last position whose res is no NaN (-1: none), res there
};
__device__ __forceinline__ void merge(Suffix& a, const Suffix& b) {
    if (b.amax > a.amax || (b.amax == a.amax && b.pmax >= 0. && (a.pmax < 0. || b.pmax < a.pmax))) { a.amax = b.amax; a.pmax = b.pmax; a.xmax = b.xmax; }
    if (b.pnn >= 0. && (a.pnn < 0. || b.pnn < a.pnn)) { a.pnn = b.pnn; a.xnn = b.xnn; }
    if (b.plast > a.plast) { a.plast = b.plast; a.xlast = b.xlast; }
}

__global__ __launch_bounds__(TPB) void k_suffix_partial(const double* __restrict__ field, int im, int jm, int km, int k0, int nk, int kg0,
                                                        int kmg, double from, Suffix* __restrict__ part) {
    __shared__ Suffix sh[TPB];
    Suffix b = { -1., -1., 0., -1., 0., -1., 0. };
    const size_t n = (size_t)(im - 2) * (jm - 2) * nk;
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (size_t)gridDim.x * blockDim.x) {
        const int kk = (int)(q % nk);
        const size_t row = q / nk;  // (i-1)(jm-2) + (j-1)
        const int kg = kg0 + kk;
        if (kg < 1 || kg > kmg - 2) continue;
        const double t = (double)row * (double)(kmg - 2) + (double)(kg - 1);
        if (t < from) continue;
        const int i = (int)(row / (jm - 2)) + 1, j = (int)(row % (jm - 2)) + 1;
        const double x = field[idx3(jm, km, i, j, k0 + kk)];
        Suffix c = { -1., -1., 0., -1., 0., -1., 0. };
        if (fabs(x) >= 0.) { c.amax = fabs(x); c.pmax = t; c.xmax = x; c.plast = t; c.xlast = x; }
        if (x >= 0.) { c.pnn = t; c.xnn = x; }
        merge(b, c);
    }
    sh[threadIdx.x] = b;
    __syncthreads();
    for (int st = TPB / 2; st > 0; st >>= 1) {
        if (threadIdx.x < st) merge(sh[threadIdx.x], sh[threadIdx.x + st]);
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = sh[0];
}
__global__ __launch_bounds__(TPB) void k_suffix_final(const Suffix* __restrict__ part, int nblocks, Suffix* __restrict__ out) {
    __shared__ Suffix sh[TPB];
    Suffix b = { -1., -1., 0., -1., 0., -1., 0. };
    for (int n = threadIdx.x; n < nblocks; n += blockDim.x) merge(b, part[n]);
    sh[threadIdx.x] = b;
    __syncthreads();
    for (int st = TPB / 2; st > 0; st >>= 1) {
        if (threadIdx.x < st) merge(sh[threadIdx.x], sh[threadIdx.x + st]);
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = sh[0];
}

struct PartArgs { const double* x[9]; const double* xn[9]; };
template <bool LAST>
__global__ __launch_bounds__(TPB) void k_steady_part(PartArgs s, int im, int jm, int km, int k0, int nk, int kg0, int kmg, Best* __restrict__ part) {
    __shared__ Best sh[TPB];
    const int f = blockIdx.y;
    Best b; b.v = -1.; b.idx = LAST ? -1 : (1LL << 62);
    const size_t n = (size_t)im * jm * nk;
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (size_t)gridDim.x * blockDim.x) {
        const int kk = (int)(q % nk);
        const size_t row = q / nk;  // i jm + j
        const size_t p = row * km + k0 + kk;
        take<LAST>(b, fabs(s.x[f][p] - s.xn[f][p]), (long long)(row * kmg + kg0 + kk));
    }
    sh[threadIdx.x] = b;
    __syncthreads();
    for (int st = TPB / 2; st > 0; st >>= 1) {
        if (threadIdx.x < st) take<LAST>(sh[threadIdx.x], sh[threadIdx.x + st].v, sh[threadIdx.x + st].idx);
        __syncthreads();
    }
    if (threadIdx.x == 0) part[(size_t)f * gridDim.x + blockIdx.x] = sh[0];
}

bool slab_ok(const atomdev* h, int k0, int nk, int kg0, int kmg) {
    return k0 >= 0 && nk >= 1 && k0 + nk <= h->km && kg0 >= 0 && kg0 + nk <= kmg;
}

}  // namespace

extern "C" int atomdev_residuum3d_field(atomdev_t* h) {
    ATOMDEV_REQUIRE(h, "atomdev_residuum3d_field: null handle");
    ATOMDEV_TRY(hipSetDevice(h->device));
    dim3 grid((h->km - 2 + 127) / 128, h->jm - 2, h->im - 2);
    hipLaunchKernelGGL(k_residuum3d, grid, dim3(128), 0, h->stream, h->f[ATOMDEV_U], h->f[ATOMDEV_V], h->f[ATOMDEV_W],
                       h->f[ATOMDEV_RESIDUUM], (double*)nullptr, h->rad, h->sinthe, h->costhe, h->k);
    ATOMDEV_TRY(hipGetLastError());
    return 0;
}

extern "C" int atomdev_residuum3d_suffix(atomdev_t* h, int k0, int nk, int kg0, int km_global, double from, double out[7]) {
    ATOMDEV_REQUIRE(h && out && slab_ok(h, k0, nk, kg0, km_global), "atomdev_residuum3d_suffix: bad argument");
    ATOMDEV_TRY(hipSetDevice(h->device));
    const int nblocks = 1024;
    ATOMDEV_REQUIRE((nblocks + 1) * sizeof(Suffix) <= h->red_bytes, "atomdev_residuum3d_suffix: scratch too small");
    Suffix* part = (Suffix*)h->red;
    hipLaunchKernelGGL(k_suffix_partial, dim3(nblocks), dim3(TPB), 0, h->stream, h->f[ATOMDEV_RESIDUUM], h->im, h->jm, h->km, k0, nk, kg0,
                       km_global, from, part);
    hipLaunchKernelGGL(k_suffix_final, dim3(1), dim3(TPB), 0, h->stream, part, nblocks, part + nblocks);
    ATOMDEV_TRY(hipGetLastError());
    Suffix r;
    ATOMDEV_TRY(hipMemcpyAsync(&r, part + nblocks, sizeof r, hipMemcpyDeviceToHost, h->stream));
    if (int rc__ = atomdev_sync_stream(h)) return rc__;
    out[0] = r.amax; out[1] = r.pmax; out[2] = r.xmax; out[3] = r.pnn; out[4] = r.xnn; out[5] = r.plast; out[6] = r.xlast;
    return 0;
}

extern "C" int atomdev_steady3d_part(atomdev_t* h, int k0, int nk, int kg0, int km_global, double* values, double* positions) {
    ATOMDEV_REQUIRE(h && values && positions && slab_ok(h, k0, nk, kg0, km_global), "atomdev_steady3d_part: bad argument");
    ATOMDEV_TRY(hipSetDevice(h->device));
    static const int FA[9] = { ATOMDEV_U, ATOMDEV_V, ATOMDEV_W, ATOMDEV_T, ATOMDEV_C, ATOMDEV_CLOUD, ATOMDEV_ICE, ATOMDEV_CO2, ATOMDEV_P_DYN };
    static const int FH[6] = { ATOMDEV_U, ATOMDEV_V, ATOMDEV_W, ATOMDEV_T, ATOMDEV_C, ATOMDEV_P_DYN };
    const bool atm = h->model == ATOMDEV_ATM;
    const int nf = atm ? 9 : 6;
    const int* F = atm ? FA : FH;
    const int nblocks = 256;
    Best* part = (Best*)h->red;
    ATOMDEV_REQUIRE(((size_t)nf * nblocks + 16) * sizeof(Best) <= h->red_bytes, "atomdev_steady3d_part: scratch too small");
    Best* fin = part + (size_t)nf * nblocks;
    PartArgs s;
    for (int f = 0; f < nf; f++) { s.x[f] = h->f[F[f]]; s.xn[f] = h->f[F[f] + 9]; }
    if (atm) {
        hipLaunchKernelGGL(k_steady_part<false>, dim3(nblocks, nf), dim3(TPB), 0, h->stream, s, h->im, h->jm, h->km, k0, nk, kg0, km_global, part);
        hipLaunchKernelGGL(k_steady_final<false>, dim3(nf), dim3(TPB), 0, h->stream, part, nblocks, nf, fin);
    } else {
        hipLaunchKernelGGL(k_steady_part<true>, dim3(nblocks, nf), dim3(TPB), 0, h->stream, s, h->im, h->jm, h->km, k0, nk, kg0, km_global, part);
        hipLaunchKernelGGL(k_steady_final<true>, dim3(nf), dim3(TPB), 0, h->stream, part, nblocks, nf, fin);
    }
    ATOMDEV_TRY(hipGetLastError());
    Best res[9];
    ATOMDEV_TRY(hipMemcpyAsync(res, fin, nf * sizeof(Best), hipMemcpyDeviceToHost, h->stream));
    if (int rc__ = atomdev_sync_stream(h)) return rc__;
    for (int f = 0; f < nf; f++) { values[f] = res[f].v; positions[f] = (double)res[f].idx; }
    return 0;
}

extern "C" int atomdev_atm_steady3d(atomdev_t* h, double out[9], int loc[27]) {
    ATOMDEV_REQUIRE(h && h->model == ATOMDEV_ATM && out && loc, "atomdev_atm_steady3d: bad argument");
    ATOMDEV_TRY(hipSetDevice(h->device));
    static const int F[9] = { ATOMDEV_U, ATOMDEV_V, ATOMDEV_W, ATOMDEV_T, ATOMDEV_C, ATOMDEV_CLOUD, ATOMDEV_ICE, ATOMDEV_CO2, ATOMDEV_P_DYN };
    return steady(h, F, 9, 0, h->n3, false, out, loc);
}
extern "C" int atomdev_atm_steady2d(atomdev_t* h, double out[3], int loc[9]) {
    ATOMDEV_REQUIRE(h && h->model == ATOMDEV_ATM && out && loc, "atomdev_atm_steady2d: bad argument");
    ATOMDEV_TRY(hipSetDevice(h->device));
    static const int F[3] = { ATOMDEV_V, ATOMDEV_W, ATOMDEV_P_DYN };
    return steady(h, F, 3, 0, h->n2, false, out, loc);
}
extern "C" int atomdev_hyd_steady3d(atomdev_t* h, double out[6], int loc[18]) {
    ATOMDEV_REQUIRE(h && h->model == ATOMDEV_HYD && out && loc, "atomdev_hyd_steady3d: bad argument");
    ATOMDEV_TRY(hipSetDevice(h->device));
    static const int F[6] = { ATOMDEV_U, ATOMDEV_V, ATOMDEV_W, ATOMDEV_T, ATOMDEV_C, ATOMDEV_P_DYN };
    return steady(h, F, 6, 0, h->n3, true, out, loc);
}
