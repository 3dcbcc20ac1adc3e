// Accuracy queries: continuity residuum (Accuracy_Atm::residuumQuery_3D/_2D,
// atmosphere/Accuracy_Atm.cpp:54-116; Accuracy_Hyd::residuumQuery_3D,
// hydrosphere/Accuracy_Hyd.cpp:88-116) and steady-state differences
// (Accuracy_Atm::steadyQuery_3D/_2D, Accuracy_Atm.cpp:118-153, 192-305;
// Accuracy_Hyd::steadyQuery_3D, Accuracy_Hyd.cpp:121-184).
//
// The residuum "minimum" of the reference is an order-dependent scan,
//     if (fabs(res) > min) { min = res; loc = here; }      (>= for the hydrosphere)
// with a SIGNED store and an unsigned compare: after a negative value is stored
// every following point replaces it.  It is not a max-reduction, so it is
// evaluated as what it is, a state machine over the lexicographic sequence, in
// three steps: (1) every point's residuum in parallel; (2) per chunk of the
// sequence, in parallel, the chunk's transfer function: for an incoming state m
// the first point that replaces it is the first strict prefix maximum of |res|
// that beats m, and what the chunk leaves behind from there on does not depend on
// m any more; (3) one thread chains the chunk functions.
#include <math.h>

#include "atomdev_internal.h"

namespace {

constexpr int TPB = 256;
constexpr int CHUNK = 2048;    // sequence elements per chunk
constexpr int MAXREC = 24;     // prefix-maximum records kept per chunk; more -> the chain step walks the chunk itself

struct ChunkFn {
    int nrec;                  // -1: too many records
    double thr[MAXREC];        // |res| at the record
    double out_val[MAXREC];    // state after the chunk if the record is the first replacement
    int out_pos[MAXREC];       // sequence index of that state
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
    seq[((size_t)(i - 1) * (jm - 2) + (j - 1)) * (km - 2) + (k - 1)] = res;
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

template <bool GE>
__global__ void k_chunk_functions(const double* __restrict__ seq, size_t n, ChunkFn* __restrict__ fn, int nchunks) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nchunks) return;
    const size_t b = (size_t)c * CHUNK;
    const size_t e = (b + CHUNK < n) ? b + CHUNK : n;
    ChunkFn f;
    f.nrec = 0;
    double best = -1.;
    size_t rec_pos[MAXREC];
    for (size_t q = b; q < e; q++) {
        double a = fabs(seq[q]);
        if (a > best) {  // strict prefix maximum (NaN never is)
            best = a;
            if (f.nrec < MAXREC) { f.thr[f.nrec] = a; rec_pos[f.nrec] = q; }
            f.nrec++;
        }
    }
    if (f.nrec > MAXREC) { fn[c].nrec = -1; return; }
    for (int r = 0; r < f.nrec; r++) {
        double m = seq[rec_pos[r]];
        size_t pos = rec_pos[r];
        for (size_t q = rec_pos[r] + 1; q < e; q++) {
            double x = seq[q];
            if (beats<GE>(x, m)) { m = x; pos = q; }
        }
        f.out_val[r] = m; f.out_pos[r] = (int)pos;
    }
    fn[c] = f;
}

template <bool GE>
__global__ void k_chain_chunks(const double* __restrict__ seq, size_t n, const ChunkFn* __restrict__ fn, int nchunks,
                               double m_in, long pos_in, double* __restrict__ out /* {min, pos} */) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    double m = m_in;
    long pos = pos_in;  // -1: state carried in from the caller, location unchanged
    for (int c = 0; c < nchunks; c++) {
        const ChunkFn& f = fn[c];
        if (f.nrec >= 0) {
            for (int r = 0; r < f.nrec; r++) {
                // first record that replaces m: records are increasing in |res|
                if (GE ? (f.thr[r] >= m) : (f.thr[r] > m)) { m = f.out_val[r]; pos = f.out_pos[r]; break; }
            }
        } else {
            const size_t b = (size_t)c * CHUNK;
            const size_t e = (b + CHUNK < n) ? b + CHUNK : n;
            for (size_t q = b; q < e; q++) {
                double x = seq[q];
                if (beats<GE>(x, m)) { m = x; pos = (long)q; }
            }
        }
    }
    out[0] = m;
    out[1] = (double)pos;
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
    for (int f = 0; f < s.n; f++) {
        Best b; b.v = -1.; b.idx = LAST ? -1 : (long long)n;
        for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (size_t)gridDim.x * blockDim.x) {
            double d = fabs(s.x[f][off + q] - s.xn[f][off + q]);
            take<LAST>(b, d, (long long)q);
        }
        sh[threadIdx.x] = b;
        __syncthreads();
        for (int st = TPB / 2; st > 0; st >>= 1) {
            if (threadIdx.x < st) take<LAST>(sh[threadIdx.x], sh[threadIdx.x + st].v, sh[threadIdx.x + st].idx);
            __syncthreads();
        }
        if (threadIdx.x == 0) part[(size_t)f * gridDim.x + blockIdx.x] = sh[0];
        __syncthreads();
    }
}

template <bool LAST>
__global__ void k_steady_final(const Best* __restrict__ part, int nblocks, int nf, Best* __restrict__ out) {
    int f = threadIdx.x;
    if (f >= nf) return;
    Best b; b.v = -1.; b.idx = LAST ? -1 : (1LL << 62);
    for (int i = 0; i < nblocks; i++) take<LAST>(b, part[(size_t)f * nblocks + i].v, part[(size_t)f * nblocks + i].idx);
    out[f] = b;
}

int residuum(atomdev* h, bool three_d, bool ge, double* min, int loc[3]) {
    const int im = h->im, jm = h->jm, km = h->km;
    const size_t n = three_d ? (size_t)(im - 2) * (jm - 2) * (km - 2) : (size_t)(jm - 2) * (km - 2);
    const int nchunks = (int)((n + CHUNK - 1) / CHUNK);
    // scratch: seq | chunk functions | result
    const size_t need = n * sizeof(double) + (size_t)nchunks * sizeof(ChunkFn) + 64;
    if (need > h->red_bytes) {
        ATOMDEV_TRY(hipStreamSynchronize(h->stream));
        ATOMDEV_TRY(hipFree(h->red));
        h->red = nullptr; h->red_bytes = 0;
        ATOMDEV_TRY(hipMalloc(&h->red, need));
        h->red_bytes = need;
    }
    double* seq = (double*)h->red;
    ChunkFn* fn = (ChunkFn*)((char*)h->red + ((n * sizeof(double) + 63) / 64) * 64);
    double* out = (double*)((char*)fn + (size_t)nchunks * sizeof(ChunkFn));
    if ((char*)(out + 2) > (char*)h->red + h->red_bytes) { atomdev_set_error("residuum: scratch layout"); return 1; }
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
    const unsigned cb = (unsigned)((nchunks + 63) / 64);
    if (ge) {
        hipLaunchKernelGGL(k_chunk_functions<true>, dim3(cb), dim3(64), 0, h->stream, seq, n, fn, nchunks);
        hipLaunchKernelGGL(k_chain_chunks<true>, dim3(1), dim3(64), 0, h->stream, seq, n, fn, nchunks, m_in, -1L, out);
    } else {
        hipLaunchKernelGGL(k_chunk_functions<false>, dim3(cb), dim3(64), 0, h->stream, seq, n, fn, nchunks);
        hipLaunchKernelGGL(k_chain_chunks<false>, dim3(1), dim3(64), 0, h->stream, seq, n, fn, nchunks, m_in, -1L, out);
    }
    ATOMDEV_TRY(hipGetLastError());
    double res[2];
    ATOMDEV_TRY(hipMemcpyAsync(res, out, sizeof(res), hipMemcpyDeviceToHost, h->stream));
    ATOMDEV_TRY(hipStreamSynchronize(h->stream));
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
        hipLaunchKernelGGL(k_steady_partial<true>, dim3(nblocks), dim3(TPB), 0, h->stream, s, off, n, part);
        hipLaunchKernelGGL(k_steady_final<true>, dim3(1), dim3(64), 0, h->stream, part, nblocks, nf, fin);
    } else {
        hipLaunchKernelGGL(k_steady_partial<false>, dim3(nblocks), dim3(TPB), 0, h->stream, s, off, n, part);
        hipLaunchKernelGGL(k_steady_final<false>, dim3(1), dim3(64), 0, h->stream, part, nblocks, nf, fin);
    }
    ATOMDEV_TRY(hipGetLastError());
    Best res[9];
    ATOMDEV_TRY(hipMemcpyAsync(res, fin, nf * sizeof(Best), hipMemcpyDeviceToHost, h->stream));
    ATOMDEV_TRY(hipStreamSynchronize(h->stream));
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

}  // namespace

extern "C" int atomdev_atm_residuum3d(atomdev_t* h, double* min, int loc[3]) {
    ATOMDEV_REQUIRE(h && h->model == ATOMDEV_ATM && min && loc, "atomdev_atm_residuum3d: bad argument");
    ATOMDEV_TRY(hipSetDevice(h->device));
    return residuum(h, true, false, min, loc);
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
