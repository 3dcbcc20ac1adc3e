// Memory-system experiment: what bandwidth does the sweep's access pattern get, apart from any arithmetic?
// NF_R fields are read and NF_W written at the points of a 41 x 181 x 361 grid, by 234 workgroups (one per level and
// block of 64 longitudes) that walk the rows j = 1 .. jm-2; a wave of a workgroup takes `rows_per_visit` consecutive rows
// at a time, the waves take the visits in turn.  Variants: fields in (i, j, k) order as in the library, or tiled so that a
// workgroup's part of a field is one contiguous stream.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int IM = 41, JM = 181, KM = 361, NF_R = 23, NF_W = 12;
struct Args { const double* r[NF_R]; double* w[NF_W]; };

// LPW > 0: lines are numbered through all levels and a workgroup takes LPW consecutive ones (a wave may span two levels)
template <int RPV, bool TILED, int LPW = 0, int MODE = 0>
__global__ __launch_bounds__(1024) void k_stream(Args a, int nwaves) {
    const int gx = (KM - 2 + 63) / 64;
    int lev = blockIdx.x / gx, bx = blockIdx.x % gx;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int k = 1 + bx * 64 + lane; if (k > KM - 2) k = KM - 2;
    if (LPW > 0) {
        int L = blockIdx.x * LPW + (lane < LPW ? lane : LPW - 1);
        if (L >= (IM - 2) * (KM - 2)) L = (IM - 2) * (KM - 2) - 1;
        lev = L / (KM - 2); k = 1 + L % (KM - 2);
    }
    const int i = 1 + lev;
    const int NR = JM - 2;
    int v0 = wave, vstep = nwaves; bool do_r = true, do_w = true;
    if (MODE == 4) { const int nst = nwaves / 4, nld = nwaves - nst; do_w = wave >= nld; do_r = !do_w; v0 = do_w ? wave - nld : wave; vstep = do_w ? nst : nld; }
    for (int v = v0; v * RPV < NR; v += vstep) {
        double acc[RPV];
#pragma unroll
        for (int q = 0; q < RPV; q++) acc[q] = 0.;
        double val[RPV][NF_R];
#pragma unroll
        for (int f = 0; f < NF_R; f++)
#pragma unroll
            for (int q = 0; q < RPV; q++) {
                int j = 1 + v * RPV + q; if (j > NR) j = NR;
                size_t idx = TILED ? (((size_t)(lev * gx + bx) * JM + j) * 64 + lane) : (((size_t)i * JM + j) * KM + k);
                if (MODE == 1) idx = (((size_t)(lev * gx + bx) * JM + j) * (NF_R + NF_W) + f) * 64 + lane;
                val[q][f] = (MODE == 3 || !do_r) ? 1. : a.r[MODE == 1 ? 0 : f][idx];
            }
#pragma unroll
        for (int f = 0; f < NF_R; f++)
#pragma unroll
            for (int q = 0; q < RPV; q++) acc[q] += val[q][f];
#pragma unroll
        for (int f = 0; f < NF_W; f++)
#pragma unroll
            for (int q = 0; q < RPV; q++) {
                int j = 1 + v * RPV + q; if (j > NR) j = NR;
                size_t idx = TILED ? (((size_t)(lev * gx + bx) * JM + j) * 64 + lane) : (((size_t)i * JM + j) * KM + k);
                if (MODE == 1) idx = (((size_t)(lev * gx + bx) * JM + j) * (NF_R + NF_W) + NF_R + f) * 64 + lane;
                if ((MODE != 2 && do_w) || acc[q] == 123.456) ((double*)(MODE == 1 ? a.r[0] : a.w[f]))[idx] = acc[q] + f;
            }
    }
}

template <int RPV, bool TILED, int LPW = 0, int MODE = 0>
void run(Args a, int nwaves, const char* name) {
    const int gx = (KM - 2 + 63) / 64;
    const int nwg = LPW > 0 ? ((IM - 2) * (KM - 2) + LPW - 1) / LPW : (IM - 2) * gx;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int it = 0; it < 3; it++) hipLaunchKernelGGL((k_stream<RPV, TILED, LPW, MODE>), dim3(nwg), dim3(64 * nwaves), 0, 0, a, nwaves);
    CHECK(hipEventRecord(e0));
    const int N = 20;
    for (int it = 0; it < N; it++) hipLaunchKernelGGL((k_stream<RPV, TILED, LPW, MODE>), dim3(nwg), dim3(64 * nwaves), 0, 0, a, nwaves);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= N;
    const double bytes = (double)(IM - 2) * (JM - 2) * (KM - 2) * 8. * (MODE == 2 ? NF_R : MODE == 3 ? NF_W : NF_R + NF_W);
    printf("%-28s wgs %3d waves %2d rows/visit %d: %.3f ms  %.2f TB/s\n", name, nwg, nwaves, RPV, ms, bytes / ms * 1e-9);
}

int main() {
    const size_t n = (size_t)IM * JM * KM + 64 * 1024;
    Args a;
    for (int f = 0; f < NF_R; f++) { double* p; CHECK(hipMalloc(&p, n * 8)); CHECK(hipMemset(p, 0, n * 8)); a.r[f] = p; }
    for (int f = 0; f < NF_W; f++) { double* p; CHECK(hipMalloc(&p, n * 8)); CHECK(hipMemset(p, 0, n * 8)); a.w[f] = p; }
    double* big; CHECK(hipMalloc(&big, n * 8 * (NF_R + NF_W))); CHECK(hipMemset(big, 0, n * 8 * (NF_R + NF_W)));
    Args b = a; b.r[0] = big;
    for (int nw : {8, 16}) {
        run<1, false>(a, nw, "(i,j,k) read + write");
        run<2, false>(a, nw, "(i,j,k) read + write");
        run<2, false, 55>(a, nw, "(i,j,k) 55 lines a workgroup");
        run<1, false, 0, 4>(a, nw, "(i,j,k) load waves, store waves");
        run<1, false, 0, 2>(a, nw, "(i,j,k) read only");
        run<1, false, 0, 3>(a, nw, "(i,j,k) write only");
        run<1, true>(a, nw, "tiled, field by field");
        run<1, false, 0, 1>(b, nw, "one allocation, interleaved");
    }
    return 0;
}
