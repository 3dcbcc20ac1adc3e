// Largest relative error of the hardware's fp64 reciprocal estimate, and after one and two Newton steps.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(double lo, double hi, int n, double* out) {
    double e0 = 0., e1 = 0., e2 = 0.;
    for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < n; q += gridDim.x * blockDim.x) {
        const double b = lo + (hi - lo) * ((double)q + .37) / n;
        double y = __builtin_amdgcn_rcp(b);
        const double ex = 1. / b;
        e0 = fmax(e0, fabs(y - ex) / ex);
        y = __builtin_fma(y, __builtin_fma(-b, y, 1.), y);
        e1 = fmax(e1, fabs(y - ex) / ex);
        y = __builtin_fma(y, __builtin_fma(-b, y, 1.), y);
        e2 = fmax(e2, fabs(y - ex) / ex);
    }
    out[3 * (blockIdx.x * blockDim.x + threadIdx.x) + 0] = e0;
    out[3 * (blockIdx.x * blockDim.x + threadIdx.x) + 1] = e1;
    out[3 * (blockIdx.x * blockDim.x + threadIdx.x) + 2] = e2;
}
int main() {
    const int nb = 256, nt = 256;
    double* d; hipMalloc(&d, sizeof(double) * 3 * nb * nt);
    static double h[3 * nb * nt];
    const double ranges[3][2] = { { 1., 2. }, { .5, 1.5 }, { 200., 400. } };
    for (auto& r : ranges) {
        hipLaunchKernelGGL(k, dim3(nb), dim3(nt), 0, 0, r[0], r[1], 1 << 26, d);
        if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) { printf("failed\n"); return 1; }
        double e[3] = { 0, 0, 0 };
        for (int q = 0; q < nb * nt; q++) for (int c = 0; c < 3; c++) e[c] = fmax(e[c], h[3 * q + c]);
        printf("[%g, %g): estimate %.3g, one step %.3g, two steps %.3g\n", r[0], r[1], e[0], e[1], e[2]);
    }
    return 0;
}
