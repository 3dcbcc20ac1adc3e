// Micro-benchmarks of fp64 VALU latencies on gfx950 (s_memtime ticks = shader cycles).
// build: hipcc -O3 --offload-arch=gfx950 -o f64lat f64lat.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define N 512

__global__ void k_fma_dep(double* out, long long* cyc, double a, double b) {
    double x = out[threadIdx.x];
    asm volatile("" : "+v"(x));
    long long t0 = __builtin_readcyclecounter();
    asm volatile("" : "+v"(x));
#pragma unroll
    for (int i = 0; i < N; i++) x = __builtin_fma(x, a, b);
    asm volatile("" : "+v"(x));
    long long t1 = __builtin_readcyclecounter();
    asm volatile("" : "+v"(x));
    out[threadIdx.x + blockIdx.x * blockDim.x] = x;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_fma_ind4(double* out, long long* cyc, double a, double b) {
    double x0 = out[threadIdx.x], x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
    asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
    long long t0 = __builtin_readcyclecounter();
    asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
#pragma unroll
    for (int i = 0; i < N / 4; i++) {
        x0 = __builtin_fma(x0, a, b); x1 = __builtin_fma(x1, a, b); x2 = __builtin_fma(x2, a, b); x3 = __builtin_fma(x3, a, b);
    }
    asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
    long long t1 = __builtin_readcyclecounter();
    asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
    out[threadIdx.x + blockIdx.x * blockDim.x] = x0 + x1 + x2 + x3;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_add_dep(double* out, long long* cyc, double a, double b) {
    double x = out[threadIdx.x];
    asm volatile("" : "+v"(x));
    long long t0 = __builtin_readcyclecounter();
    asm volatile("" : "+v"(x));
#pragma unroll
    for (int i = 0; i < N; i++) x = x + a;
    asm volatile("" : "+v"(x));
    long long t1 = __builtin_readcyclecounter();
    asm volatile("" : "+v"(x));
    out[threadIdx.x + blockIdx.x * blockDim.x] = x;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_mul_dep(double* out, long long* cyc, double a, double b) {
    double x = out[threadIdx.x];
    asm volatile("" : "+v"(x));
    long long t0 = __builtin_readcyclecounter();
    asm volatile("" : "+v"(x));
#pragma unroll
    for (int i = 0; i < N; i++) x = x * a;
    asm volatile("" : "+v"(x));
    long long t1 = __builtin_readcyclecounter();
    asm volatile("" : "+v"(x));
    out[threadIdx.x + blockIdx.x * blockDim.x] = x;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_div_dep(double* out, long long* cyc, double a, double b) {
    double x = out[threadIdx.x];
    asm volatile("" : "+v"(x));
    long long t0 = __builtin_readcyclecounter();
    asm volatile("" : "+v"(x));
#pragma unroll
    for (int i = 0; i < N / 8; i++) x = a / x;
    asm volatile("" : "+v"(x));
    long long t1 = __builtin_readcyclecounter();
    asm volatile("" : "+v"(x));
    out[threadIdx.x + blockIdx.x * blockDim.x] = x;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_rcp_dep(double* out, long long* cyc, double a, double b) {
    double x = out[threadIdx.x];
    asm volatile("" : "+v"(x));
    long long t0 = __builtin_readcyclecounter();
    asm volatile("" : "+v"(x));
#pragma unroll
    for (int i = 0; i < N / 8; i++) x = __builtin_amdgcn_rcp(x);
    asm volatile("" : "+v"(x));
    long long t1 = __builtin_readcyclecounter();
    asm volatile("" : "+v"(x));
    out[threadIdx.x + blockIdx.x * blockDim.x] = x;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
// LDS publish -> barrier -> read round trip, blockDim waves of 64
__global__ void k_lds_round(double* out, long long* cyc, double a, double b) {
    __shared__ double sh[2][512];
    double x = out[threadIdx.x];
    const int peer = (threadIdx.x + 64) % blockDim.x;
    asm volatile("" : "+v"(x));
    long long t0 = __builtin_readcyclecounter();
    asm volatile("" : "+v"(x));
#pragma unroll 8
    for (int i = 0; i < N / 8; i++) {
        sh[i & 1][threadIdx.x] = x;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        x = sh[i & 1][peer] + a;
    }
    asm volatile("" : "+v"(x));
    long long t1 = __builtin_readcyclecounter();
    asm volatile("" : "+v"(x));
    out[threadIdx.x + blockIdx.x * blockDim.x] = x;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

__global__ void k_long(double* out, long long* cyc, double a, double b, int iters) {
    double x = out[threadIdx.x];
    asm volatile("" : "+v"(x));
    long long t0 = __builtin_readcyclecounter();
    long long w0 = wall_clock64();
    asm volatile("" : "+v"(x));
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 64; i++) x = __builtin_fma(x, a, b);
    }
    asm volatile("" : "+v"(x));
    long long t1 = __builtin_readcyclecounter();
    long long w1 = wall_clock64();
    asm volatile("" : "+v"(x));
    out[threadIdx.x + blockIdx.x * blockDim.x] = x;
    if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = w1 - w0; }
}

__global__ void k_long_masked(double* out, long long* cyc, double a, double b, int iters, int active) {
    double x = out[threadIdx.x];
    if ((threadIdx.x & 63) < active) {
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int i = 0; i < 64; i++) x = __builtin_fma(x, a, b);
        }
    }
    out[threadIdx.x + blockIdx.x * blockDim.x] = x;
}
static void run_masked(int blocks, int threads, int iters, int active) {
    double* out; long long* cyc;
    hipMalloc(&out, sizeof(double) * blocks * threads + 4096);
    hipMalloc(&cyc, 2 * sizeof(long long));
    hipMemset(out, 0, sizeof(double) * blocks * threads + 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_long_masked, dim3(blocks), dim3(threads), 0, 0, out, cyc, 1.0000001, 1e-9, iters, active);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_long_masked, dim3(blocks), dim3(threads), 0, 0, out, cyc, 1.0000001, 1e-9, iters, active);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("masked fma chain blocks=%d threads=%d active lanes=%d: %.3f ms, %.2f ns/op/wave\n", blocks, threads, active, ms, ms * 1e6 / (64.0 * iters));
    hipFree(out); hipFree(cyc);
}

static void run_long(int blocks, int threads, int iters) {
    double* out; long long* cyc;
    hipMalloc(&out, sizeof(double) * blocks * threads + 4096);
    hipMalloc(&cyc, 2 * sizeof(long long));
    hipMemset(out, 0, sizeof(double) * blocks * threads + 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_long, dim3(blocks), dim3(threads), 0, 0, out, cyc, 1.0000001, 1e-9, iters);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_long, dim3(blocks), dim3(threads), 0, 0, out, cyc, 1.0000001, 1e-9, iters);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c[2];
    hipMemcpy(c, cyc, sizeof(c), hipMemcpyDeviceToHost);
    printf("long fma chain blocks=%d threads=%d: %.3f ms, %lld memtime ticks (%.1f MHz), %lld wall ticks (%.1f MHz), %.2f ns/op\n", blocks, threads, ms,
           c[0], c[0] / ms / 1e3, c[1], c[1] / ms / 1e3, ms * 1e6 / (64.0 * iters));
    hipFree(out); hipFree(cyc);
}

// one "stage round": read a peer's value from LDS, NOPS dependent div_by-like ops (mul, fma, fma), publish, barrier
template <int NOPS>
__global__ void k_round(double* out, long long* cyc, double a, double b) {
    __shared__ double sh[2][512];
    double x = out[threadIdx.x];
    const int peer = (threadIdx.x + 64) % blockDim.x;
    sh[1][threadIdx.x] = x;
    __syncthreads();
    asm volatile("" : "+v"(x));
    long long t0 = __builtin_readcyclecounter();
    asm volatile("" : "+v"(x));
    for (int i = 0; i < 256; i++) {
        double y = sh[(i & 1) ^ 1][peer];
#pragma unroll
        for (int n = 0; n < NOPS / 3; n++) {
            double q = x * a;
            double r = __builtin_fma(-q, b, x);
            x = __builtin_fma(r, a, q);
        }
        x = x + y;
        sh[i & 1][threadIdx.x] = x;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    asm volatile("" : "+v"(x));
    long long t1 = __builtin_readcyclecounter();
    asm volatile("" : "+v"(x));
    out[threadIdx.x + blockIdx.x * blockDim.x] = x;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <class K>
static void run(const char* name, K k, int blocks, int threads, int ops) {
    double* out; long long* cyc;
    hipMalloc(&out, sizeof(double) * blocks * threads + 4096);
    hipMalloc(&cyc, sizeof(long long));
    std::vector<double> h(blocks * threads + 512, 1.000001);
    hipMemcpy(out, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice);
    for (int rep = 0; rep < 3; rep++) hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, out, cyc, 1.0000001, 1e-9);
    hipDeviceSynchronize();
    long long c;
    hipMemcpy(&c, cyc, sizeof(c), hipMemcpyDeviceToHost);
    printf("%-28s blocks=%4d threads=%4d  %8lld ticks  %7.2f ticks/op\n", name, blocks, threads, c, (double)c / ops);
    hipFree(out); hipFree(cyc);
}

int main() {
    int clk = 0;
    hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    int wclk = 0;
    hipDeviceGetAttribute(&wclk, hipDeviceAttributeWallClockRate, 0);
    printf("clockRate %d kHz  wallClockRate %d kHz\n", clk, wclk);
    for (int act : {64, 48, 32, 16, 1}) run_masked(234, 512, 10000, act);
    run_long(1, 64, 20000);
    run_long(1, 512, 20000);
    run_long(234, 512, 20000);
    run_long(2048, 512, 5000);
    run("fma_f64 dependent", k_fma_dep, 1, 64, N);
    run("fma_f64 dependent 2w/SIMD", k_fma_dep, 1, 512, N);
    run("fma_f64 dep, chip full", k_fma_dep, 256, 512, N);
    run("fma_f64 4 independent", k_fma_ind4, 1, 64, N);
    run("fma_f64 4 indep 2w/SIMD", k_fma_ind4, 1, 512, N);
    run("add_f64 dependent", k_add_dep, 1, 64, N);
    run("mul_f64 dependent", k_mul_dep, 1, 64, N);
    run("div_f64 dependent", k_div_dep, 1, 64, N / 8);
    run("div_f64 dependent 2w/SIMD", k_div_dep, 1, 512, N / 8);
    run("rcp_f64 dependent", k_rcp_dep, 1, 64, N / 8);
    run("round 0 ops, 4 waves", k_round<0>, 234, 256, 256);
    run("round 30 ops, 4 waves", k_round<30>, 234, 256, 256);
    run("round 60 ops, 4 waves", k_round<60>, 234, 256, 256);
    run("round 30 ops, 8 waves", k_round<30>, 234, 512, 256);
    run("round 60 ops, 8 waves", k_round<60>, 234, 512, 256);
    run("lds round 8 waves", k_lds_round, 1, 512, N / 8);
    run("lds round 8 waves chip full", k_lds_round, 256, 512, N / 8);
    return 0;
}
