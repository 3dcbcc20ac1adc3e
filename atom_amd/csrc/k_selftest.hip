// Device-side evaluation of the library's arithmetic helpers on caller-supplied operands, for property tests:
// the double-double exp / pow of cr_math.h (which must give the bits the same source gives on the host) and the
// exact-division helper div_by against the hardware's IEEE division.
#include "atomdev_internal.h"
#include "cr_math.h"

#include <chrono>
#include <thread>

namespace {

__global__ void k_selftest(int kind, size_t n, const double* __restrict__ x, const double* __restrict__ y, double* __restrict__ out) {
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n) return;
    const double a = x[q], b = y ? y[q] : 0.;
    double r = 0.;
    switch (kind) {
    case ATOMDEV_SELFTEST_EXP: r = cr_exp(a); break;
    case ATOMDEV_SELFTEST_POW: r = cr_pow(a, b); break;
    case ATOMDEV_SELFTEST_POW4: r = cr_pow4(a); break;
    case ATOMDEV_SELFTEST_ROOT4: r = cr_root4(a); break;
    case ATOMDEV_SELFTEST_DIV_BY: r = div_by(a, b, 1. / b); break;
    case ATOMDEV_SELFTEST_DIVIDE: r = a / b; break;
    }
    out[q] = r;
}

__global__ void k_raise_error(int* err) { __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

}  // namespace

extern "C" int atomdev_selftest_raise_kernel_error(atomdev_t* h) {
    ATOMDEV_REQUIRE(h, "atomdev_selftest_raise_kernel_error: null handle");
    ATOMDEV_TRY(hipSetDevice(h->device));
    hipLaunchKernelGGL(k_raise_error, dim3(1), dim3(1), 0, h->stream, (int*)atomdev_kernel_error_word(h));
    ATOMDEV_TRY(hipGetLastError());
    return 0;
}

extern "C" int atomdev_selftest_math(atomdev_t* h, int kind, size_t n, const double* x, const double* y, double* out) {
    ATOMDEV_REQUIRE(h && x && out && n > 0, "atomdev_selftest_math: bad argument");
    ATOMDEV_REQUIRE(kind >= ATOMDEV_SELFTEST_EXP && kind <= ATOMDEV_SELFTEST_DIVIDE, "atomdev_selftest_math: unknown kind");
    const bool two = kind == ATOMDEV_SELFTEST_POW || kind == ATOMDEV_SELFTEST_DIV_BY || kind == ATOMDEV_SELFTEST_DIVIDE;
    ATOMDEV_REQUIRE(!two || y, "atomdev_selftest_math: this kind takes two operands");
    ATOMDEV_TRY(hipSetDevice(h->device));
    double *dx = nullptr, *dy = nullptr, *dout = nullptr;
    ATOMDEV_TRY(hipMalloc(&dx, n * 8));
    int rc = 0;
    if (!rc) rc = atomdev_check(hipMalloc(&dout, n * 8), "hipMalloc");
    if (!rc && two) rc = atomdev_check(hipMalloc(&dy, n * 8), "hipMalloc");
    if (!rc) rc = atomdev_check(hipMemcpyAsync(dx, x, n * 8, hipMemcpyHostToDevice, h->stream), "hipMemcpyAsync");
    if (!rc && two) rc = atomdev_check(hipMemcpyAsync(dy, y, n * 8, hipMemcpyHostToDevice, h->stream), "hipMemcpyAsync");
    if (!rc) {
        hipLaunchKernelGGL(k_selftest, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, kind, n, dx, dy, dout);
        rc = atomdev_check(hipGetLastError(), "k_selftest");
    }
    if (!rc) rc = atomdev_check(hipMemcpyAsync(out, dout, n * 8, hipMemcpyDeviceToHost, h->stream), "hipMemcpyAsync");
    if (!rc) rc = atomdev_check(hipStreamSynchronize(h->stream), "hipStreamSynchronize");
    (void)hipFree(dx); (void)hipFree(dout);
    if (dy) (void)hipFree(dy);
    return rc;
}

// Two deferred writes to one path, the first held back by 300 ms: the file must end up with the second one's text.
extern "C" int atomdev_selftest_deferred_file_order(atomdev_t* h, const char* path) {
    ATOMDEV_REQUIRE(h && path, "atomdev_selftest_deferred_file_order: bad argument");
    const std::string file(path);
    auto writer = [file](const char* text, int delay_ms) {
        return [file, text, delay_ms]() -> std::string {
            std::this_thread::sleep_for(std::chrono::milliseconds(delay_ms));
            FILE* f = std::fopen(file.c_str(), "w");
            if (!f) return "could not open " + file;
            std::fputs(text, f);
            return std::fclose(f) == 0 ? std::string() : "write failed: " + file;
        };
    };
    h->defer_files = true;
    int rc = atomdev_defer_file(h, file, writer("first\n", 300));
    if (!rc) rc = atomdev_defer_file(h, file, writer("second\n", 0));
    h->defer_files = false;
    const int rj = atomdev_join_files(h);
    return rc ? rc : rj;
}
