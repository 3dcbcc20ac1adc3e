// Atmosphere -> hydrosphere coupling: PostProcess_Atmosphere::Atmosphere_v_w_Transfer
// (atmosphere/PostProcess_Atm.cpp:98-118) writes the surface values of v * u_0, w * u_0, t, p_dyn,
// Evaporation_Dalton and Precipitation in fixed notation with 4 decimals;
// PostProcess_Hydrosphere::Atmosphere_TransferFile_read (hydrosphere/PostProcess_Hyd.cpp:642-668) reads them
// into the hydrosphere's top level (and sets p_dyn there to 0).  The rounding to 4 decimals is therefore part
// of the hydrosphere's input.  Here the same values go from one handle to the other without the text file,
// through a kernel that reproduces the decimal round trip; the file itself can be written and read too.
#include <fstream>
#include <string>

#include "atomdev_internal.h"
#include "fixed4.h"
#include <memory>

#include "host_parallel.h"
#include "text_numbers.h"

namespace {

constexpr int TPB = 256;

// The double that "%.4f" followed by strtod makes of x: the decimal r / 10^4 nearest to x (ties to the even
// r, as glibc's exact conversion does in round-to-nearest), read back as the nearest double.  The product
// x * 10^4 is inexact, so the decision between r and its neighbour is taken on exactly rounded residuals
// (fma), whose signs are those of the exact differences.  Values of 4e11 and beyond are returned as they are
// (no model field comes near).
__device__ __forceinline__ double through_4_decimals(double x) {
    if (!(fabs(x) < 4.0e11)) return x;
    double r = rint(x * 1.0e4);
    const double d = fma(x, 1.0e4, -r);
    if (d > 0.) {
        const double t = fma(x, 1.0e4, -(r + .5));
        if (t > 0. || (t == 0. && fmod(r, 2.) != 0.)) r += 1.;
    } else if (d < 0.) {
        const double t = fma(x, 1.0e4, -(r - .5));
        if (t < 0. || (t == 0. && fmod(r, 2.) != 0.)) r -= 1.;
    }
    const double y = r / 1.0e4;
    return (r == 0.) ? copysign(0., x) : y;  // "-0.0000" reads back as -0.0
}

// planes [6][n2]: v * u_0, w * u_0, t, p_dyn of level 0; Evaporation_Dalton; Precipitation
__global__ void k_transfer_planes(const double* __restrict__ v, const double* __restrict__ w, const double* __restrict__ t,
                                  const double* __restrict__ p_dyn, const double* __restrict__ evap,
                                  const double* __restrict__ precip, double u_0, size_t n2, int rounded,
                                  double* __restrict__ out) {
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n2) return;
    double x[6] = { v[q] * u_0, w[q] * u_0, t[q], p_dyn[q], evap[q], precip[q] };
#pragma unroll
    for (int n = 0; n < 6; n++) out[(size_t)n * n2 + q] = rounded ? through_4_decimals(x[n]) : x[n];
}

int planes(atomdev* atm, int rounded) {
    ATOMDEV_TRY(hipSetDevice(atm->device));
    if (!atm->xfer) ATOMDEV_TRY(hipMalloc(&atm->xfer, 6 * atm->n2 * sizeof(double)));
    hipLaunchKernelGGL(k_transfer_planes, dim3((unsigned)((atm->n2 + TPB - 1) / TPB)), dim3(TPB), 0, atm->stream,
                       atm->f[ATOMDEV_V], atm->f[ATOMDEV_W], atm->f[ATOMDEV_T], atm->f[ATOMDEV_P_DYN],
                       atm->g[ATOMDEV_EVAPORATION_DALTON], atm->g[ATOMDEV_PRECIPITATION], atm->p.u_0, atm->n2, rounded,
                       atm->xfer);
    ATOMDEV_TRY(hipGetLastError());
    if (int rc__ = atomdev_sync_stream(atm)) return rc__;
    return 0;
}

// six planes (host or device memory) into the hydrosphere's top level and 2-D fields
int load_planes(atomdev* hyd, const double* src) {
    ATOMDEV_TRY(hipSetDevice(hyd->device));
    const size_t n2 = hyd->n2, top = (size_t)(hyd->im - 1) * n2, bytes = n2 * sizeof(double);
    double* dst[6] = { hyd->f[ATOMDEV_V] + top, hyd->f[ATOMDEV_W] + top, hyd->f[ATOMDEV_T] + top, nullptr,
                       hyd->g[ATOMDEV_EVAPORATION_DALTON], hyd->g[ATOMDEV_PRECIPITATION] };
    for (int n = 0; n < 6; n++)
        if (dst[n]) ATOMDEV_TRY(hipMemcpyAsync(dst[n], src + (size_t)n * n2, bytes, hipMemcpyDefault, hyd->stream));
    ATOMDEV_TRY(hipMemsetAsync(hyd->f[ATOMDEV_P_DYN] + top, 0, bytes, hyd->stream));  // PostProcess_Hyd.cpp:666
    if (int rc__ = atomdev_sync_stream(hyd)) return rc__;
    return 0;
}

}  // namespace

extern "C" int atomdev_transfer_atm_to_hyd(atomdev_t* atm, atomdev_t* hyd) {
    ATOMDEV_REQUIRE(atm && atm->model == ATOMDEV_ATM, "atomdev_transfer_atm_to_hyd: first handle is not an atmosphere");
    ATOMDEV_REQUIRE(hyd && hyd->model == ATOMDEV_HYD, "atomdev_transfer_atm_to_hyd: second handle is not a hydrosphere");
    ATOMDEV_REQUIRE(atm->jm == hyd->jm && atm->km == hyd->km, "atomdev_transfer_atm_to_hyd: the two surfaces differ in size");
    if (int rc = planes(atm, 1)) return rc;
    return load_planes(hyd, atm->xfer);  // peer or same-device copy, whichever the two handles need
}

extern "C" int atomdev_atm_write_transfer_file(atomdev_t* atm, const char* path) {
    ATOMDEV_REQUIRE(atm && atm->model == ATOMDEV_ATM && path, "atomdev_atm_write_transfer_file: bad argument");
    if (int rc = planes(atm, 0)) return rc;
    const size_t n2 = atm->n2;
    auto xs = std::make_shared<std::vector<double>>(6 * n2);
    ATOMDEV_TRY(hipMemcpy(xs->data(), atm->xfer, 6 * n2 * sizeof(double), hipMemcpyDeviceToHost));
    // The values are on the host: inside a one-call slice the text is made and written by a thread of its own while the
    // slice goes on (the call that ends the slice waits for it).
    const std::string file(path);
    return atomdev_defer_file(atm, file, [xs, n2, file]() -> std::string {
        const std::vector<double>& x = *xs;
        // the text an ofstream with ios::fixed and precision(4) writes: six numbers a line, one space between them
        FILE* f = std::fopen(file.c_str(), "w");
        if (!f) return "atomdev_atm_write_transfer_file: could not open " + file;
        const size_t rows = 4096, pieces = (n2 + rows - 1) / rows;   // a few host threads make the text side by side
        std::vector<std::string> piece(pieces);
        host_parallel_run(pieces, [&](size_t c) {
            std::string& out = piece[c];
            out.reserve(rows * 60);
            char line[6 * 400 + 8];   // a row at a time: one append per row, not per number
            for (size_t q = c * rows; q < n2 && q < (c + 1) * rows; q++) {
                char* e = line;
                for (int n = 0; n < 6; n++) {
                    e = put_fixed4(e, x[(size_t)n * n2 + q]);
                    *e++ = n < 5 ? ' ' : '\n';
                }
                out.append(line, (size_t)(e - line));
            }
        });
        bool ok = true;
        for (const std::string& out : piece) ok = ok && std::fwrite(out.data(), 1, out.size(), f) == out.size();
        if (std::fclose(f) != 0 || !ok) return "atomdev_atm_write_transfer_file: write failed: " + file;
        return std::string();
    });
}

extern "C" int atomdev_hyd_read_transfer_file(atomdev_t* hyd, const char* path) {
    ATOMDEV_REQUIRE(hyd && hyd->model == ATOMDEV_HYD && path, "atomdev_hyd_read_transfer_file: bad argument");
    const size_t n2 = hyd->n2;
    std::vector<double> x(6 * n2);
    TextNumbers tn;
    ATOMDEV_REQUIRE(tn.open(path), std::string("atomdev_hyd_read_transfer_file: could not open ") + path);
    if (tn.plain()) {  // plain decimal tokens: converted directly, the values the stream extraction returns
        std::vector<double> every;
        tn.all(every);
        ATOMDEV_REQUIRE(every.size() >= 6 * n2, std::string("atomdev_hyd_read_transfer_file: fewer than jm * km rows of six numbers in ") + path);
        host_parallel_run(6, [&](size_t n) { for (size_t q = 0; q < n2; q++) x[n * n2 + q] = every[6 * q + n]; });
    }
    if (!tn.plain() || tn.out_of_range()) {
        std::ifstream f(path);
        for (size_t q = 0; q < n2; q++)
            for (int n = 0; n < 6; n++) f >> x[(size_t)n * n2 + q];
        ATOMDEV_REQUIRE(!f.fail(), std::string("atomdev_hyd_read_transfer_file: fewer than jm * km rows of six numbers in ") + path);
    }
    return load_planes(hyd, x.data());
}
