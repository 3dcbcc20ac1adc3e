// Point-wise kernels: land flags, fills, old->new copies, limiters, solid ground.
// All are pure streaming kernels over the flat phi-fastest arrays (HBM-bound).
#include <float.h>

#include <cstdlib>

#include "atomdev_internal.h"

namespace {

constexpr int TPB = 256;

__device__ __forceinline__ bool land_of(double h) { return fabs(h - 1) < DBL_EPSILON; }  // lib/Utils.h:22-24

// One byte of flags per grid point, from h and its neighbours.  Which
// neighbours matter is dictated by RK_RHS_3D_* (atmosphere/RHS_Atm.cpp:108-149,
// 245-312; hydrosphere/RHS_Hyd.cpp:94-139, 191-215, 281-341).
__global__ void k_derive_flags(const double* __restrict__ h, uint8_t* __restrict__ flags, int im, int jm, int km) {
    size_t n = (size_t)im * jm * km;
    size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= n) return;
    int k = (int)(id % km);
    int j = (int)((id / km) % jm);
    int i = (int)(id / ((size_t)km * jm));
    auto L = [&](int ii, int jj, int kk) -> bool {
        if (ii < 0 || ii >= im || jj < 0 || jj >= jm || kk < 0 || kk >= km) return false;
        return land_of(h[idx3(jm, km, ii, jj, kk)]);
    };
    auto F = [&](int ii, int jj, int kk) -> bool {  // fluid and inside the grid
        if (ii < 0 || ii >= im || jj < 0 || jj >= jm || kk < 0 || kk >= km) return false;
        return !land_of(h[idx3(jm, km, ii, jj, kk)]);
    };
    const double hc = h[id];
    const bool land = land_of(hc);
    uint8_t fl = 0;
    if (land) fl |= FL_LAND;
    if (hc == 0.) fl |= FL_HZERO;
    if (land && F(i, j - 1, k) && F(i, j - 2, k)) fl |= FL_ONESIDED_S;
    if (!land && (L(i, j - 1, k) || L(i, j + 1, k))) fl |= FL_H0J;
    if (!land && (L(i, j, k - 1) || L(i, j, k + 1))) fl |= FL_H0K;
    if (land && i + 1 < im && h[idx3(jm, km, i + 1, j, k)] == 0.) fl |= FL_RDEAD;
    if (land && ((F(i, j, k + 1) && F(i, j, k + 2)) || (F(i, j, k - 1) && F(i, j, k - 2)))) fl |= FL_D2ZERO;
    flags[id] = fl;
}

__global__ void k_fill(double* __restrict__ a, double v, size_t n) {
    size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id < n) a[id] = v;
}

struct PtrList { double* p[9]; };
struct CPtrList { const double* p[9]; };

// AtomUtils::move_data_to_new_arrays with coeff = 1 (lib/Utils.cpp:41-68)
__global__ void k_copy_lists(CPtrList src, PtrList dst, int nf, size_t off, size_t n) {
    size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= n) return;
    for (int f = 0; f < nf; f++) dst.p[f][off + id] = 1. * src.p[f][off + id];
}
// the same for whole fields (16-byte aligned, as every allocation of the library is), two neighbouring points a thread,
// the stores with the non-temporal hint
__global__ void k_copy_lists_pairs(CPtrList src, PtrList dst, int nf, size_t n) {
    const size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x, pairs = n >> 1;
    if (id < pairs) {
        for (int f = 0; f < nf; f++) {
            typedef double pair_t __attribute__((ext_vector_type(2)));
            // (plain loads: what is read here is what the kernels after the copy read again; the copies are read once)
            const pair_t v = reinterpret_cast<const pair_t*>(src.p[f])[id];
            const pair_t w = { 1. * v.x, 1. * v.y };
            __builtin_nontemporal_store(w, reinterpret_cast<pair_t*>(dst.p[f]) + id);
        }
    } else if (id == pairs && (n & 1)) {
        for (int f = 0; f < nf; f++) dst.p[f][n - 1] = 1. * src.p[f][n - 1];
    }
}

// Stores x only where its bits differ from what the field holds: a limiter changes few points, and the
// write-back of unchanged values would double the kernel's HBM traffic.
__device__ __forceinline__ void store_if_changed(double* p, double was, double x) {
    if (__double_as_longlong(was) != __double_as_longlong(x)) *p = x;
}

// BC_Thermo::Value_Limitation_Atm (atmosphere/BC_Thermo.cpp:3235-3275)
// REST: only what a sweep that stores limited values has left to do - the points the sweep does not store (the six
// faces of the grid) and p_dyn on land.
template <bool REST>
__global__ void k_atm_value_limitation(double* u, double* v, double* w, double* t, double* c, double* cloud, double* ice,
                                       double* co2, double* p_dyn, const uint8_t* __restrict__ flags, size_t n,
                                       int im, int jm, int km) {
    size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= n) return;
    if constexpr (REST) {
        const int k = (int)(id % (size_t)km), j = (int)((id / (size_t)km) % (size_t)jm), i = (int)(id / ((size_t)km * jm));
        if (i > 0 && i < im - 1 && j > 0 && j < jm - 1 && k > 0 && k < km - 1) {
            if (flags[id] & FL_LAND) p_dyn[id] = 0.;
            return;
        }
    }
    const double u0 = u[id], v0 = v[id], w0 = w[id], t0 = t[id], c0 = c[id], cl0 = cloud[id], ic0 = ice[id], co0 = co2[id];
    double x;
    x = u0; if (x >= .106) x = .106; if (x <= -.106) x = -.106; double uu = x;
    x = v0; if (x >= .125) x = .125; if (x <= -.125) x = -.125; double vv = x;
    x = w0; if (x >= 3.5) x = 3.5; if (x <= -.469) x = -.469; double ww = x;
    x = t0; if (x >= 1.165) x = 1.165; if (x <= -.78) x = -.78; store_if_changed(t + id, t0, x);
    x = c0; if (x >= .03) x = .03; if (x < 0.) x = 0.; store_if_changed(c + id, c0, x);
    x = cl0; if (x >= .01) x = .01; if (x < 0.) x = 0.; double cl = x;
    x = ic0; if (x >= .005) x = .005; if (x < 0.) x = 0.; double ic = x;
    x = co0; if (x >= 5.36) x = 5.36; if (x <= 1.) x = 1.; double co = x;
    if (flags[id] & FL_LAND) { uu = 0.; vv = 0.; ww = 0.; cl = 0.; ic = 0.; co = 1.; p_dyn[id] = 0.; }
    store_if_changed(u + id, u0, uu); store_if_changed(v + id, v0, vv); store_if_changed(w + id, w0, ww);
    store_if_changed(cloud + id, cl0, cl); store_if_changed(ice + id, ic0, ic); store_if_changed(co2 + id, co0, co);
}

// BC_Bathymetry_Atmosphere::BC_SolidGround (atmosphere/BC_Bath_Atm.cpp:94-119): levels 0 .. im-2
__global__ void k_atm_solid_ground(double* u, double* v, double* w, double* cloud, double* ice, double* co2,
                                   double* p_dyn, const uint8_t* __restrict__ flags, size_t n_below_top) {
    size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= n_below_top) return;
    if (flags[id] & FL_LAND) { u[id] = 0.; v[id] = 0.; w[id] = 0.; cloud[id] = 0.; ice[id] = 0.; co2[id] = 1.; p_dyn[id] = 0.; }
}

// BC_Thermohalin::Value_Limitation_Hyd (hydrosphere/BC_Thermohalin.cpp:699-722)
__global__ void k_hyd_value_limitation(double* u, double* v, double* w, double* t, double* c, size_t n) {
    size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= n) return;
    const double u0 = u[id], v0 = v[id], w0 = w[id], t0 = t[id], c0 = c[id];
    double x;
    x = u0; if (x >= 11.11) x = 11.11; if (x <= -11.11) x = -11.11; store_if_changed(u + id, u0, x);
    x = v0; if (x >= .552) x = .552; if (x <= -.552) x = -.552; store_if_changed(v + id, v0, x);
    x = w0; if (x >= .552) x = .552; if (x <= -.552) x = -.552; store_if_changed(w + id, w0, x);
    x = t0; if (x >= 1.147) x = 1.147; if (x <= -1.01464) x = -1.01464; store_if_changed(t + id, t0, x);
    x = c0; if (x >= 1.1) x = 1.1; if (x < .95) x = .95; store_if_changed(c + id, c0, x);
}

// BC_Bathymetry_Hydrosphere::BC_SolidGround (hydrosphere/BC_Bath_Hyd.cpp:120-139): old and new arrays
__global__ void k_hyd_solid_ground(PtrList x /* p t c u v w */, PtrList xn, double pa, double ta, double ca,
                                   const uint8_t* __restrict__ flags, size_t n) {
    size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= n) return;
    if (!(flags[id] & FL_LAND)) return;
    const double val[6] = { pa, ta, ca, 0., 0., 0. };
#pragma unroll
    for (int f = 0; f < 6; f++) { x.p[f][id] = val[f]; __builtin_nontemporal_store(val[f], xn.p[f] + id); }   // (the old-time-level fields are read once a step)
}

inline unsigned blocks_for(size_t n) { return (unsigned)((n + TPB - 1) / TPB); }

}  // namespace

// Solid cells as bytes (1 = solid) -> the field h (1. / 0.): the readers build their masks as bytes on the host, an
// eighth of the field to fill, rotate, smooth and send; the flags buffer stages the bytes and is derived anew afterwards.
namespace {
__global__ void k_expand_mask(const uint8_t* __restrict__ solid, double* __restrict__ hfield, size_t n) {
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q < n) hfield[q] = solid[q] ? 1. : 0.;
}
}  // namespace

int atomdev_upload_mask_bytes(atomdev* h, const uint8_t* solid) {
    ATOMDEV_TRY(hipSetDevice(h->device));
    ATOMDEV_TRY(hipMemcpyAsync(h->flags, solid, h->n3, hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(k_expand_mask, dim3(blocks_for(h->n3)), dim3(TPB), 0, h->stream, h->flags, h->f[ATOMDEV_H], h->n3);
    ATOMDEV_TRY(hipGetLastError());
    if (int rc__ = atomdev_sync_stream(h)) return rc__;  // the host buffer may be reused at once
    h->flags_dirty = true;
    return 0;
}

int atomdev_refresh_flags(atomdev* h) {
    if (!h->flags_dirty) return 0;
    hipLaunchKernelGGL(k_derive_flags, dim3(blocks_for(h->n3)), dim3(TPB), 0, h->stream, h->f[ATOMDEV_H], h->flags,
                       h->im, h->jm, h->km);
    ATOMDEV_TRY(hipGetLastError());
    h->flags_dirty = false;
    h->h0i_valid = false;
    return 0;
}

extern "C" int atomdev_fill(atomdev_t* h, int field, double value) {
    ATOMDEV_REQUIRE(h && field >= 0 && field < ATOMDEV_N3D && h->f[field], "atomdev_fill: bad field");
    ATOMDEV_TRY(hipSetDevice(h->device));
    hipLaunchKernelGGL(k_fill, dim3(blocks_for(h->n3)), dim3(TPB), 0, h->stream, h->f[field], value, h->n3);
    ATOMDEV_TRY(hipGetLastError());
    if (field == ATOMDEV_H) h->flags_dirty = true;
    return 0;
}

extern "C" int atomdev_fill2d(atomdev_t* h, int field, double value) {
    ATOMDEV_REQUIRE(h && field >= 0 && field < ATOMDEV_N2D && h->g[field], "atomdev_fill2d: bad field");
    ATOMDEV_TRY(hipSetDevice(h->device));
    hipLaunchKernelGGL(k_fill, dim3(blocks_for(h->n2)), dim3(TPB), 0, h->stream, h->g[field], value, h->n2);
    ATOMDEV_TRY(hipGetLastError());
    return 0;
}

extern "C" int atomdev_move_data_to_new_arrays(atomdev_t* h, int level) {
    ATOMDEV_REQUIRE(h != nullptr, "atomdev_move_data_to_new_arrays: null handle");
    ATOMDEV_REQUIRE(level < h->im, "atomdev_move_data_to_new_arrays: level out of range");
    ATOMDEV_TRY(hipSetDevice(h->device));
    // cAtmosphereModel.cpp:60-63 (old_arrays_3d / old_arrays_2d); the hydrosphere lists are the first six
    static const int O3[9] = { ATOMDEV_U, ATOMDEV_V, ATOMDEV_W, ATOMDEV_T, ATOMDEV_P_DYN, ATOMDEV_C,
                               ATOMDEV_CLOUD, ATOMDEV_ICE, ATOMDEV_CO2 };
    static const int O2[3] = { ATOMDEV_V, ATOMDEV_W, ATOMDEV_P_DYN };
    CPtrList s; PtrList d;
    int nf; size_t off, n;
    if (level < 0) {
        nf = h->model == ATOMDEV_ATM ? 9 : 6; off = 0; n = h->n3;
        for (int f = 0; f < nf; f++) { s.p[f] = h->f[O3[f]]; d.p[f] = h->f[O3[f] + 9]; }
    } else {
        nf = 3; off = (size_t)level * h->n2; n = h->n2;
        for (int f = 0; f < nf; f++) { s.p[f] = h->f[O2[f]]; d.p[f] = h->f[O2[f] + 9]; }
    }
    if (level < 0 && !getenv("ATOMDEV_COPY_ONE_BY_ONE"))
        hipLaunchKernelGGL(k_copy_lists_pairs, dim3(blocks_for((n >> 1) + 1)), dim3(TPB), 0, h->stream, s, d, nf, n);
    else
        hipLaunchKernelGGL(k_copy_lists, dim3(blocks_for(n)), dim3(TPB), 0, h->stream, s, d, nf, off, n);
    ATOMDEV_TRY(hipGetLastError());
    return 0;
}

extern "C" int atomdev_atm_value_limitation(atomdev_t* h) {
    ATOMDEV_REQUIRE(h && h->model == ATOMDEV_ATM, "atomdev_atm_value_limitation: not an atmosphere handle");
    ATOMDEV_TRY(hipSetDevice(h->device));
    if (int rc = atomdev_refresh_flags(h)) return rc;
    hipLaunchKernelGGL(k_atm_value_limitation<false>, dim3(blocks_for(h->n3)), dim3(TPB), 0, h->stream, h->f[ATOMDEV_U],
                       h->f[ATOMDEV_V], h->f[ATOMDEV_W], h->f[ATOMDEV_T], h->f[ATOMDEV_C], h->f[ATOMDEV_CLOUD],
                       h->f[ATOMDEV_ICE], h->f[ATOMDEV_CO2], h->f[ATOMDEV_P_DYN], h->flags, h->n3, h->im, h->jm, h->km);
    ATOMDEV_TRY(hipGetLastError());
    return 0;
}

int atomdev_atm_limit_rest(atomdev* h) {
    hipLaunchKernelGGL(k_atm_value_limitation<true>, dim3(blocks_for(h->n3)), dim3(TPB), 0, h->stream, h->f[ATOMDEV_U],
                       h->f[ATOMDEV_V], h->f[ATOMDEV_W], h->f[ATOMDEV_T], h->f[ATOMDEV_C], h->f[ATOMDEV_CLOUD],
                       h->f[ATOMDEV_ICE], h->f[ATOMDEV_CO2], h->f[ATOMDEV_P_DYN], h->flags, h->n3, h->im, h->jm, h->km);
    ATOMDEV_TRY(hipGetLastError());
    return 0;
}

extern "C" int atomdev_atm_solid_ground(atomdev_t* h) {
    ATOMDEV_REQUIRE(h && h->model == ATOMDEV_ATM, "atomdev_atm_solid_ground: not an atmosphere handle");
    ATOMDEV_TRY(hipSetDevice(h->device));
    if (int rc = atomdev_refresh_flags(h)) return rc;
    size_t n = (size_t)(h->im - 1) * h->n2;
    hipLaunchKernelGGL(k_atm_solid_ground, dim3(blocks_for(n)), dim3(TPB), 0, h->stream, h->f[ATOMDEV_U], h->f[ATOMDEV_V],
                       h->f[ATOMDEV_W], h->f[ATOMDEV_CLOUD], h->f[ATOMDEV_ICE], h->f[ATOMDEV_CO2], h->f[ATOMDEV_P_DYN],
                       h->flags, n);
    ATOMDEV_TRY(hipGetLastError());
    return 0;
}

extern "C" int atomdev_hyd_value_limitation(atomdev_t* h) {
    ATOMDEV_REQUIRE(h && h->model == ATOMDEV_HYD, "atomdev_hyd_value_limitation: not a hydrosphere handle");
    ATOMDEV_TRY(hipSetDevice(h->device));
    hipLaunchKernelGGL(k_hyd_value_limitation, dim3(blocks_for(h->n3)), dim3(TPB), 0, h->stream, h->f[ATOMDEV_U],
                       h->f[ATOMDEV_V], h->f[ATOMDEV_W], h->f[ATOMDEV_T], h->f[ATOMDEV_C], h->n3);
    ATOMDEV_TRY(hipGetLastError());
    return 0;
}

extern "C" int atomdev_hyd_solid_ground(atomdev_t* h) {
    ATOMDEV_REQUIRE(h && h->model == ATOMDEV_HYD, "atomdev_hyd_solid_ground: not a hydrosphere handle");
    ATOMDEV_TRY(hipSetDevice(h->device));
    if (int rc = atomdev_refresh_flags(h)) return rc;
    static const int X[6] = { ATOMDEV_P_DYN, ATOMDEV_T, ATOMDEV_C, ATOMDEV_U, ATOMDEV_V, ATOMDEV_W };
    PtrList x, xn;
    for (int f = 0; f < 6; f++) { x.p[f] = h->f[X[f]]; xn.p[f] = h->f[X[f] + 9]; }
    hipLaunchKernelGGL(k_hyd_solid_ground, dim3(blocks_for(h->n3)), dim3(TPB), 0, h->stream, x, xn, h->p.pa, h->p.ta,
                       h->p.ca, h->flags, h->n3);
    ATOMDEV_TRY(hipGetLastError());
    return 0;
}
