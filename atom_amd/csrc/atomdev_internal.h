// Internal definitions shared by the HIP translation units of libatomdev.so.
#ifndef ATOMDEV_INTERNAL_H_
#define ATOMDEV_INTERNAL_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <functional>
#include <future>
#include <string>
#include <vector>

#include "atomdev.h"

// Point flags derived from h once per mask upload (k_flags.hip).
enum : uint8_t {
    FL_LAND = 1 << 0,      // |h - 1| < DBL_EPSILON  (AtomUtils::is_land)
    FL_ONESIDED_S = 1 << 1,// land cell with fluid at j-1 and j-2: one-sided theta differences
    FL_H0J = 1 << 2,       // fluid cell with land at j-1 or j+1  (h_0_j = 1)
    FL_H0K = 1 << 3,       // fluid cell with land at k-1 or k+1  (h_0_k = 1)
    FL_RDEAD = 1 << 4,     // hyd: land cell with h[i+1] == 0.: radial differences multiplied by h_d_i = 0
    FL_D2ZERO = 1 << 5,    // hyd: land cell with two fluid cells east or west: theta 2nd derivatives zeroed
    FL_HZERO = 1 << 6      // h == 0. exactly (the hydrosphere surface routine tests this)
};

// Scalars every kernel may need, computed on the host in the reference's own
// operation order so that they carry the same bits.
struct DevConst {
    int im, jm, km;
    double dt, dr, dthe, dphi;
    double dr2, dthe2, dphi2, two_dr, two_dthe, two_dphi;
    double re, re_pr, re_sc_wv, re_sc_co2, re_sc;   // divisors of the diffusion brackets
    double r_air, g, Buoyancy, t_0, c_0;
    double R_Air, R_WV_over_R_Air, gam, L, topo_step;
    double exp_pressure, p_SL_factor /* r_air*R_Air */, coeff_energy, coeff_buoy, coeff_trans, coeff_vapour;
    double coeff_salinity, buoy_g /* Buoyancy*g */, r_0_water;
    double t_tropopause, ta, pa, ca;
    // reciprocals (correctly rounded 1/x) for the exact-division helper
    double i_dthe, i_two_dthe, i_dthe2, i_dr, i_two_dr, i_dr2, i_dphi2, i_re, i_re_pr, i_re_sc_wv, i_re_sc_co2, i_re_sc,
           i_r_air, i_six, i_topo_step;
};

struct atomdev {
    int model, im, jm, km, device;
    size_t n3, n2;
    atomdev_params p;
    DevConst k;
    hipStream_t stream;
    hipEvent_t ev0, ev1;
    double* f[ATOMDEV_N3D];
    double* g[ATOMDEV_N2D];
    double* alt[ATOMDEV_N3D];  // hyd RK writes out of place into these, then swaps with f[]
    uint8_t* flags;
    bool flags_dirty;
    bool h0i_valid;                 // atm: scr[1] holds h_0_i for the current mask and topography
    double *rad, *sinthe, *costhe;  // device tables: rad[max(im,41)], sin/cos(the[j]) from the host libm
    double* plane;                  // jm*km scratch (atm: snapshot of c at level 3)
    void* red;                      // reduction scratch
    size_t red_bytes;
    double* scr[2];                 // 3-D scratch written by the RK prep kernels
    double *tab_rm2sin, *tab_irm2sin, *tab_coefv;  // [i*jm + j]: rm^2 sin(the), its reciprocal, viscous v coefficient
    void* batch;                    // device copy of the item list of a batched surface sweep (lazy)
    size_t batch_bytes;
    double* xfer;                   // six surface planes of the atmosphere -> hydrosphere transfer (lazy)
    void* surf;                     // skewed copy of the surface fields for the 2-D sweep (lazy)
    void* ws_params;                // device copy of the wave-specialised sweep's argument block(s) (lazy)
    void* pipe_params = nullptr;    // device copy of the pipelined sweep's argument block(s) (lazy)
    const double* pipe_cur0 = nullptr;  // hyd: f[T] at the time the pipelined sweep's blocks were built
    const double* ws_cur0;          // hyd: f[T] at the time the blocks were built (tells the buffer orientation)
    void* static_scratch;           // this GPU's copy of the library's static scratch array (k_accuracy.hip)
    size_t static_scratch_bytes;
    void* pinned;                   // 256 bytes of page-locked host memory: results read back without staging (first half),
                                    // the word kernels set on a failure they detect themselves (byte 128)
    void* stage;                    // device staging buffer of atomdev_pack_columns / _unpack_columns for host buffers (lazy)
    size_t stage_bytes;
    void* stage_host = nullptr;     // page-locked host mirror of `stage` for the text writers' section fetches (lazy)
    size_t stage_host_bytes = 0;
    int rk_variant;                 // 0: wave-specialised sweep, 1: one thread per theta line, 2: 0 in its tolerance form,
                                    // 3: the pipelined sweep (tolerance form; k_rk_atm_pipe.hip, k_rk_hyd_pipe.hip)
    int profile_rk;                 // bracket the RK sweeps with events: 1 the sweep kernel alone, 2 with the kernels that prepare it
    bool defer_files = false;         // inside a one-call slice: text files are finished by threads of their own
    std::vector<std::future<std::string>> pending_files;  // their error texts (empty: written)
    std::vector<std::string> pending_paths;               // ... and the files they write
    std::vector<void*> field_allocs;  // what hipMalloc returned for the 3-D fields (f[] and alt[] point into these, each at its own skew)
    std::vector<hipEvent_t> prof_ev;  // pairs (start, stop), prof_used of them recorded
    size_t prof_used;
};

// Event bracket around the RK sweep kernels when profiling is on (atomdev_api.hip).
int atomdev_prof_begin(atomdev* h);
int atomdev_prof_end(atomdev* h);

void atomdev_set_error(const std::string& msg);
int atomdev_check(hipError_t e, const char* what);
int atomdev_refresh_flags(atomdev* h);  // k_pointwise.hip
int atomdev_upload_mask_bytes(atomdev* h, const uint8_t* solid);  // k_pointwise.hip: h from a byte mask (1 = solid)
int atomdev_fill_theta_neumann(atomdev* h, int field);            // k_bc_pressure.hip
int atomdev_fill_phi_seam(atomdev* h, int field, int j0, int j1);  // k_bc_pressure.hip
int atomdev_static_scratch_init(atomdev* h);  // k_accuracy.hip
int atomdev_sync_stream(atomdev* h);          // atomdev_api.hip: hipStreamSynchronize + the kernels' own failure word
int atomdev_defer_file(atomdev* h, const std::string& path, std::function<std::string()> job);  // atomdev_api.hip
int atomdev_join_files(atomdev* h);                                   // atomdev_api.hip
inline volatile int* atomdev_kernel_error_word(atomdev* h) { return (volatile int*)((char*)h->pinned + 128); }
// limit: the values stored are what Value_Limitation_Atm would make of them (interior points of the eight variables);
// limit_before: the kernel that prepares the sweep runs Value_Limitation_Atm (and with it BC_SolidGround) on its way
int atomdev_atm_rk3d_ws(atomdev* h, bool limit, bool limit_before);    // k_rk_atm_ws.hip
int atomdev_atm_rk3d_ws_fast(atomdev* h, bool limit, bool limit_before);  // k_rk_atm_ws.hip compiled with -DATOMDEV_RK_FAST
int atomdev_atm_rk3d_pipe_launch(atomdev* h, bool limit);                 // k_rk_atm_pipe.hip: the sweep kernel alone
int atomdev_atm_limit_rest(atomdev* h);  // k_pointwise.hip: the limiter where the sweep does not store, and p_dyn on land
int atomdev_hyd_rk3d_ws(atomdev* h);    // k_rk_hyd_ws.hip
int atomdev_hyd_rk3d_ws_fast(atomdev* h);  // k_rk_hyd_ws.hip compiled with -DATOMDEV_RK_FAST
int atomdev_hyd_rk3d_pipe(atomdev* h);     // k_rk_hyd_pipe.hip

#define ATOMDEV_TRY(expr)                                   \
    do {                                                    \
        if (int rc__ = atomdev_check((expr), #expr)) return rc__; \
    } while (0)

#define ATOMDEV_REQUIRE(cond, msg)        \
    do {                                  \
        if (!(cond)) {                    \
            atomdev_set_error(msg);       \
            return 1;                     \
        }                                 \
    } while (0)

#ifdef __HIPCC__
// a / b for a divisor whose correctly rounded reciprocal rb is known: one product and two fused corrections (the
// residual a - q*b is exact), at a quarter of the cost of the generic IEEE division sequence.  This is not a theorem
// for every operand pair: there are divisors for which some dividend makes the corrected quotient differ from the
// correctly rounded one.  The divisors here are grid constants and table entries; that the result has the bits of
// a / b is checked on the device over 10^7 operand pairs (dividends over the fields' range, divisors over the
// constants' range, divisors with every significand bit set among them: tests/test_gpu_math.py) and, end to end, by
// the bit-exact parity tests of every kernel that uses it.  Finite, normal, non-overflowing operands only.
// Building with -DATOMDEV_TRUE_DIV replaces it by the hardware's division sequence.
__device__ __forceinline__ double div_by(double a, double b, double rb) {
#ifdef ATOMDEV_TRUE_DIV
    (void)rb;
    return a / b;
#else
    double q = a * rb;
    double r = __builtin_fma(-q, b, a);
    return __builtin_fma(r, rb, q);
#endif
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every
// outstanding global load and store of the wave (vmcnt(0)), which would drain the
// prefetches a sweep kernel keeps in flight across its stage barriers.  LDS operations
// of a wave complete in issue order, so waiting for them before the barrier is enough
// for the other waves to read what this wave wrote.
__device__ __forceinline__ void lds_barrier() {
#ifdef ATOMDEV_EXP_NOBARRIER  // timing experiment only: results are wrong
    return;
#endif
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__host__ __device__ __forceinline__ size_t idx3(int jm, int km, int i, int j, int k) {
    return ((size_t)i * (size_t)jm + (size_t)j) * (size_t)km + (size_t)k;
}
#endif

#endif  // ATOMDEV_INTERNAL_H_
