// Field store, handle lifetime, transfers, timers: the part of the C-ABI that
// replaces lib/Array.{h,cpp}, Array_2D and Array_1D for the hot path.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <string>
#include <vector>

#include "atomdev_internal.h"

static thread_local std::string g_last_error;

void atomdev_set_error(const std::string& msg) { g_last_error = msg; }

int atomdev_check(hipError_t e, const char* what) {
    if (e == hipSuccess) return 0;
    g_last_error = std::string(what) + ": " + hipGetErrorString(e);
    return 2;
}

extern "C" const char* atomdev_last_error(void) { return g_last_error.c_str(); }

// Reference defaults: param.py:32-177; grid constants atmosphere/cAtmosphereModel.cpp:38-55
// and hydrosphere/cHydrosphereModel.cpp:145-156.
extern "C" void atomdev_default_params(int model, atomdev_params* p) {
    memset(p, 0, sizeof(*p));
    const double pi180 = 180. / M_PI;
    p->dt = 0.00001; p->dr = 0.025; p->dthe = 1. / pi180; p->dphi = 1. / pi180;
    p->r0 = 1.; p->the0 = 0.; p->phi0 = 0.;
    p->g = 9.8066; p->Buoyancy = 1.0; p->t_0 = 273.15; p->p_0 = 1013.25;
    p->ep = .623; p->hp = 6.1078; p->lv = 2.52e6; p->ls = 2.83e6; p->r_water_vapour = .0094; p->co2_0 = 280.;
    if (model == ATOMDEV_ATM) { p->re = 1000.; p->pr = .7179; p->u_0 = 8.; p->c_0 = .035; }
    else { p->re = 10.; p->pr = 6.957; p->u_0 = .25; p->c_0 = 34.6; }
    p->sc_WaterVapour = .61; p->sc_CO2 = .96; p->gam = .65; p->r_air = 1.2041; p->R_Air = 287.1;
    p->R_WaterVapour = 461.6; p->L_atm = 16000.; p->cp_l = 1005.; p->t_tropopause = .798;
    p->sc = 1.7329; p->L_hyd = 1000.; p->r_0_water = 1000.; p->pa = 0.; p->ca = 1.;
    p->ta = model == ATOMDEV_ATM ? 1. : 1.01464;  // param.py:102, 163
}

static bool field_used(int model, int f) {
    if (f < 0 || f >= ATOMDEV_N3D) return false;
    if (model == ATOMDEV_ATM) return !(f >= ATOMDEV_SALT_BALANCE && f <= ATOMDEV_R_SALT_WATER);
    if (f >= ATOMDEV_Q_LATENT && f <= ATOMDEV_S_C_C) return false;
    // hydrosphere: no cloud / ice / co2 / S_*
    if (f == ATOMDEV_CLOUD || f == ATOMDEV_ICE || f == ATOMDEV_CO2) return false;
    if (f == ATOMDEV_CLOUDN || f == ATOMDEV_ICEN || f == ATOMDEV_CO2N) return false;
    if (f >= ATOMDEV_S_V && f <= ATOMDEV_S_S) return false;
    return true;
}

static void fill_constants(atomdev* h) {
    const atomdev_params& P = h->p;
    DevConst& k = h->k;
    const bool atm = h->model == ATOMDEV_ATM;
    k.im = h->im; k.jm = h->jm; k.km = h->km;
    k.dt = P.dt; k.dr = P.dr; k.dthe = P.dthe; k.dphi = P.dphi;
    k.dr2 = P.dr * P.dr; k.dthe2 = P.dthe * P.dthe; k.dphi2 = P.dphi * P.dphi;
    k.two_dr = 2. * P.dr; k.two_dthe = 2. * P.dthe; k.two_dphi = 2. * P.dphi;
    k.re = P.re; k.re_pr = P.re * P.pr;
    k.re_sc_wv = P.sc_WaterVapour * P.re; k.re_sc_co2 = P.sc_CO2 * P.re; k.re_sc = P.sc * P.re;
    k.r_air = P.r_air; k.g = P.g; k.Buoyancy = P.Buoyancy; k.t_0 = P.t_0; k.c_0 = P.c_0;
    k.R_Air = P.R_Air; k.R_WV_over_R_Air = P.R_WaterVapour / P.R_Air; k.gam = P.gam;
    k.L = atm ? P.L_atm : P.L_hyd;
    k.topo_step = k.L / (double)(h->im - 1);
    k.exp_pressure = P.g / (1.e-2 * P.gam * P.R_Air);
    k.p_SL_factor = P.r_air * P.R_Air;
    k.coeff_energy = P.L_atm / (P.cp_l * P.t_0 * P.u_0);
    k.coeff_buoy = atm ? P.r_air * P.u_0 * P.u_0 / P.L_atm : P.r_0_water * P.u_0 * P.u_0 / P.L_hyd;
    k.coeff_trans = P.L_atm / P.u_0;
    k.coeff_vapour = 1.1574e-5 * P.L_atm / P.u_0;
    k.coeff_salinity = 1.1574e-5 * P.L_hyd / P.u_0;
    k.buoy_g = P.Buoyancy * P.g;
    k.r_0_water = P.r_0_water;
    k.t_tropopause = P.t_tropopause; k.ta = P.ta; k.pa = P.pa; k.ca = P.ca;
    k.i_dthe = 1. / k.dthe; k.i_two_dthe = 1. / k.two_dthe; k.i_dthe2 = 1. / k.dthe2;
    k.i_dr = 1. / k.dr; k.i_two_dr = 1. / k.two_dr; k.i_dr2 = 1. / k.dr2; k.i_dphi2 = 1. / k.dphi2;
    k.i_re = 1. / k.re; k.i_re_pr = 1. / k.re_pr; k.i_re_sc_wv = 1. / k.re_sc_wv;
    k.i_re_sc_co2 = 1. / k.re_sc_co2; k.i_re_sc = 1. / k.re_sc;
    k.i_r_air = 1. / k.r_air; k.i_six = 1. / 6.; k.i_topo_step = 1. / k.topo_step;
}

// Coordinate and geometry tables of a handle, (re)built from its parameters.
static int upload_tables(atomdev* h) {
    const atomdev_params* p = &h->p;
    const int im = h->im, jm = h->jm, model = h->model;
    // coordinates by repeated addition (lib/Array_1D.cpp:54-65); sin/cos from the host
    // libm, which is what the reference calls per grid point (e.g. RHS_Atm.cpp:101-103)
    const int nrad = im > 41 ? im : 41;  // RK_RHS_2D_Hydrosphere reads rad[40] on any grid
    std::vector<double> rad(nrad), st(jm), ct(jm);
    rad[0] = p->r0;
    for (int l = 1; l < nrad; l++) rad[l] = rad[l - 1] + p->dr;
    // sin and cos in loops of their own: a compiler that sees both calls on one argument may merge them
    // into sincos(), whose sine differs from sin() in the last place for some angles
    std::vector<double> the(jm);
    the[0] = p->the0;
    for (int j = 1; j < jm; j++) the[j] = the[j - 1] + p->dthe;
    for (int j = 0; j < jm; j++) st[j] = sin(the[j]);
    for (int j = 0; j < jm; j++) ct[j] = cos(the[j]);
    const size_t nb = (size_t)im * jm * sizeof(double);
    if (!h->rad) ATOMDEV_TRY(hipMalloc(&h->rad, nrad * sizeof(double)));
    if (!h->sinthe) ATOMDEV_TRY(hipMalloc(&h->sinthe, jm * sizeof(double)));
    if (!h->costhe) ATOMDEV_TRY(hipMalloc(&h->costhe, jm * sizeof(double)));
    if (!h->tab_rm2sin) ATOMDEV_TRY(hipMalloc(&h->tab_rm2sin, nb));
    if (!h->tab_irm2sin) ATOMDEV_TRY(hipMalloc(&h->tab_irm2sin, nb));
    if (!h->tab_coefv) ATOMDEV_TRY(hipMalloc(&h->tab_coefv, nb));
    ATOMDEV_TRY(hipMemcpy(h->rad, rad.data(), nrad * sizeof(double), hipMemcpyHostToDevice));
    ATOMDEV_TRY(hipMemcpy(h->sinthe, st.data(), jm * sizeof(double), hipMemcpyHostToDevice));
    ATOMDEV_TRY(hipMemcpy(h->costhe, ct.data(), jm * sizeof(double), hipMemcpyHostToDevice));
    // per-(i, j) geometry factors of the RK sweeps, in the reference's operation order
    // (RHS_Atm.cpp:97-106, 450; RHS_Hyd.cpp:83-92, 422); 1/x is the correctly rounded reciprocal
    std::vector<double> t1((size_t)im * jm), t2((size_t)im * jm), t3((size_t)im * jm);
    for (int i = 0; i < im; i++)
        for (int j = 0; j < jm; j++) {
            const double rm = rad[i], rm2 = rm * rm, sinthe = st[j], costhe = ct[j], sinthe2 = sinthe * sinthe;
            const double rm2sinthe = rm2 * sinthe;
            t1[(size_t)i * jm + j] = rm2sinthe;
            t2[(size_t)i * jm + j] = 1. / rm2sinthe;
            t3[(size_t)i * jm + j] = model == ATOMDEV_ATM ? 1. + costhe * costhe / (rm * sinthe2)
                                                          : 1. + costhe * costhe / sinthe2;
        }
    ATOMDEV_TRY(hipMemcpy(h->tab_rm2sin, t1.data(), nb, hipMemcpyHostToDevice));
    ATOMDEV_TRY(hipMemcpy(h->tab_irm2sin, t2.data(), nb, hipMemcpyHostToDevice));
    ATOMDEV_TRY(hipMemcpy(h->tab_coefv, t3.data(), nb, hipMemcpyHostToDevice));
    return 0;
}

extern "C" int atomdev_create(int model, int im, int jm, int km, const atomdev_params* p, int device,
                              atomdev_t** out) {
    ATOMDEV_REQUIRE(out != nullptr && p != nullptr, "atomdev_create: null argument");
    *out = nullptr;
    ATOMDEV_REQUIRE(model == ATOMDEV_ATM || model == ATOMDEV_HYD, "atomdev_create: unknown model");
    // the boundary fills read levels 3 and im-4 and rows 2 and jm-3
    ATOMDEV_REQUIRE(im >= 5 && jm >= 7 && km >= 7, "atomdev_create: grid too small (need im>=5, jm>=7, km>=7)");
    ATOMDEV_REQUIRE(p->dt > 0 && p->dr > 0 && p->dthe > 0 && p->dphi > 0, "atomdev_create: step sizes must be positive");
    int ndev = 0;
    ATOMDEV_TRY(hipGetDeviceCount(&ndev));
    ATOMDEV_REQUIRE(ndev > 0, "atomdev_create: no HIP device visible");
    if (device < 0) ATOMDEV_TRY(hipGetDevice(&device));
    ATOMDEV_REQUIRE(device < ndev, "atomdev_create: device index out of range");
    ATOMDEV_TRY(hipSetDevice(device));

    atomdev* h = new atomdev();
    memset(h->f, 0, sizeof(h->f)); memset(h->g, 0, sizeof(h->g)); memset(h->alt, 0, sizeof(h->alt));
    h->model = model; h->im = im; h->jm = jm; h->km = km; h->device = device;
    h->n3 = (size_t)im * jm * km; h->n2 = (size_t)jm * km;
    h->p = *p;
    h->flags = nullptr; h->flags_dirty = true; h->h0i_valid = false;
    h->profile_rk = 0; h->prof_used = 0;
    h->scr[0] = h->scr[1] = nullptr; h->tab_rm2sin = h->tab_irm2sin = h->tab_coefv = nullptr;
    // A handle starts on a tolerance form of the sweep (every field within 1e-10 of the reference after a time slice,
    // iteration counts identical): the hydrosphere on the pipelined one, the atmosphere on the wave-specialised one (its
    // pipelined form is as close per sweep, but the column physics of a whole time slice amplifies what it leaves to
    // 4e-10 in c).  The exact forms, whose hydrosphere results carry the reference's bits, are a call of
    // atomdev_set_rk_variant away, and the models' outer interfaces make that call.
    h->rk_variant = model == ATOMDEV_ATM ? 2 : 3;
    h->ws_params = nullptr; h->surf = nullptr; h->xfer = nullptr; h->batch = nullptr; h->batch_bytes = 0;
    if (const char* e = getenv("ATOMDEV_RK_VARIANT")) {
        const int v = atoi(e);
        if (v >= 0 && v <= 3) h->rk_variant = v;
    }
    h->rad = h->sinthe = h->costhe = h->plane = nullptr; h->red = nullptr;
    h->static_scratch = nullptr; h->static_scratch_bytes = 0; h->pinned = nullptr;
    h->stage = nullptr; h->stage_bytes = 0;
    fill_constants(h);

    int rc = 0;
    auto fail = [&](int code) { atomdev_destroy(h); return code; };
    if ((rc = atomdev_check(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking), "hipStreamCreate"))) { h->stream = nullptr; return fail(rc); }
    if ((rc = atomdev_check(hipEventCreate(&h->ev0), "hipEventCreate"))) return fail(rc);
    if ((rc = atomdev_check(hipEventCreate(&h->ev1), "hipEventCreate"))) return fail(rc);
    // Every 3-D field starts at its own distance from the allocation's (2 MiB-aligned) start.  A sweep streams through
    // thirty fields at the same index at once; with all of them at the same address modulo a large power of two, the
    // requests of a row would keep landing on the same memory channel and bank.
    size_t skew_unit = 0;
    if (const char* e = getenv("ATOMDEV_FIELD_SKEW")) skew_unit = (size_t)atol(e) & ~(size_t)255;
    int nalloc = 0;
    auto alloc_field = [&](double** out) -> int {
        void* raw = nullptr;
        const size_t skew = skew_unit * (size_t)nalloc++;
        if (int rc2 = atomdev_check(hipMalloc(&raw, h->n3 * sizeof(double) + skew), "hipMalloc(field)")) return rc2;
        h->field_allocs.push_back(raw);
        *out = (double*)((char*)raw + skew);
        return 0;
    };
    for (int f = 0; f < ATOMDEV_N3D; f++) {
        if (!field_used(model, f)) continue;
        if ((rc = alloc_field(&h->f[f]))) return fail(rc);
        if ((rc = atomdev_check(hipMemsetAsync(h->f[f], 0, h->n3 * sizeof(double), h->stream), "hipMemset"))) return fail(rc);
    }
    if (model == ATOMDEV_HYD) {
        const int pp[5] = { ATOMDEV_T, ATOMDEV_U, ATOMDEV_V, ATOMDEV_W, ATOMDEV_C };
        for (int n = 0; n < 5; n++)
            if ((rc = alloc_field(&h->alt[pp[n]]))) return fail(rc);
    }
    for (int g = 0; g < ATOMDEV_N2D; g++) {
        if ((rc = atomdev_check(hipMalloc(&h->g[g], h->n2 * sizeof(double)), "hipMalloc(field2d)"))) return fail(rc);
        if ((rc = atomdev_check(hipMemsetAsync(h->g[g], 0, h->n2 * sizeof(double), h->stream), "hipMemset"))) return fail(rc);
    }
    if ((rc = atomdev_check(hipMalloc(&h->flags, h->n3), "hipMalloc(flags)"))) return fail(rc);
    if ((rc = atomdev_check(hipMalloc(&h->plane, h->n2 * sizeof(double)), "hipMalloc(plane)"))) return fail(rc);
    h->red_bytes = 1 << 20;
    if ((rc = atomdev_check(hipMalloc(&h->red, h->red_bytes), "hipMalloc(red)"))) return fail(rc);

    if ((rc = atomdev_check(hipHostMalloc(&h->pinned, 256, hipHostMallocDefault), "hipHostMalloc(pinned)"))) return fail(rc);
    memset(h->pinned, 0, 256);
    if ((rc = atomdev_static_scratch_init(h))) return fail(rc);
    if ((rc = upload_tables(h))) return fail(rc);
    for (int n = 0; n < 2; n++)
        if ((rc = atomdev_check(hipMalloc(&h->scr[n], h->n3 * sizeof(double)), "hipMalloc(scr)"))) return fail(rc);
    if ((rc = atomdev_check(hipStreamSynchronize(h->stream), "hipStreamSynchronize"))) return fail(rc);
    *out = h;
    return 0;
}

extern "C" int atomdev_set_params(atomdev_t* h, const atomdev_params* p) {
    ATOMDEV_REQUIRE(h != nullptr && p != nullptr, "atomdev_set_params: null argument");
    ATOMDEV_REQUIRE(p->dt > 0 && p->dr > 0 && p->dthe > 0 && p->dphi > 0, "atomdev_set_params: step sizes must be positive");
    ATOMDEV_TRY(hipSetDevice(h->device));
    if (int rc__ = atomdev_sync_stream(h)) return rc__;
    h->p = *p;
    fill_constants(h);
    if (int rc = upload_tables(h)) return rc;
    if (h->ws_params) { ATOMDEV_TRY(hipFree(h->ws_params)); h->ws_params = nullptr; }  // holds a copy of the constants
    if (h->pipe_params) { ATOMDEV_TRY(hipFree(h->pipe_params)); h->pipe_params = nullptr; }
    h->h0i_valid = false;
    return 0;
}

extern "C" int atomdev_destroy(atomdev_t* h) {
    if (!h) return 0;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    for (void* raw : h->field_allocs) (void)hipFree(raw);
    for (int g = 0; g < ATOMDEV_N2D; g++) if (h->g[g]) (void)hipFree(h->g[g]);
    if (h->flags) (void)hipFree(h->flags);
    if (h->plane) (void)hipFree(h->plane);
    if (h->red) (void)hipFree(h->red);
    if (h->pinned) (void)hipHostFree(h->pinned);
    if (h->stage) (void)hipFree(h->stage);
    if (h->stage_host) (void)hipHostFree(h->stage_host);
    if (h->rad) (void)hipFree(h->rad);
    if (h->sinthe) (void)hipFree(h->sinthe);
    if (h->costhe) (void)hipFree(h->costhe);
    if (h->tab_rm2sin) (void)hipFree(h->tab_rm2sin);
    if (h->tab_irm2sin) (void)hipFree(h->tab_irm2sin);
    if (h->tab_coefv) (void)hipFree(h->tab_coefv);
    if (h->ws_params) (void)hipFree(h->ws_params);
    if (h->pipe_params) (void)hipFree(h->pipe_params);
    if (h->surf) (void)hipFree(h->surf);
    if (h->xfer) (void)hipFree(h->xfer);
    if (h->batch) (void)hipFree(h->batch);
    for (int n = 0; n < 2; n++) if (h->scr[n]) (void)hipFree(h->scr[n]);
    for (hipEvent_t e : h->prof_ev) (void)hipEventDestroy(e);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return 0;
}

extern "C" int atomdev_dims(const atomdev_t* h, int* im, int* jm, int* km) {
    ATOMDEV_REQUIRE(h != nullptr, "atomdev_dims: null handle");
    if (im) *im = h->im;
    if (jm) *jm = h->jm;
    if (km) *km = h->km;
    return 0;
}

static int check3d(const atomdev_t* h, int field, const void* host, const char* who) {
    if (!h || !host) { atomdev_set_error(std::string(who) + ": null argument"); return 1; }
    if (field < 0 || field >= ATOMDEV_N3D || !h->f[field]) {
        atomdev_set_error(std::string(who) + ": field " + std::to_string(field) + " does not exist in this model");
        return 1;
    }
    return 0;
}

extern "C" int atomdev_upload(atomdev_t* h, int field, const double* host) {
    if (int rc = check3d(h, field, host, "atomdev_upload")) return rc;
    ATOMDEV_TRY(hipSetDevice(h->device));
    ATOMDEV_TRY(hipMemcpyAsync(h->f[field], host, h->n3 * sizeof(double), hipMemcpyHostToDevice, h->stream));
    if (int rc__ = atomdev_sync_stream(h)) return rc__;  // the host buffer may be reused at once
    if (field == ATOMDEV_H) h->flags_dirty = true;
    return 0;
}

extern "C" int atomdev_download(atomdev_t* h, int field, double* host) {
    if (int rc = check3d(h, field, host, "atomdev_download")) return rc;
    ATOMDEV_TRY(hipSetDevice(h->device));
    ATOMDEV_TRY(hipMemcpyAsync(host, h->f[field], h->n3 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    if (int rc__ = atomdev_sync_stream(h)) return rc__;
    return 0;
}

extern "C" int atomdev_upload2d(atomdev_t* h, int field, const double* host) {
    ATOMDEV_REQUIRE(h && host && field >= 0 && field < ATOMDEV_N2D, "atomdev_upload2d: bad argument");
    ATOMDEV_TRY(hipSetDevice(h->device));
    ATOMDEV_TRY(hipMemcpyAsync(h->g[field], host, h->n2 * sizeof(double), hipMemcpyHostToDevice, h->stream));
    if (int rc__ = atomdev_sync_stream(h)) return rc__;
    if (field == ATOMDEV_TOPOGRAPHY) h->h0i_valid = false;
    return 0;
}

extern "C" int atomdev_download2d(atomdev_t* h, int field, double* host) {
    ATOMDEV_REQUIRE(h && host && field >= 0 && field < ATOMDEV_N2D, "atomdev_download2d: bad argument");
    ATOMDEV_TRY(hipSetDevice(h->device));
    ATOMDEV_TRY(hipMemcpyAsync(host, h->g[field], h->n2 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    if (int rc__ = atomdev_sync_stream(h)) return rc__;
    return 0;
}

// Columns k0 .. k0+nk-1 of a 3-D field <-> a dense [im*jm][nk] buffer in host or device memory.
static int copy_columns(atomdev_t* h, int field, int k0, int nk, void* buf, bool to_buf, const char* who) {
    ATOMDEV_REQUIRE(h && buf && field >= 0 && field < ATOMDEV_N3D && h->f[field], std::string(who) + ": bad argument");
    ATOMDEV_REQUIRE(k0 >= 0 && nk >= 1 && k0 + nk <= h->km, std::string(who) + ": column range outside the grid");
    ATOMDEV_TRY(hipSetDevice(h->device));
    const size_t rows = (size_t)h->im * h->jm, pitch = (size_t)h->km * sizeof(double), w = (size_t)nk * sizeof(double);
    double* f = h->f[field] + k0;
    if (to_buf) ATOMDEV_TRY(hipMemcpy2DAsync(buf, w, f, pitch, w, rows, hipMemcpyDefault, h->stream));
    else ATOMDEV_TRY(hipMemcpy2DAsync(f, pitch, buf, w, w, rows, hipMemcpyDefault, h->stream));
    if (int rc__ = atomdev_sync_stream(h)) return rc__;
    if (!to_buf && field == ATOMDEV_H) h->flags_dirty = true;
    return 0;
}

// ---- several fields' columns in one message buffer ----
struct PackArgs { double* f[16]; };
template <bool PACK>
__global__ void k_pack_columns(PackArgs a, double* __restrict__ buf, size_t rows, int km, int k0, int nk) {
    const size_t per_field = rows * nk;
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= per_field) return;
    const size_t row = q / nk;
    const int kk = (int)(q % nk);
    double* x = a.f[blockIdx.y] + row * km + k0 + kk;
    double* b = buf + (size_t)blockIdx.y * per_field + q;
    if (PACK) *b = *x; else *x = *b;
}

static int pack_columns(atomdev_t* h, const int* fields, int nf, int k0, int nk, void* buf, bool pack, const char* who) {
    ATOMDEV_REQUIRE(h && fields && buf && nf >= 1 && nf <= 16, std::string(who) + ": bad argument");
    ATOMDEV_REQUIRE(k0 >= 0 && nk >= 1 && k0 + nk <= h->km, std::string(who) + ": column range outside the grid");
    ATOMDEV_TRY(hipSetDevice(h->device));
    PackArgs a;
    for (int n = 0; n < nf; n++) {
        ATOMDEV_REQUIRE(fields[n] >= 0 && fields[n] < ATOMDEV_N3D && h->f[fields[n]], std::string(who) + ": a field this model does not have");
        a.f[n] = h->f[fields[n]];
        if (!pack && fields[n] == ATOMDEV_H) h->flags_dirty = true;
    }
    const size_t rows = (size_t)h->im * h->jm, bytes = (size_t)nf * rows * nk * sizeof(double);
    hipPointerAttribute_t attr;
    const bool on_device = hipPointerGetAttributes(&attr, buf) == hipSuccess && attr.type == hipMemoryTypeDevice;
    if (!on_device) (void)hipGetLastError();  // plain host memory is "not found": clear the sticky error
    double* dev = (double*)buf;
    if (!on_device) {  // host buffer: one contiguous copy through a staging buffer of the handle
        if (bytes > h->stage_bytes) {
            if (int rc__ = atomdev_sync_stream(h)) return rc__;
            if (h->stage) ATOMDEV_TRY(hipFree(h->stage));
            h->stage = nullptr; h->stage_bytes = 0;
            ATOMDEV_TRY(hipMalloc(&h->stage, bytes));
            h->stage_bytes = bytes;
        }
        dev = (double*)h->stage;
        if (!pack) ATOMDEV_TRY(hipMemcpyAsync(dev, buf, bytes, hipMemcpyHostToDevice, h->stream));
    }
    const dim3 grid((unsigned)((rows * nk + 255) / 256), nf);
    if (pack) hipLaunchKernelGGL(k_pack_columns<true>, grid, dim3(256), 0, h->stream, a, dev, rows, h->km, k0, nk);
    else hipLaunchKernelGGL(k_pack_columns<false>, grid, dim3(256), 0, h->stream, a, dev, rows, h->km, k0, nk);
    ATOMDEV_TRY(hipGetLastError());
    if (!on_device) {
        if (pack) ATOMDEV_TRY(hipMemcpyAsync(buf, dev, bytes, hipMemcpyDeviceToHost, h->stream));
        return atomdev_sync_stream(h);  // the host buffer is complete / may be reused
    }
    return 0;  // device buffer: ordered on the handle's stream, the caller waits once (atomdev_sync) before it hands the buffer on
}

extern "C" int atomdev_pack_columns(atomdev_t* h, const int* fields3d, int nf, int k0, int nk, void* dst) {
    return pack_columns(h, fields3d, nf, k0, nk, dst, true, "atomdev_pack_columns");
}
extern "C" int atomdev_unpack_columns(atomdev_t* h, const int* fields3d, int nf, int k0, int nk, const void* src) {
    return pack_columns(h, fields3d, nf, k0, nk, const_cast<void*>(src), false, "atomdev_unpack_columns");
}

extern "C" int atomdev_get_columns(atomdev_t* h, int field3d, int k0, int nk, void* dst) {
    return copy_columns(h, field3d, k0, nk, dst, true, "atomdev_get_columns");
}

extern "C" int atomdev_set_columns(atomdev_t* h, int field3d, int k0, int nk, const void* src) {
    return copy_columns(h, field3d, k0, nk, const_cast<void*>(src), false, "atomdev_set_columns");
}

// Waits for the handle's stream and then looks at the word in page-locked host memory that a kernel sets when it
// could not finish its work properly (the hydrosphere sweep's wait between workgroups).  Every entry point that hands
// data to the host or reports completion waits through this function, so such a failure surfaces with the first
// result that could contain it, once.
int atomdev_sync_stream(atomdev* h) {
    ATOMDEV_TRY(hipStreamSynchronize(h->stream));
    volatile int* err = atomdev_kernel_error_word(h);
    if (const int code = *err) {
        *err = 0;
        atomdev_set_error(code == 3 ? "atm_rk3d: a wave of the pipelined sweep gave up waiting for another wave of its workgroup; the fields of the last sweep are not valid"
                          : code == 2 ? "rk2d: a wave of the surface sweep gave up waiting for the row above it; the surface fields of the last sweep are not valid"
                                    : "hyd_rk3d: level im-2 gave up waiting for the salinity rows of level im-3; the fields of the last sweep are not valid");
        return 1;
    }
    return 0;
}

// Text files whose numbers are on the host already: inside a one-call slice (defer_files) the text is made and written
// by a thread of its own while the slice goes on, and atomdev_join_files waits for all of them; otherwise at once.
// Two jobs for one path run one after the other, in the order they were handed in (the reference writes its files one
// after the other, so where a checkpoint and the final write fall on the same name the later content stands).
int atomdev_defer_file(atomdev* h, const std::string& path, std::function<std::string()> job) {
    if (h->defer_files) {
        for (size_t n = 0; n < h->pending_files.size(); n++)
            if (h->pending_paths[n] == path && h->pending_files[n].valid()) h->pending_files[n].wait();
        h->pending_files.push_back(std::async(std::launch::async, std::move(job)));
        h->pending_paths.push_back(path);
        return 0;
    }
    const std::string e = job();
    if (!e.empty()) { atomdev_set_error(e); return 1; }
    return 0;
}

int atomdev_join_files(atomdev* h) {
    std::string first;
    for (std::future<std::string>& f : h->pending_files) {
        const std::string e = f.get();
        if (first.empty()) first = e;
    }
    h->pending_files.clear();
    h->pending_paths.clear();
    if (!first.empty()) { atomdev_set_error(first); return 1; }
    return 0;
}

extern "C" int atomdev_sync(atomdev_t* h) {
    ATOMDEV_REQUIRE(h != nullptr, "atomdev_sync: null handle");
    ATOMDEV_TRY(hipSetDevice(h->device));
    return atomdev_sync_stream(h);
}

extern "C" int atomdev_timer_start(atomdev_t* h) {
    ATOMDEV_REQUIRE(h != nullptr, "atomdev_timer_start: null handle");
    ATOMDEV_TRY(hipSetDevice(h->device));
    ATOMDEV_TRY(hipEventRecord(h->ev0, h->stream));
    return 0;
}

extern "C" int atomdev_timer_stop(atomdev_t* h, float* ms) {
    ATOMDEV_REQUIRE(h != nullptr && ms != nullptr, "atomdev_timer_stop: null argument");
    ATOMDEV_TRY(hipSetDevice(h->device));
    ATOMDEV_TRY(hipEventRecord(h->ev1, h->stream));
    ATOMDEV_TRY(hipEventSynchronize(h->ev1));
    ATOMDEV_TRY(hipEventElapsedTime(ms, h->ev0, h->ev1));
    return 0;
}

static int prof_record(atomdev* h) {
    if (h->prof_used == h->prof_ev.size()) {
        hipEvent_t e;
        ATOMDEV_TRY(hipEventCreate(&e));
        h->prof_ev.push_back(e);
    }
    ATOMDEV_TRY(hipEventRecord(h->prof_ev[h->prof_used], h->stream));
    h->prof_used++;
    return 0;
}

int atomdev_prof_begin(atomdev* h) { return h->profile_rk ? prof_record(h) : 0; }
int atomdev_prof_end(atomdev* h) { return h->profile_rk ? prof_record(h) : 0; }

extern "C" int atomdev_profile_rk(atomdev_t* h, int enable) {
    ATOMDEV_REQUIRE(h != nullptr, "atomdev_profile_rk: null handle");
    ATOMDEV_TRY(hipSetDevice(h->device));
    if (int rc__ = atomdev_sync_stream(h)) return rc__;
    h->profile_rk = enable;
    h->prof_used = 0;
    return 0;
}

extern "C" int atomdev_profile_read(atomdev_t* h, float* total_ms, int* sweeps) {
    ATOMDEV_REQUIRE(h && total_ms && sweeps, "atomdev_profile_read: null argument");
    ATOMDEV_TRY(hipSetDevice(h->device));
    if (int rc__ = atomdev_sync_stream(h)) return rc__;
    float sum = 0.f;
    for (size_t i = 0; i + 1 < h->prof_used; i += 2) {
        float ms = 0.f;
        ATOMDEV_TRY(hipEventElapsedTime(&ms, h->prof_ev[i], h->prof_ev[i + 1]));
        sum += ms;
    }
    *total_ms = sum;
    *sweeps = (int)(h->prof_used / 2);
    h->prof_used = 0;
    return 0;
}

extern "C" int atomdev_get_rk_variant(const atomdev_t* h) { return h ? h->rk_variant : -1; }

extern "C" int atomdev_set_rk_variant(atomdev_t* h, int variant) {
    ATOMDEV_REQUIRE(h != nullptr, "atomdev_set_rk_variant: null handle");
    ATOMDEV_REQUIRE(variant >= 0 && variant <= 3, "atomdev_set_rk_variant: variant must be 0, 1, 2 or 3");
    h->rk_variant = variant;
    return 0;
}
