// The ASCII VTK sections of both models: PostProcess_Atmosphere::paraview_vtk_radial / _longal / _zonal
// (atmosphere/PostProcess_Atm.cpp:240-621) and PostProcess_Hydrosphere::paraview_vtk_longal / _radial / _zonal
// (hydrosphere/PostProcess_Hyd.cpp:373-636).  Host code over downloads of the section's slices: one table of
// columns per file, the derived columns computed as the reference computes them (with the host's libm, as there),
// and what the reference leaves behind in aux_u / aux_v / aux_w written back to the device fields.
#include <cmath>
#include <algorithm>
#include <cstdio>
#include <memory>
#include <string>
#include <vector>

#include "atomdev_internal.h"
#include "fixed4.h"
#include "host_parallel.h"

namespace {

// element r of a contiguous buffer <-> element r * pitch of a field (scatter != 0: buffer to field)
__global__ void k_section_column(double* __restrict__ buf, double* __restrict__ field, size_t pitch, size_t rows, int scatter) {
    const size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    if (scatter) field[r * pitch] = buf[r]; else buf[r] = field[r * pitch];
}

// rows of `width` doubles, `pitch` doubles apart, as one dense run (a section of a 3-D field in the writers' order)
__global__ void k_section_gather(double* __restrict__ dst, const double* __restrict__ src, size_t width, size_t pitch, size_t count) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= count) return;
    dst[e] = src[(e / width) * pitch + e % width];
}

// What the radial writer of the atmosphere leaves behind at Ma = 0 (PostProcess_Atm.cpp:287-309), one thread per column:
// the Celsius copy of the NASA temperature, its difference to the ground temperature where the column has air at the
// ground or a land cell below an air cell (in aux_v at the section's level), evaporation - precipitation (in aux_w).
__global__ void k_radial_nasa_terms(const double* __restrict__ t_nasa, const double* __restrict__ t, const double* __restrict__ evap,
                                    const double* __restrict__ precip, const uint8_t* __restrict__ flags, double* __restrict__ celsius,
                                    double* __restrict__ aux_v, double* __restrict__ aux_w, double t_0, int im, size_t n2) {
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n2) return;
    celsius[q] = t_nasa[q] * t_0 - t_0;
    bool assigned = !(flags[q] & FL_LAND);  // is_air(h, 0, j, k)
    for (int i = im - 2; i >= 0 && !assigned; i--)
        if (!(flags[(size_t)(i + 1) * n2 + q] & FL_LAND) && (flags[(size_t)i * n2 + q] & FL_LAND)) assigned = true;
    if (assigned) aux_v[q] = (t[q] - t_nasa[q]) * t_0;
    aux_w[q] = evap[q] - precip[q];
}

// one section of a 3-D field in the order the writers walk it: radial [j][k], longal [i][k], zonal [i][j]
struct Section {
    atomdev* h;
    int kind, index, n_outer, n_inner;
    Section(atomdev* h_, int kind_, int index_) : h(h_), kind(kind_), index(index_) {
        n_outer = kind == ATOMDEV_RADIAL ? h->jm : h->im;
        n_inner = kind == ATOMDEV_ZONAL ? h->jm : h->km;
    }
    size_t size() const { return (size_t)n_outer * n_inner; }
    // source geometry of the section inside a field: rows of `width` bytes, `pitch` bytes apart
    void geometry(size_t& offset, size_t& width, size_t& pitch, size_t& rows) const {
        const size_t km = h->km, jm = h->jm;
        if (kind == ATOMDEV_RADIAL) { offset = (size_t)index * jm * km; width = jm * km * 8; pitch = width; rows = 1; }
        else if (kind == ATOMDEV_LONGAL) { offset = (size_t)index * km; width = km * 8; pitch = jm * km * 8; rows = h->im; }
        else { offset = (size_t)index; width = 8; pitch = km * 8; rows = (size_t)h->im * jm; }
    }
    // A zonal section is one double out of every row of the field: as a strided copy that is a transfer per
    // element, so it is gathered into (scattered from) a contiguous device buffer by a kernel instead.
    int staging(double*& dev) const {
        const size_t bytes = size() * sizeof(double);
        if (bytes > h->stage_bytes) {
            ATOMDEV_TRY(hipStreamSynchronize(h->stream));
            if (h->stage) ATOMDEV_TRY(hipFree(h->stage));
            h->stage = nullptr; h->stage_bytes = 0;
            ATOMDEV_TRY(hipMalloc(&h->stage, bytes));
            h->stage_bytes = bytes;
        }
        dev = (double*)h->stage;
        return 0;
    }
    int get(int field, std::vector<double>& out) const {
        out.resize(size());
        size_t off, width, pitch, rows;
        geometry(off, width, pitch, rows);
        if (kind == ATOMDEV_ZONAL) {
            double* dev;
            if (int rc = staging(dev)) return rc;
            hipLaunchKernelGGL(k_section_column, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, h->stream, dev, h->f[field] + off,
                               (size_t)h->km, rows, 0);
            ATOMDEV_TRY(hipGetLastError());
            ATOMDEV_TRY(hipMemcpyAsync(out.data(), dev, rows * 8, hipMemcpyDeviceToHost, h->stream));
            return 0;
        }
        ATOMDEV_TRY(hipMemcpy2DAsync(out.data(), width, h->f[field] + off, pitch, width, rows, hipMemcpyDeviceToHost, h->stream));
        return 0;
    }
    // Several columns of a file at once: every section (or 2-D field, `two_d`) is gathered into its slot of the device
    // staging buffer, the slots come over in ONE copy into page-locked memory, and the stream is waited for once -
    // instead of a strided copy into pageable memory and a wait per column.
    struct Want { int field; bool two_d; std::vector<double>* out; };
    int get_many(const std::vector<Want>& list) const {
        if (list.empty()) return 0;
        const size_t n = size(), n2 = h->n2, per = std::max(n, n2), bytes = list.size() * per * sizeof(double);
        if (bytes > h->stage_bytes) {
            ATOMDEV_TRY(hipStreamSynchronize(h->stream));
            if (h->stage) ATOMDEV_TRY(hipFree(h->stage));
            h->stage = nullptr; h->stage_bytes = 0;
            ATOMDEV_TRY(hipMalloc(&h->stage, bytes));
            h->stage_bytes = bytes;
        }
        if (bytes > h->stage_host_bytes) {
            if (h->stage_host) ATOMDEV_TRY(hipHostFree(h->stage_host));
            h->stage_host = nullptr; h->stage_host_bytes = 0;
            ATOMDEV_TRY(hipHostMalloc(&h->stage_host, bytes, hipHostMallocDefault));
            h->stage_host_bytes = bytes;
        }
        double* dev = (double*)h->stage;
        const double* host = (const double*)h->stage_host;
        size_t off, width, pitch, rows;
        geometry(off, width, pitch, rows);
        for (size_t c = 0; c < list.size(); c++) {
            double* d = dev + c * per;
            if (list[c].two_d) {
                ATOMDEV_TRY(hipMemcpyAsync(d, h->g[list[c].field], n2 * 8, hipMemcpyDeviceToDevice, h->stream));
            } else {
                hipLaunchKernelGGL(k_section_gather, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, d,
                                   (const double*)(h->f[list[c].field] + off), width / 8, pitch / 8, n);
                ATOMDEV_TRY(hipGetLastError());
            }
        }
        ATOMDEV_TRY(hipMemcpyAsync(h->stage_host, dev, bytes, hipMemcpyDeviceToHost, h->stream));
        if (int rc = atomdev_sync_stream(h)) return rc;
        for (size_t c = 0; c < list.size(); c++)
            list[c].out->assign(host + c * per, host + c * per + (list[c].two_d ? n2 : n));
        return 0;
    }
    int put(int field, const std::vector<double>& in) const {
        size_t off, width, pitch, rows;
        geometry(off, width, pitch, rows);
        if (kind == ATOMDEV_ZONAL) {
            double* dev;
            if (int rc = staging(dev)) return rc;
            ATOMDEV_TRY(hipMemcpyAsync(dev, in.data(), rows * 8, hipMemcpyHostToDevice, h->stream));
            hipLaunchKernelGGL(k_section_column, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, h->stream, dev, h->f[field] + off,
                               (size_t)h->km, rows, 1);
            return atomdev_check(hipGetLastError(), "k_section_column");
        }
        ATOMDEV_TRY(hipMemcpy2DAsync(h->f[field] + off, pitch, in.data(), width, width, rows, hipMemcpyHostToDevice, h->stream));
        return 0;
    }
};

// The file as a list of pieces - literal text, a column of numbers (one a line) or a table of triples - whose text
// is made by a few host threads side by side and written in order.
struct Text {
    struct Piece {
        std::string text;                 // literal, or the rendered numbers
        std::vector<double> a, b, c;      // column: a; triples: a, b, c (an empty one stands for zeros)
        double multiplier = 1.;
        int kind = 0;                     // 0 literal, 1 column, 2 triples
    };
    std::vector<Piece> pieces;
    std::string& s_ref() {
        if (pieces.empty() || pieces.back().kind != 0) pieces.emplace_back();
        return pieces.back().text;
    }
    void lit(const std::string& t) { s_ref() += t; }
    void scalars(const char* name) { lit(std::string("SCALARS ") + name + " float 1\nLOOKUP_TABLE default\n"); }
    void column(const char* name, const std::vector<double>& a, double multiplier) {
        scalars(name);
        pieces.emplace_back();
        pieces.back().kind = 1; pieces.back().a = a; pieces.back().multiplier = multiplier;
    }
    void column(const char* name, std::vector<double>&& a, double multiplier) {
        scalars(name);
        pieces.emplace_back();
        pieces.back().kind = 1; pieces.back().a = std::move(a); pieces.back().multiplier = multiplier;
    }
    void triples(const std::vector<double>& a, const std::vector<double>& b, const std::vector<double>& c) {
        pieces.emplace_back();
        pieces.back().kind = 2; pieces.back().a = a; pieces.back().b = b; pieces.back().c = c;
    }
    static void render(Piece& p) {
        if (p.kind == 0) return;
        const size_t n = p.kind == 1 ? p.a.size() : std::max(p.a.size(), std::max(p.b.size(), p.c.size()));
        p.text.clear();
        p.text.reserve(n * (p.kind == 1 ? 12 : 36));
        // (the digits go into a local buffer that is appended when it is nearly full: one append per value costs more
        // than the value's digits; a value may take up to 400 bytes)
        std::vector<char> buf(1 << 16);
        char* const b0 = buf.data();
        char* const high = b0 + buf.size() - 1300;
        char* e = b0;
        for (size_t q = 0; q < n; q++) {
            if (e > high) { p.text.append(b0, (size_t)(e - b0)); e = b0; }
            if (p.kind == 1) e = put_fixed4(e, p.a[q] * p.multiplier);  // precision(4), ios::fixed
            else {
                e = put_fixed4(e, p.a.empty() ? 0. : p.a[q]); *e++ = ' ';
                e = put_fixed4(e, p.b.empty() ? 0. : p.b[q]); *e++ = ' ';
                e = put_fixed4(e, p.c.empty() ? 0. : p.c[q]);
            }
            *e++ = '\n';
        }
        p.text.append(b0, (size_t)(e - b0));
        p.a = std::vector<double>(); p.b = std::vector<double>(); p.c = std::vector<double>();
    }
    bool write(FILE* f) {
        host_parallel_run(pieces.size(), [this](size_t n) { render(pieces[n]); });
        bool ok = true;
        for (const Piece& p : pieces) ok = ok && std::fwrite(p.text.data(), 1, p.text.size(), f) == p.text.size();
        return ok;
    }
};

// lib/Utils.h:79-81
double exp_func(double T_K, double co_1, double co_2) { return std::exp(co_1 * (T_K - 273.15) / (T_K - co_2)); }

struct Column { const char* name; int field; double multiplier; bool two_d; };

int write_vtk(atomdev* h, int kind, int index, int Ma, const char* path) {
    ATOMDEV_TRY(hipSetDevice(h->device));
    const bool atm = h->model == ATOMDEV_ATM;
    const int im = h->im, jm = h->jm, km = h->km;
    const atomdev_params& P = h->p;
    const Section sec(h, kind, index);
    const size_t n = sec.size(), n2 = h->n2;
    auto slice = [&](int field, std::vector<double>& v) { return sec.get(field, v); };
    auto surface = [&](int field, std::vector<double>& v) {
        v.resize(n2);
        return atomdev_check(hipMemcpyAsync(v.data(), h->g[field], n2 * 8, hipMemcpyDeviceToHost, h->stream), "hipMemcpyAsync(2-D field)");
    };
    auto sync = [&]() { return atomdev_sync_stream(h); };

    Text out;
    // header and the point coordinates, built by repeated addition as in the reference
    static const char* title[2][3] = {
        { "Radial_Data_Atmosphere_Circulation", "Longitudinal_Data_Atmosphere_Circulation", "Zonal_Data_Atmosphere_Circulation" },
        { "Radial_Data_Hydrosphere_Circulation", "Longitudinal_Data_Hydrosphere_Circulation", "Zonal_Data_Hydrosphere_Circulation" } };
    out.lit("# vtk DataFile Version 3.0\n");
    out.lit(std::string(title[atm ? 0 : 1][kind]) + "\nASCII\nDATASET STRUCTURED_GRID\n");
    out.lit("DIMENSIONS " + std::to_string(sec.n_inner) + " " + std::to_string(sec.n_outer) + " 1\n");
    out.lit("POINTS " + std::to_string(sec.n_outer * sec.n_inner) + " float\n");
    {
        double x = 0., y = 0., z = 0.;
        const double dx = .1, step = kind == ATOMDEV_RADIAL ? .1 : (kind == ATOMDEV_LONGAL ? .025 : .05);
        std::vector<double> px, py, pz;
        px.reserve(n); py.reserve(n); pz.reserve(n);
        for (int a = 0; a < sec.n_outer; a++) {
            for (int b = 0; b < sec.n_inner; b++) {
                double& moving = kind == ATOMDEV_LONGAL ? z : y;
                if (b == 0) moving = 0.; else moving = moving + step;
                px.push_back(x); py.push_back(y); pz.push_back(z);
            }
            (kind == ATOMDEV_LONGAL ? z : y) = 0.;
            x = x + dx;
        }
        out.triples(px, py, pz);
    }
    const std::string point_data = "POINT_DATA " + std::to_string(sec.n_outer * sec.n_inner) + "\n";

    std::vector<double> u, v, w, t, c, cloud, ice;
    {
        std::vector<Section::Want> first = { { ATOMDEV_U, false, &u }, { ATOMDEV_V, false, &v }, { ATOMDEV_W, false, &w }, { ATOMDEV_T, false, &t } };
        if (atm) { first.push_back({ ATOMDEV_C, false, &c }); first.push_back({ ATOMDEV_CLOUD, false, &cloud }); first.push_back({ ATOMDEV_ICE, false, &ice }); }
        if (int rc = sec.get_many(first)) return rc;
    }
    auto columns = [&](std::initializer_list<Column> list) {
        std::vector<std::vector<double>> got(list.size());
        std::vector<Section::Want> want;
        size_t q = 0;
        for (const Column& col : list) want.push_back({ col.field, col.two_d, &got[q++] });
        if (int rc = sec.get_many(want)) return rc;
        q = 0;
        for (const Column& col : list) out.column(col.name, std::move(got[q++]), col.multiplier);
        return 0;
    };
    auto temperature_celsius = [&]() {
        std::vector<double> celsius(t.size());
        for (size_t q = 0; q < t.size(); q++) celsius[q] = t[q] * P.t_0 - P.t_0;
        out.column("Temperature", celsius, 1.);
    };

    if (atm) {
        const double p_scale = P.u_0 * P.u_0 * P.r_air * .01;
        if (kind == ATOMDEV_RADIAL) {
            if (Ma == 0) {  // :287-309: Celsius copy of the NASA temperature, its difference to t, evaporation - precipitation
                if (int rc = atomdev_refresh_flags(h)) return rc;
                hipLaunchKernelGGL(k_radial_nasa_terms, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, h->stream,
                                   (const double*)h->g[ATOMDEV_TEMPERATURE_NASA], (const double*)h->f[ATOMDEV_T],
                                   (const double*)h->g[ATOMDEV_EVAPORATION_DALTON], (const double*)h->g[ATOMDEV_PRECIPITATION],
                                   (const uint8_t*)h->flags, h->g[ATOMDEV_TEMP_NASA_CELSIUS], h->f[ATOMDEV_AUX_V] + (size_t)index * n2,
                                   h->f[ATOMDEV_AUX_W] + (size_t)index * n2, P.t_0, im, n2);
                ATOMDEV_TRY(hipGetLastError());
            }
            out.lit(point_data);
            out.column("u-Component", u, 1.); out.column("v-Component", v, 1.); out.column("w-Component", w, 1.);
            temperature_celsius();
            if (int rc = columns({ { "Temp_NASA", ATOMDEV_TEMP_NASA_CELSIUS, 1., true }, { "Temp_NASA_diff", ATOMDEV_AUX_V, 1., false },
                                   { "Topography", ATOMDEV_H, 1., false }, { "Topography_m", ATOMDEV_TOPOGRAPHY, 1., true } })) return rc;
            out.column("WaterVapour", c, 1000.); out.column("CloudWater", cloud, 1000.); out.column("CloudIce", ice, 1000.);
            if (int rc = columns({ { "PrecipitationRain", ATOMDEV_P_RAIN, 86400., false }, { "PrecipitationSnow", ATOMDEV_P_SNOW, 86400., false },
                                   { "Precipitation", ATOMDEV_PRECIPITATION, 1., true }, { "PrecipitableWater_2D", ATOMDEV_PRECIPITABLE_WATER, 1., true },
                                   { "Precipitation_NASA", ATOMDEV_PRECIPITATION_NASA, 1., true }, { "PressureDynamic", ATOMDEV_P_DYN, p_scale, false },
                                   { "PressureStatic", ATOMDEV_P_STAT, 1., false }, { "Epsilon_3D", ATOMDEV_EPSILON_3D, 1., false },
                                   { "albedo_2D", ATOMDEV_ALBEDO, 1., true }, { "epsilon_2D", ATOMDEV_EPSILON, 1., true },
                                   { "Q_radiation_2D", ATOMDEV_Q_RADIATION, 1., true }, { "Q_bottom_2D", ATOMDEV_Q_BOTTOM, 1., true },
                                   { "Q_latent_2D", ATOMDEV_Q_LATENT_2D, 1., true }, { "Q_sensible_2D", ATOMDEV_Q_SENSIBLE_2D, 1., true },
                                   { "BuoyancyForce", ATOMDEV_BUOYANCY, 1., false }, { "Q_Radiation", ATOMDEV_RADIATION_3D, 1., false },
                                   { "Q_Latent", ATOMDEV_Q_LATENT, 1., false }, { "Q_Sensible", ATOMDEV_Q_SENSIBLE, 1., false },
                                   { "Evaporation_Penman", ATOMDEV_EVAPORATION_PENMAN, 1., true }, { "Evaporation_Dalton", ATOMDEV_EVAPORATION_DALTON, 1., true },
                                   { "Heat_Evaporation", ATOMDEV_Q_EVAPORATION, 1., true }, { "Evap-Precip", ATOMDEV_AUX_W, 1., false },
                                   { "Vegetation", ATOMDEV_VEGETATION, 1., true }, { "CO2-Concentration", ATOMDEV_CO2, P.co2_0, false } })) return rc;
            out.lit("VECTORS v-w-Cell float \n");
            out.triples(v, w, {});
        } else if (kind == ATOMDEV_LONGAL) {
            out.lit(point_data);
            out.column("u-Component", u, 1.); out.column("v-Component", v, 1.); out.column("w-Component", w, 1.);
            temperature_celsius();
            std::vector<double> total(n);
            for (size_t q = 0; q < n; q++) total[q] = c[q] + cloud[q] + ice[q];  // :586-592, kept in aux_w
            if (int rc = sec.put(ATOMDEV_AUX_W, total)) return rc;
            if (int rc = columns({ { "Topography", ATOMDEV_H, 1., false } })) return rc;
            out.column("WaterVapour", c, 1000.); out.column("CloudWater", cloud, 1000.); out.column("CloudIce", ice, 1000.);
            out.column("Total_Water", total, 1000.);
            if (int rc = columns({ { "PrecipitationRain", ATOMDEV_P_RAIN, 86400., false }, { "PrecipitationSnow", ATOMDEV_P_SNOW, 86400., false },
                                   { "PressureDynamic", ATOMDEV_P_DYN, p_scale, false }, { "PressureStatic", ATOMDEV_P_STAT, 1., false },
                                   { "BuoyancyForce", ATOMDEV_BUOYANCY, 1., false }, { "Q_Latent", ATOMDEV_Q_LATENT, 1., false },
                                   { "Q_Sensible", ATOMDEV_Q_SENSIBLE, 1., false }, { "Epsilon_3D", ATOMDEV_EPSILON_3D, 1., false },
                                   { "CO2-Concentration", ATOMDEV_CO2, P.co2_0, false } })) return rc;
            out.lit("VECTORS u-w-Cell float\n");
            out.triples(u, {}, w);
        } else {
            out.lit(point_data);
            out.column("u-Component", u, 1.); out.column("v-Component", v, 1.); out.column("w-Component", w, 1.);
            temperature_celsius();
            // :452-482: total water and the two saturation ratios, kept in aux_w, aux_u, aux_v
            std::vector<double> total(n), sat_water(n), sat_ice(n), t0(jm);
            {   // the ground temperatures of the section: one double a row
                double* dev;
                if (int rc = sec.staging(dev)) return rc;
                hipLaunchKernelGGL(k_section_column, dim3((unsigned)((jm + 255) / 256)), dim3(256), 0, h->stream, dev,
                                   h->f[ATOMDEV_T] + index, (size_t)km, (size_t)jm, 0);
                ATOMDEV_TRY(hipGetLastError());
                ATOMDEV_TRY(hipMemcpyAsync(t0.data(), dev, (size_t)jm * 8, hipMemcpyDeviceToHost, h->stream));
            }
            if (int rc = sync()) return rc;
            for (int i = 0; i < im; i++)
                for (int j = 0; j < jm; j++) {
                    const size_t q = (size_t)i * jm + j;
                    total[q] = c[q] + cloud[q] + ice[q];
                    const double t_u = t[q] * P.t_0, T = t_u;
                    const double p_SL = .01 * (P.r_air * P.R_Air * t0[j] * P.t_0);
                    double p_h;
                    if (i != 0) p_h = std::exp(-P.g * (double)i * (P.L_atm / (double)(im - 1)) / (P.R_Air * t_u)) * p_SL;
                    else p_h = p_SL;
                    const double E_Rain = P.hp * exp_func(T, 17.2694, 35.86), E_Ice = P.hp * exp_func(T, 21.8746, 7.66);
                    const double q_Rain = P.ep * E_Rain / (p_h - E_Rain), q_Ice = P.ep * E_Ice / (p_h - E_Ice);
                    sat_water[q] = c[q] / q_Rain;
                    sat_ice[q] = T <= P.t_0 ? c[q] / q_Ice : 0.;
                }
            if (int rc = sec.put(ATOMDEV_AUX_W, total)) return rc;
            if (int rc = sec.put(ATOMDEV_AUX_U, sat_water)) return rc;
            if (int rc = sec.put(ATOMDEV_AUX_V, sat_ice)) return rc;
            if (int rc = columns({ { "Topography", ATOMDEV_H, 1., false } })) return rc;
            out.column("WaterVapour", c, 1000.); out.column("CloudWater", cloud, 1000.); out.column("CloudIce", ice, 1000.);
            out.column("Saturation_Water", sat_water, 1.); out.column("Saturation_Ice", sat_ice, 1.); out.column("Total_Water", total, 1000.);
            if (int rc = columns({ { "PrecipitationRain", ATOMDEV_P_RAIN, 86400., false }, { "PrecipitationSnow", ATOMDEV_P_SNOW, 86400., false },
                                   { "Source_WaterVapour", ATOMDEV_S_V, 1000., false }, { "Source_CloudWater", ATOMDEV_S_C, 1000., false },
                                   { "Source_CloudIce", ATOMDEV_S_I, 1000., false }, { "Source_Rain", ATOMDEV_S_R, 1000., false },
                                   { "Source_Snow", ATOMDEV_S_S, 1000., false }, { "Source_CloudWater_CondEvap", ATOMDEV_S_C_C, 1000., false },
                                   { "BuoyancyForce", ATOMDEV_BUOYANCY, 1., false }, { "Q_Radiation", ATOMDEV_RADIATION_3D, 1., false },
                                   { "Epsilon_3D", ATOMDEV_EPSILON_3D, 1., false }, { "Q_Latent", ATOMDEV_Q_LATENT, 1., false },
                                   { "Q_Sensible", ATOMDEV_Q_SENSIBLE, 1., false }, { "PressureDynamic", ATOMDEV_P_DYN, p_scale, false },
                                   { "PressureStatic", ATOMDEV_P_STAT, 1., false }, { "CO2-Concentration", ATOMDEV_CO2, P.co2_0, false } })) return rc;
            out.lit("VECTORS u-v-Cell float\n");
            out.triples(u, v, {});
        }
    } else {
        const double p_scale = P.u_0 * P.u_0 * P.r_0_water * 1e-3;
        out.lit(point_data);
        if (kind == ATOMDEV_RADIAL) {
            temperature_celsius();
            // :514-520: evaporation - precipitation over water, kept in aux_v
            std::vector<double> evap, precip, diff(n2);
            std::vector<uint8_t> ground(n2);
            if (int rc = atomdev_refresh_flags(h)) return rc;
            if (int rc = surface(ATOMDEV_EVAPORATION_DALTON, evap)) return rc;
            if (int rc = surface(ATOMDEV_PRECIPITATION, precip)) return rc;
            ATOMDEV_TRY(hipMemcpyAsync(ground.data(), h->flags, n2, hipMemcpyDeviceToHost, h->stream));
            if (int rc = sync()) return rc;
            for (size_t q = 0; q < n2; q++) diff[q] = (ground[q] & FL_LAND) ? 0. : evap[q] - precip[q];
            if (int rc = sec.put(ATOMDEV_AUX_V, diff)) return rc;
            if (int rc = columns({ { "Topography", ATOMDEV_H, 1., false }, { "Bathymetry_m", ATOMDEV_TOPOGRAPHY, 1., true } })) return rc;
        } else {
            if (int rc = columns({ { "Topography", ATOMDEV_H, 1., false } })) return rc;
        }
        out.column("u-Component", u, 1000.); out.column("v-Component", v, 1.); out.column("w-Component", w, 1.);
        if (kind != ATOMDEV_RADIAL) out.column("Temperature", t, 1.);
        if (int rc = columns({ { "PressureDynamic", ATOMDEV_P_DYN, p_scale, false }, { "PressureStatic", ATOMDEV_P_STAT, 1., false },
                               { "Salinity", ATOMDEV_C, 1., false }, { "DensityWater", ATOMDEV_R_WATER, 1., false },
                               { "DensitySaltWater", ATOMDEV_R_SALT_WATER, 1., false }, { "SaltFinger", ATOMDEV_SALT_FINGER, 1., false },
                               { "SaltDiffusion", ATOMDEV_SALT_DIFFUSION, 1., false }, { "SaltBalance", ATOMDEV_SALT_BALANCE, 1., false },
                               { "BuoyancyForce", ATOMDEV_BUOYANCY, 1., false } })) return rc;
        if (kind == ATOMDEV_RADIAL) {
            if (int rc = columns({ { "Upwelling", ATOMDEV_UPWELLING, 1., true }, { "Downwelling", ATOMDEV_DOWNWELLING, 1., true },
                                   { "BottomWater", ATOMDEV_BOTTOM_WATER, 1., true }, { "Evaporation_Dalton", ATOMDEV_EVAPORATION_DALTON, 1., true },
                                   { "Precipitation", ATOMDEV_PRECIPITATION, 1., true }, { "Evap-Precip", ATOMDEV_AUX_V, 1., false } })) return rc;
            out.lit("VECTORS v-w-Cell float\n");
            out.triples(v, w, {});
        } else if (kind == ATOMDEV_LONGAL) {
            out.lit("VECTORS u-w-Cell float\n");
            out.triples(u, {}, w);
        } else {
            out.lit("VECTORS u-v-Cell float\n");
            out.triples(u, v, {});
        }
    }
    if (int rc = sync()) return rc;
    // everything the file holds is on the host now: its text can be made while the caller goes on
    auto text = std::make_shared<Text>(std::move(out));
    const std::string file(path);
    return atomdev_defer_file(h, file, [text, file]() -> std::string {
        FILE* f = std::fopen(file.c_str(), "w");
        if (!f) return "could not open VTK file " + file;
        const bool ok = text->write(f);
        if (std::fclose(f) != 0 || !ok) return "write failed: " + file;
        return std::string();
    });
}

}  // namespace

// ---- paraview_panorama_vts (PostProcess_Atm.cpp:120-238, dump_array :33-48; PostProcess_Hyd.cpp:269-366) ----
// The whole grid as an ASCII .vts file, one field after the other in the writers' (k, j, i) order.  One field is on
// the host at a time; its text goes through a large stdio buffer.
namespace {

struct VtsFile {
    FILE* f = nullptr;
    std::vector<char> buffer;
    std::vector<double> field;
    atomdev* h;
    explicit VtsFile(atomdev* h_) : buffer(1 << 22), field(h_->n3), h(h_) {}
    ~VtsFile() { if (f) std::fclose(f); }
    int fetch(int id) {
        ATOMDEV_TRY(hipMemcpyAsync(field.data(), h->f[id], h->n3 * 8, hipMemcpyDeviceToHost, h->stream));
        if (int rc__ = atomdev_sync_stream(h)) return rc__;
        return 0;
    }
    // values of the fetched field in (k, j, i) order, `blank` after every row of levels and after every meridian
    template <class Value>
    void walk(const char* blank, Value value) {
        const int im = h->im, jm = h->jm, km = h->km;
        for (int k = 0; k < km; k++) {
            for (int j = 0; j < jm; j++) {
                char line[4096], *p = line;  // a row of levels at a time
                for (int i = 0; i < im; i++) {
                    if (p - line > (long)sizeof(line) - 420) { std::fwrite(line, 1, (size_t)(p - line), f); p = line; }
                    p = put_fixed4(p, value(field[idx3(jm, km, i, j, k)]));
                    *p++ = '\n';
                }
                std::fwrite(line, 1, (size_t)(p - line), f);
                std::fputs(blank, f);
            }
            std::fputs(blank, f);
        }
    }
    // dump_array: single newlines between the blocks
    int array(const char* name, int id, double multiplier) {
        if (int rc = fetch(id)) return rc;
        std::fprintf(f, "    <DataArray type=\"Float32\" Name=\"%s\" format=\"ascii\">\n", name);
        walk("\n", [multiplier](double x) { return x * multiplier; });
        std::fputs("\n    </DataArray>\n", f);
        return 0;
    }
};

int write_panorama(atomdev* h, const char* path) {
    ATOMDEV_TRY(hipSetDevice(h->device));
    const bool atm = h->model == ATOMDEV_ATM;
    const int im = h->im, jm = h->jm, km = h->km;
    const atomdev_params& P = h->p;
    VtsFile out(h);
    out.f = std::fopen(path, "w");
    ATOMDEV_REQUIRE(out.f, std::string("could not open panorama_vts file ") + path);
    std::setvbuf(out.f, out.buffer.data(), _IOFBF, out.buffer.size());
    FILE* f = out.f;
    // every line the writers end with "\n" << endl carries two newlines
    std::fputs("<?xml version=\"1.0\"?>\n\n<VTKFile type=\"StructuredGrid\" version=\"0.1\" byte_order=\"LittleEndian\">\n\n", f);
    std::fprintf(f, " <StructuredGrid WholeExtent=\"1 %d 1 %d 1 %d\">\n\n  <Piece Extent=\"1 %d 1 %d 1 %d\">\n\n", im, jm, km, im, jm, km);
    std::fputs(atm ? "   <PointData Vectors=\"Velocity\" Scalars=\"Topography u-component v-component w-component Temperature "
                     "CondensationTemp EvaporationTemp Epsilon_3D PressureDynamic PressureStatic WaterVapour CloudWater CloudIce "
                     "CO2-Concentration Q_Latent Rain RainSuper Ice PrecipitationRain PrecipitationSnow PrecipitationConv Updraft "
                     "Downdraft\">\n\n"
                   : "   <PointData Vectors=\"Velocity\" Scalars=\"Topography Temperature PressureDynamic PressureStatic Salinity\">\n\n", f);
    std::fputs("    <DataArray type=\"Float32\" NumberOfComponents=\"3\" Name=\"Velocity\" format=\"ascii\">\n\n", f);
    {
        std::vector<double> v(h->n3), w(h->n3);
        ATOMDEV_TRY(hipMemcpyAsync(v.data(), h->f[ATOMDEV_V], h->n3 * 8, hipMemcpyDeviceToHost, h->stream));
        ATOMDEV_TRY(hipMemcpyAsync(w.data(), h->f[ATOMDEV_W], h->n3 * 8, hipMemcpyDeviceToHost, h->stream));
        if (int rc = out.fetch(ATOMDEV_U)) return rc;
        for (int k = 0; k < km; k++) {
            for (int j = 0; j < jm; j++) {
                for (int i = 0; i < im; i++) {
                    const size_t q = idx3(jm, km, i, j, k);
                    { char b[1300], *e = put_fixed4(b, out.field[q]); *e++ = ' '; e = put_fixed4(e, v[q]); *e++ = ' ';
                      e = put_fixed4(e, w[q]); *e++ = '\n'; std::fwrite(b, 1, (size_t)(e - b), f); }
                }
                std::fputs("\n\n", f);
            }
            std::fputs("\n\n", f);
        }
        std::fputs("\n\n    </DataArray>\n\n", f);
    }
    struct Column { const char* name; int field; double multiplier; };
    auto arrays = [&](std::initializer_list<Column> list) {
        for (const Column& c : list) if (int rc = out.array(c.name, c.field, c.multiplier)) return rc;
        return 0;
    };
    if (atm) {
        if (int rc = arrays({ { "Topography", ATOMDEV_H, 1. }, { "u-component", ATOMDEV_U, 1. }, { "v-component", ATOMDEV_V, 1. },
                              { "w-component", ATOMDEV_W, 1. } })) return rc;
        if (int rc = out.fetch(ATOMDEV_T)) return rc;
        std::fputs("    <DataArray type=\"Float32\" Name=\"Temperature\" format=\"ascii\">\n\n", f);
        const double t_0 = P.t_0;
        out.walk("\n\n", [t_0](double x) { return x * t_0 - t_0; });
        std::fputs("\n\n    </DataArray>\n\n", f);
        if (int rc = arrays({ { "Epsilon_3D", ATOMDEV_EPSILON_3D, 1. }, { "WaterVapour", ATOMDEV_C, 1000. }, { "CloudWater", ATOMDEV_CLOUD, 1000. },
                              { "CloudIce", ATOMDEV_ICE, 1000. }, { "PrecipitationRain", ATOMDEV_P_RAIN, 86400. },
                              { "PrecipitationSnow", ATOMDEV_P_SNOW, 86400. }, { "PressureDynamic", ATOMDEV_P_DYN, P.u_0 * P.u_0 * P.r_air * .01 },
                              { "PressureStatic", ATOMDEV_P_STAT, 1. }, { "BuoyancyForce", ATOMDEV_BUOYANCY, 1. },
                              { "CO2-Concentration", ATOMDEV_CO2, 1. }, { "Q_Latent", ATOMDEV_Q_LATENT, 1. },
                              { "Q_Sensible", ATOMDEV_Q_SENSIBLE, 1. } })) return rc;
    } else {
        if (int rc = arrays({ { "Topography", ATOMDEV_H, 1. }, { "u-velocity", ATOMDEV_U, 1000. }, { "v-velocity", ATOMDEV_V, 1. },
                              { "w-velocity", ATOMDEV_W, 1. }, { "Temperature", ATOMDEV_T, 1. },
                              { "PressureDynamic", ATOMDEV_P_DYN, P.u_0 * P.u_0 * P.r_0_water * 1e-3 }, { "PressureStatic", ATOMDEV_P_STAT, 1. },
                              { "Salinity", ATOMDEV_C, 1. }, { "DensityWater", ATOMDEV_R_WATER, 1. },
                              { "DensitySaltWater", ATOMDEV_R_SALT_WATER, 1. }, { "Salt_Finger", ATOMDEV_SALT_FINGER, 1. },
                              { "SaltDiffusion", ATOMDEV_SALT_DIFFUSION, 1. }, { "SaltBalance", ATOMDEV_SALT_BALANCE, 1. },
                              { "BuoyancyForce", ATOMDEV_BUOYANCY, 1. } })) return rc;
    }
    std::fputs("   </PointData>\n\n   <Points>\n\n    <DataArray type=\"Float32\" NumberOfComponents=\"3\" format=\"ascii\">\n\n", f);
    {   // the "cartesian" coordinates: x grows along a column except on the first meridian and the first row
        double x = 0., y = 0., z = 0.;
        const double dx = atm ? .1 : .025, dy = .1, dz = .1;
        for (int k = 0; k < km; k++) {
            for (int j = 0; j < jm; j++) {
                for (int i = 0; i < im; i++) {
                    if (k == 0 || j == 0) x = 0.; else x = x + dx;
                    { char b[1300], *e = put_fixed4(b, x); *e++ = ' '; e = put_fixed4(e, y); *e++ = ' ';
                      e = put_fixed4(e, z); *e++ = '\n'; std::fwrite(b, 1, (size_t)(e - b), f); }
                }
                x = 0;
                y = y + dy;
                std::fputs("\n\n", f);
            }
            y = 0;
            z = z + dz;
            std::fputs("\n\n", f);
        }
    }
    std::fputs("    </DataArray>\n\n   </Points>\n\n  </Piece>\n\n </StructuredGrid>\n\n</VTKFile>\n\n", f);
    const bool ok = std::fflush(f) == 0 && !std::ferror(f);
    const bool closed = std::fclose(f) == 0;
    out.f = nullptr;
    ATOMDEV_REQUIRE(ok && closed, std::string("write failed: ") + path);
    return 0;
}

}  // namespace

extern "C" int atomdev_write_panorama_vts(atomdev_t* h, const char* path) {
    ATOMDEV_REQUIRE(h && path, "atomdev_write_panorama_vts: bad argument");
    return write_panorama(h, path);
}

extern "C" int atomdev_write_vtk(atomdev_t* h, int section, int index, int Ma, const char* path) {
    ATOMDEV_REQUIRE(h && path, "atomdev_write_vtk: bad argument");
    ATOMDEV_REQUIRE(section >= ATOMDEV_RADIAL && section <= ATOMDEV_ZONAL, "atomdev_write_vtk: section is 0 (radial), 1 (longal) or 2 (zonal)");
    const int limit = section == ATOMDEV_RADIAL ? h->im : (section == ATOMDEV_LONGAL ? h->jm : h->km);
    ATOMDEV_REQUIRE(index >= 0 && index < limit, "atomdev_write_vtk: index outside the grid");
    return write_vtk(h, section, index, Ma, path);
}
