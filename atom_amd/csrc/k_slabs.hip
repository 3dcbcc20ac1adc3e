// One model split into phi-slabs over several handles of one process (one handle per GPU; several on one GPU for
// rehearsal): the layout, the halo exchange and the whole-grid accuracy queries.  No counterpart in the reference, which
// has one domain; what the exchange serves there: the pressure step's neighbour columns (Pressure_Atm.cpp:140-155), the
// phi-seam fills (BC_Atm.cpp:115-150, Pressure_Atm.cpp:57-66, 170-176) and the continuity residuum (Accuracy_Atm.cpp:82-116).
//
// Layout (the same as atom_amd/slabs.py, which drives one slab per process through torch.distributed).  Slab r holds
//     [copies of global columns 0 .. S-1]        only the last slab
//     [G ghost columns]                          its left neighbour's last owned columns (not slab 0)
//     [owned columns]
//     [G ghost columns]                          its right neighbour's first owned columns (not the last slab)
//     [copies of global columns km-S .. km-1]    only slab 0
// as an ordinary smaller model, and runs the unmodified single-domain operators on it.  With G = 3 every k-stencil of an
// owned column (reach 2) and every k-window test evaluates as in the single domain; the S = 5 seam copies make the local
// seam fill of the two edge slabs equal to the global one.
//
// Exchange.  Every handle has a second stream.  atomdev_slabs_begin_exchange records an event on each compute stream
// ("the fields are final"), and on the owners' exchange streams, behind that event: one pack kernel per message (all
// fields of a neighbour in one dense block), one asynchronous peer copy into the holder's receive buffer, one event.
// atomdev_slabs_finish_exchange makes each holder's compute stream wait for its messages' events and unpacks there.
// The host never waits; whatever is enqueued on the compute streams between the two calls runs beside the transfer.
#include <cmath>

#include "atomdev_internal.h"

namespace {

constexpr int G = 3, S = 5, MAX_PENDING = 4, MAX_FIELDS = 16;

struct Segment { int g0, n, owner; };
struct Transfer { int owner, holder, k_src, k_dst, n; };

struct Plan {
    int km = 0, world = 0;
    std::vector<int> lo, hi;                     // owned global columns [lo, hi) of every slab
    std::vector<std::vector<Segment>> segments;  // per slab, in local order
    std::vector<Transfer> transfers;

    int local_km(int r) const { int s = 0; for (const Segment& g : segments[r]) s += g.n; return s; }
    int owned_local(int r) const {  // first local column of the owned segment
        int off = 0;
        for (const Segment& g : segments[r]) {
            if (g.owner == r && g.g0 == lo[r]) return off;
            off += g.n;
        }
        return -1;
    }
};

bool make_plan(int km, int world, Plan& P, std::string& why) {
    if (world < 1 || km < 3) { why = "a slab plan needs at least one slab and three columns"; return false; }
    P.km = km; P.world = world;
    P.lo.resize(world); P.hi.resize(world);
    for (int r = 0; r <= world; r++) {
        const int b = (int)std::nearbyint((double)r * km / world);   // ties to even, as Python's round()
        if (r < world) P.lo[r] = b;
        if (r > 0) P.hi[r - 1] = b;
    }
    for (int r = 0; r < world; r++)
        if (world > 1 && P.hi[r] - P.lo[r] < G + S) { why = "slabs narrower than " + std::to_string(G + S) + " columns"; return false; }
    P.segments.assign(world, {});
    for (int r = 0; r < world; r++) {
        std::vector<Segment>& seg = P.segments[r];
        if (world > 1 && r == world - 1) seg.push_back({ 0, S, 0 });
        if (r > 0) seg.push_back({ P.lo[r] - G, G, r - 1 });
        seg.push_back({ P.lo[r], P.hi[r] - P.lo[r], r });
        if (r < world - 1) seg.push_back({ P.hi[r], G, r + 1 });
        if (world > 1 && r == 0) seg.push_back({ km - S, S, world - 1 });
    }
    P.transfers.clear();
    for (int holder = 0; holder < world; holder++) {
        int off = 0;
        for (const Segment& g : P.segments[holder]) {
            if (g.owner != holder) P.transfers.push_back({ g.owner, holder, P.owned_local(g.owner) + (g.g0 - P.lo[g.owner]), off, g.n });
            off += g.n;
        }
    }
    return true;
}

struct FieldList { double* f[MAX_FIELDS]; };

// nk columns from k0 of nf fields <-> one dense block [nf][rows][nk]
template <bool PACK>
__global__ void k_slab_columns(FieldList a, double* __restrict__ buf, size_t rows, int km, int k0, int nk) {
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= rows * nk) return;
    const size_t row = q / nk;
    const int k = (int)(q - row * nk);
    double* x = a.f[blockIdx.y] + row * km + k0 + k;
    double* b = buf + (size_t)blockIdx.y * rows * nk + q;
    if (PACK) *b = *x; else *x = *b;
}

struct Message {          // one transfer of one pending exchange
    double *send = nullptr, *recv = nullptr;
    size_t capacity = 0;  // bytes
    hipEvent_t arrived = nullptr, consumed = nullptr;
    bool consumed_recorded = false;
};

struct Pending { int nf; int fields[MAX_FIELDS]; };

}  // namespace

struct atomdev_slabs {
    Plan plan;
    std::vector<atomdev*> h;
    std::vector<hipStream_t> xstream;
    std::vector<hipEvent_t> ready;
    std::vector<std::vector<Message>> msg;   // [pending slot][transfer]
    std::vector<Pending> pending;
};

extern "C" int atomdev_slabs_local_km(int km_global, int world, int rank) {
    Plan P; std::string why;
    if (!make_plan(km_global, world, P, why) || rank < 0 || rank >= world) return -1;
    return P.local_km(rank);
}

extern "C" int atomdev_slabs_columns(int km_global, int world, int rank, int* global_columns, int* owned_first_local, int* owned_count) {
    Plan P; std::string why;
    if (!make_plan(km_global, world, P, why)) { atomdev_set_error("atomdev_slabs_columns: " + why); return 1; }
    ATOMDEV_REQUIRE(rank >= 0 && rank < world, "atomdev_slabs_columns: no such slab");
    if (global_columns) {
        int n = 0;
        for (const Segment& g : P.segments[rank])
            for (int k = 0; k < g.n; k++) global_columns[n++] = g.g0 + k;
    }
    if (owned_first_local) *owned_first_local = P.owned_local(rank);
    if (owned_count) *owned_count = P.hi[rank] - P.lo[rank];
    return 0;
}

extern "C" int atomdev_slabs_destroy(atomdev_slabs_t* g) {
    if (!g) return 0;
    for (size_t r = 0; r < g->h.size(); r++) {
        (void)hipSetDevice(g->h[r]->device);
        if (g->xstream.size() > r && g->xstream[r]) { (void)hipStreamSynchronize(g->xstream[r]); (void)hipStreamDestroy(g->xstream[r]); }
        if (g->ready.size() > r && g->ready[r]) (void)hipEventDestroy(g->ready[r]);
    }
    for (auto& slot : g->msg)
        for (size_t t = 0; t < slot.size(); t++) {
            Message& m = slot[t];
            const Transfer& T = g->plan.transfers[t];
            if (m.send) { (void)hipSetDevice(g->h[T.owner]->device); (void)hipFree(m.send); }
            if (m.recv) { (void)hipSetDevice(g->h[T.holder]->device); (void)hipFree(m.recv); }
            if (m.arrived) (void)hipEventDestroy(m.arrived);
            if (m.consumed) (void)hipEventDestroy(m.consumed);
        }
    delete g;
    return 0;
}

extern "C" int atomdev_slabs_create(atomdev_t* const* handles, int n, int km_global, atomdev_slabs_t** out) {
    ATOMDEV_REQUIRE(handles && out && n >= 1, "atomdev_slabs_create: bad argument");
    *out = nullptr;
    Plan P; std::string why;
    if (!make_plan(km_global, n, P, why)) { atomdev_set_error("atomdev_slabs_create: " + why); return 1; }
    for (int r = 0; r < n; r++) {
        ATOMDEV_REQUIRE(handles[r], "atomdev_slabs_create: null handle");
        ATOMDEV_REQUIRE(handles[r]->model == handles[0]->model && handles[r]->im == handles[0]->im && handles[r]->jm == handles[0]->jm,
                        "atomdev_slabs_create: the slabs of one model share its kind and its im, jm");
        ATOMDEV_REQUIRE(handles[r]->km == P.local_km(r), "atomdev_slabs_create: slab " + std::to_string(r) + " must have km = " +
                        std::to_string(P.local_km(r)) + " (atomdev_slabs_local_km)");
    }
    atomdev_slabs* g = new atomdev_slabs();
    g->plan = P;
    g->h.assign(handles, handles + n);
    g->xstream.assign(n, nullptr);
    g->ready.assign(n, nullptr);
    g->msg.assign(MAX_PENDING, std::vector<Message>(P.transfers.size()));
    int rc = 0;
    for (int r = 0; r < n && !rc; r++) {
        rc = atomdev_check(hipSetDevice(handles[r]->device), "hipSetDevice");
        if (!rc) rc = atomdev_check(hipStreamCreateWithFlags(&g->xstream[r], hipStreamNonBlocking), "hipStreamCreate");
        if (!rc) rc = atomdev_check(hipEventCreateWithFlags(&g->ready[r], hipEventDisableTiming), "hipEventCreate");
        // copies between two devices go directly where the devices can reach each other's memory
        for (int q = 0; q < n && !rc; q++) {
            if (handles[q]->device == handles[r]->device) continue;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, handles[r]->device, handles[q]->device) == hipSuccess && can) {
                const hipError_t e = hipDeviceEnablePeerAccess(handles[q]->device, 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) rc = atomdev_check(e, "hipDeviceEnablePeerAccess");
                (void)hipGetLastError();
            }
        }
    }
    if (rc) { atomdev_slabs_destroy(g); return rc; }
    *out = g;
    return 0;
}

extern "C" int atomdev_slabs_begin_exchange(atomdev_slabs_t* g, const int* fields3d, int nf) {
    ATOMDEV_REQUIRE(g && fields3d && nf >= 1 && nf <= MAX_FIELDS, "atomdev_slabs_begin_exchange: bad argument");
    if (g->plan.world == 1) return 0;
    ATOMDEV_REQUIRE((int)g->pending.size() < MAX_PENDING, "atomdev_slabs_begin_exchange: too many exchanges in flight (finish one first)");
    const int slot = (int)g->pending.size();
    Pending pd; pd.nf = nf;
    for (int n = 0; n < nf; n++) {
        pd.fields[n] = fields3d[n];
        for (atomdev* h : g->h)
            ATOMDEV_REQUIRE(fields3d[n] >= 0 && fields3d[n] < ATOMDEV_N3D && h->f[fields3d[n]], "atomdev_slabs_begin_exchange: a field this model does not have");
    }
    // "the fields are final" on every compute stream; the exchange streams go on behind it
    for (int r = 0; r < g->plan.world; r++) {
        atomdev* h = g->h[r];
        ATOMDEV_TRY(hipSetDevice(h->device));
        ATOMDEV_TRY(hipEventRecord(g->ready[r], h->stream));
        ATOMDEV_TRY(hipStreamWaitEvent(g->xstream[r], g->ready[r], 0));
    }
    for (size_t t = 0; t < g->plan.transfers.size(); t++) {
        const Transfer& T = g->plan.transfers[t];
        atomdev *o = g->h[T.owner], *hd = g->h[T.holder];
        Message& m = g->msg[slot][t];
        const size_t rows = (size_t)o->im * o->jm, bytes = (size_t)nf * rows * T.n * sizeof(double);
        if (bytes > m.capacity) {   // (first use, or more fields than before: nothing can be in flight on this slot)
            if (m.send) { ATOMDEV_TRY(hipSetDevice(o->device)); ATOMDEV_TRY(hipStreamSynchronize(g->xstream[T.owner])); ATOMDEV_TRY(hipFree(m.send)); m.send = nullptr; }
            if (m.recv) { ATOMDEV_TRY(hipSetDevice(hd->device)); ATOMDEV_TRY(hipStreamSynchronize(hd->stream)); ATOMDEV_TRY(hipFree(m.recv)); m.recv = nullptr; }
            ATOMDEV_TRY(hipSetDevice(o->device));
            ATOMDEV_TRY(hipMalloc(&m.send, bytes));
            ATOMDEV_TRY(hipSetDevice(hd->device));
            ATOMDEV_TRY(hipMalloc(&m.recv, bytes));
            m.capacity = bytes;
            if (!m.arrived) ATOMDEV_TRY(hipEventCreateWithFlags(&m.arrived, hipEventDisableTiming));
            if (!m.consumed) ATOMDEV_TRY(hipEventCreateWithFlags(&m.consumed, hipEventDisableTiming));
        }
        ATOMDEV_TRY(hipSetDevice(o->device));
        hipStream_t xs = g->xstream[T.owner];
        // the receive buffer is free once the holder has unpacked what the slot carried before
        if (m.consumed_recorded) ATOMDEV_TRY(hipStreamWaitEvent(xs, m.consumed, 0));
        FieldList fl;
        for (int n = 0; n < nf; n++) fl.f[n] = o->f[fields3d[n]];
        const dim3 grid((unsigned)((rows * T.n + 255) / 256), nf);
        hipLaunchKernelGGL(k_slab_columns<true>, grid, dim3(256), 0, xs, fl, m.send, rows, o->km, T.k_src, T.n);
        ATOMDEV_TRY(hipGetLastError());
        if (o->device == hd->device) ATOMDEV_TRY(hipMemcpyAsync(m.recv, m.send, bytes, hipMemcpyDeviceToDevice, xs));
        else ATOMDEV_TRY(hipMemcpyPeerAsync(m.recv, hd->device, m.send, o->device, bytes, xs));
        ATOMDEV_TRY(hipEventRecord(m.arrived, xs));
    }
    g->pending.push_back(pd);
    return 0;
}

extern "C" int atomdev_slabs_finish_exchange(atomdev_slabs_t* g) {
    ATOMDEV_REQUIRE(g, "atomdev_slabs_finish_exchange: null group");
    for (size_t slot = 0; slot < g->pending.size(); slot++) {
        const Pending& pd = g->pending[slot];
        for (size_t t = 0; t < g->plan.transfers.size(); t++) {
            const Transfer& T = g->plan.transfers[t];
            atomdev* hd = g->h[T.holder];
            Message& m = g->msg[slot][t];
            ATOMDEV_TRY(hipSetDevice(hd->device));
            ATOMDEV_TRY(hipStreamWaitEvent(hd->stream, m.arrived, 0));
            const size_t rows = (size_t)hd->im * hd->jm;
            FieldList fl;
            for (int n = 0; n < pd.nf; n++) {
                fl.f[n] = hd->f[pd.fields[n]];
                if (pd.fields[n] == ATOMDEV_H) hd->flags_dirty = true;
            }
            const dim3 grid((unsigned)((rows * T.n + 255) / 256), pd.nf);
            hipLaunchKernelGGL(k_slab_columns<false>, grid, dim3(256), 0, hd->stream, fl, m.recv, rows, hd->km, T.k_dst, T.n);
            ATOMDEV_TRY(hipGetLastError());
            ATOMDEV_TRY(hipEventRecord(m.consumed, hd->stream));
            m.consumed_recorded = true;
        }
    }
    g->pending.clear();
    return 0;
}

// residuumQuery_3D of the whole grid (Accuracy_Atm.cpp:82-116, Accuracy_Hyd.cpp:88-116).  The reference's scan is order
// dependent (a signed store against an unsigned compare: the running value can fall), but it is a sequence of jumps -
// from a running value r the next point that replaces it is the first with |x| >= |r| - and each slab answers "from this
// position of the whole grid's sequence on, where is my largest |res|, my first non-negative res, my last valid one"
// (atomdev_residuum3d_suffix).  A handful of jumps per query; value and location are those of the single domain.
extern "C" int atomdev_slabs_residuum3d(atomdev_slabs_t* g, double* min, int loc[3]) {
    ATOMDEV_REQUIRE(g && min && loc, "atomdev_slabs_residuum3d: bad argument");
    const int world = g->plan.world;
    const int w_field = ATOMDEV_W;
    if (world > 1) {   // w one column beyond each end of the owned columns
        if (int rc = atomdev_slabs_finish_exchange(g)) return rc;
        if (int rc = atomdev_slabs_begin_exchange(g, &w_field, 1)) return rc;
        if (int rc = atomdev_slabs_finish_exchange(g)) return rc;
    }
    for (atomdev* h : g->h)
        if (int rc = atomdev_residuum3d_field(h)) return rc;
    atomdev* h0 = g->h[0];
    const bool ge = h0->model == ATOMDEV_HYD;   // the hydrosphere compares with >= and starts from 0
    double m = ge ? 0. : *min, pos = -1., s = 0.;
    const double total = (double)(h0->im - 2) * (h0->jm - 2) * (g->plan.km - 2);
    std::vector<double> parts(7 * (size_t)world);
    while (s < total) {
        for (int r = 0; r < world; r++)
            if (int rc = atomdev_residuum3d_suffix(g->h[r], g->plan.owned_local(r), g->plan.hi[r] - g->plan.lo[r], g->plan.lo[r], g->plan.km, s, &parts[7 * r]))
                return rc;
        auto first = [&](int ip, int iv, double want_amax, double& q, double& xq) {  // smallest position >= 0 among the slabs' (position, value)
            q = -1.; xq = 0.;
            for (int r = 0; r < world; r++) {
                const double* p = &parts[7 * r];
                if (want_amax >= 0. && p[0] != want_amax) continue;
                if (p[ip] >= 0. && (q < 0. || p[ip] < q)) { q = p[ip]; xq = p[iv]; }
            }
        };
        if (m < 0.) {   // every point replaces a negative state up to the first non-negative one
            double q, xq;
            first(3, 4, -1., q, xq);
            if (q >= 0.) { m = xq; pos = q; s = q + 1.; continue; }
            double lp = -1., lv = 0.;
            for (int r = 0; r < world; r++) if (parts[7 * r + 5] > lp) { lp = parts[7 * r + 5]; lv = parts[7 * r + 6]; }
            if (lp >= 0.) { m = lv; pos = lp; }
            break;
        }
        double amax = -1.;
        for (int r = 0; r < world; r++) amax = parts[7 * r] > amax ? parts[7 * r] : amax;
        if (amax < 0. || !(ge ? amax >= m : amax > m)) break;
        double P, xP;
        first(1, 2, amax, P, xP);
        m = xP; pos = P; s = P + 1.;
    }
    *min = m;
    if (pos >= 0.) {
        const long long p = (long long)pos, nk = g->plan.km - 2, nj = h0->jm - 2;
        loc[0] = (int)(p / (nj * nk)) + 1; loc[1] = (int)((p / nk) % nj) + 1; loc[2] = (int)(p % nk) + 1;
    }
    return 0;
}

// steadyQuery_3D of the whole grid: the slabs' maxima of |x - xn| combined, the position in the whole grid as tie-break
// (the first one for the atmosphere, the last one for the hydrosphere, as the two reference classes compare).
extern "C" int atomdev_slabs_steady3d(atomdev_slabs_t* g, double* values, int* loc) {
    ATOMDEV_REQUIRE(g && values && loc, "atomdev_slabs_steady3d: bad argument");
    atomdev* h0 = g->h[0];
    const bool last = h0->model == ATOMDEV_HYD;
    const int nf = last ? 6 : 9, world = g->plan.world;
    std::vector<double> v(nf * (size_t)world), p(nf * (size_t)world);
    for (int r = 0; r < world; r++)
        if (int rc = atomdev_steady3d_part(g->h[r], g->plan.owned_local(r), g->plan.hi[r] - g->plan.lo[r], g->plan.lo[r], g->plan.km, &v[nf * r], &p[nf * r]))
            return rc;
    const long long kmg = g->plan.km, n2 = (long long)h0->jm * kmg;
    for (int f = 0; f < nf; f++) {
        double bv = -1., bp = 0.; bool have = false;
        for (int r = 0; r < world; r++) {
            const double pv = v[nf * r + f], pp = p[nf * r + f];
            if (pv > bv || (pv == bv && have && (last ? pp > bp : pp < bp))) { bv = pv; bp = pp; have = true; }
        }
        if (bv < 0. || (!last && bv == 0.)) { bv = 0.; bp = 0.; }
        const long long q = (long long)bp;
        values[f] = bv;
        loc[3 * f] = (int)(q / n2); loc[3 * f + 1] = (int)((q % n2) / kmg); loc[3 * f + 2] = (int)(q % kmg);
    }
    return 0;
}
