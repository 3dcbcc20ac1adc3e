"""Phi-slab decomposition of one model over several GPUs (one process per GPU).

What couples neighbouring longitudes in the 3-D iteration (cAtmosphereModel.cpp:756-874, cHydrosphereModel.cpp:420-610):
  * the land-mask tests of the right-hand sides and of the pressure step (k+-1, k+-2): static, part of every slab's mask;
  * the pressure Jacobi sweep: p_dynn at k+-1, aux_w at k+-1 and k+-2 (Pressure_Atm.cpp:140-155, Pressure_Hyd.cpp);
  * the phi-seam fills, which set columns 0 and km-1 to the mean of the Neumann extrapolations from columns 1, 2 and
    km-2, km-3 (BC_Atm.cpp:115-150, Pressure_Atm.cpp:57-66, 170-176);
  * the continuity residuum (w at k+-1, Accuracy_Atm.cpp:82-116), an order-dependent scan over the whole grid.
The 3-D Runge-Kutta sweeps do not: their phi derivatives are multiplied by h_d_k = 0 (RHS_Atm.cpp:146-149,
RHS_Hyd.cpp:136-139), so a column's new values depend on that column and on the mask around it only.

Layout.  Every rank holds an ordinary, smaller model (its own handle, its own km) made of
    [copies of global columns 0..S-1]       only on the last rank
    [G ghost columns]                       left neighbour's last owned columns (not on rank 0)
    [owned columns]
    [G ghost columns]                       right neighbour's first owned columns (not on the last rank)
    [copies of global columns km-S..km-1]   only on rank 0
and runs the unmodified single-domain operators on it.  With G = 3 every k-stencil of an owned column (reach 2) and
every k-window test (k >= 2, k <= km-3) evaluates exactly as in the single domain.  The seam copies make the LOCAL seam
fill of the two edge ranks equal to the global one: the copies of columns km-3 .. km-1 (0 .. 2 on the last rank) stay
valid through every column-local operator because their mask neighbourhood lies inside the S = 5 copies.  The other
non-owned columns hold whatever the operators make of them; they never feed an owned column except through the
fields named below, which their owners send first.

Exchange.  Before a pressure step: p_dynn, aux_u, aux_v, aux_w at the ghost and seam columns (the aux fields are
final after the last RK sweep, p_dynn after the old -> new copy, so the messages are posted right there and travel
while the limiter, the queries and the copy run; the pressure step waits for them).  Before a residuum query: w.
Nothing after an RK sweep, nothing for the pointwise operators and the r / theta / phi fills.

Queries.  Every rank evaluates its owned columns (atomdev_residuum3d_suffix, atomdev_steady3d_part); the reference's
scan over the whole grid's lexicographic sequence is run as its jumps, each jump one all-gather of a seven-number
summary per rank; the steady maxima combine with the position in the whole grid as tie-break.  Results and locations
are those of the single domain.

Start of a slice.  The file readers, the initial conditions and the 2-D surface loop (a phi-coupled Gauss-Seidel
sweep over 65 k points) run on one full-width model on rank 0; `distribute` hands every rank its columns of every
field, `collect` brings the owned columns back.  The column physics of the atmosphere (class members carried from
column to column in the whole grid's visiting order) is not sharded.
"""
import numpy as np

G = 3  # ghost columns per interior slab edge
S = 5  # seam copies held by the two edge ranks

ATM_STATE = ["T", "U", "V", "W", "P_DYN", "C", "CLOUD", "ICE", "CO2"]
HYD_STATE = ["T", "U", "V", "W", "P_DYN", "C"]

# fields whose ghost / seam copies must be the owners' before an operator runs
NEEDS = {
    "atm_pressure3d": ["P_DYNN", "AUX_U", "AUX_V", "AUX_W"],
    "hyd_pressure3d": ["P_DYNN", "AUX_U", "AUX_V", "AUX_W"],
    "residuum": ["W"],
}
NOT_SHARDED = {"atm_rk2d", "hyd_rk2d", "atm_pressure2d", "hyd_pressure2d"}


class SlabPlan:
    """Which global columns every rank holds, in local order, and who owns them."""

    def __init__(self, km, world):
        if world < 1:
            raise ValueError("world must be >= 1")
        self.km, self.world = km, world
        bounds = [round(r * km / world) for r in range(world + 1)]
        self.owned = [(bounds[r], bounds[r + 1]) for r in range(world)]
        if world > 1 and min(e - s for s, e in self.owned) < G + S:
            raise ValueError("slabs of %d columns over %d ranks are narrower than %d columns" % (km, world, G + S))
        self.segments = []  # per rank: list of (global_start, n, owner)
        for r, (s, e) in enumerate(self.owned):
            seg = []
            if world > 1 and r == world - 1:
                seg.append((0, S, 0))
            if r > 0:
                seg.append((s - G, G, r - 1))
            seg.append((s, e - s, r))
            if r < world - 1:
                seg.append((e, G, r + 1))
            if world > 1 and r == 0:
                seg.append((km - S, S, world - 1))
            self.segments.append(seg)

    def columns(self, rank):
        """Global column index of every local column."""
        return np.concatenate([np.arange(g0, g0 + n) for g0, n, _ in self.segments[rank]])

    def local_km(self, rank):
        return sum(n for _, n, _ in self.segments[rank])

    def owned_local(self, rank):
        """(first local column, count) of the owned segment."""
        off = 0
        for g0, n, owner in self.segments[rank]:
            if owner == rank and g0 == self.owned[rank][0]:
                return off, n
            off += n
        raise AssertionError

    def local_of(self, rank, gcol):
        """Local index on `rank` of one of its OWNED global columns."""
        off, n = self.owned_local(rank)
        s = self.owned[rank][0]
        assert s <= gcol < s + n
        return off + (gcol - s)

    def transfers(self):
        """All (owner, holder, owner_local_k0, holder_local_k0, n) copies of one refresh, in a fixed global order."""
        out = []
        for holder in range(self.world):
            off = 0
            for g0, n, owner in self.segments[holder]:
                if owner != holder:
                    out.append((owner, holder, self.local_of(owner, g0), off, n))
                    assert self.owned[owner][0] <= g0 and g0 + n <= self.owned[owner][1]
                off += n
        return out


class SlabModel:
    """One rank's slab: `engine` is that rank's model (atom_amd.DeviceModel on its GPU) built with
    km = plan.local_km(rank); `dist` is torch.distributed (initialised) or None for a single rank.
    Exchange buffers live on `xdev` ("cuda:N" for RCCL, "cpu" for gloo) and are allocated once per message."""

    def __init__(self, engine, plan, rank, dist=None, xdev="cpu", model=0):
        self.e, self.plan, self.rank, self.dist, self.xdev, self.model = engine, plan, rank, dist, xdev, model
        if plan.world > 1 and dist is None:
            raise ValueError("a plan over several ranks needs torch.distributed")
        self.cols = plan.columns(rank)
        self._transfers = [t for t in plan.transfers() if rank in t[:2]]
        self._buffers = {}   # (transfer index, number of fields) -> persistent tensor
        self._pending = []   # posted exchanges: (work handles, receives to unpack)

    # ---- data in and out ----------------------------------------------------
    def scatter(self, fields3d, fields2d):
        """Local slices (including ghosts and seam copies) of global arrays -> dicts for upload."""
        f = {k: np.ascontiguousarray(v[..., self.cols]) for k, v in fields3d.items()}
        g = {k: np.ascontiguousarray(v[..., self.cols]) for k, v in fields2d.items()}
        return f, g

    def owned_block(self, name):
        """(global first column, (im, jm, n) block) of this rank's owned columns of a 3-D field."""
        off, n = self.plan.owned_local(self.rank)
        return self.plan.owned[self.rank][0], self.e.get_columns(name, off, n)

    def distribute(self, full, names3d, names2d=()):
        """Rank 0 holds `full`, a model of the whole grid (anything with download(name)); every rank receives its local
        columns - ghosts and seam copies included - of the named fields and uploads them.  `full` is None elsewhere."""
        for name in list(names3d) + list(names2d):
            if self.plan.world == 1:
                a = full.download(name)
            else:
                parts = None
                if self.rank == 0:
                    a = full.download(name)
                    parts = [np.ascontiguousarray(a[..., self.plan.columns(r)]) for r in range(self.plan.world)]
                box = [None]
                self.dist.scatter_object_list(box, parts, src=0)
                a = box[0]
            self.e.upload(name, a)

    def collect(self, names3d):
        """The owned columns of the named 3-D fields assembled on rank 0: {name: (im, jm, km) array} there, None elsewhere."""
        mine = {k: self.owned_block(k) for k in names3d}
        if self.plan.world == 1:
            return {k: np.ascontiguousarray(b) for k, (_, b) in mine.items()}
        mine = {k: (g0, np.ascontiguousarray(b)) for k, (g0, b) in mine.items()}
        gathered = [None] * self.plan.world if self.rank == 0 else None
        self.dist.gather_object(mine, gathered, dst=0)
        if self.rank != 0:
            return None
        out = {}
        for k in names3d:
            a = np.empty((self.e.im, self.e.jm, self.plan.km))
            for part in gathered:
                g0, blk = part[k]
                a[:, :, g0:g0 + blk.shape[-1]] = blk
            out[k] = a
        return out

    # ---- exchange -------------------------------------------------------------
    def _buffer(self, index, nfields, n):
        import torch
        key = (index, nfields)
        if key not in self._buffers:
            self._buffers[key] = torch.empty((nfields, self.e.im, self.e.jm, n), dtype=torch.float64, device=self.xdev)
        return self._buffers[key]

    def _view(self, t):
        # engines take numpy arrays for host buffers and tensors (data_ptr) for device buffers
        return t.numpy() if t.device.type == "cpu" else t

    def begin_exchange(self, names):
        """Owners post the ghost / seam copies of the named 3-D fields, holders post the receives; returns at once."""
        names = list(names)
        if self.plan.world == 1 or not names:
            return
        on_device = self.xdev != "cpu"
        works, unpack = [], []
        packed = False
        for index, (owner, holder, k_src, k_dst, n) in enumerate(self._transfers):
            buf = self._buffer((index, tuple(names)), len(names), n)
            if self.rank == owner:
                if hasattr(self.e, "pack_columns"):
                    self.e.pack_columns(names, k_src, n, self._view(buf))      # one kernel, one message
                else:
                    for m, name in enumerate(names):
                        self.e.get_columns(name, k_src, n, out=self._view(buf[m]))
                packed = True
        if packed and on_device:
            self.e.sync()   # the packing copies ran on the handle's stream; the collective library reads on its own
        ops = []
        for index, (owner, holder, k_src, k_dst, n) in enumerate(self._transfers):
            buf = self._buffer((index, tuple(names)), len(names), n)
            if self.rank == owner:
                ops.append(self.dist.P2POp(self.dist.isend, buf, holder))
            else:
                ops.append(self.dist.P2POp(self.dist.irecv, buf, owner))
                unpack.append((buf, names, k_dst))
        works = self.dist.batch_isend_irecv(ops)   # one group: no ordering between the messages of different neighbours
        self._pending.append((works, unpack, on_device))

    def finish_exchange(self, only_last=False):
        """Waits for every posted exchange (or the one posted last) and stores what arrived into the ghost / seam columns."""
        todo = [self._pending.pop()] if only_last and self._pending else self._pending
        if not only_last:
            self._pending = []
        for works, unpack, on_device in todo:
            for w in works:
                w.wait()
            if on_device:
                import torch
                torch.cuda.current_stream(self.xdev).synchronize()   # wait() orders the collective against torch's stream on the buffers' device only
            for buf, names, k_dst in unpack:
                if hasattr(self.e, "unpack_columns"):
                    self.e.unpack_columns(names, k_dst, self._view(buf))
                else:
                    for m, name in enumerate(names):
                        self.e.set_columns(name, k_dst, self._view(buf[m]))

    def refresh(self, names):
        """begin_exchange and the wait for it; exchanges posted earlier stay in flight."""
        if self.plan.world == 1 or not list(names):
            return
        self.begin_exchange(names)
        self.finish_exchange(only_last=True)

    # ---- operators ----------------------------------------------------------
    def op(self, name):
        if name in NOT_SHARDED and self.plan.world > 1:
            raise ValueError("%s is not sharded: the surface loop runs on the full-width model of rank 0" % name)
        if name in NEEDS:
            self.finish_exchange()
            self.refresh(NEEDS[name])
        self.e.op(name)

    def move_data_to_new_arrays(self, level=-1):
        self.e.move_data_to_new_arrays(level)

    # ---- queries --------------------------------------------------------------
    def _all(self, values):
        """Every rank's list of numbers (the same count on every rank), in rank order: one all-gather of a small tensor."""
        if self.plan.world == 1:
            return [list(values)]
        import torch
        mine = torch.tensor([float(v) for v in values], dtype=torch.float64, device=self.xdev)
        box = [torch.empty_like(mine) for _ in range(self.plan.world)]
        self.dist.all_gather(box, mine)
        return [b.tolist() for b in box]

    def residuum(self, min0=0.0, loc0=(0, 0, 0)):
        """residuumQuery_3D of the whole grid (Accuracy_Atm.cpp:82-116, Accuracy_Hyd.cpp:88-116): (value, location)."""
        self.refresh(NEEDS["residuum"])
        self.e.residuum_field()
        off, n = self.plan.owned_local(self.rank)
        kg0, kmg = self.plan.owned[self.rank][0], self.plan.km
        ge = self.model == 1               # the hydrosphere compares with >= and starts from 0
        m = 0.0 if ge else float(min0)
        pos = -1
        s = 0.0
        total = float((self.e.im - 2) * (self.e.jm - 2) * (kmg - 2))

        def first(cands):                  # smallest position among (position, value) pairs with position >= 0
            cands = [c for c in cands if c[0] >= 0]
            return min(cands) if cands else (-1.0, 0.0)

        while s < total:
            parts = self._all(self.e.residuum_suffix(off, n, kg0, kmg, s))
            if m < 0.0:                    # every point replaces a negative state up to the first non-negative one
                q, xq = first([(p[3], p[4]) for p in parts])
                if q >= 0:
                    m, pos, s = xq, q, q + 1
                    continue
                last = max([(p[5], p[6]) for p in parts])
                if last[0] >= 0:
                    m, pos = last[1], last[0]
                break
            amax = max(p[0] for p in parts)
            if amax < 0 or not (amax >= m if ge else amax > m):
                break
            P, xP = first([(p[1], p[2]) for p in parts if p[0] == amax])
            m, pos, s = xP, P, P + 1
        if pos < 0:
            return m, tuple(loc0)
        pos = int(pos)
        nk, nj = kmg - 2, self.e.jm - 2
        return m, (pos // (nj * nk) + 1, (pos // nk) % nj + 1, pos % nk + 1)

    def steady(self):
        """steadyQuery_3D of the whole grid: (values, locations), fields in the order of atomdev_*_steady3d."""
        off, n = self.plan.owned_local(self.rank)
        kg0, kmg = self.plan.owned[self.rank][0], self.plan.km
        v, p = self.e.steady_part(off, n, kg0, kmg)
        parts = self._all(list(v) + list(p))
        nf = len(v)
        last = self.model == 1
        values, locs = [], []
        n2 = self.e.jm * kmg
        for f in range(nf):
            best_v, best_p = -1.0, None
            for part in parts:
                pv, pp = part[f], part[nf + f]
                if pv > best_v or (pv == best_v and best_p is not None and (pp > best_p if last else pp < best_p)):
                    best_v, best_p = pv, pp
            if best_v < 0.0 or (not last and best_v == 0.0):
                best_v, best_p = 0.0, 0
            q = int(best_p)
            values.append(best_v)
            locs.append((q // n2, (q % n2) // kmg, q % kmg))
        return values, locs

    # ---- the 3-D loop of the flow solver --------------------------------------
    def run_3d_loop(self, pressure_iters=2, velocity_iters=2, queries=True):
        """The pressure / velocity iterations of run_3D_loop without the column physics, exchanges posted as early as their
        data is final.  Returns the queries' results in call order: (residuum, location) before and after every sweep and
        (values, locations) of the steady query."""
        atm = self.model == 0
        pre = "atm_" if atm else "hyd_"
        log = []
        if atm:
            self.move_data_to_new_arrays(-1)    # cAtmosphereModel.cpp:753; the hydrosphere's loop has no such copy
        for _ in range(pressure_iters):
            for v in range(1, velocity_iters + 1):
                if queries:
                    first = self.residuum()
                    log.append(first)
                if atm:
                    ops = ["bc_radius", "bc_theta", "bc_phi", "value_limitation", "solid_ground", "rk3d", "value_limitation"]
                else:
                    ops = ["bc_radius", "bc_theta", "bc_phi", "pressure_density", "value_limitation", "rk3d", "solid_ground"]
                for op in ops:
                    self.e.op(pre + op)
                    if op == "rk3d" and v == velocity_iters:
                        self.begin_exchange(["AUX_U", "AUX_V", "AUX_W"])   # final until the next sweep
                if queries:
                    log.append(self.residuum(*first) if atm else self.residuum())
                    log.append(self.steady())
                if atm:
                    self.e.msl_data(1)      # Results_MSL_Atm::run_MSL_data, column by column (:854-859)
                self.move_data_to_new_arrays(-1)
            self.begin_exchange(["P_DYNN"])
            self.finish_exchange()
            self.e.op(pre + "pressure3d")
        return log


class LocalSlabs:
    """One model split into phi-slabs over several handles of THIS process - one per GPU of the node, or several on one
    GPU - with the exchange and the whole-grid queries done by the library (atomdev_slabs_*: a second stream per handle,
    events, peer copies between persistent device buffers; the host never waits).  A thin caller: layout, exchange and
    queries are the C side's."""

    def __init__(self, model, im, jm, km_global, params, world, devices=None):
        import ctypes as C
        from .device import DeviceModel, F3, load_library
        self.lib = load_library()
        self.model, self.im, self.jm, self.km, self.world = model, im, jm, km_global, world
        self.F3 = F3
        devices = list(devices) if devices is not None else [-1] * world
        self.e, self.cols, self.owned = [], [], []
        for r in range(world):
            kml = self.lib.atomdev_slabs_local_km(km_global, world, r)
            if kml < 0:
                raise ValueError("slabs of %d columns over %d handles are too narrow" % (km_global, world))
            cols = (C.c_int * kml)()
            first, count = C.c_int(), C.c_int()
            if self.lib.atomdev_slabs_columns(km_global, world, r, cols, C.byref(first), C.byref(count)):
                raise RuntimeError(self.lib.atomdev_last_error().decode())
            self.cols.append(np.array(cols[:], dtype=np.int64))
            self.owned.append((first.value, count.value))
            self.e.append(DeviceModel(model, im, jm, kml, params, device=devices[r]))
        handles = (C.c_void_p * world)(*[e._h for e in self.e])
        self._g = C.c_void_p()
        self._check(self.lib.atomdev_slabs_create(handles, world, km_global, C.byref(self._g)))

    def _check(self, rc):
        if rc:
            raise RuntimeError(self.lib.atomdev_last_error().decode())

    def close(self):
        if self._g:
            self.lib.atomdev_slabs_destroy(self._g)
            self._g = None
        for e in self.e:
            e.close()
        self.e = []

    def upload_all(self, fields3d, fields2d):
        """Every slab's columns (ghosts and seam copies included) of whole-grid arrays."""
        for r, e in enumerate(self.e):
            e.upload_all({k: np.ascontiguousarray(v[..., self.cols[r]]) for k, v in fields3d.items()},
                         {k: np.ascontiguousarray(v[..., self.cols[r]]) for k, v in fields2d.items()})

    def collect(self, name):
        """A 3-D field of the whole grid from the slabs' owned columns."""
        a = np.empty((self.im, self.jm, self.km))
        for r, e in enumerate(self.e):
            first, count = self.owned[r]
            g0 = int(self.cols[r][first])
            a[:, :, g0:g0 + count] = e.get_columns(name, first, count)
        return a

    def op(self, name):
        for e in self.e:
            e.op(name)

    def move_data_to_new_arrays(self, level=-1):
        for e in self.e:
            e.move_data_to_new_arrays(level)

    def sync(self):
        for e in self.e:
            e.sync()

    def begin_exchange(self, names):
        import ctypes as C
        ids = (C.c_int * len(names))(*[self.F3[n] for n in names])
        self._check(self.lib.atomdev_slabs_begin_exchange(self._g, ids, len(names)))

    def finish_exchange(self):
        self._check(self.lib.atomdev_slabs_finish_exchange(self._g))

    def residuum(self, min0=0.0, loc0=(0, 0, 0)):
        import ctypes as C
        m, loc = C.c_double(min0), (C.c_int * 3)(*loc0)
        self._check(self.lib.atomdev_slabs_residuum3d(self._g, C.byref(m), loc))
        return m.value, tuple(loc)

    def steady(self):
        import ctypes as C
        nf = 9 if self.model == 0 else 6
        v, loc = (C.c_double * nf)(), (C.c_int * (3 * nf))()
        self._check(self.lib.atomdev_slabs_steady3d(self._g, v, loc))
        return list(v), [tuple(loc[3 * f:3 * f + 3]) for f in range(nf)]

    def run_3d_loop(self, pressure_iters=2, velocity_iters=2, queries=True):
        """The pressure / velocity iterations of run_3D_loop without the column physics (as SlabModel.run_3d_loop), the
        exchanges posted as early as their data is final."""
        atm = self.model == 0
        pre = "atm_" if atm else "hyd_"
        log = []
        if atm:
            self.move_data_to_new_arrays(-1)
        for _ in range(pressure_iters):
            for v in range(1, velocity_iters + 1):
                if queries:
                    first = self.residuum()
                    log.append(first)
                if atm:
                    ops = ["bc_radius", "bc_theta", "bc_phi", "value_limitation", "solid_ground", "rk3d", "value_limitation"]
                else:
                    ops = ["bc_radius", "bc_theta", "bc_phi", "pressure_density", "value_limitation", "rk3d", "solid_ground"]
                for op in ops:
                    self.op(pre + op)
                    if op == "rk3d" and v == velocity_iters:
                        self.begin_exchange(["AUX_U", "AUX_V", "AUX_W"])
                if queries:
                    log.append(self.residuum(*first) if atm else self.residuum())
                    log.append(self.steady())
                if atm:
                    for e in self.e:
                        e.msl_data(1)
                self.move_data_to_new_arrays(-1)
            self.begin_exchange(["P_DYNN"])
            self.finish_exchange()
            self.op(pre + "pressure3d")
        return log


def create_device_slab(model, im, jm, km_global, params, rank, world, dist=None, device=-1, xdev=None):
    """The product constructor: this rank's slab on its GPU.  A process that exchanges through device buffers (RCCL) must
    have imported torch before the first DeviceModel was created: torch brings a HIP runtime of its own, which the library
    shares when it is loaded first and which finds no device when it is loaded second."""
    from .device import DeviceModel
    if dist is not None and dist.get_backend() == "nccl":
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("torch sees no GPU: import torch before the first atom_amd.DeviceModel of the process is created")
    plan = SlabPlan(km_global, world)
    engine = DeviceModel(model, im, jm, plan.local_km(rank), params, device=device)
    if xdev is None:
        if dist is None or dist.get_backend() != "nccl":
            xdev = "cpu"
        else:
            import torch
            xdev = "cuda:%d" % (torch.cuda.current_device() if device < 0 else device)
    return SlabModel(engine, plan, rank, dist, xdev, model)
