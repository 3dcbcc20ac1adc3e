"""Phi-slab decomposition of one model over several GPUs (one process per GPU).

The 3-D Runge-Kutta sweeps have no phi coupling (the phi derivatives of RHS_Atm.cpp:146-149 and
RHS_Hyd.cpp:136-139 are multiplied by h_d_k = 0), so a slab of longitudes can be swept without any
exchange.  What does couple neighbouring longitudes is
  * the land-mask tests of the right-hand sides (k+-1, k+-2; static),
  * the pressure Jacobi sweep (p_dynn at k+-1, aux_w at k+-1 and k+-2: Pressure_Atm.cpp:140-155),
  * the phi-seam fills, which set columns 0 and km-1 to the mean of the Neumann extrapolations
    from columns 1, 2 and km-2, km-3 (BC_Atm.cpp:115-150, Pressure_Atm.cpp:57-66, 170-176).

Layout.  Every rank holds an ordinary, smaller model (its own handle, its own km) made of
    [copies of global columns 0..S-1]   only on the last rank
    [G ghost columns]                   left neighbour's last owned columns (not on rank 0)
    [owned columns]
    [G ghost columns]                   right neighbour's first owned columns (not on the last rank)
    [copies of global columns km-S..km-1]   only on rank 0
and runs the unmodified single-domain operators on it.  With G = 3 every k-stencil of an owned
column (reach 2) and every k-window test (k >= 2, k <= km-3) evaluates exactly as in the single
domain; the seam copies make the LOCAL seam fill of the two edge ranks equal to the global one, and
S = 5 keeps the copies of columns 1, 2 / km-2, km-3 exact through one stencil application inside an
operator (the pressure step fills the seam of p_dyn right after its Jacobi sweep).  Columns next to
the junction between two segments compute garbage; they are never owned and are overwritten by the
next refresh.  Results at owned columns are bit-identical to the single-domain run.

Refresh.  Ghost and seam columns are re-sent by their owner after the operators whose results
at those columns are not reproducible locally: the RK sweeps (mask tests two columns beyond the
slab edge) and the pressure step.  Pointwise operators and the r / theta fills keep valid ghosts valid.

Not sharded: the 2-D surface loop (phi-coupled Gauss-Seidel, 65 k points) and the order-dependent
residuum / steady queries; they stay on one GPU.
"""
import numpy as np

G = 3  # ghost columns per interior slab edge
S = 5  # seam copies held by the two edge ranks

ATM_STATE = ["T", "U", "V", "W", "P_DYN", "C", "CLOUD", "ICE", "CO2"]
HYD_STATE = ["T", "U", "V", "W", "P_DYN", "C"]

# fields whose ghost copies an operator invalidates (everything else keeps valid ghosts valid)
NONLOCAL_WRITES = {
    "atm_rk3d": ["T", "U", "V", "W", "C", "CLOUD", "ICE", "CO2", "AUX_U", "AUX_V", "AUX_W", "BUOYANCY"],
    "hyd_rk3d": ["T", "U", "V", "W", "C", "AUX_U", "AUX_V", "AUX_W", "BUOYANCY", "SALT_BALANCE", "SALT_FINGER",
                 "SALT_DIFFUSION", "R_WATER", "R_SALT_WATER"],
    "atm_pressure3d": ["P_DYN", "AUX_U", "AUX_V", "AUX_W"],
    "hyd_pressure3d": ["P_DYN", "AUX_U", "AUX_V", "AUX_W"],
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
    Exchange buffers live on `xdev` ("cuda:N" for RCCL, "cpu" for gloo)."""

    def __init__(self, engine, plan, rank, dist=None, xdev="cpu"):
        self.e, self.plan, self.rank, self.dist, self.xdev = engine, plan, rank, dist, xdev
        if plan.world > 1 and dist is None:
            raise ValueError("a plan over several ranks needs torch.distributed")
        self.cols = plan.columns(rank)
        self._transfers = plan.transfers()

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

    # ---- operators ----------------------------------------------------------
    def op(self, name):
        if name in NOT_SHARDED and self.plan.world > 1:
            raise ValueError("%s is not sharded: run the surface loop on one GPU" % name)
        self.e.op(name)
        self.refresh(NONLOCAL_WRITES.get(name, ()))

    def move_data_to_new_arrays(self, level=-1):
        self.e.move_data_to_new_arrays(level)

    def refresh(self, names):
        """Owners re-send the ghost / seam copies of the named 3-D fields."""
        names = list(names)
        if self.plan.world == 1 or not names:
            return
        import torch
        for owner, holder, k_src, k_dst, n in self._transfers:
            if self.rank != owner and self.rank != holder:
                continue
            buf = torch.empty((len(names), self.e.im, self.e.jm, n), dtype=torch.float64, device=self.xdev)
            if self.rank == owner:
                for m, name in enumerate(names):
                    self.e.get_columns(name, k_src, n, out=self._view(buf[m]))
                self.dist.send(buf, dst=holder)
            else:
                self.dist.recv(buf, src=owner)
                for m, name in enumerate(names):
                    self.e.set_columns(name, k_dst, self._view(buf[m]))

    def _view(self, t):
        # engines take numpy arrays for host buffers and tensors (data_ptr) for device buffers
        return t.numpy() if t.device.type == "cpu" else t


def create_device_slab(model, im, jm, km_global, params, rank, world, dist=None, device=-1, xdev=None):
    """The product constructor: this rank's slab on its GPU."""
    from .device import DeviceModel
    plan = SlabPlan(km_global, world)
    engine = DeviceModel(model, im, jm, plan.local_km(rank), params, device=device)
    if xdev is None:
        xdev = "cpu" if dist is None or dist.get_backend() != "nccl" else "cuda:%d" % max(device, 0)
    return SlabModel(engine, plan, rank, dist, xdev)
