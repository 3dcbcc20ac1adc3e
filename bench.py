#!/usr/bin/env python3
"""Benchmark of the dynamical-core hot path on MI355X.

Metric (BASELINE.json): grid-point RK updates/s on the 41x181x361 fp64 grid.
One *step* = one 3-D velocity iteration of the atmosphere's hot path on synthetic
fields (seed 1234) with the real 0 Ma land mask, in the order of
cAtmosphereModel::run_3D_loop (atmosphere/cAtmosphereModel.cpp:756-874):
  BC_radius, BC_theta, BC_phi -> Value_Limitation_Atm -> BC_SolidGround ->
  solveRungeKutta_3D_Atmosphere -> Value_Limitation_Atm -> move_data_to_new_arrays,
and computePressure_3D after every second step (velocity_iter_max = 2).
One step advances 39*179*359 = 2,506,179 grid points through all four RK stages of
all eight prognostic variables.  Fields are resident in HBM before the timed
region.  With N > 1 every rank runs an independent slice on its own GPU (the
140->0 Ma sweep of independent time slices): weak scaling, no data-path collective;
the same run then times ONE 0.25-degree atmosphere split into N phi-slabs (strong
scaling, halo exchange around the pressure step) and reports it as `slabs_0p25`.

The timed region of K steps is repeated (--windows, default 5, each bracketed by the same barrier + synchronize);
`value` and `ms_per_step` are the median window's, the fastest and slowest are reported beside them.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--no-cpu]
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
  python bench.py --mode slabs --grid 0.25 --slabs 2          one model as 2 phi-slabs of this process (library-side exchange)
  python bench.py --mode slabs --grid 0.1 --slab-of 16        rank 0's slab of the 81x1801x3601 grid split 16 ways, alone
"""
import argparse
import json
import os
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

IM, JM, KM = 41, 181, 361
# degrees -> grid; the refined ones take the 0 Ma mask upsampled (nearest neighbour; 0.1 degree: 81 levels)
GRIDS = {"1": (41, 181, 361), "0.25": (41, 721, 1441), "0.1": (81, 1801, 3601)}
UPDATES = (IM - 2) * (JM - 2) * (KM - 2)
ATM_BYTES_PER_UPDATE = 280   # reads 9 state + 8 xn + h + 5 sources, writes 8 state + 3 aux + buoyancy
HYD_BYTES_PER_UPDATE = 208
HBM_PEAK_GBS = 8000.0

ATM_STEP = ["atm_bc_radius", "atm_bc_theta", "atm_bc_phi", "atm_value_limitation", "atm_solid_ground",
            "atm_rk3d", "atm_value_limitation", "move3d"]
# on the device the three boundary fills are one call (atomdev_atm_bc_radius_theta_phi) and so are limiter, solid ground,
# sweep and limiter (atomdev_atm_limit_ground_rk3d_limit): same bits as the calls one by one
ATM_DEV_STEP = ["atm_bc_radius_theta_phi", "atm_limit_ground_rk3d_limit", "move3d"]
HYD_STEP = ["hyd_bc_radius_theta_phi", "hyd_pressure_density", "hyd_value_limitation", "hyd_rk3d",
            "hyd_solid_ground", "move3d"]   # (the three boundary fills as one call: same bits)
SWEEP_KERNEL_LOAD_LIMIT = "k_atm_rk3d_wsf<true, true>"
SWEEP_KERNEL = {0: "k_atm_rk3d_ws<true>", 1: "k_atm_rk3d", 2: "k_atm_rk3d_wsf<true>", 3: "k_atm_rk3d_pipe<true>"}
SLAB_LEG_LIMIT_S = 300   # N > 1: how long the slab measurement behind the slices may take before the line is printed without it
SWEEP_FORM = {0: "exact form", 1: "one thread per line, exact form", 2: "tolerance form", 3: "pipelined tolerance form"}


def set_grid(name):
    global IM, JM, KM, UPDATES
    IM, JM, KM = GRIDS[name]
    UPDATES = (IM - 2) * (JM - 2) * (KM - 2)


def grid_mask(S, model, columns=None):
    """The real 0 Ma mask of the 1-degree grid at the points of the current grid (nearest neighbour in all three
    directions), restricted to the given global columns."""
    h, topo = S.load_mask_0Ma(model)
    if (IM, JM, KM) == GRIDS["1"] and columns is None:
        return h, topo
    ii = np.rint(np.arange(IM) * 40. / (IM - 1)).astype(int)
    jj = np.rint(np.arange(JM) * 180. / (JM - 1)).astype(int)
    kk = np.rint((np.arange(KM) if columns is None else np.asarray(columns)) * 360. / (KM - 1)).astype(int)
    return np.ascontiguousarray(h[ii][:, jj][:, :, kk]), np.ascontiguousarray(topo[jj][:, kk])


def slab_fields(S, model, columns, seed, p):
    """Seeded fields of a model that holds the given global columns (every slab its own seed), the mask at those columns."""
    f, g = S.make_inputs(model, IM, JM, len(columns), seed, "allfluid", p)
    f = {k: f[k] for k in (S.ATM_FIELDS if model == S.ATM else S.HYD_FIELDS)}
    full = (IM, JM, KM) == GRIDS["1"] and len(columns) == KM
    f["H"], g["TOPOGRAPHY"] = grid_mask(S, model, None if full else columns)
    if (IM, JM, KM) != GRIDS["1"]:
        # the 1-degree noise shrunk by (refinement)^4 (1/256 at 0.25 degree): second differences as on the 1-degree grid next to the poles
        S.shrink_noise(f, ((JM - 1) / 180.) ** 4)
    return f, g


def grid_params(atom_amd, model):
    """The product's own defaults (param.py; no checker involved) with the steps of the current grid.  On the 0.1-degree
    grid the time step goes with dthe^2 (1e-7): the reference's 1e-5 belongs to its 1-degree grid, and with it the
    coastal damping term dt v / dthe^2 is beyond the four-stage scheme's stability limit there (a sweep overflows)."""
    p = atom_amd.default_params(model, IM, JM, KM)
    if (IM, JM, KM) == GRIDS["0.1"]:
        p.dt = 1e-7
    return p


def make_model(S, atom_amd, model, device, seed):
    p = grid_params(atom_amd, model)
    f, g = slab_fields(S, model, np.arange(KM), seed, p)
    d = atom_amd.DeviceModel(model, IM, JM, KM, p, device=device)
    d.upload_all(f, g)
    d.sync()
    return d, p, f, g


def dev_step(d, seq, s, pressure_op):
    for op in seq:
        if op == "move3d":
            d.move_data_to_new_arrays(-1)
        else:
            d.op(op)
    if (s + 1) % 2 == 0:
        d.op(pressure_op)


def slab_step(sm, s):
    """One velocity iteration of a slabbed atmosphere (atom_amd.slabs.SlabModel: this process's slab among the ranks';
    LocalSlabs: all slabs in this process); the exchange for the pressure step is posted as soon as its data is final."""
    pressure = (s + 1) % 2 == 0
    target = sm if isinstance(getattr(sm, "e", None), list) else sm.e
    for op in ATM_DEV_STEP:
        if op == "move3d":
            target.move_data_to_new_arrays(-1)
        else:
            target.op(op)
            if op == "atm_limit_ground_rk3d_limit" and pressure:
                sm.begin_exchange(["AUX_U", "AUX_V", "AUX_W"])
    if pressure:
        sm.begin_exchange(["P_DYNN"])
        sm.finish_exchange()
        target.op("atm_pressure3d")


def timed_windows(step, sync, barrier, reduce_max, steps, windows, first_step=0):
    """`windows` timed regions of exactly `steps` steps each, every one bracketed by barrier + synchronize on both sides;
    returns the regions' wall times (the maximum over the ranks of each)."""
    out = []
    s = first_step
    for _ in range(windows):
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            step(s)
            s += 1
        sync()
        barrier()
        out.append(reduce_max(time.perf_counter() - t0))
    return out


def window_stats(times, units_per_window):
    med = statistics.median(times)
    return med, {"n": len(times), "seconds": times, "value_min": units_per_window / max(times), "value_max": units_per_window / min(times)}


def cpu_baseline(p, f, g, budget_s=12.0):
    """The same step on the host, 1 thread, with the C restatement of the reference
    (oracle/atom_oracle.c, bit-identical to the compiled reference; loaded through the tests' harness: checker code,
    used here as the reported baseline only).  Where the compiled reference itself travelled with the tree
    (oracle/_ref), its own solveRungeKutta_3D_Atmosphere is timed once as well."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import harness as H
    m = H.CpuModel("oracle", H.ATM, IM, JM, KM, H.Params.from_buffer_copy(bytes(p)), f, g)
    n, t0 = 0, time.perf_counter()
    rk_s = 0.0
    while True:
        for op in ATM_STEP:
            t1 = time.perf_counter()
            H.run_op(m, op)
            if op == "atm_rk3d":
                rk_s += time.perf_counter() - t1
        n += 1
        if time.perf_counter() - t0 > budget_s or n >= 4:
            break
    dt = time.perf_counter() - t0
    m.close()
    try:
        host_cores = len(os.sched_getaffinity(0))
    except AttributeError:
        host_cores = os.cpu_count()
    out = {"value": UPDATES * n / dt, "unit": "grid-point RK updates/s", "cores": 1, "kind": "port",
           "host_cores": host_cores, "host_cores_machine": os.cpu_count(),
           "sample": "%d step(s) of the same workload (0 Ma mask), %.1f s on 1 of %d cores this process may use "
                     "(the sweep is one lexicographic Gauss-Seidel chain: the reference runs it on one core); RK sweep alone "
                     "%.2f s/sweep" % (n, dt, host_cores, rk_s / n)}
    if H.have_ref():
        r = H.CpuModel("ref", H.ATM, IM, JM, KM, H.Params.from_buffer_copy(bytes(p)), f, g)
        t1 = time.perf_counter()
        r.op("atm_rk3d")
        out["reference_rk_sweep_s"] = time.perf_counter() - t1  # includes flat<->jagged Array copies of the shim
        r.close()
    return out


def measure_slabs(args, S, atom_amd, dist, rank, world, local_rank, barrier, reduce_max, steps, warmup, windows):
    """Strong scaling: ONE atmosphere on the current grid, its longitudes split into phi-slabs.  world > 1: one slab per
    rank (atom_amd.slabs.SlabModel over torch.distributed); world == 1: --slabs N slabs as N handles of this process with
    the library's own exchange (atomdev_slabs_*), or --slab-of W: slab 0 of a W-way split alone (what one GPU of W would
    hold and compute; nothing to exchange with).  Returns rank 0's record (None elsewhere)."""
    from atom_amd import slabs
    p = grid_params(atom_amd, S.ATM)
    nslabs, alone, local = world, False, None
    if world > 1:
        sm = slabs.create_device_slab(S.ATM, IM, JM, KM, p, rank, world, dist, device=local_rank, xdev=None if args.backend == "nccl" else "cpu")
        f, g = slab_fields(S, S.ATM, sm.cols, 1234 + rank, p)
        sm.e.upload_all(f, g)
        engines, own = [sm.e], sm.plan.owned[0][1] - sm.plan.owned[0][0]
        mode = "one slab per process over torch.distributed (%s)" % args.backend
    elif args.slab_of > 1:
        alone, nslabs = True, args.slab_of
        plan = slabs.SlabPlan(KM, args.slab_of)
        cols = plan.columns(0)
        e = atom_amd.DeviceModel(S.ATM, IM, JM, len(cols), p, device=local_rank)
        f, g = slab_fields(S, S.ATM, cols, 1234, p)
        e.upload_all(f, g)
        sm = slabs.SlabModel(e, slabs.SlabPlan(len(cols), 1), 0, None, "cpu", S.ATM)   # a single domain of the slab's width
        engines, own = [e], plan.owned[0][1] - plan.owned[0][0]
        mode = "slab 0 of %d alone on one GPU (%d owned + %d ghost and seam columns), no exchange partner" % (args.slab_of, own, len(cols) - own)
    else:
        nslabs = max(1, args.slabs)
        local = sm = slabs.LocalSlabs(S.ATM, IM, JM, KM, p, nslabs, devices=[local_rank] * nslabs)
        for r, e in enumerate(sm.e):
            f, g = slab_fields(S, S.ATM, sm.cols[r], 1234 + r, p)
            e.upload_all(f, g)
        engines, own = list(sm.e), sm.owned[0][1]
        mode = "%d slabs as handles of one process on one GPU, exchange by the library (second stream, events, device-to-device copies)" % nslabs
    del f, g

    def sync():
        for e in engines:
            e.sync()

    for s in range(warmup + warmup % 2):
        slab_step(sm, s)
    sync()
    engines[0].profile_rk(True)
    times = timed_windows(lambda s: slab_step(sm, s), sync, barrier, reduce_max, steps, windows)
    rk_ms, rk_n = engines[0].profile_read()
    variant = engines[0].rk_variant()
    exchanged = None
    if world > 1:
        exchanged = sum(n for o, hd, a, b2, n in sm._transfers if o == 0) * IM * JM * 8 * 4   # four fields per pressure step
    if local is not None:
        local.close()
    else:
        for e in engines:
            e.close()
    if rank != 0:
        return None
    units = (IM - 2) * (JM - 2) * (own if alone else KM - 2) * steps
    med, stats = window_stats(times, units)
    slab_updates = (IM - 2) * (JM - 2) * own
    rk_s = rk_ms / rk_n * 1e-3
    achieved = slab_updates * ATM_BYTES_PER_UPDATE / rk_s / 1e9
    return {"value": units / med, "ms_per_step": 1e3 * med / steps, "windows": stats, "slabs": nslabs, "mode": mode, "alone": alone,
            "grid": [IM, JM, KM], "owned_columns_slab0": own, "exchange_bytes_per_pressure_step_rank0": exchanged,
            "roofline": {"bound": "hbm", "kernel": "%s (slab 0: %d owned columns)" % (SWEEP_KERNEL.get(variant, "?"), own),
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": None, "bytes_per_update": ATM_BYTES_PER_UPDATE, "kernel_ms": rk_s * 1e3, "launches": rk_n,
                         "rk_variant": variant}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--windows", type=int, default=5, help="how many times the timed region of --steps steps is repeated (median reported)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo to rehearse "
                                                      "several ranks on one GPU)")
    ap.add_argument("--grid", default="1", choices=sorted(GRIDS), help="horizontal resolution in degrees (default 1: the metric's grid)")
    ap.add_argument("--mode", default="slices", choices=["slices", "slabs"],
                    help="slices (default): every GPU its own time slice, weak scaling; slabs: ONE model split into phi-slabs "
                         "with halo exchange around the pressure step, strong scaling (meant for --grid 0.25 and 0.1)")
    ap.add_argument("--slabs", type=int, default=2, help="--mode slabs in one process: how many slabs (handles) share this GPU")
    ap.add_argument("--slab-of", type=int, default=0, help="--mode slabs in one process: run slab 0 of a split this many ways, alone "
                                                           "(the grids that do not fit one GPU)")
    ap.add_argument("--no-slabs", action="store_true", help="N > 1: skip the 0.25-degree slab measurement behind the slices")
    args = ap.parse_args()
    set_grid(args.grid)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
        args.gpus = world

    import torch
    import atom_amd
    from atom_amd import synthetic as S

    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (there is no CPU fallback for the product path)")
    local_rank %= torch.cuda.device_count()  # ranks beyond the GPU count share cards (rehearsal with --backend gloo)
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=args.backend)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def reduce_max(seconds):
        if dist is None:
            return seconds
        t = torch.tensor([seconds], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def finish():
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()

    if args.mode == "slabs":
        if args.grid == "0.1" and world == 1 and args.slab_of < 2:
            sys.exit("the 0.1-degree grid (4.2 GB a field) does not fit one GPU: --slab-of W runs one slab of a W-way split")
        rec = measure_slabs(args, S, atom_amd, dist, rank, world, local_rank, barrier, reduce_max, args.steps, args.warmup, args.windows)
        if rank == 0:
            alone = rec["alone"]
            print(json.dumps({
                "metric": "grid-point RK updates/s (%dx%dx%d fp64)" % (IM, JM, KM),
                "value": rec["value"], "unit": "grid-point RK updates/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": rec["ms_per_step"], "windows": rec["windows"],
                "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
                "data": "synthetic fields (numpy PCG64 seed 1234+slab per slab), real 0 Ma land mask" + ("" if args.grid == "1" else " upsampled"),
                "config": {"workload": ("ONE slab of an atmosphere" if alone else "ONE atmosphere") + ", 3-D hot-path iteration, %dx%dx%d split into %d "
                                       "phi-slabs (3 ghost columns per edge, 5 seam copies on the edge slabs); per pressure step every slab receives "
                                       "p_dynn, aux_u, aux_v, aux_w at its ghost columns; %s" % (IM, JM, KM, rec["slabs"], rec["mode"])
                                       + ("; value counts the slab's own updates" if alone else ""),
                           "grid": [IM, JM, KM], "updates_per_step": UPDATES, "parallelism": "phi-slabs x%d" % rec["slabs"],
                           "backend": args.backend if world > 1 else "in-process", "owned_columns_slab0": rec["owned_columns_slab0"],
                           "exchange_bytes_per_pressure_step_rank0": rec["exchange_bytes_per_pressure_step_rank0"]},
                "roofline": rec["roofline"], "cpu_baseline": None}))
        finish()
        return

    atm, p, f, g = make_model(S, atom_amd, S.ATM, local_rank, 1234 + rank)
    for s in range(args.warmup + args.warmup % 2):
        dev_step(atm, ATM_DEV_STEP, s, "atm_pressure3d")
    atm.sync()
    atm.profile_rk(True)
    times = timed_windows(lambda s: dev_step(atm, ATM_DEV_STEP, s, "atm_pressure3d"), atm.sync, barrier, reduce_max, args.steps, args.windows)
    rk_ms, rk_n = atm.profile_read()
    atm.profile_rk(False)
    variant = atm.rk_variant()
    elapsed, stats = window_stats(times, UPDATES * args.steps * world)

    # outside the timed region: the sweep with the kernel that prepares its inputs in one bracket, and the other forms
    # (operator by operator, so that the bracket of the operator holds atomdev_atm_rk3d and nothing else)
    def sweep_ms(profile_mode, rk_variant, steps=40):
        atm.set_rk_variant(rk_variant)
        for s in range(4):
            dev_step(atm, ATM_STEP, s, "atm_pressure3d")
        atm.profile_rk(profile_mode)
        for s in range(steps):
            dev_step(atm, ATM_STEP, s, "atm_pressure3d")
        ms, n = atm.profile_read()
        atm.profile_rk(False)
        atm.set_rk_variant(variant)
        return ms / n

    operator_ms = sweep_ms(2, variant) if rank == 0 else None
    exact_ms = sweep_ms(1, 0) if rank == 0 else None
    pipelined_ms = sweep_ms(1, 3) if rank == 0 else None
    tolerance_ms = sweep_ms(1, 2) if rank == 0 else None

    # hydrosphere sweep, reported beside the headline (not part of `value`)
    hyd_info = None
    if rank == 0:
        hyd, _, _, _ = make_model(S, atom_amd, S.HYD, local_rank, 4321)
        for s in range(3):
            dev_step(hyd, HYD_STEP, s, "hyd_pressure3d")
        hyd.profile_rk(True)
        hyd.sync()
        th = time.perf_counter()
        nh = 100
        for s in range(nh):
            dev_step(hyd, HYD_STEP, s, "hyd_pressure3d")
        hyd.sync()
        th = time.perf_counter() - th
        hms, hn = hyd.profile_read()
        hyd_variant = hyd.rk_variant()
        # the same sweep in its other forms: 0 exact (the reference's bits), 2 wave-specialised tolerance form, 3 pipelined
        other = {}
        for v in (0, 2, 3):
            if v == hyd_variant:
                other[v] = (hms, hn)
                continue
            hyd.set_rk_variant(v)
            for s in range(4):
                dev_step(hyd, HYD_STEP, s, "hyd_pressure3d")
            hyd.profile_read()
            for s in range(40):
                dev_step(hyd, HYD_STEP, s, "hyd_pressure3d")
            other[v] = hyd.profile_read()
        hyd.close()
        hyd_traffic = None
        try:
            rec = json.load(open(os.path.join(ROOT, "profiles", "pmc_hyd_rk3d.json")))   # (of the pipelined form's kernel)
            hyd_traffic = rec.get("hbm_bytes_per_launch") if args.grid == "1" and hyd_variant == 3 and "pipe" in rec.get("kernel", "") else None
        except Exception:
            pass
        hyd_info = {"updates_per_s": UPDATES * nh / th, "ms_per_step": 1e3 * th / nh, "rk_traffic": hyd_traffic,
                    "rk_traffic_source": "profiles/pmc_hyd_rk3d.json (offline PMC pass, not measured in this run)",
                    "rk_variant": hyd_variant, "rk_form": SWEEP_FORM.get(hyd_variant, "?"),
                    "rk_sweep_ms": hms / hn,
                    "rk_hbm_frac": UPDATES * HYD_BYTES_PER_UPDATE / (hms / hn * 1e-3) / 1e9 / HBM_PEAK_GBS}
        for v, tag in ((0, "exact_form"), (2, "tolerance_form"), (3, "pipelined_form")):
            ms, n = other[v]
            hyd_info["rk_sweep_ms_" + tag] = ms / n
            hyd_info["rk_hbm_frac_" + tag] = UPDATES * HYD_BYTES_PER_UPDATE / (ms / n * 1e-3) / 1e9 / HBM_PEAK_GBS

    out = None
    if rank == 0:
        rk_avg_s = rk_ms / rk_n * 1e-3
        achieved = UPDATES * ATM_BYTES_PER_UPDATE / rk_avg_s / 1e9
        traffic, issue = None, None
        # (variant 2 inside the four-in-one call: the limiter in front of the sweep runs on the sweep's loads)
        load_limit = variant == 2 and not os.environ.get("ATOMDEV_NO_LOAD_LIMIT")
        pmc = os.path.join(ROOT, "profiles", "pmc_atm_rk3d_load_limit.json" if load_limit else "pmc_atm_rk3d.json")
        if os.path.exists(pmc) and args.grid == "1":
            try:
                rec = json.load(open(pmc))
                traffic = rec.get("hbm_bytes_per_launch")
                # the second bound of this kernel: fp64 vector instructions issue at one per 4.2 cycles per SIMD (measured),
                # 234 workgroups x 4 SIMDs are in use
                valu = rec.get("sq_insts_valu_per_launch")
                if valu:
                    simds, clock_hz, cycles = 936, 2.07e9, 4.2
                    floor_s = valu * cycles / simds / clock_hz
                    issue = {"bound": "fp64 issue", "valu_wave_insts": valu, "cycles_per_inst": cycles, "simds_used": simds,
                             "clock_ghz": clock_hz / 1e9, "floor_ms": floor_s * 1e3, "frac": floor_s / rk_avg_s,
                             "source": "profiles/%s (offline PMC pass)" % os.path.basename(pmc)}
            except Exception:
                traffic = None
        out = {
            "metric": "grid-point RK updates/s (%dx%dx%d fp64)" % (IM, JM, KM),
            "value": UPDATES * args.steps * world / elapsed,
            "unit": "grid-point RK updates/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "windows": stats,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic fields (numpy PCG64 seed 1234+rank), real 0 Ma land mask",
            "config": {"workload": "atmosphere 3-D hot-path iteration (BC fills - one call -, limiter, solid ground, fused RK4+RHS sweep, "
                                   "limiter - the four as one call: first limiter in the sweep's preparing kernel (tolerance form: there on the six faces, on the sweep's loads elsewhere), last one in its stores -, "
                                   "old->new copy; pressure Jacobi every 2nd step), %dx%dx%d, one slice per GPU; sweep in its %s "
                                   "(rk_variant %d: every field within 1e-10 of the reference)" % (IM, JM, KM, SWEEP_FORM.get(variant, "?"), variant),
                       "grid": [IM, JM, KM], "updates_per_step_per_gpu": UPDATES, "parallelism": "independent slices"},
            "roofline": {"bound": "hbm", "kernel": SWEEP_KERNEL_LOAD_LIMIT if load_limit else SWEEP_KERNEL.get(variant, "?"),
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": "profiles/%s (offline PMC pass, not measured in this run)" % os.path.basename(pmc),
                         "bytes_per_update": ATM_BYTES_PER_UPDATE, "kernel_ms": rk_avg_s * 1e3, "launches": rk_n,
                         "rk_variant": variant,
                         "operator_ms": operator_ms,    # the sweep and the kernel that prepares its pressure-height term
                         "frac_operator": UPDATES * ATM_BYTES_PER_UPDATE / (operator_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "exact_form_kernel_ms": exact_ms,
                         "pipelined_form_kernel_ms": pipelined_ms,
                         "wave_specialised_tolerance_form_kernel_ms": tolerance_ms,
                         # the timed kernel also does the work of the limiter in front of the sweep (on the old values it
                         # loads): it is priced with the sweep's 280 bytes all the same; the entry above is the sweep alone
                         "limiter_on_loads": load_limit,
                         # what a kernel that only moves the sweep's bytes in the sweep's pattern reaches (profiles/r03_stream_pattern.txt)
                         "pattern_bound_note": "35 row-strided streams read and written together by 234 workgroups reach 3.75 TB/s = 0.47 of "
                                               "peak from HBM whatever the kernel does between load and store (tools/ubench/stream_pattern.hip; offline "
                                               "measurement); on the 1-degree grid the sweeps get past that because the fields the step reads again "
                                               "(0.22 GB) stay in the 256 MB last-level cache: everything a step touches once carries the non-temporal hint",
                         "fp64_issue": issue},
            "hydrosphere": hyd_info,
        }
    atm.close()

    if rank == 0:
        out["cpu_baseline"] = None if args.no_cpu else cpu_baseline(p, f, g)

    # N > 1: the strong-scaling figure of the same run - one 0.25-degree atmosphere split into N phi-slabs.  The slices'
    # line is complete at this point and must not be lost to this leg: an exception is reported in the leg's field, and
    # a leg (or a closing barrier) that does not come back - a rank that failed alone leaves the others waiting in a
    # collective - is ended by a watchdog that prints the line as it stands and ends the process.
    import threading
    printed = threading.Lock()

    def emit():
        if printed.acquire(blocking=False) and rank == 0:
            print(json.dumps(out), flush=True)

    def give_up(what):
        if rank == 0 and out is not None and "slabs_0p25" not in out and world > 1 and not args.no_slabs:
            out["slabs_0p25"] = {"error": "%s did not finish within %d s" % (what, SLAB_LEG_LIMIT_S)}
        emit()
        sys.stdout.flush()
        os._exit(0)

    if world > 1 and not args.no_slabs:
        watchdog = threading.Timer(SLAB_LEG_LIMIT_S, give_up, args=("the 0.25-degree slab leg",))
        watchdog.daemon = True
        watchdog.start()
        rec = None
        try:
            set_grid("0.25")
            rec = measure_slabs(args, S, atom_amd, dist, rank, world, local_rank, barrier, reduce_max, steps=20, warmup=4, windows=3)
        except Exception as exc:
            rec = {"error": "%s: %s" % (type(exc).__name__, exc)}
        set_grid(args.grid)
        watchdog.cancel()
        if rank == 0:
            out["slabs_0p25"] = rec

    emit()
    if world > 1:
        closing = threading.Timer(60, give_up, args=("the closing barrier",))
        closing.daemon = True
        closing.start()
    finish()


if __name__ == "__main__":
    main()
