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
140->0 Ma sweep of independent time slices): weak scaling, no data-path collective.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--no-cpu]
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

IM, JM, KM = 41, 181, 361
GRIDS = {"1": (41, 181, 361), "0.25": (41, 721, 1441)}   # degrees -> grid; the refined one takes the 0 Ma mask upsampled by 4
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


def make_slab(H, atom_amd, dist, rank, world, device, seed, backend):
    """This rank's phi-slab of the atmosphere on the global grid (atom_amd/slabs.py): the mask columns it holds, seeded fields."""
    from atom_amd import slabs
    p = atom_amd.default_params(H.ATM, IM, JM, KM)
    sm = slabs.create_device_slab(H.ATM, IM, JM, KM, p, rank, world, dist, device=device, xdev=None if backend == "nccl" else "cpu")
    kml = sm.e.km
    f, g = H.make_inputs(H.ATM, IM, JM, kml, seed, "allfluid", p)
    f = {k: f[k] for k in H.ATM_FIELDS}
    h, topo = H.load_mask_0Ma(H.ATM)
    jj = np.rint(np.arange(JM) * 180. / (JM - 1)).astype(int)
    kk = np.rint(sm.cols * 360. / (KM - 1)).astype(int)
    f["H"], g["TOPOGRAPHY"] = np.ascontiguousarray(h[:, jj][:, :, kk]), np.ascontiguousarray(topo[jj][:, kk])
    if (IM, JM, KM) != GRIDS["1"]:
        H.shrink_noise(f, 256.)
    sm.e.upload_all(f, g)
    sm.e.sync()
    return sm


def slab_step(sm, s):
    """One velocity iteration of the slab; the exchange for the pressure step is posted as soon as its data is final."""
    pressure = (s + 1) % 2 == 0
    for op in ATM_DEV_STEP:
        if op == "move3d":
            sm.e.move_data_to_new_arrays(-1)
        else:
            sm.e.op(op)
            if op == "atm_limit_ground_rk3d_limit" and pressure:
                sm.begin_exchange(["AUX_U", "AUX_V", "AUX_W"])
    if pressure:
        sm.begin_exchange(["P_DYNN"])
        sm.finish_exchange()
        sm.e.op("atm_pressure3d")


def make_model(H, atom_amd, model, device, seed):
    p = atom_amd.default_params(model, IM, JM, KM)   # the product's own defaults (param.py), no checker involved
    f, g = H.make_inputs(model, IM, JM, KM, seed, "allfluid", p)
    names = H.ATM_FIELDS if model == H.ATM else H.HYD_FIELDS
    f = {k: f[k] for k in names}
    f["H"], g["TOPOGRAPHY"] = H.load_mask_0Ma(model)
    if (IM, JM, KM) != GRIDS["1"]:
        jj = np.rint(np.arange(JM) * 180. / (JM - 1)).astype(int)
        kk = np.rint(np.arange(KM) * 360. / (KM - 1)).astype(int)
        f["H"], g["TOPOGRAPHY"] = np.ascontiguousarray(f["H"][:, jj][:, :, kk]), np.ascontiguousarray(g["TOPOGRAPHY"][jj][:, kk])
        for k in names:   # the 1-degree noise at 1/256 of its amplitude: second differences as on the 1-degree grid next to the poles
            if k not in ("H", "RESIDUUM", "R_WATER", "R_SALT_WATER", "P_STAT"):
                m = f[k].mean()
                f[k] = m + (f[k] - m) / 256.
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


def cpu_baseline(H, p, f, g, budget_s=12.0):
    """The same step on the host, 1 thread, with the C restatement of the reference
    (oracle/atom_oracle.c, bit-identical to the compiled reference).  Where the
    compiled reference itself travelled with the tree (oracle/_ref), its own
    solveRungeKutta_3D_Atmosphere is timed once as well."""
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


def run_slabs(args, H, atom_amd, torch, dist, rank, world, local_rank, barrier):
    """Strong scaling: one atmosphere on the chosen grid, its longitudes split over the ranks."""
    sm = make_slab(H, atom_amd, dist, rank, world, local_rank, 1234 + rank, args.backend)
    for s in range(args.warmup + args.warmup % 2):
        slab_step(sm, s)
    sm.e.sync()
    sm.e.profile_rk(True)
    barrier()
    t0 = time.perf_counter()
    for s in range(args.steps):
        slab_step(sm, s)
    sm.e.sync()
    barrier()
    elapsed = time.perf_counter() - t0
    rk_ms, rk_n = sm.e.profile_read()
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        own = sm.plan.owned[0][1] - sm.plan.owned[0][0]
        slab_updates = (IM - 2) * (JM - 2) * own
        rk_s = rk_ms / rk_n * 1e-3
        achieved = slab_updates * ATM_BYTES_PER_UPDATE / rk_s / 1e9
        exchanged = sum(n for o, hd, a, b2, n in sm._transfers if o == 0) * IM * JM * 8 * 4   # four fields per pressure step
        print(json.dumps({
            "metric": "grid-point RK updates/s (%dx%dx%d fp64)" % (IM, JM, KM),
            "value": UPDATES * args.steps / elapsed, "unit": "grid-point RK updates/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic fields (numpy PCG64 seed 1234+rank per slab), real 0 Ma land mask" + ("" if args.grid == "1" else " upsampled"),
            "config": {"workload": "ONE atmosphere, 3-D hot-path iteration, %dx%dx%d split into %d phi-slabs (3 ghost columns per edge, "
                                   "5 seam copies on the edge ranks); per pressure step every rank sends p_dynn, aux_u, aux_v, aux_w at its "
                                   "neighbours' ghost columns" % (IM, JM, KM, world),
                       "grid": [IM, JM, KM], "updates_per_step": UPDATES, "parallelism": "phi-slabs x%d" % world,
                       "backend": args.backend, "exchange_bytes_per_pressure_step_rank0": exchanged},
            "roofline": {"bound": "hbm", "kernel": "k_atm_rk3d_ws (rank 0's slab of %d owned columns)" % own, "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "bytes_per_update": ATM_BYTES_PER_UPDATE, "kernel_ms": rk_s * 1e3, "launches": rk_n,
                         "rk_variant": sm.e.rk_variant()},
            "cpu_baseline": None}))
    sm.e.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo to rehearse "
                                                      "several ranks on one GPU)")
    ap.add_argument("--grid", default="1", choices=sorted(GRIDS), help="horizontal resolution in degrees (default 1: the metric's grid)")
    ap.add_argument("--mode", default="slices", choices=["slices", "slabs"],
                    help="slices (default): every GPU its own time slice, weak scaling; slabs: ONE model split into phi-slabs over "
                         "the GPUs with halo exchange around the pressure step, strong scaling (meant for --grid 0.25)")
    args = ap.parse_args()
    global IM, JM, KM, UPDATES
    IM, JM, KM = GRIDS[args.grid]
    UPDATES = (IM - 2) * (JM - 2) * (KM - 2)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
        args.gpus = world

    import torch
    import atom_amd
    import harness as H

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

    if args.mode == "slabs":
        run_slabs(args, H, atom_amd, torch, dist, rank, world, local_rank, barrier)
        return
    atm, p, f, g = make_model(H, atom_amd, H.ATM, local_rank, 1234 + rank)

    for s in range(args.warmup):
        dev_step(atm, ATM_DEV_STEP, s, "atm_pressure3d")
    atm.sync()

    atm.profile_rk(True)
    barrier()
    t0 = time.perf_counter()
    for s in range(args.steps):
        dev_step(atm, ATM_DEV_STEP, s, "atm_pressure3d")
    atm.sync()
    barrier()
    elapsed = time.perf_counter() - t0
    rk_ms, rk_n = atm.profile_read()
    atm.profile_rk(False)
    variant = atm.rk_variant()

    # outside the timed region: the sweep with the kernel that prepares its inputs in one bracket, and the exact form
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

    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # hydrosphere sweep, reported beside the headline (not part of `value`)
    hyd_info = None
    if rank == 0:
        hyd, _, _, _ = make_model(H, atom_amd, H.HYD, local_rank, 4321)
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
        # the same sweep in its tolerance form (ATOMDEV_RK_VARIANT=2; the hydrosphere's default is the exact form)
        hyd.set_rk_variant(2)
        for s in range(3):
            dev_step(hyd, HYD_STEP, s, "hyd_pressure3d")
        hyd.profile_read()
        for s in range(40):
            dev_step(hyd, HYD_STEP, s, "hyd_pressure3d")
        hms2, hn2 = hyd.profile_read()
        hyd.close()
        hyd_traffic = None
        try:
            hyd_traffic = json.load(open(os.path.join(ROOT, "profiles", "pmc_hyd_rk3d.json"))).get("hbm_bytes_per_launch") if args.grid == "1" else None
        except Exception:
            pass
        hyd_info = {"updates_per_s": UPDATES * nh / th, "ms_per_step": 1e3 * th / nh, "rk_traffic": hyd_traffic,
                    "rk_traffic_source": "profiles/pmc_hyd_rk3d.json (offline PMC pass, not measured in this run)",
                    "rk_sweep_ms": hms / hn,
                    "rk_hbm_frac": UPDATES * HYD_BYTES_PER_UPDATE / (hms / hn * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "rk_sweep_ms_tolerance_form": hms2 / hn2,
                    "rk_hbm_frac_tolerance_form": UPDATES * HYD_BYTES_PER_UPDATE / (hms2 / hn2 * 1e-3) / 1e9 / HBM_PEAK_GBS}

    if rank == 0:
        rk_avg_s = rk_ms / rk_n * 1e-3
        achieved = UPDATES * ATM_BYTES_PER_UPDATE / rk_avg_s / 1e9
        traffic, issue = None, None
        pmc = os.path.join(ROOT, "profiles", "pmc_atm_rk3d.json")
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
                             "source": "profiles/pmc_atm_rk3d.json (offline PMC pass)"}
            except Exception:
                traffic = None
        out = {
            "metric": "grid-point RK updates/s (%dx%dx%d fp64)" % (IM, JM, KM),
            "value": UPDATES * args.steps * world / elapsed,
            "unit": "grid-point RK updates/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic fields (numpy PCG64 seed 1234+rank), real 0 Ma land mask",
            "config": {"workload": "atmosphere 3-D hot-path iteration (BC fills - one call -, limiter, solid ground, fused RK4+RHS sweep, "
                                   "limiter - the four as one call: first limiter in the sweep's preparing kernel, last one in its stores -, "
                                   "old->new copy; pressure Jacobi every 2nd step), %dx%dx%d, one slice per GPU" % (IM, JM, KM),
                       "grid": [IM, JM, KM], "updates_per_step_per_gpu": UPDATES, "parallelism": "independent slices"},
            "roofline": {"bound": "hbm", "kernel": ("k_atm_rk3d_wsf<true>" if variant == 2 else "k_atm_rk3d_ws<true>") if variant != 1 else "k_atm_rk3d",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": "profiles/pmc_atm_rk3d.json (offline PMC pass, not measured in this run)",
                         "bytes_per_update": ATM_BYTES_PER_UPDATE, "kernel_ms": rk_avg_s * 1e3, "launches": rk_n,
                         "rk_variant": variant,
                         "operator_ms": operator_ms,    # the sweep and the kernel that prepares its pressure-height term
                         "frac_operator": UPDATES * ATM_BYTES_PER_UPDATE / (operator_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "exact_form_kernel_ms": exact_ms,
                         "fp64_issue": issue},
            "hydrosphere": hyd_info,
        }
        if not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(H, p, f, g)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    atm.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
