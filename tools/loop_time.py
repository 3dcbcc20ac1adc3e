#!/usr/bin/env python3
"""Times whole velocity iterations of the atmosphere loop (cAtmosphereModel.cpp:756-874) on the 1-degree grid with
every operator on the device, the BC_Thermo column physics included: tools/loop_time.py [pressure_iters] ->
ms per velocity iteration (wall clock around the loop, fields resident)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import atom_amd  # noqa: E402
import harness as H  # noqa: E402
import test_loops as TL  # noqa: E402

im, jm, km = 41, 181, 361
n_press = int(sys.argv[1]) if len(sys.argv) > 1 else 4
p, f, g, i_trop = TL.physics_loop_inputs(im, jm, km, 75)
i_trop[:] = 30
d = atom_amd.DeviceModel(H.ATM, im, jm, km, atom_amd.Params.from_buffer_copy(bytes(p)))
d.upload_all(f, g)
H.atm_3d_loop(d, 1, 2, physics=True, i_trop=i_trop)  # warm-up
for physics in (False, True):
    d.upload_all(f, g)
    d.sync()
    t0 = time.perf_counter()
    H.atm_3d_loop(d, n_press, 2, physics=physics, i_trop=i_trop)
    d.sync()
    ms = (time.perf_counter() - t0) * 1e3 / (2 * n_press)
    upd = (im - 2) * (jm - 2) * (km - 2)
    print("atmosphere loop %s the column physics: %.3f ms per velocity iteration (%.2f G grid-point updates/s), "
          "%d iterations" % ("with" if physics else "without", ms, upd / ms / 1e6, 2 * n_press))
d.close()

# where the time goes: the same loop with a synchronisation after every operator
import collections  # noqa: E402
d = atom_amd.DeviceModel(H.ATM, im, jm, km, atom_amd.Params.from_buffer_copy(bytes(p)))
d.upload_all(f, g)
phys = H._atm_physics(d)
spent = collections.OrderedDict()


def timed(name, fn):
    d.sync(); t0 = time.perf_counter()
    r = fn()
    d.sync(); spent[name] = spent.get(name, 0.) + (time.perf_counter() - t0) * 1e3
    return r


q, st, n_iter = 0.0, (0.0, 0.0), 0
H._move3d(d)
for pit in range(n_press):
    for vit in (1, 2):
        n_iter += 1
        mn, loc = timed("residuum3d (x2)", lambda: d.residuum("atm_residuum3d"))
        for op in ("atm_bc_radius", "atm_bc_theta", "atm_bc_phi"):
            timed("bc fills", lambda: d.op(op))
        if vit % 2 == 0:
            q = timed("saturation adjustment (every 2nd)", lambda: phys[0](q))
        timed("limiter (x2)", lambda: d.op("atm_value_limitation"))
        timed("solid ground", lambda: d.op("atm_solid_ground"))
        timed("rk3d", lambda: d.op("atm_rk3d"))
        timed("limiter (x2)", lambda: d.op("atm_value_limitation"))
        timed("radiation", lambda: phys[2](i_trop))
        timed("residuum3d (x2)", lambda: d.residuum("atm_residuum3d", mn, loc))
        timed("steady3d", lambda: d.steady("atm_steady3d", 9))
        timed("latent heat", lambda: d.op("atm_latent_heat"))
        timed("min/max + vegetation", lambda: phys[4](phys[3]()))
        timed("msl data", lambda: d.msl_data(1))
        if vit % 2 == 0:
            st = timed("ice scheme (every 2nd)", lambda: phys[1](st))
        timed("old -> new copy", lambda: H._move3d(d))
    timed("pressure3d (per pressure iteration)", lambda: d.op("atm_pressure3d"))
print("per velocity iteration, synchronised after every operator (ms):")
for k, v in spent.items():
    print("  %-40s %.3f" % (k, v / n_iter))
print("  %-40s %.3f" % ("sum", sum(spent.values()) / n_iter))
d.close()
