#!/usr/bin/env python3
"""Times every operator of the C-ABI on the 1-degree grid (real 0 Ma mask) with HIP events on the handle's
stream and prices it with its algorithmic bytes: tools/op_table.py [reps] -> markdown table on stdout."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import atom_amd  # noqa: E402
import harness as H  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
im, jm, km = 41, 181, 361
n3, n2 = im * jm * km, jm * km
upd = (im - 2) * (jm - 2) * (km - 2)
# algorithmic bytes per call (reads + writes of the reference's own data movement)
BYTES = {
    "atm_rk3d": upd * 280, "hyd_rk3d": upd * 208, "atm_rk3d_limited": upd * 280,
    "atm_pressure3d": upd * 48, "hyd_pressure3d": upd * 48,
    # the limiters store a value only where it changes: called again on limited fields (as here) they read their fields
    # and the mask and write p_dyn on land (the atmosphere's) or nothing
    "atm_value_limitation": n3 * (8 * 8 + 1), "hyd_value_limitation": n3 * (5 * 8),
    "move3d(atm)": n3 * 16 * 9, "move3d(hyd)": n3 * 16 * 6,
    "hyd_pressure_density": n3 * 40,
    "atm_msl_data": n3 * (7 * 8 + 1),  # second walk: c, cloud, ice, P_rain, P_snow, p_stat, t + the mask byte
    # column physics: every field the function reads plus every field it writes, once (the reference's own
    # passes re-read them: five passes in the ice scheme, up to 20 iterations per point in the adjustment)
    "atm_latent_heat": n3 * (5 * 8 + 1 + 2 * 8),
    "atm_radiation_multi_layer": n3 * (6 * 8 + 1 + 6 * 8),
    "atm_ice_water_saturation_adjustment": n3 * (5 * 8 + 1 + 8 * 8),
    "atm_two_category_ice_scheme": n3 * (7 * 8 + 1 + 11 * 8),
}
only_physics = len(sys.argv) > 2 and sys.argv[2] == "physics"
print("| operator | ms per call | algorithmic GB/s | % of 8 TB/s |")
print("|---|---|---|---|")
for model, pre in (() if only_physics else ((H.ATM, "atm"), (H.HYD, "hyd"))):
    p = H.default_params(model, im, jm, km)
    f, g = H.make_inputs(model, im, jm, km, 1234, "allfluid", p)
    names = H.ATM_FIELDS if model == H.ATM else H.HYD_FIELDS
    f = {k: f[k] for k in names}
    f["H"], g["TOPOGRAPHY"] = H.load_mask_0Ma(model)
    d = atom_amd.DeviceModel(model, im, jm, km, atom_amd.Params.from_buffer_copy(bytes(p)))
    d.upload_all(f, g)
    ops = [pre + "_" + o for o in ("bc_radius", "bc_theta", "bc_phi", "value_limitation", "solid_ground", "rk3d",
                                    "pressure3d", "rk2d", "pressure2d")] + ["move3d(%s)" % pre, "residuum3d", "steady3d"]
    if model == H.HYD:
        ops.append("hyd_pressure_density")
    else:
        ops += ["atm_msl_data", "atm_rk3d_limited"]
    for op in ops:
        def call():
            if op.startswith("move3d"):
                d.move_data_to_new_arrays(-1)
            elif op == "residuum3d":
                d.residuum(pre + "_residuum3d")
            elif op == "atm_msl_data":
                d.msl_data(1)
            elif op == "steady3d":
                d.steady(pre + "_steady3d", 9 if model == H.ATM else 6)
            else:
                d.op(op)
        for _ in range(3):
            call()
        d.timer_start()
        for _ in range(reps):
            call()
        ms = d.timer_stop() / reps
        b = BYTES.get(op)
        name = op if "(" in op or op.startswith(pre) else pre + "_" + op
        if b:
            # a call that touches less than the 256 MB last-level cache in front of the HBM finds its data there when it is
            # repeated: its figure is marked and is not an HBM figure
            mark = " (*)" if b < 256e6 else ""
            print("| %s%s | %.4f | %.0f | %.1f |" % (name, mark, ms, b / ms / 1e6, b / ms / 1e6 / 80.0))
        else:
            print("| %s | %.4f | | |" % (name, ms))
    d.close()

# the BC_Thermo column physics, on a state it keeps bounded (tests/test_loops.py); the state is restored before
# every call because the iteration counts of the adjustment depend on it
import numpy as np  # noqa: E402
import test_loops as TL  # noqa: E402
p, f, g, i_trop = TL.physics_loop_inputs(im, jm, km, 75)
i_trop[:] = 30
d = atom_amd.DeviceModel(H.ATM, im, jm, km, atom_amd.Params.from_buffer_copy(bytes(p)))
d.upload_all(f, g)
state = ("T", "C", "CLOUD", "ICE", "CN", "CLOUDN", "ICEN", "P_RAIN", "P_SNOW", "S_C_C")
calls = {
    "atm_latent_heat": lambda: d.op("atm_latent_heat"),
    "atm_radiation_multi_layer": lambda: d.radiation_multi_layer(i_trop),
    "atm_vegetation_distribution": lambda: d.vegetation_distribution(0.8),
    "atm_ice_water_saturation_adjustment": lambda: d.ice_water_saturation_adjustment(0.0),
    "atm_two_category_ice_scheme": lambda: d.two_category_ice_scheme((0.0, 0.0)),
}
for name, call in calls.items():
    ms = []
    for rep in range(min(reps, 5) + 1):
        for k in state:
            d.upload(k, f[k])
        d.timer_start()
        call()
        ms.append(d.timer_stop())
    ms = float(np.mean(ms[1:]))
    b = BYTES.get(name)
    if b:
        print("| %s | %.4f | %.0f | %.1f |" % (name, ms, b / ms / 1e6, b / ms / 1e6 / 80.0))
    else:
        print("| %s | %.4f | | |" % (name, ms))
d.close()
print()
print("(*) the call touches less than the 256 MB last-level cache in front of the HBM and is repeated on the same data: "
      "the figure is a cache figure, not an HBM figure")
