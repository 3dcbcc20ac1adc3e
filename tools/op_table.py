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
    "atm_rk3d": upd * 280, "hyd_rk3d": upd * 208,
    "atm_pressure3d": upd * 48, "hyd_pressure3d": upd * 48,
    "atm_value_limitation": n3 * 152, "hyd_value_limitation": n3 * 88,
    "move3d(atm)": n3 * 16 * 9, "move3d(hyd)": n3 * 16 * 6,
    "hyd_pressure_density": n3 * 40,
    "atm_msl_data": n3 * (7 * 8 + 1),  # second walk: c, cloud, ice, P_rain, P_snow, p_stat, t + the mask byte
}
print("| operator | ms per call | algorithmic GB/s | % of 8 TB/s |")
print("|---|---|---|---|")
for model, pre in ((H.ATM, "atm"), (H.HYD, "hyd")):
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
        ops.append("atm_msl_data")
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
            print("| %s | %.4f | %.0f | %.1f |" % (name, ms, b / ms / 1e6, b / ms / 1e6 / 80.0))
        else:
            print("| %s | %.4f | | |" % (name, ms))
    d.close()
