#!/usr/bin/env python3
"""Runs a few RK sweeps on the full 1-degree grid (for rocprofv3 passes).
usage: python3 tools/run_rk.py atm|hyd [sweeps] [IMxJMxKM]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import atom_amd  # noqa: E402
import harness as H  # noqa: E402

model = H.ATM if sys.argv[1] == "atm" else H.HYD
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dims = tuple(int(x) for x in sys.argv[3].split("x")) if len(sys.argv) > 3 else (41, 181, 361)
im, jm, km = dims
p = H.default_params(model, im, jm, km)
f, g = H.make_inputs(model, im, jm, km, 1234, "allfluid", p)
names = H.ATM_FIELDS if model == H.ATM else H.HYD_FIELDS
f = {k: f[k] for k in names}
if dims == (41, 181, 361):
    f["H"], g["TOPOGRAPHY"] = H.load_mask_0Ma(model)
d = atom_amd.DeviceModel(model, im, jm, km, atom_amd.Params.from_buffer_copy(bytes(p)))
d.upload_all(f, g)
d.profile_rk(2 if os.environ.get("RUN_RK_OPERATOR") else True)  # RUN_RK_OPERATOR: the preparing kernel is timed too
for _ in range(n):
    d.op(os.environ.get("RUN_RK_OP") or ("atm_rk3d" if model == H.ATM else "hyd_rk3d"))  # RUN_RK_OP: another call that holds the sweep
    if os.environ.get("RUN_RK_MOVE"):  # a streaming copy of known size, to calibrate byte counters against
        d.move_data_to_new_arrays(-1)
ms, cnt = d.profile_read()
print("%s %s: %.3f ms per sweep (%d sweeps)" % (sys.argv[1], "x".join(map(str, dims)), ms / cnt, cnt))
d.close()
