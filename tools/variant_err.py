#!/usr/bin/env python3
"""Time and per-field difference of RK sweep variants on the full 1-degree grid with the 0 Ma mask.
usage: python3 tools/variant_err.py atm|hyd <variant> [sweeps]  (difference against variant 0 after `sweeps` sweeps)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import atom_amd, harness as H
model = H.ATM if sys.argv[1] == "atm" else H.HYD
variant = int(sys.argv[2]); sweeps = int(sys.argv[3]) if len(sys.argv) > 3 else 1
im, jm, km = 41, 181, 361
p = H.default_params(model, im, jm, km)
f, g = H.make_inputs(model, im, jm, km, 1234, "allfluid", p)
names = H.ATM_FIELDS if model == H.ATM else H.HYD_FIELDS
f = {k: f[k] for k in names}
f["H"], g["TOPOGRAPHY"] = H.load_mask_0Ma(model)
op = "atm_rk3d" if model == H.ATM else "hyd_rk3d"
out = []
for v in (0, variant):
    d = atom_amd.DeviceModel(model, im, jm, km, atom_amd.Params.from_buffer_copy(bytes(p)))
    d.upload_all(f, g)
    d.set_rk_variant(v)
    for _ in range(sweeps):
        d.op(op)
    out.append({k: d.download(k) for k in names})
    d.profile_rk(True)
    for _ in range(20):
        d.op(op)
    ms, cnt = d.profile_read()
    print("variant %d: %.3f ms per sweep" % (v, ms / cnt))
    d.close()
errs = {k: H.max_rel_err(out[1][k], out[0][k]) for k in names if k != "RESIDUUM"}
print("largest differences:", sorted(errs.items(), key=lambda kv: -kv[1])[:6])
