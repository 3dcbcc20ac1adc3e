#!/usr/bin/env python3
"""Prints the cycle stamps a -DATOMDEV_EXP_STAMP build of the library leaves in level 0 of aux_w."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import atom_amd, harness as H
im, jm, km = 41, 181, 361
model = H.HYD if len(sys.argv) > 1 and sys.argv[1] == "hyd" else H.ATM
p = H.default_params(model, im, jm, km)
f, g = H.make_inputs(model, im, jm, km, 1234, "allfluid", p)
f = {k: f[k] for k in (H.ATM_FIELDS if model == H.ATM else H.HYD_FIELDS)}
f["H"], g["TOPOGRAPHY"] = H.load_mask_0Ma(model)
d = atom_amd.DeviceModel(model, im, jm, km, atom_amd.Params.from_buffer_copy(bytes(p)))
d.upload_all(f, g)
for _ in range(3):
    d.op("atm_rk3d" if model == H.ATM else "hyd_rk3d")
st = d.download("AUX_W").reshape(-1)[:128].reshape(8, 16)
names = ["V", "T", "U", "C", "W", "CO2", "ICE", "CLOUD"] if model == H.ATM else ["T", "V", "U", "C", "W", "-", "-", "-"]
t0 = st[:4, 0].min()
print("role  " + " ".join("%6s" % x for x in ["top", "loads", "s0 arr", "s0 go", "s1 arr", "s1 go", "s2 arr", "s2 go", "s3 arr", "s3 go", "vmwait", "s1 adv", "s1 rhs"]))
for w in range(8):
    print("%-5s " % names[w] + " ".join("%6d" % (st[w, n] - t0) for n in range(13)))
d.close()
