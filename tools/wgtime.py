#!/usr/bin/env python3
"""Prints per-workgroup durations a -DATOMDEV_EXP_WGTIME build leaves in level 0 of aux_u (cycles), by level."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import atom_amd, harness as H
im, jm, km = 41, 181, 361
M = H.HYD if "hyd" in sys.argv[1:] else H.ATM
p = H.default_params(M, im, jm, km)
f, g = H.make_inputs(M, im, jm, km, 1234, "allfluid", p)
f = {k: f[k] for k in (H.ATM_FIELDS if M == H.ATM else H.HYD_FIELDS)}
if "allfluid" not in sys.argv[1:]:
    f["H"], g["TOPOGRAPHY"] = H.load_mask_0Ma(M)
d = atom_amd.DeviceModel(M, im, jm, km, atom_amd.Params.from_buffer_copy(bytes(p)))
d.upload_all(f, g)
for _ in range(3):
    d.op("atm_rk3d" if M == H.ATM else "hyd_rk3d")
a = d.download("AUX_U").reshape(-1)
gx = 6
for base, tag in ((0, "first kernel"), (16384, "second kernel")):
    t = a[base:base + 39 * gx * 8].reshape(39, gx, 8)
    if t.max() <= 0: continue
    print(tag, "max over waves per workgroup, kilo-cycles; rows = level 1..39, columns = longitude block")
    for lev in range(39):
        print("%2d " % (lev + 1) + " ".join("%6.0f" % (t[lev, b].max() / 1e3) for b in range(gx)))
t0 = a[16384:16384 + 39 * gx]; t1 = a[32768:32768 + 39 * gx]
if t0.max() > 0:
    base = t0.min()
    print("start of workgroups after the first one, kilo-ticks: min %.1f median %.1f max %.1f" % (0, np.median(t0 - base) / 1e3, (t0 - base).max() / 1e3))
    print("end: min %.1f median %.1f max %.1f; ticks are 10 ns" % ((t1 - base).min() / 1e3, np.median(t1 - base) / 1e3, (t1 - base).max() / 1e3))
    print("per level: start .. end of its workgroups in microseconds after the first start (min start, max end)")
    for lev in range(39):
        a0 = (t0[lev * gx:(lev + 1) * gx] - base) / 100.; a1 = (t1[lev * gx:(lev + 1) * gx] - base) / 100.
        print("%2d  %7.1f .. %7.1f" % (lev + 1, a0.min(), a1.max()))
d.close()
