#!/usr/bin/env python3
"""How far a whole atmosphere time slice (1-degree grid, tests/golden/atm_slice_inputs.npz) ends from the exact form of
the sweep when the sweep runs in another form: tools/slice_variant_err.py [Ma] [variants...] -> largest per-field
relative differences against variant 0, with the location of the worst point of the worst field."""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import atom_amd  # noqa: E402
import harness as H  # noqa: E402

Ma = int(sys.argv[1]) if len(sys.argv) > 1 else 0
variants = [int(x) for x in sys.argv[2:]] or [2, 3]
prev = dict(H.ATM_SLICE_CASES)[Ma]
DIMS = (41, 181, 361)
tmp = tempfile.mkdtemp()
_, _, curve = H.write_atm_slice_inputs(tmp)
p = H.default_params(H.ATM)
s, i_trop = H.atm_slice_start(Ma, prev, curve)
name = "%dMa_Golonka.xyz" % Ma
res = {}
for variant in [0] + variants:
    d = atom_amd.DeviceModel(H.ATM, *DIMS, atom_amd.Params.from_buffer_copy(bytes(p)))
    d.set_rk_variant(variant)
    f, g = H.atm_reset_arrays(DIMS, p)
    d.upload_all(f, g)
    out = os.path.join(tmp, "out_%d" % variant)
    os.makedirs(out)
    sl = d.atm_slice(os.path.join(tmp, name), os.path.join(tmp, "SurfaceTemperature_NASA.xyz"),
                     os.path.join(tmp, "SurfacePrecipitation_NASA.xyz"), out, name,
                     coast_correction=1 if Ma > 0 else 0, mean_temperature=float(np.float32(curve[Ma])))
    for fld, _ in atom_amd.AtmStart._fields_[:-1]:
        setattr(sl.start, fld, getattr(s, fld))
    d.atm_run_time_slice(sl)
    res[variant] = {k: d.download(k) for k in f if k != "RESIDUUM"}
    d.close()
for v in variants:
    errs = {k: H.max_rel_err(res[v][k], res[0][k]) for k in res[0]}
    top = sorted(errs.items(), key=lambda kv: -kv[1])[:6]
    print("variant %d against variant 0 after the %d Ma slice:" % (v, Ma), ", ".join("%s %.2e" % kv for kv in top))
    k = top[0][0]
    diff = np.abs(res[v][k] - res[0][k])
    loc = np.unravel_index(np.argmax(diff), diff.shape)
    print("   worst point of %s: %r, %r against %r; points of it above 1e-11 of the field's maximum: %d" %
          (k, tuple(int(x) for x in loc), res[v][k][loc], res[0][k][loc], int((diff > 1e-11 * np.abs(res[0][k]).max()).sum())))

# one sweep from the slice's final state (smooth, realistic fields): the forms against the exact one, point by point
base = res[0]
one = {}
for variant in [0] + variants:
    d = atom_amd.DeviceModel(H.ATM, *DIMS, atom_amd.Params.from_buffer_copy(bytes(p)))
    d.set_rk_variant(variant)
    d.upload_all({k: v for k, v in base.items() if v.ndim == 3}, {k: v for k, v in base.items() if v.ndim == 2})
    d.op("atm_rk3d")
    one[variant] = {k: d.download(k) for k in ("T", "U", "V", "W", "C", "CLOUD", "ICE", "CO2")}
    d.close()
for v in variants:
    line = []
    for k in one[0]:
        a, b = one[v][k], one[0][k]
        scale = np.abs(b).max()
        pt = np.abs(a - b) / np.maximum(np.abs(b), 1e-30 + 1e-12 * scale)
        line.append("%s %.1e (field) %.1e (point)" % (k, np.abs(a - b).max() / scale, pt.max()))
    print("one sweep from that state, variant %d against 0: %s" % (v, "; ".join(line)))

# where the largest point-by-point differences of one sweep are
H_mask = base["H"]
for v in variants:
    for k in ("CLOUD", "ICE", "V"):
        a, b = one[v][k], one[0][k]
        scale = np.abs(b).max()
        pt = np.abs(a - b) / np.maximum(np.abs(b), 1e-30 + 1e-12 * scale)
        order = np.argsort(pt.reshape(-1))[::-1][:5]
        txt = []
        for o in order:
            i, j, kk = np.unravel_index(o, pt.shape)
            txt.append("(%d,%d,%d) %.1e x=%.3e xm1=%.3e land=%d landm1=%d" % (i, j, kk, pt[i, j, kk], b[i, j, kk], b[i, j - 1, kk],
                                                                             int(H_mask[i, j, kk] == 1.), int(H_mask[i, j - 1, kk] == 1.)))
        print("variant %d %s worst points: %s" % (v, k, "; ".join(txt)))
