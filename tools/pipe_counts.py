#!/usr/bin/env python3
"""Prints what a -DATOMDEV_EXP_PIPE_COUNT=<level> build of the library leaves in level 0 of aux_w (pipelined atmosphere
sweep, one workgroup of that level): how often each chain wave found its inputs not there yet and why (packet not ready /
slot's results not stored yet / the other chain not far enough), and how often the memory waves polled for their chain."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import atom_amd, harness as H
im, jm, km = 41, 181, 361
p = H.default_params(H.ATM, im, jm, km)
f, g = H.make_inputs(H.ATM, im, jm, km, 1234, "allfluid", p)
f = {k: f[k] for k in H.ATM_FIELDS}
f["H"], g["TOPOGRAPHY"] = H.load_mask_0Ma(H.ATM)
d = atom_amd.DeviceModel(H.ATM, im, jm, km, atom_amd.Params.from_buffer_copy(bytes(p)))
d.upload_all(f, g)
d.set_rk_variant(3)
for _ in range(3):
    d.op("atm_rk3d")
raw = d.download("AUX_W").reshape(-1)
print("179 rows.  chain A: %d failed polls (packet %d, stored %d, other chain %d); chain B: %d (packet %d, stored %d, chain A %d)"
      % tuple(int(x) for x in raw[:8]))
print("memory waves' polls for their chain (pair 0..3, A and B):", [int(x) for x in raw[16:24]])
d.close()
