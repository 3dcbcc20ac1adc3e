#!/usr/bin/env python3
"""Prints the cycle stamps a -DATOMDEV_EXP_PIPE_STAMP build of the library leaves in level 0 of aux_w
(pipelined atmosphere sweep: four epochs of one workgroup; per wave: epoch top, inputs there, work done, loads issued,
barrier reached, barrier passed)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
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
st = raw[:10 * 32].reshape(10, 4, 8)
print("kernel: %d wall ticks (100 MHz), %d cycles -> %.2f GHz" % (raw[400], raw[401], raw[401] / raw[400] / 10.))
names = ["chainA", "chainB", "prep0A", "prep0B", "prep1A", "prep1B", "prep2A", "prep2B", "prep3A", "prep3B"]
t0 = st[0, 0, 0]
for e in range(4):
    print("epoch %d" % (60 + e))
    for w in range(10):
        print("  %-6s " % names[w] + " ".join("%7d" % (st[w, e, n] - t0) if st[w, e, n] else "      -" for n in range(6)))
d.close()
