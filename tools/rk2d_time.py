#!/usr/bin/env python3
"""Times the surface sweep of both models on the 1-degree grid (real 0 Ma mask) and, with an argument, of an
n-row surface: tools/rk2d_time.py [reps] [jm km]."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import atom_amd  # noqa: E402
import harness as H  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
jm, km = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (181, 361)
for model, pre in ((H.ATM, "atm"), (H.HYD, "hyd")):
    im = 41 if model == H.HYD else 5
    p = H.default_params(model, im, jm, km)
    mask = "allfluid" if (jm, km) == (181, 361) else "blobs"
    f, g = H.make_inputs(model, im, jm, km, 1234, mask, p)
    names = H.ATM_FIELDS if model == H.ATM else H.HYD_FIELDS
    f = {k: f[k] for k in names}
    if (jm, km) == (181, 361) and im == 41:
        f["H"], g["TOPOGRAPHY"] = H.load_mask_0Ma(model)
    d = atom_amd.DeviceModel(model, im, jm, km, atom_amd.Params.from_buffer_copy(bytes(p)))
    d.upload_all(f, g)
    for variant in (0, 1, 2):
        d.set_rk_variant(variant)
        for _ in range(3):
            d.op(pre + "_rk2d")
        d.timer_start()
        for _ in range(reps):
            d.op(pre + "_rk2d")
        ms = d.timer_stop() / reps
        print("%s_rk2d %dx%d variant %d: %.4f ms per sweep" % (pre, jm, km, variant, ms), flush=True)
    d.close()
