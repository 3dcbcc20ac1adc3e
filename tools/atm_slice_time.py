#!/usr/bin/env python3
"""Times a whole atmosphere time slice on the 1-degree grid through atomdev_atm_run_time_slice (one call: input files,
initial conditions, 2 x 10 surface and 2 x 2 volume iterations with all column physics, the six output files), with the
input files of tests/golden/atm_slice_inputs.npz: tools/atm_slice_time.py [Ma] [reps]."""
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import atom_amd  # noqa: E402
import harness as H  # noqa: E402

Ma = int(sys.argv[1]) if len(sys.argv) > 1 else 0
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
prev = dict(H.ATM_SLICE_CASES)[Ma]
DIMS = (41, 181, 361)
tmp = tempfile.mkdtemp()
_, _, curve = H.write_atm_slice_inputs(tmp)
p = H.default_params(H.ATM)
s, i_trop = H.atm_slice_start(Ma, prev, curve)
name = "%dMa_Golonka.xyz" % Ma
for variant in (2, 0):
    d = atom_amd.DeviceModel(H.ATM, *DIMS, atom_amd.Params.from_buffer_copy(bytes(p)))
    d.set_rk_variant(variant)
    for rep in range(reps):
        f, g = H.atm_reset_arrays(DIMS, p)
        d.upload_all(f, g)
        d.sync()
        out = os.path.join(tmp, "out_%d_%d" % (variant, rep))
        os.makedirs(out)
        sl = d.atm_slice(os.path.join(tmp, name), os.path.join(tmp, "SurfaceTemperature_NASA.xyz"),
                         os.path.join(tmp, "SurfacePrecipitation_NASA.xyz"), out, name,
                         coast_correction=1 if Ma > 0 else 0, mean_temperature=float(np.float32(curve[Ma])))
        for fld, _ in atom_amd.AtmStart._fields_[:-1]:
            setattr(sl.start, fld, getattr(s, fld))
        t0 = time.perf_counter()
        d.atm_run_time_slice(sl)
        d.sync()
        ms = 1e3 * (time.perf_counter() - t0)
        print("atmosphere time slice %d Ma, 41x181x361, sweep variant %d, run %d: %.1f ms" % (Ma, variant, rep, ms), flush=True)
    d.close()
