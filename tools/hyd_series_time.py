#!/usr/bin/env python3
"""A series of independent hydrosphere time slices on one GPU (the paleo series: one slice per Ma): N handles advanced
in lockstep by atomdev_hyd_run_time_slices against N calls of atomdev_hyd_run_time_slice one after the other.
tools/hyd_series_time.py [N ...]   (1-degree grid, 0 Ma bathymetry, the reference's iteration counts)"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import atom_amd  # noqa: E402
import harness as H  # noqa: E402

counts = [int(x) for x in sys.argv[1:]] or [1, 4, 8]
(im, jm, km), p, f, g = H.make_hyd_slice_inputs_1deg(91)
tmp = tempfile.mkdtemp()
bath, salt = os.path.join(tmp, "bath.xyz"), os.path.join(tmp, "salt.xyz")
H.write_bathymetry_file_0Ma(bath)
H.write_salinity_file(salt, jm, km, 92)
models = []
for n in range(max(counts)):
    d = atom_amd.DeviceModel(H.HYD, im, jm, km, atom_amd.Params.from_buffer_copy(bytes(p)))
    models.append(d)


def reload(ms):
    for d in ms:
        d.upload_all(f, g)
    for d in ms:
        d.sync()


reload(models)
atom_amd.hyd_run_time_slices(models, [(bath, salt, None, 0)] * len(models))  # lazy allocations
for n in counts:
    ms = models[:n]
    reload(ms)
    t0 = time.perf_counter()
    for d in ms:
        d.run_time_slice(bath, salt)
    one_by_one = 1e3 * (time.perf_counter() - t0)
    reload(ms)
    t0 = time.perf_counter()
    atom_amd.hyd_run_time_slices(ms, [(bath, salt, None, 0)] * n)
    lockstep = 1e3 * (time.perf_counter() - t0)
    print("%2d slices: one after the other %.1f ms (%.1f per slice); in lockstep %.1f ms (%.1f per slice)"
          % (n, one_by_one, one_by_one / n, lockstep, lockstep / n), flush=True)
for d in models:
    d.close()
