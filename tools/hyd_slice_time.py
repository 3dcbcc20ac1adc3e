#!/usr/bin/env python3
"""Times a whole hydrosphere time slice on the 1-degree grid (0 Ma bathymetry) with the reference's iteration counts
(10 x 2 surface iterations, 2 x 2 volume iterations; param.py:135-138), every operator on the device, from the
bathymetry and salinity files to the last pressure step: tools/hyd_slice_time.py [--cpu] (the C restatement of the
reference beside it, and the largest difference between the two after the slice)."""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import atom_amd  # noqa: E402
import harness as H  # noqa: E402

(im, jm, km), p, f, g = H.make_hyd_slice_inputs_1deg(91)
tmp = tempfile.mkdtemp()
bath, salt = os.path.join(tmp, "bath.xyz"), os.path.join(tmp, "salt.xyz")
H.write_bathymetry_file_0Ma(bath)
H.write_salinity_file(salt, jm, km, 92)
st = H.hyd_start(im)


def time_slice(m, start):
    sync = m.sync if hasattr(m, "sync") else (lambda: None)
    t = [time.perf_counter()]
    H.hyd_start_of_slice(m, start, bath, salt)
    sync(); t.append(time.perf_counter())
    H.hyd_2d_loop(m, 10, 2)
    sync(); t.append(time.perf_counter())
    H.hyd_3d_loop(m, 2, 2, run_data=True)
    sync(); t.append(time.perf_counter())
    return [1e3 * (b - a) for a, b in zip(t, t[1:])]


d = atom_amd.DeviceModel(H.HYD, im, jm, km, atom_amd.Params.from_buffer_copy(bytes(p)))
dst = atom_amd.HydStart.from_buffer_copy(bytes(st))
for rep in range(2):  # the first pass pays the lazy allocations
    t0 = time.perf_counter()
    d.upload_all(f, g)
    d.sync()
    up = 1e3 * (time.perf_counter() - t0)
    parts = time_slice(d, dst)
print("hydrosphere time slice, 41x181x361, device: upload of the transfer-file state %.1f ms; readers + initial conditions %.1f ms; "
      "surface loop (20 sweeps) %.1f ms; volume loop (4 sweeps, queries, run_data) %.1f ms; slice %.1f ms"
      % (up, parts[0], parts[1], parts[2], sum(parts)))
d.upload_all(f, g)
d.sync()
t0 = time.perf_counter()
d.run_time_slice(bath, salt)
print("the same slice through atomdev_hyd_run_time_slice (one call): %.1f ms" % (1e3 * (time.perf_counter() - t0)))
d.upload_all(f, g)
time_slice(d, dst)  # the state the comparison below expects
if "--cpu" in sys.argv:
    o = H.CpuModel("oracle", H.HYD, im, jm, km, p, f, g)
    cp = time_slice(o, st)
    print("same slice, C restatement of the reference, 1 thread: readers + initial conditions %.0f ms; surface loop %.0f ms; "
          "volume loop %.0f ms; slice %.0f ms (x%.0f)" % (cp[0], cp[1], cp[2], sum(cp), sum(cp) / sum(parts)))
    worst = 0.0
    for k in list(f) + list(g):
        if k != "RESIDUUM":
            worst = max(worst, H.max_rel_err(d.download(k), o.f[k] if k in f else o.g[k]))
    print("largest relative difference of a field after the slice: %.2e" % worst)
d.close()
