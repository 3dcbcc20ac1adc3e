#!/usr/bin/env python3
"""How well do independent models on one GPU overlap?  tools/multi_handle.py <op> [max handles]: N atmosphere handles
(one stream each), the operator enqueued on all of them, wall time until all have finished."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import atom_amd, harness as H
op = sys.argv[1] if len(sys.argv) > 1 else "atm_rk2d"
nmax = int(sys.argv[2]) if len(sys.argv) > 2 else 16
im, jm, km = 41, 181, 361
p = H.default_params(H.ATM, im, jm, km)
f, g = H.make_inputs(H.ATM, im, jm, km, 1234, "allfluid", p)
f = {k: f[k] for k in H.ATM_FIELDS}
f["H"], g["TOPOGRAPHY"] = H.load_mask_0Ma(H.ATM)
models = []
for n in range(nmax):
    d = atom_amd.DeviceModel(H.ATM, im, jm, km, atom_amd.Params.from_buffer_copy(bytes(p)))
    d.upload_all(f, g)
    models.append(d)
for n in (1, 2, 4, 8, 16, 32):
    if n > nmax: break
    for d in models[:n]: d.op(op)
    for d in models[:n]: d.sync()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        for d in models[:n]: d.op(op)
    for d in models[:n]: d.sync()
    dt = (time.perf_counter() - t0) / reps
    print("%s on %2d handles: %.3f ms per round, %.3f ms per handle" % (op, n, dt * 1e3, dt * 1e3 / n))
    if op == "atm_rk2d":
        atom_amd.rk2d_batch(models[:n])
        for d in models[:n]: d.sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            atom_amd.rk2d_batch(models[:n])
        for d in models[:n]: d.sync()
        dt = (time.perf_counter() - t0) / reps
        print("   batched launch    : %.3f ms per round, %.3f ms per handle" % (dt * 1e3, dt * 1e3 / n))
