#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the COMPILED REFERENCE (oracle/_ref/libatomref.so).

Run in the build container only (needs /root/reference):  python oracle/make_golden.py
The outputs are data: inputs are re-derived from the seeds by tests/harness.py
(their SHA-256 is stored so drift of the generator is detected), expected outputs
are what the reference's own classes computed.
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import harness as H  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
XYZ = "/root/reference/data/Paleotopography_bathymetry/Golonka_rev210/0Ma_Golonka.xyz"


def digest(fields, names):
    h = hashlib.sha256()
    for k in names:
        h.update(np.ascontiguousarray(fields[k]).tobytes())
    return h.hexdigest()


def small_case(model, dims, seq, seed, fname, level=None):
    im, jm, km = dims
    p = H.default_params(model, im, jm, km)
    f, g = H.make_inputs(model, im, jm, km, seed, "blobs", p)
    names = H.ATM_FIELDS if model == H.ATM else H.HYD_FIELDS
    f = {k: f[k] for k in names}
    m = H.CpuModel("ref", model, im, jm, km, p, f, g)
    lvl = None if level is None else (im - 1 if level == "top" else 0)
    snaps = H.run_sequence(m, seq, names, lvl)
    snaps["input_sha256"] = np.array(digest(f, names) + digest(g, H.FIELDS2D))
    snaps["dims"] = np.array(dims)
    snaps["seed"] = np.array(seed)
    if model == H.HYD and level == "top":
        snaps["p_dyn_level0_lastrow"] = m.f["P_DYN"][0, jm - 1, :].copy()  # Pressure_Hyd.cpp:304
    np.savez_compressed(os.path.join(GOLD, fname), **snaps)
    print(fname, len(snaps), "entries")


def residuum_case():
    im, jm, km = 7, 13, 17
    p = H.default_params(H.ATM, im, jm, km)
    f, g = H.make_inputs(H.ATM, im, jm, km, 77, "blobs", p)
    f = {k: f[k] for k in H.ATM_FIELDS}
    # second velocity set for the second query on the same object
    f["UN"], f["VN"], f["WN"] = -0.5 * f["U"], 0.25 * f["V"], -0.125 * f["W"]
    m = H.CpuModel("ref", H.ATM, im, jm, km, p, f, g)
    out = {}
    mn, loc = m.residuum("atm_residuum3d")
    out["res3d_min"], out["res3d_loc"], out["res3d_field"] = np.array(mn), np.array(loc), m.f["RESIDUUM"].copy()
    import ctypes as C
    mins = (C.c_double * 2)(); locs = (C.c_int * 6)()
    m.op("atm_residuum3d_twice", mins, locs)
    out["res3d_twice_min"], out["res3d_twice_loc"] = np.array(mins), np.array(locs)
    mn, loc = m.residuum("atm_residuum2d")
    out["res2d_min"], out["res2d_loc"], out["res2d_field"] = np.array(mn), np.array(loc), m.f["RESIDUUM"][0].copy()
    ph = H.default_params(H.HYD, im, jm, km)
    fh, gh = H.make_inputs(H.HYD, im, jm, km, 78, "blobs", ph)
    fh = {k: fh[k] for k in H.HYD_FIELDS}
    mh = H.CpuModel("ref", H.HYD, im, jm, km, ph, fh, gh)
    mn, loc = mh.residuum("hyd_residuum3d")
    out["hyd_res3d_min"], out["hyd_res3d_loc"] = np.array(mn), np.array(loc)
    np.savez_compressed(os.path.join(GOLD, "residuum_small.npz"), **out)
    print("residuum_small.npz")


def masks():
    out = {}
    for model, tag in ((H.ATM, "atm"), (H.HYD, "hyd")):
        im, jm, km = 41, 181, 361
        p = H.default_params(model)
        f = {"H": np.zeros((im, jm, km))}
        g = {"TOPOGRAPHY": np.zeros((jm, km))}
        m = H.CpuModel("ref", model, im, jm, km, p, f, g)
        m.lib.ref_read_mask(m.ctx, XYZ.encode())
        h = m.f["H"]
        assert set(np.unique(h)) <= {0.0, 1.0}
        out[tag + "_h_bits"] = np.packbits(h.astype(np.uint8).ravel())
        topo = m.g["TOPOGRAPHY"]
        out[tag + "_topo"] = topo
        print(tag, "land cells", int(h.sum()))
    np.savez_compressed(os.path.join(GOLD, "mask_0Ma.npz"), **out)


def full_case(model, tag):
    im, jm, km = 41, 181, 361
    p = H.default_params(model)
    f, g = H.make_inputs(model, im, jm, km, 1234, "allfluid", p)
    names = H.ATM_FIELDS if model == H.ATM else H.HYD_FIELDS
    f = {k: f[k] for k in names}
    h, topo = H.load_mask_0Ma(model)
    f["H"], g["TOPOGRAPHY"] = h, topo
    m = H.CpuModel("ref", model, im, jm, km, p, f, g)
    before = {k: m.f[k].copy() for k in names}
    m.op("atm_rk3d" if model == H.ATM else "hyd_rk3d")
    out = {"input_sha256": np.array(digest(f, names) + digest(g, H.FIELDS2D))}
    for k in names:
        if not np.array_equal(m.f[k], before[k]):
            a = m.f[k]
            out[k + "/sha256"] = np.array(hashlib.sha256(a.tobytes()).hexdigest())
            out[k + "/stats"] = np.array([a.sum(), a.min(), a.max(), np.abs(a).max()])
            for pk, pv in H.sample_planes(a).items():
                out[k + "/" + pk] = pv
    np.savez_compressed(os.path.join(GOLD, "full_1deg_%s_rk3d.npz" % tag), **out)
    print("full", tag, len(out))


def msl_case():
    """Results_MSL_Atm::run_MSL_data on the 1-degree surface with five levels (the function prints at fixed
    1-degree locations): SHA-256 and statistics of every field it changed, plus sample slices."""
    import ctypes as C
    im, jm, km = 5, 181, 361
    p = H.default_params(H.ATM, im, jm, km)
    f, g = H.make_msl_inputs(im, jm, km, 51, p)
    m = H.CpuModel("ref", H.ATM, im, jm, km, p, f, g)
    m.op("atm_msl_data", C.c_int(1))
    out = {"input_sha256": np.array(digest(f, sorted(f)) + digest(g, sorted(g)))}
    for k in sorted(f):
        if not np.array_equal(m.f[k], f[k]):
            a = m.f[k]
            out[k + "/sha256"] = np.array(hashlib.sha256(a.tobytes()).hexdigest())
            out[k + "/stats"] = np.array([a.sum(), a.min(), a.max(), np.abs(a).max()])
            out[k + "/j62"] = a[:, 62, :].copy()
    for k in sorted(g):
        if not np.array_equal(m.g[k], g[k]):
            a = m.g[k]
            out[k + "/sha256"] = np.array(hashlib.sha256(a.tobytes()).hexdigest())
            out[k + "/stats"] = np.array([a.sum(), a.min(), a.max(), np.abs(a).max()])
            out[k + "/s4"] = a[::4, ::4].copy()
    np.savez_compressed(os.path.join(GOLD, "atm_msl_data.npz"), **out)
    print("atm_msl_data.npz", len(out), "entries")


def transfer_case():
    """Atmosphere -> hydrosphere transfer through the reference's own writer and reader: what arrives in the
    hydrosphere, and the SHA-256 of the file the reference wrote."""
    import tempfile
    (pa, fa, ga), (ph, fh, gh) = H.make_transfer_inputs((5, 13, 17), (6, 13, 17), 61)
    ra, rb = H.CpuModel("ref", H.ATM, 5, 13, 17, pa, fa, ga), H.CpuModel("ref", H.HYD, 6, 13, 17, ph, fh, gh)
    with tempfile.TemporaryDirectory() as d:
        ra.lib.ref_transfer_atm_to_hyd(ra.ctx, rb.ctx, d.encode())
        text = open(os.path.join(d, "[bridge]_Transfer_Atm.vw"), "rb").read()
    out = {"input_sha256": np.array(digest(fa, sorted(fa)) + digest(ga, sorted(ga)) + digest(fh, sorted(fh)) + digest(gh, sorted(gh))),
           "file_sha256": np.array(hashlib.sha256(text).hexdigest()), "file_head": np.array(text[:600].decode())}
    for k in ("V", "W", "T", "P_DYN"):
        out[k] = rb.f[k][5].copy()
    for k in ("EVAPORATION_DALTON", "PRECIPITATION"):
        out[k] = rb.g[k].copy()
    np.savez_compressed(os.path.join(GOLD, "transfer_small.npz"), **out)
    print("transfer_small.npz", len(out), "entries")


def thermo_case():
    """BC_Thermo's in-loop functions (atmosphere/BC_Thermo.cpp:166-424, 2418-3143, 3235-3275) through the compiled class:
    SHA-256 of every field each call changed, the members carried between calls, one slice per field."""
    out = H.thermo_golden_run("ref")
    np.savez_compressed(os.path.join(GOLD, "bc_thermo.npz"), **out)
    print("bc_thermo.npz", len(out), "entries")


def atm_slice_inputs():
    """The reference's own input files of the atmosphere as compact arrays (tests/harness.py writes them back digit for
    digit): the three 1-degree surface files (temperature, precipitation, salinity) in hundredths, the Golonka topography of the golden time slices (the doubles
    their text stands for), and the mean temperatures the model's curve gives for the time slices involved."""
    import ctypes as C
    data = "/root/reference/data"
    out = {}
    for key, name in (("temperature_c100", "SurfaceTemperature_NASA.xyz"), ("precipitation_c100", "SurfacePrecipitation_NASA.xyz"),
                      ("salinity_c100", "SurfaceSalinity_NASA.xyz")):
        v = np.loadtxt(os.path.join(data, name))[:, 2]
        c = np.round(v * 100.).astype(np.int64)
        assert np.abs(c).max() < 2 ** 31 and np.array_equal(c / 100., v)
        out[key] = c.astype(np.int32)
    for Ma in sorted({m for case in H.ATM_SLICE_CASES for m in case[:1]}):
        v = np.loadtxt(os.path.join(data, "Paleotopography_bathymetry/Golonka_rev210/%dMa_Golonka.xyz" % Ma))[:, 2]
        out["topography_%d" % Ma] = v
    fn = H.load_ref().ref_atm_mean_temperature_from_curve
    fn.restype = C.c_float; fn.argtypes = [C.c_float]
    times = sorted({m for case in H.ATM_SLICE_CASES for m in case if m is not None} | {0})
    out["curve_Ma"] = np.array(times, dtype=np.int32)
    out["curve_temperature"] = np.array([fn(float(m)) for m in times], dtype=np.float32)
    np.savez_compressed(H.ATM_SLICE_INPUTS, **out)
    print("atm_slice_inputs.npz", len(out), "entries")
    # the files written back are the reference's, byte for byte
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        H.write_atm_slice_inputs(tmp)
        for name, ref in [(n, n) for n in ("SurfaceTemperature_NASA.xyz", "SurfacePrecipitation_NASA.xyz", "SurfaceSalinity_NASA.xyz")] + \
                         [("%dMa_Golonka.xyz" % m, "Paleotopography_bathymetry/Golonka_rev210/%dMa_Golonka.xyz" % m) for m, _ in H.ATM_SLICE_CASES]:
            a, b = open(os.path.join(tmp, name), "rb").read(), open(os.path.join(data, ref), "rb").read()
            print("  %s: %s" % (name, "same bytes" if a == b else "same values" if np.array_equal(np.loadtxt(os.path.join(tmp, name))[:, 2], np.loadtxt(os.path.join(data, ref))[:, 2]) else "DIFFERENT"))


def atm_slice_case():
    """cAtmosphereModel::RunTimeSlice of the compiled reference (plate reconstruction off) for the golden time slices at the
    reference's iteration counts, from the input files as tests/harness.py writes them back."""
    import tempfile
    out = {}
    dims = (41, 181, 361)
    with tempfile.TemporaryDirectory() as tmp:
        _, _, curve = H.write_atm_slice_inputs(tmp)
        for Ma, prev in H.ATM_SLICE_CASES:
            p = H.default_params(H.ATM)
            f, g = H.atm_reset_arrays(dims, p)
            r = H.CpuModel("ref", H.ATM, *dims, p, f, g)
            s, keep = H.atm_slice_start(Ma, prev, curve)
            outdir = os.path.join(tmp, "out_%d" % Ma)
            os.makedirs(outdir)
            H.run_ref_atm_slice(r, s, (2, 10, 2, 2), tmp, outdir)
            out.update(H.atm_slice_record(r, outdir, "Ma%d" % Ma))
            print("  Ma", Ma, "done")
    np.savez_compressed(H.ATM_SLICE_GOLDEN, **out)
    print("atm_time_slice.npz", len(out), "entries")


def hyd_slice_case():
    """cHydrosphereModel::RunTimeSlice of the compiled reference at 0 Ma behind the compiled reference's atmosphere slice
    (which leaves the transfer file it reads), both with the reference's default parameters and iteration counts, from
    the input files as tests/harness.py writes them back: SHA-256 and a digest of the numbers of every file the
    hydrosphere writes (its checkpoint files and its final ones)."""
    import ctypes as C
    import hashlib
    import tempfile
    out = {}
    dims = (41, 181, 361)
    with tempfile.TemporaryDirectory() as tmp:
        _, _, curve = H.write_atm_slice_inputs(tmp)
        p = H.default_params(H.ATM)
        f, g = H.atm_reset_arrays(dims, p)
        r = H.CpuModel("ref", H.ATM, *dims, p, f, g)
        s, keep = H.atm_slice_start(0, None, curve)
        outdir = os.path.join(tmp, "out")
        os.makedirs(outdir)
        H.run_ref_atm_slice(r, s, (2, 10, 2, 2), tmp, outdir)
        before = set(os.listdir(outdir))
        lib = H.load_ref()
        lib.ref_hyd_run_time_slice.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int]
        assert lib.ref_hyd_run_time_slice(tmp.encode(), os.path.join(tmp, "SurfaceSalinity_NASA.xyz").encode(), outdir.encode(), 0) == 0
        for fn in sorted(set(os.listdir(outdir)) - before):
            path = os.path.join(outdir, fn)
            out["Ma0/file/" + fn] = np.array(hashlib.sha256(open(path, "rb").read()).hexdigest())
            out["Ma0/numbers/" + fn] = H.text_file_numbers(path)
            print("  ", fn, os.path.getsize(path))
    np.savez_compressed(H.HYD_SLICE_GOLDEN, **out)
    print("hyd_time_slice.npz", len(out), "entries")


if __name__ == "__main__":
    assert H.have_ref(), "build oracle/_ref first: make -C oracle ref"
    os.makedirs(GOLD, exist_ok=True)
    if len(sys.argv) > 1:   # single cases by name
        for name in sys.argv[1:]:
            globals()[name]()
        sys.exit(0)
    masks()
    small_case(H.ATM, (7, 13, 17), H.ATM_SEQ, 11, "atm_small.npz")
    small_case(H.HYD, (7, 13, 17), H.HYD_SEQ, 12, "hyd_small.npz")
    small_case(H.HYD, (41, 43, 9), H.HYD_SURFACE_SEQ, 13, "hyd_surface.npz", level="top")
    residuum_case()
    full_case(H.ATM, "atm")
    full_case(H.HYD, "hyd")
    msl_case()
    transfer_case()
    thermo_case()
    atm_slice_inputs()
    atm_slice_case()
