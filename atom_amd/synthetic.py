"""Seeded synthetic inputs of the hot path: fields drawn inside the limiter bounds, synthetic land masks, the real 0 Ma
mask of the 1-degree grid.  Used by bench.py, the tools and the tests (tests/harness.py re-exports these names)."""
import os

import numpy as np

from .device import ATM, HYD

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

ATM_FIELDS = ["T", "U", "V", "W", "P_DYN", "C", "CLOUD", "ICE", "CO2",
              "TN", "UN", "VN", "WN", "P_DYNN", "CN", "CLOUDN", "ICEN", "CO2N",
              "H", "AUX_U", "AUX_V", "AUX_W", "BUOYANCY", "S_V", "S_C", "S_I", "S_R", "S_S", "RESIDUUM"]
HYD_FIELDS = ["T", "U", "V", "W", "P_DYN", "C", "TN", "UN", "VN", "WN", "P_DYNN", "CN",
              "H", "AUX_U", "AUX_V", "AUX_W", "BUOYANCY",
              "SALT_BALANCE", "SALT_FINGER", "SALT_DIFFUSION", "R_WATER", "R_SALT_WATER", "RESIDUUM", "P_STAT"]


def synthetic_mask(model, im, jm, km, L, rng, kind="blobs"):
    """Land mask h (exactly 0. / 1.) and the 2-D height field in metres.

    blobs: smooth random terrain thresholded per level plus sprinkled single-cell
    land and fluid pockets, so that every one-sided / coast branch is reached.
    allfluid: h = 0 everywhere.
    """
    h = np.zeros((im, jm, km))
    topo = np.zeros((jm, km))
    if kind == "allfluid":
        return h, topo
    step = L / (im - 1)
    # smooth terrain from a few random cosines
    jj, kk = np.meshgrid(np.arange(jm), np.arange(km), indexing="ij")
    z = np.zeros((jm, km))
    for _ in range(6):
        a, b = rng.uniform(0.5, 4.0, 2)
        ph1, ph2 = rng.uniform(0, 2 * np.pi, 2)
        z += rng.uniform(0.3, 1.0) * np.cos(a * 2 * np.pi * jj / jm + ph1) * np.cos(b * 2 * np.pi * kk / km + ph2)
    z = (z - z.min()) / (z.max() - z.min())
    nlev = max(2, im // 3)
    height_levels = np.where(z > 0.55, (z - 0.55) / 0.45 * nlev, 0.0)  # in levels
    topo = np.round(height_levels * step * rng.uniform(0.7, 1.3, (jm, km)), 1)
    topo[height_levels == 0] = 0.0
    for i in range(im):
        if model == ATM:
            h[i] = (height_levels > 0) & (i <= np.floor(height_levels))
        else:
            # ocean floor rises from below; continents fill the whole column
            h[i] = (height_levels > 0) & ((i <= np.floor(height_levels) + im // 3) | (z > 0.85))
    # sprinkle isolated cells
    n = max(4, h.size // 200)
    idx = (rng.integers(0, im, n), rng.integers(0, jm, n), rng.integers(0, km, n))
    h[idx] = 1.0 - h[idx]
    return np.ascontiguousarray(h, dtype=np.float64), np.ascontiguousarray(topo)


def make_inputs(model, im, jm, km, seed=1234, mask="blobs", params=None):
    """Fields drawn inside the limiter bounds (uniform draws; the source terms within 1e-6; the hydrosphere's two densities around 1000 and 1028)."""
    rng = np.random.default_rng(seed)
    shp = (im, jm, km)
    U = lambda lo, hi: rng.uniform(lo, hi, shp)
    f = {}
    if model == ATM:
        L = params.L_atm if params is not None else 16000.0
        f["T"], f["U"], f["V"], f["W"] = U(0.8, 1.1), U(-0.1, 0.1), U(-0.1, 0.1), U(-0.4, 0.4)
        f["P_DYN"], f["C"], f["CLOUD"], f["ICE"], f["CO2"] = U(-0.1, 0.1), U(0, 0.03), U(0, 0.01), U(0, 0.005), U(1.0, 1.2)
        names = ["T", "U", "V", "W", "P_DYN", "C", "CLOUD", "ICE", "CO2"]
        for s in ["S_V", "S_C", "S_I", "S_R", "S_S"]:
            f[s] = U(-1e-6, 1e-6)
    else:
        L = params.L_hyd if params is not None else 1000.0
        f["T"], f["U"], f["V"], f["W"] = U(0.98, 1.1), U(-0.1, 0.1), U(-0.1, 0.1), U(-0.1, 0.1)
        f["P_DYN"], f["C"] = U(-0.1, 0.1), U(0.95, 1.1)
        names = ["T", "U", "V", "W", "P_DYN", "C"]
        f["R_WATER"] = U(995.0, 1005.0)
        f["R_SALT_WATER"] = U(1020.0, 1035.0)
        f["P_STAT"] = np.random.default_rng(seed + 7919).uniform(1.0, 100.0, shp)  # own stream: the other draws stay as they were
        for s in ["SALT_BALANCE", "SALT_FINGER", "SALT_DIFFUSION"]:
            f[s] = np.zeros(shp)
    for nme in names:
        f[nme + "N"] = f[nme] + 1e-3 * U(-1, 1) * np.abs(f[nme]).max()
    for s in ["AUX_U", "AUX_V", "AUX_W"]:
        f[s] = U(-1, 1)
    f["BUOYANCY"] = np.zeros(shp)
    f["RESIDUUM"] = np.zeros(shp)
    h, topo = synthetic_mask(model, im, jm, km, L, rng, mask)
    f["H"] = h
    g = {"TOPOGRAPHY": topo,
         "EVAPORATION_DALTON": rng.uniform(0.0, 8.0, (jm, km)),
         "PRECIPITATION": rng.uniform(0.0, 3.0, (jm, km))}
    return f, g


def load_mask_0Ma(model):
    """The real 0 Ma land mask on the 1-degree grid, as the reference's readers
    build it from 0Ma_Golonka.xyz (a committed fixture under tests/golden)."""
    z = np.load(os.path.join(ROOT, "tests", "golden", "mask_0Ma.npz"))
    tag = "atm" if model == ATM else "hyd"
    n = 41 * 181 * 361
    h = np.unpackbits(z[tag + "_h_bits"])[:n].astype(np.float64).reshape(41, 181, 361)
    return np.ascontiguousarray(h), np.ascontiguousarray(z[tag + "_topo"], dtype=np.float64)


def shrink_noise(f, factor):
    """Every field of f (but the mask and the densities) pulled towards its mean: white noise of the 1-degree tests'
    amplitude leaves every bound within two sweeps on a refined grid (second differences in phi grow with 1 / dphi^2, divided
    by a sin^2(theta) that shrinks next to the poles)."""
    for k in f:
        if k not in ("H", "RESIDUUM", "R_WATER", "R_SALT_WATER", "P_STAT"):
            m = f[k].mean()
            f[k] = m + (f[k] - m) / factor
