"""The 3-D sweeps and pressure steps on the refined 0.25-degree grid (41 x 721 x 1441, 16 times the points of the
1-degree grid) against the oracle: the 0 Ma land masks of the 1-degree grid upsampled by 4, seeded fields.  The
hydrosphere in its exact form must carry the oracle's bits - 23 longitude blocks x 40 rounds = 920 workgroups, more than the chip
holds at once, so the hand-off of new salinity rows between the workgroups of levels im-3 and im-2 runs with workgroups
that start at different times; the atmosphere (pow per point) is held to 1e-10 per field."""
import numpy as np
import pytest

import atom_amd
import harness as H

pytestmark = pytest.mark.gpu

DIMS = (41, 721, 1441)


def _upsampled_mask(model):
    h, topo = H.load_mask_0Ma(model)
    jj = np.rint(np.arange(DIMS[1]) / 4.).astype(int)
    kk = np.rint(np.arange(DIMS[2]) / 4.).astype(int)
    return np.ascontiguousarray(h[:, jj][:, :, kk]), np.ascontiguousarray(topo[jj][:, kk])


def _pair(model, seed, variants):
    """The oracle and one device model per sweep variant on the same inputs (the oracle's sweeps take most of the time
    here: they are run once for all variants)."""
    p = H.default_params(model, *DIMS)
    f, g = H.make_inputs(model, *DIMS, seed, "allfluid", p)
    names = H.ATM_FIELDS if model == H.ATM else H.HYD_FIELDS
    f = {k: f[k] for k in names}
    # white noise of the 1-degree tests' amplitude has 16 times their second differences in phi here, divided by a
    # sin^2(theta) that is 16 times smaller on the rows next to the poles, and leaves every bound within two sweeps: the same
    # noise at 1/256 of the amplitude around each field's mean
    for k in names:
        if k not in ("H", "RESIDUUM", "R_WATER", "R_SALT_WATER", "P_STAT"):
            m = f[k].mean()
            f[k] = m + (f[k] - m) / 256.
    f["H"], g["TOPOGRAPHY"] = _upsampled_mask(model)
    o = H.CpuModel("oracle", model, *DIMS, p, f, g)
    devices = []
    for v in variants:
        d = atom_amd.DeviceModel(model, *DIMS, atom_amd.Params.from_buffer_copy(bytes(p)))
        d.upload_all(f, g)
        if v is None:
            assert d.rk_variant() == (2 if model == H.ATM else 3)     # the handle's default
        else:
            d.set_rk_variant(v)
        devices.append(d)
    return o, devices, names


def test_hyd_sweeps_and_pressure_quarter_degree():
    """Two sweeps (both orientations of the buffer pair) and a pressure step: the oracle's bits with the sweep in its
    exact form, every field within 1e-10 in the handle's default form (the pipelined one)."""
    o, (exact, default), names = _pair(H.HYD, 301, (0, None))
    worst = 0.0
    for op in ("hyd_rk3d", "hyd_rk3d", "hyd_pressure3d"):
        o.op(op); exact.op(op); default.op(op)
        for k in names:
            if k != "RESIDUUM":
                assert np.isfinite(o.f[k]).all(), (op, k)
                got = exact.download(k)
                assert H.bits_equal(got, o.f[k]), (op, k, H.max_rel_err(got, o.f[k]))
                e = H.max_rel_err(default.download(k), o.f[k])
                worst = max(worst, e)
                assert e <= 1e-10, (op, k, e)
    print("hydrosphere sweeps + pressure at 0.25 degrees, pipelined form: max rel err %.3e" % worst)
    exact.close(); default.close()


def test_atm_sweep_and_pressure_quarter_degree():
    """A sweep and a pressure step with the sweep in its pipelined (3), wave-specialised tolerance (2, the default) and
    exact (0) forms: every field within 1e-10 of the oracle."""
    variants = (3, 2, 0)
    o, devices, names = _pair(H.ATM, 302, variants)
    worst = dict.fromkeys(variants, 0.0)
    for op in ("atm_rk3d", "atm_pressure3d"):
        o.op(op)
        for v, d in zip(variants, devices):
            d.op(op)
            for k in names:
                if k != "RESIDUUM":
                    e = H.max_rel_err(d.download(k), o.f[k])
                    worst[v] = max(worst[v], e)
                    assert e <= 1e-10, (v, op, k, e)
    print("atmosphere sweep + pressure at 0.25 degrees: max rel err by sweep variant %r" % worst)
    for d in devices:
        d.close()
