"""The 0.1-degree grid with 81 radial levels (81 x 1801 x 3601: 4.2 GB a field, 185 GB an atmosphere) runs as phi-slabs
only, one per GPU.  Here: the slab that rank 0 of a 16-way split holds - 225 owned columns, 3 ghost columns of its
neighbour and the 5 seam copies of an edge slab, 233 columns in all, 272 MB a field (the widest slab whose oracle run,
one core, still takes well under a minute per sweep) - through the 3-D sweeps and pressure steps of both models against
the oracle: the 0 Ma masks of the 1-degree grid upsampled by 10 (by 2 in the radial direction), seeded fields, the time step scaled with dthe^2.  The
hydrosphere must carry the oracle's bits (81 levels: the level -> XCD maps, the two-blocks-per-workgroup form and the
hand-off of salinity rows between levels im-3 and im-2 at a grid none of the other tests reaches); the atmosphere is held
to 1e-10 per field, in its default form and in the pipelined one.  Reference: lib/Array.cpp:54-71 (the size asserts such a
grid needs lifted), RungeKutta_Atm.cpp:47-49, RungeKutta_Hyd.cpp:30-124, RHS_Hyd.cpp:355-373."""
import numpy as np
import pytest

import atom_amd
import harness as H
from atom_amd import slabs

pytestmark = pytest.mark.gpu

IM, JM, KM_GLOBAL, WORLD = 81, 1801, 3601, 16


def _slab_pair(model, seed):
    plan = slabs.SlabPlan(KM_GLOBAL, WORLD)
    cols = plan.columns(0)
    km = len(cols)
    assert km == 233 and plan.owned[0] == (0, 225)
    p = H.default_params(model, IM, JM, KM_GLOBAL)       # the steps of the whole grid
    # The reference's time step (1e-5) belongs to its 1-degree grid: with dthe 10 times smaller the coastal damping term
    # dt v / dthe^2 is 3.3 per stage, beyond the stability limit of the four-stage scheme, and the theta recurrence of a
    # sweep overflows.  The time step goes with dthe^2, as it would in any run on this grid.
    p.dt = 1e-5 / 100.
    f, g = H.make_inputs(model, IM, JM, km, seed, "allfluid", p)
    names = H.ATM_FIELDS if model == H.ATM else H.HYD_FIELDS
    f = {k: f[k] for k in names}
    H.shrink_noise(f, 10. ** 4)                          # (refinement)^4, as 4^4 on the 0.25-degree grid
    h, topo = H.load_mask_0Ma(model)
    ii = np.rint(np.arange(IM) / 2.).astype(int)
    jj = np.rint(np.arange(JM) / 10.).astype(int)
    kk = np.rint(cols / 10.).astype(int)
    f["H"], g["TOPOGRAPHY"] = np.ascontiguousarray(h[ii][:, jj][:, :, kk]), np.ascontiguousarray(topo[jj][:, kk])
    o = H.CpuModel("oracle", model, IM, JM, km, p, f, g)
    d = atom_amd.DeviceModel(model, IM, JM, km, atom_amd.Params.from_buffer_copy(bytes(p)))
    d.upload_all(f, g)
    return o, d, names


def test_hyd_slab_of_the_tenth_degree_grid_bitexact():
    o, d, names = _slab_pair(H.HYD, 501)
    start = {k: o.f[k].copy() for k in names}
    assert d.rk_variant() == 3
    d.set_rk_variant(0)                                         # the exact form: the oracle's bits
    for op in ("hyd_rk3d", "hyd_rk3d", "hyd_pressure3d"):       # two sweeps: both orientations of the buffer pair
        o.op(op); d.op(op)
        for k in names:
            if k != "RESIDUUM":
                assert np.isfinite(o.f[k]).all(), (op, k)
                got = d.download(k)
                assert H.bits_equal(got, o.f[k]), (op, k, H.max_rel_err(got, o.f[k]))
    d.upload_all(start)                                         # the default form (pipelined): within the tolerance
    d.set_rk_variant(3)
    for op in ("hyd_rk3d", "hyd_rk3d", "hyd_pressure3d"):
        d.op(op)
    worst = 0.0
    for k in names:
        if k != "RESIDUUM":
            e = H.max_rel_err(d.download(k), o.f[k])
            worst = max(worst, e)
            assert e <= 1e-10, (k, e)
    print("hydrosphere sweeps + pressure on a slab of the 0.1-degree grid, pipelined form: max rel err %.3e" % worst)
    d.close(); o.close()


def test_atm_slab_of_the_tenth_degree_grid():
    o, d, names = _slab_pair(H.ATM, 502)
    start = {k: o.f[k].copy() for k in names}
    for op in ("atm_rk3d", "atm_pressure3d"):
        o.op(op)
    worst = {}
    assert d.rk_variant() == 2
    for variant in (2, 3):      # the default form of the sweep and the pipelined one
        d.upload_all(start)
        d.set_rk_variant(variant)
        for op in ("atm_rk3d", "atm_pressure3d"):
            d.op(op)
        worst[variant] = 0.0
        for k in names:
            if k != "RESIDUUM":
                assert np.isfinite(o.f[k]).all(), k
                e = H.max_rel_err(d.download(k), o.f[k])
                worst[variant] = max(worst[variant], e)
                assert e <= 1e-10, (variant, k, e)
    print("atmosphere sweep + pressure on a slab of the 0.1-degree grid: max rel err by sweep variant %r" % worst)
    d.close(); o.close()
