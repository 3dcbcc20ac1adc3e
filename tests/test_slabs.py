"""Phi-slab decomposition (atom_amd/slabs.py): a run split over N ranks must give, at every owned
column, the bits of the single-domain run.

CPU tests (gloo, 2 and 3 ranks): the slab logic over the oracle as the per-rank engine, against the
oracle on the whole domain.  GPU test: two ranks on one card through the C-ABI, same comparison
(hydrosphere and the pow-free atmosphere operators bit-exact; the atmosphere sweep to the stated
1e-10 per field because device and host pow differ in the last place)."""
import os
import subprocess
import sys

import numpy as np
import pytest

import harness as H
from atom_amd import slabs

HERE = os.path.dirname(os.path.abspath(__file__))
ATM_OPS = ["atm_bc_radius", "atm_bc_theta", "atm_bc_phi", "atm_value_limitation", "atm_solid_ground", "atm_rk3d",
           "atm_value_limitation", "atm_pressure3d", "move3d", "atm_bc_radius", "atm_bc_theta", "atm_bc_phi", "atm_rk3d",
           "atm_pressure3d"]
HYD_OPS = ["hyd_bc_radius", "hyd_bc_theta", "hyd_bc_phi", "hyd_value_limitation", "hyd_rk3d", "hyd_solid_ground",
           "hyd_pressure3d", "move3d", "hyd_bc_radius", "hyd_bc_theta", "hyd_bc_phi", "hyd_rk3d", "hyd_pressure3d"]


def _launch(world, engine, model, dims, seed, mask, ops, out, port):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), OMP_NUM_THREADS="1")
    procs = []
    for r in range(world):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "slab_worker.py"), engine, model,
                                       "x".join(map(str, dims)), str(seed), mask, ",".join(ops), out], env=e))
    rcs = [p.wait(timeout=600) for p in procs]
    assert rcs == [0] * world, rcs


def _single_domain(model, dims, seed, mask, ops):
    im, jm, km = dims
    m = H.ATM if model == "atm" else H.HYD
    p = H.default_params(m, im, jm, km)
    f, g = H.make_inputs(m, im, jm, km, seed, mask, p)
    names = H.ATM_FIELDS if m == H.ATM else H.HYD_FIELDS
    o = H.CpuModel("oracle", m, im, jm, km, p, {k: f[k] for k in names}, g)
    for op in ops:
        H.run_op(o, op)
    return o


def test_plan_layout():
    plan = slabs.SlabPlan(361, 4)
    assert plan.owned[0][0] == 0 and plan.owned[-1][1] == 361
    assert sum(e - s for s, e in plan.owned) == 361
    c0, c3 = plan.columns(0), plan.columns(3)
    assert list(c0[-slabs.S:]) == list(range(361 - slabs.S, 361)) and c0[0] == 0
    assert list(c3[:slabs.S]) == list(range(slabs.S)) and c3[-1] == 360
    for r in range(4):
        off, n = plan.owned_local(r)
        assert list(plan.columns(r)[off:off + n]) == list(range(*plan.owned[r]))
    for owner, holder, ks, kd, n in plan.transfers():
        assert owner != holder
        assert list(plan.columns(owner)[ks:ks + n]) == list(plan.columns(holder)[kd:kd + n])
    assert slabs.SlabPlan(361, 1).local_km(0) == 361 and slabs.SlabPlan(361, 1).transfers() == []
    with pytest.raises(ValueError):
        slabs.SlabPlan(20, 4)


@pytest.mark.parametrize("world,model,dims,seed,port", [(2, "atm", (7, 13, 40), 51, 29511), (3, "atm", (6, 11, 57), 52, 29512),
                                                        (2, "hyd", (7, 13, 40), 53, 29513), (3, "hyd", (6, 11, 57), 54, 29514)])
def test_slabs_over_oracle_bitexact(tmp_path, world, model, dims, seed, port):
    ops = ATM_OPS if model == "atm" else HYD_OPS
    out = str(tmp_path / "slabs.npz")
    _launch(world, "oracle", model, dims, seed, "blobs", ops, out, port)
    o = _single_domain(model, dims, seed, "blobs", ops)
    got = np.load(out)
    for k in got.files:
        assert H.bits_equal(got[k], o.f[k]), (k, H.max_rel_err(got[k], o.f[k]))
    o.close()


@pytest.mark.gpu
@pytest.mark.parametrize("model,dims,seed,port", [("hyd", (7, 13, 70), 55, 29521), ("atm", (7, 13, 70), 56, 29522)])
def test_slabs_two_ranks_one_gpu(tmp_path, model, dims, seed, port):
    ops = ATM_OPS if model == "atm" else HYD_OPS
    out = str(tmp_path / "slabs.npz")
    _launch(2, "device", model, dims, seed, "blobs", ops, out, port)
    o = _single_domain(model, dims, seed, "blobs", ops)
    got = np.load(out)
    for k in got.files:
        if model == "hyd":
            assert H.bits_equal(got[k], o.f[k]), (k, H.max_rel_err(got[k], o.f[k]))
        else:
            assert H.max_rel_err(got[k], o.f[k]) <= 1e-10, (k, H.max_rel_err(got[k], o.f[k]))
    o.close()
