"""Phi-slab decomposition (atom_amd/slabs.py): a run split over N ranks must give, at every owned
column, the bits of the single-domain run, and the queries the single domain's results and locations.

CPU tests (gloo, 2 and 3 ranks): the slab logic over the oracle as the per-rank engine, against the
oracle on the whole domain - operator sequences, and the whole 3-D loop of the flow solver with its queries
after `distribute` from a full-width model on rank 0.  GPU tests: two ranks on one card through the C-ABI, host-staged
over gloo and with device-resident buffers over RCCL, same comparisons (hydrosphere and the pow-free atmosphere
operators bit-exact; the atmosphere sweep to the stated 1e-10 per field because device and host pow differ in the
last place), and the loop on the 0.25-degree grid."""
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


def _launch(world, engine, model, dims, seed, mask, ops, out, port, timeout=600):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), OMP_NUM_THREADS="1")
    if model == "hyd":   # the slabs must reproduce the single domain bit for bit: the hydrosphere's sweep in its exact form
        env["ATOMDEV_RK_VARIANT"] = "0"
    procs = []
    for r in range(world):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "slab_worker.py"), engine, model,
                                       "x".join(map(str, dims)), str(seed), mask, ops if isinstance(ops, str) else ",".join(ops), out],
                                      env=e))
    rcs = [p.wait(timeout=timeout) for p in procs]
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


def _single_domain_loop(model, dims, seed, counts=(2, 2), refined=False):
    im, jm, km = dims
    m = H.ATM if model == "atm" else H.HYD
    p = H.default_params(m, im, jm, km)
    if m == H.ATM:
        f, g = H.make_msl_inputs(im, jm, km, seed, p)
    else:
        f, g = H.make_inputs(m, im, jm, km, seed, "blobs", p)
        f = {k: f[k] for k in H.HYD_FIELDS}
    if refined:
        H.shrink_noise(f, 256.)
    o = H.CpuModel("oracle", m, im, jm, km, p, f, g)
    log = H.atm_3d_loop(o, *counts) if m == H.ATM else H.hyd_3d_loop(o, *counts)
    return o, log


def _same_log(got, want, exact):
    """The slab run's query log against the single domain's: residuum (value, location) and steady (values, locations)."""
    assert len(got) == len(want)
    for a, b in zip(got, want):
        if isinstance(b[0], np.ndarray) or isinstance(a[0], list):   # steady
            va, la = np.array(a[0]), np.array(a[1]).reshape(-1, 3)
            vb, lb = np.array(b[0]), np.array(b[1]).reshape(-1, 3)
            if exact:
                assert np.array_equal(va, vb) and np.array_equal(la, lb), (a, b)
            else:
                assert np.allclose(va, vb, rtol=1e-9, atol=0)
        else:
            if exact:
                assert a[0] == b[0] and tuple(a[1]) == tuple(b[1]), (a, b)
            else:
                assert abs(a[0] - b[0]) <= 1e-9 * abs(b[0])


@pytest.mark.parametrize("world,model,dims,seed,port", [(2, "atm", (7, 13, 40), 61, 29531), (3, "atm", (6, 11, 57), 62, 29532),
                                                        (2, "hyd", (7, 13, 40), 63, 29533), (3, "hyd", (6, 11, 57), 64, 29534)])
def test_slab_loop_over_oracle_bitexact(tmp_path, world, model, dims, seed, port):
    """distribute -> 2 x 2 iterations of the 3-D loop with the residuum and steady queries -> collect: fields, query values
    and query locations of the single domain."""
    import pickle
    out = str(tmp_path / "slabs.npz")
    _launch(world, "oracle", model, dims, seed, "blobs", "loop", out, port)
    o, log = _single_domain_loop(model, dims, seed)
    got = np.load(out)
    for k in got.files:
        assert H.bits_equal(got[k], o.f[k]), (k, H.max_rel_err(got[k], o.f[k]))
    _same_log(pickle.load(open(out + ".log", "rb")), log, exact=True)
    o.close()


@pytest.mark.gpu
@pytest.mark.parametrize("engine,model,dims,seed,port", [("device", "hyd", (7, 13, 70), 65, 29541), ("device", "atm", (9, 21, 70), 67, 29543)])
def test_slab_loop_two_ranks_one_gpu(tmp_path, engine, model, dims, seed, port):
    """The same loop with two device slabs (two processes) on one card, host-staged over gloo.  RCCL refuses two ranks on
    one device ("Duplicate GPU detected"), so the device-resident exchange is exercised in one process below."""
    import pickle
    out = str(tmp_path / "slabs.npz")
    _launch(2, engine, model, dims, seed, "blobs", "loop", out, port)
    o, log = _single_domain_loop(model, dims, seed)
    got = np.load(out)
    for k in got.files:
        if model == "hyd":
            assert H.bits_equal(got[k], o.f[k]), (k, H.max_rel_err(got[k], o.f[k]))
        else:
            assert H.max_rel_err(got[k], o.f[k]) <= 1e-10, (k, H.max_rel_err(got[k], o.f[k]))
    _same_log(pickle.load(open(out + ".log", "rb")), log, exact=model == "hyd")
    o.close()


@pytest.mark.gpu
def test_slab_loop_quarter_degree_two_ranks_one_gpu(tmp_path):
    """One iteration of the hydrosphere's 3-D loop (sweep, queries, pressure step) on the 0.25-degree grid split over two
    device slabs: the single domain's bits, query values and locations."""
    import pickle
    dims = (41, 721, 1441)
    out = str(tmp_path / "slabs.npz")
    _launch(2, "device", "hyd", dims, 68, "blobs", "loop:1x1:refined", out, 29551, timeout=1500)
    o, log = _single_domain_loop("hyd", dims, 68, (1, 1), refined=True)
    got = np.load(out)
    for k in got.files:
        assert H.bits_equal(got[k], o.f[k]), (k, H.max_rel_err(got[k], o.f[k]))
    _same_log(pickle.load(open(out + ".log", "rb")), log, exact=True)
    o.close()


class _ThreadDist:
    """torch.distributed for ranks that are threads of one process, with the stream behaviour of the RCCL backend: a send
    reads its buffer on a communication stream that is ordered against torch's current stream only; wait() of a receive
    orders torch's current stream against the arrival and blocks neither the host nor any other stream."""

    class P2POp:
        def __init__(self, op, tensor, peer):
            self.op, self.tensor, self.peer = op, tensor, peer

    class _Work:
        def __init__(self, fn):
            self.fn = fn

        def wait(self):
            self.fn()

    isend, irecv = "isend", "irecv"

    def __init__(self, shared, rank):
        self.s, self.rank = shared, rank

    @staticmethod
    def shared(world):
        import collections
        import threading
        import torch
        return dict(world=world, cv=threading.Condition(), mail={}, sent=collections.Counter(), posted=collections.Counter(),
                    barrier=threading.Barrier(world), box={}, side=[torch.cuda.Stream() for _ in range(world)])

    def get_backend(self):
        return "nccl"

    def batch_isend_irecv(self, ops):
        import torch
        side = self.s["side"][self.rank]
        works = []
        for o in ops:
            if o.op == "isend":
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    staged = o.tensor.clone()
                    ev = torch.cuda.Event(); ev.record(side)
                with self.s["cv"]:   # messages between two ranks match in the order they were posted
                    n = self.s["sent"][(self.rank, o.peer)]
                    self.s["sent"][(self.rank, o.peer)] += 1
                    self.s["mail"][(self.rank, o.peer, n)] = (staged, ev)
                    self.s["cv"].notify_all()
                works.append(self._Work(lambda: None))
            else:
                with self.s["cv"]:
                    n = self.s["posted"][(o.peer, self.rank)]
                    self.s["posted"][(o.peer, self.rank)] += 1

                def arrive(o=o, n=n):
                    with self.s["cv"]:
                        assert self.s["cv"].wait_for(lambda: (o.peer, self.rank, n) in self.s["mail"], timeout=120)
                        staged, ev = self.s["mail"].pop((o.peer, self.rank, n))
                    with torch.cuda.stream(side):
                        side.wait_event(ev)
                        o.tensor.copy_(staged, non_blocking=True)
                        done = torch.cuda.Event(); done.record(side)
                    torch.cuda.current_stream().wait_event(done)
                works.append(self._Work(arrive))
        return works

    def _exchange(self, key, value):
        self.s["box"][(key, self.rank)] = value
        self.s["barrier"].wait()
        out = [self.s["box"][(key, r)] for r in range(self.s["world"])]
        self.s["barrier"].wait()
        return out

    def all_gather_object(self, box, obj):
        box[:] = self._exchange("ag", obj)

    def all_gather(self, box, tensor):
        tensor.device.type == "cuda" and __import__("torch").cuda.current_stream().synchronize()
        for b, t in zip(box, self._exchange("agt", tensor.clone())):
            b.copy_(t)

    def scatter_object_list(self, box, parts, src=0):
        box[0] = self._exchange("sc", parts)[src][self.rank]

    def gather_object(self, obj, gathered, dst=0):
        got = self._exchange("ga", obj)
        if self.rank == dst:
            gathered[:] = got


@pytest.mark.gpu
@pytest.mark.parametrize("model,dims,seed", [("hyd", (7, 13, 70), 69), ("atm", (9, 21, 70), 70)])
def test_slab_loop_device_resident_exchange(model, dims, seed):
    """Two slabs as two threads of one process, exchange buffers in device memory (xdev = cuda), the messages moved by
    device-to-device copies on communication streams with the ordering rules of the RCCL backend: packing on the handle's
    stream, sending on another, unpacking on the handle's stream again.  Fields and queries of the single domain."""
    import threading
    import torch
    import atom_amd
    im, jm, km = dims
    m = H.ATM if model == "atm" else H.HYD
    p = H.default_params(m, im, jm, km)
    if m == H.ATM:
        f, g = H.make_msl_inputs(im, jm, km, seed, p)
    else:
        f, g = H.make_inputs(m, im, jm, km, seed, "blobs", p)
        f = {k: f[k] for k in H.HYD_FIELDS}
    shared = _ThreadDist.shared(2)
    full = atom_amd.DeviceModel(m, im, jm, km, atom_amd.Params.from_buffer_copy(bytes(p)), device=0)
    full.upload_all(f, g)
    results, errors = [None, None], []

    def rank_main(rank):
        try:
            torch.cuda.set_device(0)
            sm = slabs.create_device_slab(m, im, jm, km, atom_amd.Params.from_buffer_copy(bytes(p)), rank, 2, _ThreadDist(shared, rank), device=-1)
            assert sm.xdev == "cuda:0"
            if model == "hyd":
                sm.e.set_rk_variant(0)   # the exact form: the comparison below is bit for bit
            sm.distribute(full if rank == 0 else None, list(f), list(g))
            log = sm.run_3d_loop(2, 2, queries=True)
            results[rank] = (sm.collect([k for k in f if k not in ("H", "RESIDUUM")]), log)
            sm.e.close()
        except BaseException as exc:   # a failing rank must not leave the other one waiting
            errors.append(exc)
            shared["barrier"].abort()

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(2)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=300)
    assert not errors, errors
    o, log = _single_domain_loop(model, dims, seed)
    got, got_log = results[0]
    for k in got:
        if model == "hyd":
            assert H.bits_equal(got[k], o.f[k]), (k, H.max_rel_err(got[k], o.f[k]))
        else:
            assert H.max_rel_err(got[k], o.f[k]) <= 1e-10, (k, H.max_rel_err(got[k], o.f[k]))
    _same_log(got_log, log, exact=model == "hyd")
    assert results[1][1] == got_log         # every rank sees the same query results
    full.close(); o.close()


@pytest.mark.gpu
@pytest.mark.parametrize("world,model,dims,seed", [(2, "hyd", (7, 13, 70), 71), (3, "hyd", (6, 11, 80), 72), (2, "atm", (9, 21, 70), 73),
                                                   (3, "atm", (7, 13, 80), 74)])
def test_slab_group_in_one_process(world, model, dims, seed):
    """The slabs as handles of ONE process, exchange and queries by the library (atomdev_slabs_*: a second stream per
    handle, events, device-to-device copies between persistent buffers, no host wait): 2 x 2 iterations of the 3-D loop,
    fields and queries of the single domain."""
    import atom_amd
    im, jm, km = dims
    m = H.ATM if model == "atm" else H.HYD
    p = H.default_params(m, im, jm, km)
    if m == H.ATM:
        f, g = H.make_msl_inputs(im, jm, km, seed, p)
    else:
        f, g = H.make_inputs(m, im, jm, km, seed, "blobs", p)
        f = {k: f[k] for k in H.HYD_FIELDS}
    ls = slabs.LocalSlabs(m, im, jm, km, atom_amd.Params.from_buffer_copy(bytes(p)), world, devices=[0] * world)
    plan = slabs.SlabPlan(km, world)
    for r in range(world):   # the library's layout is the plan of atom_amd/slabs.py
        assert list(ls.cols[r]) == list(plan.columns(r)) and ls.owned[r] == plan.owned_local(r)
    ls.upload_all(f, g)
    if model == "hyd":
        for e in ls.e:
            e.set_rk_variant(0)   # the exact form: the comparison below is bit for bit
    got_log = ls.run_3d_loop(2, 2, queries=True)
    o, log = _single_domain_loop(model, dims, seed)
    for k in f:
        if k in ("H", "RESIDUUM"):
            continue
        got = ls.collect(k)
        if model == "hyd":
            assert H.bits_equal(got, o.f[k]), (k, H.max_rel_err(got, o.f[k]))
        else:
            assert H.max_rel_err(got, o.f[k]) <= 1e-10, (k, H.max_rel_err(got, o.f[k]))
    _same_log(got_log, log, exact=model == "hyd")
    ls.close(); o.close()


@pytest.mark.gpu
def test_slab_group_refuses_what_it_cannot_do():
    import atom_amd
    p = H.default_params(H.HYD, 5, 7, 40)
    a = atom_amd.DeviceModel(H.HYD, 5, 7, 28, atom_amd.Params.from_buffer_copy(bytes(p)))
    lib = a.lib
    import ctypes as C
    assert lib.atomdev_slabs_local_km(40, 2, 0) == 28 and lib.atomdev_slabs_local_km(40, 2, 1) == 28
    assert lib.atomdev_slabs_local_km(20, 4, 0) == -1            # slabs narrower than ghosts + seam copies
    handles, g = (C.c_void_p * 2)(a._h, a._h), C.c_void_p()
    b = atom_amd.DeviceModel(H.HYD, 5, 7, 30, atom_amd.Params.from_buffer_copy(bytes(p)))
    handles[1] = b._h
    assert lib.atomdev_slabs_create(handles, 2, 40, C.byref(g)) != 0 and b"must have km = 28" in lib.atomdev_last_error()
    a.close(); b.close()
