"""One rank of a phi-slab run (started by tests/test_slabs.py, one process per rank, rendezvous on 127.0.0.1).
engine = "oracle": the slab logic of atom_amd.slabs driven over the CPU oracle (checks partition, ghost / seam layout,
exchange schedule and the sharded queries without a GPU; gloo).  engine = "device": the product path, every rank a
DeviceModel on cuda:0 with host-staged exchange (gloo).  engine = "device-nccl": the same with device-resident exchange
buffers over RCCL (the ranks share one card).
mode = a comma-separated operator list: the operators one by one on scattered inputs.  mode = "loop": rank 0 holds the
whole grid in a full-width model, `distribute` hands out the columns, every rank runs the 3-D loop of the flow solver with
its queries, `collect` brings the owned columns back.  Rank 0 writes fields and query log to <out>.npz."""
import os
import pickle
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


class OracleEngine:
    """The oracle behind the engine interface SlabModel expects."""

    def __init__(self, H, model, im, jm, km, p, f, g):
        self.m = H.CpuModel("oracle", model, im, jm, km, p, f, g)
        self.H, self.model, self.im, self.jm, self.km = H, model, im, jm, km

    def op(self, name):
        self.H.run_op(self.m, name)

    def msl_data(self, radiation_model):
        import ctypes as C
        self.m.op("atm_msl_data", C.c_int(radiation_model))

    def move_data_to_new_arrays(self, level=-1):
        self.H.run_op(self.m, "move3d")

    def _field(self, name):
        return self.m.f[name] if name in self.m.f else self.m.g[name]

    def upload(self, name, a):
        self._field(name)[...] = a

    def download(self, name):
        return self._field(name).copy()

    def sync(self):
        pass

    def get_columns(self, name, k0, nk, out=None):
        blk = self.m.f[name][:, :, k0:k0 + nk]
        if out is None:
            return np.ascontiguousarray(blk)
        out[...] = blk
        return out

    def set_columns(self, name, k0, block):
        self.m.f[name][:, :, k0:k0 + block.shape[-1]] = block

    # the sharded queries, in numpy (the device has kernels for them)
    def residuum_field(self):
        self.m.residuum("atm_residuum3d" if self.model == self.H.ATM else "hyd_residuum3d")

    def residuum_suffix(self, k0, nk, kg0, kmg, start):
        res = self.m.f["RESIDUUM"][1:-1, 1:-1, k0:k0 + nk]
        kg = np.arange(kg0, kg0 + nk)
        row = np.arange(res.shape[0] * res.shape[1], dtype=np.float64).reshape(res.shape[0], res.shape[1], 1)
        t = np.broadcast_to(row * (kmg - 2) + (kg - 1), res.shape)
        ok = np.broadcast_to((kg >= 1) & (kg <= kmg - 2), res.shape) & (t >= start)
        x, t = res[ok], t[ok]
        out = [-1., -1., 0., -1., 0., -1., 0.]
        num = ~np.isnan(x)
        if num.any():
            a = np.abs(x[num]); tn = t[num]; xn = x[num]
            top = a == a.max()
            n = np.argmin(np.where(top, tn, np.inf))
            out[0], out[1], out[2] = float(a.max()), float(tn[n]), float(xn[n])
            n = np.argmax(tn)
            out[5], out[6] = float(tn[n]), float(xn[n])
        nn = x >= 0
        if nn.any():
            n = np.argmin(np.where(nn, t, np.inf))
            out[3], out[4] = float(t[n]), float(x[n])
        return out

    def steady_part(self, k0, nk, kg0, kmg):
        names = ["U", "V", "W", "T", "C", "CLOUD", "ICE", "CO2", "P_DYN"] if self.model == self.H.ATM else ["U", "V", "W", "T", "C", "P_DYN"]
        last = self.model != self.H.ATM
        rows = np.arange(self.im * self.jm, dtype=np.float64).reshape(self.im, self.jm, 1)
        pos = rows * kmg + np.arange(kg0, kg0 + nk)
        v, p = [], []
        for nme in names:
            d = np.abs(self.m.f[nme][:, :, k0:k0 + nk] - self.m.f[nme + "N"][:, :, k0:k0 + nk])
            top = d == d.max()
            v.append(float(d.max()))
            p.append(float(pos[top].max() if last else pos[top].min()))
        return v, p


def main():
    engine, model_s, dims_s, seed, mask, mode, out = sys.argv[1:8]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    import torch
    import torch.distributed as dist
    nccl = engine == "device-nccl"
    if nccl:
        torch.cuda.set_device(0)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group(backend="gloo")
    import harness as H
    from atom_amd import slabs

    loop_counts, refined = (2, 2), False
    if mode.startswith("loop:"):      # loop:<pressure iterations>x<velocity iterations>[:refined]
        parts = mode.split(":")
        loop_counts = tuple(int(x) for x in parts[1].split("x"))
        refined = len(parts) > 2 and parts[2] == "refined"
        mode = "loop"
    model = H.ATM if model_s == "atm" else H.HYD
    im, jm, km = (int(x) for x in dims_s.split("x"))
    p = H.default_params(model, im, jm, km)
    if model == H.ATM and mode == "loop":
        f, g = H.make_msl_inputs(im, jm, km, int(seed), p)      # the loop runs run_MSL_data
    else:
        f, g = H.make_inputs(model, im, jm, km, int(seed), mask, p)  # the same global inputs on every rank
        names = H.ATM_FIELDS if model == H.ATM else H.HYD_FIELDS
        f = {k: f[k] for k in names}
    if refined:
        H.shrink_noise(f, 256.)
    names = list(f)
    plan = slabs.SlabPlan(km, world)
    kml = plan.local_km(rank)
    zeros3 = {k: np.zeros((im, jm, kml)) for k in f}
    zeros2 = {k: np.zeros((jm, kml)) for k in g}
    if engine == "oracle":
        sm = slabs.SlabModel(None, plan, rank, dist, "cpu", model)
        lf, lg = sm.scatter(f, g) if mode != "loop" else (zeros3, zeros2)
        sm.e = OracleEngine(H, model, im, jm, kml, p, lf, lg)
    else:
        import atom_amd
        sm = slabs.create_device_slab(model, im, jm, km, atom_amd.Params.from_buffer_copy(bytes(p)), rank, world, dist,
                                      device=0, xdev=None if nccl else "cpu")
        assert sm.xdev == ("cuda:0" if nccl else "cpu")
        if mode != "loop":
            lf, lg = sm.scatter(f, g)
            sm.e.upload_all(lf, lg)
    log = None
    if mode == "loop":
        full = None
        if rank == 0:   # the whole grid on rank 0, as the start of a time slice leaves it
            if engine == "oracle":
                full = OracleEngine(H, model, im, jm, km, p, f, g)
            else:
                import atom_amd
                full = atom_amd.DeviceModel(model, im, jm, km, atom_amd.Params.from_buffer_copy(bytes(p)), device=0)
                full.upload_all(f, g)
        sm.distribute(full, list(f), list(g))
        log = sm.run_3d_loop(loop_counts[0], loop_counts[1], queries=True)
    else:
        for op in mode.split(","):
            if op == "move3d":
                sm.move_data_to_new_arrays(-1)
            else:
                sm.op(op)
    res = sm.collect([k for k in names if k not in ("H", "RESIDUUM")])
    if rank == 0:
        np.savez(out, **res)
        with open(out + ".log", "wb") as fh:
            pickle.dump(log, fh)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
