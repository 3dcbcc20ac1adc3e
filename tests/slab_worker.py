"""One rank of a phi-slab run (started by tests/test_slabs.py, one process per rank, gloo rendezvous on
127.0.0.1).  engine = "oracle": the slab logic of atom_amd.slabs driven over the CPU oracle (checks
partition, ghost / seam layout and refresh schedule without a GPU).  engine = "device": the product
path, every rank a DeviceModel on cuda:0 (host-staged exchange, since the ranks share one card).
Rank 0 gathers the owned columns of every field and writes them to <out>.npz."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


class OracleEngine:
    """The oracle behind the engine interface SlabModel expects (op, move, get/set columns)."""

    def __init__(self, H, model, im, jm, km, p, f, g):
        self.m = H.CpuModel("oracle", model, im, jm, km, p, f, g)
        self.H, self.im, self.jm, self.km = H, im, jm, km

    def op(self, name):
        self.H.run_op(self.m, name)

    def move_data_to_new_arrays(self, level=-1):
        self.H.run_op(self.m, "move3d")

    def get_columns(self, name, k0, nk, out=None):
        blk = self.m.f[name][:, :, k0:k0 + nk]
        if out is None:
            return np.ascontiguousarray(blk)
        out[...] = blk
        return out

    def set_columns(self, name, k0, block):
        self.m.f[name][:, :, k0:k0 + block.shape[-1]] = block


def main():
    engine, model_s, dims_s, seed, mask, seq_s, out = sys.argv[1:8]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    import torch.distributed as dist
    dist.init_process_group(backend="gloo")
    import harness as H
    from atom_amd import slabs

    model = H.ATM if model_s == "atm" else H.HYD
    im, jm, km = (int(x) for x in dims_s.split("x"))
    p = H.default_params(model, im, jm, km)
    f, g = H.make_inputs(model, im, jm, km, int(seed), mask, p)  # the same global inputs on every rank
    names = H.ATM_FIELDS if model == H.ATM else H.HYD_FIELDS
    f = {k: f[k] for k in names}
    plan = slabs.SlabPlan(km, world)
    kml = plan.local_km(rank)
    if engine == "oracle":
        sm = slabs.SlabModel(None, plan, rank, dist, "cpu")
        lf, lg = sm.scatter(f, g)
        sm.e = OracleEngine(H, model, im, jm, kml, p, lf, lg)
    else:
        import atom_amd
        sm = slabs.create_device_slab(model, im, jm, km, atom_amd.Params.from_buffer_copy(bytes(p)), rank, world, dist,
                                      device=0, xdev="cpu")
        lf, lg = sm.scatter(f, g)
        sm.e.upload_all(lf, lg)
    for op in seq_s.split(","):
        if op == "move3d":
            sm.move_data_to_new_arrays(-1)
        else:
            sm.op(op)
    mine = {}
    for k in names:
        if k in ("H", "RESIDUUM"):
            continue
        g0, blk = sm.owned_block(k)
        mine[k] = (g0, np.ascontiguousarray(blk))
    gathered = [None] * world if rank == 0 else None
    dist.gather_object(mine, gathered, dst=0)
    if rank == 0:
        res = {}
        for k in mine:
            a = np.empty((im, jm, km))
            for part in gathered:
                g0, blk = part[k]
                a[:, :, g0:g0 + blk.shape[-1]] = blk
            res[k] = a
        np.savez(out, **res)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
