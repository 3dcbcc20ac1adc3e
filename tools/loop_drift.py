"""Where the atmosphere loop's device/oracle difference comes from: the loop of tests/test_loops.py run in lock step
on both, the largest per-field relative difference printed after every operator."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import ctypes as C
import numpy as np
import atom_amd
import harness as H
import test_loops as TL


class Pair:
    """Forwards every call to the device and the oracle model and reports the difference afterwards."""
    def __init__(self, d, o, names3, names2):
        self.d, self.o, self.n3, self.n2 = d, o, names3, names2

    def report(self, what):
        errs = {}
        for k in self.n3 + self.n2:
            if k == "RESIDUUM":
                continue
            ref = self.o.f[k] if k in self.n3 else self.o.g[k]
            errs[k] = H.max_rel_err(self.d.download(k), ref)
        top = sorted(errs.items(), key=lambda kv: -kv[1])[:3]
        print("%-34s %s" % (what, "  ".join("%s %.1e" % kv for kv in top)), flush=True)


def main():
    im, jm, km = 12, 21, 34
    p, f, g, i_trop = TL.physics_loop_inputs(im, jm, km, 75)
    o = H.CpuModel("oracle", H.ATM, im, jm, km, p, f, g)
    d = atom_amd.DeviceModel(H.ATM, im, jm, km, atom_amd.Params.from_buffer_copy(bytes(p)))
    d.upload_all(f, g)
    pair = Pair(d, o, list(f), list(g))
    phys_d, phys_o = H._atm_physics(d), H._atm_physics(o)
    qd = qo = 0.0
    sd = so = (0.0, 0.0)
    H._move3d(d); H._move3d(o)
    for pit in range(2):
        for vit in (1, 2):
            for op in ("atm_bc_radius", "atm_bc_theta", "atm_bc_phi"):
                d.op(op); o.op(op)
            if vit % 2 == 0:
                qd, qo = phys_d[0](qd), phys_o[0](qo); pair.report("saturation adjustment")
            for op in ("atm_value_limitation", "atm_solid_ground", "atm_rk3d", "atm_value_limitation"):
                d.op(op); o.op(op)
                if op == "atm_rk3d":
                    pair.report("rk3d")
            phys_d[2](i_trop); phys_o[2](i_trop); pair.report("radiation")
            d.op("atm_latent_heat"); o.op("atm_latent_heat")
            phys_d[4](phys_d[3]()); phys_o[4](phys_o[3]())
            d.msl_data(1); o.op("atm_msl_data", C.c_int(1)); pair.report("latent heat, vegetation, msl")
            if vit % 2 == 0:
                sd, so = phys_d[1](sd), phys_o[1](so); pair.report("ice scheme")
            H._move3d(d); H._move3d(o)
        d.op("atm_pressure3d"); o.op("atm_pressure3d"); pair.report("pressure")


if __name__ == "__main__":
    main()
