import os, sys
ROOT = "/root/repo" if os.path.exists("/root/repo/tests") else os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import atom_amd, harness as H
im, jm, km = 41, 181, 361
p = H.default_params(H.ATM, im, jm, km)
f, g = H.make_inputs(H.ATM, im, jm, km, 1234, "allfluid", p)
f = {k: f[k] for k in H.ATM_FIELDS}
d = atom_amd.DeviceModel(H.ATM, im, jm, km, atom_amd.Params.from_buffer_copy(bytes(p)))
d.upload_all(f, g)
for ops in (["atm_bc_radius_theta_phi"], ["atm_bc_radius", "atm_bc_theta", "atm_bc_phi"]):
    for _ in range(3):
        for o in ops: d.op(o)
    d.timer_start()
    for _ in range(50):
        for o in ops: d.op(o)
    print(ops, d.timer_stop() / 50, "ms")
d.close()
