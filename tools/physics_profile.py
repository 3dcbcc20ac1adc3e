"""The BC_Thermo column physics on the 1-degree grid, a few calls each, for rocprofv3 --kernel-trace --stats."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import atom_amd
import harness as H
import test_loops as TL

im, jm, km = 41, 181, 361
p, f, g, i_trop = TL.physics_loop_inputs(im, jm, km, 75)
i_trop[:] = 30
d = atom_amd.DeviceModel(H.ATM, im, jm, km, atom_amd.Params.from_buffer_copy(bytes(p)))
d.upload_all(f, g)
state = ("T", "C", "CLOUD", "ICE", "CN", "CLOUDN", "ICEN", "P_RAIN", "P_SNOW", "S_C_C")
for rep in range(4):
    for k in state:
        d.upload(k, f[k])
    d.ice_water_saturation_adjustment(0.0)
    d.radiation_multi_layer(i_trop)
    d.op("atm_latent_heat")
    d.two_category_ice_scheme((0.0, 0.0))
d.close()
print("done")
