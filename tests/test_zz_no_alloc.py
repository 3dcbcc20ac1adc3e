"""atomdev_atm_residuum3d_noalloc: the residuum query of Accuracy_Atm::residuumQuery_3D with its intermediate
results in the library's static device array instead of an allocated buffer.  Same kernels as
atomdev_atm_residuum3d, so the same bits: checked on two calls in a row, the second on a smaller grid (the
static array still holds the first call's sequence), with the carried-in state of the reference's reused
Accuracy_Atm object, and the size limit is refused with a message."""
import numpy as np
import pytest

import atom_amd
import harness as H

pytestmark = pytest.mark.gpu


def _model(dims, seed):
    im, jm, km = dims
    p = H.default_params(H.ATM, im, jm, km)
    f, g = H.make_inputs(H.ATM, im, jm, km, seed, "blobs", p)
    f = {k: f[k] for k in H.ATM_FIELDS}
    d = atom_amd.DeviceModel(H.ATM, im, jm, km, atom_amd.Params.from_buffer_copy(bytes(p)))
    d.upload_all(f, g)
    return d, f


def test_residuum3d_noalloc_matches_residuum3d():
    for dims, seed in (((23, 61, 97), 91), ((7, 13, 17), 92)):   # the second call works on a smaller sequence
        d, f = _model(dims, seed)
        mn_a, loc_a = d.residuum("atm_residuum3d")
        field_a = d.download("RESIDUUM")
        d.upload("RESIDUUM", np.zeros(dims))
        mn_b, loc_b = d.residuum("atm_residuum3d_noalloc")
        field_b = d.download("RESIDUUM")
        assert np.array([mn_a]).tobytes() == np.array([mn_b]).tobytes() and loc_a == loc_b, (dims, mn_a, mn_b, loc_a, loc_b)
        assert field_a.tobytes() == field_b.tobytes()
        assert np.abs(field_b).max() > 0
        # the reference reuses one Accuracy_Atm object: the second query starts from the first one's state
        d.upload("U", -0.5 * f["U"]); d.upload("V", 0.25 * f["V"]); d.upload("W", -0.125 * f["W"])
        again_a = d.residuum("atm_residuum3d", mn_a, loc_a)
        again_b = d.residuum("atm_residuum3d_noalloc", mn_b, loc_b)
        assert again_a == again_b
        # and against the oracle
        o = H.CpuModel("oracle", H.ATM, *dims, H.default_params(H.ATM, *dims), f, {})
        mn_o, loc_o = o.residuum("atm_residuum3d")
        assert (mn_o, loc_o) == (mn_b, loc_b)
        d.close()


def test_residuum3d_noalloc_refuses_grids_beyond_the_static_scratch():
    im, jm, km = 43, 261, 261     # 41 * 259 * 259 = 2 750 321 interior points > 2 621 440
    p = H.default_params(H.ATM, im, jm, km)
    d = atom_amd.DeviceModel(H.ATM, im, jm, km, atom_amd.Params.from_buffer_copy(bytes(p)))
    with pytest.raises(atom_amd.AtomDevError, match="static scratch holds at most"):
        d.residuum("atm_residuum3d_noalloc")
    d.residuum("atm_residuum3d")   # the allocating entry point still takes it
    d.close()
