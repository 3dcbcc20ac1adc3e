"""The reference's call sites (cAtmosphereModel.cpp:777-874, cHydrosphereModel.cpp:437-610) compiled
twice from one function template: with the reference's own operator classes and with the same-named
classes of include/atomdev_mirror.hpp (GPU through the C-ABI), on the same seeded jagged Arrays.

The binary (oracle/_ref/bridge_test) contains compiled reference sources, so it is built where
/root/reference exists (`make -C oracle bridge`, done by __graft_entry__.build()) and travels with
the tree; without it the test is skipped.  Bars are printed by the binary: hydrosphere and the
pow-free atmosphere operators bit-exact, the atmosphere 3-D sweep 1e-10 relative per field."""
import os
import subprocess

import pytest

import harness as H

BIN = os.path.join(H.ROOT, "oracle", "_ref", "bridge_test")


def test_mirror_header_cites_every_reference_class():
    src = open(os.path.join(H.ROOT, "include", "atomdev_mirror.hpp")).read()
    for cls in ("RHS_Atmosphere", "RungeKutta_Atmosphere", "Pressure_Atm", "BC_Atmosphere",
                "RHS_Hydrosphere", "RungeKutta_Hydrosphere", "Pressure_Hyd", "BC_Hydrosphere"):
        assert "class %s " % cls in src, cls
    for method in ("solveRungeKutta_3D_Atmosphere", "solveRungeKutta_2D_Atmosphere", "computePressure_3D",
                   "computePressure_2D", "BC_radius", "BC_theta", "BC_phi", "solveRungeKutta_3D_Hydrosphere",
                   "solveRungeKutta_2D_Hydrosphere", "RB_radius", "RB_theta", "RB_phi"):
        assert method in src, method


@pytest.mark.gpu
def test_reference_call_sites_through_mirror_classes(tmp_path):
    if not os.path.exists(BIN):
        pytest.skip("oracle/_ref/bridge_test not built (needs the reference sources)")
    r = subprocess.run([BIN, str(tmp_path)], capture_output=True, text=True, timeout=300)
    print(r.stdout)
    print(r.stderr)
    assert r.returncode == 0 and "BRIDGE PASS" in r.stdout, r.stdout + r.stderr
