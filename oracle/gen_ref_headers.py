"""Runs the reference's own parameter generator (param.py:5-436) so that its generated C++ pieces
exist for the compiled-reference checker: AtmosphereParams.h.inc, cAtmosphereDefaults.cpp.inc,
AtmosphereLoadConfig.cpp.inc, AtmParameters.{h,cpp} and the hydrosphere twins.

Test infrastructure, build container only (the reference does not exist on the GPU box).  Nothing
is copied from the reference and nothing here writes C++: param.py is imported from where it lies
and its main() runs with a scratch directory under oracle/_ref/gen as the working directory, which is
where its relative output paths then point.  main() writes every C++ file first and then fails in
write_pxi (it reads python/pyatom.pyx.template relative to the working directory, and under
Python 3 would mix bytes and str at param.py:288-308); that failure is expected and ignored once
the C++ files are there.
"""
import importlib.util
import os
import sys

REF = os.environ.get("ATOM_REF", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
GEN = os.path.join(HERE, "_ref", "gen")

EXPECTED = [
    "atmosphere/cAtmosphereDefaults.cpp.inc", "atmosphere/AtmosphereLoadConfig.cpp.inc",
    "atmosphere/AtmParameters.cpp", "atmosphere/AtmosphereParams.h.inc", "atmosphere/AtmParameters.h",
    "hydrosphere/cHydrosphereDefaults.cpp.inc", "hydrosphere/HydrosphereLoadConfig.cpp.inc",
    "hydrosphere/HydParameters.cpp", "hydrosphere/HydrosphereParams.h.inc", "hydrosphere/HydParameters.h",
]


def main():
    for d in ("atmosphere", "hydrosphere", "python", "cli", "benchmark"):
        os.makedirs(os.path.join(GEN, d), exist_ok=True)
    spec = importlib.util.spec_from_file_location("atom_ref_param", os.path.join(REF, "param.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    cwd = os.getcwd()
    os.chdir(GEN)
    try:
        mod.main()
    except (IOError, OSError, TypeError) as e:
        print("param.py stopped after the C++ files (expected): %s: %s" % (type(e).__name__, e))
    finally:
        os.chdir(cwd)
    missing = [f for f in EXPECTED if not os.path.isfile(os.path.join(GEN, f))]
    if missing:
        sys.exit("param.py did not produce: " + ", ".join(missing))
    print("generated under %s: %s" % (GEN, ", ".join(EXPECTED)))


if __name__ == "__main__":
    main()
