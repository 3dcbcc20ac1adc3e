#!/usr/bin/env python3
"""Writes include/atom_params.def - the models' configuration parameters (section, type, name, default) as one table - from
the files the reference's own param.py generates (oracle/gen_ref_headers.py puts them under oracle/_ref/gen).  Run in
the build container when the reference's parameter list changes; tests/test_models.py checks the committed table against
the generated files whenever they are present."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GEN = os.path.join(ROOT, "oracle", "_ref", "gen")


def read_model(subdir, stem_params, stem_defaults):
    types, order, section_of = {}, [], {}
    section = None
    for line in open(os.path.join(GEN, subdir, stem_params)):
        m = re.match(r"//\s*(\w+) section", line)
        if m:
            section = m.group(1)
            continue
        m = re.match(r"\s*(string|double|int|bool)\s+(\w+);", line)
        if m:
            types[m.group(2)] = m.group(1)
            order.append(m.group(2))
            section_of[m.group(2)] = section
    defaults = {}
    for line in open(os.path.join(GEN, subdir, stem_defaults)):
        m = re.match(r"\s*(\w+)\s*=\s*(.*);\s*$", line)
        if m:
            defaults[m.group(1)] = m.group(2)
    return [(section_of[n], types[n], n, defaults[n]) for n in order]


def main():
    atm = read_model("atmosphere", "AtmosphereParams.h.inc", "cAtmosphereDefaults.cpp.inc")
    hyd = read_model("hydrosphere", "HydrosphereParams.h.inc", "cHydrosphereDefaults.cpp.inc")
    common = [p for p in atm if p[0] == "common"]
    assert common == [p for p in hyd if p[0] == "common"]
    out = ["/* The configuration parameters of the two models (param.py:9-178): section, C++ type, name, default.  The",
           " * <common> section belongs to both models, <atmosphere> and <hydrosphere> to one each.  Written by",
           " * tools/gen_params_def.py.  Expand with ATOM_PARAM(section, type, name, default) defined. */"]
    for sec, rows in (("common", common), ("atmosphere", [p for p in atm if p[0] == "atmosphere"]),
                      ("hydrosphere", [p for p in hyd if p[0] == "hydrosphere"])):
        out.append("")
        for _, t, n, d in rows:
            out.append("ATOM_PARAM(%s, %s, %s, %s)" % (sec, t, n, d))
    open(os.path.join(ROOT, "include", "atom_params.def"), "w").write("\n".join(out) + "\n")
    print("common %d, atmosphere %d, hydrosphere %d" % (len(common), len(atm) - len(common), len(hyd) - len(common)))


if __name__ == "__main__":
    main()
