#!/usr/bin/env python3
"""Prints per-kernel averages of the counters collected by tools/pmc.sh: pmc_summary.py <tag> [kernel-substring]"""
import collections, csv, glob, sys
tag = sys.argv[1]; sub = sys.argv[2] if len(sys.argv) > 2 else "rk3d"
for f in sorted(glob.glob("gpurun_out/pmc_%s_*/*counter_collection.csv" % tag)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            acc[(r["Kernel_Name"].split("(")[1 if r["Kernel_Name"].startswith("(") else 0][:40], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print("%-14s %-42s %-22s %16.1f  x%d" % (f.split("/")[1][4:], k[0], k[1], sum(v) / len(v), len(v)))
