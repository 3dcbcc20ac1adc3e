#!/usr/bin/env python3
"""HBM traffic per launch of a kernel from the FETCH_SIZE / WRITE_SIZE passes of tools/pmc.sh, calibrated
against the streaming copy kernel (k_copy_lists: 9 fields read and 9 written, 8 bytes per lane like the sweeps)
that ran in the same pass:  pmc_traffic.py <tag> <kernel-substring> <im> <jm> <km> [fields copied: 9 atm, 6 hyd] > profiles/pmc_<...>.json"""
import collections, csv, glob, json, sys
tag, sub = sys.argv[1], sys.argv[2]
n3 = int(sys.argv[3]) * int(sys.argv[4]) * int(sys.argv[5])
copy_bytes = (int(sys.argv[6]) if len(sys.argv) > 6 else 9) * n3 * 8  # each way
def avg(passname, counter, sub):
    vals = collections.defaultdict(float)
    for f in glob.glob("gpurun_out/pmc_%s_%s/*counter_collection.csv" % (tag, passname)):
        for r in csv.DictReader(open(f)):
            if sub in r["Kernel_Name"] and r["Counter_Name"] == counter:
                vals[r["Dispatch_Id"]] += float(r["Counter_Value"])
    v = list(vals.values())
    return sum(v) / len(v), len(v)
out = {"kernel": sub, "tag": tag}
for passname, counter, key in (("fetch", "FETCH_SIZE", "read"), ("write", "WRITE_SIZE", "write")):
    k, nk = avg(passname, counter, sub)
    c, nc = avg(passname, counter, "k_copy_lists")
    scale = copy_bytes / c  # bytes per counter unit for 8-byte-per-lane streaming access
    out[key] = {"counter": counter, "kernel_avg": k, "kernel_launches": nk, "copy_avg": c, "copy_launches": nc,
                "copy_known_bytes": copy_bytes, "bytes_per_unit": scale, "bytes_per_launch": k * scale}
out["hbm_bytes_per_launch"] = out["read"]["bytes_per_launch"] + out["write"]["bytes_per_launch"]
try:  # wave-level vector instructions per launch, for the fp64-issue bound (pass sq1)
    out["sq_insts_valu_per_launch"], _ = avg("sq1", "SQ_INSTS_VALU", sub)
    out["sq_wave_cycles_per_launch"], _ = avg("sq1", "SQ_WAVE_CYCLES", sub)
    out["sq_wait_any_per_launch"], _ = avg("sq2", "SQ_WAIT_ANY", sub)
except ZeroDivisionError:
    pass
print(json.dumps(out, indent=1))
