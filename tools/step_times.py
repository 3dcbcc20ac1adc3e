#!/usr/bin/env python3
"""Sums the "step <ms> <call>" lines a run with ATOMDEV_STEP_TIMES=1 wrote to stderr, by call: tools/step_times.py file."""
import collections
import re
import sys

tot, cnt = collections.Counter(), collections.Counter()
for line in open(sys.argv[1]):
    m = re.match(r"step ([0-9.]+) (\w+)", line)
    if m:
        tot[m.group(2)] += float(m.group(1)); cnt[m.group(2)] += 1
print("| call | calls | ms in all | ms per call |\n|---|---|---|---|")
for k, v in tot.most_common():
    print("| %s | %d | %.1f | %.3f |" % (k, cnt[k], v, v / cnt[k]))
print("| sum | | %.1f | |" % sum(tot.values()))
