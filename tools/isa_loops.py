#!/usr/bin/env python3
"""Lists memory operations, waits and barriers inside each loop of an AMDGPU assembly file (tools/isa_loops.py file.s [prefix])."""
import re, sys
lines = open(sys.argv[1]).read().split('\n')
pref = sys.argv[2] if len(sys.argv) > 2 else '.LBB1_'
hdr = {}
for n, l in enumerate(lines):
    m = re.match(r'^(' + re.escape(pref) + r'\d+):.*Loop Header', l)
    if m: hdr[m.group(1)] = n
for lab, n0 in hdr.items():
    ends = [n for n, l in enumerate(lines) if re.search(r's_c?branch\w* ' + re.escape(lab) + r'$', l.strip())]
    n1 = max(ends)
    body = lines[n0:n1 + 1]
    ev = []
    for n, l in enumerate(body):
        if re.search(r'global_load|global_store|flat_|vmcnt|s_barrier|s_load|v_readlane|v_writelane|scratch_', l):
            ev.append('%d:%s' % (n, re.sub(r'\s+', ' ', l.strip())[:44]))
    nf64 = sum(1 for l in body if re.search(r'v_\w+_f64', l))
    print(lab, 'lines', len(body), 'f64', nf64)
    print('   ' + '\n   '.join(ev))
