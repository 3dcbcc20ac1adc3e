#!/bin/bash
# On the GPU box, from the repo root: the slab measurements of bench.py - the 0.1-degree grid's slab (with a kernel trace),
# the 0.25-degree grid as one domain and as two slabs of one process on one card.
set -u
R=$PWD
O=$R/gpurun_out/slabs
mkdir -p $O
timeout -k 10 400 python bench.py --mode slabs --grid 0.1 --slab-of 16 --steps 10 --warmup 2 --windows 3 > $O/tenth_degree_slab_bench.json 2> $O/tenth.err || { tail -5 $O/tenth.err; exit 1; }
tail -c 700 $O/tenth_degree_slab_bench.json; echo
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o tenth -- python3 $R/bench.py --mode slabs --grid 0.1 --slab-of 16 --steps 6 --warmup 2 --windows 1 > $O/tenth_prof.log 2>&1 || { tail -5 $O/tenth_prof.log; exit 1; }
cd $R
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/tenth_degree_slab_kernel_stats.csv && cut -c1-160 "$f" | head -12
timeout -k 10 300 python bench.py --mode slabs --grid 0.25 --slabs 1 --steps 40 --warmup 4 --windows 5 > $O/quarter_degree_one_domain_bench.json 2>> $O/tenth.err || exit 1
timeout -k 10 300 python bench.py --mode slabs --grid 0.25 --slabs 2 --steps 40 --warmup 4 --windows 5 > $O/quarter_degree_two_slabs_one_card_bench.json 2>> $O/tenth.err || exit 1
python3 - <<'PY'
import json
a=json.load(open("gpurun_out/slabs/quarter_degree_one_domain_bench.json")); b=json.load(open("gpurun_out/slabs/quarter_degree_two_slabs_one_card_bench.json"))
print("0.25 degree: one domain %.3f ms/step, two slabs on one card %.3f ms/step, ratio %.3f" % (a["ms_per_step"], b["ms_per_step"], b["ms_per_step"]/a["ms_per_step"]))
PY
