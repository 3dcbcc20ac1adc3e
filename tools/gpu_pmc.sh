#!/bin/bash
# On the GPU box, from the repo root: the counter passes of both sweeps on the 1-degree grid (tools/pmc.sh), one rocprofv3
# run per counter group.  tools/pmc_traffic.py turns what comes back under gpurun_out/ into profiles/pmc_*_rk3d.json.
set -u
tag=${1:-r02}
tools/pmc.sh ${tag}a atm || exit 1
tools/pmc.sh ${tag}h hyd || exit 1
echo "PMC PASSES DONE"
