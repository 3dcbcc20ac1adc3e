#!/bin/bash
# usage (GPU box, repo root): tools/kstats.sh <tag> <command...> -- rocprofv3 kernel trace + stats of the command as csv
tag=$1; shift
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ks_$tag -o ks -- "$@" > $R/gpurun_out/ks_$tag.log 2>&1
rc=$?
f=$(find $R/gpurun_out/ks_$tag -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" $R/gpurun_out/ks_$tag.csv && cut -c1-220 "$f" | head -30
exit $rc
