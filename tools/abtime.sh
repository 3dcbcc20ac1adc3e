#!/bin/bash
# usage (GPU box): tools/abtime.sh atm|hyd "<variant numbers>" [lib names under variants/...] -- sweep time of each library and variant, same box, same run
model=$1; vars=$2; shift 2
R=$GRAFT_REPO_ROOT
for lib in default "$@"; do
  for v in $vars; do
    if [ $lib = default ]; then L=$R/atom_amd/libatomdev.so; else L=$R/variants/libatomdev_$lib.so; fi
    echo -n "$lib variant $v: "
    ATOMDEV_RK_VARIANT=$v ATOMDEV_LIBRARY=$L timeout -k 10 60 python3 $R/tools/run_rk.py $model 30 || exit 1
  done
done
