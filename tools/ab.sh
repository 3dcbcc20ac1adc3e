#!/bin/bash
# usage (GPU box): tools/ab.sh <model> <variant>... -- times run_rk.py with each variants/libatomdev_<variant>.so
model=$1; shift
R=$GRAFT_REPO_ROOT
timeout -k 10 60 python3 $R/tools/run_rk.py $model 10 || exit 1
for v in "$@"; do
  echo "== $v"
  ATOMDEV_LIBRARY=$R/variants/libatomdev_$v.so timeout -k 10 60 python3 $R/tools/run_rk.py $model 10 || exit 1
done
