#!/bin/bash
# usage (on the GPU box, from the repo root): tools/pmc.sh <tag> <model> [IMxJMxKM] -- one rocprofv3 --pmc pass per counter group
set -u
export RUN_RK_MOVE=1
export ATOMDEV_COPY_ONE_BY_ONE=1   # the calibrating copy moves 8 bytes a lane, like the sweeps
tag=$1; model=$2; dims=${3:-41x181x361}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
pass() { # name counters...
  local name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $R/gpurun_out/pmc_${tag}_${name} -o p -- python3 $R/tools/run_rk.py $model 3 $dims > $R/gpurun_out/pmc_${tag}_${name}.log 2>&1
  echo "pass $name rc=$?"
}
pass sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES || exit 1
pass sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM || exit 1
[ "${PMC_SQ_ONLY:-0}" = 1 ] && exit 0
pass fetch FETCH_SIZE || exit 1
pass write WRITE_SIZE || exit 1
pass grbm GRBM_GUI_ACTIVE || exit 1
