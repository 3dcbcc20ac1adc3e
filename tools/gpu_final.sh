#!/bin/bash
# On the GPU box, from the repo root: the whole GPU suite, then (only while everything passes) the measurements the
# committed profiles come from.  Every GPU step has its own time limit; the script ends at the first failing step.
set -u
R=$PWD
mkdir -p gpurun_out/final
step() {  # step <seconds> <log> <command...>
  local secs=$1 log=$2; shift 2
  echo "== $* (limit ${secs}s)"
  timeout -k 10 $secs "$@" > $log 2>&1
  local rc=$?
  if [ $rc -ne 0 ]; then echo "FAILED rc=$rc: $*"; tail -30 $log; exit $rc; fi
}
# usage: tools/gpu_final.sh tests | measure   (two calls: a call on the GPU box is limited to 20 minutes)
if [ "${1:-tests}" = tests ]; then
step 1000 gpurun_out/final/pytest.log python -u -m pytest tests -m gpu -x -q --timeout 900 --timeout-method thread
tail -3 gpurun_out/final/pytest.log
step 150 gpurun_out/final/smoke.log python -u -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')"
tail -3 gpurun_out/final/smoke.log
echo "ALL STEPS PASSED"
exit 0
fi
step 300 gpurun_out/final/bench.json python -u bench.py
tail -1 gpurun_out/final/bench.json
cd /tmp && export TMPDIR=/tmp
step 300 $R/gpurun_out/final/rocprof.log rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final/prof -o bench -- python3 $R/bench.py --steps 200 --warmup 20
cd $R
step 300 gpurun_out/final/bench_quarter_degree.json python -u bench.py --grid 0.25 --steps 40 --warmup 6 --no-cpu
tail -c 600 gpurun_out/final/bench_quarter_degree.json
step 400 gpurun_out/final/op_table.md python -u tools/op_table.py 20
tail -36 gpurun_out/final/op_table.md
step 300 gpurun_out/final/loop_time.txt python -u tools/loop_time.py 4
cat gpurun_out/final/loop_time.txt
step 120 gpurun_out/final/rk2d_time.txt python -u tools/rk2d_time.py 20
cat gpurun_out/final/rk2d_time.txt
step 200 gpurun_out/final/atm_slice_time.txt python -u tools/atm_slice_time.py 0 3
cat gpurun_out/final/atm_slice_time.txt
step 200 gpurun_out/final/hyd_slice_time.txt python -u tools/hyd_slice_time.py
cat gpurun_out/final/hyd_slice_time.txt
step 300 gpurun_out/final/hyd_series.txt python -u tools/hyd_series_time.py
cat gpurun_out/final/hyd_series.txt
# the N > 1 path of bench.py, rehearsed with two ranks sharing the card (gloo: RCCL refuses two ranks on one GPU)
step 300 gpurun_out/final/bench_2ranks_gloo.json python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29581 bench.py --gpus 2 --backend gloo --steps 100 --warmup 10 --no-cpu
tail -1 gpurun_out/final/bench_2ranks_gloo.json
echo "ALL STEPS PASSED"
