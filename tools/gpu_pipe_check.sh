#!/bin/bash
# On the GPU box, from the repo root: the parity tests that cover the atmosphere sweep's variants, then the sweep time
# of the tolerance form (2) and the pipelined form (3) on the 1-degree grid.
set -u
R=$PWD
mkdir -p gpurun_out/pipe
timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 300 --timeout-method thread \
  -k "rk_variants_same_bits or limited_is_sweep or full_1deg_real_mask_rk3d" > gpurun_out/pipe/pytest.log 2>&1
rc=$?
tail -15 gpurun_out/pipe/pytest.log
[ $rc -ne 0 ] && exit $rc
bash tools/abtime.sh atm "2 3" || exit 1
