#!/bin/bash
# On the GPU box: the radiation test, then the column-physics timings, with every library under variants/; ends at
# the first failing step.
set -u
mkdir -p gpurun_out
for lib in variants/libatomdev_*.so; do
  echo "== $lib"
  ATOMDEV_LIBRARY=$PWD/$lib timeout -k 10 120 python -u -m pytest tests/test_gpu_parity.py -x -q --timeout 100 --timeout-method thread -k radiation -s > gpurun_out/variant_test.log 2>&1
  rc=$?
  grep -E "max rel|passed|failed" gpurun_out/variant_test.log
  if [ $rc -ne 0 ]; then echo "test failed (rc=$rc): stopping"; tail -20 gpurun_out/variant_test.log; exit $rc; fi
  ATOMDEV_LIBRARY=$PWD/$lib timeout -k 10 120 python -u tools/op_table.py 5 physics > gpurun_out/variant.md 2> gpurun_out/variant.err
  rc=$?
  if [ $rc -ne 0 ]; then echo "timing failed (rc=$rc): stopping"; tail -5 gpurun_out/variant.err; exit $rc; fi
  grep radiation gpurun_out/variant.md
done
