#!/bin/bash
# On the GPU box: times the column physics with every library under variants/; ends at the first failing step.
set -u
mkdir -p gpurun_out
for lib in variants/libatomdev_*.so; do
  echo "== $lib"
  ATOMDEV_LIBRARY=$PWD/$lib timeout -k 10 120 python -u tools/op_table.py 5 physics 2> gpurun_out/variant.err | tail -2
  rc=${PIPESTATUS[0]}
  if [ $rc -ne 0 ]; then echo "failed (rc=$rc): stopping"; tail -5 gpurun_out/variant.err; exit $rc; fi
done
