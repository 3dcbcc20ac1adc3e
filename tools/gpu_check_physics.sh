#!/bin/bash
# On the GPU box, from the repo root: the column-physics tests, then (only if they pass) the timing table.
# Every GPU step has its own time limit; the script ends at the first step that fails.
set -u
mkdir -p gpurun_out
timeout -k 10 200 python -u -m pytest tests/test_gpu_parity.py tests/test_loops.py tests/test_bridge.py -x -v --timeout 120 \
    --timeout-method thread -k "ice_scheme or column_physics or saturation or bridge or mirror or radiation" -s > gpurun_out/ice.log 2>&1
rc=$?
grep -E "PASS|FAIL|Error|error|max rel" gpurun_out/ice.log | head -30
if [ $rc -ne 0 ]; then echo "tests failed (rc=$rc): stopping"; exit $rc; fi
timeout -k 10 200 python -u tools/op_table.py 5 physics > gpurun_out/op_phys.md 2> gpurun_out/op_phys.err
rc=$?
tail -6 gpurun_out/op_phys.md
exit $rc
