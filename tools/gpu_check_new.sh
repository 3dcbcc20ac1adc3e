#!/bin/bash
# On the GPU box: the tests named by $1 (a pytest -k expression), one step, its own time limit.
set -u
mkdir -p gpurun_out
timeout -k 10 200 python -u -m pytest tests -m gpu -x -v --timeout 120 --timeout-method thread -k "$1" -s > gpurun_out/new.log 2>&1
rc=$?
grep -E "PASS|FAIL|Error|error|max rel" gpurun_out/new.log | head -40
if [ $rc -ne 0 ]; then tail -30 gpurun_out/new.log; fi
exit $rc
