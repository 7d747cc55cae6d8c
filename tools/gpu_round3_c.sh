#!/bin/bash
# round 3, batch C (GPU box): the GPU suite (PYTEST_K: a -k expression)
set -o pipefail
out=${GRAFT_REPO_ROOT:-.}/gpurun_out/r03c
mkdir -p $out
if [ -n "$PYTEST_K" ]; then
  timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q -k "$PYTEST_K" > $out/tests.log 2>&1
else
  timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $out/tests.log 2>&1
fi
rc=$?
tail -25 $out/tests.log
exit $rc
