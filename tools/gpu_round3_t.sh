#!/bin/bash
# round 3, batch T (GPU box): the seed-index parity tests (SKIP_TESTS=1: not), then the C4 sweep under knobs
set -o pipefail
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r03t
if [ -z "$SKIP_TESTS" ]; then
  timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -k "index or c4 or large_k or device_slots_large" > gpurun_out/r03t/pytest.log 2>&1; rc=$?; tail -4 gpurun_out/r03t/pytest.log; [ $rc = 0 ] || exit 1
fi
printf "${KNOBS:-IPCR_INDEX_TWO_STEP=1\nIPCR_INDEX_TWO_STEP=0\nIPCR_INDEX_TWO_STEP=1\n}" | bash tools/c4_knobs.sh
