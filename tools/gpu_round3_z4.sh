#!/bin/bash
# round 3, batch Z4 (GPU box): the next unit taken and loaded under the last chunk's walk (IPCR_INDEX_AHEAD) -- parity, sweep time
set -o pipefail
cd ${GRAFT_REPO_ROOT:-.}
out=gpurun_out/r03z4
mkdir -p $out
timeout -k 10 400 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -k "index or c4 or large_panel or slots" > $out/tests.log 2>&1 || { tail -20 $out/tests.log; exit 1; }
tail -2 $out/tests.log
printf "${KNOBS:-IPCR_INDEX_AHEAD=0\nIPCR_INDEX_AHEAD=1\nIPCR_INDEX_AHEAD=0\nIPCR_INDEX_AHEAD=1\n}" | bash tools/c4_knobs.sh || exit 1
