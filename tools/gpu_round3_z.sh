#!/bin/bash
# round 3, batch Z (GPU box): the index drain as a stack (IPCR_INDEX_STACK_DRAIN) -- parity, sweep time, counters
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout -k 10 400 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -k "index or c4 or large_panel or slots" > gpurun_out/z_tests.log 2>&1 || { tail -20 gpurun_out/z_tests.log; exit 1; }
tail -2 gpurun_out/z_tests.log
printf "IPCR_INDEX_STACK_DRAIN=0\nIPCR_INDEX_STACK_DRAIN=1\nIPCR_INDEX_STACK_DRAIN=0\nIPCR_INDEX_STACK_DRAIN=1\n" | bash tools/c4_knobs.sh || exit 1
bash tools/pmc_c4.sh
