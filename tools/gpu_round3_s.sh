#!/bin/bash
# round 3, batch S (GPU box): the full GPU suite, then the driver's bench command
set -o pipefail
cd ${GRAFT_REPO_ROOT:-.}
out=gpurun_out/r03s
mkdir -p $out
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; rc=$?; tail -3 $out/pytest.log; [ $rc = 0 ] || exit 1
WORKLOADS=driver bash tools/gpu_round3_g.sh
cp gpurun_out/r03g/bench_driver.json $out/ 2>/dev/null
