#!/bin/bash
# round 3, batch Z10 (GPU box): the specialised filter with the next block's column 0 in the stash (IPCR_JIT_NEIGHBOUR) -- parity, sweep time
set -o pipefail
cd ${GRAFT_REPO_ROOT:-.}
out=gpurun_out/r03z10
mkdir -p $out
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -k "not index and not slots" > $out/tests.log 2>&1 || { tail -20 $out/tests.log; exit 1; }
tail -2 $out/tests.log
for w in c2 c3; do
  for r in 0 1 0 1; do
    IPCR_JIT_NEIGHBOUR=$r IPCR_JIT_ASYNC=0 timeout -k 10 200 python3 bench.py --workload $w --no-cpu-baseline --no-others --no-traffic --steps 600 --warmup 100 2>$out/knob.err | python3 -c "import sys,json; d=json.loads(sys.stdin.read().splitlines()[-1]); print('$w NEIGHBOUR=$r', d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['ms_per_step'], d['config']['products_per_step'], d['config']['panel_compile']['kernels_built_cold_s'])" || { tail -3 $out/knob.err; exit 1; }
  done
done
