#!/bin/bash
# round 3, batch A (GPU box): where does the C4 index sweep lose its time?  static vs dynamic units, wave timelines
set -o pipefail
out=${GRAFT_REPO_ROOT:-.}/gpurun_out/r03a
mkdir -p $out
run() { # name, env...
  name=$1; shift
  env "$@" timeout -k 10 300 python3 bench.py --workload c4 --no-cpu-baseline --no-others --no-traffic --steps 12 --warmup 2 > $out/$name.json 2> $out/$name.err || { echo "$name FAILED"; tail -5 $out/$name.err; return 1; }
  python3 -c "import sys,json; d=json.loads(open('$out/$name.json').read().splitlines()[-1]); print('$name', 'sweep_ms', d['roofline']['avg_launch_ms'], 'ms_per_step', d['ms_per_step'], 'products', d['config']['products_per_step'])"
}
run static IPCR_INDEX_DYNAMIC=0 &&
run dynamic IPCR_INDEX_DYNAMIC=1 &&
run static_stamps IPCR_INDEX_DYNAMIC=0 IPCR_INDEX_STAMPS=$out/static.stamps &&
run dynamic_stamps IPCR_INDEX_DYNAMIC=1 IPCR_INDEX_STAMPS=$out/dynamic.stamps &&
python3 tools/index_stamps.py $out/static.stamps | tail -6 && python3 tools/index_stamps.py $out/dynamic.stamps | tail -6
rm -f $out/*.stamps
