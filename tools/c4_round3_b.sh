#!/bin/bash
# round 3, batch B (GPU box): index tests, then the C4 sweep under generator knobs (one "VAR=value ..." line per run in $KNOBS)
set -o pipefail
out=${GRAFT_REPO_ROOT:-.}/gpurun_out/r03b
mkdir -p $out
if [ -z "$SKIP_TESTS" ]; then
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "index or c4 or shards or leftovers" > $out/tests.log 2>&1 || { tail -30 $out/tests.log; exit 1; }
tail -2 $out/tests.log
fi
run() { # name, env...
  name=$1; shift
  env "$@" timeout -k 10 300 python3 bench.py --workload c4 --no-cpu-baseline --no-others --no-traffic --steps 12 --warmup 2 > $out/$name.json 2> $out/$name.err || { echo "$name FAILED"; tail -5 $out/$name.err; return 1; }
  python3 -c "import sys,json; d=json.loads(open('$out/$name.json').read().splitlines()[-1]); print('$name', 'sweep_ms', d['roofline']['avg_launch_ms'], 'ms_per_step', d['ms_per_step'], 'products', d['config']['products_per_step'])"
}
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  echo "run$i: $line"
  run run$i $line || exit 1
done <<< "${KNOBS:-A=0}"
