#!/bin/bash
# round 3, batch H (GPU box): specialised-filter generator knobs on a workload (one "VAR=value ..." line per run in $KNOBS)
set -o pipefail
out=${GRAFT_REPO_ROOT:-.}/gpurun_out/r03h
mkdir -p $out
w=${WORKLOAD:-c3}
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  res=$(env $line IPCR_JIT_ASYNC=0 timeout -k 10 200 python3 bench.py --workload $w --no-cpu-baseline --no-others --no-traffic --steps 600 --warmup 100 2>$out/knob$i.err | python3 -c "import sys,json; d=json.loads(sys.stdin.read().splitlines()[-1]); print(d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['ms_per_step'], d['config']['products_per_step'])") || { tail -3 $out/knob$i.err; exit 1; }
  echo "$w $line -> $res"
done <<< "${KNOBS:-A=0}"
