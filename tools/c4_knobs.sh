#!/bin/bash
# dev tool (GPU box): C4 index sweep under generator knobs, one "VAR=value [VAR=value ...]" line per run on stdin
#   printf "A=0\nIPCR_INDEX_XLDS=1\nIPCR_INDEX_XVALU=16\n" | bash tools/c4_knobs.sh
out=${GRAFT_REPO_ROOT:-.}/gpurun_out
mkdir -p $out
while read -r line; do
  [ -z "$line" ] && continue
  res=$(env $line timeout -k 10 200 python3 bench.py --workload c4 --no-cpu-baseline --no-others --no-traffic --steps 12 --warmup 2 2>$out/knob.err | python3 -c "import sys,json; d=json.loads(sys.stdin.read().splitlines()[-1]); print(d['roofline']['avg_launch_ms'], d['ms_per_step'], d['config']['products_per_step'])") || { tail -3 $out/knob.err; exit 1; }
  echo "$line -> $res"
done
