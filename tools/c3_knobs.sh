#!/bin/bash
# dev tool (GPU box): C2 / C3 sweep under generator knobs, one "WORKLOAD VAR=value [VAR=value ...]" line per run on stdin
#   printf "c3 A=0\nc3 IPCR_JIT_FILTER_LEN=16\n" | bash tools/c3_knobs.sh
out=${GRAFT_REPO_ROOT:-.}/gpurun_out
mkdir -p $out
while read -r w line; do
  [ -z "$w" ] && continue
  res=$(env $line timeout -k 10 200 python3 bench.py --workload $w --no-cpu-baseline --no-others --no-traffic --steps 300 --warmup 50 2>$out/knob.err | python3 -c "import sys,json; d=json.loads(sys.stdin.read().splitlines()[-1]); print(d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['ms_per_step'], d['config']['products_per_step'])") || { tail -3 $out/knob.err; exit 1; }
  echo "$w $line -> $res"
done
