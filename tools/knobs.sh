#!/bin/bash
# dev tool (GPU box): A/B runs of one bench workload under generator / runtime knobs -- one run per stdin line,
#   "<workload> [VAR=value ...]"        workload: c2 | c2n | c3 | c4 | c5 | chunk (the native worker pool: args after "--")
# e.g.
#   printf "c3\nc3 IPCR_JIT_PEEL=0\nc4\nc4 IPCR_INDEX_XCD=0\nchunk GPU_MAX_HW_QUEUES=8 -- 125000000 4000000 8 16\n" | bash tools/knobs.sh
# Alternate the lines (A, B, A, B): consecutive runs on one box differ by a per cent or two, boxes of the pool by five.
# Prints sweep ms, roofline fraction, ms per step and the product count (every pass of every run is checked by bench.py).
cd ${GRAFT_REPO_ROOT:-.}
out=gpurun_out
mkdir -p $out
while read -r w line; do
  [ -z "$w" ] && continue
  if [ "$w" = chunk ]; then
    envs=${line%%--*}; args=${line#*--}; [ "$args" = "$line" ] && args="125000000 4000000 8 16"
    res=$(env $envs timeout -k 10 300 ipcr_amd/chunk_workers $args 2>$out/knob.err | python3 -c "import sys,json; d=json.loads(sys.stdin.read().splitlines()[-1]); print({k: v for k, v in d.items() if k.startswith('gbases') or k.startswith('probe')})") || { tail -3 $out/knob.err; exit 1; }
  else
    steps=300; warm=50; case "$w" in c4*) steps=12; warm=2;; esac
    res=$(env IPCR_JIT_ASYNC=0 $line timeout -k 10 280 python3 bench.py --workload $w --no-cpu-baseline --no-others --no-traffic --steps $steps --warmup $warm 2>$out/knob.err | python3 -c "import sys,json; d=json.loads(sys.stdin.read().splitlines()[-1]); print(d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['ms_per_step'], d['config']['products_per_step'])") || { tail -3 $out/knob.err; exit 1; }
  fi
  echo "$w $line -> $res"
done
