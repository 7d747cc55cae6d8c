#!/bin/bash
# round 3, batch G (GPU box): the driver's bench command, and the other workloads as the main line
set -o pipefail
out=${GRAFT_REPO_ROOT:-.}/gpurun_out/r03g
mkdir -p $out
for w in ${WORKLOADS:-driver c4 c3}; do
  if [ "$w" = driver ]; then args="--gpus 1 --steps 20 --warmup 5"; else args="--workload $w --no-others"; fi
  timeout -k 10 900 python3 bench.py $args > $out/bench_$w.json 2> $out/bench_$w.err || { echo "bench $w failed"; tail -15 $out/bench_$w.err; exit 1; }
  python3 - <<PY
import json
d = json.loads(open("$out/bench_$w.json").read().splitlines()[-1])
c = d["config"]
print("$w", {k: d[k] for k in ("value", "ms_per_step", "windows", "window_ms_min", "window_ms_max")}, "roofline", d["roofline"]["frac"], d["roofline"]["avg_launch_ms"])
print("  compile", c.get("panel_compile"), "cpu", {k: d.get("cpu_baseline", {}).get(k) for k in ("value", "cores", "threads_busy", "panel_compile_s", "products_in_sample")})
for k, v in c.get("other_workloads", {}).items():
    print("  other", k, {kk: v.get(kk) for kk in ("ms_per_step", "gbases_per_s", "sweep_ms", "panel_compile", "gbases_per_s_8_workers", "gbases_per_s_16_workers", "gbases_per_s_1_worker", "gbases_per_s_whole_record", "load_s", "total_s") if kk in v})
PY
done
