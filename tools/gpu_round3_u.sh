#!/bin/bash
# round 3, batch U (GPU box): what the driver runs at round end -- smoke(), the GPU suite, the bench command
set -o pipefail
cd ${GRAFT_REPO_ROOT:-.}
out=gpurun_out/r03u
mkdir -p $out
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 || exit 1
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; rc=$?; tail -3 $out/pytest.log; [ $rc = 0 ] || exit 1
s=$(date +%s)
timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver.json 2> $out/bench_driver.err || { tail -5 $out/bench_driver.err; exit 1; }
echo "driver bench took $(( $(date +%s) - s )) s"
python3 - <<PY
import json
d = json.loads(open("$out/bench_driver.json").read().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "windows", "window_ms_min", "window_ms_max")}, d["roofline"]["frac"], d["roofline"]["traffic"], d["cpu_baseline"]["value"])
for k, v in d["config"]["other_workloads"].items():
    print(" ", k, {kk: v.get(kk) for kk in ("gbases_per_s", "sweep_ms", "gbases_per_s_8_workers", "gbases_per_s_16_workers", "gbases_per_s_whole_record", "load_s", "total_s") if kk in v})
PY
