#!/bin/bash
# Collects the evidence profiles/ holds for one round, on the GPU box:
#   gpurun --timeout 1100 -- 'bash tools/profile_round.sh r01'
# then, back in the container:  python tools/collect_profiles.py r01
# Separate passes: bench line (with cpu_baseline), rocprofv3 kernel trace + stats, then one --pmc pass per counter
# (never mixed with other trace domains).
set -o pipefail
tag=${1:-r01}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
python3 bench.py > "$out/bench.json" 2> "$out/bench.err" || { tail -5 "$out/bench.err"; exit 1; }
tail -1 "$out/bench.json"
python3 bench.py --no-pipeline --no-cpu-baseline > "$out/bench_serial.json" 2> "$out/bench_serial.err" || exit 1
tail -1 "$out/bench_serial.json"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- python3 bench.py --steps 400 --warmup 50 --no-cpu-baseline > "$out/stats.log" 2>&1 || { tail -5 "$out/stats.log"; exit 1; }
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline > "$out/pmc_fetch.log" 2>&1 || { tail -5 "$out/pmc_fetch.log"; exit 1; }
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write" -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline > "$out/pmc_write.log" 2>&1 || { tail -5 "$out/pmc_write.log"; exit 1; }
# keep the merge-back small: the per-dispatch traces are not needed, the summaries are
find "$out" -name "*kernel_trace.csv" -size +8M -delete
echo done
