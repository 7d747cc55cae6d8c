#!/bin/bash
# round 3, batch Z3 (GPU box): per-XCD unit counters of the index kernel (IPCR_INDEX_XCD) -- parity, sweep time, FETCH_SIZE
set -o pipefail
cd ${GRAFT_REPO_ROOT:-.}
out=gpurun_out/r03z3
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 400 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -k "index or c4 or large_panel or slots" > $out/tests.log 2>&1 || { tail -20 $out/tests.log; exit 1; }
tail -2 $out/tests.log
printf "IPCR_INDEX_XCD=0\nIPCR_INDEX_XCD=1\nIPCR_INDEX_XCD=0\nIPCR_INDEX_XCD=1\n" | bash tools/c4_knobs.sh || exit 1
args="--workload c4 --no-cpu-baseline --no-others --no-traffic --steps 3 --warmup 1"
for x in 0 1; do
  export IPCR_INDEX_XCD=$x
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch$x -- python3 bench.py $args > $out/fetch$x.log 2>&1 || { tail -5 $out/fetch$x.log; exit 1; }
  python3 - <<PY
import csv, glob
v=[float(r["Counter_Value"]) for f in glob.glob("$out/fetch$x/**/*counter_collection.csv", recursive=True) for r in csv.DictReader(open(f, newline="")) if "ipcr_index_filter" in r["Kernel_Name"] and r["Counter_Name"]=="FETCH_SIZE"]
print("IPCR_INDEX_XCD=$x FETCH_SIZE KiB avg", sum(v)/len(v), "launches", len(v), "-> bytes x2:", sum(v)/len(v)*2048)
PY
done
find $out -name "*.csv" -size +2M -delete
