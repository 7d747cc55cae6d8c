#!/bin/bash
# quick look at the C4 index kernel's SQ / LDS counters (dev tool; profile_round.sh collects what profiles/ holds)
set -o pipefail
out=$GRAFT_REPO_ROOT/gpurun_out/c4pmc
mkdir -p "$out"
export TMPDIR=/tmp
args="--workload c4 --no-cpu-baseline --no-others --no-traffic --steps 3 --warmup 1"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --output-format csv -d "$out/sq" -- python3 bench.py $args > "$out/sq.log" 2>&1 || { tail -5 "$out/sq.log"; exit 1; }
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY --output-format csv -d "$out/lds" -- python3 bench.py $args > "$out/lds.log" 2>&1 || { tail -5 "$out/lds.log"; exit 1; }
python3 - <<'PY'
import csv, glob, os
out = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "c4pmc")
for d in ("sq", "lds"):
    acc = {}
    for f in glob.glob(os.path.join(out, d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f, newline="")):
            if "ipcr_index_filter" in row["Kernel_Name"]:
                acc.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print(k, len(v), "%.4g" % (sum(v) / len(v)))
PY
find "$out" -name "*.csv" -size +2M -delete
