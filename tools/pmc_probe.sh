#!/bin/bash
# PMC passes over a short bench run (one --pmc set per pass, nothing else traced): tools/pmc_probe.sh "SET1" "SET2" ...
export TMPDIR=/tmp
i=0
for set in "$@"; do
  i=$((i+1))
  out=$GRAFT_REPO_ROOT/gpurun_out/pmcp_$i
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline > $out.log 2>&1 || { tail -3 $out.log; continue; }
  python3 - "$out" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "ipcr_filter" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, v in sorted(acc.items()):
    print("%-28s avg %.4g over %d launches" % (k, sum(v) / len(v), len(v)))
PY
done
