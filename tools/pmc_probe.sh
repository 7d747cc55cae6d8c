#!/bin/bash
# PMC passes (one --pmc set per pass, nothing else traced) over a short run:
#   tools/pmc_probe.sh [-k kernel_substring] [-c "python3 script args"] "SET1" "SET2" ...
# default command: the bench loop; default kernel: ipcr_filter
export TMPDIR=/tmp
kernel=ipcr_filter
cmd="python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline"
while getopts "k:c:" o; do case $o in k) kernel=$OPTARG;; c) cmd=$OPTARG;; esac; done
shift $((OPTIND-1))
i=0
for set in "$@"; do
  i=$((i+1))
  out=$GRAFT_REPO_ROOT/gpurun_out/pmcp_$i
  rm -rf $out
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out -- $cmd > $out.log 2>&1 || { tail -3 $out.log; continue; }
  python3 - "$out" "$kernel" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if sys.argv[2] in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, v in sorted(acc.items()):
    print("%-28s avg %.4g over %d launches" % (k, sum(v) / len(v), len(v)))
PY
done
