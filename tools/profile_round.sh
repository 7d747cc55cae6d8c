#!/bin/bash
# Collects the evidence profiles/ holds for one round, on the GPU box:
#   gpurun --timeout 1150 -- 'bash tools/profile_round.sh r04 bench'; gpurun --timeout 1150 -- 'bash tools/profile_round.sh r04 prof "c2 c2n c3"'; ... prof "c4 c4n"
# then, back in the container:  python tools/collect_profiles.py r03
# Separate passes: bench line (all workloads + cpu_baseline), then per workload a rocprofv3 kernel trace + stats and
# one --pmc pass per counter set (never mixed with other trace domains).
set -o pipefail
tag=${1:-r04}
stage=${2:-all}   # bench | prof | all (a gpurun call is limited to 20 minutes: two calls for one round)
wls=${3:-"c2 c2n c3 c4 c4n"}   # workloads of the prof stage (a kernel that has not changed keeps its profile)
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
export IPCR_JIT_ASYNC=0   # every pass on the panel's own kernels, from the first one
if [ $stage != prof ]; then
python3 bench.py > "$out/bench.json" 2> "$out/bench.err" || { tail -5 "$out/bench.err"; exit 1; }
tail -c 600 "$out/bench.json"; echo
python3 bench.py --no-pipeline --no-cpu-baseline --no-others --no-traffic > "$out/bench_serial.json" 2> "$out/bench_serial.err" || exit 1
for w in c3 c4; do
  python3 bench.py --workload $w --no-others > "$out/bench_$w.json" 2> "$out/bench_$w.err" || { tail -5 "$out/bench_$w.err"; exit 1; }
done
fi
[ $stage = bench ] && { echo done; exit 0; }
for w in $wls; do
  steps=400; warm=50; psteps=4
  case $w in c4*) steps=20; warm=3; psteps=3;; esac
  bw=$w
  unset IPCR_SPECIALIZE
  # c2g: C2 through the table-driven kernel (what runs while hiprtc builds a small panel's kernels, and for panels beyond every limit)
  case $w in c2g) bw=c2; steps=12; warm=2; psteps=2; export IPCR_SPECIALIZE=0;; esac
  args="--workload $bw --no-cpu-baseline --no-others --no-traffic"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats_$w" -- python3 bench.py $args --steps $steps --warmup $warm > "$out/stats_$w.log" 2>&1 || { tail -5 "$out/stats_$w.log"; exit 1; }
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch_$w" -- python3 bench.py $args --steps $psteps --warmup 1 > "$out/pmc_fetch_$w.log" 2>&1 || { tail -5 "$out/pmc_fetch_$w.log"; exit 1; }
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write_$w" -- python3 bench.py $args --steps $psteps --warmup 1 > "$out/pmc_write_$w.log" 2>&1 || { tail -5 "$out/pmc_write_$w.log"; exit 1; }
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --output-format csv -d "$out/pmc_sq_$w" -- python3 bench.py $args --steps $psteps --warmup 1 > "$out/pmc_sq_$w.log" 2>&1 || { tail -5 "$out/pmc_sq_$w.log"; exit 1; }
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY --output-format csv -d "$out/pmc_lds_$w" -- python3 bench.py $args --steps $psteps --warmup 1 > "$out/pmc_lds_$w.log" 2>&1 || { tail -5 "$out/pmc_lds_$w.log"; exit 1; }
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d "$out/pmc_grbm_$w" -- python3 bench.py $args --steps $psteps --warmup 1 > "$out/pmc_grbm_$w.log" 2>&1 || { tail -5 "$out/pmc_grbm_$w.log"; exit 1; }
  echo "profiled $w"
done
# keep the merge-back small (gpurun merges at most 64 MiB): of the per-dispatch counter files only the rows of our kernels and
# the three columns collect_profiles.py reads; the per-dispatch traces are not needed, the summaries are
python3 - "$out" <<'PY'
import csv, glob, os, sys
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    rows = []
    with open(f, newline="") as fh:
        for row in csv.DictReader(fh):
            if "ipcr_" in row.get("Kernel_Name", "") or "filter_generic" in row.get("Kernel_Name", ""):
                rows.append({k: row[k] for k in ("Kernel_Name", "Counter_Name", "Counter_Value")})
    with open(f, "w", newline="") as fh:
        w = csv.DictWriter(fh, fieldnames=["Kernel_Name", "Counter_Name", "Counter_Value"])
        w.writeheader()
        w.writerows(rows)
PY
find "$out" -name "*kernel_trace.csv" -delete
find "$out" -name "*agent_info.csv" -delete
unset IPCR_SPECIALIZE
echo done
