#!/bin/bash
# round 3, batch Y (GPU box): the resident-genome lanes of bench.py against the number of hardware queues (C2, then C5)
cd ${GRAFT_REPO_ROOT:-.}
for w in c2 c5; do
  for q in 2 4 8 16; do
    r=$(GPU_MAX_HW_QUEUES=$q timeout -k 10 200 python3 bench.py --workload $w --no-cpu-baseline --no-others --no-traffic 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['avg_launch_ms'], d['value'])") || exit 1
    echo "$w queues $q -> $r"
  done
done
