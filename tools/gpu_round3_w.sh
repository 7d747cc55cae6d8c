#!/bin/bash
# round 3, batch W (GPU box): chunk worker pool against the number of hardware queues the runtime may use
cd ${GRAFT_REPO_ROOT:-.}
for rep in 1 2 3; do
  for q in ${QUEUES:-4 6 8 12 16}; do
    r=$(GPU_MAX_HW_QUEUES=$q timeout -k 10 300 ipcr_amd/chunk_workers 125000000 4000000 ${WORKERS:-8 16} | python3 -c "import sys,json; d=json.loads(sys.stdin.read().splitlines()[-1]); print({k: v for k, v in d.items() if k.startswith('gbases')})") || exit 1
    echo "queues $q -> $r"
  done
done
