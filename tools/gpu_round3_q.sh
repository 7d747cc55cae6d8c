#!/bin/bash
# round 3, batch Q (GPU box): chunk worker pool, knobs one per line in $KNOBS, each run twice (A/B on one box)
set -o pipefail
cd ${GRAFT_REPO_ROOT:-.}
for rep in 1 2; do
while read -r line; do
  [ -z "$line" ] && continue
  r=$(env $line timeout -k 10 300 ipcr_amd/chunk_workers ${CW_ARGS:-} 125000000 4000000 ${WORKERS:-1 8 16} | python3 -c "import sys,json; d=json.loads(sys.stdin.read().splitlines()[-1]); print({k:v for k,v in d.items() if k.startswith('gbases') or k.startswith('workers')})") || exit 1
  echo "$line -> $r"
done <<< "${KNOBS:-A=0}"
done
