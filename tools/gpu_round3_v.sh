#!/bin/bash
# round 3, batch V (GPU box): chunk worker pool bound to the device's socket (main thread included) against unbound, alternating
cd ${GRAFT_REPO_ROOT:-.}
for rep in 1 2 3; do
  for a in "" "--bind"; do
    r=$(timeout -k 10 300 ipcr_amd/chunk_workers $a 125000000 4000000 8 16 | python3 -c "import sys,json; d=json.loads(sys.stdin.read().splitlines()[-1]); print({k:v for k,v in d.items() if k.startswith('gbases') or k.startswith('workers')})") || exit 1
    echo "'$a' -> $r"
  done
done
