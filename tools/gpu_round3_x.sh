#!/bin/bash
# round 3, batch X (GPU box): the index kernel's cost map over panels -- rows, k, terminal window (tools/c4_probe.py)
cd ${GRAFT_REPO_ROOT:-.}
for cfg in "64 2 3" "256 2 3" "1024 2 3" "1024 1 3" "1024 3 3" "1024 2 0" "1024 2 5"; do
  set -- $cfg
  echo "== rows $1 k $2 tw $3"
  C4_K=$2 C4_TW=$3 timeout -k 10 280 python3 tools/c4_probe.py $1 2>/dev/null | grep "^scan\|CompilePanel" | tail -2 | cut -c1-200
done
