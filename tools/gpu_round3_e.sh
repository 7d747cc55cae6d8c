#!/bin/bash
# round 3, batch E (GPU box): chunk path knobs under the native worker pool (one "VAR=value ..." line per run in $KNOBS)
set -o pipefail
out=${GRAFT_REPO_ROOT:-.}/gpurun_out/r03e
mkdir -p $out
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  env $line timeout -k 10 300 ./ipcr_amd/chunk_workers 125000000 4000000 ${WORKERS:-8 16} > $out/cw$i.json 2> $out/cw$i.err || { echo "run $i failed"; tail -5 $out/cw$i.err; exit 1; }
  python3 -c "import json; d=json.load(open('$out/cw$i.json')); print('$line', {k: v for k, v in d.items() if k.startswith('gbases')})"
done <<< "${KNOBS:-A=0}"
