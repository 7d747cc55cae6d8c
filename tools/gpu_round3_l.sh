#!/bin/bash
# round 3, batch L (GPU box): FASTA loader knobs (one "VAR=value ..." line per run in $KNOBS), stage times on stderr
set -o pipefail
out=${GRAFT_REPO_ROOT:-.}/gpurun_out/r03l
mkdir -p $out
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  env $line IPCR_DEBUG_TIMES=${DEBUG_TIMES:-1} timeout -k 10 300 python3 tools/e2e_fasta.py --records 8 > $out/run$i.json 2> $out/run$i.err || { tail -5 $out/run$i.err; exit 1; }
  echo "$line -> $(python3 -c "import json; d=json.load(open('$out/run$i.json')); print(d['warm'])")"
  grep "fasta loader" $out/run$i.err | tail -1
done <<< "${KNOBS:-A=0}"
