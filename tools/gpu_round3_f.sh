#!/bin/bash
# round 3, batch F (GPU box): FASTA -> tiles -> scan -> TSV under loader knobs (one "VAR=value ..." line per run in $KNOBS)
set -o pipefail
out=${GRAFT_REPO_ROOT:-.}/gpurun_out/r03f
mkdir -p $out
export TMPDIR=/tmp
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  env $line IPCR_DEBUG_TIMES=1 timeout -k 10 400 python3 tools/e2e_fasta.py --records ${RECORDS:-8} > $out/e2e$i.json 2> $out/e2e$i.err || { echo "run $i failed"; tail -5 $out/e2e$i.err; exit 1; }
  python3 -c "import json; d=json.load(open('$out/e2e$i.json')); print('$line', d['warm'])"
  grep "fasta loader" $out/e2e$i.err | tail -1
done <<< "${KNOBS:-A=0}"
