#!/bin/bash
# round 3, batch N (GPU box): the FASTA loader from the native tool, knobs one per line in $KNOBS
set -o pipefail
cd ${GRAFT_REPO_ROOT:-.}
f=/tmp/ipcr_m.fa
python3 tools/fasta_load.py make $f || exit 1
cat $f > /dev/null
while read -r line; do
  [ -z "$line" ] && continue
  echo "== $line"
  env $line IPCR_DEBUG_TIMES=${DEBUG_TIMES:-1} timeout -k 10 120 tools/ubench/fasta_load $f 3 > /tmp/o.txt 2>&1; grep "native load\|copy of slab  [5-9]\|rror\|fasta loader" /tmp/o.txt | tail -${TAIL:-4}
done <<< "${KNOBS:-A=0}"
