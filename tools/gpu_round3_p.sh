#!/bin/bash
# round 3, batch P (GPU box): threads on the device's own CPUs -- FASTA loader (native, and end to end from Python), chunk worker pool
set -o pipefail
cd ${GRAFT_REPO_ROOT:-.}
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q -k 'fasta or Fasta or bind_thread or chunk' 2>&1 | tail -3 || exit 1
f=/tmp/ipcr_m.fa
python3 tools/fasta_load.py make $f || exit 1
cat $f > /dev/null
for k in A=0 IPCR_BIND_THREADS=0; do
  echo "== loader $k"
  env $k IPCR_DEBUG_TIMES=1 timeout -k 10 120 tools/ubench/fasta_load $f 4 > /tmp/o.txt 2>&1; grep "native load" /tmp/o.txt | tail -3
done
echo "== e2e"
timeout -k 10 300 python3 tools/e2e_fasta.py --records 8 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.read().splitlines()[-1])['warm'])"
echo "== chunk workers, bound"
GPU_MAX_HW_QUEUES=8 timeout -k 10 300 ipcr_amd/chunk_workers 125000000 4000000 8 16 | python3 -c "import sys,json; d=json.loads(sys.stdin.read().splitlines()[-1]); print({k:v for k,v in d.items() if k.startswith('gbases') or k.startswith('workers')})"
echo "== chunk workers, --no-bind"
GPU_MAX_HW_QUEUES=8 timeout -k 10 300 ipcr_amd/chunk_workers --no-bind 125000000 4000000 8 16 | python3 -c "import sys,json; d=json.loads(sys.stdin.read().splitlines()[-1]); print({k:v for k,v in d.items() if k.startswith('gbases') or k.startswith('workers')})"
