#!/bin/bash
# round 3, batch D (GPU box): the chunk path with host packing forced on (parity), then the native worker pool
set -o pipefail
out=${GRAFT_REPO_ROOT:-.}/gpurun_out/r03d
mkdir -p $out
if [ -z "$SKIP_TESTS" ]; then
IPCR_CHUNK_HOSTPACK=1 timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fasta.py -m gpu -x -q -k "not fullsize" > $out/tests.log 2>&1 || { tail -30 $out/tests.log; exit 1; }
tail -2 $out/tests.log
fi
for hp in 1 0; do
  IPCR_CHUNK_HOSTPACK=$hp timeout -k 10 300 ./ipcr_amd/chunk_workers 125000000 4000000 1 8 16 > $out/cw_hp$hp.json 2> $out/cw_hp$hp.err || { echo "chunk_workers hp=$hp failed"; tail -5 $out/cw_hp$hp.err; exit 1; }
  python3 -c "import json; d=json.load(open('$out/cw_hp$hp.json')); print('hostpack=$hp', {k: v for k, v in d.items() if k.startswith('gbases') or k.startswith('pinned')})"
done
timeout -k 10 300 ./ipcr_amd/chunk_workers 125000000 4000000 8 16 24 > $out/cw_auto.json 2> $out/cw_auto.err && python3 -c "import json; d=json.load(open('$out/cw_auto.json')); print('auto', {k: v for k, v in d.items() if k.startswith('gbases') or k.startswith('call_ms')})"
nproc
